// Sparse paged single-token decode attention over the top-K selected pages + the current page,
// and the handler that plans its split over the chip.
//
// Reference behaviour restated (not translated):
//   kernel   BatchDecodeWithPagedKVCacheKernel        kernels/include/decode/decode_attn.cuh:440-646
//   indexing protective_get_k_ptr_heads              kernels/include/decode/decode_page.cuh:325-351
//   planner  BatchDecodeWithPagedKVCacheWorkEstimation / PartitionPagedKVCacheComputeAuxiliaryInfo
//                                                      decode_attn.cuh:675-893
//   merge    VariableLengthMergeStates (flashinfer cascade.cuh, un-vendored) call :992-1001
//   handler  BatchDecodeHandler                        kernels/include/decode/decode_handler.cuh:39-244
//
// gfx950 design.  The op is a gather of (page, head) K/V tiles -- 16 rows x 256 B each for
// D=128 -- and a batch-1 vector-matrix product: HBM-bound, no MFMA.  Per head the page list
// (n_sel selected pages + the current page as one extra slot, so no pseudo-batch) is cut into
// chunks of `pages_per_chunk`; one 4-wave workgroup per (chunk, head).  A wave owns one page at a
// time: each 16 B-per-lane load instruction fetches 4 token rows (1 KiB), all 8 loads of a page
// (4 K + 4 V) are issued before any is consumed and the next page's loads are issued before the
// current page is reduced (register double buffer, no LDS round trip -- K/V are used once).
// The 16 lanes of a row reduce q.k with a DPP rotation tree (row_allreduce_sum_fast); every row keeps its own online-softmax
// state (m, d, acc[8]/lane), rows merge by shuffles, waves through LDS, chunks by a small second
// kernel that also normalises and casts (VariableLengthMergeStates' job).
#include <cstdlib>
#include <new>
#include <vector>

#include "append_device.cuh"
#include "decode_device.cuh"
#include "layer_device.cuh"

namespace quest {

// Workgroups go to the 8 XCDs round-robin by linear workgroup id, and each XCD has its own L2.  With GQA the query
// heads of a kv-head group select overlapping page sets (measured: 21 % of their pages are shared, scripts/gqa_overlap.py),
// so their workgroups should sit on the SAME XCD, at the same time: the host picks `xcd_period` such that grid rows
// y, y + period, y + 2 period, ... (same XCD for every chunk index) serve a run of consecutive query heads.
// (head_dim 256 keeps twice the K/V registers in flight per lane: its 8-wave instantiations are built for 2 waves per SIMD
// -- up to 256 VGPRs, one workgroup per CU -- instead of spilling 41-46 registers to scratch at the 128 of 4 waves per SIMD)
template <int D, int S_T, int FC, int NW, int VF = -1>
__global__ __launch_bounds__(NW* kWave, (D >= 256 && NW >= 8) ? NW / 4 : NW / 2) void sparse_decode_kernel(QUEST_DECODE_HEAD_PARAMS, DecodeParams p) {
    QUEST_DECODE_HEAD_TAKE(p);
    uint32_t hq = blockIdx.y;
    const uint32_t period = p.xcd_period & 255u, slow = (p.xcd_period >> 8) & 15u, group_l2 = p.xcd_period >> 12;
    if (period > 1) {
        hq = (hq % period) * (a_num_qo_heads / period) + hq / period;
    } else if (slow) {
        // slow-class kv heads first (plan_decode): the first quarter of the rows serves the kv heads = slow - 1 mod 4 (with
        // all query heads of their groups), the rest follow in order
        const uint32_t sr = slow - 1u, unit = hq >> group_l2, quarter = a_num_qo_heads >> (2u + group_l2);
        uint32_t hk;
        if (unit < quarter) {
            hk = 4u * unit + sr;
        } else {
            const uint32_t r = unit - quarter;
            hk = 4u * (r / 3u) + ((sr + 1u + r % 3u) & 3u);
        }
        hq = (hk << group_l2) + (hq & ((1u << group_l2) - 1u));
    }
    p.xcd_period = period;
    if constexpr (VF == 8) sparse_decode_tiles_body<D, NW>(p, blockIdx.x, hq, blockIdx.z, a_num_qo_heads);
    else sparse_decode_body<D, S_T, FC, NW, VF>(p, blockIdx.x, hq, blockIdx.z, a_num_qo_heads);
}

// One launch per layer of a batched step: the workgroup of a (sequence, query head) appends, scores its head's pages into
// LDS, selects from LDS and gathers (layer_device.cuh).  grid (1, Hq, n_seqs) -- the per-head-list kernel's, so the same
// XCD-aware row order applies.
template <int D, int FC, int NW>
__global__ __launch_bounds__(NW* kWave, NW / 2) void layer_decode_kernel(QUEST_LAYER_HEAD_PARAMS, LayerParams p) {
    layer_decode_body<D, FC, NW>(a_q, a_state, a_meta, a_meta_tables, a_kv_tables, a_n_cap, a_meta_table_stride,
                                 a_kv_table_stride, a_pack, a_num_qo_heads, p);
}

// ------------------------------------------------------------------------------------------------
// Group-shared variant: all GS query heads of a kv head attend the SAME page list (full-KV decode: the
// model's dense first layers, contexts shorter than the budget).  One workgroup per (chunk, kv head)
// fetches each K/V tile once and folds it into GS online-softmax states, instead of once per query head
// as the per-head-list kernel above would (4x the traffic at Llama-3 GQA).  Same work split, same partial
// records, same merge kernel.
//
// APPEND (round 4): the decode append rides in the same launch.  The reference issues append_kv_cache_decode and the
// attention as two calls (QuestAttention.py:106 then :125-132, page.cu:6-99); here a dense layer of a captured step was
// three launches (append, attention, merge).  The new token's row of the current page is touched by exactly ONE
// workgroup per kv head -- the one whose chunk holds the last slot -- so that workgroup takes the row from the inputs
// instead of from the pool, writes it to the pool and folds the key into the page's (max, min) metadata entry
// (append_decode_body's arithmetic, same bits); nobody else reads or writes those bytes during the launch.
template <int D, int GS, int NW, bool APPEND = false>
__global__ __launch_bounds__(NW* kWave, NW / 2) void shared_decode_kernel(QUEST_DECODE_HEAD_PARAMS, DecodeParams p) {
    // MHA (one query head per kv head: registers to spare): the next page's rows are requested before the current page is
    // folded.  cfg 2: 14.35 -> 14.11 us per launch, 16.40 -> 16.20 per layer; the 32K dense launch 85.0 -> 84.4 us
    // (profiles/r05_ab_dense_kernel_prefetch.txt).
    constexpr bool PREFETCH = GS == 1;
    QUEST_DECODE_HEAD_TAKE(p);
    constexpr int LPR = D / kVec, R = kWave / LPR, S_T = 16, T = (S_T + R - 1) / R;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = lane / LPR, col = lane % LPR;
    // grid order (launch_shared): (chunks, kv heads) or, with the NHD pool, (kv heads, chunks) -- p.xcd_period != 0, the
    // preloaded field the per-head kernel uses for its own grid order -- so that workgroups adjacent in dispatch order read
    // the 256-byte pieces of the SAME token rows next to each other (a page's rows are 8 KiB of heads side by side)
    const bool heads_first = p.xcd_period != 0;
    const uint32_t chunk = heads_first ? blockIdx.y : blockIdx.x, hk = heads_first ? blockIdx.x : blockIdx.y;
    const SeqView sv = select_sequence(p, a_num_qo_heads, D, blockIdx.z);
    if (p.state) {  // state-driven launch: the plan (chunks) was made for the pool capacity; workgroups whose
                    // chunk lies past the live page list write an empty partial (weight 0 in the merge)
        const quest_step_state_t st = *sv.state;
        p.n_sel = (uint32_t)(st.n_pages - 1);
        p.last_page_len = (uint32_t)st.kv_last_page_len;
        p.last_page_idx = st.kv_last_page_idx;
        if constexpr (APPEND) {
            p.meta_last_page_len = (uint32_t)st.meta_last_page_len;
            p.meta_last_page_idx = st.meta_last_page_idx;
        }
    }
    const uint32_t n_slots = p.n_sel + 1;
    const uint32_t slot_begin = min(n_slots, chunk * p.pages_per_chunk);
    const uint32_t slot_end = min(n_slots, slot_begin + p.pages_per_chunk);

    float8 qv[GS];
#pragma unroll
    for (int g = 0; g < GS; ++g) {
        qv[g] = to_f32(ld8(sv.q + ((size_t)hk * GS + g) * D + col * kVec));
        qv[g] *= p.scale_log2;
    }
    // page base + uni[t] + lane_off (+ v_off): the walk of quest_common.cuh (walk_*), the same for every pool layout
    const uint32_t lane_off = walk_lane_off(p.st, hk, R, row, col * kVec);
    const uint32_t v_off = pool_v_off(p.st, hk);
    uint32_t uni[T];
#pragma unroll
    for (int t = 0; t < T; ++t) uni[t] = walk_uniform(p.st, hk, R, t);
    RowState<D> st[GS];

    [[maybe_unused]] const size_t app_in_off = ((size_t)blockIdx.z * (a_num_qo_heads / GS) + hk) * D + col * kVec;
    // APPEND: the lanes (of the ONE wave per kv head that attends the current page) that hold the new token's row.  Their
    // metadata loads are issued here and consumed after the page loop; the stores come after the loop as well -- issued in
    // front of it they sat in the wave's vmcnt queue ahead of its K/V loads, whose first wait then also waited for the
    // write acknowledgements (measured: the fused launch lost its gain whenever the last chunk was long).
    [[maybe_unused]] bool app_mine = false;
    [[maybe_unused]] uint16_t* app_mmax = nullptr;
    [[maybe_unused]] ushort8 app_mx = (ushort8)(kHalfNegMax), app_mn = (ushort8)(kHalfMax);
    if constexpr (APPEND) {
        const uint32_t e = p.last_page_len - 1u;
        app_mine = slot_begin <= p.n_sel && p.n_sel < slot_end && (uint32_t)wave == (p.n_sel - slot_begin) % NW &&
                   (uint32_t)row == e % R;
        if (app_mine) {
            // metadata entry of the page: entry meta_last_page_len - 1 of the last metadata page (same pool geometry as
            // the KV pool); a token that opens a page starts from the sentinels, not from stale pool bytes
            const uint32_t me = p.meta_last_page_len - 1u;
            app_mmax = reinterpret_cast<uint16_t*>(p.app_meta) + (size_t)p.meta_last_page_idx * p.st.page +
                       (size_t)pool_slot(p.st, hk, me) * p.st.head + (size_t)me * p.st.entry + col * kVec;
            if (e > 0) {
                app_mx = *reinterpret_cast<const ushort8*>(app_mmax);
                app_mn = *reinterpret_cast<const ushort8*>(app_mmax + v_off);
            }
        }
    }

    // request the K/V rows of slot s0
    auto issue = [&](uint32_t s0, half8 (&k)[T], half8 (&v)[T]) {
        const int32_t pg = __builtin_amdgcn_readfirstlane(s0 < p.n_sel ? sv.indices[s0] : p.last_page_idx);
        const half_t* b0 = p.kv + (size_t)pg * p.st.page;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            k[t] = ld8_stream(b0 + uni[t] + lane_off);
            v[t] = ld8_stream(b0 + uni[t] + lane_off + v_off);
        }
    };
    // fold the rows of slot s0 into the GS states
    auto fold = [&](uint32_t s0, half8 (&k)[T], half8 (&v)[T]) {
        const int len = s0 < p.n_sel ? S_T : (int)p.last_page_len;
        if constexpr (APPEND) {
            if (s0 >= p.n_sel) {  // wave-uniform: the current page -- its newest row is the token being decoded, which is
                                  // not in the pool yet (written after this loop): it comes from the inputs
                const uint32_t e = (uint32_t)len - 1u;
                const half8 kn = ld8(p.app_k + app_in_off), vn = ld8(p.app_v + app_in_off);
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if ((uint32_t)(t * R + row) == e) {
                        k[t] = kn;
                        v[t] = vn;
                    }
            }
        }
        float8 kf[T], vf[T];
        bool valid[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            valid[t] = row < len - t * R;
            kf[t] = to_f32(k[t]);
            vf[t] = valid[t] ? to_f32(v[t]) : (float8)(0.f);  // stale rows past the page length: select
        }
#pragma unroll
        for (int g = 0; g < GS; ++g) {
            float sc[T];
            float m_new = st[g].m;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float dot = 0.f;
#pragma unroll
                for (int i = 0; i < kVec; ++i) dot = __builtin_fmaf(qv[g][i], kf[t][i], dot);
                dot = row_allreduce_sum_fast<LPR>(dot);
                sc[t] = valid[t] ? dot : kNegFloor;
                m_new = __builtin_fmaxf(m_new, sc[t]);
            }
            const float scale = __builtin_amdgcn_exp2f(st[g].m - m_new);
            st[g].d *= scale;
            st[g].acc *= scale;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float pr = valid[t] ? __builtin_amdgcn_exp2f(sc[t] - m_new) : 0.f;
                st[g].d += pr;
#pragma unroll
                for (int i = 0; i < kVec; ++i) st[g].acc[i] = __builtin_fmaf(pr, vf[t][i], st[g].acc[i]);
            }
            st[g].m = m_new;
        }
    };
    if constexpr (PREFETCH) {
        // the next page's rows are requested before the current page is folded (two register sets, 16 loads in flight per
        // lane); same pages in the same order per wave, hence the same bits.  Straight-line issue -> fold segments: a load
        // under a branch would be waited for at the join
        uint32_t s0 = slot_begin + wave;
        if (s0 < slot_end) {
            half8 k0[T], v0[T], k1[T], v1[T];
            issue(s0, k0, v0);
            while (true) {
                uint32_t s1 = s0 + NW;
                if (s1 >= slot_end) {
                    fold(s0, k0, v0);
                    break;
                }
                issue(s1, k1, v1);
                fold(s0, k0, v0);
                s0 = s1 + NW;
                if (s0 >= slot_end) {
                    fold(s1, k1, v1);
                    break;
                }
                issue(s0, k0, v0);
                fold(s1, k1, v1);
            }
        }
    } else {
        for (uint32_t s0 = slot_begin + wave; s0 < slot_end; s0 += NW) {
            half8 k[T], v[T];
            issue(s0, k, v);
            fold(s0, k, v);
        }
    }

    if constexpr (APPEND) {
        if (app_mine) {  // write the new token to the pool and fold its key into the page's (max, min) entry
            const uint32_t e = p.last_page_len - 1u;
            const half8 kn = ld8(p.app_k + app_in_off), vn = ld8(p.app_v + app_in_off);
            half_t* dst = const_cast<half_t*>(p.kv) + (size_t)p.last_page_idx * p.st.page + lane_off + walk_uniform(p.st, hk, R, e / R);
            st8(dst, kn);
            st8(dst + v_off, vn);
            const ushort8 k8 = __builtin_bit_cast(ushort8, kn);
            *reinterpret_cast<ushort8*>(app_mmax) = fold_max(app_mx, k8);
            *reinterpret_cast<ushort8*>(app_mmax + v_off) = fold_min(app_mn, k8);
        }
    }

    __shared__ float s_acc[NW][GS][D];
    __shared__ float s_md[NW][GS][2];
#pragma unroll
    for (int g = 0; g < GS; ++g) {
        for_each_row_distance<LPR>([&](auto off_c) {  // rows of the wave -> one state
            constexpr int OFF = decltype(off_c)::value;
            const float m_o = lane_xor<OFF>(st[g].m, lane), d_o = lane_xor<OFF>(st[g].d, lane);
            const float m_n = __builtin_fmaxf(st[g].m, m_o);
            const float a = __builtin_amdgcn_exp2f(st[g].m - m_n), b = __builtin_amdgcn_exp2f(m_o - m_n);
            st[g].d = st[g].d * a + d_o * b;
#pragma unroll
            for (int i = 0; i < kVec; ++i) st[g].acc[i] = st[g].acc[i] * a + lane_xor<OFF>(st[g].acc[i], lane) * b;
            st[g].m = m_n;
        });
        if (row == 0) {
#pragma unroll
            for (int i = 0; i < kVec; ++i) s_acc[wave][g][col * kVec + i] = st[g].acc[i];
            if (col == 0) {
                s_md[wave][g][0] = st[g].m;
                s_md[wave][g][1] = st[g].d;
            }
        }
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < (uint32_t)(GS * D); t += NW * kWave) {
        const uint32_t g = t / D, f = t % D, hq = hk * GS + g;
        float M = s_md[0][g][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = __builtin_fmaxf(M, s_md[w][g][0]);
        float acc = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float e = __builtin_amdgcn_exp2f(s_md[w][g][0] - M);
            acc += e * s_acc[w][g][f];
            den += e * s_md[w][g][1];
        }
        if (p.n_chunks == 1) {
            sv.o[(size_t)hq * D + f] = (half_t)(acc / den);
            if (p.lse && f == 0) sv.lse[hq] = (M + __builtin_amdgcn_logf(den)) * 0.6931471805599453f;
        } else {
            float* w = sv.ws + ((size_t)hq * p.n_chunks + chunk) * p.ws_stride;
            w[f] = acc;
            if (f == 0) {
                w[D] = M;
                w[D + 1] = den;
            }
        }
    }
}

// Merge the per-chunk partial states of a head, normalise, cast to fp16 (the job of flashinfer's
// VariableLengthMergeStates).  kMergeGroups thread groups each take every kMergeGroups-th chunk with all of
// their loads independent (unrolled), then combine through LDS: two memory round trips total.
// Merging INSIDE the attention launch (the workgroup of a head that draws the last arrival ticket merges; partial
// records leave by write-through stores) was built twice -- round 1, and round 3 with the recipe of VERDICT r2 item 1a
// -- bit-identical and slower: every workgroup's tail gains a store acknowledgement and an atomic round trip that
// queue behind the launch's own streaming reads.  Same box, us per launch incl. merge: cfg 3 14.7 (this launch) vs
// 22.4 (in-kernel), cfg 2 17.7 vs 20.6, cfg 4 23.8 vs 26.9 (gpurun_out/r3a_*).
template <int D>
__global__ __launch_bounds__(D* kMergeGroups) void merge_states_kernel(const float* __restrict__ ws,
                                                                        half_t* __restrict__ o,
                                                                        float* __restrict__ lse, uint32_t n_chunks,
                                                                        uint32_t ws_stride, StepAdvance adv) {
    __shared__ float s_w[1024];  // per-chunk weight exp2(m_c - M); planner keeps n_chunks <= 1024
    __shared__ float s_red[kMergeGroups][D + 1];
    __shared__ float s_M;
    const uint32_t hq = blockIdx.x, tid = threadIdx.x;
    const uint32_t f = tid % D, g = tid / D;
    const float* w = ws + (size_t)hq * n_chunks * ws_stride;
    // the next step's reservation (StepAdvance, append_device.cuh): nothing in or after this launch reads the state
    if (adv.st && hq == 0 && tid < adv.batch.n_seqs)
        step_state_advance_one(adv.st + tid, adv.kv_table + (size_t)tid * adv.batch.kv_table_stride,
                               adv.meta_table + (size_t)tid * adv.batch.meta_table_stride, adv.page_size, adv.max_kv_pages,
                               adv.max_meta_pages);
    if (n_chunks <= kMergeFastChunks) {  // launch-uniform
        merge_head_fast<D, D * kMergeGroups>(w, o + (size_t)hq * D, lse ? lse + hq : nullptr, n_chunks, ws_stride, tid,
                                                    &s_red[0][0]);
        return;
    }
    // the first kPre partial rows of this thread's chunks are requested up front
    constexpr int kPre = kMergePre;
    float pre[kPre];
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
        const uint32_t c = g + j * kMergeGroups, cc = c < n_chunks ? c : n_chunks - 1;  // clamped: no branch
        pre[j] = w[(size_t)cc * ws_stride + f];
    }
    // pass 1: chunk maxima -> M, weights, denominator
    float M = kNegFloor;
    for (uint32_t c = tid; c < n_chunks; c += blockDim.x) M = __builtin_fmaxf(M, w[(size_t)c * ws_stride + D]);
    M = wave_allreduce_max(M, (int)(tid & 63));
    if ((tid & 63) == 0) s_red[0][tid >> 6] = M;
    __syncthreads();
    if (tid == 0) {
        float mm = s_red[0][0];
        for (uint32_t i = 1; i < (blockDim.x >> 6); ++i) mm = __builtin_fmaxf(mm, s_red[0][i]);
        s_M = mm;
    }
    __syncthreads();
    M = s_M;
    float den = 0.f;
    for (uint32_t c = tid; c < n_chunks; c += blockDim.x) {
        const float e = __builtin_amdgcn_exp2f(w[(size_t)c * ws_stride + D] - M);
        s_w[c] = e;
        den += e * w[(size_t)c * ws_stride + D + 1];
    }
    __syncthreads();
    // pass 2: weighted sum of the partial outputs
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < kPre; ++j) {
        const uint32_t c = g + j * kMergeGroups;
        if (c < n_chunks) acc += s_w[c] * pre[j];
    }
#pragma unroll 8
    for (uint32_t c = g + kPre * kMergeGroups; c < n_chunks; c += kMergeGroups)
        acc += s_w[c] * w[(size_t)c * ws_stride + f];
    den = wave_allreduce_sum(den, (int)(tid & 63));
    s_red[g][f] = acc;
    __shared__ float s_den[16];
    if ((tid & 63) == 0) s_den[tid >> 6] = den;
    __syncthreads();
    if (g == 0) {
        float a = s_red[0][f];
#pragma unroll
        for (int j = 1; j < kMergeGroups; ++j) a += s_red[j][f];
        float dn = 0.f;
        for (uint32_t i = 0; i < (blockDim.x >> 6); ++i) dn += s_den[i];
        o[(size_t)hq * D + f] = (half_t)(a / dn);
        if (lse && f == 0) lse[hq] = (M + __builtin_amdgcn_logf(dn)) * 0.6931471805599453f;
    }
}

}  // namespace quest

using namespace quest;

namespace quest {
int check_pool(const quest_paged_kv_t& p);  // append.hip
}

struct quest_decode_handler {
    uint32_t layout = 0;
    bool started = false;
    uint32_t n_sel = 0, num_qo_heads = 0, num_kv_heads = 0, head_dim = 0, page_size = 0;
    uint32_t pages_per_chunk = 0, n_chunks = 0;
    uint32_t forced_ppc = 0;
    float* ws = nullptr;
    size_t ws_bytes = 0;
    uint32_t ws_stride = 0;
    // Workspaces outgrown by a later plan.  Launches captured in a hipGraph hold the workspace pointer BY VALUE
    // (DecodeParams.ws), so a buffer that any launch was issued on may still be written by a replay: it is
    // retired, never freed before the handler itself is destroyed.  (A few hundred KiB each.)
    std::vector<float*> retired_ws;
    uint32_t dec_waves = 4;
    uint32_t shared_ppc = 0, shared_chunks = 0;  // plan of the group-shared kernel (grid.y = kv heads)
    uint32_t batch = 1;                          // sequences per launch the plan / workspace are made for
    uint32_t num_cus = 256;                      // compute units of the current device (MI355X: 256)
    bool skip_merge = false;                     // measurement aid: leave the partial states unmerged
    int front_end = 0;                           // fused top-k front end: 0 = by row length, 1 / 2 = forced generation, 3 = 2 + pre-filter
    uint32_t last_launch[6] = {0, 0, 0, 0, 0, 0};  // quest_decode_last_launch_info
    void* sel_val_out = nullptr;                 // inspection aid (quest_decode_set_selection_out)
    int32_t* sel_idx_out = nullptr;
    StepAdvance armed_advance = {};              // quest_decode_arm_step_advance: rides in the next merge launch, once
};

// the armed reservation, handed to the merge launch that is about to be issued (and disarmed)
static StepAdvance take_armed_advance(quest_decode_handler* h) {
    const StepAdvance adv = h->armed_advance;
    h->armed_advance = StepAdvance{};
    return adv;
}
// a launch that turned out to have no merge launch (one workgroup per head, merge skipped): the reservation still happens
// behind this layer's kernels, as its own launch
static int flush_armed_advance(quest_decode_handler* h, hipStream_t s) {
    if (!h->armed_advance.st) return 0;
    const StepAdvance a = take_armed_advance(h);
    if (a.batch.n_seqs == 1)  // (one sequence: its tables are plain arrays, no strides to check)
        return quest_step_state_advance(a.st, a.kv_table, a.meta_table, a.page_size, a.max_kv_pages, a.max_meta_pages,
                                        (quest_stream_t)s);
    return quest_step_state_advance_batched(a.st, a.kv_table, a.meta_table, a.page_size, a.max_kv_pages, a.max_meta_pages,
                                            a.batch, (quest_stream_t)s);
}

// Workgroups the planner aims for.  One sequence: the kernel is built for 2 workgroups (8 waves) per CU,
// so 2 x CUs workgroups (512) are one fully resident round, each wave with 16 x 1 KiB loads in flight.
// A batch: ONE workgroup per CU (256) -- measured at cfg-3 shapes for 2/4/8/16 sequences:
// every extra workgroup of a head repeats the fused selection and adds a partial record to merge, and
// with 8 x 32 (sequence, head) pairs the split disappears altogether (one workgroup per head, no merge
// launch).
static uint32_t target_workgroups(const quest_decode_handler* h) { return h->batch > 1 ? h->num_cus : 2 * h->num_cus; }
static constexpr uint32_t kMaxChunks = 1024;  // merge kernel's LDS weight table
extern "C" int quest_decode_arm_step_advance(quest_decode_handler_t* h, quest_step_state_t* state, const int32_t* kv_tables,
                                             const int32_t* meta_tables, uint32_t page_size, uint32_t max_kv_pages,
                                             uint32_t max_meta_pages, quest_batch_t batch) {
    if (!h) return QUEST_EINVAL;
    if (!state) {  // disarm
        h->armed_advance = StepAdvance{};
        return 0;
    }
    if (!kv_tables || !meta_tables || page_size == 0 || max_kv_pages == 0 || max_meta_pages == 0 || batch.n_seqs == 0)
        return QUEST_EINVAL;
    if (batch.n_seqs > 1 && (batch.kv_table_stride < max_kv_pages || batch.meta_table_stride < max_meta_pages)) return QUEST_EINVAL;
    if (!h->started) return QUEST_ESTATE;
    // the reservation rides in the next MERGE launch (one thread per sequence of its first workgroup); a forward whose plan
    // has no merge launch issues it as its own launch behind the attention kernel instead (correct, nothing saved)
    if (batch.n_seqs > 64) return QUEST_EUNSUPPORTED;
    h->armed_advance = StepAdvance{state, kv_tables, meta_tables, page_size, max_kv_pages, max_meta_pages, batch};
    return 0;
}

extern "C" int quest_decode_handler_create(quest_decode_handler_t** out, uint32_t layout) {
    if (!out || layout > QUEST_LAYOUT_NHD_ROT) return QUEST_EINVAL;
    quest_decode_handler* h = new (std::nothrow) quest_decode_handler();
    if (!h) return (int)hipErrorOutOfMemory;
    h->layout = layout;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        h->num_cus = (uint32_t)cus;
    else
        (void)hipGetLastError();  // no device (CPU-only import): keep the MI355X default
    if (const char* e = quest_tuning_env("QUEST_DEC_WAVES")) h->dec_waves = atoi(e) == 8 ? 8 : 4;  // tuning knob
    *out = h;
    return 0;
}

extern "C" void quest_decode_handler_destroy(quest_decode_handler_t* h) {
    if (!h) return;
    if (h->ws) (void)hipFree(h->ws);
    for (float* w : h->retired_ws) (void)hipFree(w);
    delete h;
}

extern "C" int quest_decode_set_pages_per_chunk(quest_decode_handler_t* h, uint32_t ppc) {
    if (!h) return QUEST_EINVAL;
    h->forced_ppc = ppc;
    return 0;
}

extern "C" int quest_decode_set_skip_merge(quest_decode_handler_t* h, int skip) {
    if (!h) return QUEST_EINVAL;
    h->skip_merge = skip != 0;
    return 0;
}

extern "C" int quest_decode_set_front_end(quest_decode_handler_t* h, int generation) {
    if (!h || generation < 0 || generation > 3) return QUEST_EINVAL;
    h->front_end = generation;
    return 0;
}

extern "C" int quest_decode_set_selection_out(quest_decode_handler_t* h, void* val_out, int32_t* idx_out) {
    if (!h) return QUEST_EINVAL;
    h->sel_val_out = val_out;
    h->sel_idx_out = idx_out;
    return 0;
}

extern "C" int quest_decode_set_batch(quest_decode_handler_t* h, uint32_t n_seqs) {
    if (!h || n_seqs == 0 || n_seqs > 65535u) return QUEST_EINVAL;  // grid.z limit
    h->batch = n_seqs;
    return 0;
}

extern "C" int quest_decode_begin_forward(quest_decode_handler_t* h, uint32_t n_selected_pages, uint32_t num_qo_heads,
                                          uint32_t num_kv_heads, uint32_t head_dim, uint32_t page_size,
                                          quest_stream_t stream) {
    (void)stream;
    if (!h || num_qo_heads == 0 || num_kv_heads == 0 || page_size == 0) return QUEST_EINVAL;
    if (num_qo_heads % num_kv_heads != 0) return QUEST_EINVAL;  // decode_attn.cuh:1045-1050
    if (head_dim != 64 && head_dim != 128 && head_dim != 256) return QUEST_EUNSUPPORTED;
    h->n_sel = n_selected_pages;
    h->num_qo_heads = num_qo_heads;
    h->num_kv_heads = num_kv_heads;
    h->head_dim = head_dim;
    h->page_size = page_size;
    const uint32_t n_slots = n_selected_pages + 1;
    uint32_t ppc;
    if (h->forced_ppc) {
        ppc = h->forced_ppc;
    } else {
        // a batch brings its own parallelism: the workgroups are spread over the sequences (fewer, longer
        // chunks per head -> fewer partials and fewer repeats of the fused selection)
        uint32_t chunks = target_workgroups(h) / (num_qo_heads * h->batch);
        if (chunks < 1) chunks = 1;
        if (chunks > n_slots) chunks = n_slots;
        ppc = (n_slots + chunks - 1) / chunks;
        // keep the fused front end usable for every plan (it stages one chunk's page ids in LDS): eager callers
        // can fall back to two launches, the state-driven / batched entries cannot
        if (ppc > (uint32_t)kFusedMaxPpc) ppc = kFusedMaxPpc;
    }
    if ((n_slots + ppc - 1) / ppc > kMaxChunks) ppc = (n_slots + kMaxChunks - 1) / kMaxChunks;
    h->pages_per_chunk = ppc;
    h->n_chunks = (n_slots + ppc - 1) / ppc;
    {   // the group-shared kernel runs one workgroup per (chunk, KV head): plan it for the same 512 workgroups
        uint32_t chunks = 2 * h->num_cus / (num_kv_heads * h->batch);
        if (chunks < 1) chunks = 1;
        if (chunks > n_slots) chunks = n_slots;
        if (chunks > kMaxChunks) chunks = kMaxChunks;
        uint32_t sp = h->forced_ppc ? h->forced_ppc : (n_slots + chunks - 1) / chunks;
        if ((n_slots + sp - 1) / sp > kMaxChunks) sp = (n_slots + kMaxChunks - 1) / kMaxChunks;
        h->shared_ppc = sp;
        h->shared_chunks = (n_slots + sp - 1) / sp;
    }
    h->ws_stride = (head_dim + 2 + 31) / 32 * 32;
    const uint32_t max_chunks = h->n_chunks > h->shared_chunks ? h->n_chunks : h->shared_chunks;
    const size_t need = (size_t)h->batch * num_qo_heads * max_chunks * h->ws_stride * sizeof(float);
#ifdef QUEST_WALLSTAMPS
    const bool want_ws = true;  // one-chunk plans leave their per-workgroup stamps in the workspace as well
#else
    const bool want_ws = max_chunks > 1;
#endif
    if (want_ws && need > h->ws_bytes) {  // grow-only; reused across begin/end cycles
        float* bigger = nullptr;
        hipError_t e = hipMalloc((void**)&bigger, need);
        if (e != hipSuccess) return (int)e;
        if (h->ws) {
            try {
                h->retired_ws.push_back(h->ws);  // a captured graph may still reference it (see retired_ws)
            } catch (...) {
                (void)hipFree(bigger);
                return (int)hipErrorOutOfMemory;
            }
        }
        h->ws = bigger;
        h->ws_bytes = need;
    }
    h->started = true;
    return 0;
}

extern "C" int quest_decode_end_forward(quest_decode_handler_t* h) {
    if (!h) return QUEST_EINVAL;
    h->started = false;  // workspace is kept for the next begin_forward (freed in destroy)
    return 0;
}

extern "C" int quest_decode_plan_info(const quest_decode_handler_t* h, uint32_t* pages_per_chunk,
                                      uint32_t* chunks_per_head) {
    if (!h || !h->started) return QUEST_ESTATE;
    if (pages_per_chunk) *pages_per_chunk = h->pages_per_chunk;
    if (chunks_per_head) *chunks_per_head = h->n_chunks;
    return 0;
}

extern "C" int quest_decode_debug_workspace(const quest_decode_handler_t* h, void** ptr, uint64_t* bytes, uint32_t* record_floats) {
    if (!h || !ptr || !bytes) return QUEST_EINVAL;
    *ptr = h->ws;
    *bytes = h->ws_bytes;
    if (record_floats) *record_floats = h->ws_stride;
    return 0;
}

extern "C" int quest_decode_last_launch_info(const quest_decode_handler_t* h, uint32_t info[6]) {
    if (!h || !info) return QUEST_EINVAL;
    for (int i = 0; i < 6; ++i) info[i] = h->last_launch[i];
    return 0;
}

template <int D, int FC>
static int launch_decode_fc(quest_decode_handler* h, const DecodeParams& p, uint32_t num_qo_heads, uint32_t waves,
                            hipStream_t s, uint32_t n_seqs) {
    dim3 grid(h->n_chunks, num_qo_heads, n_seqs);
    const size_t lds = FC > 0 ? (size_t)p.ids_lds_offset + (p.stage_ids ? (size_t)((p.n_scores + 4u) & ~3u) * 4 : 0) : 0;  // the table has n_scores + 1 entries, staged in granules of 4
    // what this launch is (quest_decode_last_launch_info): keys per thread, waves, front-end variant, one-variant
    // instantiation or the generic kernel, workgroups per head, sequences
    uint32_t* info = h->last_launch;
    info[0] = (uint32_t)FC, info[1] = waves, info[2] = FC > 0 ? p.vec_front : 0u, info[3] = 0u, info[4] = h->n_chunks, info[5] = n_seqs;
    // The common (keys per thread, front-end variant) pairs of 8-wave page-16 launches as their own compact
    // instantiations -- the generic kernel carries every front-end variant (27 KiB of code in round 3; the one-variant
    // kernels are 10-14 KiB) and measured 0.57 us per launch slower at the headline shape:
    //   FC 8,  variant 3: single-sequence rows <= 4096 pages, keys and page ids straight into registers (cfg 3)
    //   FC 8,  variant 1: the same rows in batched launches, staging arrays fed by vector loads (cfg 5)
    //   FC 8,  variant 8: tiles front end, rows that carry their tile maxima (cfg 4)
    //   FC 16 / 24 / 32, variant 2: second-generation front end of long rows without tile maxima
    // QUEST_FE_SPECIALIZE=0 launches the generic kernel instead (A/B).
    static const bool specialize = [] { const char* e = quest_tuning_env("QUEST_FE_SPECIALIZE"); return !e || atoi(e) != 0; }();
    if (p.vec_front == 8u) {  // tiles front end (plan_decode: page size 16, 8 waves)
        if constexpr (FC == 8) {
            info[3] = 1u;
            hipLaunchKernelGGL((sparse_decode_kernel<D, 16, 8, 8, 8>), grid, dim3(8 * kWave), 0, s, QUEST_DECODE_HEAD_ARGS(p, num_qo_heads), p);
            QUEST_LAUNCH_CHECK();
            goto merge;
        } else {
            return QUEST_EUNSUPPORTED;
        }
    }
    if (specialize && p.page_size == 16 && waves == 8) {
        constexpr int VFA = FC == 8 ? 3 : 2, VFB = FC == 8 ? 1 : 2;
        if constexpr (FC == 8 || FC == 16 || FC == 24 || FC == 32) {
            info[3] = 1u;
            if (p.vec_front == (uint32_t)VFA) {
                hipLaunchKernelGGL((sparse_decode_kernel<D, 16, FC, 8, VFA>), grid, dim3(8 * kWave), lds, s, QUEST_DECODE_HEAD_ARGS(p, num_qo_heads), p);
                QUEST_LAUNCH_CHECK();
                goto merge;
            }
            if (VFB != VFA && p.vec_front == (uint32_t)VFB) {
                hipLaunchKernelGGL((sparse_decode_kernel<D, 16, FC, 8, VFB>), grid, dim3(8 * kWave), lds, s, QUEST_DECODE_HEAD_ARGS(p, num_qo_heads), p);
                QUEST_LAUNCH_CHECK();
                goto merge;
            }
            info[3] = 0u;
        }
    }
    if (p.page_size == 16 && waves == 8)
        hipLaunchKernelGGL((sparse_decode_kernel<D, 16, FC, 8>), grid, dim3(8 * kWave), lds, s, QUEST_DECODE_HEAD_ARGS(p, num_qo_heads), p);
    else if (p.page_size == 16)
        hipLaunchKernelGGL((sparse_decode_kernel<D, 16, FC, 4>), grid, dim3(4 * kWave), lds, s, QUEST_DECODE_HEAD_ARGS(p, num_qo_heads), p);
    else
        hipLaunchKernelGGL((sparse_decode_kernel<D, 0, FC, 4>), grid, dim3(4 * kWave), lds, s, QUEST_DECODE_HEAD_ARGS(p, num_qo_heads), p);
    QUEST_LAUNCH_CHECK();
merge:
    if (h->n_chunks > 1 && !h->skip_merge) {  // o / lse / partials of a batch are contiguous over (sequence, head): one grid
        hipLaunchKernelGGL((merge_states_kernel<D>), dim3(num_qo_heads * n_seqs), dim3(D * kMergeGroups), 0, s,
                           (const float*)p.ws, p.o, QUEST_LSE_ENABLED ? p.lse : nullptr, h->n_chunks, p.ws_stride,
                           take_armed_advance(h));
        QUEST_LAUNCH_CHECK();
    }
    return flush_armed_advance(h, s);
}

// fc: fused top-k front end variant (0 = page ids come from an index tensor)
template <int D>
static int launch_decode(quest_decode_handler* h, const DecodeParams& p, uint32_t num_qo_heads, int fc,
                         uint32_t waves, hipStream_t s, uint32_t n_seqs) {
    switch (fc) {
        case 0: return launch_decode_fc<D, 0>(h, p, num_qo_heads, waves, s, n_seqs);
        case 8: return launch_decode_fc<D, 8>(h, p, num_qo_heads, waves, s, n_seqs);
        case 16: return launch_decode_fc<D, 16>(h, p, num_qo_heads, waves, s, n_seqs);
        case 24: return launch_decode_fc<D, 24>(h, p, num_qo_heads, waves, s, n_seqs);
        case 32: return launch_decode_fc<D, 32>(h, p, num_qo_heads, waves, s, n_seqs);
        case 64: return launch_decode_fc<D, 64>(h, p, num_qo_heads, waves, s, n_seqs);
        default: return QUEST_EUNSUPPORTED;
    }
}

// Checks + kernel parameters of a per-head-list decode launch; fc = keys per thread of the fused front end (0 = page ids
// come from an index tensor), waves = workgroup size.
static int plan_decode(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t kv, uint32_t num_qo_heads,
                       const void* scores, uint32_t n_scores, void* topk_val_out, int32_t* topk_idx_out, float* lse,
                       uint32_t score_stride, const quest_step_state_t* state, quest_batch_t batch, DecodeParams& p, int& fc,
                       uint32_t& waves, uint32_t tile_off = 0) {
    if (state) kv.last_page_len = 1;  // placeholder; the kernel reads the real one from `state`
    if (!h || batch.n_seqs == 0) return QUEST_EINVAL;
    if (!h->started) return QUEST_ESTATE;
    if (batch.n_seqs > h->batch) return QUEST_ESTATE;  // workspace / plan were made for fewer sequences
    if (!q || !o || !kv.data) return QUEST_EINVAL;
    const bool fused = scores != nullptr;
    if (fused) {
        if (!kv.indices || h->n_sel == 0 || h->n_sel > n_scores) return QUEST_EINVAL;
        if (n_scores > QUEST_TOPK_MAX_ROW) return QUEST_ETOOLARGE;
        if (h->pages_per_chunk > (uint32_t)kFusedMaxPpc) return QUEST_EUNSUPPORTED;
        // (rows beyond 4096 columns need the second-generation front end, i.e. aligned score rows: checked below)
    } else if (h->n_sel > 0 && (!kv.indices || (!state && kv.page_budget < h->n_sel))) {
        return QUEST_EINVAL;  // (state-driven: one shared list, row stride kv.page_budget == 0)
    }
    if (kv.layout != h->layout || kv.head_dim != h->head_dim || kv.page_size != h->page_size ||
        kv.num_heads != h->num_kv_heads || num_qo_heads != h->num_qo_heads)
        return QUEST_EINVAL;
    if (kv.last_page_len == 0 || kv.last_page_len > kv.page_size) return QUEST_EINVAL;
    p = DecodeParams{};
    p.q = (const half_t*)q;
    p.o = (half_t*)o;
    p.lse = lse;
    p.kv = (const half_t*)kv.data;
    p.indices = kv.indices;
    p.ws = h->ws;
    p.st = pool_strides(kv);
    p.idx_stride = fused ? 0 : kv.page_budget;
    p.n_sel = h->n_sel;
    p.last_page_len = kv.last_page_len;
    p.last_page_idx = kv.last_page_idx;
    p.page_size = kv.page_size;
    p.group = num_qo_heads / kv.num_heads;
    p.pages_per_chunk = h->pages_per_chunk;
    p.n_chunks = h->n_chunks;
    p.scale_log2 = (float)(1.4426950408889634 / sqrt((double)kv.head_dim));
    p.scores = (const uint16_t*)scores;
    p.n_scores = n_scores;
    p.sel_val_out = (uint16_t*)topk_val_out;
    p.sel_idx_out = topk_idx_out;
    if (fused && state && !topk_idx_out) {  // state-driven launches have no output arguments: handler-level aid
        p.sel_val_out = (uint16_t*)h->sel_val_out;
        p.sel_idx_out = h->sel_idx_out;
    }
    p.sel_stride = h->n_sel;
    p.ws_stride = h->ws_stride;
    p.score_stride = score_stride ? score_stride : n_scores;
    p.stage_ids = n_scores <= 4096 ? 1u : 0u;  // keys always staged (2 B each); ids (4 B each) up to 16 KiB
    p.ids_lds_offset = (uint32_t)((((size_t)n_scores * 2) + 15) & ~(size_t)15);
    p.vec_front = 0;
    int forced = 0;
    bool table_vec = false, rows_aligned = false;
    if (fused) {
        // second-generation front end (topk_bitmap.cuh): 8-byte loads of 4 scores straight from the row -> the rows
        // must be 8-byte aligned and readable up to the next multiple of 4 columns (the row stride covers it)
        // Measured on MI355X (state-driven launches, us per launch gen 1 -> gen 2): cfg 3 (2047 columns, 16 workgroups
        // per head) 12.7 -> 13.6, 8 x cfg 3 batched 47.9 -> 50.1, cfg 4 (8191 columns) 28.6 -> 21.9.  The second
        // generation has fewer barriers but its per-wave scans are redundant work on an issue-bound CU (4 waves per
        // SIMD), which costs more than it saves on short rows -> gen 2 from 4097 columns up (gen 1 spills there).
        // quest_decode_set_front_end / QUEST_FRONT_END=1 / 2 force a generation where it is applicable (tuning, tests).
        static const int env_forced = [] { const char* e = quest_tuning_env("QUEST_FRONT_END"); return e ? atoi(e) : 0; }();
        forced = h->front_end ? h->front_end : env_forced;
        const uint32_t stride = p.score_stride;
        const bool aligned = ((uintptr_t)scores & 7u) == 0 && stride % 4u == 0 && stride >= ((n_scores + 3u) & ~3u);
        const bool table_aligned = ((uintptr_t)kv.indices & 15u) == 0 && (batch.n_seqs == 1 || batch.kv_table_stride % 4u == 0);
        table_vec = table_aligned;
        rows_aligned = aligned;
        int gen = 1;
        bool lead_rows = false;
        if (aligned) {
            if (forced == 2 || forced == 3 || (forced == 0 && n_scores > 4096u)) gen = 2;
        } else if (n_scores > 4096u && n_scores + 3u <= QUEST_TOPK_MAX_ROW && forced != 1) {  // (the bitmaps cover positions)
            // the reference's own score layout beyond 4096 pages (contiguous [Hq][pages - 1] rows: 2-byte aligned,
            // quest/utils/__init__.py:194): the second generation on the aligned stream BELOW each row (row_lead) -- one
            // fused launch where rounds 2-4 needed topk_filtering + forward
            gen = 2;
            lead_rows = true;
        }
        if (gen != 1) {
            p.vec_front = 2;
            p.ids_lds_offset = 0;  // no key staging
            p.row_lead = lead_rows ? 1u : 0u;
            // page ids are staged with 16-byte loads: the table(s) must be 16-byte aligned (and columns = positions)
            if (!table_aligned || lead_rows) p.stage_ids = 0;
            // (gen 2 stages ids only in the instantiations with <= 16 keys per thread: rows <= 4096 columns)
            // pre-filter of the histogram: pays where a thread holds many keys (rows beyond 4096 columns); QUEST_FE2_PREFILTER=0
            // turns it off, =2 turns it on for every second-generation launch (tuning, tests)
            static const int pre_env = [] { const char* e = quest_tuning_env("QUEST_FE2_PREFILTER"); return e ? atoi(e) : 1; }();
            p.fe2_prefilter = forced == 3 || pre_env == 2 || (pre_env == 1 && forced != 2 && n_scores > 4096u) ? 1u : 0u;
        } else if (aligned && table_aligned && p.stage_ids && forced != 1) {
            p.vec_front = 1;  // generation 1 with its staging arrays filled by the granule loads (8 / 16 bytes per lane)
        }
    }
    // rows beyond 4096 columns are served by the second-generation front end only (the first one spills there):
    // state-driven callers own the score scratch and give it an 8-byte aligned row stride (a multiple of 4 columns)
    if (fused && !p.vec_front && n_scores > 4096u && !tile_off) return QUEST_EUNSUPPORTED;
    if (tile_off) {  // tiles front end (sparse_decode_tiles_body): the rows carry their tile maxima behind the scores
        const uint32_t tiles_cap = (n_scores + 7u) / 8u;
        if (!fused || tile_off != ((n_scores + 7u) & ~7u) || ((uintptr_t)scores & 15u) != 0 || p.score_stride % 8u != 0 ||
            p.score_stride < tile_off + ((tiles_cap + 3u) & ~3u))
            return QUEST_EINVAL;
        if (kv.page_size != 16 || h->n_sel > 256u || tiles_cap > 4u * 512u || batch.n_seqs != 1) return QUEST_EUNSUPPORTED;
        p.vec_front = 8;
        p.stage_ids = 0;
        p.ids_lds_offset = 0;
    }
    p.state = state;
    p.table_stride = batch.kv_table_stride;
    p.budgets = state ? batch.page_budgets : nullptr;
    {   // XCD-aware row order for GQA (see sparse_decode_kernel); QUEST_XCD_GROUP=0 keeps the plain order (tuning)
        static const bool xcd_group = [] { const char* e = quest_tuning_env("QUEST_XCD_GROUP"); return !e || atoi(e) != 0; }();
        uint32_t gcd = 8, c = h->n_chunks % 8u;
        while (c) { const uint32_t t = gcd % c; gcd = c; c = t; }
        const uint32_t period = 8u / gcd;
        p.xcd_period = (xcd_group && p.group > 1 && period > 1 && num_qo_heads % period == 0 &&
                        (num_qo_heads / period) % p.group == 0) ? period : 1u;
        // Dispatch order by address class (round 5, scripts/probe/addr_class_probe.hip): on the NHD pool a head's K/V rows
        // are 256-byte pieces at one offset inside every 1 KiB, and pieces whose address bits 8-9 are 01 are served ~20 %
        // slower than the others under mixed traffic -- heads 1, 5, 9, ... of a 128-wide MHA pool.  A single sequence's
        // launch is two workgroups per CU, dispatched rows-first: the first 16 rows' workgroups get ahead of the second
        // 16 (they start their gather ~1 us earlier and finish ~2 us earlier), so the slow heads go FIRST, where they
        // have that slack, instead of deciding the end of the launch from the second half.
        // Measured at cfg 3, same box, us per launch: 12.04 -> 11.82 (profiles/r05_ab_slow_class_heads_first.txt);
        // QUEST_SLOW_FIRST=0 keeps the plain order (A/B).  Same work per head, same bits.
        static const int slow_env = [] { const char* e = quest_tuning_env("QUEST_SLOW_FIRST"); return e ? atoi(e) : 1; }();
        if (slow_env && p.xcd_period == 1 && (p.group & (p.group - 1u)) == 0 && kv.layout == QUEST_LAYOUT_NHD &&
            kv.head_dim == 128 && kv.num_heads % 4u == 0 && batch.n_seqs == 1) {
            const uint32_t c0 = (uint32_t)(((uintptr_t)kv.data >> 8) & 3u);  // class of kv head 0
            p.xcd_period |= (1u + ((1u - c0) & 3u)) << 8 | (uint32_t)__builtin_ctz(p.group) << 12;
        }
    }
    // fc > 0: capacity (keys per thread) of the fused top-k front end; 0 = page ids come from an index tensor
    fc = 0;
    waves = h->dec_waves;
    p.cpt = 0;
    if (fused && p.vec_front == 8u) {
        waves = 8;
        fc = 8;
        p.cpt = (n_scores + 7u) / 8u <= 2u * 512u ? 2u : 4u;  // tile keys per thread
        p.table_vec = table_vec ? 1u : 0u;
    } else if (fused) {
        // the selection is VALU-issue bound (~1000 instructions per wave at 8 keys per thread), so rows
        // beyond 1024 pages get 8 waves (<= 4 keys per thread up to 2048 pages, <= 8 up to 4096); the
        // attention part runs the same with 4 or 8 waves.  Measured at cfg 3: 15.2 vs 15.8 us.
        if (kv.page_size == 16 && n_scores > 4u * 4u * kWave) waves = 8;
        const uint32_t nt = (kv.page_size == 16 ? waves : 4u) * kWave;
        const uint32_t per_thread = (n_scores + (p.row_lead ? 3u : 0u) + nt - 1) / nt;  // (+ the skipped leading columns)
        // (24: rows of 8193-12288 columns at 512 threads -- cfg 4's capacity of ~8320 pages needs 17 keys per thread; the
        // 32-key instantiation carries two more dead load / histogram / bitmap rounds in every unrolled phase.
        // QUEST_FC24=0 takes the 32-key instantiation instead: A/B)
        static const bool fc24 = [] { const char* e = quest_tuning_env("QUEST_FC24"); return !e || atoi(e) != 0; }();
        fc = per_thread <= 8 ? 8 : per_thread <= 16 ? 16 : (per_thread <= 24 && fc24) ? 24 : per_thread <= 32 ? 32 : 64;
        // ownership chunk: a multiple of 4 columns when the register capacity allows, so a thread's keys are one
        // 8/16-byte LDS read (topk_load_keys)
        const uint32_t r4 = (per_thread + 3) / 4 * 4;
        p.cpt = r4 <= (uint32_t)fc ? r4 : per_thread;
        // rows of <= 8 columns per thread with vector-loadable chunks: each thread takes its own columns and page ids
        // straight into registers (no staging arrays).  Measured (us per launch, staged -> direct): cfg 3 12.79 ->
        // 12.38; 8 sequences batched 48.6 -> 48.85, cfg 5 51.2 -> 51.2 -> single-sequence launches only.
        // QUEST_FE1_DIRECT=0 keeps the staged variant, =2 takes the direct one for batches too (tuning, tests).
        static const int direct_env = [] { const char* e = quest_tuning_env("QUEST_FE1_DIRECT"); return e ? atoi(e) : 1; }();
        const bool direct_ok = direct_env == 2 || (direct_env == 1 && batch.n_seqs == 1);
        const bool direct_possible = rows_aligned && p.vec_front != 2 && forced != 1 && fc == 8 &&
            (p.cpt == 4 || (p.cpt == 8 && ((uintptr_t)scores & 15u) == 0 && p.score_stride % 8u == 0 &&
                            p.score_stride >= ((n_scores + 7u) & ~7u)));
        if (direct_possible && direct_ok && p.vec_front == 1) p.vec_front = 3;
        // (Round 4 built two more front ends and measured both slower: column-range ownership -- a workgroup gathers the
        // selected pages of ITS range of columns, no rank scan, no page-list hand-off: cfg 3 13.27 vs 12.07 us, cfg 4 20.34
        // vs 19.54, the per-CU page imbalance costs more than the earlier page list returns -- and a third generation for
        // long rows, pre-filter -> compact -> short-row select: 22.4 vs 19.6 us at cfg 4.  Removed from the tree in round 5,
        // DESIGN.md 3.5; history: git log -S sparse_decode_colrange_body, -S fe3_select.)
        p.table_vec = table_vec ? 1u : 0u;
    }
    return 0;
}

static int decode_entry(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t kv,
                        uint32_t num_qo_heads, const void* scores, uint32_t n_scores, void* topk_val_out,
                        int32_t* topk_idx_out, float* lse, hipStream_t s, uint32_t score_stride = 0,
                        const quest_step_state_t* state = nullptr, quest_batch_t batch = {1, 0, 0, 0}, uint32_t tile_off = 0) {
    DecodeParams p;
    int fc;
    uint32_t waves;
    if (int e = plan_decode(h, q, o, kv, num_qo_heads, scores, n_scores, topk_val_out, topk_idx_out, lse, score_stride, state,
                            batch, p, fc, waves, tile_off))
        return e;
    switch (kv.head_dim) {
        case 64: return launch_decode<64>(h, p, num_qo_heads, fc, waves, s, batch.n_seqs);
        case 128: return launch_decode<128>(h, p, num_qo_heads, fc, waves, s, batch.n_seqs);
        case 256: return launch_decode<256>(h, p, num_qo_heads, fc, waves, s, batch.n_seqs);
        default: return QUEST_EUNSUPPORTED;
    }
}

extern "C" int quest_decode_forward(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t kv,
                                    uint32_t num_qo_heads, float* lse, quest_stream_t stream) {
    return decode_entry(h, q, o, kv, num_qo_heads, nullptr, 0, nullptr, nullptr, lse, (hipStream_t)stream);
}

template <int D>
static int launch_shared(const quest_decode_handler* h, const DecodeParams& p_in, uint32_t num_qo_heads, uint32_t gs,
                         hipStream_t s, uint32_t n_seqs) {
    // (8-wave workgroups for group sizes 1 / 2 -- 16 waves per CU instead of 8, each with one page in flight -- measured
    // slower in round 4: cfg 2 18.4 vs 17.8 us per layer, the 32K dense kernel 94.4 vs 92.4 us)
    static const char* order_env = quest_tuning_env("QUEST_SHARED_HEADS_FIRST");
    const bool nhd = p_in.st.head < p_in.st.entry;
    const bool heads_first = order_env ? atoi(order_env) != 0 : nhd && num_qo_heads / gs > 1;
    DecodeParams p = p_in;
    p.xcd_period = heads_first ? 1u : 0u;
    dim3 grid(p.n_chunks, num_qo_heads / gs, n_seqs), block(4 * kWave);
    if (heads_first) grid = dim3(num_qo_heads / gs, p.n_chunks, n_seqs);
#define QUEST_SHARED_CASE(GS)                                                                                                  \
    case GS:                                                                                                                   \
        if (p.app_k)                                                                                                           \
            hipLaunchKernelGGL((shared_decode_kernel<D, GS, 4, true>), grid, block, 0, s, QUEST_DECODE_HEAD_ARGS(p, num_qo_heads), p); \
        else                                                                                                                   \
            hipLaunchKernelGGL((shared_decode_kernel<D, GS, 4, false>), grid, block, 0, s, QUEST_DECODE_HEAD_ARGS(p, num_qo_heads), p); \
        break;
    switch (gs) {
        QUEST_SHARED_CASE(1)
        QUEST_SHARED_CASE(2)
        QUEST_SHARED_CASE(4)
        QUEST_SHARED_CASE(8)
        default: return QUEST_EUNSUPPORTED;
    }
#undef QUEST_SHARED_CASE
    QUEST_LAUNCH_CHECK();
    if (p.n_chunks > 1 && !h->skip_merge) {
        hipLaunchKernelGGL((merge_states_kernel<D>), dim3(num_qo_heads * n_seqs), dim3(D * kMergeGroups), 0, s,
                           (const float*)p.ws, p.o, p.lse, p.n_chunks, p.ws_stride,
                           take_armed_advance(const_cast<quest_decode_handler*>(h)));
        QUEST_LAUNCH_CHECK();
    }
    return flush_armed_advance(const_cast<quest_decode_handler*>(h), s);
}

struct SharedAppend {  // optional decode append riding in the group-shared launch
    const void* k = nullptr;
    const void* v = nullptr;
    const quest_paged_kv_t* metadata = nullptr;
};

static int shared_entry(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t kv, uint32_t num_qo_heads,
                        float* lse, const quest_step_state_t* state, quest_stream_t stream,
                        quest_batch_t batch = {1, 0, 0, 0}, SharedAppend app = {}) {
    if (!h || batch.n_seqs == 0) return QUEST_EINVAL;
    if (state) kv.last_page_len = 1;  // placeholder; read from `state` in the kernel
    if (!h->started) return QUEST_ESTATE;
    if (batch.n_seqs > h->batch) return QUEST_ESTATE;
    if (!q || !o || !kv.data || (h->n_sel > 0 && !kv.indices)) return QUEST_EINVAL;
    if (kv.layout != h->layout || kv.head_dim != h->head_dim || kv.page_size != h->page_size ||
        kv.num_heads != h->num_kv_heads || num_qo_heads != h->num_qo_heads)
        return QUEST_EINVAL;
    if (kv.last_page_len == 0 || kv.last_page_len > kv.page_size) return QUEST_EINVAL;
    if (kv.page_size != 16 || (kv.head_dim != 64 && kv.head_dim != 128)) {
        // outside the group-shared kernel's set.  Eager callers get EUNSUPPORTED and pass the per-head index
        // tensor to quest_decode_forward; a state-driven launch has no such tensor, so it runs the per-head-list
        // kernel with every head reading the ONE page table (row stride 0) and the live length from `state`.
        if (!state || app.k) return QUEST_EUNSUPPORTED;  // (with an append: the caller issues it as its own launch)
        kv.page_budget = 0;
        return decode_entry(h, q, o, kv, num_qo_heads, nullptr, 0, nullptr, nullptr, lse, (hipStream_t)stream, 0,
                            state, batch);
    }
    DecodeParams p{};
    p.q = (const half_t*)q;
    p.o = (half_t*)o;
    p.lse = lse;
    p.kv = (const half_t*)kv.data;
    p.indices = kv.indices;
    p.ws = h->ws;
    p.st = pool_strides(kv);
    p.n_sel = h->n_sel;
    p.last_page_len = kv.last_page_len;
    p.last_page_idx = kv.last_page_idx;
    p.page_size = kv.page_size;
    p.group = num_qo_heads / kv.num_heads;
    p.pages_per_chunk = h->shared_ppc;
    p.n_chunks = h->shared_chunks;
    p.scale_log2 = (float)(1.4426950408889634 / sqrt((double)kv.head_dim));
    p.ws_stride = h->ws_stride;
    p.state = state;
    p.table_stride = batch.kv_table_stride;
    if (app.k) {
        const quest_paged_kv_t& m = *app.metadata;
        if (!app.v || !m.data) return QUEST_EINVAL;
        // the append addresses the metadata pool with the KV pool's strides: same geometry required (it is: the
        // controller builds both from one KvCache class with the same page size, heads, dim and layout)
        if (m.layout != kv.layout || m.head_dim != kv.head_dim || m.page_size != kv.page_size || m.num_heads != kv.num_heads)
            return QUEST_EINVAL;
        if (!state && (m.last_page_len == 0 || m.last_page_len > m.page_size)) return QUEST_EINVAL;
        p.app_k = (const half_t*)app.k;
        p.app_v = (const half_t*)app.v;
        p.app_meta = (half_t*)m.data;
        p.meta_last_page_len = m.last_page_len;
        p.meta_last_page_idx = m.last_page_idx;
    }
    hipStream_t s = (hipStream_t)stream;
    return kv.head_dim == 64 ? launch_shared<64>(h, p, num_qo_heads, p.group, s, batch.n_seqs)
                             : launch_shared<128>(h, p, num_qo_heads, p.group, s, batch.n_seqs);
}

extern "C" int quest_decode_forward_shared(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t kv,
                                           uint32_t num_qo_heads, float* lse, quest_stream_t stream) {
    return shared_entry(h, q, o, kv, num_qo_heads, lse, nullptr, stream);
}

extern "C" int quest_decode_forward_shared_dyn(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t kv,
                                               uint32_t num_qo_heads, const quest_step_state_t* state, float* lse,
                                               quest_stream_t stream) {
    if (!state) return QUEST_EINVAL;
    return shared_entry(h, q, o, kv, num_qo_heads, lse, state, stream);
}

extern "C" int quest_decode_append_forward_shared_dyn(quest_decode_handler_t* h, const void* k, const void* v,
                                                      quest_paged_kv_t metadata, const void* q, void* o, quest_paged_kv_t kv,
                                                      uint32_t num_qo_heads, const quest_step_state_t* state, float* lse,
                                                      quest_stream_t stream) {
    if (!state || !k || !v) return QUEST_EINVAL;
    SharedAppend app;
    app.k = k, app.v = v, app.metadata = &metadata;
    return shared_entry(h, q, o, kv, num_qo_heads, lse, state, stream, {1, 0, 0, 0}, app);
}

extern "C" int quest_decode_append_forward_shared_batched(quest_decode_handler_t* h, const void* k, const void* v,
                                                          quest_paged_kv_t metadata, const void* q, void* o,
                                                          quest_paged_kv_t kv, uint32_t num_qo_heads,
                                                          const quest_step_state_t* state, quest_batch_t batch, float* lse,
                                                          quest_stream_t stream) {
    if (!state || !k || !v) return QUEST_EINVAL;
    if (h && batch.n_seqs > 1 && batch.kv_table_stride < h->n_sel + 1) return QUEST_EINVAL;
    SharedAppend app;
    app.k = k, app.v = v, app.metadata = &metadata;
    return shared_entry(h, q, o, kv, num_qo_heads, lse, state, stream, batch, app);
}

extern "C" int quest_decode_forward_fused_topk_strided(quest_decode_handler_t* h, const void* q, void* o,
                                                       quest_paged_kv_t kv, uint32_t num_qo_heads, const void* scores,
                                                       uint32_t n_scores, uint32_t score_stride, void* topk_val_out,
                                                       int32_t* topk_idx_out, float* lse, quest_stream_t stream) {
    if (!scores || (score_stride != 0 && score_stride < n_scores)) return QUEST_EINVAL;
    return decode_entry(h, q, o, kv, num_qo_heads, scores, n_scores, topk_val_out, topk_idx_out, lse,
                        (hipStream_t)stream, score_stride);
}

extern "C" int quest_decode_forward_fused_topk(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t kv,
                                               uint32_t num_qo_heads, const void* scores, uint32_t n_scores,
                                               void* topk_val_out, int32_t* topk_idx_out, float* lse,
                                               quest_stream_t stream) {
    return quest_decode_forward_fused_topk_strided(h, q, o, kv, num_qo_heads, scores, n_scores, 0, topk_val_out,
                                                   topk_idx_out, lse, stream);
}

extern "C" int quest_decode_forward_fused_topk_dyn(quest_decode_handler_t* h, const void* q, void* o,
                                                   quest_paged_kv_t kv, uint32_t num_qo_heads, const void* scores,
                                                   uint32_t score_stride, uint32_t max_n_scores,
                                                   const quest_step_state_t* state, float* lse, quest_stream_t stream) {
    if (!scores || !state || score_stride < max_n_scores) return QUEST_EINVAL;
    // dispatch (keys per thread) is chosen for the longest row the graph will see
    return decode_entry(h, q, o, kv, num_qo_heads, scores, max_n_scores, nullptr, nullptr, lse, (hipStream_t)stream,
                        score_stride, state);
}

extern "C" int quest_decode_forward_fused_topk_tiles_dyn(quest_decode_handler_t* h, const void* q, void* o,
                                                         quest_paged_kv_t kv, uint32_t num_qo_heads, const void* scores,
                                                         uint32_t score_stride, uint32_t max_n_scores,
                                                         uint32_t tile_max_offset, const quest_step_state_t* state, float* lse,
                                                         quest_stream_t stream) {
    if (!scores || !state || score_stride < max_n_scores || tile_max_offset == 0) return QUEST_EINVAL;
    return decode_entry(h, q, o, kv, num_qo_heads, scores, max_n_scores, nullptr, nullptr, lse, (hipStream_t)stream,
                        score_stride, state, {1, 0, 0, 0}, tile_max_offset);
}

extern "C" int quest_decode_forward_fused_topk_batched(quest_decode_handler_t* h, const void* q, void* o,
                                                       quest_paged_kv_t kv, uint32_t num_qo_heads, const void* scores,
                                                       uint32_t score_stride, uint32_t max_n_scores,
                                                       const quest_step_state_t* state, quest_batch_t batch, float* lse,
                                                       quest_stream_t stream) {
    if (!scores || !state || score_stride < max_n_scores) return QUEST_EINVAL;
    if (batch.n_seqs > 1 && batch.kv_table_stride < max_n_scores + 1) return QUEST_EINVAL;
    return decode_entry(h, q, o, kv, num_qo_heads, scores, max_n_scores, nullptr, nullptr, lse, (hipStream_t)stream,
                        score_stride, state, batch);
}

// append + estimate + top-k + sparse attention of one layer for a whole batch in ONE launch (layer_device.cuh).
// QUEST_EUNSUPPORTED when the plan or the shape is outside what the launch serves: the caller then issues
// quest_append_estimate_batched + quest_decode_forward_fused_topk_batched (same bits).
extern "C" int quest_decode_layer_fused_batched(quest_decode_handler_t* h, const void* k, const void* v,
                                                quest_paged_kv_t metadata, const void* q, void* o, quest_paged_kv_t kv,
                                                uint32_t num_qo_heads, uint32_t max_n_scores,
                                                const quest_step_state_t* state, quest_batch_t batch, void* scores_out,
                                                uint32_t score_stride, float* lse, quest_stream_t stream) {
    if (!h || !k || !v || !q || !o || !state || batch.n_seqs == 0 || max_n_scores == 0) return QUEST_EINVAL;
    if (!h->started) return QUEST_ESTATE;
    if (batch.n_seqs > h->batch) return QUEST_ESTATE;
    kv.last_page_len = metadata.last_page_len = 1;  // placeholders; the kernel reads the real ones from `state`
    if (int e = check_pool(kv)) return e;
    if (int e = check_pool(metadata)) return e;
    if (kv.layout != h->layout || kv.head_dim != h->head_dim || kv.page_size != h->page_size ||
        kv.num_heads != h->num_kv_heads || num_qo_heads != h->num_qo_heads)
        return QUEST_EINVAL;
    // the append and the estimate address the metadata pool with the KV pool's strides: same geometry required
    if (metadata.layout != kv.layout || metadata.head_dim != kv.head_dim || metadata.page_size != kv.page_size ||
        metadata.num_heads != kv.num_heads)
        return QUEST_EINVAL;
    if (scores_out && score_stride < max_n_scores) return QUEST_EINVAL;
    if (batch.n_seqs > 1 && (batch.kv_table_stride < max_n_scores + 1 ||
                             (uint64_t)batch.meta_table_stride * metadata.page_size < max_n_scores))
        return QUEST_EINVAL;
    // served: one workgroup per head (the planner's split for a batch that fills the chip), the whole selection in one
    // workgroup's LDS list, page size 16, head_dim 128 / 64, rows the 8-keys-per-thread selection covers
    constexpr uint32_t NW = 8, NT = NW * kWave, FC = 8;
    if (h->n_chunks != 1 || h->n_sel + 1 > (uint32_t)kFusedMaxPpc || h->n_sel == 0) return QUEST_EUNSUPPORTED;
    if (kv.page_size != 16 || (kv.head_dim != 128 && kv.head_dim != 64)) return QUEST_EUNSUPPORTED;
    if (max_n_scores > FC * NT) return QUEST_EUNSUPPORTED;
    LayerParams p{};
    p.q = (const half_t*)q;
    p.state = state;
    p.meta = (const half_t*)metadata.data;
    p.meta_tables = metadata.indices;
    p.kv_tables = kv.indices;
    p.n_cap = max_n_scores;
    p.meta_table_stride = batch.meta_table_stride;
    p.kv_table_stride = batch.kv_table_stride;
    const uint32_t per_thread = (max_n_scores + NT - 1) / NT, r4 = (per_thread + 3) / 4 * 4;
    p.cpt = r4 <= FC ? r4 : per_thread;
    p.group = num_qo_heads / kv.num_heads;
    {   // XCD-aware row order for GQA: one workgroup per head -> period 8 (see plan_decode)
        static const bool xcd_group = [] { const char* e = quest_tuning_env("QUEST_XCD_GROUP"); return !e || atoi(e) != 0; }();
        p.xcd_period = (xcd_group && p.group > 1 && num_qo_heads % 8u == 0 && (num_qo_heads / 8u) % p.group == 0) ? 8u : 1u;
    }
    p.k_new = (const half_t*)k;
    p.v_new = (const half_t*)v;
    p.kv = (half_t*)kv.data;
    p.o = (half_t*)o;
    p.lse = lse;
#ifdef QUEST_WALLSTAMPS
    p.ws = h->ws;
#endif
    p.budgets = batch.page_budgets;
    p.scores_out = (uint16_t*)scores_out;
    p.score_stride = score_stride;
    p.sel_val_out = (uint16_t*)h->sel_val_out;
    p.sel_idx_out = h->sel_idx_out;
    p.sel_stride = h->n_sel;
    p.st = pool_strides(kv);
    p.n_sel = h->n_sel;
    p.num_kv_heads = kv.num_heads;
    p.ids_lds_offset = (uint32_t)((((size_t)max_n_scores * 2) + 15) & ~(size_t)15);
    p.ws_stride = h->ws_stride;
    p.scale_log2 = (float)(1.4426950408889634 / sqrt((double)kv.head_dim));
    const size_t lds = (size_t)p.ids_lds_offset + (size_t)(max_n_scores + 1u) * 4u;
    dim3 grid(1, num_qo_heads, batch.n_seqs), block(NT);
    hipStream_t s = (hipStream_t)stream;
    if (kv.head_dim == 128)
        hipLaunchKernelGGL((layer_decode_kernel<128, FC, NW>), grid, block, lds, s, QUEST_LAYER_HEAD_ARGS(p, num_qo_heads), p);
    else
        hipLaunchKernelGGL((layer_decode_kernel<64, FC, NW>), grid, block, lds, s, QUEST_LAYER_HEAD_ARGS(p, num_qo_heads), p);
    QUEST_LAUNCH_CHECK();
    uint32_t* info = h->last_launch;  // front-end variant 7 = the one-launch layer (keys from LDS)
    info[0] = FC, info[1] = NW, info[2] = 7u, info[3] = 1u, info[4] = 1u, info[5] = batch.n_seqs;
    return flush_armed_advance(h, s);  // (no merge launch to ride in: the armed reservation follows as its own launch)
}

extern "C" int quest_decode_forward_batched(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t kv,
                                            uint32_t num_qo_heads, const int32_t* indices, uint32_t idx_stride,
                                            const quest_step_state_t* state, quest_batch_t batch, float* lse,
                                            quest_stream_t stream) {
    if (!h || !state || !indices) return QUEST_EINVAL;
    if (idx_stride < h->n_sel) return QUEST_EINVAL;  // rows must hold the plan's selected-page count
    kv.indices = indices;        // [n_seqs][num_qo_heads][idx_stride]
    kv.page_budget = idx_stride;  // row stride of a head's list
    batch.kv_table_stride = num_qo_heads * idx_stride;  // entries between the sequences' index blocks
    return decode_entry(h, q, o, kv, num_qo_heads, nullptr, 0, nullptr, nullptr, lse, (hipStream_t)stream, 0, state, batch);
}

extern "C" int quest_decode_forward_shared_batched(quest_decode_handler_t* h, const void* q, void* o,
                                                   quest_paged_kv_t kv, uint32_t num_qo_heads,
                                                   const quest_step_state_t* state, quest_batch_t batch, float* lse,
                                                   quest_stream_t stream) {
    if (!state) return QUEST_EINVAL;
    if (h && batch.n_seqs > 1 && batch.kv_table_stride < h->n_sel + 1) return QUEST_EINVAL;
    return shared_entry(h, q, o, kv, num_qo_heads, lse, state, stream, batch);
}

extern "C" const char* quest_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case QUEST_EINVAL: return "quest: invalid argument";
        case QUEST_EUNSUPPORTED: return "quest: unsupported head_dim/page_size/group size";
        case QUEST_ESTATE: return "quest: begin_forward() must be called before forward()";
        case QUEST_ETOOLARGE: return "quest: top-k row exceeds QUEST_TOPK_MAX_ROW";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "quest: unknown error";
    }
}

// "quest_hip gfx950 src=<16 hex digits> r4": the digits are build.py's hash of csrc/*, include/quest_hip.h and the compiler
// flags; quest_amd/_lib.py recomputes it from the sources at import and refuses a library built from anything else.
#ifndef QUEST_SRC_HASH
#define QUEST_SRC_HASH "unhashed-build!!"
#endif
extern "C" const char* quest_build_info(void) { return "quest_hip gfx950 src=" QUEST_SRC_HASH " r4"; }
