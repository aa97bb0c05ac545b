// One persistent launch per layer for the whole decode-token chain:
//   append -> estimate -> top-k select -> sparse attention -> merge           (QuestAttention.py:106-157)
//
// Why: with one sequence every kernel of the chain moves only 32 MiB (~5 us of HBM time) but pays ~2 us
// of launch boundary plus 1-4 us of dependent latency (page-table fetch, selection) during which HBM
// idles chip-wide because all workgroups of a launch sit in the same phase.  Here 2 workgroups per CU
// stay resident and pull work items from one queue; items of different kinds are interleaved in the queue
// so that while some workgroups run the latency-bound selection of one head tile, others stream metadata
// of the next:
//
//   A            (append the new token; 1 item per 256 lanes of work)
//   E(ht=0), E(ht=1), S(ht=0), E(ht=2), S(ht=1), ... , S(last)
//
//   E(ht) = estimate tiles of kv-head tile ht (8 entries x 8 kv heads each, as estimate_kernel)
//   S(ht) = (query head, chunk) attention items of that tile's heads: wait until all E(ht) items are done,
//           select the head's top-k pages (the shared routine of topk_select.cuh, recomputed per item),
//           gather the chunk's pages, leave a partial state; the LAST item of a head merges them.
//
// Cross-workgroup hand-offs follow the measured-valid form of the MI355X guide (inter-workgroup
// visibility, "valid forms", first table row): producer data is stored write-through (agent-scope / sc1
// stores), every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane bumps an
// agent-scope counter; consumers poll the counter (one lane, bounded spin, s_sleep), barrier, and read the
// produced bytes with sc1 loads only.  An E item never waits, and an S item only waits for E items that
// were dequeued earlier, so the queue order makes progress independent of dispatch order.  Every spin is
// bounded: on time-out the error word is set and the item proceeds (wrong numbers, no hang).
//
// Scope of this first version: head_dim 128, page size 16, kv heads a multiple of 8, query groups 1 or 4,
// rows <= 4096 pages, NHD or HND.  Everything else stays on the three-launch path.
#include <new>

#include "append_device.cuh"
#include "topk_select.cuh"

namespace quest {

constexpr int kLfThreads = 256;
constexpr int kLfWaves = kLfThreads / kWave;
constexpr int kLfD = 128;
constexpr int kLfLPR = kLfD / kVec;      // 16 lanes per row
constexpr int kLfR = kWave / kLfLPR;     // 4 rows per load instruction
constexpr int kLfIter = 4;               // estimate: load instructions per tensor per wave
constexpr int kLfRows = kLfWaves * kLfIter * kLfR;  // 64 rows per estimate tile
constexpr int kLfHW = 8;                 // kv heads per estimate tile
constexpr int kLfEW = kLfRows / kLfHW;   // 8 entries per estimate tile
constexpr int kLfMaxChunks = 64;
constexpr int kFusedMaxPpcLf = 64;  // pages per attention item
constexpr uint32_t kLfSpinLimit = 1u << 22;
constexpr float kLfNegFloor = -1.0e30f;

// sync words (uint32, device memory, all zero between launches)
// cntE is sharded 16 ways per head tile (an atomic hot word saturates at ~88 ops/us on this part; 256
// estimate tiles finishing together on one word cost ~3 us of serialisation per round)
constexpr int kSyDone = 1, kSyError = 2, kSyCntA = 3, kSyCntE = 16 /* [head tiles][16 shards] */,
              kSyCntS = 16 + 16 * 16 /* [query heads] */;
constexpr int kLfShards = 16;

struct LfSegment {
    uint32_t begin, count, kind, ht;  // kind: 0 = append, 1 = estimate tiles, 2 = attention items
};

struct LayerParams {
    const uint16_t* k_new;
    const uint16_t* v_new;
    quest_paged_kv_t kv;    // data, indices = full page table, last_page_len / last_page_idx
    quest_paged_kv_t meta;  // data, indices, last_page_len / last_page_idx
    const half_t* q;
    half_t* o;
    uint32_t* scores;       // [Hq][score_stride/2] pairs of fp16 scores (internal workspace)
    float* ws;              // [Hq][n_chunks][ws_stride]
    uint32_t* sync;
    uint32_t Hq, n_out, score_stride, n_sel, pages_per_chunk, n_chunks, ws_stride;
    uint32_t n_et, n_ht, n_a_items, total_items, n_segments;
    float scale_log2;
    LfSegment seg[12];
};

__device__ __forceinline__ uint32_t ld_u32_agent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_u32_agent(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_f32_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_f32_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte row piece through sc1 loads (only for the sequence's last page, which this launch wrote)
__device__ __forceinline__ half8 ld8_agent(const half_t* p) {
    const uint32_t* u = reinterpret_cast<const uint32_t*>(p);
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 r;
    r[0] = ld_u32_agent(u);
    r[1] = ld_u32_agent(u + 1);
    r[2] = ld_u32_agent(u + 2);
    r[3] = ld_u32_agent(u + 3);
    return __builtin_bit_cast(half8, r);
}

// One lane waits until *cnt >= target (bounded), then the workgroup proceeds.
__device__ __forceinline__ void wait_count(uint32_t* cnt, uint32_t target, uint32_t* err) {
    if (threadIdx.x == 0) {
        uint32_t spins = 0;
        while (ld_u32_agent(cnt) < target) {
            __builtin_amdgcn_s_sleep(8);
            if (++spins > kLfSpinLimit) {
                st_u32_agent(err, 1u);
                break;
            }
        }
    }
    __syncthreads();
}

// Wait until every shard s of a 16-way sharded counter has reached its share of `total` arrivals
// (arrival i goes to shard i % 16): lanes 0..15 of wave 0 poll one shard each.
__device__ __forceinline__ void wait_shards(uint32_t* cnt, uint32_t total, uint32_t* err) {
    if (threadIdx.x < kWave) {
        const uint32_t sh = threadIdx.x;
        const uint32_t target = sh < (uint32_t)kLfShards ? (total + kLfShards - 1 - sh) / kLfShards : 0u;
        uint32_t spins = 0;
        for (;;) {
            const bool ok = sh >= (uint32_t)kLfShards || ld_u32_agent(cnt + sh) >= target;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > kLfSpinLimit) {
                if (sh == 0) st_u32_agent(err, 1u);
                break;
            }
        }
    }
    __syncthreads();
}

// Producer side of a hand-off: all sc1 stores of this workgroup are drained, then ONE lane counts.
__device__ __forceinline__ uint32_t publish(uint32_t* cnt) {
    __shared__ uint32_t s_old;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return s_old;
}

struct LfRowState {
    float m = kLfNegFloor, d = 0.f;
    float8 acc = (float8)(0.f);
};

template <int G, bool HND, int FC>
__global__ __launch_bounds__(kLfThreads, 2) void decode_layer_kernel(LayerParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lf_smem[];
    __shared__ TopkSmem<kLfThreads> tk;
    __shared__ int32_t s_sel[kFusedMaxPpcLf];
    __shared__ float s_acc[kLfWaves][kLfD];
    __shared__ float s_md[kLfWaves][2];
    __shared__ float s_w[kLfMaxChunks];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int row = lane / kLfLPR, col = lane % kLfLPR;
    const uint32_t Hkv = p.kv.num_heads;
    const PoolStrides ks = pool_strides(p.kv), ms = pool_strides(p.meta);
    uint32_t* err = p.sync + kSyError;

    // Static round-robin assignment (item i -> workgroup i mod grid): no queue word to fight over.  Progress:
    // an attention item only waits for estimate items of LOWER index; by induction on the index every item
    // completes provided the grid is co-resident (the host caps it at 2 workgroups per CU).
    for (uint32_t item = blockIdx.x; item < p.total_items; item += gridDim.x) {
        uint32_t kind = 0, ht = 0, local = 0;
#pragma unroll 1
        for (uint32_t sgi = 0; sgi < p.n_segments; ++sgi)
            if (item >= p.seg[sgi].begin && item < p.seg[sgi].begin + p.seg[sgi].count) {
                kind = p.seg[sgi].kind;
                ht = p.seg[sgi].ht;
                local = item - p.seg[sgi].begin;
            }

        if (kind == 0) {
            // ------------------------------------------------------------------ append (write-through)
            const uint32_t t = local * kLfThreads + tid;
            if (t < Hkv * kLfLPR) {
                const uint32_t h = t / kLfLPR, f = (t % kLfLPR) * kVec;
                const uint32_t entry = p.kv.last_page_len - 1, mentry = p.meta.last_page_len - 1;
                uint16_t* kv_data = reinterpret_cast<uint16_t*>(p.kv.data);
                uint16_t* m_data = reinterpret_cast<uint16_t*>(p.meta.data);
                uint16_t* kdst = kv_data + (size_t)p.kv.last_page_idx * ks.page + (size_t)h * ks.head + (size_t)entry * ks.entry + f;
                uint16_t* mmax = m_data + (size_t)p.meta.last_page_idx * ms.page + (size_t)h * ms.head + (size_t)mentry * ms.entry + f;
                uint16_t* mmin = mmax + ms.v_off;
                const ushort8 k8 = *reinterpret_cast<const ushort8*>(p.k_new + (size_t)h * kLfD + f);
                const ushort8 v8 = *reinterpret_cast<const ushort8*>(p.v_new + (size_t)h * kLfD + f);
                ushort8 mx, mn;
                if (entry > 0) {
                    mx = *reinterpret_cast<const ushort8*>(mmax);
                    mn = *reinterpret_cast<const ushort8*>(mmin);
                } else {
                    mx = (ushort8)(kHalfNegMax);
                    mn = (ushort8)(kHalfMax);
                }
                mx = fold_max(mx, k8);
                mn = fold_min(mn, k8);
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                const u4 kw = __builtin_bit_cast(u4, k8), vw = __builtin_bit_cast(u4, v8);
                const u4 xw = __builtin_bit_cast(u4, mx), nw = __builtin_bit_cast(u4, mn);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    st_u32_agent(reinterpret_cast<uint32_t*>(kdst) + i, kw[i]);
                    st_u32_agent(reinterpret_cast<uint32_t*>(kdst + ks.v_off) + i, vw[i]);
                    st_u32_agent(reinterpret_cast<uint32_t*>(mmax) + i, xw[i]);
                    st_u32_agent(reinterpret_cast<uint32_t*>(mmin) + i, nw[i]);
                }
            }
            publish(p.sync + kSyCntA);
        } else if (kind == 1) {
            // ------------------------------------------------------------------ estimate tile (et, ht)
            half_t* qp_s = reinterpret_cast<half_t*>(lf_smem);      // [HW*G][D]
            half_t* qn_s = qp_s + kLfHW * G * kLfD;
            half_t* out_s = qn_s + kLfHW * G * kLfD;                 // [HW*G][EW]
            const uint32_t e0 = local * kLfEW, h0 = ht * kLfHW, S = p.meta.page_size;
            const half_t* data = reinterpret_cast<const half_t*>(p.meta.data);
            constexpr int QV = (kLfHW * G * kLfLPR + kLfThreads - 1) / kLfThreads;
            half8 qreg[QV];
#pragma unroll
            for (int t = 0; t < QV; ++t) {
                const uint32_t vi = tid + t * kLfThreads;
                qreg[t] = ld8(p.q + (size_t)h0 * G * kLfD + (size_t)(vi < kLfHW * G * kLfLPR ? vi : 0) * kVec);
            }
            uint32_t el[kLfIter], hl[kLfIter], ecl[kLfIter];
            size_t page[kLfIter];
#pragma unroll
            for (int j = 0; j < kLfIter; ++j) {
                const uint32_t r = (wave * kLfIter + j) * kLfR + row;
                el[j] = HND ? r % kLfEW : r / kLfHW;
                hl[j] = HND ? r / kLfEW : r % kLfHW;
                const uint32_t e = e0 + el[j];
                ecl[j] = e < p.n_out ? e : p.n_out - 1;
                page[j] = (size_t)p.meta.indices[ecl[j] / S];
            }
            half8 mx[kLfIter], mn[kLfIter];
#pragma unroll
            for (int j = 0; j < kLfIter; ++j) {
                const half_t* src = data + page[j] * ms.page + (size_t)(h0 + hl[j]) * ms.head +
                                    (size_t)(ecl[j] % S) * ms.entry + col * kVec;
                mx[j] = ld8_stream(src);
                mn[j] = ld8_stream(src + ms.v_off);
            }
#pragma unroll
            for (int t = 0; t < QV; ++t) {
                const uint32_t vi = tid + t * kLfThreads;
                if (vi < kLfHW * G * kLfLPR) {
                    half8 pos, neg;
#pragma unroll
                    for (int i = 0; i < kVec; ++i) {
                        const half_t x = qreg[t][i];
                        pos[i] = x > (half_t)0 ? x : (half_t)0;
                        neg[i] = x < (half_t)0 ? x : (half_t)0;
                    }
                    st8(qp_s + (size_t)vi * kVec, pos);
                    st8(qn_s + (size_t)vi * kVec, neg);
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kLfIter; ++j) {
                const float8 a = to_f32(mx[j]), b = to_f32(mn[j]);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t qh = hl[j] * G + g;
                    const float8 qp = to_f32(ld8(qp_s + (size_t)qh * kLfD + col * kVec));
                    const float8 qn = to_f32(ld8(qn_s + (size_t)qh * kLfD + col * kVec));
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < kVec; ++i) {
                        acc = __builtin_fmaf(qp[i], a[i], acc);
                        acc = __builtin_fmaf(qn[i], b[i], acc);
                    }
                    acc = row_allreduce_sum_fast<kLfLPR>(acc);
                    if (col == 0) out_s[qh * kLfEW + el[j]] = (half_t)acc;
                }
            }
            __syncthreads();
            // EW = 8 consecutive scores per query head leave as 4 write-through dwords (entries past n_out
            // land in the row's padding and are never read)
            const uint32_t* out_w = reinterpret_cast<const uint32_t*>(out_s);
            for (uint32_t t = tid; t < kLfHW * G * kLfEW / 2; t += kLfThreads) {
                const uint32_t qh = t / (kLfEW / 2), pr = t % (kLfEW / 2);
                st_u32_agent(p.scores + ((size_t)(h0 * G + qh) * p.score_stride + e0) / 2 + pr, out_w[t]);
            }
            publish(p.sync + kSyCntE + ht * kLfShards + local % kLfShards);
        } else {
            // ------------------------------------------------------------------ attention item (hq, chunk)
            const uint32_t hq = ht * kLfHW * G + local / p.n_chunks, chunk = local % p.n_chunks;
            const uint32_t hk = hq / G;
            const uint32_t n_slots = p.n_sel + 1;
            const uint32_t slot_begin = chunk * p.pages_per_chunk;
            const uint32_t slot_end = min(n_slots, slot_begin + p.pages_per_chunk);
            const half8 q_raw = ld8(p.q + (size_t)hq * kLfD + col * kVec);

            wait_shards(p.sync + kSyCntE + ht * kLfShards, p.n_et, err);
            {   // top-k front end: scores through sc1 loads, two keys per dword
                uint16_t* keys_s = reinterpret_cast<uint16_t*>(lf_smem);
                int32_t* ids_s = reinterpret_cast<int32_t*>(lf_smem + 4096 * 2);
                const uint32_t n = p.n_out;
                const uint32_t* srow = p.scores + (size_t)hq * p.score_stride / 2;
                uint32_t kraw[FC / 2];
                int32_t iraw[FC];
#pragma unroll
                for (int i = 0; i < FC / 2; ++i) {
                    const uint32_t w = tid + i * kLfThreads;  // dword index = columns 2w, 2w+1
                    kraw[i] = ld_u32_agent(srow + (2 * w < n ? w : (n - 1) / 2));
                }
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    const uint32_t e = tid + i * kLfThreads;
                    iraw[i] = p.kv.indices[e < n ? e : n - 1];
                }
                topk_clear<kLfThreads>(tk);
#pragma unroll
                for (int i = 0; i < FC / 2; ++i) {
                    const uint32_t w = tid + i * kLfThreads;
                    if (2 * w < n) keys_s[2 * w] = (uint16_t)half_key((uint16_t)(kraw[i] & 0xffffu));
                    if (2 * w + 1 < n) keys_s[2 * w + 1] = (uint16_t)half_key((uint16_t)(kraw[i] >> 16));
                }
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    const uint32_t e = tid + i * kLfThreads;
                    if (e < n) ids_s[e] = iraw[i];
                }
                __syncthreads();
                const uint32_t cpt = topk_cols_per_thread<kLfThreads>(n), c0 = tid * cpt;
                uint32_t key[FC];
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    const uint32_t c = c0 + i;
                    key[i] = keys_s[c < n ? c : n - 1];
                }
                TopkCursor cur = topk_select<kLfThreads, FC>(tk, key, n, p.n_sel, cpt);
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    uint32_t slot;
                    if (topk_take(cur, key[i], (uint32_t)i < cpt && c0 + i < n, slot) && slot >= slot_begin && slot < slot_end)
                        s_sel[slot - slot_begin] = ids_s[c0 + i];
                }
                __syncthreads();
            }
            const bool has_last = slot_end == n_slots;  // this chunk contains the page the launch appended to
            if (has_last) wait_count(p.sync + kSyCntA, p.n_a_items, err);

            float8 qv = to_f32(q_raw);
            qv *= p.scale_log2;
            const half_t* head_base = reinterpret_cast<const half_t*>(p.kv.data) + (size_t)hk * ks.head;
            const uint32_t lane_off = row * ks.entry + col * kVec;
            const uint32_t step = kLfR * ks.entry;
            LfRowState st;
            constexpr int T = 16 / kLfR;  // 4 load instructions per page per tensor
            for (uint32_t s0 = slot_begin + wave; s0 < slot_end; s0 += kLfWaves) {
                const bool last = s0 >= p.n_sel;
                const int32_t pg = __builtin_amdgcn_readfirstlane(last ? p.kv.last_page_idx : s_sel[s0 - slot_begin]);
                const int len = last ? (int)p.kv.last_page_len : 16;
                const half_t* b0 = head_base + (size_t)pg * ks.page;
                half8 k[T], v[T];
                if (last) {  // wave-uniform: bytes written by this launch -> sc1 loads
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        k[t] = ld8_agent(b0 + lane_off + t * step);
                        v[t] = ld8_agent(b0 + lane_off + t * step + ks.v_off);
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        k[t] = ld8_stream(b0 + lane_off + t * step);
                        v[t] = ld8_stream(b0 + lane_off + t * step + ks.v_off);
                    }
                }
                float sc[T];
                float m_new = st.m;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float8 kf = to_f32(k[t]);
                    float dot = 0.f;
#pragma unroll
                    for (int i = 0; i < kVec; ++i) dot = __builtin_fmaf(qv[i], kf[i], dot);
                    dot = row_allreduce_sum_fast<kLfLPR>(dot);
                    sc[t] = row < len - t * kLfR ? dot : kLfNegFloor;
                    m_new = __builtin_fmaxf(m_new, sc[t]);
                }
                const float scale = __builtin_amdgcn_exp2f(st.m - m_new);
                st.d *= scale;
                st.acc *= scale;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const bool valid = row < len - t * kLfR;
                    const float pr = valid ? __builtin_amdgcn_exp2f(sc[t] - m_new) : 0.f;
                    st.d += pr;
                    const float8 vf = valid ? to_f32(v[t]) : (float8)(0.f);
#pragma unroll
                    for (int i = 0; i < kVec; ++i) st.acc[i] = __builtin_fmaf(pr, vf[i], st.acc[i]);
                }
                st.m = m_new;
            }
#pragma unroll
            for (int off = kLfLPR; off < kWave; off <<= 1) {
                const float m_o = __shfl_xor(st.m, off, kWave), d_o = __shfl_xor(st.d, off, kWave);
                const float m_n = __builtin_fmaxf(st.m, m_o);
                const float a = __builtin_amdgcn_exp2f(st.m - m_n), b = __builtin_amdgcn_exp2f(m_o - m_n);
                st.d = st.d * a + d_o * b;
#pragma unroll
                for (int i = 0; i < kVec; ++i) st.acc[i] = st.acc[i] * a + __shfl_xor(st.acc[i], off, kWave) * b;
                st.m = m_n;
            }
            if (row == 0) {
#pragma unroll
                for (int i = 0; i < kVec; ++i) s_acc[wave][col * kVec + i] = st.acc[i];
                if (col == 0) {
                    s_md[wave][0] = st.m;
                    s_md[wave][1] = st.d;
                }
            }
            __syncthreads();
            float* wrec = p.ws + ((size_t)hq * p.n_chunks + chunk) * p.ws_stride;
            if (tid < kLfD) {
                float M = s_md[0][0];
#pragma unroll
                for (int w = 1; w < kLfWaves; ++w) M = __builtin_fmaxf(M, s_md[w][0]);
                float acc = 0.f, den = 0.f;
#pragma unroll
                for (int w = 0; w < kLfWaves; ++w) {
                    const float e = __builtin_amdgcn_exp2f(s_md[w][0] - M);
                    acc += e * s_acc[w][tid];
                    den += e * s_md[w][1];
                }
                st_f32_agent(wrec + tid, acc);
                if (tid == 0) {
                    st_f32_agent(wrec + kLfD, M);
                    st_f32_agent(wrec + kLfD + 1, den);
                }
            }
            const uint32_t old = publish(p.sync + kSyCntS + hq);
            if (old == p.n_chunks - 1) {
                // last item of this head: every partial is in memory -> merge, normalise, cast
                const float* wh = p.ws + (size_t)hq * p.n_chunks * p.ws_stride;
                float m_c = kLfNegFloor, d_c = 0.f;
                if ((uint32_t)lane < p.n_chunks) {
                    m_c = ld_f32_agent(wh + (size_t)lane * p.ws_stride + kLfD);
                    d_c = ld_f32_agent(wh + (size_t)lane * p.ws_stride + kLfD + 1);
                }
                float Mw = m_c;
#pragma unroll
                for (int off = kWave / 2; off > 0; off >>= 1) Mw = __builtin_fmaxf(Mw, __shfl_xor(Mw, off, kWave));
                const float e_c = (uint32_t)lane < p.n_chunks ? __builtin_amdgcn_exp2f(m_c - Mw) : 0.f;
                float dn = e_c * d_c;
#pragma unroll
                for (int off = kWave / 2; off > 0; off >>= 1) dn += __shfl_xor(dn, off, kWave);
                if (wave == 0 && (uint32_t)lane < p.n_chunks) s_w[lane] = e_c;
                __syncthreads();
                // two thread groups split the chunk loop; combine through LDS
                const uint32_t f = tid % kLfD, g2 = tid / kLfD;
                float a = 0.f;
                for (uint32_t c = g2; c < p.n_chunks; c += kLfThreads / kLfD)
                    a += s_w[c] * ld_f32_agent(wh + (size_t)c * p.ws_stride + f);
                s_acc[g2][f] = a;
                __syncthreads();
                if (tid < kLfD) {
                    float tot = 0.f;
#pragma unroll
                    for (int j = 0; j < kLfThreads / kLfD; ++j) tot += s_acc[j][tid];
                    p.o[(size_t)hq * kLfD + tid] = (half_t)(tot / dn);
                }
            }
            __syncthreads();
        }
    }
    // the last workgroup to leave zeroes the sync words for the next launch (nobody polls them any more)
    if (tid == 0) {
        const uint32_t d = __hip_atomic_fetch_add(p.sync + kSyDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == gridDim.x - 1) {
            for (uint32_t i = 0; i < kSyCntS + p.Hq; ++i)
                if (i != (uint32_t)kSyError) st_u32_agent(p.sync + i, 0u);
        }
    }
}

}  // namespace quest

using namespace quest;

// Workspace of the fused layer launch (lives beside the decode handler's plan).
struct quest_layer_ws {
    uint32_t* scores = nullptr;
    size_t scores_bytes = 0;
    float* ws = nullptr;
    size_t ws_bytes = 0;
    uint32_t* sync = nullptr;
};

extern "C" int quest_layer_ws_create(quest_layer_ws** out) {
    if (!out) return QUEST_EINVAL;
    quest_layer_ws* w = new (std::nothrow) quest_layer_ws();
    if (!w) return (int)hipErrorOutOfMemory;
    hipError_t e = hipMalloc((void**)&w->sync, 1024 * sizeof(uint32_t));
    if (e != hipSuccess) { delete w; return (int)e; }
    e = hipMemset(w->sync, 0, 1024 * sizeof(uint32_t));
    if (e != hipSuccess) { (void)hipFree(w->sync); delete w; return (int)e; }
    *out = w;
    return 0;
}

extern "C" void quest_layer_ws_destroy(quest_layer_ws* w) {
    if (!w) return;
    if (w->scores) (void)hipFree(w->scores);
    if (w->ws) (void)hipFree(w->ws);
    if (w->sync) (void)hipFree(w->sync);
    delete w;
}

// Reads (and clears) the time-out flag of the last launches: 0 = every hand-off completed.  Synchronises.
extern "C" int quest_layer_ws_error(quest_layer_ws* w, uint32_t* flag) {
    if (!w || !flag) return QUEST_EINVAL;
    hipError_t e = hipMemcpy(flag, w->sync + kSyError, sizeof(uint32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return (int)e;
    if (*flag) e = hipMemset(w->sync, 0, 1024 * sizeof(uint32_t));
    return (int)e;
}

template <int G, int FC>
static void launch_layer(bool hnd, dim3 grid, size_t lds, hipStream_t s, const LayerParams& p) {
    if (hnd)
        hipLaunchKernelGGL((decode_layer_kernel<G, true, FC>), grid, dim3(kLfThreads), lds, s, p);
    else
        hipLaunchKernelGGL((decode_layer_kernel<G, false, FC>), grid, dim3(kLfThreads), lds, s, p);
}

// The whole chain of one layer for one decode token in ONE launch (see the header of this file).
//   k, v: [1][Hkv][128] new token; q, o: [1][Hq][128]
//   kv.indices = the sequence's page table [n_pages]; kv/meta last_page_len/idx as for the separate ops
//   n_selected_pages = page budget - 1 (must be <= n_pages - 1: the sparse regime)
// Results are those of quest_append_kv_cache_decode + quest_estimate_attn_score + quest_topk_filtering +
// quest_decode_forward.  QUEST_EUNSUPPORTED when the shape is outside this version's set (callers then use
// the three-launch path).  Must not be in flight twice on the same workspace.
extern "C" int quest_decode_layer_fused(quest_layer_ws* w, const void* k, const void* v, const void* q, void* o,
                                        quest_paged_kv_t kv, quest_paged_kv_t meta, uint32_t num_qo_heads,
                                        uint32_t n_pages, uint32_t n_selected_pages, quest_stream_t stream) {
    if (!w || !k || !v || !q || !o || !kv.data || !kv.indices || !meta.data || !meta.indices) return QUEST_EINVAL;
    if (kv.layout > QUEST_LAYOUT_HND || kv.layout != meta.layout) return QUEST_EINVAL;
    if (kv.num_heads == 0 || num_qo_heads % kv.num_heads != 0 || kv.num_heads != meta.num_heads) return QUEST_EINVAL;
    if (n_pages < 2 || n_selected_pages == 0 || n_selected_pages > n_pages - 1) return QUEST_EINVAL;
    const uint32_t G = num_qo_heads / kv.num_heads, n_out = n_pages - 1;
    if (kv.head_dim != 128 || kv.page_size != 16 || meta.page_size != 16 || kv.num_heads % kLfHW != 0) return QUEST_EUNSUPPORTED;
    if ((G != 1 && G != 4) || n_out > 4096) return QUEST_EUNSUPPORTED;
    if (kv.last_page_len == 0 || kv.last_page_len > 16 || meta.last_page_len == 0 || meta.last_page_len > 16) return QUEST_EINVAL;

    LayerParams p{};
    p.k_new = (const uint16_t*)k;
    p.v_new = (const uint16_t*)v;
    p.kv = kv;
    p.meta = meta;
    p.q = (const half_t*)q;
    p.o = (half_t*)o;
    p.Hq = num_qo_heads;
    p.n_out = n_out;
    p.n_sel = n_selected_pages;
    p.score_stride = (n_out + 7) / 8 * 8;
    // plan: 512 resident workgroups; a head's page list in chunks of >= 4 pages, <= 64 chunks
    const uint32_t n_slots = n_selected_pages + 1;
    uint32_t chunks = 512 / num_qo_heads;
    if (chunks < 1) chunks = 1;
    if (chunks > n_slots) chunks = n_slots;
    if (chunks > (uint32_t)kLfMaxChunks) chunks = kLfMaxChunks;
    uint32_t ppc = (n_slots + chunks - 1) / chunks;
    if (ppc > (uint32_t)kFusedMaxPpcLf) return QUEST_EUNSUPPORTED;
    p.pages_per_chunk = ppc;
    p.n_chunks = (n_slots + ppc - 1) / ppc;
    p.ws_stride = (kLfD + 2 + 31) / 32 * 32;
    p.scale_log2 = (float)(1.4426950408889634 / sqrt(128.0));
    p.n_et = (n_out + kLfEW - 1) / kLfEW;
    p.n_ht = kv.num_heads / kLfHW;
    p.n_a_items = (kv.num_heads * kLfLPR + kLfThreads - 1) / kLfThreads;
    if (p.n_ht > 16 || kSyCntS + num_qo_heads > 1024) return QUEST_EUNSUPPORTED;

    const size_t need_scores = (size_t)num_qo_heads * p.score_stride * sizeof(uint16_t);
    if (need_scores > w->scores_bytes) {
        if (w->scores) (void)hipFree(w->scores);
        w->scores = nullptr;
        w->scores_bytes = 0;
        hipError_t e = hipMalloc((void**)&w->scores, need_scores);
        if (e != hipSuccess) return (int)e;
        w->scores_bytes = need_scores;
    }
    const size_t need_ws = (size_t)num_qo_heads * p.n_chunks * p.ws_stride * sizeof(float);
    if (need_ws > w->ws_bytes) {
        if (w->ws) (void)hipFree(w->ws);
        w->ws = nullptr;
        w->ws_bytes = 0;
        hipError_t e = hipMalloc((void**)&w->ws, need_ws);
        if (e != hipSuccess) return (int)e;
        w->ws_bytes = need_ws;
    }
    p.scores = w->scores;
    p.ws = w->ws;
    p.sync = w->sync;

    // queue order: A, E(0), E(1), S(0), E(2), S(1), ..., S(n_ht-1)
    const uint32_t s_items = kLfHW * G * p.n_chunks;
    uint32_t pos = 0, ns = 0;
    auto add = [&](uint32_t kind, uint32_t ht, uint32_t count) {
        p.seg[ns++] = LfSegment{pos, count, kind, ht};
        pos += count;
    };
    add(0, 0, p.n_a_items);
    add(1, 0, p.n_et);
    if (p.n_ht > 1) add(1, 1, p.n_et);
    for (uint32_t t = 0; t < p.n_ht; ++t) {
        add(2, t, s_items);
        if (t + 2 < p.n_ht) add(1, t + 2, p.n_et);
    }
    if (ns > 12) return QUEST_EUNSUPPORTED;
    p.n_segments = ns;
    p.total_items = pos;

    static int cu_count = 0;
    if (!cu_count) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return QUEST_EINVAL;
        cu_count = prop.multiProcessorCount;
    }
    const uint32_t resident = 2u * (uint32_t)cu_count;  // __launch_bounds__(256, 2): two workgroups per CU
    const uint32_t grid = pos < resident ? pos : resident;
    const size_t lds_e = (size_t)kLfHW * G * (2 * kLfD + kLfEW) * sizeof(uint16_t);
    const size_t lds_s = 4096 * 2 + 4096 * 4;
    const size_t lds = lds_e > lds_s ? lds_e : lds_s;
    const bool hnd = kv.layout == QUEST_LAYOUT_HND;
    hipStream_t s = (hipStream_t)stream;
    const int fc = n_out <= 2048 ? 8 : 16;
    if (G == 1 && fc == 8) launch_layer<1, 8>(hnd, dim3(grid), lds, s, p);
    else if (G == 1) launch_layer<1, 16>(hnd, dim3(grid), lds, s, p);
    else if (fc == 8) launch_layer<4, 8>(hnd, dim3(grid), lds, s, p);
    else launch_layer<4, 16>(hnd, dim3(grid), lds, s, p);
    QUEST_LAUNCH_CHECK();
    return 0;
}
