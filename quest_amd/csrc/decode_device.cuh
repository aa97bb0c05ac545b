// Device code of the sparse paged decode attention: kernel parameters, the online-softmax row state, the hand-off of
// the chained launch and the workgroup body shared by sparse_decode_kernel and chain_kernel (sparse_attn.hip, which
// also holds the reference citations and the design notes).
#pragma once
#include "estimate_device.cuh"
#include "topk_bitmap.cuh"

// Developer aid (scripts/timeline.py): -DQUEST_TIMELINE makes one workgroup of sparse_decode_kernel write
// clock stamps of its phases into the `lse` buffer instead of the log-sum-exp.
#ifdef QUEST_TIMELINE
#define QUEST_LSE_ENABLED false
#define QUEST_STAMP(i) \
    do { __builtin_amdgcn_s_waitcnt(0); tl[i] = clock64(); } while (0)
#else
#define QUEST_LSE_ENABLED true
#define QUEST_STAMP(i) \
    do { } while (0)
#endif

namespace quest {

// K/V tile loads of the per-head-list kernel: streaming (`nt`) by default; -DQUEST_KV_CACHED keeps them in the L2's normal
// replacement order (tuning: GQA heads of a group re-read each other's pages)
__device__ __forceinline__ half8 ld8_kv(const half_t* p) {
#ifdef QUEST_KV_CACHED
    return ld8(p);
#else
    return ld8_stream(p);
#endif
}

constexpr float kNegFloor = -1.0e30f;  // finite "-inf": exp2(floor - floor) stays finite, weights it carries are 0

struct DecodeParams {
    // The pointers come first: with -amdgpu-kernarg-preload-count=16 (build.py) the first 16 kernarg dwords
    // arrive in SGPRs at wave launch, so the first loads of a workgroup do not wait for a scalar load of the
    // argument block (0.3 us per launch on the fused kernel).
    const half_t* q;
    const half_t* kv;
    const int32_t* indices;
    const uint16_t* scores;           // fused front end: [Hq][n_scores] fp16 estimate output
    const quest_step_state_t* state;  // optional device-resident lengths / current page (graph replay)
    // dwords 10-15, still preloaded: what the first loads of the fused front end need besides the pointers (row
    // stride and capacity of the score rows, table stride of a batch, which loads to issue)
    uint32_t n_scores;      // fused top-k front end (FC > 0): `indices` is then the sequence's page table [n_scores + 1]
    uint32_t score_stride;  // row stride of `scores`
    uint32_t table_stride;  // batched launches (blockIdx.z = sequence): entries between page tables
    uint32_t stage_ids;     // fused front end: page ids staged in LDS next to the keys
    uint32_t vec_front;     // fused front end: 0 = first generation (topk_select.cuh); 1 = the same, staging arrays fed by
                            // 8/16-byte granule loads; 3 = the same, each thread loads its OWN cpt (4 or 8) columns and
                            // page ids straight into registers (no LDS staging); 2 = second generation
                            // (topk_bitmap.cuh).  1-3 need aligned score rows
    uint32_t cpt;           // fused front end: columns per thread (thread t owns [t*cpt, t*cpt + cpt)), host-chosen
    // ---- beyond the preloaded block (one scalar load of the argument segment)
    uint32_t idx_stride;
    half_t* o;
    float* ws;  // [Hq][n_chunks][ws_stride] fp32 partial (acc[D], m, d)
    float* lse;
    PoolStrides st;
    uint32_t n_sel;
    uint32_t last_page_len;
    int32_t last_page_idx;
    uint32_t page_size;
    uint32_t group;  // qo heads per kv head
    uint32_t pages_per_chunk;
    uint32_t n_chunks;
    float scale_log2;  // 1/sqrt(D) * log2(e)
    uint16_t* sel_val_out;   // optional [Hq][n_sel]
    int32_t* sel_idx_out;    // optional [Hq][n_sel]
    uint32_t ws_stride;  // floats per partial record (>= D + 2, multiple of 32 -> records own whole 128 B lines)
    uint32_t ids_lds_offset;
    uint32_t sel_stride;  // row stride of sel_val_out / sel_idx_out (the plan's n_sel; the live n_sel may be smaller)
    uint32_t xcd_period;  // > 1: grid row y serves query head (y % period) * (Hq / period) + y / period (see sparse_decode_kernel)
};

// Batched state-driven launch: blockIdx.z selects the sequence; every per-sequence operand is a row of a
// batched tensor ([n_seqs][Hq][...]), the pools are shared.  A single-sequence launch has blockIdx.z == 0.
// (Returned as a fresh set of pointers: rewriting the by-value kernel argument in place trips an
// address-space inference bug in this compiler.)
struct SeqView {
    const half_t* q;
    half_t* o;
    float* lse;
    const int32_t* indices;
    const uint16_t* scores;
    float* ws;
    const quest_step_state_t* state;
};
__device__ __forceinline__ SeqView select_sequence(const DecodeParams& p, uint32_t num_qo_heads, uint32_t head_dim,
                                                   uint32_t seq_index) {
    const size_t seq = seq_index, row = seq * num_qo_heads;
    SeqView v;
    v.q = p.q + row * head_dim;
    v.o = p.o + row * head_dim;
    v.lse = p.lse + row;  // only dereferenced when p.lse != nullptr
    v.indices = p.indices + seq * p.table_stride;
    v.scores = p.scores + row * p.score_stride;
    v.ws = p.ws + row * p.n_chunks * p.ws_stride;
    v.state = p.state + seq;  // only dereferenced when p.state != nullptr
    return v;
}

constexpr int kFusedMaxPpc = 128;  // pages per workgroup the fused front end can stage in LDS

template <int D>
struct RowState {
    float m = kNegFloor, d = 0.f;
    float8 acc = (float8)(0.f);
};

// Fold NG groups of R token rows (one load instruction each) into the row state with ONE rescale.
// len[g] is the number of valid rows counted from group g's first row (<= 0: none).
template <int D, int NG>
__device__ __forceinline__ void fold_groups(RowState<D>& st, const float8& qv, const half8 (&k)[NG], const half8 (&v)[NG],
                                            const int (&rows_left)[NG], int row) {
    constexpr int LPR = D / kVec;
    float s[NG];
    float m_new = st.m;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const float8 kf = to_f32(k[g]);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < kVec; ++i) dot = __builtin_fmaf(qv[i], kf[i], dot);
        dot = row_allreduce_sum_fast<LPR>(dot);
        s[g] = row < rows_left[g] ? dot : kNegFloor;
        m_new = __builtin_fmaxf(m_new, s[g]);
    }
    const float scale = __builtin_amdgcn_exp2f(st.m - m_new);
    st.d *= scale;
    st.acc *= scale;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        // rows past the page length hold stale pool bytes (possibly NaN/Inf): select, never multiply by 0
        const bool valid = row < rows_left[g];
        const float p = valid ? __builtin_amdgcn_exp2f(s[g] - m_new) : 0.f;
        st.d += p;
        const float8 vf = valid ? to_f32(v[g]) : (float8)(0.f);
#pragma unroll
        for (int i = 0; i < kVec; ++i) st.acc[i] = __builtin_fmaf(p, vf[i], st.acc[i]);
    }
    st.m = m_new;
}

constexpr uint32_t kChainReplicas = 16, kChainLineWords = 32;  // see ChainWait
#ifdef QUEST_CHAIN_TRACE  // developer aid (scripts/chain_trace.py): per workgroup {role/group, start, past the wait, end} wall-clock stamps
constexpr uint32_t kChainTraceBlocks = 4096;
constexpr uint32_t kChainMaxGroups = 32, kChainCounterWords = kChainMaxGroups * kChainReplicas * kChainLineWords, kChainSyncWords = kChainCounterWords + 2 + 8 * kChainTraceBlocks;
#define QUEST_CHAIN_STAMP(slot)                                                                          \
    do {                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < kChainTraceBlocks)                                          \
            reinterpret_cast<long long*>(cw.error + 2)[4 * blockIdx.x + (slot)] = wall_clock64();        \
    } while (0)
#define QUEST_CHAIN_ROLE(role, group)                                                                    \
    do {                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < kChainTraceBlocks)                                          \
            reinterpret_cast<long long*>(cw.error + 2)[4 * blockIdx.x] = ((long long)(role) << 32) | (group); \
    } while (0)
#else
constexpr uint32_t kChainMaxGroups = 32, kChainCounterWords = kChainMaxGroups * kChainReplicas * kChainLineWords, kChainSyncWords = kChainCounterWords + 2;
#define QUEST_CHAIN_STAMP(slot) \
    do { } while (0)
#define QUEST_CHAIN_ROLE(role, group) \
    do { } while (0)
#endif

// Chained launch (chain_kernel below): the estimate and the attention of a step run in ONE grid.  Workgroups are
// dispatched in index order; the estimate (and append) workgroups of a head group come before the attention
// workgroups that consume their scores and never wait for anything.  Hand-off, built from what was measured on MI355X
// (scripts/chain_trace.py):
//   * a device-scope RELEASE fence writes back the XCD's whole L2 (`buffer_wbl2 sc1`): ~8 us per workgroup, serialised
//     -> 28 us per launch.  So the producers' data (scores; the appended token) leave by write-through stores
//     (relaxed agent-scope atomic stores, `sc1`), each wave waits for its own stores to complete (`s_waitcnt`), the
//     workgroup meets at a barrier, and only then are the counters bumped (relaxed, no fence);
//   * several hundred workgroups polling ONE word queue up behind each other at its memory channel (4-6 us per poll,
//     and the estimate's own loads to that channel wait in the same queue) -> every group's counter exists in
//     kChainReplicas copies on separate 128-byte lines; producers bump all of them, a consumer polls one;
//   * a consumer's first thread polls (bounded: a stuck grid sets *error and carries on instead of hanging the GPU),
//     then an ACQUIRE fence (`buffer_inv sc1`, cheap) and a workgroup barrier;
//   * the counters are re-armed by the merge launch that follows (kernel boundary), so nothing is counted twice.
struct ChainWait {
    uint32_t* done;      // [groups][kChainReplicas] counters, one 128-byte line each
    uint32_t* error;     // set to 1 when a wait timed out
    uint32_t target;     // producers per group (estimate workgroups of the group + append workgroups)
};
#ifndef QUEST_CHAIN_SLEEP
#define QUEST_CHAIN_SLEEP 4  // x 64 cycles between two polls of the counter
#endif
constexpr long long kChainTimeoutTicks = 5000000;  // 50 ms of the 100 MHz wall clock

__device__ __forceinline__ uint32_t* chain_counter(const ChainWait& cw, uint32_t group, uint32_t replica) {
    return cw.done + ((size_t)group * kChainReplicas + replica) * kChainLineWords;
}
// all threads of a producer workgroup, after their last write-through store
__device__ __forceinline__ void chain_signal(const ChainWait& cw, uint32_t group_begin, uint32_t group_end) {
    __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have completed (device-wide: they are write-through)
    __syncthreads();
    if (threadIdx.x < kChainReplicas)
        for (uint32_t g = group_begin; g < group_end; ++g)
            __hip_atomic_fetch_add(chain_counter(cw, g, threadIdx.x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_wait(const ChainWait& cw, uint32_t group) {
    if (threadIdx.x == 0) {
        uint32_t* flag = chain_counter(cw, group, blockIdx.x % kChainReplicas);
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cw.target) {
            __builtin_amdgcn_s_sleep(QUEST_CHAIN_SLEEP);
            if (wall_clock64() - t0 > kChainTimeoutTicks) {
                *cw.error = 1u;
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// S_T = compile-time page size (16) or 0 for the generic run-time path.
//
// Addressing: the (page, kv head) tile base is wave-uniform (page id through readfirstlane -> SGPR
// pair), the per-lane part (row*entry_stride + col*8) is a 32-bit offset fixed for the whole kernel,
// so every load is `global_load_dwordx4 v, v_off, s[base] offset:imm` and the 16 loads of a slot pair
// cost 64 data VGPRs and no address VGPRs.
//
// FC > 0 enables the fused top-k front end: the 256 threads own FC columns each of the head's score
// row, run the shared selection routine (topk_select.cuh -- the same code as the stand-alone top-k
// kernel, hence the same pages), and the columns whose output slot falls in this workgroup's chunk
// drop their physical page id into LDS.  All workgroups of a head repeat the (cheap, L2-resident)
// selection instead of waiting for one another.
//
// The body is a device function so that the chained launch can run it as one role of a larger grid: (chunk, hq, seq)
// are the workgroup's coordinates, CHAIN = wait for the head group's scores (chain_wait) before touching them.
template <int D, int S_T, int FC, int NW, bool CHAIN>
__device__ __forceinline__ void sparse_decode_body(DecodeParams p, const uint32_t chunk, const uint32_t hq, const uint32_t seq,
                                                   const uint32_t num_qo_heads, const ChainWait& cw, const uint32_t wait_group) {
    constexpr int LPR = D / kVec, R = kWave / LPR;
    // wave index as an SGPR so per-wave control flow below is scalar branching
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = lane / LPR, col = lane % LPR;
    const uint32_t hk = hq / p.group;
#ifdef QUEST_TIMELINE
    long long tl[10] = {};
    long long sub_out[9] = {};
    const long long wall0 = wall_clock64();
    QUEST_STAMP(0);
#endif
    const SeqView sv = select_sequence(p, num_qo_heads, D, seq);
    // state-driven launches pass the longest row the graph will see in p.n_scores (it sizes FC); the live
    // row length comes from the state
    if (FC == 0 && p.state) {  // no front end, state-driven: full-KV decode of shapes the group-shared kernel
                               // does not cover (shared_entry's fallback); the list is the page table itself
        const quest_step_state_t st = *sv.state;
        p.n_scores = (uint32_t)(st.n_pages - 1);
        p.n_sel = min(p.n_sel, p.n_scores);
        p.last_page_len = (uint32_t)st.kv_last_page_len;
        p.last_page_idx = st.kv_last_page_idx;
    }
    // slots = selected pages + the current page; a state-driven launch on a sequence still shorter than the
    // budget selects ALL of its pages (k = n: the reference's full-attention branch, QuestAttention.py:123-132)
    // and workgroups whose chunk lies past the live list write an empty partial (weight 0 in the merge)
    uint32_t n_slots = p.n_sel + 1;
    const uint32_t slot_begin = chunk * p.pages_per_chunk;
    uint32_t slot_end = min(n_slots, slot_begin + p.pages_per_chunk);

    // q is requested now but first used after the top-k front end, so its latency hides under the selection
    const half8 q_raw = ld8(sv.q + (size_t)hq * D + col * kVec);

    const half_t* head_base = p.kv + (size_t)hk * p.st.head;       // uniform
    const uint32_t lane_off = row * p.st.entry + col * kVec;        // per lane, loop invariant
    const int32_t* idx_row = sv.indices + (size_t)hq * p.idx_stride;  // uniform
    RowState<D> st;

    __shared__ int32_t s_sel[FC > 0 ? kFusedMaxPpc : 1];
    if constexpr (FC > 0) {
        __shared__ TopkSmem<NW * kWave> sm;
        const uint32_t n_cap = p.n_scores;  // as launched: the longest row this launch may see (buffers cover it)
        constexpr int NT = NW * kWave;
        __shared__ uint32_t s_bm[2][kBmWords];
        extern __shared__ __attribute__((aligned(16))) unsigned char fe_dyn[];
        const uint16_t* srow = sv.scores + (size_t)hq * p.score_stride;
        Fe2Raw<fe2_has_ids(FC)> raw[FC / 4];
        uint4 own_keys = make_uint4(0u, 0u, 0u, 0u), own_ids[2] = {own_keys, own_keys};  // vec_front == 3
        const bool own_cols = FC == 8 && p.vec_front == 3;
        // the thread's own contiguous columns (vec_front 3): their page ids (1-2 x 16 bytes) and 16 bytes of scores per
        // lane (8 at 4 columns per thread) -- a coalesced sweep of the row -- all of it kept in registers
        const uint32_t own_c0 = threadIdx.x * p.cpt, own_cc = own_c0 < n_cap ? own_c0 : 0u;
        if (own_cols) {
            const uint32_t table_len = n_cap + 1u;
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if ((uint32_t)(4 * g) < p.cpt) {
                    const uint32_t c = own_cc + 4u * g;
                    if (c + 4u <= table_len) {
                        own_ids[g] = *reinterpret_cast<const uint4*>(sv.indices + c);
                    } else {  // tail of a table whose length is not a multiple of 4
                        const uint32_t last = table_len - 1u;
                        own_ids[g].x = (uint32_t)sv.indices[c < last ? c : last];
                        own_ids[g].y = (uint32_t)sv.indices[c + 1u < last ? c + 1u : last];
                        own_ids[g].z = (uint32_t)sv.indices[c + 2u < last ? c + 2u : last];
                        own_ids[g].w = (uint32_t)sv.indices[c + 3u < last ? c + 3u : last];
                    }
                }
        }
        // chained launch: everything above is independent of the estimate; the score row (and the appended token's
        // K/V further down) is not
        if constexpr (CHAIN) {
            QUEST_CHAIN_STAMP(1);
            chain_wait(cw, wait_group);
            QUEST_CHAIN_STAMP(2);
        }
        if (own_cols) {
            if (p.cpt == 8) {
                own_keys = *reinterpret_cast<const uint4*>(srow + own_cc);
            } else {
                const uint2 k2 = *reinterpret_cast<const uint2*>(srow + own_cc);
                own_keys.x = k2.x, own_keys.y = k2.y;
            }
            fe2_clear<NT>(sm);
        } else if (p.vec_front) {  // loads first: their addresses depend on the capacity only, not on the state below
            fe2_issue<NT, FC / 4, fe2_has_ids(FC)>(srow, sv.indices, n_cap + 1u, p.stage_ids != 0, n_cap, raw);
            fe2_clear<NT>(sm);
        }
        if (p.state) {  // live lengths
            const quest_step_state_t st = *sv.state;
            p.n_scores = (uint32_t)(st.n_pages - 1);
            p.last_page_len = (uint32_t)st.kv_last_page_len;
            p.last_page_idx = st.kv_last_page_idx;
            p.n_sel = min(p.n_sel, p.n_scores);
            n_slots = p.n_sel + 1;
            slot_end = min(n_slots, slot_begin + p.pages_per_chunk);
        }
        const uint32_t n = p.n_scores;
        if (n > 0 && p.vec_front == 2) {
            QUEST_STAMP(1);
            const size_t out_row = ((size_t)seq * num_qo_heads + hq) * p.sel_stride;
            const bool ids_staged = p.stage_ids && fe2_has_ids(FC);
            uint16_t* val_row = p.sel_val_out ? p.sel_val_out + out_row : nullptr;
            int32_t* idx_row_out = p.sel_idx_out ? p.sel_idx_out + out_row : nullptr;
            fe2_select<NT, FC>(sm, s_bm, raw, srow, sv.indices,
                               ids_staged ? reinterpret_cast<int32_t*>(fe_dyn + p.ids_lds_offset) : nullptr, n_cap, n,
                               p.n_sel, slot_begin, slot_end, s_sel, val_row, idx_row_out
#ifdef QUEST_TIMELINE
                               , sub_out
#endif
            );
            QUEST_STAMP(4);
            __syncthreads();
            if (!ids_staged) {  // block-uniform: columns -> pages, one parallel round trip
                fe2_resolve_pages(srow, sv.indices, slot_begin, slot_end, p.n_sel, s_sel, val_row, idx_row_out);
                __syncthreads();
            }
            QUEST_STAMP(5);
        } else if (FC <= 16 && n > 0) {  // first-generation front end (unaligned score rows; rows <= 4096 columns)
            // block-uniform; a one-page sequence has no row to select from (only the current page)
            // Ownership is fixed by the host from the row CAPACITY (p.cpt; NT * cpt >= n_cap >= n).
            const uint32_t cpt = p.cpt;
            const uint32_t c0 = threadIdx.x * cpt;
            const int32_t* table = sv.indices;
            uint16_t* keys_s = reinterpret_cast<uint16_t*>(fe_dyn);
            const bool stage_ids = p.stage_ids != 0;
            int32_t* ids_s = reinterpret_cast<int32_t*>(fe_dyn + p.ids_lds_offset);
            uint32_t key[FC];
            uint32_t mm = kMmNeutral;
            const bool direct = own_cols;  // keys / ids already in this thread's registers
            if (direct) {
                const uint32_t w[4] = {own_keys.x, own_keys.y, own_keys.z, own_keys.w};
#pragma unroll
                for (int i = 0; i < FC; ++i) {
                    key[i] = i < 8 ? half_key((uint16_t)((i & 1) ? w[(i >> 1) & 3] >> 16 : w[(i >> 1) & 3] & 0xffffu)) : 0u;
                    if ((uint32_t)i < cpt && c0 + i < n) mm = pk_max_u16(mm, mm_pack(key[i]));
                }
                QUEST_STAMP(1);
            } else if (p.vec_front == 1) {
                // aligned score rows: the granule loads issued at the top of the kernel (fe2_issue) feed the staging arrays
                if constexpr (fe2_has_ids(FC)) mm = fe1_stage_vector<NT, FC / 4>(raw, keys_s, stage_ids ? ids_s : nullptr, n_cap, n);
                QUEST_STAMP(1);
            } else {
            // coalesced loads (element t + i*NT), parked in LDS as keys (+ page ids when they fit)
            uint16_t kraw[FC];
            int32_t iraw[FC];
            // FC is the power-of-two register capacity; only the first ceil(n_cap / NT) rounds hold columns
            // (5 of 8 at 2058 pages x 512 threads): the rest are skipped by a wave-uniform test
            const uint32_t rounds = (n_cap + NT - 1) / NT;
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                kraw[i] = 0;
                iraw[i] = 0;
                if ((uint32_t)i < rounds) {
                    // clamped (unconditional) to the CAPACITY, not the live length: the addresses do not wait for
                    // the state load of a state-driven launch; columns in [n, n_cap) are readable and masked below
                    const uint32_t e = threadIdx.x + i * NT, ec = e < n_cap ? e : n_cap - 1;
                    kraw[i] = srow[ec];
                    iraw[i] = stage_ids ? table[ec] : 0;
                }
            }
            topk_clear<NT>(sm);  // overlaps the score / page-id loads above
            QUEST_STAMP(1);
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const uint32_t e = threadIdx.x + i * NT;
                if ((uint32_t)i < rounds && e < n) {
                    const uint32_t kk = half_key(kraw[i]);
                    mm = pk_max_u16(mm, mm_pack(kk));
                    keys_s[e] = (uint16_t)kk;
                    if (stage_ids) ids_s[e] = iraw[i];
                }
            }
            }
            topk_publish_range<NT>(sm, mm);
            QUEST_STAMP(2);
            __syncthreads();
            QUEST_STAMP(3);
            if (!direct) topk_load_keys<FC>(keys_s, c0, n, cpt, key);
#ifdef QUEST_TIMELINE
            long long sub[9] = {};
            TopkCursor cur = topk_select<NT, FC>(sm, key, n, p.n_sel, cpt, sub);
#else
            TopkCursor cur = topk_select<NT, FC>(sm, key, n, p.n_sel, cpt);
#endif
#ifdef QUEST_TIMELINE
            for (int i = 0; i < 9; ++i) sub_out[i] = sub[i];
#endif
            QUEST_STAMP(4);
            uint32_t my_slot[FC];
            bool mine[FC];
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                uint32_t slot;
                mine[i] = topk_take(cur, key[i], (uint32_t)i < cpt && c0 + i < n, slot) && slot >= slot_begin && slot < slot_end;
                my_slot[i] = slot;
                if (mine[i]) {
                    // two separate loads: a select between an LDS and a global ADDRESS becomes one flat load
                    // whose address-space cast this compiler miscompiles (illegal v_cmp on src_shared_base)
                    int32_t pg;
                    if (direct) {
                        const uint4 q4 = own_ids[(i >> 2) & 1];
                        pg = (int32_t)((i & 3) == 0 ? q4.x : (i & 3) == 1 ? q4.y : (i & 3) == 2 ? q4.z : q4.w);
                    } else if (stage_ids) {
                        pg = ids_s[c0 + i];
                    } else {
                        pg = table[c0 + i];
                        asm volatile("" : "+v"(pg));
                    }
                    s_sel[slot - slot_begin] = pg;
                }
            }
            __syncthreads();
            QUEST_STAMP(5);
            // optional copy of the selection for callers that inspect it: issued after the barrier so no
            // workgroup waits on these stores before it starts fetching K/V
            if (p.sel_idx_out) {
                const size_t out_row = ((size_t)seq * num_qo_heads + hq) * p.sel_stride;
#pragma unroll
                for (int i = 0; i < FC; ++i)
                    if (mine[i]) {
                        p.sel_idx_out[out_row + my_slot[i]] = s_sel[my_slot[i] - slot_begin];
                        if (p.sel_val_out) p.sel_val_out[out_row + my_slot[i]] = key_to_half_bits(key[i]);
                    }
            }
        }  // n > 0
    }
    float8 qv = to_f32(q_raw);
    qv *= p.scale_log2;
    // physical page of a slot: selected list (global index row, or the LDS list of the fused front end)
    auto slot_page = [&](uint32_t slot) -> int32_t {
        if (slot >= p.n_sel) return p.last_page_idx;
        if constexpr (FC > 0) return s_sel[slot - slot_begin];
        else return idx_row[slot];
    };

    if constexpr (S_T > 0) {
        constexpr int T = (S_T + R - 1) / R;  // load instructions per page per tensor
        const uint32_t step = R * p.st.entry;
        for (uint32_t s0 = slot_begin + wave; s0 < slot_end; s0 += 2 * NW) {
            const uint32_t s1 = s0 + NW;
            const bool has1 = s1 < slot_end;
            const int32_t pg0 = __builtin_amdgcn_readfirstlane(slot_page(s0));
            const int32_t pg1 = __builtin_amdgcn_readfirstlane(has1 ? slot_page(s1) : pg0);
            const int len0 = s0 < p.n_sel ? S_T : (int)p.last_page_len;
            const int len1 = has1 ? (s1 < p.n_sel ? S_T : (int)p.last_page_len) : 0;
            const half_t* b0 = head_base + (size_t)pg0 * p.st.page;
            const half_t* b1 = head_base + (size_t)pg1 * p.st.page;
            half8 k[2 * T], v[2 * T];
            int left[2 * T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                left[t] = len0 - t * R;
                left[T + t] = len1 - t * R;
                // rows past the page's length exist in the pool (the page is allocated) but hold
                // stale bytes; they are fetched only for the sequence's last page and masked in fold
                k[t] = ld8_kv(b0 + lane_off + t * step);
                v[t] = ld8_kv(b0 + lane_off + t * step + p.st.v_off);
            }
            if (has1) {  // wave-uniform
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    k[T + t] = ld8_kv(b1 + lane_off + t * step);
                    v[T + t] = ld8_kv(b1 + lane_off + t * step + p.st.v_off);
                }
                fold_groups<D, 2 * T>(st, qv, k, v, left, row);
            } else {
                half8 k0[T], v0[T];
                int left0[T];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    k0[t] = k[t];
                    v0[t] = v[t];
                    left0[t] = left[t];
                }
                fold_groups<D, T>(st, qv, k0, v0, left0, row);
            }
        }
    } else {
        const uint32_t S = p.page_size;
        for (uint32_t slot = slot_begin + wave; slot < slot_end; slot += NW) {
            const bool sel = slot < p.n_sel;
            const int32_t pg = __builtin_amdgcn_readfirstlane(slot_page(slot));
            const int len = sel ? (int)S : (int)p.last_page_len;
            const half_t* b = head_base + (size_t)pg * p.st.page;
            for (int t0 = 0; t0 < len; t0 += R) {
                half8 k1[1], v1[1];
                const int left[1] = {len - t0};
                // rows past `len` stay inside the (allocated) page for page sizes that are a multiple of R;
                // otherwise clamp to the page's first row -- masked in fold either way
                const uint32_t r_in = (uint32_t)(t0 + row) < S ? (uint32_t)t0 : 0u;
                const half_t* ptr = b + lane_off + (size_t)r_in * p.st.entry - (r_in == (uint32_t)t0 ? 0 : (size_t)row * p.st.entry);
                k1[0] = ld8(ptr);
                v1[0] = ld8(ptr + p.st.v_off);
                fold_groups<D, 1>(st, qv, k1, v1, left, row);
            }
        }
    }

    QUEST_STAMP(6);
    // rows of the wave -> one state (xor butterfly across rows; both partners get the same bits)
    for_each_row_distance<LPR>([&](auto off_c) {
        constexpr int OFF = decltype(off_c)::value;
        const float m_o = lane_xor<OFF>(st.m, lane), d_o = lane_xor<OFF>(st.d, lane);
        const float m_n = __builtin_fmaxf(st.m, m_o);
        const float a = __builtin_amdgcn_exp2f(st.m - m_n), b = __builtin_amdgcn_exp2f(m_o - m_n);
        st.d = st.d * a + d_o * b;
#pragma unroll
        for (int i = 0; i < kVec; ++i) st.acc[i] = st.acc[i] * a + lane_xor<OFF>(st.acc[i], lane) * b;
        st.m = m_n;
    });

    // waves -> workgroup through LDS
    __shared__ float s_acc[NW][D];
    __shared__ float s_md[NW][2];
    if (row == 0) {
#pragma unroll
        for (int i = 0; i < kVec; ++i) s_acc[wave][col * kVec + i] = st.acc[i];
        if (col == 0) {
            s_md[wave][0] = st.m;
            s_md[wave][1] = st.d;
        }
    }
    QUEST_STAMP(7);
    __syncthreads();
    QUEST_STAMP(8);
    const int f = threadIdx.x;
    if (f < D) {
        float M = s_md[0][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = __builtin_fmaxf(M, s_md[w][0]);
        float acc = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float e = __builtin_amdgcn_exp2f(s_md[w][0] - M);
            acc += e * s_acc[w][f];
            den += e * s_md[w][1];
        }
        if (p.n_chunks == 1) {
            sv.o[(size_t)hq * D + f] = (half_t)(acc / den);
            if (QUEST_LSE_ENABLED && p.lse && f == 0) sv.lse[hq] = (M + __builtin_amdgcn_logf(den)) * 0.6931471805599453f;
        } else {
            float* w = sv.ws + ((size_t)hq * p.n_chunks + chunk) * p.ws_stride;
            w[f] = acc;
            if (f == 0) {
                w[D] = M;
                w[D + 1] = den;
            }
        }
    }
#ifdef QUEST_TIMELINE
    QUEST_STAMP(9);
    if (p.lse && chunk == p.n_chunks / 2 && hq == num_qo_heads / 2 && seq == 0 && threadIdx.x == 0) {
        for (int i = 0; i < 10; ++i) p.lse[i] = (float)(tl[i] - tl[0]);
        if constexpr (FC > 0)
            for (int i = 0; i < 9; ++i) p.lse[16 + i] = (float)(sub_out[i] - tl[0]);
        p.lse[10] = (float)(wall_clock64() - wall0);  // 100 MHz ticks over the same span as tl[9] - tl[0]
    }
#endif
}

}  // namespace quest
