// Device code of the sparse paged decode attention: kernel parameters, the online-softmax row state, the merge of the
// per-workgroup partial states (merge_states_kernel's fast path) and the workgroup body of sparse_decode_kernel (sparse_attn.hip, which also holds the reference citations and the design notes).
#pragma once
#include "estimate_device.cuh"
#include "stamps.cuh"
#include "topk_bitmap.cuh"

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
    // Kernarg preload (-amdgpu-kernarg-preload-count, build.py) only covers SCALAR kernel arguments -- a struct passed
    // by value is fetched with s_load whatever its layout (round 2 believed otherwise; the ISA showed three serialised
    // scalar loads before the first vector load).  So the kernels take the fields their first loads need as separate
    // leading arguments (QUEST_DECODE_HEAD_*, 14 dwords = what the hardware preloads beside the kernarg pointer) and
    // copy them over the struct's fields; the rest of the struct arrives by s_load while those first loads fly.
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
    uint32_t row_lead;      // second generation on score rows that are only 2-byte aligned: every workgroup rounds its row
                            // pointer down to 8 bytes and skips the 0-3 leading columns that belong to the previous row
    uint32_t vec_front;     // fused front end: 0 = first generation (topk_select.cuh); 1 = the same, staging arrays fed by
                            // 8/16-byte granule loads; 3 = the same, each thread loads its OWN cpt (4 or 8) columns and
                            // page ids straight into registers (no LDS staging); 2 = second generation
                            // (topk_bitmap.cuh); 8 = tiles (sparse_decode_tiles_body: the rows carry tile maxima).
                            // 1-3 and 8 need aligned score rows
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
    uint32_t fe2_prefilter;  // second-generation front end: histogram only the keys above a per-wave lower bound (topk_bitmap.cuh)
    const int32_t* budgets;  // optional per-sequence page budgets (pages incl. the current one) of a batched launch
    uint32_t table_vec;      // page table(s) 16-byte aligned: a lane's 4 page ids are one load
    // decode append folded into the group-shared (full-KV) launch: shared_decode_kernel<.., APPEND = true>
    const half_t* app_k;     // [n_seqs][kv heads][D]: the new token's key / value, not yet in the pool
    const half_t* app_v;
    half_t* app_meta;        // metadata pool of this layer (same layout / page size as the KV pool)
    uint32_t meta_last_page_len;  // host-planned launches; state-driven ones read both from the step state
    int32_t meta_last_page_idx;
};

// leading scalar kernel arguments (preloaded into SGPRs at wave launch) and their hand-over to the struct; a_pack =
// vec_front | cpt << 4 | stage_ids << 12 | row_lead << 13 | xcd_period << 16 (one dword, so that the query-head count fits as well:
// gridDim.y would be an s_load of the hidden arguments in front of the score-row address)
#define QUEST_DECODE_HEAD_PARAMS                                                                                        \
    const half_t *a_q, const int32_t *a_indices, const uint16_t *a_scores, const quest_step_state_t *a_state,           \
        uint32_t a_n_scores, uint32_t a_score_stride, uint32_t a_table_stride, uint32_t a_pack, uint32_t a_num_qo_heads
#define QUEST_DECODE_HEAD_ARGS(p, num_qo_heads)                                                  \
    (p).q, (p).indices, (p).scores, (p).state, (p).n_scores, (p).score_stride, (p).table_stride, \
        ((p).vec_front | (p).cpt << 4 | ((p).stage_ids ? 1u : 0u) << 12 | ((p).row_lead ? 1u : 0u) << 13 | (p).xcd_period << 16), (uint32_t)(num_qo_heads)
#define QUEST_DECODE_HEAD_TAKE(p)                                                                                    \
    do {                                                                                                             \
        (p).q = a_q, (p).indices = a_indices, (p).scores = a_scores, (p).state = a_state, (p).n_scores = a_n_scores; \
        (p).score_stride = a_score_stride, (p).table_stride = a_table_stride;                                        \
        (p).vec_front = a_pack & 15u, (p).cpt = (a_pack >> 4) & 255u, (p).stage_ids = (a_pack >> 12) & 1u;           \
        (p).row_lead = (a_pack >> 13) & 1u;                                                                          \
        (p).xcd_period = a_pack >> 16;                                                                               \
    } while (0)

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

constexpr int kFusedMaxPpc = 256;  // pages per workgroup the fused front end can stage in LDS (round 6: 128 -> 256, so that a
                                    // token budget of 4096 -- the reference's largest, scripts/passkey.sh:11 -- is one workgroup per head)

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

// ---- merge of the per-chunk partial states of a head (flashinfer's VariableLengthMergeStates, call site
// decode_attn.cuh:992-1001): weights exp2(m_c - M), weighted sum, normalise, cast.
constexpr int kMergeGroups = 4;   // the chunks of a head are dealt round-robin over 4 partial sums per feature
constexpr int kMergePre = 8;      // chunks per partial sum requested up front (one memory round trip)
constexpr uint32_t kMergeFastChunks = 32;  // = kMergePre * kMergeGroups <= kWave: the one-round-trip form below

// Merge of one head with n_chunks <= kMergeFastChunks by NT threads (whole waves; D >= 64 so a wave lies inside one
// feature group).  Every wave derives the chunk weights by itself -- lane c holds chunk c's (m, d) -- so the only
// barrier is the cross-group sum.  The association order is fixed (chunks g, g+4, g+8, ... ascending into partial sum
// g; then ((s0 + s1) + s2) + s3), independent of NT.
// s_red: kMergeGroups * D floats of LDS.
template <int D, int NT>
__device__ __forceinline__ void merge_head_fast(const float* __restrict__ w, half_t* __restrict__ o_row, float* lse_ptr,
                                                uint32_t n_chunks, uint32_t ws_stride, uint32_t tid, float* s_red) {
    constexpr int ITEMS = kMergeGroups * D, PASSES = (ITEMS + NT - 1) / NT;
    float pre[PASSES][kMergePre];
#pragma unroll
    for (int r = 0; r < PASSES; ++r) {
        const uint32_t item = tid + r * NT, f = item % D, g = item / D;
#pragma unroll
        for (int j = 0; j < kMergePre; ++j) {
            const uint32_t c = g + j * kMergeGroups, cc = c < n_chunks ? c : n_chunks - 1;  // clamped: no branch
            pre[r][j] = w[(size_t)cc * ws_stride + f];
        }
    }
    const uint32_t lane = tid & 63, lc = lane < n_chunks ? lane : n_chunks - 1;
    const float m_c = w[(size_t)lc * ws_stride + D], d_c = w[(size_t)lc * ws_stride + D + 1];
    const float Mw = wave_allreduce_max(lane < n_chunks ? m_c : kNegFloor, (int)lane);
    const float e_c = lane < n_chunks ? __builtin_amdgcn_exp2f(m_c - Mw) : 0.f;
    const float dn = wave_allreduce_sum(e_c * d_c, (int)lane);
#pragma unroll
    for (int r = 0; r < PASSES; ++r) {
        const uint32_t item = tid + r * NT, f = item % D;
        if (ITEMS % NT == 0 || item < (uint32_t)ITEMS) {
            const uint32_t g_u = __builtin_amdgcn_readfirstlane(item / D);
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < kMergePre; ++j) {
                const uint32_t c = g_u + j * kMergeGroups;  // wave-uniform: the chunk weight is a scalar broadcast
                const float wc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, e_c),
                                                                                   (int)(c < n_chunks ? c : 0)));
                if (c < n_chunks) a += wc * pre[r][j];
            }
            s_red[g_u * D + f] = a;
        }
    }
    __syncthreads();
    for (uint32_t f = tid; f < (uint32_t)D; f += NT) {
        float tot = s_red[f];
#pragma unroll
        for (int j = 1; j < kMergeGroups; ++j) tot += s_red[j * D + f];
        o_row[f] = (half_t)(tot / dn);
        if (lse_ptr && f == 0) *lse_ptr = (Mw + __builtin_amdgcn_logf(dn)) * 0.6931471805599453f;
    }
}

// Gather of the slots [slot_begin, slot_end) of one head by the NW waves of a workgroup + the workgroup's result: slot s
// is the listed page page_of(s) for s < n_listed, the sequence's current page (p.last_page_idx, p.last_page_len rows)
// otherwise.  Writes o (one workgroup per head) or the workgroup's partial state.
// (The kernel parameters it needs travel BY VALUE in AttendArgs, built field by field by the caller: handing the
// callers' modified copy of DecodeParams over by reference left a 32-byte slice of it in memory, which the compiler then
// "promoted" to LDS -- 16 KiB per workgroup, one workgroup per CU instead of two, 24 us instead of 12 at cfg 3.)
struct AttendArgs {
    const half_t* kv;
    PoolStrides st;
    uint32_t group, last_page_len;
    int32_t last_page_idx;
    uint32_t page_size, n_chunks, ws_stride;
    float scale_log2;
    float* lse;
};
__device__ __forceinline__ AttendArgs attend_args(const DecodeParams& p) {
    AttendArgs a;
    a.kv = p.kv;
    a.st.page = p.st.page, a.st.v_off = p.st.v_off, a.st.head = p.st.head, a.st.entry = p.st.entry;
    a.st.rot = p.st.rot, a.st.vflip = p.st.vflip;
    a.group = p.group, a.last_page_len = p.last_page_len, a.last_page_idx = p.last_page_idx;
    a.page_size = p.page_size, a.n_chunks = p.n_chunks, a.ws_stride = p.ws_stride;
    a.scale_log2 = p.scale_log2;
    a.lse = p.lse;
    return a;
}
// APPEND (layer_decode_kernel, layer_device.cuh): the decode append rides in the gather.  The new token's row of the
// sequence's current page is not in the pool yet: the wave that folds the current page takes it from the inputs (k / v of
// this kv head) instead, and the ONE workgroup per kv head marked `writer` stores it -- and the page's folded (max, min)
// metadata entry -- after the page loop (stores in front of it would sit in the wave's vmcnt queue ahead of its K/V
// loads).  `mx` / `mn`: the metadata entry as it was before this token (requested at kernel entry, unconditionally).
struct AppendRow {
    const half_t* k;   // the kv head's new key / value [D]
    const half_t* v;
    half_t* meta_entry;  // the current page's metadata entry of this kv head (max; min is meta_v_off halves further)
    uint32_t meta_v_off;
    ushort8 mx, mn;
    bool writer;
};
struct NoAppend {};
template <int D, int S_T, int NW, bool APPEND = false, typename PageOf, typename App = NoAppend>
__device__ __forceinline__ void attend_slots(const AttendArgs p, const SeqView sv, const half8 q_raw, const uint32_t chunk,
                                             const uint32_t hq, const uint32_t slot_begin, const uint32_t slot_end,
                                             const uint32_t n_listed, const int wave, const int lane,
                                             PageOf page_of QUEST_TL_PARAM QUEST_WS_PARAM, const App app = App{}) {
    constexpr int LPR = D / kVec, R = kWave / LPR;
    const int row = lane / LPR, col = lane % LPR;
    RowState<D> st;
    const uint32_t hk = hq / p.group;
    // address of (page, entry t R + row) of this kv head = pool + page * st.page + uni[t] + lane_off (+ v_off for V): the head
    // term is split between the two so that the row-rotated pool (st.rot, quest_common.cuh walk_*) walks like the plain ones
    const uint32_t lane_off = walk_lane_off(p.st, hk, R, row, col * kVec);  // per lane, loop invariant
    const uint32_t v_off = pool_v_off(p.st, hk);                              // uniform
    float8 qv = to_f32(q_raw);
    qv *= p.scale_log2;
    [[maybe_unused]] half8 app_kn, app_vn;
    [[maybe_unused]] const uint32_t app_e = p.last_page_len - 1u;  // row of the new token in the current page
    if constexpr (APPEND) {
        static_assert(S_T > 0, "the fused append serves the compile-time page size");
        app_kn = ld8(app.k + col * kVec);
        app_vn = ld8(app.v + col * kVec);
    }
    // physical page of a slot: a listed page, or the sequence's current page for slots at or beyond n_listed
    auto slot_page = [&](uint32_t slot) -> int32_t {
        if (slot >= n_listed) return p.last_page_idx;
        return page_of(slot);
    };

    if constexpr (S_T > 0) {
        constexpr int T = (S_T + R - 1) / R;  // load instructions per page per tensor
        uint32_t uni[T];                        // uniform, loop invariant
#pragma unroll
        for (int t = 0; t < T; ++t) uni[t] = walk_uniform(p.st, hk, R, t);
        for (uint32_t s0 = slot_begin + wave; s0 < slot_end; s0 += 2 * NW) {
            const uint32_t s1 = s0 + NW;
            const bool has1 = s1 < slot_end;
            const int32_t pg0 = __builtin_amdgcn_readfirstlane(slot_page(s0));
            const int32_t pg1 = __builtin_amdgcn_readfirstlane(has1 ? slot_page(s1) : pg0);
            const int len0 = s0 < n_listed ? S_T : (int)p.last_page_len;
            const int len1 = has1 ? (s1 < n_listed ? S_T : (int)p.last_page_len) : 0;
            const half_t* b0 = p.kv + (size_t)pg0 * p.st.page;
            const half_t* b1 = p.kv + (size_t)pg1 * p.st.page;
            half8 k[2 * T], v[2 * T];
            int left[2 * T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                left[t] = len0 - t * R;
                left[T + t] = len1 - t * R;
                // rows past the page's length exist in the pool (the page is allocated) but hold
                // stale bytes; they are fetched only for the sequence's last page and masked in fold
                k[t] = ld8_kv(b0 + uni[t] + lane_off);
                v[t] = ld8_kv(b0 + uni[t] + lane_off + v_off);
            }
            if (has1) {  // wave-uniform (a likely-taken hint here cost 3.5 us per batched launch: measured, r3j)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    k[T + t] = ld8_kv(b1 + uni[t] + lane_off);
                    v[T + t] = ld8_kv(b1 + uni[t] + lane_off + v_off);
                }
                if constexpr (APPEND) {
                    if (s0 >= n_listed || s1 >= n_listed) {  // wave-uniform: one of the two is the current page
                        const int at = s0 >= n_listed ? 0 : T;
#pragma unroll
                        for (int t = 0; t < 2 * T; ++t)
                            if (t >= at && t < at + T && (uint32_t)((t - at) * R + row) == app_e) k[t] = app_kn, v[t] = app_vn;
                    }
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
                if constexpr (APPEND) {
                    if (s0 >= n_listed) {
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            if ((uint32_t)(t * R + row) == app_e) k0[t] = app_kn, v0[t] = app_vn;
                    }
                }
                fold_groups<D, T>(st, qv, k0, v0, left0, row);
            }
        }
    } else {
        const uint32_t S = p.page_size;
        for (uint32_t slot = slot_begin + wave; slot < slot_end; slot += NW) {
            const bool sel = slot < n_listed;
            const int32_t pg = __builtin_amdgcn_readfirstlane(slot_page(slot));
            const int len = sel ? (int)S : (int)p.last_page_len;
            const half_t* b = p.kv + (size_t)pg * p.st.page;
            for (int t0 = 0; t0 < len; t0 += R) {
                half8 k1[1], v1[1];
                const int left[1] = {len - t0};
                // rows past `len` stay inside the (allocated) page for page sizes that are a multiple of R;
                // otherwise clamp to the page's first row -- masked in fold either way
                const uint32_t e = (uint32_t)(t0 + row) < S ? (uint32_t)(t0 + row) : 0u;
                const half_t* ptr = b + (size_t)e * p.st.entry + (size_t)pool_slot(p.st, hk, e) * p.st.head + col * kVec;
                k1[0] = ld8(ptr);
                v1[0] = ld8(ptr + v_off);
                fold_groups<D, 1>(st, qv, k1, v1, left, row);
            }
        }
    }

    if constexpr (APPEND) {
        // the wave that folded the current page (slot n_listed), the row of lanes at the new token's position
        if (app.writer && slot_begin <= n_listed && n_listed < slot_end && (uint32_t)wave == (n_listed - slot_begin) % NW &&
            (uint32_t)row == app_e % R) {
            half_t* dst = const_cast<half_t*>(p.kv) + (size_t)p.last_page_idx * p.st.page + lane_off + walk_uniform(p.st, hk, R, app_e / R);
            st8(dst, app_kn);
            st8(dst + v_off, app_vn);
            const ushort8 k8 = __builtin_bit_cast(ushort8, app_kn);
            // a token that opens a page starts from the sentinels, not from stale pool bytes
            const ushort8 mx0 = app_e > 0 ? app.mx : (ushort8)(kHalfNegMax), mn0 = app_e > 0 ? app.mn : (ushort8)(kHalfMax);
            uint16_t* me = reinterpret_cast<uint16_t*>(app.meta_entry) + col * kVec;
            *reinterpret_cast<ushort8*>(me) = fold_max(mx0, k8);
            *reinterpret_cast<ushort8*>(me + app.meta_v_off) = fold_min(mn0, k8);
        }
    }
    QUEST_STAMP(6);
    QUEST_WS_NOW(ws_gather)
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
            // (a wall-stamp build allocates the workspace for one-chunk plans too)
            if (f == 0) QUEST_WS_RECORD(sv.ws ? sv.ws + (size_t)hq * p.ws_stride : nullptr, D, p.ws_stride, slot_end > slot_begin ? slot_end - slot_begin : 0u);
        } else {
            float* w = sv.ws + ((size_t)hq * p.n_chunks + chunk) * p.ws_stride;
            w[f] = acc;
            if (f == 0) {
                w[D] = M;
                w[D + 1] = den;
                QUEST_WS_RECORD(w, D, p.ws_stride, slot_end > slot_begin ? slot_end - slot_begin : 0u);
            }
        }
    }
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
// (chunk, hq, seq) are the workgroup's coordinates.
// VF >= 0: the front-end variant (DecodeParams.vec_front) is a compile-time constant and the other variants' code is
// not part of the kernel (the generic fused kernel carries all four: 27 KiB of code against 11 for one).
template <int D, int S_T, int FC, int NW, int VF = -1>
__device__ __forceinline__ void sparse_decode_body(DecodeParams p, const uint32_t chunk, const uint32_t hq, const uint32_t seq,
                                                   const uint32_t num_qo_heads) {
    constexpr int LPR = D / kVec;
    const uint32_t vec_front = VF >= 0 ? (uint32_t)VF : p.vec_front;
    // wave index as an SGPR so per-wave control flow below is scalar branching
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int col = lane % LPR;
    // Source order matters up to the first loads: everything above them uses only the preloaded arguments (see
    // DecodeParams); values that need the rest of the struct (kv head, slots, pool strides) are derived after them.
    QUEST_TL_BEGIN
    QUEST_WS_ENTRY
    const SeqView sv = select_sequence(p, num_qo_heads, D, seq);
    // state-driven launches pass the longest row the graph will see in p.n_scores (it sizes FC); the live
    // row length comes from the state
    if (FC == 0 && p.state) {  // no front end, state-driven: full-KV decode of shapes the group-shared kernel
                               // does not cover (shared_entry's fallback); the list is the page table itself
        const quest_step_state_t st = *sv.state;
        p.n_scores = (uint32_t)(st.n_pages - 1);
        if (p.budgets) p.n_sel = min(p.n_sel, (uint32_t)max(p.budgets[seq] - 1, 0));
        p.n_sel = min(p.n_sel, p.n_scores);
        p.last_page_len = (uint32_t)st.kv_last_page_len;
        p.last_page_idx = st.kv_last_page_idx;
    }
    // slots = selected pages + the current page; a state-driven launch on a sequence still shorter than the
    // budget selects ALL of its pages (k = n: the reference's full-attention branch, QuestAttention.py:123-132)
    // and workgroups whose chunk lies past the live list write an empty partial (weight 0 in the merge)
    uint32_t n_slots = 0, slot_begin = 0, slot_end = 0;
    auto plan_slots = [&]() {
        n_slots = p.n_sel + 1;
        slot_begin = chunk * p.pages_per_chunk;
        slot_end = min(n_slots, slot_begin + p.pages_per_chunk);
    };
    if constexpr (FC == 0) plan_slots();

    // q is requested now but first used after the top-k front end, so its latency hides under the selection
    const half8 q_raw = ld8(sv.q + (size_t)hq * D + col * kVec);

    __shared__ int32_t s_sel[FC > 0 ? kFusedMaxPpc : 1];
    if constexpr (FC > 0) {
        __shared__ TopkSmem<NW * kWave> sm;
        const uint32_t n_cap = p.n_scores;  // as launched: the longest row this launch may see (buffers cover it)
        constexpr int NT = NW * kWave;
        __shared__ uint32_t s_bm[2][kBmWords];
        extern __shared__ __attribute__((aligned(16))) unsigned char fe_dyn[];
        const uint16_t* srow = sv.scores + (size_t)hq * p.score_stride;
        // (second generation on 2-byte aligned rows: the aligned stream below the row, see topk_bitmap.cuh)
        const uint32_t lead = p.row_lead ? (uint32_t)((reinterpret_cast<uintptr_t>(srow) >> 1) & 3u) : 0u;
        Fe2Raw<fe2_has_ids(FC)> raw[FC / 4];
        // live lengths of a state-driven launch: ONE scalar load (n_pages, last page's length and id are adjacent),
        // issued before the vector loads below and consumed after them
        // (unconditional, from q's bytes when there is no state: a load under a branch is waited for at the join)
        const int4 live = *(p.state ? reinterpret_cast<const int4*>(sv.state)
                                    : reinterpret_cast<const int4*>(p.q));  // seq_len, n_pages, kv_last_page_len, kv_last_page_idx

        uint4 own_keys = make_uint4(0u, 0u, 0u, 0u), own_ids[2] = {own_keys, own_keys};  // vec_front == 3
        const bool own_cols = FC == 8 && vec_front == 3;
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
        if (own_cols) {
            {   // two UNCONDITIONAL 8-byte loads (the second one repeats the first when the thread owns 4 columns): a
                // 16-byte load under `if (cpt == 8)` was compiled as "upper half under the branch, s_waitcnt vmcnt(0) at
                // the join, then the lower half" -- two serialised round trips in front of the selection at every shape
                // whose capacity exceeds 2048 pages (the bench's: cpt = 8)
                const uint2 lo = *reinterpret_cast<const uint2*>(srow + own_cc);
                const uint2 hi = *reinterpret_cast<const uint2*>(srow + own_cc + (p.cpt == 8 ? 4u : 0u));
                own_keys = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
            fe2_clear<NT>(sm);
        } else if (vec_front) {  // loads first: their addresses depend on the capacity only, not on the state below
            fe2_issue<NT, FC / 4, fe2_has_ids(FC)>(srow - lead, sv.indices, n_cap + 1u, p.stage_ids != 0, n_cap + lead, raw);
            fe2_clear<NT>(sm);
        }
        // the per-sequence budget of a batched launch (from q's first bytes when there is none): requested here, after the
        // vector loads above have left, so that its round trip -- the pointer sits at the end of the argument struct --
        // runs under theirs instead of being fetched and waited for at its use
        const int32_t budget_raw = ld_uniform_i32(p.budgets ? p.budgets + seq : reinterpret_cast<const int32_t*>(p.q));
        if (p.state) {  // live lengths
            p.n_scores = (uint32_t)(live.y - 1);
            p.last_page_len = (uint32_t)live.z;
            p.last_page_idx = live.w;
            if (p.budgets) p.n_sel = min(p.n_sel, (uint32_t)max(budget_raw - 1, 0));
            p.n_sel = min(p.n_sel, p.n_scores);
        }
        plan_slots();
        const uint32_t n = p.n_scores;
        const bool second_gen = vec_front == 2;
        if (n > 0 && second_gen) {
            QUEST_STAMP(1);
            const size_t out_row = ((size_t)seq * num_qo_heads + hq) * p.sel_stride;
            const bool ids_staged = p.stage_ids && fe2_has_ids(FC) && vec_front == 2;
            uint16_t* val_row = p.sel_val_out ? p.sel_val_out + out_row : nullptr;
            int32_t* idx_row_out = p.sel_idx_out ? p.sel_idx_out + out_row : nullptr;
            fe2_select<NT, FC>(sm, s_bm, raw, srow - lead, sv.indices - lead,
                               ids_staged ? reinterpret_cast<int32_t*>(fe_dyn + p.ids_lds_offset) : nullptr, n_cap + lead, n,
                               p.n_sel, slot_begin, slot_end, s_sel, val_row, idx_row_out, p.fe2_prefilter != 0, QUEST_TL_SUB,
                               lead);
            QUEST_STAMP(4);
            __syncthreads();
            if (!ids_staged) {  // block-uniform: columns (positions of the aligned stream) -> pages, one parallel round trip
                fe2_resolve_pages(srow - lead, sv.indices - lead, slot_begin, slot_end, p.n_sel, s_sel, val_row, idx_row_out);
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
            } else if (vec_front == 1) {
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
            TopkCursor cur = topk_select<NT, FC>(sm, key, n, p.n_sel, cpt, QUEST_TL_SUB);
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
    const int32_t* idx_row = sv.indices + (size_t)hq * p.idx_stride;  // uniform
    attend_slots<D, S_T, NW>(attend_args(p), sv, q_raw, chunk, hq, slot_begin, slot_end, p.n_sel, wave, lane, [&](uint32_t slot) -> int32_t {
        if constexpr (FC > 0) return s_sel[slot - slot_begin];
        else return idx_row[slot];
    } QUEST_TL_ARG QUEST_WS_ARG);
    QUEST_TL_REPORT(p, chunk, hq, seq, num_qo_heads);
}

// ---- tiles front end (round 5): long score rows whose producer also left TILE MAXIMA -- per run of 8 columns the largest
// key of the run (quest_append_estimate_tiles_dyn: the estimate workgroup that holds those 8 scores writes it beside the
// row).  Two passes of the SAME selection routine instead of four issue-bound passes over 16-32 keys per thread:
//   1. top-k1 of the row's ceil(n / 8) tile keys, k1 = min(k, tiles), under the declared tie rule (lowest tile first among
//      equal maxima).  These k1 tiles are a sufficient candidate set: with B = the k1-th largest maximum, at least k scores
//      reach B (one per selected tile), so the threshold T >= B; every score > B lies in a tile whose maximum is > B (all of
//      them selected); and the ties the rule needs at T == B are taken lowest column first, of which the a tiles with a
//      maximum > B hold at least a scores above B -- at most k - a tied scores are needed, and the first k - a tied tiles in
//      column order already hold that many at lower columns than any later tile.
//   2. the k1 candidate tiles arrive compacted in ascending column order (a tile's slot of pass 1 = its rank); thread j takes
//      half a tile (tile j / 2, columns 4 (j & 1) ..): one 8-byte load of scores and one 16-byte load of page ids, then the
//      short-row selection over <= 4 keys per thread with the validity mask -- same threshold, same ties, same slots as the
//      selection over the whole row (tests/test_gpu_long_rows.py, test_gpu_full_size.py cfg 4).
// Serves k <= 256 (two threads per candidate tile at 512 threads) and rows up to 16384 columns (4 tile keys per thread).
constexpr int kTileCols = 8;
template <int D, int NW>
__device__ __forceinline__ void sparse_decode_tiles_body(DecodeParams p, const uint32_t chunk, const uint32_t hq,
                                                         const uint32_t seq, const uint32_t num_qo_heads) {
    constexpr int LPR = D / kVec, NT = NW * kWave, TM = 4, KC = 4;
    static_assert(NT == 512, "two threads per candidate tile, k <= 256");
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int col = lane % LPR;
    const uint32_t tid = threadIdx.x;
    QUEST_TL_BEGIN
    QUEST_WS_ENTRY
    const SeqView sv = select_sequence(p, num_qo_heads, D, seq);
    const half8 q_raw = ld8(sv.q + (size_t)hq * D + col * kVec);  // first used after the selection
    __shared__ TopkSmem<NT> sm;
    __shared__ int32_t s_sel[kFusedMaxPpc];
    __shared__ uint32_t s_tiles[NT / 2];
    const uint32_t n_cap = p.n_scores, tiles_cap = (n_cap + kTileCols - 1) / kTileCols;
    const uint16_t* srow = sv.scores + (size_t)hq * p.score_stride;
    const uint16_t* trow = srow + ((n_cap + 7u) & ~7u);  // the maxima follow the scores padded to 8 columns (host-checked)
    const int4 live = *(p.state ? reinterpret_cast<const int4*>(sv.state) : reinterpret_cast<const int4*>(p.q));
    // pass-1 loads: the thread's own p.cpt (2 or 4) tile keys, two unconditional 4-byte loads (the second repeats the
    // first at 2 keys per thread); addresses depend on the capacity only
    const uint32_t cpt_t = p.cpt, t0 = tid * cpt_t, t0c = t0 < tiles_cap ? t0 : 0u;
    const uint32_t tlo = *reinterpret_cast<const uint32_t*>(trow + t0c);
    const uint32_t thi = *reinterpret_cast<const uint32_t*>(trow + t0c + (cpt_t == 4u ? 2u : 0u));
    topk_clear<NT>(sm);
    const int32_t budget_raw = ld_uniform_i32(p.budgets ? p.budgets + seq : reinterpret_cast<const int32_t*>(p.q));
    if (p.state) {
        p.n_scores = (uint32_t)(live.y - 1);
        p.last_page_len = (uint32_t)live.z;
        p.last_page_idx = live.w;
        if (p.budgets) p.n_sel = min(p.n_sel, (uint32_t)max(budget_raw - 1, 0));
        p.n_sel = min(p.n_sel, p.n_scores);
    }
    const uint32_t n = p.n_scores;
    const uint32_t n_slots = p.n_sel + 1, slot_begin = chunk * p.pages_per_chunk;
    const uint32_t slot_end = min(n_slots, slot_begin + p.pages_per_chunk);
    if (n > 0 && p.n_sel > 0) {  // block-uniform
        const uint32_t n_tiles = (n + kTileCols - 1) / kTileCols, k1 = min(p.n_sel, n_tiles);
        // ---- pass 1: the k1 tiles with the largest maxima
        uint32_t tkey[TM] = {tlo & 0xffffu, tlo >> 16, thi & 0xffffu, thi >> 16};
        uint32_t mm = kMmNeutral;
#pragma unroll
        for (int i = 0; i < TM; ++i)
            if ((uint32_t)i < cpt_t && t0 + i < n_tiles) mm = pk_max_u16(mm, mm_pack(tkey[i]));
        QUEST_STAMP(1);
        topk_publish_range<NT>(sm, mm);
        __syncthreads();
        TopkCursor cur = topk_select<NT, TM>(sm, tkey, n_tiles, k1, cpt_t, QUEST_TL_SUB);
        QUEST_STAMP(2);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            uint32_t slot;
            if (topk_take(cur, tkey[i], (uint32_t)i < cpt_t && t0 + i < n_tiles, slot)) s_tiles[slot] = t0 + i;
        }
        topk_clear<NT>(sm);  // (pass 1 has read its histograms: its last barrier lies behind us)
        __syncthreads();
        QUEST_STAMP(3);
        // ---- pass 2: half a candidate tile per thread
        const uint32_t cand = tid >> 1;
        const bool active = cand < k1;
        const uint32_t c0 = s_tiles[active ? cand : 0u] * kTileCols + KC * (tid & 1u);  // < n_cap rounded up to 8
        const uint2 kraw = *reinterpret_cast<const uint2*>(srow + c0);
        uint4 ids;
        const uint32_t table_len = n_cap + 1u;
        if (p.table_vec && c0 + 4u <= table_len) {
            ids = *reinterpret_cast<const uint4*>(sv.indices + c0);
        } else {  // unaligned tables, or the tail of a table whose length is not a multiple of 4
            const uint32_t last = table_len - 1u;
            ids.x = (uint32_t)sv.indices[c0 < last ? c0 : last];
            ids.y = (uint32_t)sv.indices[c0 + 1u < last ? c0 + 1u : last];
            ids.z = (uint32_t)sv.indices[c0 + 2u < last ? c0 + 2u : last];
            ids.w = (uint32_t)sv.indices[c0 + 3u < last ? c0 + 3u : last];
        }
        uint32_t key[KC] = {half_key((uint16_t)(kraw.x & 0xffffu)), half_key((uint16_t)(kraw.x >> 16)),
                            half_key((uint16_t)(kraw.y & 0xffffu)), half_key((uint16_t)(kraw.y >> 16))};
        auto valid = [&](int i) { return active && c0 + (uint32_t)i < n; };
        mm = kMmNeutral;
#pragma unroll
        for (int i = 0; i < KC; ++i)
            if (valid(i)) mm = pk_max_u16(mm, mm_pack(key[i]));
        topk_publish_range<NT>(sm, mm);
        __syncthreads();
        QUEST_STAMP(4);
        TopkCursor cur2 = topk_select_v<NT, KC>(sm, key, valid, p.n_sel);
        const size_t out_row = ((size_t)seq * num_qo_heads + hq) * p.sel_stride;
        const uint32_t idw[KC] = {ids.x, ids.y, ids.z, ids.w};
#pragma unroll
        for (int i = 0; i < KC; ++i) {
            uint32_t slot;
            if (topk_take(cur2, key[i], valid(i), slot)) {
                if (slot >= slot_begin && slot < slot_end) s_sel[slot - slot_begin] = (int32_t)idw[i];
                if (p.sel_idx_out && chunk == 0) {  // the inspection copy is written once per head
                    p.sel_idx_out[out_row + slot] = (int32_t)idw[i];
                    if (p.sel_val_out) p.sel_val_out[out_row + slot] = key_to_half_bits(key[i]);
                }
            }
        }
    }
    __syncthreads();
    QUEST_STAMP(5);
    attend_slots<D, 16, NW>(attend_args(p), sv, q_raw, chunk, hq, slot_begin, slot_end, p.n_sel, wave, lane,
                            [&](uint32_t slot) -> int32_t { return s_sel[slot - slot_begin]; } QUEST_TL_ARG QUEST_WS_ARG);
    QUEST_TL_REPORT(p, chunk, hq, seq, num_qo_heads);
}

}  // namespace quest
