// One launch per layer of a batched decode step: a workgroup owns one (sequence, query head) END TO END --
//   append (QuestAttention.py:106, decode_page.cuh:398-449) -> page-criticality estimate (:136, decode_attn.cuh:245-401,
//   arithmetic :137-168) -> top-k (:144, decode_select_k.cuh:25-62) -> sparse attention over the selected pages + the
//   current one (:147-157, decode_attn.cuh:440-646)
// -- with NO cross-workgroup dependency, so there is nothing to hand over: the head's page scores never leave the CU
// (they go to LDS as order-preserving 16-bit keys, 4 KiB at 2047 pages, instead of to a global score row), the selection
// reads them from there, the gather follows.  One ramp and one drain per layer instead of two of each, no score scratch
// round trip, no second dispatch.
//
// When it applies: the launch must fill the chip with (sequence, head) pairs -- the planner's one-workgroup-per-head
// split (8 sequences x 32 heads = 256 workgroups = one per CU).  A single sequence (32 workgroups) keeps the two tiled
// launches (estimate_kernel + sparse_decode_kernel), where the estimate is spread over 1024 workgroups.
//
// Same bits as the two launches: the estimate is estimate_tile's arithmetic per (entry, head) row -- per lane 8 features
// left to right in fp32, the row_ror 8/4/2/1 tree, one RNE cast (oracle qo_estimate) --, the selection is topk_select
// (the routine of the stand-alone top-k kernel and of every fused front end), the gather is attend_slots with the plan's
// one-workgroup-per-head split.  The estimate phase has the gather's own memory shape: a metadata page IS a page whose
// 16 "tokens" are the (max, min) entries of 16 KV pages (controller.py:29-37), so a wave streams two metadata pages per
// round with the same 16 x 1 KiB loads in flight and the same SGPR-base + fixed-lane-offset addressing.
#pragma once
#include "decode_device.cuh"

namespace quest {

struct LayerParams {
    const half_t* q;                  // [n_seqs][Hq][D]
    const quest_step_state_t* state;  // [n_seqs]
    const half_t* meta;               // metadata pool of this layer (same geometry as the KV pool)
    const int32_t* meta_tables;       // [n_seqs][meta_table_stride]
    const int32_t* kv_tables;         // [n_seqs][kv_table_stride]
    uint32_t n_cap;                   // longest score row the launch may see (page capacity - 1): sizes the LDS arrays
    uint32_t meta_table_stride, kv_table_stride;
    uint32_t cpt;                     // selection: columns per thread (thread t owns [t*cpt, t*cpt + cpt)), from n_cap
    uint32_t xcd_period;              // see sparse_decode_kernel
    // ---- beyond the preloaded block
    const half_t* k_new;              // [n_seqs][Hkv][D] the token being decoded (not in the pool yet)
    const half_t* v_new;
    half_t* kv;                       // KV pool of this layer
    half_t* o;                        // [n_seqs][Hq][D]
    float* lse;                       // optional [n_seqs][Hq]
    float* ws;                        // wall-stamp builds only
    const int32_t* budgets;           // optional per-sequence page budgets (pages incl. the current one)
    uint16_t* scores_out;             // optional inspection copy of the page scores [n_seqs][Hq][score_stride] fp16
    uint16_t* sel_val_out;            // optional inspection copies of the selection [n_seqs][Hq][sel_stride]
    int32_t* sel_idx_out;
    PoolStrides st;
    uint32_t score_stride, sel_stride;
    uint32_t n_sel;                   // the plan's selected-page count (budget - 1)
    uint32_t group;                   // query heads per kv head
    uint32_t num_kv_heads;
    uint32_t ids_lds_offset;          // byte offset of the staged page table behind the keys
    uint32_t ws_stride;
    float scale_log2;
};

typedef int int4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int4v ld_uniform_i32x4(const void* p) {  // one s_load_dwordx4 of a wave-uniform address
    typedef const int4v __attribute__((address_space(4))) * cptr;
    return *(cptr)(uintptr_t)p;
}

// fp16 bits of one page score from the row's 16 lanes (estimate_tile's arithmetic, estimate_device.cuh: the monotone form
// for finite non-zero q, the literal form of decode_attn.cuh:152-156 otherwise -- bit-identical where both apply)
template <int LPR>
__device__ __forceinline__ uint16_t page_score_bits(const half8 mx, const half8 mn, const float8 qf, const uint4 nm,
                                                    const bool literal) {
    float acc = 0.f;
    if (QUEST_LIKELY(!literal)) {
        const half8 hi = __builtin_elementwise_max(mx, mn), lo = __builtin_elementwise_min(mx, mn);
        const uint4 hb = __builtin_bit_cast(uint4, hi), lb = __builtin_bit_cast(uint4, lo);
        uint4 sb;  // per 16-bit feature: q < 0 ? lo : hi
        sb.x = (lb.x & nm.x) | (hb.x & ~nm.x);
        sb.y = (lb.y & nm.y) | (hb.y & ~nm.y);
        sb.z = (lb.z & nm.z) | (hb.z & ~nm.z);
        sb.w = (lb.w & nm.w) | (hb.w & ~nm.w);
        const float8 x = to_f32(__builtin_bit_cast(half8, sb));
#pragma unroll
        for (int i = 0; i < kVec; ++i) acc = __builtin_fmaf(qf[i], x[i], acc);
    } else {
        const float8 a = to_f32(mx), b = to_f32(mn);
#pragma unroll
        for (int i = 0; i < kVec; ++i) acc += __builtin_fmaxf(qf[i] * a[i], qf[i] * b[i]);
    }
    acc = row_allreduce_sum_fast<LPR>(acc);
    return half_bits((half_t)acc);
}

// Leading scalar kernel arguments = what the first loads need (preloaded into SGPRs at wave launch, see DecodeParams).
#define QUEST_LAYER_HEAD_PARAMS                                                                                            \
    const half_t *a_q, const quest_step_state_t *a_state, const half_t *a_meta, const int32_t *a_meta_tables,              \
        const int32_t *a_kv_tables, uint32_t a_n_cap, uint32_t a_meta_table_stride, uint32_t a_kv_table_stride,            \
        uint32_t a_pack, uint32_t a_num_qo_heads
#define QUEST_LAYER_HEAD_ARGS(p, num_qo_heads)                                                                         \
    (p).q, (p).state, (p).meta, (p).meta_tables, (p).kv_tables, (p).n_cap, (p).meta_table_stride, (p).kv_table_stride, \
        ((p).cpt | (p).xcd_period << 8 | (p).group << 16), (uint32_t)(num_qo_heads)

template <int D, int FC, int NW>
__device__ __forceinline__ void layer_decode_body(QUEST_LAYER_HEAD_PARAMS, const LayerParams& p) {
    constexpr int LPR = D / kVec, R = kWave / LPR, S_T = 16, T = S_T / R, NT = NW * kWave;
    static_assert(D == 128 || D == 64, "row = 16 or 8 lanes");
    const uint32_t cpt = a_pack & 255u, xcd_period = (a_pack >> 8) & 255u, group = a_pack >> 16;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = lane / LPR, col = lane % LPR;
    const uint32_t tid = threadIdx.x, seq = blockIdx.z;
    uint32_t hq = blockIdx.y;
    if (xcd_period > 1) hq = (hq % xcd_period) * (a_num_qo_heads / xcd_period) + hq / xcd_period;
    const uint32_t hk = hq / group;
    QUEST_TL_BEGIN
    QUEST_WS_ENTRY

    // ---- everything the first loads need comes from the preloaded arguments
    const quest_step_state_t* stp = a_state + seq;
    const int4v live = ld_uniform_i32x4(stp);                                      // seq_len, n_pages, kv last len / idx
    const int4v live_m = ld_uniform_i32x4(reinterpret_cast<const int4*>(stp) + 1);  // n_meta_pages, meta last len / idx
    const half8 q_raw = ld8(a_q + ((size_t)seq * a_num_qo_heads + hq) * D + col * kVec);
    const int32_t* kv_table = a_kv_tables + (size_t)seq * a_kv_table_stride;
    const int32_t* meta_table = a_meta_tables + (size_t)seq * a_meta_table_stride;

    extern __shared__ __attribute__((aligned(16))) unsigned char layer_dyn[];
    __shared__ TopkSmem<NT> sm;
    __shared__ int32_t s_sel[kFusedMaxPpc];
    uint16_t* keys_s = reinterpret_cast<uint16_t*>(layer_dyn);
    int32_t* ids_s = reinterpret_cast<int32_t*>(layer_dyn + p.ids_lds_offset);

    // the sequence's page table -> LDS: the n_cap entries a selected COLUMN can name (columns < n <= n_cap; the current page,
    // table entry n, comes from the step state, never from LDS) -- n_cap <= FC * NT by the host's check, so FC rounds cover
    // it; addresses depend on the capacity only: clamped, unconditional
    const uint32_t n_cap = a_n_cap, table_len = n_cap;
    int32_t iraw[FC];
    const uint32_t id_rounds = (table_len + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < FC; ++i) {
        iraw[i] = 0;
        if ((uint32_t)i < id_rounds) {
            const uint32_t e = tid + i * NT;
            iraw[i] = kv_table[e < table_len ? e : table_len - 1u];
        }
    }
    topk_clear<NT>(sm);

    const uint32_t n = (uint32_t)(live.y - 1);  // pages to score: all but the current one
    uint32_t n_sel = p.n_sel;
    if (p.budgets) n_sel = min(n_sel, (uint32_t)max(ld_uniform_i32(p.budgets + seq) - 1, 0));
    n_sel = min(n_sel, n);

    // the current page's metadata entry of this kv head, as it is before this token (the append's read half): every lane
    // asks for its 16 bytes of the entry (the rows of a wave repeat each other) -- unconditional, held until the stores
    // (metadata and KV pools share their geometry: the same walk -- quest_common.cuh walk_* -- serves both phases)
    const uint32_t lane_off = walk_lane_off(p.st, hk, R, row, col * kVec);
    const uint32_t v_off = pool_v_off(p.st, hk);
    const uint32_t m_entry = (uint32_t)(live_m.y - 1);
    half_t* meta_entry = const_cast<half_t*>(a_meta) + (size_t)live_m.z * p.st.page + (size_t)m_entry * p.st.entry +
                         (size_t)pool_slot(p.st, hk, m_entry) * p.st.head;
    AppendRow app;
    app.mx = *reinterpret_cast<const ushort8*>(reinterpret_cast<const uint16_t*>(meta_entry) + col * kVec);
    app.mn = *reinterpret_cast<const ushort8*>(reinterpret_cast<const uint16_t*>(meta_entry) + v_off + col * kVec);
    app.meta_entry = meta_entry;
    app.meta_v_off = v_off;
    app.k = p.k_new + ((size_t)seq * p.num_kv_heads + hk) * D;
    app.v = p.v_new + ((size_t)seq * p.num_kv_heads + hk) * D;
    app.writer = hq % group == 0;

    // ---- estimate: wave w streams the metadata pages w, w + NW, ... (two per round), scores -> keys in LDS
    const float8 qf = to_f32(q_raw);
    uint4 nm;  // per 16-bit feature: q < 0 ? 0xffff : 0
    bool odd_q = false;  // a zero or non-finite query element (the monotone form needs finite non-zero q)
    {
        const uint4 w = __builtin_bit_cast(uint4, q_raw);
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
        uint32_t m[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            m[i] = ((ww[i] >> 15) & 0x00010001u) * 0xffffu;
            odd_q |= ((ww[i] & 0x7fffu) - 1u >= 0x7bffu) | (((ww[i] >> 16) & 0x7fffu) - 1u >= 0x7bffu);
        }
        nm = make_uint4(m[0], m[1], m[2], m[3]);
    }
    const bool literal = __ballot(odd_q) != 0ull;  // every wave holds the whole q: same answer in all of them
    uint16_t* score_row = p.scores_out ? p.scores_out + ((size_t)seq * a_num_qo_heads + hq) * p.score_stride : nullptr;
    uint32_t mm = kMmNeutral;
    const uint32_t n_mp = (n + S_T - 1) / S_T;  // metadata pages that hold entries to score
    uint32_t uni[T];  // uniform, loop invariant
#pragma unroll
    for (int t = 0; t < T; ++t) uni[t] = walk_uniform(p.st, hk, R, t);
    bool ids_parked = false;
    auto park_ids = [&]() {  // page table -> LDS; called after the first round's loads have left
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            const uint32_t e = tid + i * NT;
            if ((uint32_t)i < id_rounds && e < table_len) ids_s[e] = iraw[i];
        }
    };
    // Metadata pages are handed out ONE per round from an LDS counter (the first round is the static one: wave w takes
    // page w): a wave that is served faster scores more pages, so the waves reach the selection's barrier together -- a
    // page's scores do not depend on who computes them.  Measured against the static walk (wave w takes pages w, w + NW, ...
    // two per round, 16 loads in flight per lane): 86.2 vs 87.5 us per launch at 8 x cfg 3; two pages per round from the
    // counter: 86.8 (profiles/r05_ab_layer_kernel_variants.txt).
    __shared__ uint32_t s_next;
    if (tid == 0) s_next = NW;
    __syncthreads();
    // Tuning builds (GQA batches in one launch): -DQUEST_LAYER_GQA_STAGGER=<pages> starts the metadata
    // walk of the j-th query head of a kv-head group j * <pages> metadata pages further (0: a j-th of the row further), so
    // that the heads of a group -- co-located on one XCD by the grid order -- find each other's metadata in its L2 instead of
    // streaming it four times; -DQUEST_LAYER_META_CACHED keeps those loads in the L2's normal replacement order.  Measured at
    // cfg 5: profiles/r06_ab_gqa_one_launch_staggered_walk.txt.  (A page's scores do not depend on when they are computed.)
#ifdef QUEST_LAYER_GQA_STAGGER
    const uint32_t walk0 = n_mp == 0 ? 0u : ((hq % group) * (QUEST_LAYER_GQA_STAGGER > 0 ? (uint32_t)QUEST_LAYER_GQA_STAGGER : n_mp / group)) % n_mp;
#else
    constexpr uint32_t walk0 = 0;
#endif
    auto issue = [&](uint32_t mp, half8 (&mx)[T], half8 (&mn)[T]) {  // request the 16 entries of metadata page mp
        const int32_t pg = ld_uniform_i32(meta_table + mp);
        const half_t* b0 = a_meta + (size_t)pg * p.st.page;
#pragma unroll
        for (int t = 0; t < T; ++t) {
#ifdef QUEST_LAYER_META_CACHED
            mx[t] = ld8(b0 + uni[t] + lane_off);
            mn[t] = ld8(b0 + uni[t] + lane_off + v_off);
#else
            mx[t] = ld8_stream(b0 + uni[t] + lane_off);
            mn[t] = ld8_stream(b0 + uni[t] + lane_off + v_off);
#endif
        }
    };
    auto fetch = [&]() -> uint32_t {  // a page from the counter (broadcast from lane 0)
        uint32_t nx = 0;
        if (lane == 0) nx = atomicAdd(&s_next, 1u);
        return __builtin_amdgcn_readfirstlane(nx);
    };
    auto score = [&](uint32_t mp, const half8 (&mx)[T], const half8 (&mn)[T]) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const uint32_t e = mp * S_T + (uint32_t)(t * R + row);
            const uint16_t sb = page_score_bits<LPR>(mx[t], mn[t], qf, nm, literal);
            if (col == 0 && e < n) {
                const uint32_t kk = half_key(sb);
                keys_s[e] = (uint16_t)kk;
                mm = pk_max_u16(mm, mm_pack(kk));
                if (score_row) score_row[e] = sb;
            }
        }
    };
    // (Requesting the next page before the current one is scored -- two register sets, 16 loads in flight per lane -- is
    // SLOWER: 89.9-92.3 vs 86.4 us per launch.  This access shape wants shallow queues: scripts/probe/addr_class_probe.hip.)
#ifdef QUEST_LAYER_TWO_PAGES
    // tuning build: TWO metadata pages per round from the counter (16 loads in flight per lane).  Round 5 measured 86.8 vs 86.2 us
    // on the NHD pool ("this access shape wants shallow queues"); re-measured on the row-rotated pool in round 6:
    // profiles/r06_ab_layer_kernel_two_pages_per_round.txt
    for (uint32_t mp = wave; mp < n_mp;) {
        half8 mx[T], mn[T], mx2[T], mn2[T];
        issue(mp, mx, mn);
        const uint32_t mp2 = fetch();
        const bool has2 = mp2 < n_mp;
        if (has2) issue(mp2, mx2, mn2);
        const uint32_t nx = has2 ? fetch() : mp2;
        if (!ids_parked) {
            park_ids();
            ids_parked = true;
        }
        score(mp, mx, mn);
        if (has2) score(mp2, mx2, mn2);
        mp = nx;
    }
#else
    for (uint32_t mp = wave; mp < n_mp;) {
        half8 mx[T], mn[T];
        const uint32_t mpw = mp + walk0 < n_mp ? mp + walk0 : mp + walk0 - n_mp;  // (walk0 = 0 outside the tuning builds)
        issue(mpw, mx, mn);
        const uint32_t nx = fetch();  // the next round's page, asked for while this round's loads fly
        if (!ids_parked) {
            park_ids();
            ids_parked = true;
        }
        score(mpw, mx, mn);
        mp = nx;
    }
#endif
    if (!ids_parked) park_ids();  // waves without a round (short sequences)
    QUEST_STAMP(1);
    QUEST_WS_NOW(ws_est)  // this wave's share of the head's pages is scored

    // ---- top-k over the head's keys, straight from LDS
    if (n > 0 && n_sel > 0) {  // block-uniform
        topk_publish_range<NT>(sm, mm);
        __syncthreads();  // keys, page ids, cleared histograms, wave ranges
        const uint32_t c0 = tid * cpt;
        uint32_t key[FC];
        topk_load_keys<FC>(keys_s, c0, n, cpt, key);
        TopkCursor cur = topk_select<NT, FC>(sm, key, n, n_sel, cpt, nullptr);
        const size_t out_row = ((size_t)seq * a_num_qo_heads + hq) * p.sel_stride;
#pragma unroll
        for (int i = 0; i < FC; ++i) {
            uint32_t slot;
            if (topk_take(cur, key[i], (uint32_t)i < cpt && c0 + i < n, slot)) {
                const int32_t pg = ids_s[c0 + i];
                s_sel[slot] = pg;
                if (p.sel_idx_out) {
                    p.sel_idx_out[out_row + slot] = pg;
                    if (p.sel_val_out) p.sel_val_out[out_row + slot] = key_to_half_bits(key[i]);
                }
            }
        }
    }
    __syncthreads();
    QUEST_STAMP(5);

    // ---- gather of the selected pages + the current page (which carries the append)
    AttendArgs aa;
    aa.kv = p.kv;
    aa.st.page = p.st.page, aa.st.v_off = p.st.v_off, aa.st.head = p.st.head, aa.st.entry = p.st.entry;
    aa.st.rot = p.st.rot, aa.st.vflip = p.st.vflip;
    aa.group = group, aa.last_page_len = (uint32_t)live.z, aa.last_page_idx = live.w;
    aa.page_size = S_T, aa.n_chunks = 1u, aa.ws_stride = p.ws_stride;
    aa.scale_log2 = p.scale_log2;
    aa.lse = p.lse;
    SeqView sv;
    const size_t seq_row = (size_t)seq * a_num_qo_heads;
    sv.q = a_q + seq_row * D;
    sv.o = p.o + seq_row * D;
    sv.lse = p.lse + seq_row;
    sv.indices = kv_table;
    sv.scores = nullptr;
    sv.ws = p.ws ? p.ws + seq_row * p.ws_stride : nullptr;
    sv.state = stp;
    attend_slots<D, S_T, NW, true>(aa, sv, q_raw, 0u, hq, 0u, n_sel + 1u, n_sel, wave, lane,
                                   [&](uint32_t slot) -> int32_t { return s_sel[slot]; } QUEST_TL_ARG QUEST_WS_ARG, app);
    if (tid == 0) QUEST_WS_RECORD_EXTRA(sv.ws ? sv.ws + (size_t)hq * p.ws_stride : nullptr, D, p.ws_stride, 5, ws_est);
}

}  // namespace quest
