// Device code of the page-criticality estimate (estimate.hip).
// See estimate.hip for the reference citations and the arithmetic.
#pragma once
#include "append_device.cuh"

namespace quest {

#ifndef QUEST_EST_ITER
#define QUEST_EST_ITER 4
#endif
#ifndef QUEST_EST_WAVES
#define QUEST_EST_WAVES 2  // preferred workgroup width of the estimate launch (estimate.hip launch_estimate: 2, else 4)
#endif
constexpr int kEstIter = QUEST_EST_ITER;    // load instructions per tensor per wave, all in flight together
constexpr int kEstWaves = QUEST_EST_WAVES;  // preferred waves per workgroup (a template parameter of the kernel: 2 or 4)
#ifndef QUEST_EST_ITER_GQA
#define QUEST_EST_ITER_GQA 4
#endif
#ifndef QUEST_EST_MIN_WAVES
#define QUEST_EST_MIN_WAVES 1
#endif
template <int G>
constexpr int est_iter() { return G >= 2 ? QUEST_EST_ITER_GQA : kEstIter; }

struct AppendTail {  // optional decode-append riding in the same launch (blocks >= est_blocks)
    quest_paged_kv_t kv;
    const uint16_t* key;
    const uint16_t* value;
    uint32_t est_blocks;
    uint32_t enabled;
    const quest_step_state_t* state;  // optional device-resident lengths / last-page ids (graph replay)
    uint32_t o_stride;                // row stride of o (== n_out unless state-driven)
    uint32_t tile_heads;              // kv heads per workgroup tile (power of two dividing num_heads, <= 8)
    uint32_t tile_log2;               // log2(tile_heads): tile rows and heads are powers of two -> shifts, not divisions
    uint32_t meta_table_stride;       // batched launches (blockIdx.y = sequence): entries between page tables
    uint32_t tile_off;                // > 0: column offset (in every row of o) of the row's TILE MAXIMA: per run of 8 columns
                                      // the largest order-preserving key of the run's scores (tiles front end, decode_device.cuh)
};

// Workgroup tile = EW entries x HW kv heads = 64 rows (D = 128): HW = tile_heads (8 for 8 or 32 kv heads), so
// a tile reads HW*256 B contiguous per entry (NHD) or EW*256 B contiguous per head (HND) for max and for
// min, and produces EW consecutive scores for each of its HW*G query heads.  Row order inside the tile
// follows memory: NHD head-fastest, HND entry-fastest.
//
//   1. the tile's query vectors are requested first and parked in LDS together with a 16-bit sign mask per
//      feature (0xffff where q < 0).  For a finite non-zero q, x -> q*x is strictly monotone on the extended
//      reals and exact in fp32 (fp16 x fp16 has 22 significant bits, |q*x| <= 65504^2), so
//      max(q*Kmax, q*Kmin) == q * (q > 0 ? max(Kmax,Kmin) : min(Kmax,Kmin)) bit for bit, NaN entries included
//      (maxNum/minNum drop a NaN operand exactly as fmaxf drops a NaN product).  hi/lo are two packed fp16
//      instructions per feature PAIR shared by the G query heads of the row, the select is one v_bfi_b32 per
//      pair and the product-accumulate one v_fma_mix_f32 (both operands read as fp16; the product is exact, so
//      the FMA rounds like `acc += q*x`).  A tile whose query vectors contain a zero or a non-finite element
//      (0*inf = NaN breaks the monotonicity argument) takes the literal form instead: two products, fmaxf, add
//      -- the workgroup-uniform flag is found while staging q;
//   2. page-table entries, then all 2*kEstIter metadata loads of the wave (streaming, nt) are issued;
//   3. scores are transposed through LDS and leave as EW-long contiguous runs per query head (the 2-byte
//      scattered stores of the first version cost 0.7 us at MHA and 1.5 us at GQA-4 in write amplification).
// EWV = waves per tile, TPB = tiles per workgroup (tile t of the workgroup is run by waves [t*EWV, (t+1)*EWV)); the
// tile is (entry tile `et`, head tile `ht`).  With TPB > 1 the tiles of a workgroup share its barriers, so a tile
// that lies past the live length runs through with clamped loads and masked stores instead of returning.
// `smem` = the tile's dynamic LDS (est_tile_lds_bytes), `s_literal` = EWV words of LDS.
template <int D, int G, bool HND, int EWV, int TPB>
__device__ __forceinline__ void estimate_tile(const half_t* __restrict__ q, half_t* __restrict__ o, const quest_paged_kv_t& meta,
                                              uint32_t n_out, const AppendTail& tail, uint32_t et, uint32_t ht, uint32_t tid,
                                              unsigned char* est_smem, uint32_t* s_literal, uint32_t o_row0 = 0) {
    // o_row0: first row of `o` that belongs to this launch's sequence (batched launches).  It is applied at the final
    // store: tail.o_stride arrives by s_load (not preloaded), and nothing before the metadata loads may wait for it.
    constexpr int LPR = D / kVec;   // lanes per row
    constexpr int R = kWave / LPR;  // rows per load instruction
    constexpr int kEstIter = est_iter<G>();  // (shadows the namespace constant: GQA instantiations may differ)
    constexpr int ROWS = EWV * kEstIter * R;
    // n_out as passed bounds every address (state-driven launches pass the largest n_out the graph will
    // see; page tables and pools cover it); the live n_out comes from the state further down.
    const uint32_t n_cap = n_out;
    static_assert((ROWS & (ROWS - 1)) == 0, "tile rows must be a power of two");
    const uint32_t hw_l2 = tail.tile_log2, ew_l2 = (uint32_t)__builtin_ctz(ROWS) - hw_l2;
    const uint32_t HW = 1u << hw_l2, EW = 1u << ew_l2;
    half_t* q_s = reinterpret_cast<half_t*>(est_smem);                              // [HW*G][D]  q
    uint16_t* neg_s = reinterpret_cast<uint16_t*>(q_s + (size_t)HW * G * D);        // [HW*G][D]  q < 0 ? 0xffff : 0
    half_t* out_s = reinterpret_cast<half_t*>(neg_s + (size_t)HW * G * D);          // [HW*G][EW] scores
    // s_literal[w]: some q element staged by wave w is zero or non-finite

    const int wave = tid >> 6, lane = tid & 63;
    const int row = lane / LPR, col = lane % LPR;
    const uint32_t S = meta.page_size;
    const uint32_t e0 = et * EW, h0 = ht * HW;
    const PoolStrides ms = pool_strides(meta);
    const half_t* data = reinterpret_cast<const half_t*>(meta.data);
    const int32_t* idx = meta.indices;  // batch_size == 1: indptr[0] == 0 (estimate.cu:14)

    // (1) query vectors of this tile: HW*G contiguous heads starting at h0*G
    constexpr int QV_PER_THREAD = 2;  // covers HW*G*D/8 <= 512 16-byte vectors
    half8 qreg[QV_PER_THREAD];
    const uint32_t q_vecs = HW * G * LPR;
#pragma unroll
    for (int t = 0; t < QV_PER_THREAD; ++t) {
        const uint32_t vi = tid + t * (EWV * kWave);
        qreg[t] = ld8(q + (size_t)h0 * G * D + (size_t)(vi < q_vecs ? vi : 0) * kVec);
    }

    // (2) rows of this wave: page-table entries first, then the metadata itself
    uint32_t el[kEstIter], hl[kEstIter];
    size_t page[kEstIter];
    uint32_t ecl[kEstIter];
#pragma unroll
    for (int j = 0; j < kEstIter; ++j) {
        const uint32_t r = (wave * kEstIter + j) * R + row;  // row inside the tile
        el[j] = HND ? r & (EW - 1) : r >> hw_l2;
        hl[j] = HND ? r >> ew_l2 : r & (HW - 1);
        const uint32_t e = e0 + el[j];
        // Loads are unconditional from a clamped entry: a predicated load compiles to branch + load + wait
        // and serialises the round trips.  Clamped rows re-read the last entry (tail tiles only).
        ecl[j] = e < n_cap ? e : n_cap - 1;
    }
    // live length of a state-driven launch: requested together with the page-table entry (one scalar round trip for
    // both; from the table's first word when there is no state -- a load under a branch would be waited for on its own)
    const int32_t live_pages = ld_uniform_i32(tail.state ? &tail.state->n_pages : idx);
    if (QUEST_LIKELY((S & (EW - 1)) == 0)) {
        // the tile's EW consecutive entries (and the clamped ones of a tail tile) lie in ONE metadata page: a
        // single wave-uniform (scalar) table load instead of a vector load per row at the head of the
        // table -> metadata dependency chain
        int32_t pg_raw = ld_uniform_i32(idx + e0 / S), live_dep = live_pages;
        asm volatile("" : "+s"(pg_raw), "+s"(live_dep));  // both requested before either is waited for (no sinking)
        const size_t pg = (size_t)pg_raw;
#pragma unroll
        for (int j = 0; j < kEstIter; ++j) page[j] = pg;
    } else {
#pragma unroll
        for (int j = 0; j < kEstIter; ++j) page[j] = (size_t)idx[ecl[j] / S];
    }
    if (tail.state) {  // live length (<= n_cap); whole tiles past it have nothing to do
        n_out = (uint32_t)(live_pages - 1);  // (same register as live_dep above: already there)
        if (TPB == 1 && e0 >= n_out) return;
    }
    half8 mx[kEstIter], mn[kEstIter];
#pragma unroll
    for (int j = 0; j < kEstIter; ++j) {
        // rows follow MEMORY order: row (head slot h0 + hl, entry) holds the max vector of kv head slot ^ (entry & rot) --
        // a head of the same tile (rot < HW, checked by the host) -- and its min vector sits pool_v_off further (plain
        // layouts: rot = 0, v_off)
        const uint32_t e_in = ecl[j] % S, slot = h0 + hl[j];
        const half_t* p = data + page[j] * ms.page + (size_t)slot * ms.head + (size_t)e_in * ms.entry + col * kVec;
        mx[j] = ld8_stream(p);
        mn[j] = ld8_stream(p + pool_v_off(ms, slot));
        hl[j] ^= e_in & ms.rot;  // from here on: the row's kv HEAD inside the tile
    }

    // q -> LDS (these loads are older than the metadata loads, so this does not wait for them)
    bool odd_q = false;  // a zero or non-finite query element among the ones this thread stages
#pragma unroll
    for (int t = 0; t < QV_PER_THREAD; ++t) {
        const uint32_t vi = tid + t * (EWV * kWave);
        if (vi < q_vecs) {
            const uint4 w = __builtin_bit_cast(uint4, qreg[t]);
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
            uint32_t m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m[i] = ((ww[i] >> 15) & 0x00010001u) * 0xffffu;
                // (|x| bits) - 1 >= 0x7bff  <=>  x is +-0, +-inf or NaN
                odd_q |= ((ww[i] & 0x7fffu) - 1u >= 0x7bffu) | (((ww[i] >> 16) & 0x7fffu) - 1u >= 0x7bffu);
            }
            st8(q_s + (size_t)vi * kVec, qreg[t]);
            *reinterpret_cast<uint4*>(neg_s + (size_t)vi * kVec) = make_uint4(m[0], m[1], m[2], m[3]);
        }
    }
    {
        const unsigned long long any = __ballot(odd_q);
        if (lane == 0) s_literal[wave] = any != 0ull;
    }
    __syncthreads();
    uint32_t literal = 0;
#pragma unroll
    for (int w = 0; w < EWV; ++w) literal |= s_literal[w];
    literal = __builtin_amdgcn_readfirstlane(literal);

    // (3) scores
    if (QUEST_LIKELY(!literal)) {
#pragma unroll
        for (int j = 0; j < kEstIter; ++j) {
            const half8 hi = __builtin_elementwise_max(mx[j], mn[j]), lo = __builtin_elementwise_min(mx[j], mn[j]);
            const uint4 hb = __builtin_bit_cast(uint4, hi), lb = __builtin_bit_cast(uint4, lo);
            float accs[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t qh = hl[j] * G + g;  // query head inside the tile
                const float8 qf = to_f32(ld8(q_s + (size_t)qh * D + col * kVec));
                const uint4 nm = *reinterpret_cast<const uint4*>(neg_s + (size_t)qh * D + col * kVec);
                uint4 sb;  // per 16-bit feature: q < 0 ? lo : hi
                sb.x = (lb.x & nm.x) | (hb.x & ~nm.x);
                sb.y = (lb.y & nm.y) | (hb.y & ~nm.y);
                sb.z = (lb.z & nm.z) | (hb.z & ~nm.z);
                sb.w = (lb.w & nm.w) | (hb.w & ~nm.w);
                const float8 x = to_f32(__builtin_bit_cast(half8, sb));
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < kVec; ++i) acc = __builtin_fmaf(qf[i], x[i], acc);
                accs[g] = acc;
            }
            if constexpr (LPR == 16 && G > 1) {
                // the G row sums share one reduction tree walk (same association order as below, hence the same bits)
                int g_mine;
                const float total = row_segmented_sum16<G>(accs, col, g_mine);
                if ((col & (16 / G - 1)) == 0) out_s[(hl[j] * G + g_mine) * EW + el[j]] = (half_t)total;
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float total = row_allreduce_sum_fast<LPR>(accs[g]);
                    if (col == 0) out_s[(hl[j] * G + g) * EW + el[j]] = (half_t)total;
                }
            }
        }
    } else {  // literal form of decode_attn.cuh:152-156 (zero / non-finite query element in the tile)
#pragma unroll
        for (int j = 0; j < kEstIter; ++j) {
            const float8 a = to_f32(mx[j]), b = to_f32(mn[j]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t qh = hl[j] * G + g;
                const float8 qf = to_f32(ld8(q_s + (size_t)qh * D + col * kVec));
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < kVec; ++i) acc += __builtin_fmaxf(qf[i] * a[i], qf[i] * b[i]);
                acc = row_allreduce_sum_fast<LPR>(acc);
                if (col == 0) out_s[qh * EW + el[j]] = (half_t)acc;
            }
        }
    }
    __syncthreads();
    const uint32_t n_scores = HW * G * EW;
    for (uint32_t t = tid; t < n_scores; t += EWV * kWave) {
        const uint32_t qh = t >> ew_l2, e = e0 + (t & (EW - 1));
        const half_t sc = out_s[t];
#ifdef QUEST_EST_NOSTORE
        if (e < n_out && sc == (half_t)12345.f)
#else
        if (e < n_out)
#endif
        {
            o[((size_t)o_row0 + (size_t)h0 * G + qh) * tail.o_stride + e] = sc;
        }
        if (tail.tile_off) {  // launch-uniform: one key per (query head, run of 8 columns): the largest key of the run's valid
                              // scores.  8 consecutive t = 8 consecutive lanes = one run (EW is a multiple of 8): three DPP steps
            int key = e < n_out ? (int)half_key(half_bits(sc)) : 0;
            key = max(key, dpp_i<kDppQuadXor1>(key));
            key = max(key, dpp_i<kDppQuadXor2>(key));
            key = max(key, dpp_i<kDppHalfMirror>(key));
            if ((t & 7u) == 0 && e < n_out)
                reinterpret_cast<uint16_t*>(o)[((size_t)o_row0 + (size_t)h0 * G + qh) * tail.o_stride + tail.tile_off + (e >> 3)] =
                    (uint16_t)key;
        }
    }
}

// kv heads per tile: the largest power of two <= 8 that divides num_heads and whose query vectors
// (hw * G * D/8 sixteen-byte vectors) fit the two-per-thread staging pass
static inline uint32_t pick_tile_heads(uint32_t num_heads, uint32_t G, uint32_t lpr, uint32_t tile_waves = kEstWaves) {
    for (uint32_t hw = 8; hw > 1; hw >>= 1)
        if (num_heads % hw == 0 && hw * G * lpr <= 2u * tile_waves * kWave) return hw;
    return 1;
}

// dynamic LDS of one tile: q + sign masks + score transpose
__host__ __device__ inline size_t est_tile_lds_bytes(uint32_t hw, uint32_t G, uint32_t D, uint32_t ew) {
    return (size_t)hw * G * (2 * D + ew) * sizeof(half_t);
}

}  // namespace quest
