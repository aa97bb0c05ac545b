// Third-generation top-k front end of the sparse attention kernel, for LONG score rows (4097 .. 16384 columns: 16 - 32
// keys per thread at 512 threads; BASELINE cfg 4 has 8191).  Same contract as the others (the build's declared tie rule,
// SURVEY.md 8a T-tie / oracle qo_topk_row; bit-identical page lists in the same order).
//
// Why: on such rows every phase of the second generation (topk_bitmap.cuh) that walks all of a thread's keys -- histogram
// atomics, bitmaps, rank scan -- is VALU-issue bound at ~1-2 us each (24 keys x 512 threads x 16 waves per CU), and the
// page list of a cfg-4 workgroup was known 8.7 us after kernel entry (profiles/r04_wallstamps_cfg4_second_generation.log).  Its
// histogram pre-filter already showed that only ~4 % of the keys matter: a lower bound LB of the threshold key found per
// wave by bisection over the per-thread maxima.  Here that bound is used to COMPACT: one pass over the thread's keys
// emits the candidates (key >= LB, a few hundred of 8191, with their columns) in column order into the wave's own
// LDS segment; the lanes of the wave then take 4 consecutive candidates each, and from there on the selection is the
// first generation's (topk_select.cuh) on <= 4 keys per thread: histogram, threshold, ranks, slots.
//
//   ownership     wave w holds the contiguous columns [64 w FC, 64 (w+1) FC); inside it lane l takes, in round r < FC / 4,
//                 the 4 columns at 256 r + 4 l -- a wave load instruction is 512 contiguous bytes, 8 per lane, like the
//                 second generation's granule loads (16-byte loads of the score row, per-thread contiguous or per-wave
//                 contiguous alike, took ~5 us to arrive in the timeline against 0.8 for the 8-byte ones).  Wave-major
//                 order of (round, lane, column) IS column order: the tie rule's order.
//   LB            every wave: the ceil(k / waves)-th largest of its 64 per-thread maxima (bisection with ballots); the
//                 minimum over the waves is reached by at least k keys of the row, so T >= LB and every selected column
//                 is a candidate.  Published with the wave maxima as the selection's key range [LB, max] (one barrier).
//   compaction    per lane and round: count, ONE packed wave prefix (DPP) for all rounds, branch-free emits of
//                 (key << 14 | column) -- non-candidates go to a dump slot per lane -- wave-local, no block barrier; then
//                 lane l reads entries 4 l .. 4 l + 3 of the segment.
//   fallback      a wave with more than 256 candidates (rows of many equal scores) or k > 64 x waves: block-uniform
//                 flag, checked behind the histogram barrier; the caller runs the second generation instead.
#pragma once
#include "topk_bitmap.cuh"

namespace quest {

constexpr int kFe3Seg = 256;       // candidates a wave can compact (4 per lane)
constexpr int kFe3Cpt = 4;         // candidates per thread after compaction
constexpr uint32_t kFe3ColBits = 14;  // columns < 16384 = QUEST_TOPK_MAX_ROW
static_assert(QUEST_TOPK_MAX_ROW <= (1u << kFe3ColBits), "column field of a packed candidate");

template <int NWV>
struct Fe3Smem {
    uint32_t seg[NWV][kFe3Seg + kWave];  // per wave: packed candidates (key << 14 | column) in column order + a dump slot per lane
    uint32_t abort;                      // set by a wave whose candidates do not fit its segment
};

// First column of the thread's 4-column granule of round j (ownership: see the header).
template <int FC>
__device__ __forceinline__ uint32_t fe3_col0(int j) {
    return (threadIdx.x >> 6) * (uint32_t)(kWave * FC) + 256u * (uint32_t)j + 4u * (threadIdx.x & 63u);
}
// Issue the loads of the thread's FC / 4 granules (call first, before anything waits on memory).  Rows are 8-byte aligned
// with a stride covering the next multiple of 4 columns; granules at or beyond n_cap are clamped to granule 0 and masked
// by the caller's live length.
template <int FC>
__device__ __forceinline__ void fe3_issue(const uint16_t* srow, uint32_t n_cap, uint2 (&raw)[FC / 4]) {
#pragma unroll
    for (int j = 0; j < FC / 4; ++j) {
        const uint32_t c = fe3_col0<FC>(j), cc = c < n_cap ? c : 0u;
        raw[j] = *reinterpret_cast<const uint2*>(srow + cc);
    }
}

// Selection.  On success (returns true, block-uniform) the physical pages of the output slots [slot_begin, slot_end) are
// in s_sel[slot - slot_begin] and the caller's next barrier publishes them.  Returns false when the row must take the
// second-generation front end (nothing was written to s_sel; sm's histograms are dirty).
// Preconditions: histograms cleared (fe2_clear) and f3.abort zeroed by thread 0 BEFORE this call's first barrier.
template <int NT, int FC>
__device__ __forceinline__ bool fe3_select(TopkSmem<NT>& sm, Fe3Smem<NT / kWave>& f3, const uint2 (&raw)[FC / 4],
                                           const uint16_t* srow, const int32_t* table, uint32_t n, uint32_t k,
                                           uint32_t slot_begin, uint32_t slot_end, int32_t* s_sel, uint16_t* sel_val_row,
                                           int32_t* sel_idx_row, long long* sub = nullptr) {
    constexpr int NWV = NT / kWave, NW2 = FC / 2;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NR = FC / 4;  // rounds = 4-column granules per thread
    static_assert(NR <= 8, "packed per-round prefixes: two words of four 8-bit fields");

    // ---- keys, packed two per register; columns at or beyond the live length become key 0 (below every real key)
    uint32_t key2[NW2];
    uint32_t pmax = 0u;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const uint32_t w[2] = {raw[j].x, raw[j].y};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const uint32_t c = fe3_col0<FC>(j) + 2u * q;
            uint32_t kk = half_key2(w[q]);
            if (c + 1u >= n) kk = c >= n ? 0u : (kk & 0xffffu);  // (rare: the granule that straddles the row's end)
            key2[2 * j + q] = kk;
            pmax = pk_max_u16(pmax, kk);
        }
    }
    const uint32_t tmax = max(pmax & 0xffffu, pmax >> 16);  // the thread's largest key (0: it holds no live column)
    // ---- per-wave lower bound of the threshold: the jw-th largest of the 64 per-thread maxima
    const uint32_t jw = (k + (uint32_t)NWV - 1u) / (uint32_t)NWV;
    uint32_t lb = 0u;
#pragma unroll
    for (int bit = 15; bit >= 0; --bit) {
        const uint32_t cand = lb | (1u << bit);
        const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(tmax >= cand));
        lb = cnt >= jw ? cand : lb;
    }
    {   // wave maximum (DPP / row swaps), published with the bound as the wave's key range [lb, max]
        int v = (int)tmax;
        v = max(v, dpp_i<kDppRowRor + 8>(v));
        v = max(v, dpp_i<kDppRowRor + 4>(v));
        v = max(v, dpp_i<kDppRowRor + 2>(v));
        v = max(v, dpp_i<kDppRowRor + 1>(v));
        v = max(v, __builtin_bit_cast(int, lane_xor<16>(__builtin_bit_cast(float, v), (int)lane)));
        v = max(v, __builtin_bit_cast(int, lane_xor<32>(__builtin_bit_cast(float, v), (int)lane)));
        if (lane == 0u) sm.wave_mm[wave] = ((uint32_t)v << 16) | (0xffffu - lb);
    }
    QUEST_SUBSTAMP(0);
    __syncthreads();  // A: bounds + cleared histograms + cleared abort flag visible
    QUEST_SUBSTAMP(1);
    uint32_t mm = kMmNeutral;
#pragma unroll
    for (int w = 0; w < NWV; ++w) mm = pk_max_u16(mm, sm.wave_mm[w]);
    const uint32_t LB = 0xffffu - (mm & 0xffffu);  // min over the waves: at least k keys of the row reach it

    // ---- compaction: the thread's candidates (key >= LB), in column order, into the wave's segment.  Column order inside
    // a wave is (round, lane, column): per-round counts, one packed prefix over the lanes (8 bits per round: a round of
    // a wave holds 512 columns, but more than 255 candidates in one means the wave does not fit its segment anyway --
    // the counts are clamped so that the fields cannot carry into each other, and the clamp makes the wave abort)
    const uint32_t LB2 = LB | (LB << 16);
    uint32_t cnt_r[NR], total_lane = 0u;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        uint32_t c = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            // both halves at once: max(key, LB) == key  <=>  key >= LB
            const uint32_t x = pk_max_u16(key2[2 * j + q], LB2) ^ key2[2 * j + q];
            c += ((x & 0xffffu) == 0u ? 1u : 0u) + ((x >> 16) == 0u ? 1u : 0u);
        }
        cnt_r[j] = c;
        total_lane += c;
    }
    // the wave's total first (it decides whether the packed prefixes are safe), then the packed per-round prefixes
    uint32_t n_w = 0u;
    bool fits = jw <= (uint32_t)kWave;
    {
        const uint32_t incl_all = wave_scan_incl_dpp(total_lane);
        n_w = (uint32_t)__builtin_amdgcn_readlane((int)incl_all, kWave - 1);
        fits = fits && n_w <= 255u;  // <= 255: every per-round prefix fits its 8-bit field (and the segment's 256 entries)
    }
    uint32_t packed[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < NR; ++j) packed[j / 4] |= cnt_r[j] << (8 * (j % 4));
    uint32_t incl_p[2] = {0u, 0u}, tot_p[2] = {0u, 0u};
    if (fits) {
#pragma unroll
        for (int h = 0; h < (NR + 3) / 4; ++h) {
            incl_p[h] = wave_scan_incl_dpp(packed[h]);
            tot_p[h] = (uint32_t)__builtin_amdgcn_readlane((int)incl_p[h], kWave - 1);
        }
    }
    if (!fits && lane == 0u) f3.abort = 1u;
    uint32_t* seg = f3.seg[wave];
    if (fits) {  // wave-uniform
        uint32_t base = 0;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            uint32_t off = base + (((incl_p[j / 4] - packed[j / 4]) >> (8 * (j % 4))) & 0xffu);
            base += (tot_p[j / 4] >> (8 * (j % 4))) & 0xffu;
            const uint32_t col0 = fe3_col0<FC>(j);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t kk = key_at(key2, 4 * j + i);
                const bool cand = kk >= LB;
                // branch-free: a non-candidate goes to the lane's dump slot behind the segment
                seg[cand ? off : (uint32_t)kFe3Seg + lane] = (kk << kFe3ColBits) | (col0 + (uint32_t)i);
                off += cand ? 1u : 0u;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();  // the reads below follow this wave's writes (same wave: LDS operations stay in order)
    uint32_t key[kFe3Cpt], colv[kFe3Cpt];
    int32_t pg[kFe3Cpt];
    {
        const uint4 e = *reinterpret_cast<const uint4*>(seg + 4u * lane);
        const uint32_t ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
        for (int i = 0; i < kFe3Cpt; ++i) {
            const bool live = 4u * lane + (uint32_t)i < n_w;
            key[i] = live ? ev[i] >> kFe3ColBits : 0u;
            colv[i] = live ? ev[i] & ((1u << kFe3ColBits) - 1u) : 0u;
            pg[i] = table[colv[i]];  // unconditional (column 0 for dead entries): in flight under the selection below
        }
    }
    auto valid = [&](int i) { return 4u * lane + (uint32_t)i < n_w; };
    QUEST_SUBSTAMP(2);
    TopkCursor cur = topk_select_v<NT, kFe3Cpt>(sm, key, valid, k, &f3.abort, nullptr);
    if (cur.T == 0xffffffffu) return false;  // block-uniform
    QUEST_SUBSTAMP(3);
#pragma unroll
    for (int i = 0; i < kFe3Cpt; ++i) {
        uint32_t slot;
        if (topk_take(cur, key[i], valid(i), slot) && slot >= slot_begin && slot < slot_end) {
            s_sel[slot - slot_begin] = pg[i];
            if (sel_idx_row) {
                sel_idx_row[slot] = pg[i];
                if (sel_val_row) sel_val_row[slot] = key_to_half_bits(key[i]);
            }
        }
    }
    QUEST_SUBSTAMP(4);
    (void)srow;
    (void)sub;
    return true;
}

}  // namespace quest
