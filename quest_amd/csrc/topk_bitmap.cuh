// Second-generation top-k front end of the sparse attention kernel: same contract as topk_select.cuh (the
// build's declared tie rule, SURVEY.md 8a T-tie / oracle qo_topk_row: every key above the threshold key T plus the
// `need` LOWEST columns among keys equal to T, output slots in ascending column order), built for latency:
//
//   * ownership is by GRANULES of 4 columns dealt round-robin over the threads (granule g -> thread g % NT, round
//     g / NT), so a thread's keys arrive by one 8-byte load per round straight from the score row -- coalesced,
//     address independent of the live row length (state-driven launches know only the capacity when they issue
//     their loads), no LDS staging pass, and every thread holds live columns as soon as the row has 4*NT of them;
//   * the threshold comes from one LDS histogram over the row's own key range (as before) but the suffix scan is
//     done by every WAVE for itself with conflict-free rotated reads -- no block scan, no publish barrier;
//   * selected / tied columns are written as two BITMAPS in column order (8 lanes combine their 4-bit nibbles
//     with three DPP ORs, no atomics); output slots are popcount ranks over the bitmap, so only the slots of
//     this workgroup's chunk are ever materialised.
//
// Barriers in the common case (row spans < 2048 key values): range, histogram, bitmaps, page list = 4 (the first
// generation: 7).  Wants score rows whose base and stride are 8-byte aligned and readable up to the next multiple of 4
// columns.  Rows that are only 2-byte aligned (the reference's contiguous [Hq][pages - 1] score tensor) are served through
// `lead`: the caller passes the row pointer rounded DOWN to 8 bytes and the number of columns it skipped (0-3); position p
// of that aligned stream is column p - lead, positions below `lead` belong to the previous row and are never live.
#pragma once
#include "topk_select.cuh"

namespace quest {

constexpr int kBmWords = QUEST_TOPK_MAX_ROW / 32;  // 512 bitmap words cover the longest row

template <bool IDS>
struct Fe2Raw {  // what a thread has in flight after fe2_issue
    uint2 k;     // 4 fp16 scores of the granule
};
template <>
struct Fe2Raw<true> {
    uint2 k;
    uint4 ids;   // their physical page ids (rows short enough for the ids to be staged in LDS)
};
// page ids are staged in LDS for rows of up to 4096 columns: 8 keys per thread at 512 threads, 16 at 256.  (Staging
// them for rows up to 8192 columns as well -- 32 KiB -- was measured in round 3 at cfg 4: 22.5 vs 21.6 us per launch.
// The 2.3 us the timeline shows between "selection done" and "page list in LDS" is not the column -> page round trip
// of fe2_resolve_pages but the younger waves of the SIMD catching up with wave 0 on an issue-bound phase.)
constexpr bool fe2_has_ids(int FC) { return FC <= 16; }

// Issue the loads of the thread's granules (rounds < RMAX): call first, before anything waits on memory.
template <int NT, int RMAX, bool IDS>
__device__ __forceinline__ void fe2_issue(const uint16_t* srow, const int32_t* table, uint32_t table_len, bool stage_ids,
                                          uint32_t n_cap, Fe2Raw<IDS> (&raw)[RMAX]) {
    const uint32_t rounds = (n_cap + 4 * NT - 1) / (4 * NT);
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        raw[r].k = make_uint2(0u, 0u);
        if constexpr (IDS) raw[r].ids = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)r < rounds) {  // wave-uniform
            const uint32_t c0 = 4u * (threadIdx.x + (uint32_t)r * NT);
            const uint32_t cc = c0 < n_cap ? c0 : 0u;  // clamped, unconditional (masked later)
            raw[r].k = *reinterpret_cast<const uint2*>(srow + cc);
            if constexpr (IDS) if (stage_ids) {
                if (cc + 4u <= table_len) {
                    raw[r].ids = *reinterpret_cast<const uint4*>(table + cc);
                } else {  // last granule of a table whose length is not a multiple of 4: element-wise, clamped
                    const uint32_t last = table_len - 1u;
                    raw[r].ids.x = (uint32_t)table[cc < last ? cc : last];
                    raw[r].ids.y = (uint32_t)table[cc + 1u < last ? cc + 1u : last];
                    raw[r].ids.z = (uint32_t)table[cc + 2u < last ? cc + 2u : last];
                    raw[r].ids.w = (uint32_t)table[cc + 3u < last ? cc + 3u : last];
                }
            }
        }
    }
}

// Zero the histograms (overlaps the loads; the first barrier of fe2_select publishes it).
template <int NT>
__device__ __forceinline__ void fe2_clear(TopkSmem<NT>& sm) {
    if constexpr (4 * NT > kBins1) {  // 1024 threads: the first half clears
        if (threadIdx.x < kBins1 / 4) reinterpret_cast<uint4*>(sm.hist1)[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    } else {
        static_assert(kBins1 % (4 * NT) == 0, "one or more 16-byte stores per thread");
#pragma unroll
        for (int v = 0; v < kBins1 / (4 * NT); ++v)
            reinterpret_cast<uint4*>(sm.hist1)[threadIdx.x + v * NT] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (threadIdx.x < kBins2) sm.hist2[threadIdx.x] = 0;
}

typedef uint16_t ushort2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(ushort2v, a),
                                                                   __builtin_bit_cast(ushort2v, b)));
}
// half_key (quest_common.cuh) of two fp16 bit patterns at once: key = b ^ (b < 0 ? 0xffff : 0x8000)
__device__ __forceinline__ uint32_t half_key2(uint32_t w) {
    return w ^ ((((w >> 15) & 0x00010001u) * 0x7fffu) | 0x80008000u);
}
// key number i of a packed key array (i is a compile-time constant at every call site)
template <int N>
__device__ __forceinline__ uint32_t key_at(const uint32_t (&key2)[N], int i) {
    return (i & 1) ? key2[i >> 1] >> 16 : key2[i >> 1] & 0xffffu;
}

// First-generation front end fed by the vector loads of fe2_issue (aligned score rows): converts the granules to keys,
// parks keys (8 bytes per granule) and page ids (16 bytes) in the LDS staging arrays topk_select.cuh reads its
// contiguous ownership from, and returns the thread's packed (max, 0xffff - min) key range.  2 load rounds instead of
// the 5 scalar ones at 2179 columns of capacity x 512 threads.
template <int NT, int RMAX>
__device__ __forceinline__ uint32_t fe1_stage_vector(const Fe2Raw<true> (&raw)[RMAX], uint16_t* keys_s, int32_t* ids_s,
                                                     uint32_t n_cap, uint32_t n) {
    const uint32_t rounds = (n_cap + 4 * NT - 1) / (4 * NT);
    uint32_t pmax = 0u, pmin = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
        if ((uint32_t)r < rounds) {
            const uint32_t c0 = 4u * (threadIdx.x + (uint32_t)r * NT);
            if (c0 < n_cap) {
                const uint32_t k0 = half_key2(raw[r].k.x), k1 = half_key2(raw[r].k.y);
                *reinterpret_cast<uint2*>(keys_s + c0) = make_uint2(k0, k1);
                if (ids_s) *reinterpret_cast<uint4*>(ids_s + c0) = raw[r].ids;
                if (c0 + 3u < n) {
                    pmax = pk_max_u16(pmax, pk_max_u16(k0, k1));
                    pmin = pk_min_u16(pmin, pk_min_u16(k0, k1));
                } else {
                    const uint32_t kk[4] = {k0 & 0xffffu, k0 >> 16, k1 & 0xffffu, k1 >> 16};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (c0 + i < n) {
                            pmax = pk_max_u16(pmax, kk[i]);
                            pmin = pk_min_u16(pmin, kk[i] | 0xffff0000u);
                        }
                }
            }
        }
    const uint32_t xl = pmax & 0xffffu, xh = pmax >> 16, nl = pmin & 0xffffu, nh = pmin >> 16;
    return ((xl > xh ? xl : xh) << 16) | (0xffffu - (nl < nh ? nl : nh));
}

// OR over the 8 lanes of an aligned lane octet; every lane of the octet gets the result.
__device__ __forceinline__ uint32_t octet_or(uint32_t x) {
    x |= (uint32_t)dpp_i<kDppQuadXor1>((int)x);
    x |= (uint32_t)dpp_i<kDppQuadXor2>((int)x);
    x |= (uint32_t)dpp_i<kDppHalfMirror>((int)x);
    return x;
}

// Keep the `keep` lowest set bits of x.
__device__ __forceinline__ uint32_t lowest_bits(uint32_t x, uint32_t keep) {
    while ((uint32_t)__builtin_popcount(x) > keep) x &= ~(0x80000000u >> __builtin_clz(x));
    return x;
}

// Selection proper.  n = live row length (<= n_cap), k = pages to select (<= n).  Writes the physical page of every
// output slot in [slot_begin, slot_end) to s_sel[slot - slot_begin]; the caller's next barrier publishes them.
// bm: two bitmaps of kBmWords words.  ids_s: LDS copy of the page table (filled here) or nullptr (read from `table`).
// SOLO: the two scans (threshold over the histogram, ranks over the bitmaps) are done by wave 0 alone and published
// through LDS (one more barrier) instead of by every wave redundantly.
#ifdef QUEST_FE2_SOLO
constexpr bool kFe2Solo = true;
#else
constexpr bool kFe2Solo = false;
#endif

template <int NT, int FC>
__device__ __forceinline__ void fe2_select(TopkSmem<NT>& sm, uint32_t (&bm)[2][kBmWords],
                                           const Fe2Raw<fe2_has_ids(FC)> (&raw)[FC / 4],
                                           const uint16_t* srow, const int32_t* table, int32_t* ids_s, uint32_t n_cap,
                                           uint32_t n, uint32_t k, uint32_t slot_begin, uint32_t slot_end, int32_t* s_sel,
                                           uint16_t* sel_val_row, int32_t* sel_idx_row, bool prefilter,
                                           long long* sub = nullptr, const uint32_t lead = 0) {
    // srow / table / n_cap are those of the ALIGNED stream when lead > 0 (srow and table moved down by `lead` entries,
    // n_cap grown by it); n stays the live COLUMN count: position c is live iff c - lead < n (unsigned: c < lead wraps)
    constexpr int RMAX = FC / 4, NWV = NT / kWave;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t rounds = (n_cap + 4 * NT - 1) / (4 * NT);

    // ---- keys (kept PACKED, two 16-bit keys per register), row range, page ids -> LDS
    uint32_t key2[FC / 2];
    uint32_t pmax = 0u, pmin = 0xffffffffu;  // packed running max / min of the valid keys
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        key2[2 * r] = key2[2 * r + 1] = 0u;
        if ((uint32_t)r < rounds) {
            const uint32_t c0 = 4u * (tid + (uint32_t)r * NT);
            key2[2 * r] = half_key2(raw[r].k.x);
            key2[2 * r + 1] = half_key2(raw[r].k.y);
            if (c0 >= lead && c0 + 3u - lead < n) {  // whole granule inside the row (all but one or two granules of the row)
                pmax = pk_max_u16(pmax, pk_max_u16(key2[2 * r], key2[2 * r + 1]));
                pmin = pk_min_u16(pmin, pk_min_u16(key2[2 * r], key2[2 * r + 1]));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c0 + i - lead < n) {
                        const uint32_t kk = key_at(key2, 4 * r + i);
                        pmax = pk_max_u16(pmax, kk);                   // low half only
                        pmin = pk_min_u16(pmin, kk | 0xffff0000u);
                    }
            }
            if constexpr (fe2_has_ids(FC))
                if (ids_s && c0 < n_cap) *reinterpret_cast<uint4*>(ids_s + c0) = raw[r].ids;
        }
    }
    uint32_t mm;
    {
        const uint32_t xl = pmax & 0xffffu, xh = pmax >> 16, nl = pmin & 0xffffu, nh = pmin >> 16;
        mm = ((xl > xh ? xl : xh) << 16) | (0xffffu - (nl < nh ? nl : nh));  // (max, 0xffff - min) of the thread's keys
    }
    topk_publish_range<NT>(sm, mm);
    // Pre-filter of long rows (a thread holds many keys, the histogram's LDS atomics dominate: 8191 of them cost 1.85 us
    // at cfg 4).  Every wave finds the j-th largest of its 64 per-thread MAXIMA, j = ceil(k / waves), by bisection over
    // the 16 key bits with ballots: at least j of the wave's keys reach that value, so at least k keys of the row reach
    // the minimum over the waves -- a guaranteed lower bound LB of the threshold key.  Only keys >= LB enter the
    // histogram (a few hundred instead of thousands; bins then start at LB, which usually makes a bin a single key
    // value).  Threshold, tie count and bitmaps are exact as before: same T, same pages.  No extra barrier (LB rides
    // with the range).
    const uint32_t jw = (k + (uint32_t)NWV - 1u) / (uint32_t)NWV;
    prefilter = prefilter && jw <= (uint32_t)kWave;
    if (prefilter) {  // block-uniform
        const uint32_t tmax = mm >> 16;  // the thread's largest valid key (0 when it holds none)
        uint32_t m = 0u;
#pragma unroll
        for (int bit = 15; bit >= 0; --bit) {
            const uint32_t cand = m | (1u << bit);
            const uint32_t cnt = (uint32_t)__builtin_popcountll(__ballot(tmax >= cand));
            m = cnt >= jw ? cand : m;
        }
        if (lane == 0u) sm.wave_lb[wave] = m;
    }
    QUEST_SUBSTAMP(0);
    __syncthreads();  // A: range + cleared histograms (+ ids) visible
    QUEST_SUBSTAMP(1);

    mm = kMmNeutral;
#pragma unroll
    for (int w = 0; w < NWV; ++w) mm = pk_max_u16(mm, sm.wave_mm[w]);
    uint32_t kmin = 0xffffu - (mm & 0xffffu);
    const uint32_t kmax = mm >> 16;
    if (prefilter) {  // bins start at the lower bound of the threshold; keys below it are neither counted nor selected
        uint32_t lb = 0xffffu;
#pragma unroll
        for (int w = 0; w < NWV; ++w) lb = min(lb, sm.wave_lb[w]);
        kmin = max(kmin, lb);
    }
    const uint32_t range = kmax - kmin;
    const uint32_t bits = 32u - (uint32_t)__builtin_clz(range | 1u);
    const uint32_t shift = bits > 11u ? bits - 11u : 0u;
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
        if ((uint32_t)r < rounds) {
            const uint32_t c0 = 4u * (tid + (uint32_t)r * NT);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t kk = key_at(key2, 4 * r + i);
                if (c0 + i - lead < n && kk >= kmin) atomicAdd(&sm.hist1[(kk - kmin) >> shift], 1u);
            }
        }
    QUEST_SUBSTAMP(2);
    __syncthreads();  // B: histogram complete
    QUEST_SUBSTAMP(3);

    // ---- threshold bin, by every wave for itself.  Lane l owns the 32 bins [2048 - 32(l+1), 2048 - 32l): reads are
    // rotated by the lane number so that the 32 lanes of a read group hit 32 different banks.
    uint32_t T, need;
    {
        uint32_t thr_bin = 0, above = 0;
        if (!kFe2Solo || wave == 0) {
            // Only the bins [0, nb) can hold keys (nb = the row's key range in bins; with the pre-filter the range starts
            // at the lower bound of the threshold and is a few hundred bins, not 2048): the lanes share THOSE bins, top
            // down -- lane l owns the bpl bins below top - bpl l, bpl = ceil(nb / 64) <= 32 -- so the loop runs bpl times
            // instead of 32 (round 4: the threshold phase of a cfg-4 workgroup was 0.96 us of 32 LDS reads per lane).
            // Reads are rotated by the lane number so that the lanes of a read group hit different banks.
            const uint32_t nb = (range >> shift) + 1u;                 // <= kBins1
            const uint32_t bpl = (nb + 63u) >> 6;                      // wave-uniform, 1 .. 32
            const uint32_t top = bpl << 6;                             // bins [nb, top) are empty (cleared), top <= kBins1
            const uint32_t base = top - bpl * (lane + 1u);
            const uint32_t rot = lane % bpl;                           // (one division per thread, outside the loop)
            uint32_t tot = 0;
            for (uint32_t j = 0; j < bpl; ++j) {
                uint32_t jj = j + rot;
                jj = jj >= bpl ? jj - bpl : jj;
                tot += sm.hist1[base + jj];
            }
            const uint32_t incl = wave_scan_incl_dpp(tot);
            const unsigned long long m1 = __ballot(incl >= k);
            const uint32_t L = (uint32_t)__builtin_ctzll(m1);  // m1 != 0: the row holds n >= k keys
            const uint32_t above_l = (uint32_t)__builtin_amdgcn_readlane((int)(incl - tot), (int)L);
            const uint32_t base_l = top - bpl * (L + 1u);
            const uint32_t c = lane < bpl ? sm.hist1[base_l + bpl - 1u - lane] : 0u;  // bins of lane L, descending
            const uint32_t incl2 = wave_scan_incl_dpp(c);
            const unsigned long long m2 = __ballot(lane < bpl && above_l + incl2 >= k);
            const uint32_t I = (uint32_t)__builtin_ctzll(m2);
            thr_bin = base_l + bpl - 1u - I;
            above = above_l + (uint32_t)__builtin_amdgcn_readlane((int)(incl2 - c), (int)I);
            if (kFe2Solo && lane == 0) {
                sm.misc[0] = thr_bin;
                sm.misc[1] = above;
            }
        }
        if (kFe2Solo) {
            __syncthreads();  // C: threshold bin published
            thr_bin = sm.misc[0];
            above = sm.misc[1];
        }
        if (shift == 0) {  // a bin IS a key value (the usual case: one or two binades of fp16 scores)
            T = kmin + thr_bin;
            need = k - above;
        } else {
            const uint32_t low_mask = (1u << shift) - 1u;
#pragma unroll
            for (int r = 0; r < RMAX; ++r)
                if ((uint32_t)r < rounds) {
                    const uint32_t c0 = 4u * (tid + (uint32_t)r * NT);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t kk = key_at(key2, 4 * r + i);
                        if (c0 + i - lead < n && kk >= kmin && ((kk - kmin) >> shift) == thr_bin)
                            atomicAdd(&sm.hist2[(kk - kmin) & low_mask], 1u);
                    }
                }
            __syncthreads();  // (only rows spanning >= 2048 key values)
            const uint32_t c3 = lane < (uint32_t)kBins2 ? sm.hist2[kBins2 - 1 - lane] : 0u;
            const uint32_t incl3 = wave_scan_incl_dpp(c3);
            const unsigned long long m3 = __ballot(lane < (uint32_t)kBins2 && above + incl3 >= k);
            const uint32_t J = (uint32_t)__builtin_ctzll(m3);
            T = kmin + ((thr_bin << shift) | (kBins2 - 1 - J));
            need = k - (above + (uint32_t)__builtin_amdgcn_readlane((int)(incl3 - c3), (int)J));
        }
    }
    QUEST_SUBSTAMP(4);
    // ---- bitmaps in column order: bit (c & 31) of word (c >> 5) for column c.  A granule is a nibble; the 8 lanes
    // of an octet own 8 consecutive granules of a round = one word.
    const uint32_t sh = (lane & 7u) * 4u;
#pragma unroll
    for (int r = 0; r < RMAX; ++r)
        if ((uint32_t)r < rounds) {
            const uint32_t g = tid + (uint32_t)r * NT, c0 = 4u * g;
            uint32_t ng = 0, ne = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool in = c0 + i - lead < n;
                const uint32_t kk = key_at(key2, 4 * r + i);
                ng |= (uint32_t)(in && kk > T) << i;
                ne |= (uint32_t)(in && kk == T) << i;
            }
            const uint32_t wg = octet_or(ng << sh), we = octet_or(ne << sh);
            if ((lane & 7u) == 0u) {
                bm[0][g >> 3] = wg;
                bm[1][g >> 3] = we;
            }
        }
    QUEST_SUBSTAMP(5);
    __syncthreads();  // D: bitmaps complete
    QUEST_SUBSTAMP(6);

    // ---- ranks: a wave scans the whole bitmap (a few words per lane) and extracts the slots of this workgroup's
    // chunk from the lanes it is responsible for (redundant form: lane % waves == wave; solo form: wave 0, all lanes)
    if (kFe2Solo && wave != 0) return;
    const uint32_t W = (n + lead + 31u) >> 5, wpl = (W + 63u) >> 6;  // words per lane, <= 8
    uint32_t sel[8], eqc = 0;
    uint32_t eqw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sel[j] = eqw[j] = 0u;
        if ((uint32_t)j < wpl) {  // wave-uniform
            const uint32_t wi = lane * wpl + (uint32_t)j;
            if (wi < W) {
                sel[j] = bm[0][wi];
                eqw[j] = bm[1][wi];
            }
            eqc += (uint32_t)__builtin_popcount(eqw[j]);
        }
    }
    int allowed = (int)need - (int)(wave_scan_incl_dpp(eqc) - eqc);  // ties still wanted when this lane's words begin
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if ((uint32_t)j < wpl) {
            const uint32_t c = (uint32_t)__builtin_popcount(eqw[j]);
            if (eqw[j]) sel[j] |= lowest_bits(eqw[j], (uint32_t)(allowed > 0 ? allowed : 0));
            allowed -= (int)c;
            cnt += (uint32_t)__builtin_popcount(sel[j]);
        }
    uint32_t rank = wave_scan_incl_dpp(cnt) - cnt;  // output slot of this lane's first selected column
    QUEST_SUBSTAMP(7);
    if ((kFe2Solo || (lane % NWV) == wave) && rank < slot_end && rank + cnt > slot_begin) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if ((uint32_t)j >= wpl) break;
            uint32_t x = sel[j];
            while (x) {
                const uint32_t b = (uint32_t)__builtin_ctz(x);
                x &= x - 1u;
                if (rank >= slot_begin && rank < slot_end) {
                    const uint32_t col = 32u * (lane * wpl + (uint32_t)j) + b;
                    // without an LDS copy of the page table the COLUMN is recorded here and fe2_resolve_pages turns
                    // the chunk's columns into pages with one parallel round trip (a dependent global load per
                    // selected bit inside this loop serialised them: 3.7 us at 8191 columns)
                    s_sel[rank - slot_begin] = ids_s ? ids_s[col] : (int32_t)col;
                    if (ids_s && sel_idx_row) {
                        sel_idx_row[rank] = ids_s[col];
                        if (sel_val_row) sel_val_row[rank] = srow[col];
                    }
                }
                ++rank;
            }
        }
    }
}

// Second half of the extraction when the page ids are not staged in LDS: call after the barrier that follows
// fe2_select, follow with another barrier.  s_sel[i] holds a column on entry, the physical page on exit.
__device__ __forceinline__ void fe2_resolve_pages(const uint16_t* srow, const int32_t* table, uint32_t slot_begin,
                                                  uint32_t slot_end, uint32_t k, int32_t* s_sel, uint16_t* sel_val_row,
                                                  int32_t* sel_idx_row) {
    const uint32_t slot = slot_begin + threadIdx.x;
    if (slot < slot_end && slot < k) {
        const uint32_t col = (uint32_t)s_sel[threadIdx.x];
        const int32_t pg = table[col];
        s_sel[threadIdx.x] = pg;
        if (sel_idx_row) {
            sel_idx_row[slot] = pg;
            if (sel_val_row) sel_val_row[slot] = srow[col];
        }
    }
}

}  // namespace quest
