// Column-range ownership for the fused top-k front end of the sparse attention kernel (round 4).
//
// Same contract as topk_select.cuh / topk_bitmap.cuh (the build's declared tie rule, SURVEY.md 8a T-tie / oracle
// qo_topk_row: every key above the threshold key T plus the `need` LOWEST columns among keys equal to T) -- but a
// workgroup no longer owns a range of OUTPUT SLOTS of its head's page list, it owns a range of COLUMNS of the score
// row.  Slot ownership needs every selected column's rank among ALL selected columns (a block-wide scan of (>T, ==T)
// counts, a slot walk, a page list in LDS and a barrier: 1.4 of the 5.4 us a cfg-3 workgroup spent before its first
// K/V byte, profiles/r03_timeline_front_end_cfg3.log).  With column ownership a workgroup needs, besides T and `need`,
// only the number of ties in LOWER columns than its range -- and the histogram that finds T delivers that for free:
// every LDS atomic adds 1 | (column < range start) << 16, so a bin's word holds (keys in the bin, of which in lower
// columns) and the suffix sums hold (keys above, of which in lower columns).  No rank scan, no slot walk, no barrier
// after the one that publishes T; every WAVE then looks at the whole range (4 columns per lane, <= 256 columns) by
// itself and takes list entries wave, wave + NW, ... -- balanced over the waves without any exchange.
//
// The page SET of a head is bit-identical to the other front ends' (same T, same ties); what changes is which
// workgroup folds which page, i.e. the fp32 association order of the partial-state merge (tests: <= 2e-3 against
// the fp64 oracle, as for any other work split).  Output slots (sel_idx_out / sel_val_out, inspection only) follow
// from the same packed words: slot of the range's first selected column = (keys above T in lower columns) +
// min(ties in lower columns, need).
#pragma once
#include "topk_bitmap.cuh"

namespace quest {

constexpr int kColRangeMax = 256;  // columns of a workgroup's range: 4 per lane of a wave

struct ColRangeSel {
    uint32_t T, need;   // threshold key; ties (keys == T) to take, lowest columns first
    uint32_t gt_lower;  // keys > T in columns below the workgroup's range
    uint32_t eq_lower;  // keys == T in columns below the workgroup's range
};

// Packed increment of a histogram word for column `col` of a workgroup whose range starts at `range_start`.
__device__ __forceinline__ uint32_t colrange_inc(uint32_t col, uint32_t range_start) {
    return 1u + (col < range_start ? 0x10000u : 0u);
}

// First-generation ownership (thread t holds the contiguous columns [t*cpt, t*cpt + cpt) in key[]): threshold and the
// lower-column counts.  Precondition as topk_select: histograms cleared, topk_publish_range() + __syncthreads() done.
// Row lengths up to 16384 keep both halves of every packed word below 65536.
template <int NT, int C>
__device__ __forceinline__ ColRangeSel topk_threshold_colrange(TopkSmem<NT>& sm, const uint32_t (&key)[C], uint32_t n,
                                                               uint32_t k, uint32_t cpt, uint32_t range_start,
                                                               long long* sub = nullptr) {
    constexpr int BPT = kBins1 / NT;
    static_assert(kBins1 % NT == 0 && NT >= kWave, "thread count must divide the bin count");
    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = tid * cpt;
    uint32_t mm = kMmNeutral;
#pragma unroll
    for (int w = 0; w < NT / kWave; ++w) mm = pk_max_u16(mm, sm.wave_mm[w]);
    const uint32_t kmax = mm >> 16, kmin = 0xffffu - (mm & 0xffffu);
    const uint32_t range = kmax - kmin;
    const uint32_t bits = 32u - (uint32_t)__builtin_clz(range | 1u);
    const uint32_t shift = bits > 11u ? bits - 11u : 0u;
    const uint32_t low_mask = (1u << shift) - 1u;
#pragma unroll
    for (int i = 0; i < C; ++i)
        if ((uint32_t)i < cpt && c0 + i < n) atomicAdd(&sm.hist1[(key[i] - kmin) >> shift], colrange_inc(c0 + i, range_start));
    QUEST_SUBSTAMP(0);
    __syncthreads();
    QUEST_SUBSTAMP(1);
    {   // suffix scan of the packed words from the top bin (thread t owns bins kBins1-1-BPT*t .. descending)
        uint32_t h[BPT], tot = 0;
        {
            const uint32_t* blk = &sm.hist1[kBins1 - BPT * (tid + 1)];
            uint32_t asc[BPT];
            if constexpr (BPT % 4 == 0) {
#pragma unroll
                for (int v = 0; v < BPT / 4; ++v) {
                    const uint4 q4 = reinterpret_cast<const uint4*>(blk)[v];
                    asc[4 * v] = q4.x, asc[4 * v + 1] = q4.y, asc[4 * v + 2] = q4.z, asc[4 * v + 3] = q4.w;
                }
            } else if constexpr (BPT == 2) {
                const uint2 q2 = *reinterpret_cast<const uint2*>(blk);
                asc[0] = q2.x, asc[1] = q2.y;
            } else {
#pragma unroll
                for (int j = 0; j < BPT; ++j) asc[j] = blk[j];
            }
#pragma unroll
            for (int j = 0; j < BPT; ++j) {
                h[j] = asc[BPT - 1 - j];
                tot += h[j];
            }
        }
        QUEST_SUBSTAMP(2);
        const uint32_t incl = block_scan_incl<NT>(tot, sm.wave_tot[0]);
        QUEST_SUBSTAMP(3);
        uint32_t run = incl - tot;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            const uint32_t lo = run & 0xffffu;
            if (lo < k && k <= lo + (h[j] & 0xffffu)) {
                sm.misc[0] = kBins1 - 1 - (BPT * tid + j);
                sm.misc[1] = run;   // (keys above the bin, of which in lower columns)
                sm.misc[2] = h[j];  // (keys in the bin, of which in lower columns)
            }
            run += h[j];
        }
    }
    __syncthreads();
    QUEST_SUBSTAMP(4);
    const uint32_t thr_bin = sm.misc[0], above = sm.misc[1];
    ColRangeSel cs;
    if (shift == 0) {  // a bin is a key value
        const uint32_t hb = sm.misc[2];
        cs.T = kmin + thr_bin;
        cs.need = k - (above & 0xffffu);
        cs.gt_lower = above >> 16;
        cs.eq_lower = hb >> 16;
        QUEST_SUBSTAMP(5);
        QUEST_SUBSTAMP(6);
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i)
            if ((uint32_t)i < cpt && c0 + i < n && ((key[i] - kmin) >> shift) == thr_bin)
                atomicAdd(&sm.hist2[(key[i] - kmin) & low_mask], colrange_inc(c0 + i, range_start));
        __syncthreads();
        QUEST_SUBSTAMP(5);
        if (tid < kWave) {  // wave 0: lane l looks at low digit 31-l
            const uint32_t cnt = tid < kBins2 ? sm.hist2[kBins2 - 1 - tid] : 0u;
            const uint32_t incl = wave_scan_incl_dpp(cnt);
            const uint32_t excl = incl - cnt;
            const uint32_t a0 = (above & 0xffffu) + (excl & 0xffffu);
            if (tid < kBins2 && a0 < k && k <= a0 + (cnt & 0xffffu)) {
                sm.misc[4] = kmin + ((thr_bin << shift) | (kBins2 - 1 - tid));
                sm.misc[5] = k - a0;
                sm.misc[6] = (above >> 16) + (excl >> 16);
                sm.misc[7] = cnt >> 16;
            }
        }
        __syncthreads();
        QUEST_SUBSTAMP(6);
        cs.T = sm.misc[4];
        cs.need = sm.misc[5];
        cs.gt_lower = sm.misc[6];
        cs.eq_lower = sm.misc[7];
    }
    (void)sub;
    return cs;
}

// Second-generation ownership (granules of 4 columns dealt round-robin over the threads, topk_bitmap.cuh) with its
// histogram pre-filter: threshold and lower-column counts, found by every wave for itself after the histogram barrier.
// Call right after fe2_issue + fe2_clear; contains the range barrier (A) and the histogram barrier (B).
template <int NT, int FC>
__device__ __forceinline__ ColRangeSel fe2_threshold_colrange(TopkSmem<NT>& sm, const Fe2Raw<false> (&raw)[FC / 4],
                                                              uint32_t n_cap, uint32_t n, uint32_t k, uint32_t range_start,
                                                              bool prefilter, long long* sub = nullptr) {
    constexpr int RMAX = FC / 4, NWV = NT / kWave;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t rounds = (n_cap + 4 * NT - 1) / (4 * NT);
    uint32_t key2[FC / 2];
    uint32_t pmax = 0u, pmin = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        key2[2 * r] = key2[2 * r + 1] = 0u;
        if ((uint32_t)r < rounds) {
            const uint32_t c0 = 4u * (tid + (uint32_t)r * NT);
            key2[2 * r] = half_key2(raw[r].k.x);
            key2[2 * r + 1] = half_key2(raw[r].k.y);
            if (c0 + 3u < n) {
                pmax = pk_max_u16(pmax, pk_max_u16(key2[2 * r], key2[2 * r + 1]));
                pmin = pk_min_u16(pmin, pk_min_u16(key2[2 * r], key2[2 * r + 1]));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (c0 + i < n) {
                        const uint32_t kk = key_at(key2, 4 * r + i);
                        pmax = pk_max_u16(pmax, kk);
                        pmin = pk_min_u16(pmin, kk | 0xffff0000u);
                    }
            }
        }
    }
    uint32_t mm;
    {
        const uint32_t xl = pmax & 0xffffu, xh = pmax >> 16, nl = pmin & 0xffffu, nh = pmin >> 16;
        mm = ((xl > xh ? xl : xh) << 16) | (0xffffu - (nl < nh ? nl : nh));
    }
    topk_publish_range<NT>(sm, mm);
    const uint32_t jw = (k + (uint32_t)NWV - 1u) / (uint32_t)NWV;
    prefilter = prefilter && jw <= (uint32_t)kWave;
    if (prefilter) {  // per-wave lower bound of the threshold key (fe2_select has the argument)
        const uint32_t tmax = mm >> 16;
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
    __syncthreads();  // A
    QUEST_SUBSTAMP(1);
    mm = kMmNeutral;
#pragma unroll
    for (int w = 0; w < NWV; ++w) mm = pk_max_u16(mm, sm.wave_mm[w]);
    uint32_t kmin = 0xffffu - (mm & 0xffffu);
    const uint32_t kmax = mm >> 16;
    if (prefilter) {
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
            const uint32_t lower = c0 < range_start ? 0x10000u : 0u;  // ranges start at multiples of 4: whole granule
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t kk = key_at(key2, 4 * r + i);
                if (c0 + i < n && kk >= kmin) atomicAdd(&sm.hist1[(kk - kmin) >> shift], 1u + lower);
            }
        }
    QUEST_SUBSTAMP(2);
    __syncthreads();  // B
    QUEST_SUBSTAMP(3);
    ColRangeSel cs;
    const uint32_t base = kBins1 - 32u * (lane + 1u);
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) tot += sm.hist1[base + ((j + lane) & 31u)];
    const uint32_t incl = wave_scan_incl_dpp(tot);
    const unsigned long long m1 = __ballot((incl & 0xffffu) >= k);
    const uint32_t L = (uint32_t)__builtin_ctzll(m1);  // m1 != 0: the row holds n >= k keys, all of them >= kmin
    const uint32_t above_l = (uint32_t)__builtin_amdgcn_readlane((int)(incl - tot), (int)L);
    const uint32_t base_l = kBins1 - 32u * (L + 1u);
    const uint32_t c = lane < 32u ? sm.hist1[base_l + 31u - lane] : 0u;
    const uint32_t incl2 = wave_scan_incl_dpp(c);
    const unsigned long long m2 = __ballot(lane < 32u && (above_l & 0xffffu) + (incl2 & 0xffffu) >= k);
    const uint32_t I = (uint32_t)__builtin_ctzll(m2);
    const uint32_t thr_bin = base_l + 31u - I;
    const uint32_t above = above_l + (uint32_t)__builtin_amdgcn_readlane((int)(incl2 - c), (int)I);
    const uint32_t hb = (uint32_t)__builtin_amdgcn_readlane((int)c, (int)I);
    if (shift == 0) {
        cs.T = kmin + thr_bin;
        cs.need = k - (above & 0xffffu);
        cs.gt_lower = above >> 16;
        cs.eq_lower = hb >> 16;
    } else {
        const uint32_t low_mask = (1u << shift) - 1u;
#pragma unroll
        for (int r = 0; r < RMAX; ++r)
            if ((uint32_t)r < rounds) {
                const uint32_t c0 = 4u * (tid + (uint32_t)r * NT);
                const uint32_t lower = c0 < range_start ? 0x10000u : 0u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint32_t kk = key_at(key2, 4 * r + i);
                    if (c0 + i < n && kk >= kmin && ((kk - kmin) >> shift) == thr_bin)
                        atomicAdd(&sm.hist2[(kk - kmin) & low_mask], 1u + lower);
                }
            }
        __syncthreads();  // (only rows spanning >= 2048 key values above the lower bound)
        const uint32_t c3 = lane < (uint32_t)kBins2 ? sm.hist2[kBins2 - 1 - lane] : 0u;
        const uint32_t incl3 = wave_scan_incl_dpp(c3);
        const unsigned long long m3 = __ballot(lane < (uint32_t)kBins2 && (above & 0xffffu) + (incl3 & 0xffffu) >= k);
        const uint32_t J = (uint32_t)__builtin_ctzll(m3);
        const uint32_t above3 = above + (uint32_t)__builtin_amdgcn_readlane((int)(incl3 - c3), (int)J);
        cs.T = kmin + ((thr_bin << shift) | (kBins2 - 1 - J));
        cs.need = k - (above3 & 0xffffu);
        cs.gt_lower = above3 >> 16;
        cs.eq_lower = (uint32_t)__builtin_amdgcn_readlane((int)c3, (int)J) >> 16;
    }
    QUEST_SUBSTAMP(4);
    (void)sub;
    return cs;
}

// The loads a lane needs for its 4 columns of the workgroup's range: scores (8 bytes) and page ids (16 bytes).
struct ColRangeRaw {
    uint2 k;
    uint4 ids;
};
// gc0 = the lane's first column (a multiple of 4); columns at or beyond n_cap are clamped to granule 0 and masked by
// the caller's column bound.  Score rows are 8-byte aligned with a stride covering the next multiple of 4 columns; the
// page table has table_len entries (its tail granule is read element-wise).
__device__ __forceinline__ ColRangeRaw colrange_issue(const uint16_t* srow, const int32_t* table, uint32_t table_len,
                                                      uint32_t n_cap, uint32_t gc0, bool table_aligned) {
    ColRangeRaw r;
    const uint32_t cc = gc0 < n_cap ? gc0 : 0u;
    r.k = *reinterpret_cast<const uint2*>(srow + cc);
    if (table_aligned && cc + 4u <= table_len) {
        r.ids = *reinterpret_cast<const uint4*>(table + cc);
    } else {
        const uint32_t last = table_len - 1u;
        r.ids.x = (uint32_t)table[cc < last ? cc : last];
        r.ids.y = (uint32_t)table[cc + 1u < last ? cc + 1u : last];
        r.ids.z = (uint32_t)table[cc + 2u < last ? cc + 2u : last];
        r.ids.w = (uint32_t)table[cc + 3u < last ? cc + 3u : last];
    }
    return r;
}

// Every wave for itself: which of the range's columns are selected -- a lane holds NG granules of 4 columns, granule g
// at gc0 + 256 g, rounds below range_len, columns bounded by col_end; the selected columns' page ids go to list[0 .. count) in column order (all waves
// write the same values to the same words: a wave's reads follow its own writes in program order, so no barrier is
// needed).  Returns the count (wave-uniform).  sel_idx_row / sel_val_row: optional global rows of the head's selection
// (inspection aid), written by the wave(s) with write_out.
template <int NG>
__device__ __forceinline__ uint32_t colrange_collect(const ColRangeSel& cs, const ColRangeRaw (&raw)[NG], uint32_t gc0,
                                                     uint32_t range_len, uint32_t col_end, int32_t* list, bool write_out,
                                                     int32_t* sel_idx_row, uint16_t* sel_val_row) {
    // ties still to be taken when this granule round's columns begin (signed: lower columns may have used them up)
    int allowed0 = (int)cs.need - (int)cs.eq_lower;
    // output slot of the range's first selected column (ascending column order over the whole row)
    const uint32_t slot0 = cs.gt_lower + (cs.eq_lower < cs.need ? cs.eq_lower : cs.need);
    uint32_t base = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g > 0 && (uint32_t)(g * kColRangeMax) >= range_len) break;  // block-uniform: rounds beyond the live range
        const uint32_t k01 = half_key2(raw[g].k.x), k23 = half_key2(raw[g].k.y);
        const uint32_t kk[4] = {k01 & 0xffffu, k01 >> 16, k23 & 0xffffu, k23 >> 16};
        const uint32_t id[4] = {raw[g].ids.x, raw[g].ids.y, raw[g].ids.z, raw[g].ids.w};
        const uint32_t c = gc0 + (uint32_t)(g * kColRangeMax);
        bool gt[4], eq[4];
        uint32_t ne = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = c + i < col_end;
            gt[i] = in && kk[i] > cs.T;
            eq[i] = in && kk[i] == cs.T;
            ne += eq[i] ? 1u : 0u;
        }
        int allowed = allowed0;
        if (__ballot(ne != 0u)) {  // wave-uniform: most ranges hold no tie
            const uint32_t incl_e = wave_scan_incl_dpp(ne);
            allowed -= (int)(incl_e - ne);
            allowed0 -= __builtin_amdgcn_readlane((int)incl_e, kWave - 1);
        }
        bool take[4];
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            take[i] = gt[i] || (eq[i] && allowed > 0);
            allowed -= eq[i] ? 1 : 0;
            cnt += take[i] ? 1u : 0u;
        }
        const uint32_t incl = wave_scan_incl_dpp(cnt);
        uint32_t rank = base + incl - cnt;
        base += (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (take[i]) {
                list[rank] = (int32_t)id[i];
                if (write_out && sel_idx_row) {
                    sel_idx_row[slot0 + rank] = (int32_t)id[i];
                    if (sel_val_row) sel_val_row[slot0 + rank] = key_to_half_bits(kk[i]);
                }
                ++rank;
            }
    }
    return base;
}

}  // namespace quest
