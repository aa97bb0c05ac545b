// Third top-k front end of the sparse attention kernel: ONE wave of the workgroup does the whole selection.
//
// Why: the block-wide front ends (topk_select.cuh, topk_bitmap.cuh) are a chain of 4-7 barrier-separated phases,
// each with a latency floor of an LDS round trip plus the barrier (measured 0.25-0.7 us per phase with 16 waves
// per CU sharing the issue slots), whatever the number of keys per thread -- 2.8 us for a 2047-column row.  A row
// of <= 4096 columns fits one wave's registers (4 columns per lane per round of 256, keys packed two per VGPR),
// and with every key in one wave nothing has to cross a barrier:
//   * range, threshold and counts are wave reductions (DPP, ballots, scalar popcounts);
//   * the LDS histogram is private to the wave (LDS operations of one wave execute in order);
//   * a round's ballot masks ARE the selection bitmap in column order up to the 4-way interleave of a granule,
//     which v_mbcnt undoes per lane; only the rounds that intersect this workgroup's chunk of output slots are
//     ranked at all.
// The other waves meanwhile copy the sequence's page table into LDS and wait at the single barrier.
// Same contract as the other two (declared tie rule, ascending-column output slots): bit-identical page lists.
//
// Requirements (host-checked): score rows 8-byte aligned with a stride covering the next multiple of 4 columns,
// row capacity <= 64 * 4 * kFe3Rounds columns.
#pragma once
#include "topk_bitmap.cuh"

namespace quest {

constexpr int kFe3Rounds = 16;                       // rounds of 256 columns one wave can hold: 4096 columns
constexpr uint32_t kFe3MaxRow = 256u * kFe3Rounds;

struct Fe3Raw {
    uint2 k[kFe3Rounds];
};

// Loads of the selector wave (wave 0): round r, lane l <- columns 4(64r + l) .. +3.  Call first.
__device__ __forceinline__ void fe3_issue(const uint16_t* srow, uint32_t n_cap, Fe3Raw& raw) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rounds = (n_cap + 255u) >> 8;
#pragma unroll
    for (int r = 0; r < kFe3Rounds; ++r) {
        raw.k[r] = make_uint2(0u, 0u);
        if ((uint32_t)r < rounds) {
            const uint32_t c0 = 4u * (64u * (uint32_t)r + lane);
            raw.k[r] = *reinterpret_cast<const uint2*>(srow + (c0 < n_cap ? c0 : 0u));
        }
    }
}

// The other waves: copy the page table [n_cap + 1 entries, 16-byte aligned] into LDS (16 bytes per thread and trip).
template <int NT>
__device__ __forceinline__ void fe3_stage_ids(const int32_t* table, uint32_t table_len, int32_t* ids_s) {
    const uint32_t granules = (table_len + 3u) >> 2;
    for (uint32_t g = threadIdx.x - kWave; g < granules; g += NT - kWave) {
        const uint32_t c = 4u * g;
        uint4 v;
        if (c + 4u <= table_len) {
            v = *reinterpret_cast<const uint4*>(table + c);
        } else {
            const uint32_t last = table_len - 1u;
            v.x = (uint32_t)table[c < last ? c : last];
            v.y = (uint32_t)table[c + 1u < last ? c + 1u : last];
            v.z = (uint32_t)table[c + 2u < last ? c + 2u : last];
            v.w = (uint32_t)table[c + 3u < last ? c + 3u : last];
        }
        *reinterpret_cast<uint4*>(ids_s + c) = v;
    }
}

// Wave-wide max of x over the 64 lanes, as a wave-uniform value.
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
    int v = (int)x;
#define QUEST_WMAX(ctrl)                                   \
    {                                                      \
        const uint32_t o = (uint32_t)dpp_i<ctrl>(v);       \
        v = (int)((uint32_t)v > o ? (uint32_t)v : o);      \
    }
    QUEST_WMAX(kDppRowRor + 8)
    QUEST_WMAX(kDppRowRor + 4)
    QUEST_WMAX(kDppRowRor + 2)
    QUEST_WMAX(kDppRowRor + 1)
#undef QUEST_WMAX
    uint32_t m = (uint32_t)__builtin_amdgcn_readlane(v, 0);
    const uint32_t m1 = (uint32_t)__builtin_amdgcn_readlane(v, 16), m2 = (uint32_t)__builtin_amdgcn_readlane(v, 32),
                   m3 = (uint32_t)__builtin_amdgcn_readlane(v, 48);
    m = m > m1 ? m : m1;
    m = m > m2 ? m : m2;
    return m > m3 ? m : m3;
}

// Selection by the calling wave (all 64 lanes active; call from wave 0 only).  n = live row length, k = pages to
// select (1 <= k <= n).  For every output slot s in [slot_begin, slot_end) writes the COLUMN of the selected page
// to s_col[s - slot_begin]; the caller maps columns to pages after its barrier (LDS table or global table).
template <int NT>
__device__ __forceinline__ void fe3_select(TopkSmem<NT>& sm, const Fe3Raw& raw, uint32_t n, uint32_t k,
                                           uint32_t slot_begin, uint32_t slot_end, int32_t* s_col,
                                           long long* sub = nullptr) {
    uint32_t* const hist1 = sm.hist1;
    uint32_t* const hist2 = sm.hist2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t rounds = (n + 255u) >> 8;           // live rounds
    const uint32_t full = n >> 8;                      // rounds whose 256 columns are all inside the row
    const uint32_t tail_valid = n & 255u;              // valid columns of round `full` (0: none)

    // zero the wave's histogram while the loads are in flight
#pragma unroll
    for (int j = 0; j < kBins1 / 256; ++j)
        reinterpret_cast<uint4*>(hist1)[lane + 64u * j] = make_uint4(0u, 0u, 0u, 0u);
    if (lane < (uint32_t)kBins2) hist2[lane] = 0u;

    // ---- keys (packed two per register) and the row's key range
    uint32_t key2[2 * kFe3Rounds];
    uint32_t pmax = 0u, pmin = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < kFe3Rounds; ++r) {
        key2[2 * r] = key2[2 * r + 1] = 0u;
        if ((uint32_t)r < rounds) {  // wave-uniform
            key2[2 * r] = half_key2(raw.k[r].x);
            key2[2 * r + 1] = half_key2(raw.k[r].y);
            if ((uint32_t)r < full) {
                pmax = pk_max_u16(pmax, pk_max_u16(key2[2 * r], key2[2 * r + 1]));
                pmin = pk_min_u16(pmin, pk_min_u16(key2[2 * r], key2[2 * r + 1]));
            } else {  // the row's last, partial round
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4u * lane + i < tail_valid) {
                        const uint32_t kk = key_at(key2, 4 * r + i);
                        pmax = pk_max_u16(pmax, kk);
                        pmin = pk_min_u16(pmin, kk | 0xffff0000u);
                    }
            }
        }
    }
    const uint32_t xl = pmax & 0xffffu, xh = pmax >> 16, nl = pmin & 0xffffu, nh = pmin >> 16;
    const uint32_t kmax = wave_max_u32(xl > xh ? xl : xh);
    const uint32_t kmin = 0xffffu - wave_max_u32(0xffffu - (nl < nh ? nl : nh));
    const uint32_t range = kmax - kmin;
    const uint32_t bits = 32u - (uint32_t)__builtin_clz(range | 1u);
    const uint32_t shift = bits > 11u ? bits - 11u : 0u;

    QUEST_SUBSTAMP(0);
    // ---- histogram of (key - kmin) >> shift
#pragma unroll
    for (int r = 0; r < kFe3Rounds; ++r)
        if ((uint32_t)r < rounds) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((uint32_t)r < full || 4u * lane + i < tail_valid)
                    atomicAdd(&hist1[(key_at(key2, 4 * r + i) - kmin) >> shift], 1u);
        }

    QUEST_SUBSTAMP(1);
    // ---- threshold key T and the number of ties to take (LDS operations of one wave are processed in order, so
    // the reads below see the adds above)
    uint32_t T, need;
    {
        const uint32_t base = kBins1 - 32u * (lane + 1u);
        uint32_t tot = 0;
#pragma unroll
        for (int j = 0; j < 32; ++j) tot += hist1[base + ((j + lane) & 31u)];  // rotated: conflict-free
        const uint32_t incl = wave_scan_incl_dpp(tot);
        const uint32_t L = (uint32_t)__builtin_ctzll(__ballot(incl >= k));
        const uint32_t above_l = (uint32_t)__builtin_amdgcn_readlane((int)(incl - tot), (int)L);
        const uint32_t base_l = kBins1 - 32u * (L + 1u);
        const uint32_t c = lane < 32u ? hist1[base_l + 31u - lane] : 0u;
        const uint32_t incl2 = wave_scan_incl_dpp(c);
        const uint32_t I = (uint32_t)__builtin_ctzll(__ballot(lane < 32u && above_l + incl2 >= k));
        const uint32_t thr_bin = base_l + 31u - I;
        const uint32_t above = above_l + (uint32_t)__builtin_amdgcn_readlane((int)(incl2 - c), (int)I);
        if (shift == 0) {
            T = kmin + thr_bin;
            need = k - above;
        } else {  // rows spanning >= 2048 key values: the low `shift` bits of the threshold bin's members
            const uint32_t low_mask = (1u << shift) - 1u;
#pragma unroll
            for (int r = 0; r < kFe3Rounds; ++r)
                if ((uint32_t)r < rounds) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t d = key_at(key2, 4 * r + i) - kmin;
                        if (((uint32_t)r < full || 4u * lane + i < tail_valid) && (d >> shift) == thr_bin)
                            atomicAdd(&hist2[d & low_mask], 1u);
                    }
                }
            const uint32_t c3 = lane < (uint32_t)kBins2 ? hist2[kBins2 - 1 - lane] : 0u;
            const uint32_t incl3 = wave_scan_incl_dpp(c3);
            const uint32_t J = (uint32_t)__builtin_ctzll(__ballot(lane < (uint32_t)kBins2 && above + incl3 >= k));
            T = kmin + ((thr_bin << shift) | (kBins2 - 1 - J));
            need = k - (above + (uint32_t)__builtin_amdgcn_readlane((int)(incl3 - c3), (int)J));
        }
    }

    QUEST_SUBSTAMP(2);
    // ---- per round: how many columns are above T / equal to T (scalar popcounts of the ballots), running totals,
    // and the ranking of the rounds that reach into this workgroup's slots
    uint32_t sel_before = 0, eq_before = 0;  // selected columns / tied columns in earlier rounds (wave-uniform)
#pragma unroll
    for (int r = 0; r < kFe3Rounds; ++r) {
        if ((uint32_t)r >= rounds) continue;  // wave-uniform
        unsigned long long mg[4], me[4];
        bool gtb[4], eqb[4];
        uint32_t cg = 0, ce = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t kk = key_at(key2, 4 * r + i);
            const bool in = (uint32_t)r < full || 4u * lane + i < tail_valid;
            gtb[i] = in && kk > T;
            eqb[i] = in && kk == T;
            mg[i] = __ballot(gtb[i]);
            me[i] = __ballot(eqb[i]);
            cg += (uint32_t)__builtin_popcountll(mg[i]);
            ce += (uint32_t)__builtin_popcountll(me[i]);
        }
        const uint32_t eq_left = need > eq_before ? need - eq_before : 0u;  // ties still to take when the round begins
        const uint32_t eq_take = eq_left < ce ? eq_left : ce;
        const uint32_t sel_round = cg + eq_take;
        if (sel_before < slot_end && sel_before + sel_round > slot_begin) {  // wave-uniform: rank this round's columns
            // column order inside the round: lane-major, then i.  below_x = set bits of the 4 masks in lower lanes.
            uint32_t mine_eq[4], eq_rank = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                mine_eq[i] = eqb[i] ? 1u : 0u;
                eq_rank += __builtin_amdgcn_mbcnt_hi((uint32_t)(me[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)me[i], 0u));
            }
            uint32_t sel_bit[4];
            unsigned long long ms[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool tie_taken = mine_eq[i] && eq_rank < eq_left;  // lowest columns first
                eq_rank += mine_eq[i];
                sel_bit[i] = (uint32_t)(gtb[i] || tie_taken);
                ms[i] = eq_take == ce ? (mg[i] | me[i]) : (eq_take == 0u ? mg[i] : __ballot(sel_bit[i] != 0u));
            }
            uint32_t rank = sel_before;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                rank += __builtin_amdgcn_mbcnt_hi((uint32_t)(ms[i] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms[i], 0u));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (sel_bit[i] && rank >= slot_begin && rank < slot_end)
                    s_col[rank - slot_begin] = (int32_t)(4u * (64u * (uint32_t)r + lane) + (uint32_t)i);
                rank += sel_bit[i];
            }
        }
        sel_before += sel_round;
        eq_before += ce;
    }
    QUEST_SUBSTAMP(3);
}

}  // namespace quest
