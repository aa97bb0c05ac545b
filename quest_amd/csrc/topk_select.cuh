// Block-wide deterministic top-k SELECTION over one row of 16-bit order-preserving keys held in
// registers.  Shared by the stand-alone top-k kernel (topk.hip) and by the sparse attention kernel's
// fused front end (sparse_attn.hip), so both produce bit-identical selections.
//
// Contract (the build's declared tie rule, SURVEY.md 8a T-tie, oracle qo_topk_row): select every key
// above the threshold key T, plus the `need` LOWEST columns among keys equal to T; a selected column's
// output slot is its rank among selected columns in ascending column order.
//
// Thread t of NT owns the contiguous columns [t*cpt, t*cpt+cpt) with cpt = ceil(n / NT) <= C at run time
// (C is only the compile-time capacity of the register arrays, so a kernel built for long rows does not
// do long-row work on short ones).  Method: 11-bit histogram (2048 bins, LDS
// atomics) + block suffix scan -> threshold bin; 5-bit histogram of that bin's members -> exact T;
// packed block scan of per-thread (>T, ==T) counts -> output slots.
#pragma once
#include "quest_common.cuh"

namespace quest {

constexpr int kLowBits = 5;
constexpr int kBins1 = 1 << (16 - kLowBits);  // 2048
constexpr int kBins2 = 1 << kLowBits;         // 32

template <int NT>
struct TopkSmem {
    uint32_t hist1[kBins1];
    uint32_t hist2[kBins2];
    uint32_t wave_tot[NT / kWave];
    uint32_t misc[4];  // thr_bin, above, T, need_eq
};

// Inclusive block scan of one uint32 per thread.
template <int NT>
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t x, uint32_t* wave_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    x = wave_scan_incl_dpp(x);
    if (lane == kWave - 1) wave_tot[wave] = x;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < NT / kWave; ++w) base += (w < wave) ? wave_tot[w] : 0u;
    __syncthreads();  // wave_tot is reused by the next scan
    return x + base;
}

// Result for the calling thread: T/need (block-uniform), and the output slot / tie rank its first
// owned column would get.  Walk the owned columns in order with topk_take() to emit.
struct TopkCursor {
    uint32_t T, need, pos, eq_rank;
};

// Clear the histograms.  Call BEFORE the first use of the keys (i.e. while the global loads that
// produce them are still in flight) and follow with one __syncthreads() -- the barrier's wait is then
// the load latency the kernel pays anyway.
template <int NT>
__device__ __forceinline__ void topk_clear(TopkSmem<NT>& sm) {
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < kBins1; i += NT) sm.hist1[i] = 0;
    if (tid < kBins2) sm.hist2[tid] = 0;
}

// Columns per thread for a row of n columns.
template <int NT>
__device__ __forceinline__ uint32_t topk_cols_per_thread(uint32_t n) {
    return (n + NT - 1) / NT;
}

// Precondition: topk_clear() + __syncthreads() already done; key[i] holds column tid*cpt + i for i < cpt.
template <int NT, int C>
__device__ __forceinline__ TopkCursor topk_select(TopkSmem<NT>& sm, const uint32_t (&key)[C], uint32_t n, uint32_t k,
                                                  uint32_t cpt) {
    constexpr int BPT = kBins1 / NT;  // histogram bins per thread in the suffix scan
    static_assert(kBins1 % NT == 0 && NT >= kWave, "thread count must divide the bin count");
    const uint32_t tid = threadIdx.x;
    const uint32_t c0 = tid * cpt;

#pragma unroll
    for (int i = 0; i < C; ++i)
        if ((uint32_t)i < cpt && c0 + i < n) atomicAdd(&sm.hist1[key[i] >> kLowBits], 1u);
    __syncthreads();

    {  // suffix scan from the top bin: thread t owns bins kBins1-1-BPT*t .. kBins1-BPT*(t+1), descending
        uint32_t h[BPT], tot = 0;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            h[j] = sm.hist1[kBins1 - 1 - (BPT * tid + j)];
            tot += h[j];
        }
        const uint32_t incl = block_scan_incl<NT>(tot, sm.wave_tot);
        uint32_t run = incl - tot;
#pragma unroll
        for (int j = 0; j < BPT; ++j) {
            if (run < k && k <= run + h[j]) {
                sm.misc[0] = kBins1 - 1 - (BPT * tid + j);
                sm.misc[1] = run;
            }
            run += h[j];
        }
    }
    __syncthreads();
    const uint32_t thr_bin = sm.misc[0];
#pragma unroll
    for (int i = 0; i < C; ++i)
        if ((uint32_t)i < cpt && c0 + i < n && (key[i] >> kLowBits) == thr_bin)
            atomicAdd(&sm.hist2[key[i] & (kBins2 - 1)], 1u);
    __syncthreads();

    if (tid < kWave) {  // wave 0: lane l looks at low digit 31-l, suffix sums by shuffle
        const uint32_t above0 = sm.misc[1];
        const uint32_t cnt = tid < kBins2 ? sm.hist2[kBins2 - 1 - tid] : 0u;
        const uint32_t incl = wave_scan_incl_dpp(cnt);
        const uint32_t excl = incl - cnt;
        if (tid < kBins2 && above0 + excl < k && k <= above0 + incl) {
            sm.misc[2] = (thr_bin << kLowBits) | (kBins2 - 1 - tid);
            sm.misc[3] = k - (above0 + excl);
        }
    }
    __syncthreads();
    TopkCursor cur;
    cur.T = sm.misc[2];
    cur.need = sm.misc[3];

    uint32_t gt = 0, eq = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const bool in = (uint32_t)i < cpt && c0 + i < n;
        gt += in && key[i] > cur.T;
        eq += in && key[i] == cur.T;
    }
    const uint32_t packed = gt | (eq << 16);  // both totals < 65536 (n <= 16384)
    const uint32_t before = block_scan_incl<NT>(packed, sm.wave_tot) - packed;
    cur.eq_rank = before >> 16;
    cur.pos = (before & 0xffffu) + (cur.eq_rank < cur.need ? cur.eq_rank : cur.need);
    return cur;
}

// Advance the cursor over one owned column (in column order); returns true when the column is
// selected, in which case `slot` is its output position.
__device__ __forceinline__ bool topk_take(TopkCursor& cur, uint32_t key, bool in_range, uint32_t& slot) {
    bool take = key > cur.T;
    if (key == cur.T && in_range) {
        take = cur.eq_rank < cur.need;
        ++cur.eq_rank;
    }
    take = take && in_range;
    slot = cur.pos;
    cur.pos += take ? 1u : 0u;
    return take;
}

__device__ __forceinline__ uint16_t key_to_half_bits(uint32_t key) {
    return (key & 0x8000u) ? (uint16_t)(key & 0x7fffu) : (uint16_t)~key;
}

// ---------------------------------------------------------------------------------------------
// Single-wave variant (rows up to 64*C columns): the same selection, but one wavefront does it all,
// so there is no workgroup barrier anywhere -- phases are ordered by the wave's own program order.
// Lane l owns columns i*64 + l (i = 0..C-1): global loads are fully coalesced and a column's output
// slot comes from ballots (ascending column order = i-major, then lane).
struct TopkWaveSmem {
    uint32_t hist1[kBins1];
    uint32_t hist2[kBins2];
};

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in program order; this only stops the compiler from
    // reordering them across phases.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x, int lane) {
    (void)lane;
    return wave_scan_incl_dpp(x);
}

struct TopkWaveResult {
    uint32_t T, need;
};

// All 64 lanes of ONE wave must call this (other waves of the workgroup must not touch `sm`).
template <int C>
__device__ __forceinline__ TopkWaveResult topk_select_wave(TopkWaveSmem& sm, const uint32_t (&key)[C], uint32_t n,
                                                           uint32_t k, int lane) {
    constexpr int BPL = kBins1 / kWave;  // 32 histogram bins per lane
    static_assert(BPL == 32, "transposed histogram indexing assumes 32 bins per lane");
#pragma unroll
    for (int j = 0; j < BPL; ++j) sm.hist1[j * kWave + lane] = 0;
    if (lane < kBins2) sm.hist2[lane] = 0;
    wave_lds_fence();
#pragma unroll
    // hist1 is indexed by the REVERSED bin rb = kBins1-1-bin (so ascending rb = descending score) and
    // stored transposed, word (rb % 32) * 64 + rb / 32: lane l then owns rb = 32l .. 32l+31 and reads
    // them as hist1[j*64 + l] -- consecutive lanes on consecutive banks, no conflicts.
    for (int i = 0; i < C; ++i)
        if ((uint32_t)(i * kWave + lane) < n) {
            const uint32_t rb = kBins1 - 1 - (key[i] >> kLowBits);
            atomicAdd(&sm.hist1[((rb & (BPL - 1)) << 6) | (rb >> 5)], 1u);
        }
    wave_lds_fence();

    // suffix scan from the top: lane l owns reversed bins BPL*l .. BPL*l+BPL-1
    uint32_t thr_bin = 0, above = 0;
    {
        uint32_t h[BPL], tot = 0;
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            h[j] = sm.hist1[j * kWave + lane];
            tot += h[j];
        }
        uint32_t run = wave_scan_incl(tot, lane) - tot;
        uint32_t found_bin = 0, found_above = 0;
        bool found = false;
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
            if (run < k && k <= run + h[j]) {
                found = true;
                found_bin = kBins1 - 1 - (BPL * lane + j);
                found_above = run;
            }
            run += h[j];
        }
        const unsigned long long m = __ballot(found);  // exactly one lane when k <= n
        const int src = m ? __builtin_ctzll(m) : 0;
        thr_bin = __shfl(found_bin, src, kWave);
        above = __shfl(found_above, src, kWave);
    }
#pragma unroll
    for (int i = 0; i < C; ++i)
        if ((uint32_t)(i * kWave + lane) < n && (key[i] >> kLowBits) == thr_bin)
            atomicAdd(&sm.hist2[key[i] & (kBins2 - 1)], 1u);
    wave_lds_fence();
    TopkWaveResult r;
    {
        const uint32_t cnt = lane < kBins2 ? sm.hist2[kBins2 - 1 - lane] : 0u;
        const uint32_t incl = wave_scan_incl(cnt, lane);
        const uint32_t excl = incl - cnt;
        const bool found = lane < kBins2 && above + excl < k && k <= above + incl;
        const unsigned long long m = __ballot(found);
        const int src = m ? __builtin_ctzll(m) : 0;
        r.T = (thr_bin << kLowBits) | (uint32_t)(kBins2 - 1 - src);
        r.need = k - (above + __shfl(excl, src, kWave));
    }
    return r;
}

// Ranking cursor for the wave variant: call once per i (ascending), every lane of the wave.
struct TopkWaveCursor {
    uint32_t pos_base = 0, eq_base = 0;
};

__device__ __forceinline__ bool topk_wave_take(TopkWaveCursor& cur, const TopkWaveResult& r, uint32_t key, bool in_range,
                                               int lane, uint32_t& slot) {
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool is_eq = in_range && key == r.T;
    const unsigned long long eq_mask = __ballot(is_eq);
    const uint32_t eq_rank = cur.eq_base + (uint32_t)__builtin_popcountll(eq_mask & lt);
    const bool take = in_range && (key > r.T || (is_eq && eq_rank < r.need));
    const unsigned long long t_mask = __ballot(take);
    slot = cur.pos_base + (uint32_t)__builtin_popcountll(t_mask & lt);
    cur.pos_base += (uint32_t)__builtin_popcountll(t_mask);
    cur.eq_base += (uint32_t)__builtin_popcountll(eq_mask);
    return take;
}

}  // namespace quest
