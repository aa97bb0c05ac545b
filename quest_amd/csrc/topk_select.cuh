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
    uint32_t wave_tot[2][NT / kWave];  // one row per block scan, so a scan needs a single barrier
    uint32_t misc[4];                  // thr_bin, above, T, need_eq
};

// Inclusive block scan of one uint32 per thread; `wave_tot` must not be reused by a later scan of the
// same kernel (each call site gets its own row), which is what lets it run with ONE barrier.
template <int NT>
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t x, uint32_t* wave_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    x = wave_scan_incl_dpp(x);
    if (lane == kWave - 1) wave_tot[wave] = x;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < NT / kWave; ++w) base += (w < wave) ? wave_tot[w] : 0u;
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
        const uint32_t incl = block_scan_incl<NT>(tot, sm.wave_tot[0]);
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
    const uint32_t before = block_scan_incl<NT>(packed, sm.wave_tot[1]) - packed;
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

}  // namespace quest
