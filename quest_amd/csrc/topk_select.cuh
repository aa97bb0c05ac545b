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
// do long-row work on short ones).  Method: histogram of (key - row min) >> shift over <= 2048 bins (LDS
// atomics) + block suffix scan -> threshold bin; histogram of that bin's members' low `shift` (<= 5) bits
// -> exact T; packed block scan of per-thread (>T, ==T) counts -> output slots.
#pragma once
#include "quest_common.cuh"
#include "stamps.cuh"

namespace quest {

constexpr int kLowBits = 5;
constexpr int kBins1 = 1 << (16 - kLowBits);  // 2048
constexpr int kBins2 = 1 << kLowBits;         // 32

template <int NT>
struct TopkSmem {
    alignas(16) uint32_t hist1[kBins1];
    uint32_t hist2[kBins2];
    uint32_t wave_tot[2][NT / kWave];  // one row per block scan, so a scan needs a single barrier
    uint32_t wave_mm[NT / kWave];      // per wave: (max key << 16) | (0xffff - min key)
    uint32_t wave_lb[NT / kWave];      // per wave: a key that at least ceil(k / waves) of the wave's keys reach (fe2 pre-filter)
    uint32_t misc[8];                  // thr_bin, above, T, need_eq (topk_select)
};

// Packed (max, 0xffff - min) of 16-bit keys: one v_pk_max_u16 combines both halves.
typedef uint16_t ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, a),
                                                                   __builtin_bit_cast(ushort2_t, b)));
}
__device__ __forceinline__ uint32_t mm_pack(uint32_t key) { return (key << 16) | (0xffffu - key); }
constexpr uint32_t kMmNeutral = 0u;  // max = 0, min = 0xffff

// Publish the wave's key range (call with the packed range of the thread's VALID keys, in any ownership --
// e.g. straight from the coalesced loads) before the barrier that precedes topk_select.
template <int NT>
__device__ __forceinline__ void topk_publish_range(TopkSmem<NT>& sm, uint32_t mm) {
    int v = (int)mm;
    v = (int)pk_max_u16((uint32_t)v, (uint32_t)dpp_i<kDppRowRor + 8>(v));
    v = (int)pk_max_u16((uint32_t)v, (uint32_t)dpp_i<kDppRowRor + 4>(v));
    v = (int)pk_max_u16((uint32_t)v, (uint32_t)dpp_i<kDppRowRor + 2>(v));
    v = (int)pk_max_u16((uint32_t)v, (uint32_t)dpp_i<kDppRowRor + 1>(v));
    const int lane = threadIdx.x & 63;
    v = (int)pk_max_u16((uint32_t)v, __builtin_bit_cast(uint32_t, lane_xor<16>(__builtin_bit_cast(float, v), lane)));
    v = (int)pk_max_u16((uint32_t)v, __builtin_bit_cast(uint32_t, lane_xor<32>(__builtin_bit_cast(float, v), lane)));
    if ((threadIdx.x & 63) == 0) sm.wave_mm[threadIdx.x >> 6] = (uint32_t)v;
}

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

// Fetch the thread's cpt contiguous keys from the LDS staging array (16-byte aligned, padded to a multiple of
// 8 keys).  cpt of 2/4/8 -- the common cases -- is one 4/8/16-byte read per thread; 2-byte reads at a stride
// of cpt keys between lanes are bank conflicts (4-way at cpt = 8).  Chunks that start past the row re-read
// the row's last chunk; callers mask columns >= n themselves.
template <int C>
__device__ __forceinline__ void topk_load_keys(const uint16_t* keys_s, uint32_t c0, uint32_t n, uint32_t cpt,
                                               uint32_t (&key)[C]) {
    if (cpt == 8 && C >= 8) {
        const uint32_t cb = c0 < n ? c0 : (n - 1) & ~7u;
        const uint4 q4 = *reinterpret_cast<const uint4*>(keys_s + cb);
        const uint32_t w[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
        for (int i = 0; i < C; ++i) key[i] = i < 8 ? ((i & 1) ? w[(i >> 1) & 3] >> 16 : w[(i >> 1) & 3] & 0xffffu) : 0u;
    } else if (cpt == 4 && C >= 4) {
        const uint32_t cb = c0 < n ? c0 : (n - 1) & ~3u;
        const uint2 q2 = *reinterpret_cast<const uint2*>(keys_s + cb);
        const uint32_t w[2] = {q2.x, q2.y};
#pragma unroll
        for (int i = 0; i < C; ++i) key[i] = i < 4 ? ((i & 1) ? w[(i >> 1) & 1] >> 16 : w[(i >> 1) & 1] & 0xffffu) : 0u;
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const uint32_t c = c0 + i;
            key[i] = keys_s[c < n ? c : n - 1];
        }
    }
}

// Precondition: topk_clear() + topk_publish_range() + __syncthreads() already done; key[i] holds column
// tid*cpt + i for i < cpt.
// Shared tail of the selection routines: with T / need known (cur), count the thread's keys above and at the threshold
// and turn the block-wide exclusive prefix into the cursor (output slot of the thread's first selected column, rank of
// its first tied column).
// `valid(i)`: whether key[i] is a live entry of the thread (the contiguous-ownership callers pass "i < cpt && c0 + i < n";
// the tiles front end owns compacted candidates and passes its own mask).
template <int NT, int C, typename Valid>
__device__ __forceinline__ TopkCursor topk_finish_v(TopkSmem<NT>& sm, const uint32_t (&key)[C], Valid valid, TopkCursor cur,
                                                   long long* sub) {
    uint32_t gt = 0, eq = 0;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const bool in = valid(i);
        gt += in && key[i] > cur.T;
        eq += in && key[i] == cur.T;
    }
    const uint32_t packed = gt | (eq << 16);  // both totals < 65536 (n <= 16384)
    QUEST_SUBSTAMP(7);
    const uint32_t before = block_scan_incl<NT>(packed, sm.wave_tot[1]) - packed;
    QUEST_SUBSTAMP(8);
    cur.eq_rank = before >> 16;
    cur.pos = (before & 0xffffu) + (cur.eq_rank < cur.need ? cur.eq_rank : cur.need);
    (void)sub;
    return cur;
}

template <int NT, int C>
__device__ __forceinline__ TopkCursor topk_finish(TopkSmem<NT>& sm, const uint32_t (&key)[C], uint32_t n, uint32_t cpt,
                                                 TopkCursor cur, long long* sub) {
    const uint32_t c0 = threadIdx.x * cpt;
    return topk_finish_v<NT, C>(sm, key, [&](int i) { return (uint32_t)i < cpt && c0 + i < n; }, cur, sub);
}

template <int NT, int C, typename Valid>
__device__ __forceinline__ TopkCursor topk_select_v(TopkSmem<NT>& sm, const uint32_t (&key)[C], Valid valid, uint32_t k,
                                                    long long* sub = nullptr) {
    constexpr int BPT = kBins1 / NT;  // histogram bins per thread in the suffix scan
    static_assert(kBins1 % NT == 0 && NT >= kWave, "thread count must divide the bin count");
    const uint32_t tid = threadIdx.x;

    // Bins are taken over the row's own key range [kmin, kmax], not over the 16-bit key space: page scores
    // of one head sit in one or two binades, where fixed top-11-bit bins put ~100 keys on each of ~20
    // addresses and the LDS atomics serialise (measured 1.2 us of the 3 us selection); spread over up to
    // 2048 bins of width 2^shift they do not collide, and the threshold bin holds a handful of keys.
    uint32_t mm = kMmNeutral;
#pragma unroll
    for (int w = 0; w < NT / kWave; ++w) mm = pk_max_u16(mm, sm.wave_mm[w]);
    const uint32_t kmax = mm >> 16, kmin = 0xffffu - (mm & 0xffffu);
    const uint32_t range = kmax - kmin;  // < 65536
    const uint32_t bits = 32u - (uint32_t)__builtin_clz(range | 1u);
    const uint32_t shift = bits > 11u ? bits - 11u : 0u;  // <= kLowBits
    const uint32_t low_mask = (1u << shift) - 1u;
#pragma unroll
    for (int i = 0; i < C; ++i)
        if (valid(i)) atomicAdd(&sm.hist1[(key[i] - kmin) >> shift], 1u);
    QUEST_SUBSTAMP(0);
    __syncthreads();
    QUEST_SUBSTAMP(1);

    {  // suffix scan from the top bin: thread t owns bins kBins1-1-BPT*t .. kBins1-BPT*(t+1), descending
        // the thread's BPT bins are one contiguous, BPT*4-byte aligned block: fetch it with 8/16-byte LDS
        // reads (word reads at a stride of BPT words are BPT-way bank conflicts: ~1 us at BPT = 8)
        uint32_t h[BPT], tot = 0;
        {
            const uint32_t* blk = &sm.hist1[kBins1 - BPT * (tid + 1)];  // ascending bins; h[] is descending
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
            if (run < k && k <= run + h[j]) {
                sm.misc[0] = kBins1 - 1 - (BPT * tid + j);
                sm.misc[1] = run;
            }
            run += h[j];
        }
    }
    __syncthreads();
    QUEST_SUBSTAMP(4);
    const uint32_t thr_bin = sm.misc[0];
    TopkCursor cur;
    if (shift == 0) {
        // the row's keys span fewer than 2048 values (the usual case: one binade of fp16 scores is 1024 keys):
        // a bin IS a key value, so the threshold bin is T itself -- no second histogram, two barriers less
        cur.T = kmin + thr_bin;
        cur.need = k - sm.misc[1];
        QUEST_SUBSTAMP(5);
        QUEST_SUBSTAMP(6);
    } else {
#pragma unroll
        for (int i = 0; i < C; ++i)
            if (valid(i) && ((key[i] - kmin) >> shift) == thr_bin)
                atomicAdd(&sm.hist2[(key[i] - kmin) & low_mask], 1u);
        __syncthreads();
        QUEST_SUBSTAMP(5);

        if (tid < kWave) {  // wave 0: lane l looks at low digit 31-l, suffix sums by shuffle
            const uint32_t above0 = sm.misc[1];
            const uint32_t cnt = tid < kBins2 ? sm.hist2[kBins2 - 1 - tid] : 0u;
            const uint32_t incl = wave_scan_incl_dpp(cnt);
            const uint32_t excl = incl - cnt;
            if (tid < kBins2 && above0 + excl < k && k <= above0 + incl) {
                sm.misc[2] = kmin + ((thr_bin << shift) | (kBins2 - 1 - tid));
                sm.misc[3] = k - (above0 + excl);
            }
        }
        __syncthreads();
        QUEST_SUBSTAMP(6);
        cur.T = sm.misc[2];
        cur.need = sm.misc[3];
    }

    return topk_finish_v<NT, C>(sm, key, valid, cur, sub);
}

template <int NT, int C>
__device__ __forceinline__ TopkCursor topk_select(TopkSmem<NT>& sm, const uint32_t (&key)[C], uint32_t n, uint32_t k,
                                                  uint32_t cpt, long long* sub = nullptr) {
    const uint32_t c0 = threadIdx.x * cpt;
    return topk_select_v<NT, C>(sm, key, [&](int i) { return (uint32_t)i < cpt && c0 + i < n; }, k, sub);
}

// (A low-bits variant -- bins = key & 2047, so that the range publish and the histogram atomics share one barrier, with
// the cleared histogram published under the score loads' latency -- was built in round 3 for the register-ownership front
// end: bit-identical, 0.75 us FASTER per workgroup in the warm in-kernel timeline, 0.7-0.8 us SLOWER per launch in the
// bench (12.80 vs 12.02 us; no fallback taken, no extra registers; unexplained).  Removed)

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
