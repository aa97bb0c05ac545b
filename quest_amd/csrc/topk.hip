// Device top-K page select, one workgroup per head row, deterministic.
//
// Replaces decode_select_k (kernels/include/topk/decode_select_k.cuh:25-62), which calls RAFT's
// radix_topk_one_block_kernel<half,int32,8,512> (raft branch-24.02, un-vendored; no libraft on
// ROCm).  Same contract -- the k largest of n fp16 scores per row, emitted as
// (score, in_idx[column]) -- plus the tie/output rule this build declares (SURVEY.md 8a T-tie,
// oracle qo_topk_row): everything above the threshold key, then the LOWEST columns among keys
// equal to it, written in ascending column order.  RAFT leaves both to atomic arrival order.
//
// Method: scores become order-preserving 16-bit keys kept in LDS; an 11-bit histogram
// (2048 bins, LDS atomics) + block suffix scan finds the threshold bin, a 5-bit histogram of
// that bin's members finds the exact threshold key; a packed block scan of (>thr, ==thr)
// counts gives every selected column its output slot.  Integer work; bound: latency
// (32-row launch, ~0.4 MiB), not HBM.
#include "quest_common.cuh"

namespace quest {

constexpr int kTkThreads = 1024;
constexpr int kTkWaves = kTkThreads / kWave;
constexpr int kLowBits = 5;
constexpr int kBins1 = 1 << (16 - kLowBits);  // 2048
constexpr int kBins2 = 1 << kLowBits;         // 32

// Inclusive block scan of one uint32 per thread (1024 threads).  `wave_tot` is LDS[kTkWaves].
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t x, uint32_t* wave_tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, kWave);
        if (lane >= off) x += y;
    }
    if (lane == kWave - 1) wave_tot[wave] = x;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int w = 0; w < kTkWaves; ++w) base += (w < wave) ? wave_tot[w] : 0u;
    __syncthreads();  // wave_tot may be reused by the caller
    return x + base;
}

__global__ __launch_bounds__(kTkThreads) void topk_kernel(const uint16_t* __restrict__ vals,
                                                          const int32_t* __restrict__ in_idx,
                                                          uint16_t* __restrict__ out_val,
                                                          int32_t* __restrict__ out_idx, uint32_t n, uint32_t k) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist1 = reinterpret_cast<uint32_t*>(smem);           // [kBins1]
    uint32_t* hist2 = hist1 + kBins1;                               // [kBins2]
    uint32_t* wave_tot = hist2 + kBins2;                            // [kTkWaves]
    uint32_t* misc = wave_tot + kTkWaves;                           // [4]: thr_bin, above, T, need_eq
    uint16_t* keys = reinterpret_cast<uint16_t*>(misc + 4);         // [n]

    const uint32_t tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const uint16_t* v = vals + row * n;

    for (uint32_t i = tid; i < kBins1; i += kTkThreads) hist1[i] = 0;
    if (tid < kBins2) hist2[tid] = 0;
    for (uint32_t i = tid; i < n; i += kTkThreads) keys[i] = (uint16_t)half_key(v[i]);
    __syncthreads();

    for (uint32_t i = tid; i < n; i += kTkThreads) atomicAdd(&hist1[keys[i] >> kLowBits], 1u);
    __syncthreads();

    // suffix scan from the top bin: thread t owns bins (kBins1-1-2t) and (kBins1-2-2t)
    {
        const uint32_t c0 = hist1[kBins1 - 1 - 2 * tid], c1 = hist1[kBins1 - 2 - 2 * tid];
        const uint32_t incl = block_scan_incl(c0 + c1, wave_tot);
        const uint32_t excl = incl - (c0 + c1);
        if (excl < k && k <= excl + c0) {
            misc[0] = kBins1 - 1 - 2 * tid;
            misc[1] = excl;
        } else if (excl + c0 < k && k <= incl) {
            misc[0] = kBins1 - 2 - 2 * tid;
            misc[1] = excl + c0;
        }
    }
    __syncthreads();
    const uint32_t thr_bin = misc[0];

    for (uint32_t i = tid; i < n; i += kTkThreads)
        if ((uint32_t)(keys[i] >> kLowBits) == thr_bin) atomicAdd(&hist2[keys[i] & (kBins2 - 1)], 1u);
    __syncthreads();

    if (tid == 0) {
        uint32_t above = misc[1];
        int d = kBins2 - 1;
        for (; d > 0; --d) {
            if (above + hist2[d] >= k) break;
            above += hist2[d];
        }
        misc[2] = (thr_bin << kLowBits) | (uint32_t)d;
        misc[3] = k - above;
    }
    __syncthreads();
    const uint32_t T = misc[2], need = misc[3];

    // emit in ascending column order: each thread owns a contiguous run of columns
    const uint32_t chunk = (n + kTkThreads - 1) / kTkThreads;
    const uint32_t c_begin = tid * chunk < n ? tid * chunk : n;
    const uint32_t c_end = c_begin + chunk < n ? c_begin + chunk : n;
    uint32_t gt = 0, eq = 0;
    for (uint32_t c = c_begin; c < c_end; ++c) {
        const uint32_t key = keys[c];
        gt += key > T;
        eq += key == T;
    }
    const uint32_t packed = gt | (eq << 16);
    const uint32_t before = block_scan_incl(packed, wave_tot) - packed;
    uint32_t eq_rank = before >> 16;
    uint32_t pos = (before & 0xffffu) + (eq_rank < need ? eq_rank : need);
    uint16_t* ov = out_val + row * k;
    int32_t* oi = out_idx + row * k;
    const int32_t* ii = in_idx + row * n;
    for (uint32_t c = c_begin; c < c_end; ++c) {
        const uint32_t key = keys[c];
        bool take = key > T;
        if (key == T) {
            take = eq_rank < need;
            ++eq_rank;
        }
        if (take) {
            ov[pos] = (key & 0x8000u) ? (uint16_t)(key & 0x7fffu) : (uint16_t)~key;
            oi[pos] = ii[c];
            ++pos;
        }
    }
}

}  // namespace quest

using namespace quest;

extern "C" int quest_topk_filtering(const void* estimated_value, const int32_t* estimated_indices, void* d_out,
                                    int32_t* indices_out, void* buf, uint32_t num_heads, uint32_t num_pages,
                                    uint32_t page_budget, quest_stream_t stream) {
    (void)buf;
    if (!estimated_value || !estimated_indices) return QUEST_EINVAL;
    if (num_heads == 0 || num_pages == 0) return QUEST_EINVAL;
    if (page_budget > num_pages) return QUEST_EINVAL;  // CHECK_GE(num_pages, page_budget), topk.cu:26
    if (page_budget == 0) return 0;
    if (!d_out || !indices_out) return QUEST_EINVAL;
    if (num_pages > QUEST_TOPK_MAX_ROW) return QUEST_ETOOLARGE;
    const size_t lds = (kBins1 + kBins2 + kTkWaves + 4) * sizeof(uint32_t) + (((size_t)num_pages + 7) & ~(size_t)7) * sizeof(uint16_t);
    hipLaunchKernelGGL(topk_kernel, dim3(num_heads), dim3(kTkThreads), lds, (hipStream_t)stream,
                       (const uint16_t*)estimated_value, estimated_indices, (uint16_t*)d_out, indices_out, num_pages,
                       page_budget);
    QUEST_LAUNCH_CHECK();
    return 0;
}
