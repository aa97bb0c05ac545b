// Device top-K page select, one workgroup per head row, deterministic.
//
// Replaces decode_select_k (kernels/include/topk/decode_select_k.cuh:25-62), which calls RAFT's
// radix_topk_one_block_kernel<half,int32,8,512> (raft branch-24.02, un-vendored; no libraft on
// ROCm).  Same contract -- the k largest of n fp16 scores per row, emitted as
// (score, in_idx[column]) -- plus the tie/output rule this build declares (SURVEY.md 8a T-tie,
// oracle qo_topk_row): everything above the threshold key, then the LOWEST columns among keys
// equal to it, written in ascending column order.  RAFT leaves both to atomic arrival order.
//
// Method: scores become order-preserving 16-bit keys kept in registers; an 11-bit histogram
// (2048 bins, LDS atomics) + block suffix scan finds the threshold bin, a 5-bit histogram of
// that bin's members finds the exact threshold key; a packed block scan of (>thr, ==thr)
// counts gives every selected column its output slot.  Integer work; bound: latency
// (32-row launch, ~0.4 MiB), not HBM.
#include "topk_select.cuh"

namespace quest {

constexpr int kTkThreads = 1024;

// C = columns per thread (compile-time bound, n <= C * 1024).  Thread t owns the contiguous columns
// [t*C, t*C+C): their keys and page ids stay in registers for the whole kernel, so the only global
// traffic is one round of loads at the top and the k output stores at the bottom.
template <int C>
__global__ __launch_bounds__(kTkThreads) void topk_kernel(const uint16_t* __restrict__ vals,
                                                          const int32_t* __restrict__ in_idx,
                                                          uint16_t* __restrict__ out_val,
                                                          int32_t* __restrict__ out_idx, uint32_t n, uint32_t k) {
    __shared__ TopkSmem<kTkThreads> sm;
    const uint32_t tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const uint16_t* v = vals + row * n;
    const int32_t* ii = in_idx + row * n;
    const uint32_t cpt = topk_cols_per_thread<kTkThreads>(n);
    const uint32_t c0 = tid * cpt;

    uint32_t key[C];
    int32_t pid[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        // unconditional loads from a clamped column (a predicated load becomes branch + load + wait,
        // which serialises the round trips); out-of-range columns are masked later
        const uint32_t c = c0 + i, cc = c < n ? c : n - 1;
        key[i] = half_key(v[cc]);
        pid[i] = ii[cc];
    }
    topk_clear<kTkThreads>(sm);  // overlaps the loads above
    __syncthreads();
    TopkCursor cur = topk_select<kTkThreads, C>(sm, key, n, k, cpt);
    uint16_t* ov = out_val + row * k;
    int32_t* oi = out_idx + row * k;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        uint32_t slot;
        if (topk_take(cur, key[i], (uint32_t)i < cpt && c0 + i < n, slot)) {
            ov[slot] = key_to_half_bits(key[i]);
            oi[slot] = pid[i];
        }
    }
}

// Rows up to 64*C columns: one 64-thread workgroup (a single wave) per row, no barriers.
template <int C>
__global__ __launch_bounds__(kWave) void topk_wave_kernel(const uint16_t* __restrict__ vals,
                                                          const int32_t* __restrict__ in_idx,
                                                          uint16_t* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                          uint32_t n, uint32_t k) {
    __shared__ TopkWaveSmem sm;
    const int lane = threadIdx.x;
    const size_t row = blockIdx.x;
    const uint16_t* v = vals + row * n;
    const int32_t* ii = in_idx + row * n;
    uint32_t key[C];
    int32_t pid[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const uint32_t c = i * kWave + lane, cc = c < n ? c : n - 1;  // clamped, unconditional (see above)
        key[i] = half_key(v[cc]);
        pid[i] = ii[cc];
    }
    const TopkWaveResult r = topk_select_wave<C>(sm, key, n, k, lane);
    TopkWaveCursor cur;
    uint16_t* ov = out_val + row * k;
    int32_t* oi = out_idx + row * k;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        uint32_t slot;
        if (topk_wave_take(cur, r, key[i], (uint32_t)(i * kWave + lane) < n, lane, slot)) {
            ov[slot] = key_to_half_bits(key[i]);
            oi[slot] = pid[i];
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
    hipStream_t s = (hipStream_t)stream;
    const uint16_t* ev = (const uint16_t*)estimated_value;
    uint16_t* dv = (uint16_t*)d_out;
#define QUEST_TOPK_LAUNCH(CC)                                                                              \
    hipLaunchKernelGGL((topk_kernel<CC>), dim3(num_heads), dim3(kTkThreads), 0, s, ev, estimated_indices, dv, \
                       indices_out, num_pages, page_budget)
#define QUEST_TOPK_WAVE(CC)                                                                                 \
    hipLaunchKernelGGL((topk_wave_kernel<CC>), dim3(num_heads), dim3(kWave), 0, s, ev, estimated_indices, dv, \
                       indices_out, num_pages, page_budget)
    // (a single-wave variant, topk_wave_kernel, exists for rows <= 4096 but measured 2x slower on
    // MI355X: one wave issues ~1 instruction / 4-5 cycles, so ~3.5k instructions cost > 8 us)
    if (num_pages <= 64) QUEST_TOPK_WAVE(16);
    else if (num_pages <= 1 * kTkThreads) QUEST_TOPK_LAUNCH(1);
    else if (num_pages <= 2 * kTkThreads) QUEST_TOPK_LAUNCH(2);
    else if (num_pages <= 4 * kTkThreads) QUEST_TOPK_LAUNCH(4);
    else if (num_pages <= 8 * kTkThreads) QUEST_TOPK_LAUNCH(8);
    else QUEST_TOPK_LAUNCH(16);
#undef QUEST_TOPK_LAUNCH
#undef QUEST_TOPK_WAVE
    QUEST_LAUNCH_CHECK();
    return 0;
}
