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

// C = columns per thread (compile-time bound, n <= C * 1024).
//
// Loads are COALESCED (element t + i*1024 for thread t) and parked in LDS as 16-bit keys (+ the page ids
// when they fit); every thread then reads back the contiguous columns [t*cpt, t*cpt+cpt) it owns for the
// selection.  (Loading the owned columns directly is a 16 B-strided 2-byte gather per instruction: at
// 8191 columns that alone cost ~6 us of address-unit time.)
constexpr uint32_t kTkStageIdsMax = 8192;  // page ids are staged in LDS up to this row length (32 KiB)

template <int C>
__global__ __launch_bounds__(kTkThreads) void topk_kernel(const uint16_t* __restrict__ vals,
                                                          const int32_t* __restrict__ in_idx,
                                                          uint16_t* __restrict__ out_val,
                                                          int32_t* __restrict__ out_idx, uint32_t n, uint32_t k) {
    __shared__ TopkSmem<kTkThreads> sm;
    extern __shared__ __attribute__((aligned(16))) unsigned char tk_dyn[];
    uint16_t* keys_s = reinterpret_cast<uint16_t*>(tk_dyn);
    const bool stage_ids = n <= kTkStageIdsMax;
    int32_t* ids_s = reinterpret_cast<int32_t*>(tk_dyn + (((size_t)n * 2 + 15) & ~(size_t)15));

    const uint32_t tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const uint16_t* v = vals + row * n;
    const int32_t* ii = in_idx + row * n;
    const uint32_t cpt = topk_cols_per_thread<kTkThreads>(n);
    const uint32_t c0 = tid * cpt;

    uint16_t kraw[C];
    int32_t iraw[C];
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const uint32_t e = tid + i * kTkThreads, ec = e < n ? e : n - 1;  // clamped, unconditional
        kraw[i] = v[ec];
        iraw[i] = stage_ids ? ii[ec] : 0;
    }
    topk_clear<kTkThreads>(sm);  // overlaps the loads above
    uint32_t mm = kMmNeutral;
#pragma unroll
    for (int i = 0; i < C; ++i) {
        const uint32_t e = tid + i * kTkThreads;
        if (e < n) {
            const uint32_t kk = half_key(kraw[i]);
            mm = pk_max_u16(mm, mm_pack(kk));
            keys_s[e] = (uint16_t)kk;
            if (stage_ids) ids_s[e] = iraw[i];
        }
    }
    topk_publish_range<kTkThreads>(sm, mm);
    __syncthreads();
    uint32_t key[C];
    topk_load_keys<C>(keys_s, c0, n, cpt, key);
    TopkCursor cur = topk_select<kTkThreads, C>(sm, key, n, k, cpt);
    // Selected (value, id) pairs are compacted in LDS by output slot (k <= n), then written out by the
    // first k threads: coalesced stores instead of a divergent scatter of 2- and 4-byte writes.
    uint16_t* selv_s = keys_s;  // the key array is dead after the reads above ...
    __syncthreads();            // ... once every thread has read its keys
    int32_t* seli_s = reinterpret_cast<int32_t*>(tk_dyn + (((size_t)n * 2 + 15) & ~(size_t)15) + (stage_ids ? (size_t)n * 4 : 0));
#pragma unroll
    for (int i = 0; i < C; ++i) {
        uint32_t slot;
        if (topk_take(cur, key[i], (uint32_t)i < cpt && c0 + i < n, slot)) {
            selv_s[slot] = key_to_half_bits(key[i]);
            seli_s[slot] = stage_ids ? ids_s[c0 + i] : ii[c0 + i];
        }
    }
    __syncthreads();
    uint16_t* ov = out_val + row * k;
    int32_t* oi = out_idx + row * k;
    for (uint32_t t = tid; t < k; t += kTkThreads) {
        ov[t] = selv_s[t];
        oi[t] = seli_s[t];
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
    const size_t tk_lds = (((size_t)num_pages * 2 + 15) & ~(size_t)15) + (num_pages <= kTkStageIdsMax ? (size_t)num_pages * 4 : 0) +
                          (size_t)page_budget * 4;  // keys, staged ids, compacted selected ids
#define QUEST_TOPK_LAUNCH(CC)                                                                                   \
    hipLaunchKernelGGL((topk_kernel<CC>), dim3(num_heads), dim3(kTkThreads), tk_lds, s, ev, estimated_indices, dv, \
                       indices_out, num_pages, page_budget)
    // (a single-wave selection variant was built and measured 2x slower on MI355X: one wave issues ~1
    // instruction per 4-5 cycles, so its ~3.5k instructions cost > 8 us; removed)
    if (num_pages <= 1 * kTkThreads) QUEST_TOPK_LAUNCH(1);
    else if (num_pages <= 2 * kTkThreads) QUEST_TOPK_LAUNCH(2);
    else if (num_pages <= 4 * kTkThreads) QUEST_TOPK_LAUNCH(4);
    else if (num_pages <= 8 * kTkThreads) QUEST_TOPK_LAUNCH(8);
    else QUEST_TOPK_LAUNCH(16);
#undef QUEST_TOPK_LAUNCH
    QUEST_LAUNCH_CHECK();
    return 0;
}
