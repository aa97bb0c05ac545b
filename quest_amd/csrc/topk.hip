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

// One row: v / ii = the row's n scores and page ids, ov / oi = its k outputs.  All threads of the workgroup call it.
template <int C>
__device__ __forceinline__ void topk_row(TopkSmem<kTkThreads>& sm, unsigned char* tk_dyn, const uint16_t* __restrict__ v,
                                         const int32_t* __restrict__ ii, uint16_t* __restrict__ ov,
                                         int32_t* __restrict__ oi, uint32_t n, uint32_t k, bool stage_ids) {
    // stage_ids: the page ids fit the dynamic LDS next to the keys (decided by the host from the row CAPACITY)
    uint16_t* keys_s = reinterpret_cast<uint16_t*>(tk_dyn);
    int32_t* ids_s = reinterpret_cast<int32_t*>(tk_dyn + (((size_t)n * 2 + 15) & ~(size_t)15));

    const uint32_t tid = threadIdx.x;
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
    for (uint32_t t = tid; t < k; t += kTkThreads) {
        ov[t] = selv_s[t];
        oi[t] = seli_s[t];
    }
}

template <int C>
__global__ __launch_bounds__(kTkThreads) void topk_kernel(const uint16_t* __restrict__ vals,
                                                          const int32_t* __restrict__ in_idx,
                                                          uint16_t* __restrict__ out_val,
                                                          int32_t* __restrict__ out_idx, uint32_t n, uint32_t k,
                                                          uint32_t val_stride) {
    __shared__ TopkSmem<kTkThreads> sm;
    extern __shared__ __attribute__((aligned(16))) unsigned char tk_dyn[];
    const size_t row = blockIdx.x;
    topk_row<C>(sm, tk_dyn, vals + row * val_stride, in_idx + row * n, out_val + row * k, out_idx + row * k, n, k, n <= kTkStageIdsMax);
}

// Batched, state-driven form (EXTENSION; replaces the per-request Python loop around topk_filtering of a batch of
// `constexpr batch_size = 1` calls, quest/utils/controller.py:80-129 + utils/__init__.py:207-237): grid (head, sequence);
// the row length is the sequence's live page count minus the current page, read from its device-resident state; the
// page ids are the sequence's own page table (one row for all heads); k = min(budget of the sequence - 1, row length),
// the budget from `budgets` (pages INCLUDING the current one) or the launch-wide `k_plan`.  Outputs are rows of
// `out_stride` entries of which the first k are written.
template <int C>
__global__ __launch_bounds__(kTkThreads) void topk_batched_kernel(const uint16_t* __restrict__ vals, uint32_t val_stride,
                                                                  const int32_t* __restrict__ tables, uint32_t table_stride,
                                                                  uint16_t* __restrict__ out_val, int32_t* __restrict__ out_idx,
                                                                  uint32_t out_stride, const quest_step_state_t* __restrict__ state,
                                                                  const int32_t* __restrict__ budgets, uint32_t k_plan,
                                                                  uint32_t stage_ids) {
    __shared__ TopkSmem<kTkThreads> sm;
    extern __shared__ __attribute__((aligned(16))) unsigned char tk_dyn[];
    const uint32_t head = blockIdx.x, seq = blockIdx.y, heads = gridDim.x;
    const uint32_t n = (uint32_t)(state[seq].n_pages - 1);
    uint32_t k = budgets ? (uint32_t)max(budgets[seq] - 1, 0) : k_plan;
    k = min(min(k, n), out_stride);
    if (n == 0 || k == 0) return;  // block-uniform: a one-page sequence has nothing to select from
    const size_t row = (size_t)seq * heads + head;
    topk_row<C>(sm, tk_dyn, vals + row * val_stride, tables + (size_t)seq * table_stride, out_val + row * out_stride,
                out_idx + row * out_stride, n, k, stage_ids != 0);
}

}  // namespace quest

using namespace quest;

extern "C" int quest_topk_filtering_strided(const void* estimated_value, uint32_t value_stride,
                                            const int32_t* estimated_indices, void* d_out, int32_t* indices_out, void* buf,
                                            uint32_t num_heads, uint32_t num_pages, uint32_t page_budget,
                                            quest_stream_t stream) {
    (void)buf;
    if (!estimated_value || !estimated_indices) return QUEST_EINVAL;
    if (num_heads == 0 || num_pages == 0) return QUEST_EINVAL;
    if (value_stride == 0) value_stride = num_pages;
    if (value_stride < num_pages) return QUEST_EINVAL;
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
                       indices_out, num_pages, page_budget, value_stride)
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

extern "C" int quest_topk_filtering(const void* estimated_value, const int32_t* estimated_indices, void* d_out,
                                    int32_t* indices_out, void* buf, uint32_t num_heads, uint32_t num_pages,
                                    uint32_t page_budget, quest_stream_t stream) {
    return quest_topk_filtering_strided(estimated_value, 0, estimated_indices, d_out, indices_out, buf, num_heads, num_pages,
                                        page_budget, stream);
}

extern "C" int quest_topk_filtering_batched(const void* scores, uint32_t score_stride, uint32_t max_num_pages,
                                            const int32_t* kv_tables, void* d_out, int32_t* indices_out, uint32_t out_stride,
                                            uint32_t num_heads, uint32_t page_budget, const quest_step_state_t* state,
                                            quest_batch_t batch, quest_stream_t stream) {
    if (!scores || !kv_tables || !state || !d_out || !indices_out) return QUEST_EINVAL;
    if (num_heads == 0 || batch.n_seqs == 0 || batch.n_seqs > 65535u || max_num_pages == 0 || out_stride == 0) return QUEST_EINVAL;
    if (score_stride < max_num_pages || (batch.n_seqs > 1 && batch.kv_table_stride < max_num_pages)) return QUEST_EINVAL;
    if (max_num_pages > QUEST_TOPK_MAX_ROW) return QUEST_ETOOLARGE;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t k_plan = page_budget > 0 ? page_budget - 1 : 0;  // pages besides the current one
    const size_t tk_lds = (((size_t)max_num_pages * 2 + 15) & ~(size_t)15) +
                          (max_num_pages <= kTkStageIdsMax ? (size_t)max_num_pages * 4 : 0) + (size_t)out_stride * 4;
    const dim3 grid(num_heads, batch.n_seqs);
#define QUEST_TOPK_LAUNCH_B(CC)                                                                                           \
    hipLaunchKernelGGL((topk_batched_kernel<CC>), grid, dim3(kTkThreads), tk_lds, s, (const uint16_t*)scores, score_stride, \
                       kv_tables, batch.kv_table_stride, (uint16_t*)d_out, indices_out, out_stride, state, batch.page_budgets, k_plan, \
                       max_num_pages <= kTkStageIdsMax ? 1u : 0u)
    if (max_num_pages <= 1 * kTkThreads) QUEST_TOPK_LAUNCH_B(1);
    else if (max_num_pages <= 2 * kTkThreads) QUEST_TOPK_LAUNCH_B(2);
    else if (max_num_pages <= 4 * kTkThreads) QUEST_TOPK_LAUNCH_B(4);
    else if (max_num_pages <= 8 * kTkThreads) QUEST_TOPK_LAUNCH_B(8);
    else QUEST_TOPK_LAUNCH_B(16);
#undef QUEST_TOPK_LAUNCH_B
    QUEST_LAUNCH_CHECK();
    return 0;
}
