// KV append with fused per-page Key min/max metadata maintenance.
//
// Reference behaviour restated (not translated):
//   decode : AppendPagedKVCacheDecodeKernel   kernels/include/decode/decode_page.cuh:398-449
//   prefill: AppendPagedKVCachePrefillKernel  kernels/include/decode/decode_page.cuh:471-562
// The metadata pool has the same paged struct as the KV pool; its "entries" are KV pages, the
// K slot holds the element-wise max of the page's keys and the V slot the min (:443-446).
//
// gfx950 mapping: one row of D/8 lanes owns one (page, head) vector, 16 B per lane, so a wave
// moves 4 heads x 256 B = 1 KiB contiguous (NHD) per instruction.  Bound: HBM (pure copy + RMW).
#include "append_device.cuh"

namespace quest {

__global__ __launch_bounds__(256) void append_decode_kernel(quest_paged_kv_t kv, quest_paged_kv_t meta,
                                                            const uint16_t* __restrict__ key,
                                                            const uint16_t* __restrict__ value,
                                                            const quest_step_state_t* state) {
    if (state) {  // state-driven launch: lengths / last-page ids from device memory; blockIdx.y = sequence
        const quest_step_state_t st = state[blockIdx.y];
        key += (size_t)blockIdx.y * kv.num_heads * kv.head_dim;
        value += (size_t)blockIdx.y * kv.num_heads * kv.head_dim;
        kv.last_page_len = (uint32_t)st.kv_last_page_len;
        kv.last_page_idx = st.kv_last_page_idx;
        meta.last_page_len = (uint32_t)st.meta_last_page_len;
        meta.last_page_idx = st.meta_last_page_idx;
    }
    append_decode_body(kv, meta, key, value, blockIdx.x * blockDim.x + threadIdx.x);
}

// One row of lanes per (page, head); rows of a wave are consecutive heads of one page so source
// and (NHD) destination accesses are 1 KiB contiguous per wave instruction.
__global__ __launch_bounds__(256) void append_prefill_kernel(quest_paged_kv_t kv, quest_paged_kv_t meta,
                                                             const uint16_t* __restrict__ key,
                                                             const uint16_t* __restrict__ value,
                                                             uint32_t append_len) {
    const uint32_t D = kv.head_dim, H = kv.num_heads, S = kv.page_size, MS = meta.page_size;
    const uint32_t lpr = D / kVec;
    const uint32_t rows_per_block = blockDim.x / lpr;
    const uint32_t row = blockIdx.x * rows_per_block + threadIdx.x / lpr;
    const uint32_t f = (threadIdx.x % lpr) * kVec;

    const int32_t kv_begin = kv.indptr[0];
    const int32_t page_nums = kv.indptr[1] - kv_begin;
    const int32_t seq_len = (page_nums - 1) * (int32_t)S + (int32_t)kv.last_page_len;
    const int32_t start_seq = seq_len - (int32_t)append_len;
    const int32_t first_page = start_seq / (int32_t)S;
    const int32_t po = first_page + (int32_t)(row / H);
    const uint32_t h = row % H;
    if (po >= page_nums) return;

    const PoolStrides ks = pool_strides(kv), ms = pool_strides(meta);
    const size_t page = (size_t)kv.indices[kv_begin + po];
    const size_t mpage = (size_t)meta.indices[meta.indptr[0] + po / (int32_t)MS];
    const uint32_t mentry = (uint32_t)po % MS;
    int32_t e0 = start_seq - po * (int32_t)S;
    if (e0 < 0) e0 = 0;
    int32_t e1 = seq_len - po * (int32_t)S;
    if (e1 > (int32_t)S) e1 = (int32_t)S;

    uint16_t* kv_data = reinterpret_cast<uint16_t*>(kv.data);
    uint16_t* m_data = reinterpret_cast<uint16_t*>(meta.data);
    uint16_t* kdst = kv_data + page * ks.page + f;  // + entry * ks.entry + (the entry's slot of head h) * ks.head
    const uint32_t mslot = pool_slot(ms, h, mentry);
    uint16_t* mmax = m_data + mpage * ms.page + (size_t)mslot * ms.head + (size_t)mentry * ms.entry + f;
    uint16_t* mmin = mmax + pool_v_off(ms, mslot);
    const uint32_t kv_v_off = pool_v_off(ks, h);  // (the rotation leaves the flipped slot bits alone: one distance per head)

    ushort8 mx, mn;
    if (e0 > 0) {
        mx = *reinterpret_cast<const ushort8*>(mmax);
        mn = *reinterpret_cast<const ushort8*>(mmin);
    } else {
        mx = (ushort8)(kHalfNegMax);
        mn = (ushort8)(kHalfMax);
    }
    const size_t src0 = ((size_t)(po * (int32_t)S - start_seq) * H + h) * D + f;
#pragma unroll 4
    for (int32_t e = e0; e < e1; ++e) {
        const size_t src = src0 + (size_t)e * H * D;
        const ushort8 k8 = *reinterpret_cast<const ushort8*>(key + src);
        const ushort8 v8 = *reinterpret_cast<const ushort8*>(value + src);
        mx = fold_max(mx, k8);
        mn = fold_min(mn, k8);
        uint16_t* dst = kdst + (size_t)e * ks.entry + (size_t)pool_slot(ks, h, (uint32_t)e) * ks.head;
        *reinterpret_cast<ushort8*>(dst) = k8;
        *reinterpret_cast<ushort8*>(dst + kv_v_off) = v8;
    }
    *reinterpret_cast<ushort8*>(mmax) = mx;
    *reinterpret_cast<ushort8*>(mmin) = mn;
}

__global__ void step_state_advance_kernel(quest_step_state_t* st, const int32_t* __restrict__ kv_table,
                                          const int32_t* __restrict__ meta_table, uint32_t S, uint32_t max_kv_pages,
                                          uint32_t max_meta_pages, quest_batch_t batch) {
    // prepare_metadata(1) of quest/utils/controller.py:72-76 + kv_cache.py:115-126, on the device;
    // one thread per sequence (body: append_device.cuh step_state_advance_one)
    const uint32_t seq = blockIdx.x * blockDim.x + threadIdx.x;
    if (seq >= batch.n_seqs) return;
    step_state_advance_one(st + seq, kv_table + (size_t)seq * batch.kv_table_stride,
                           meta_table + (size_t)seq * batch.meta_table_stride, S, max_kv_pages, max_meta_pages);
}

extern "C" uint32_t quest_pool_slot(uint32_t layout, uint32_t num_heads, uint32_t head, uint32_t entry, int v_slot) {
    if (layout > QUEST_LAYOUT_NHD_ROT || num_heads == 0 || head >= num_heads) return 0xffffffffu;
    quest_paged_kv_t p{};
    p.layout = layout, p.num_heads = num_heads, p.page_size = 1, p.head_dim = 8;  // (only the layout and the head count matter)
    const PoolStrides st = pool_strides(p);
    const uint32_t slot = pool_slot(st, head, entry);
    return v_slot ? slot ^ st.vflip : slot;
}

int check_pool(const quest_paged_kv_t& p) {
    if (!p.data || !p.indices) return QUEST_EINVAL;
    if (p.layout > QUEST_LAYOUT_NHD_ROT) return QUEST_EINVAL;
    if (p.num_heads == 0 || p.page_size == 0) return QUEST_EINVAL;
    if (p.head_dim != 64 && p.head_dim != 128 && p.head_dim != 256) return QUEST_EUNSUPPORTED;
    if (p.last_page_len == 0 || p.last_page_len > p.page_size) return QUEST_EINVAL;
    return 0;
}

}  // namespace quest

using namespace quest;

extern "C" int quest_append_kv_cache_decode(const void* k, const void* v, quest_paged_kv_t kv,
                                            quest_paged_kv_t metadata, quest_stream_t stream) {
    if (!k || !v) return QUEST_EINVAL;
    if (int e = check_pool(kv)) return e;
    if (int e = check_pool(metadata)) return e;
    if (kv.num_heads != metadata.num_heads || kv.head_dim != metadata.head_dim) return QUEST_EINVAL;
    const uint32_t threads = kv.num_heads * (kv.head_dim / kVec);
    const uint32_t block = 256, grid = (threads + block - 1) / block;
    hipLaunchKernelGGL(append_decode_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, kv, metadata,
                       (const uint16_t*)k, (const uint16_t*)v, (const quest_step_state_t*)nullptr);
    QUEST_LAUNCH_CHECK();
    return 0;
}

extern "C" int quest_append_kv_cache_decode_dyn(const void* k, const void* v, quest_paged_kv_t kv,
                                                quest_paged_kv_t metadata, const quest_step_state_t* state,
                                                quest_stream_t stream) {
    if (!k || !v || !state) return QUEST_EINVAL;
    kv.last_page_len = metadata.last_page_len = 1;  // placeholders; the kernel reads the real ones from `state`
    if (int e = check_pool(kv)) return e;
    if (int e = check_pool(metadata)) return e;
    if (kv.num_heads != metadata.num_heads || kv.head_dim != metadata.head_dim) return QUEST_EINVAL;
    const uint32_t threads = kv.num_heads * (kv.head_dim / kVec);
    const uint32_t block = 256, grid = (threads + block - 1) / block;
    hipLaunchKernelGGL(append_decode_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, kv, metadata,
                       (const uint16_t*)k, (const uint16_t*)v, state);
    QUEST_LAUNCH_CHECK();
    return 0;
}

extern "C" int quest_append_kv_cache_prefill(const void* k, const void* v, uint32_t append_len,
                                             uint32_t n_pages_host, quest_paged_kv_t kv,
                                             quest_paged_kv_t metadata, quest_stream_t stream) {
    if (!k || !v || append_len == 0 || n_pages_host == 0) return QUEST_EINVAL;
    if (!kv.indptr || !metadata.indptr) return QUEST_EINVAL;
    if (int e = check_pool(kv)) return e;
    if (int e = check_pool(metadata)) return e;
    if (kv.num_heads != metadata.num_heads || kv.head_dim != metadata.head_dim) return QUEST_EINVAL;
    // pages touched: from the page holding the first appended token to the last page
    const uint64_t seq_len = (uint64_t)(n_pages_host - 1) * kv.page_size + kv.last_page_len;
    if (append_len > seq_len) return QUEST_EINVAL;
    const uint64_t first_page = (seq_len - append_len) / kv.page_size;
    const uint64_t pages = n_pages_host - first_page;
    const uint32_t block = 256;
    const uint32_t rows_per_block = block / (kv.head_dim / kVec);
    const uint64_t rows = pages * kv.num_heads;
    const uint32_t grid = (uint32_t)((rows + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL(append_prefill_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream, kv, metadata,
                       (const uint16_t*)k, (const uint16_t*)v, append_len);
    QUEST_LAUNCH_CHECK();
    return 0;
}

extern "C" int quest_step_state_advance(quest_step_state_t* state, const int32_t* kv_table, const int32_t* meta_table,
                                        uint32_t page_size, uint32_t max_kv_pages, uint32_t max_meta_pages,
                                        quest_stream_t stream) {
    if (!state || !kv_table || !meta_table || page_size == 0 || max_kv_pages == 0 || max_meta_pages == 0) return QUEST_EINVAL;
    const quest_batch_t one = {1, 0, 0, 0};
    hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, kv_table, meta_table,
                       page_size, max_kv_pages, max_meta_pages, one);
    QUEST_LAUNCH_CHECK();
    return 0;
}

extern "C" int quest_step_state_advance_batched(quest_step_state_t* state, const int32_t* kv_tables,
                                                const int32_t* meta_tables, uint32_t page_size, uint32_t max_kv_pages,
                                                uint32_t max_meta_pages, quest_batch_t batch, quest_stream_t stream) {
    if (!state || !kv_tables || !meta_tables || page_size == 0 || max_kv_pages == 0 || max_meta_pages == 0)
        return QUEST_EINVAL;
    if (batch.n_seqs == 0 || batch.kv_table_stride < max_kv_pages || batch.meta_table_stride < max_meta_pages)
        return QUEST_EINVAL;
    hipLaunchKernelGGL(step_state_advance_kernel, dim3((batch.n_seqs + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       state, kv_tables, meta_tables, page_size, max_kv_pages, max_meta_pages, batch);
    QUEST_LAUNCH_CHECK();
    return 0;
}

extern "C" int quest_append_kv_cache_decode_batched(const void* k, const void* v, quest_paged_kv_t kv,
                                                    quest_paged_kv_t metadata, const quest_step_state_t* state,
                                                    quest_batch_t batch, quest_stream_t stream) {
    if (!k || !v || !state || batch.n_seqs == 0) return QUEST_EINVAL;
    kv.last_page_len = metadata.last_page_len = 1;  // placeholders; the kernel reads the real ones from `state`
    if (int e = check_pool(kv)) return e;
    if (int e = check_pool(metadata)) return e;
    if (kv.num_heads != metadata.num_heads || kv.head_dim != metadata.head_dim) return QUEST_EINVAL;
    const uint32_t threads = kv.num_heads * (kv.head_dim / kVec);
    const uint32_t block = 256, grid = (threads + block - 1) / block;
    hipLaunchKernelGGL(append_decode_kernel, dim3(grid, batch.n_seqs), dim3(block), 0, (hipStream_t)stream, kv, metadata,
                       (const uint16_t*)k, (const uint16_t*)v, state);
    QUEST_LAUNCH_CHECK();
    return 0;
}
