// Device code of the decode-step KV append (shared by append.hip and the fused append+estimate
// launch in estimate.hip).  See append.hip for the reference citations.
#pragma once
#include "quest_common.cuh"

namespace quest {

// fp16 max/min on bit patterns: NaN-suppressing, -0 < +0 -- the exact rule the oracle states
// (oracle/quest_oracle.c qo_hmax/qo_hmin) so metadata is bit-identical.
__device__ __forceinline__ uint16_t hmax_bits(uint16_t a, uint16_t b) {
    const bool an = (a & 0x7fffu) > 0x7c00u, bn = (b & 0x7fffu) > 0x7c00u;
    uint16_t r = half_key(a) >= half_key(b) ? a : b;
    if (an) r = bn ? (uint16_t)0x7fffu : b;
    else if (bn) r = a;
    return r;
}
__device__ __forceinline__ uint16_t hmin_bits(uint16_t a, uint16_t b) {
    const bool an = (a & 0x7fffu) > 0x7c00u, bn = (b & 0x7fffu) > 0x7c00u;
    uint16_t r = half_key(a) <= half_key(b) ? a : b;
    if (an) r = bn ? (uint16_t)0x7fffu : b;
    else if (bn) r = a;
    return r;
}

typedef uint16_t ushort8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ ushort8 fold_max(ushort8 m, ushort8 k) {
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = hmax_bits(m[i], k[i]);
    return m;
}
__device__ __forceinline__ ushort8 fold_min(ushort8 m, ushort8 k) {
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = hmin_bits(m[i], k[i]);
    return m;
}

constexpr uint16_t kHalfMax = 0x7bffu;     // +65504 (CUDART_MAX_NORMAL_FP16, decode_page.cuh:430-431)
constexpr uint16_t kHalfNegMax = 0xfbffu;  // -65504

// Body of the decode append for global thread id `tid` (one thread = 8 halves of one head).
__device__ __forceinline__ void append_decode_body(const quest_paged_kv_t& kv, const quest_paged_kv_t& meta,
                                                   const uint16_t* __restrict__ key,
                                                   const uint16_t* __restrict__ value, uint32_t tid) {
    const uint32_t D = kv.head_dim, H = kv.num_heads;
    const uint32_t lpr = D / kVec;  // lanes per row
    if (tid >= H * lpr) return;
    const uint32_t h = tid / lpr, f = (tid % lpr) * kVec;

    const PoolStrides ks = pool_strides(kv), ms = pool_strides(meta);
    // The token lands in entry last_page_len-1 of the sequence's last page and its metadata in entry
    // meta.last_page_len-1 of the last metadata page (decode_page.cuh:408-420).  Both page ids are
    // passed by value (last_page_idx == indices[indptr[1]-1], the invariant the reference's decode
    // kernel also relies on, decode_page.cuh:331-333), so no dependent index load precedes the data.
    const uint32_t entry = kv.last_page_len - 1;
    const size_t page = (size_t)kv.last_page_idx;
    const size_t mpage = (size_t)meta.last_page_idx;
    const uint32_t mentry = meta.last_page_len - 1;

    uint16_t* kv_data = reinterpret_cast<uint16_t*>(kv.data);
    uint16_t* m_data = reinterpret_cast<uint16_t*>(meta.data);
    const uint32_t kslot = pool_slot(ks, h, entry), mslot = pool_slot(ms, h, mentry);
    uint16_t* kdst = kv_data + page * ks.page + (size_t)kslot * ks.head + (size_t)entry * ks.entry + f;
    uint16_t* mmax = m_data + mpage * ms.page + (size_t)mslot * ms.head + (size_t)mentry * ms.entry + f;
    uint16_t* mmin = mmax + pool_v_off(ms, mslot);

    const ushort8 k8 = *reinterpret_cast<const ushort8*>(key + (size_t)h * D + f);
    const ushort8 v8 = *reinterpret_cast<const ushort8*>(value + (size_t)h * D + f);
    ushort8 mx, mn;
    if (entry > 0) {
        mx = *reinterpret_cast<const ushort8*>(mmax);
        mn = *reinterpret_cast<const ushort8*>(mmin);
    } else {  // the token opens a new page: start from the sentinels, not from stale pool bytes
        mx = (ushort8)(kHalfNegMax);
        mn = (ushort8)(kHalfMax);
    }
    mx = fold_max(mx, k8);
    mn = fold_min(mn, k8);
    *reinterpret_cast<ushort8*>(kdst) = k8;
    *reinterpret_cast<ushort8*>(kdst + pool_v_off(ks, kslot)) = v8;
    *reinterpret_cast<ushort8*>(mmax) = mx;
    *reinterpret_cast<ushort8*>(mmin) = mn;
}

// prepare_metadata(1) of quest/utils/controller.py:72-76 + kv_cache.py:115-126 for ONE sequence, on the device: reserve room
// for one more token (a new KV page, and a new metadata page for its entry, where the last ones are full).  Run by the
// step_state_advance launch (append.hip) or by one thread of a step's LAST launch (StepAdvance below).
__device__ __forceinline__ void step_state_advance_one(quest_step_state_t* st, const int32_t* __restrict__ kv_table,
                                                       const int32_t* __restrict__ meta_table, uint32_t S,
                                                       uint32_t max_kv_pages, uint32_t max_meta_pages) {
    quest_step_state_t s = *st;
    if (s.kv_last_page_len == (int32_t)S &&
        ((uint32_t)s.n_pages >= max_kv_pages ||
         (s.meta_last_page_len == (int32_t)S && (uint32_t)s.n_meta_pages >= max_meta_pages))) {
        // pool exhausted: stay on the last token (memory-safe; the host mirror raises "KvPool exhausted"
        // right after the replay) and flag it
        st->reserved = 1;
        return;
    }
    s.seq_len += 1;
    if (s.kv_last_page_len == (int32_t)S) {  // the token opens a new KV page ...
        s.n_pages += 1;
        s.kv_last_page_len = 1;
        s.kv_last_page_idx = kv_table[s.n_pages - 1];
        if (s.meta_last_page_len == (int32_t)S) {  // ... whose metadata entry may open a new metadata page
            s.n_meta_pages += 1;
            s.meta_last_page_len = 1;
            s.meta_last_page_idx = meta_table[s.n_meta_pages - 1];
        } else {
            s.meta_last_page_len += 1;
        }
    } else {
        s.kv_last_page_len += 1;
    }
    *st = s;
}

// The NEXT step's reservation riding in the last launch of this step (round 6): the 1-thread
// step_state_advance launch at the head of a captured step costs 4.7 us of dependent latency per token.  No launch after a
// layer's merge reads the step state, so thread `seq` of the merge launch's workgroup 0 runs it after its own work; the device
// state (and its host mirror) then describe the cache one reserved token ahead between steps.
struct StepAdvance {
    quest_step_state_t* st;  // nullptr: nothing rides
    const int32_t* kv_table;
    const int32_t* meta_table;
    uint32_t page_size, max_kv_pages, max_meta_pages;
    quest_batch_t batch;
};

}  // namespace quest
