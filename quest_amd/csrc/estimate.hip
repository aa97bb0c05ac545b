// Page-criticality estimate: o[h][p] = fp16( sum_d max(q[h][d]*Kmax[p][h][d], q[h][d]*Kmin[p][h][d]) )
// for every KV page p except the current (last) one.
//
// Reference behaviour restated (not translated): MaxPossibleSampleWithPagedKVCacheKernel,
// kernels/include/decode/decode_attn.cuh:245-401, arithmetic compute_max_possible :137-168.
// The reference launches one block per kv head (grid (1, H), :1131) which starves a 256-CU
// part; here the (entry, head) rows are tiled in memory order: one workgroup = 64 rows
// (2 entries x 32 heads at cfg 3 -> 1024 workgroups), each streaming 2 x 16 KiB contiguous
// with all of its loads in flight at once (8 x 16 B per lane), no LDS.
//
// Bit-exactness: per lane 8 consecutive features are accumulated left to right in fp32 (as the
// reference kernel does, decode_attn.cuh:152-156); the 16 lanes of a row are then reduced with the
// DPP rotation tree of row_allreduce_sum_fast (row_ror 8,4,2,1 -- one VALU op per step instead of an
// LDS-crossbar ds_bpermute round trip per step of the reference's xor butterfly, which cost 0.6 us at
// MHA and 2.3 us at GQA-4 here), lane 0's value is cast once to fp16 (RNE).  The oracle (qo_estimate)
// restates exactly this tree, so HIP == oracle bit for bit.  Bound: HBM.
#include "append_device.cuh"

namespace quest {

#ifndef QUEST_EST_ITER
#define QUEST_EST_ITER 4
#endif
constexpr int kEstIter = QUEST_EST_ITER;  // load instructions per tensor per wave, all in flight together

// Work is the flat list of (entry, kv head) rows in MEMORY order, so consecutive rows are
// consecutive 256 B vectors: NHD -> entry-major (all heads of an entry are 8 KiB contiguous for
// H=32), HND -> (page, head, slot).  A wave takes kEstIter*R consecutive rows, a workgroup 4 waves.
struct AppendTail {  // optional decode-append riding in the same launch (blocks >= est_blocks)
    quest_paged_kv_t kv;
    const uint16_t* key;
    const uint16_t* value;
    uint32_t est_blocks;
    uint32_t enabled;
    const quest_step_state_t* state;  // optional device-resident lengths / last-page ids (graph replay)
    uint32_t o_stride;                // row stride of o (== n_out unless state-driven)
};

template <int D, int G, bool HND>
__global__ __launch_bounds__(256) void estimate_kernel(const half_t* __restrict__ q, half_t* __restrict__ o,
                                                       quest_paged_kv_t meta, uint32_t n_out, AppendTail tail) {
    // n_out as passed bounds every address (state-driven launches pass the largest n_out the graph will
    // see; page tables and pools cover it).  The live n_out is read from the state AFTER the page-table
    // loads are issued, so that scalar load overlaps them instead of preceding them.
    const uint32_t n_cap = n_out;
    if (tail.state && tail.enabled && blockIdx.x >= tail.est_blocks) {
        const quest_step_state_t st = *tail.state;
        meta.last_page_len = (uint32_t)st.meta_last_page_len;
        meta.last_page_idx = st.meta_last_page_idx;
        tail.kv.last_page_len = (uint32_t)st.kv_last_page_len;
        tail.kv.last_page_idx = st.kv_last_page_idx;
    }
    if (tail.enabled && blockIdx.x >= tail.est_blocks) {
        // The appended token only touches the CURRENT page's KV entry and metadata entry (index n_out),
        // which the estimate excludes (e < n_out), so the two halves of the launch share no byte.
        append_decode_body(tail.kv, meta, tail.key, tail.value, (blockIdx.x - tail.est_blocks) * 256 + threadIdx.x);
        return;
    }
    constexpr int LPR = D / kVec;   // lanes per row
    constexpr int R = kWave / LPR;  // rows per load instruction

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane / LPR, col = lane % LPR;
    const uint32_t Hkv = meta.num_heads, S = meta.page_size;
    const PoolStrides ms = pool_strides(meta);
    const half_t* data = reinterpret_cast<const half_t*>(meta.data);
    const int32_t* idx = meta.indices;  // batch_size == 1: indptr[0] == 0 (estimate.cu:14)
    const uint32_t row0 = (blockIdx.x * 4 + wave) * (kEstIter * R) + row;
    half8 mx[kEstIter], mn[kEstIter];
    // MHA: q is requested together with the metadata (one extra 16 B load per row).  GQA: a row needs the
    // G query vectors of its kv head; all Hq vectors (8 KiB for 32 x 128) are staged in LDS once per
    // workgroup -- issued after the metadata loads so both are in flight together -- and read per (row, g).
    constexpr int QPRE = (G == 1) ? 1 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char est_smem[];
    half_t* q_s = reinterpret_cast<half_t*>(est_smem);
    float8 qv[kEstIter][QPRE ? 1 : 1];
    uint32_t ent[kEstIter], head[kEstIter];
    bool ok[kEstIter];
    size_t page[kEstIter];
    uint32_t ecl[kEstIter];
#pragma unroll
    for (int j = 0; j < kEstIter; ++j) {
        const uint32_t r = row0 + j * R;
        uint32_t e, hk;
        if (HND) {
            const uint32_t per_page = Hkv * S;
            const uint32_t pg = r / per_page, rem = r % per_page;
            hk = rem / S;
            e = pg * S + rem % S;
        } else {
            e = r / Hkv;
            hk = r % Hkv;
        }
        ent[j] = e;
        head[j] = hk;
        // Loads are unconditional from a clamped entry: a predicated load compiles to branch + load +
        // wait and serialises the kEstIter round trips.  Clamped rows re-read the last entry (tail only).
        ecl[j] = e < n_cap ? e : n_cap - 1;
        page[j] = (size_t)idx[ecl[j] / S];
    }
    if (tail.state) {  // live length (<= n_cap); whole workgroups past it have nothing to do
        n_out = (uint32_t)(tail.state->n_pages - 1);
        const uint32_t first = blockIdx.x * 4 * (kEstIter * R);
        const uint32_t first_entry = HND ? (first / (Hkv * S)) * S : first / Hkv;
        if (first_entry >= n_out) return;
    }
#pragma unroll
    for (int j = 0; j < kEstIter; ++j) {
        ok[j] = ent[j] < n_out;
        const half_t* p = data + page[j] * ms.page + (size_t)head[j] * ms.head + (size_t)(ecl[j] % S) * ms.entry + col * kVec;
        mx[j] = ld8_stream(p);
        mn[j] = ld8_stream(p + ms.v_off);
        if (QPRE) qv[j][0] = to_f32(ld8(q + (size_t)head[j] * D + col * kVec));
    }
    if (!QPRE) {
        const uint32_t total = Hkv * G * D;
        for (uint32_t i = threadIdx.x * kVec; i < total; i += 256 * kVec) st8(q_s + i, ld8(q + i));
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < kEstIter; ++j) {
        const float8 a = to_f32(mx[j]), b = to_f32(mn[j]);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float8 qg;
            if (QPRE) qg = qv[j][0];
            else qg = to_f32(ld8(q_s + ((size_t)head[j] * G + g) * D + col * kVec));
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < kVec; ++i) acc += __builtin_fmaxf(qg[i] * a[i], qg[i] * b[i]);
            acc = row_allreduce_sum_fast<LPR>(acc);
            if (ok[j] && col == 0) o[((size_t)head[j] * G + g) * tail.o_stride + ent[j]] = (half_t)acc;
        }
    }
}

template <int D, int G>
static int launch_estimate(const void* q, void* o, uint32_t n_out, const quest_paged_kv_t& meta, AppendTail tail,
                           hipStream_t s) {
    constexpr int R = kWave / (D / kVec);
    const bool hnd = meta.layout == QUEST_LAYOUT_HND;
    if (!tail.state) tail.o_stride = n_out;
    const uint64_t entries = hnd ? (uint64_t)((n_out + meta.page_size - 1) / meta.page_size) * meta.page_size : n_out;
    const uint64_t rows = entries * meta.num_heads;
    const uint32_t rows_per_block = 4 * kEstIter * R;
    tail.est_blocks = (uint32_t)((rows + rows_per_block - 1) / rows_per_block);
    uint32_t blocks = tail.est_blocks;
    if (tail.enabled) blocks += (meta.num_heads * (D / kVec) + 255) / 256;
    if (blocks == 0) return 0;
    dim3 grid(blocks);
    const size_t lds = G == 1 ? 0 : (size_t)meta.num_heads * G * D * sizeof(half_t);  // staged q (GQA only)
    if (lds > 64 * 1024) return QUEST_EUNSUPPORTED;
    if (hnd)
        hipLaunchKernelGGL((estimate_kernel<D, G, true>), grid, dim3(256), lds, s, (const half_t*)q, (half_t*)o, meta, n_out, tail);
    else
        hipLaunchKernelGGL((estimate_kernel<D, G, false>), grid, dim3(256), lds, s, (const half_t*)q, (half_t*)o, meta, n_out, tail);
    QUEST_LAUNCH_CHECK();
    return 0;
}

template <int D>
static int dispatch_group(uint32_t G, const void* q, void* o, uint32_t n_out, const quest_paged_kv_t& meta,
                          const AppendTail& tail, hipStream_t s) {
    switch (G) {
        case 1: return launch_estimate<D, 1>(q, o, n_out, meta, tail, s);
        case 2: return launch_estimate<D, 2>(q, o, n_out, meta, tail, s);
        case 4: return launch_estimate<D, 4>(q, o, n_out, meta, tail, s);
        case 8: return launch_estimate<D, 8>(q, o, n_out, meta, tail, s);
        default: return QUEST_EUNSUPPORTED;
    }
}

}  // namespace quest

using namespace quest;

namespace quest {
int check_pool(const quest_paged_kv_t& p);  // append.hip
}

static int estimate_entry(const void* q, void* o, uint32_t num_qo_heads, uint32_t n_out, const quest_paged_kv_t& metadata,
                          const AppendTail& tail, hipStream_t s) {
    if (!q || !metadata.data || !metadata.indices) return QUEST_EINVAL;
    if (metadata.layout > QUEST_LAYOUT_HND || metadata.num_heads == 0 || metadata.page_size == 0) return QUEST_EINVAL;
    if (num_qo_heads == 0 || num_qo_heads % metadata.num_heads != 0) return QUEST_EINVAL;
    if (n_out > 0 && !o) return QUEST_EINVAL;
    if (n_out == 0 && !tail.enabled) return 0;  // nothing to score (single page)
    const uint32_t G = num_qo_heads / metadata.num_heads;
    switch (metadata.head_dim) {
        case 64: return dispatch_group<64>(G, q, o, n_out, metadata, tail, s);
        case 128: return dispatch_group<128>(G, q, o, n_out, metadata, tail, s);
        case 256: return dispatch_group<256>(G, q, o, n_out, metadata, tail, s);
        default: return QUEST_EUNSUPPORTED;
    }
}

extern "C" int quest_estimate_attn_score(const void* q, void* o, uint32_t num_qo_heads, uint32_t n_out,
                                         quest_paged_kv_t metadata, quest_stream_t stream) {
    AppendTail tail{};
    return estimate_entry(q, o, num_qo_heads, n_out, metadata, tail, (hipStream_t)stream);
}

extern "C" int quest_append_estimate_dyn(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                         uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out,
                                         quest_paged_kv_t metadata, const quest_step_state_t* state,
                                         quest_stream_t stream) {
    if (!k || !v || !state || max_n_out == 0 || o_stride < max_n_out) return QUEST_EINVAL;
    kv.last_page_len = metadata.last_page_len = 1;  // placeholders; the kernel reads the real ones from `state`
    if (int e = check_pool(kv)) return e;
    if (int e = check_pool(metadata)) return e;
    if (kv.num_heads != metadata.num_heads || kv.head_dim != metadata.head_dim) return QUEST_EINVAL;
    AppendTail tail{};
    tail.kv = kv;
    tail.key = (const uint16_t*)k;
    tail.value = (const uint16_t*)v;
    tail.enabled = 1;
    tail.state = state;
    tail.o_stride = o_stride;
    return estimate_entry(q, o, num_qo_heads, max_n_out, metadata, tail, (hipStream_t)stream);
}

extern "C" int quest_append_estimate(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                     uint32_t num_qo_heads, uint32_t n_out, quest_paged_kv_t metadata,
                                     quest_stream_t stream) {
    if (!k || !v) return QUEST_EINVAL;
    if (int e = check_pool(kv)) return e;
    if (int e = check_pool(metadata)) return e;
    if (kv.num_heads != metadata.num_heads || kv.head_dim != metadata.head_dim) return QUEST_EINVAL;
    AppendTail tail{};
    tail.kv = kv;
    tail.key = (const uint16_t*)k;
    tail.value = (const uint16_t*)v;
    tail.enabled = 1;
    return estimate_entry(q, o, num_qo_heads, n_out, metadata, tail, (hipStream_t)stream);
}
