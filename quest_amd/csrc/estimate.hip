// Page-criticality estimate: o[h][p] = fp16( sum_d max(q[h][d]*Kmax[p][h][d], q[h][d]*Kmin[p][h][d]) )
// for every KV page p except the current (last) one.
//
// Reference behaviour restated (not translated): MaxPossibleSampleWithPagedKVCacheKernel,
// kernels/include/decode/decode_attn.cuh:245-401, arithmetic compute_max_possible :137-168.
// The reference launches one block per kv head (grid (1, H), :1131) which starves a 256-CU
// part; here the page axis is tiled too: one workgroup = 16 metadata entries x HPB heads, so
// cfg 3 (2047 entries, 32 heads) is 1024 workgroups, each streaming 32 KiB with all of its
// loads in flight at once (8 x 16 B per lane), no LDS.
//
// Bit-exactness: per lane 8 consecutive features are accumulated left to right in fp32, rows
// are reduced with the xor butterfly (offsets LPR/2..1), one fp32->fp16 RNE cast -- the
// reference kernel's order, which the oracle (qo_estimate) restates.  Bound: HBM.
#include "quest_common.cuh"

namespace quest {

constexpr int kEstTile = 16;  // metadata entries per workgroup

template <int D, int G, bool HND>
__global__ __launch_bounds__(256) void estimate_kernel(const half_t* __restrict__ q, half_t* __restrict__ o,
                                                       quest_paged_kv_t meta, uint32_t n_out) {
    constexpr int LPR = D / kVec;      // lanes per row
    constexpr int R = kWave / LPR;     // rows per wave instruction
    constexpr int ITER = HND ? kEstTile / R : kEstTile / 4;
    constexpr int HPB = HND ? 4 : R;   // heads per block

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane / LPR, col = lane % LPR;
    const uint32_t e0 = blockIdx.x * kEstTile;
    const uint32_t hk = blockIdx.y * HPB + (HND ? wave : row);
    const uint32_t Hkv = meta.num_heads, S = meta.page_size;
    const bool head_ok = hk < Hkv;
    const PoolStrides ms = pool_strides(meta);
    const half_t* data = reinterpret_cast<const half_t*>(meta.data);
    const int32_t* idx = meta.indices + meta.indptr[0];

    half8 mx[ITER], mn[ITER];
    uint32_t ent[ITER];
    bool ok[ITER];
#pragma unroll
    for (int j = 0; j < ITER; ++j) {
        const uint32_t e = HND ? e0 + j * R + row : e0 + wave * ITER + j;
        ent[j] = e;
        ok[j] = head_ok && e < n_out;
        mx[j] = (half8)(0);
        mn[j] = (half8)(0);
        if (ok[j]) {
            const size_t page = (size_t)idx[e / S];
            const half_t* p = data + page * ms.page + (size_t)hk * ms.head + (size_t)(e % S) * ms.entry + col * kVec;
            mx[j] = ld8(p);
            mn[j] = ld8(p + ms.v_off);
        }
    }

    float8 qv[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
        qv[g] = head_ok ? to_f32(ld8(q + ((size_t)hk * G + g) * D + col * kVec)) : (float8)(0.f);

#pragma unroll
    for (int j = 0; j < ITER; ++j) {
        const float8 a = to_f32(mx[j]), b = to_f32(mn[j]);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < kVec; ++i) acc += __builtin_fmaxf(qv[g][i] * a[i], qv[g][i] * b[i]);
            acc = row_allreduce_sum<LPR>(acc);
            if (ok[j] && col == 0) o[((size_t)hk * G + g) * n_out + ent[j]] = (half_t)acc;
        }
    }
}

template <int D, int G>
static int launch_estimate(const void* q, void* o, uint32_t n_out, const quest_paged_kv_t& meta, hipStream_t s) {
    constexpr int R = kWave / (D / kVec);
    const bool hnd = meta.layout == QUEST_LAYOUT_HND;
    const uint32_t hpb = hnd ? 4 : R;
    dim3 grid((n_out + kEstTile - 1) / kEstTile, (meta.num_heads + hpb - 1) / hpb);
    if (hnd)
        hipLaunchKernelGGL((estimate_kernel<D, G, true>), grid, dim3(256), 0, s, (const half_t*)q, (half_t*)o, meta, n_out);
    else
        hipLaunchKernelGGL((estimate_kernel<D, G, false>), grid, dim3(256), 0, s, (const half_t*)q, (half_t*)o, meta, n_out);
    QUEST_LAUNCH_CHECK();
    return 0;
}

template <int D>
static int dispatch_group(uint32_t G, const void* q, void* o, uint32_t n_out, const quest_paged_kv_t& meta, hipStream_t s) {
    switch (G) {
        case 1: return launch_estimate<D, 1>(q, o, n_out, meta, s);
        case 2: return launch_estimate<D, 2>(q, o, n_out, meta, s);
        case 4: return launch_estimate<D, 4>(q, o, n_out, meta, s);
        case 8: return launch_estimate<D, 8>(q, o, n_out, meta, s);
        default: return QUEST_EUNSUPPORTED;
    }
}

}  // namespace quest

using namespace quest;

extern "C" int quest_estimate_attn_score(const void* q, void* o, uint32_t num_qo_heads, uint32_t n_out,
                                         quest_paged_kv_t metadata, quest_stream_t stream) {
    if (!q || !metadata.data || !metadata.indices || !metadata.indptr) return QUEST_EINVAL;
    if (metadata.layout > QUEST_LAYOUT_HND || metadata.num_heads == 0 || metadata.page_size == 0) return QUEST_EINVAL;
    if (num_qo_heads == 0 || num_qo_heads % metadata.num_heads != 0) return QUEST_EINVAL;
    if (n_out == 0) return 0;  // nothing to score (single page)
    if (!o) return QUEST_EINVAL;
    const uint32_t G = num_qo_heads / metadata.num_heads;
    hipStream_t s = (hipStream_t)stream;
    switch (metadata.head_dim) {
        case 64: return dispatch_group<64>(G, q, o, n_out, metadata, s);
        case 128: return dispatch_group<128>(G, q, o, n_out, metadata, s);
        case 256: return dispatch_group<256>(G, q, o, n_out, metadata, s);
        default: return QUEST_EUNSUPPORTED;
    }
}
