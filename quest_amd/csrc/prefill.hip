// Prefill attention over the paged KV cache: prefill_with_paged_kv_cache (bsk_ops.h:78-86, batch_prefill.cu:27-117 ->
// BatchPrefillWithPagedKVCache, kernels/include/prefill/prefill.cuh:1008-1119, kernel :688-882) -- SURVEY 8(f)-4's
// "HIP flash-prefill", the one GEMM-shaped operator of the `_kernels` surface and so the one that belongs on MFMA.
//
// What the reference computes (restated, not translated): o[i] = softmax_j(q[i] . k[j] / sqrt(D)) v[j] over the keys
// j <= kv_len - n + i (causal; every key otherwise) of the sequence whose pages `indices` lists; the new tokens' K/V are
// already in the cache (utils/__init__.py:127-170), no rotary (RotaryMode::kNone, batch_prefill.cu:101), fp16 in/out.
//
// Shape of the kernel (gfx950): a workgroup = 4 waves = 128 query rows of one query head; a wave owns 32 of them.  Per
// 64-key tile a wave computes the TRANSPOSED scores S^T = K Q^T with v_mfma_f32_32x32x16_f16 (K rows from LDS as the A
// operand, its Q rows -- loaded once, 32 registers -- as B): a 32x32 result has its column, i.e. the QUERY, on the lane
// and 16 keys in the lane's registers, so the row maximum and sum of the online softmax are per-lane loops plus one
// exchange with lane ^ 32, the rescale factor of the output is one scalar per lane, and the fp16 probabilities are,
// register for register, the B operand of O^T += V^T P^T (guide: "an accumulator tile as the next MFMA's operand") --
// no LDS round trip for P.  V^T fragments come out of the row-major V tile with ds_read_b64_tr_b16.  K and V tiles are
// staged global -> registers -> LDS one tile ahead (two LDS buffers, one barrier per tile) in the guide's dual-use
// image (256-byte rows, 16-byte chunks XOR-swizzled by the row), conflict-free for both kinds of read.
#include "quest_common.cuh"

namespace quest {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

constexpr int kPfRows = 128;  // query rows per workgroup
constexpr int kPfKeys = 64;   // keys per tile

struct PrefillParams {
    const half_t* q;
    half_t* o;
    const half_t* kv;
    const int32_t* table;
    uint32_t n_q, kv_len, num_qo_heads, group, page_size, q_blocks, causal;
    PoolStrides st;
    float scale_log2;  // log2(e) / sqrt(D)
};

// Byte offset of 16-byte chunk `ch` of row `row` in a [rows][128 halves] tile image (guide T10, image (b)).
__device__ __forceinline__ uint32_t img_off(uint32_t row, uint32_t ch) {
    return 256u * row + 16u * (ch ^ (((row & 3u) << 2) | ((row >> 2) & 3u)));
}

template <bool S16>
__global__ __launch_bounds__(256, 2) void prefill_kernel(const PrefillParams p) {
    constexpr int D = 128;
    __shared__ __attribute__((aligned(16))) unsigned char s_img[2][2][kPfKeys * 256];  // [buffer][K, V]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;

    // workgroup -> (head, query block): the heads of an XCD are neighbours (a GQA group shares its K/V through one L2),
    // and the query blocks with the most keys under the causal mask are dispatched first
    uint32_t head, qb;
    {
        const uint32_t id = blockIdx.x;
        if ((p.num_qo_heads & 7u) == 0) {
            const uint32_t hpx = p.num_qo_heads >> 3, slot = id >> 3;
            head = (id & 7u) * hpx + slot % hpx;
            qb = slot / hpx;
        } else {
            head = id % p.num_qo_heads;
            qb = id / p.num_qo_heads;
        }
        qb = p.q_blocks - 1 - qb;
    }
    const uint32_t kv_head = head / p.group;
    const uint32_t q0 = qb * kPfRows, q0w = q0 + wave * 32;
    const uint32_t last_q = p.n_q - 1, last_key = p.kv_len - 1;
    const uint32_t shift = p.kv_len - p.n_q;  // query i sees keys <= shift + i
    const uint32_t qi = min(q0w + r, last_q);
    const uint32_t limit = p.causal ? shift + qi : last_key;                                   // this lane's query
    const uint32_t limit_lo = p.causal ? shift + min(q0w, last_q) : last_key;                  // first query of the wave
    const uint32_t limit_hi = p.causal ? shift + min(q0w + 31, last_q) : last_key;             // last query of the wave
    const uint32_t limit_wg = p.causal ? shift + min(q0 + kPfRows - 1, last_q) : last_key;
    const uint32_t n_tiles = limit_wg / kPfKeys + 1;
    const bool wave_live = q0w < p.n_q;

    // Q^T fragments (B operand): lane (r, h) holds q[row r][16 s + 8 h ..] of k-step s
    half8 qf[D / 16];
    {
        const half_t* qrow = p.q + ((size_t)qi * p.num_qo_heads + head) * D + 8 * h;
#pragma unroll
        for (int s = 0; s < D / 16; ++s) qf[s] = ld8(qrow + 16 * s);
    }

    const uint32_t srow = tid >> 4, sch = tid & 15;  // staging: row srow + 16 i, chunk sch
    const half_t* kv_head_base = p.kv + (size_t)kv_head * p.st.head + sch * 8;
    auto issue = [&](uint32_t t, half8(&kr)[4], half8(&vr)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t key = min(t * kPfKeys + srow + 16 * i, last_key);
            uint32_t pg, slot;
            if constexpr (S16) {
                pg = (uint32_t)ld_uniform_i32(p.table + min(t * 4 + i, last_key >> 4));
                slot = key & 15u;
            } else {
                const uint32_t pi = key / p.page_size;
                pg = (uint32_t)p.table[pi];
                slot = key - pi * p.page_size;
            }
            const half_t* src = kv_head_base + (size_t)pg * p.st.page + (size_t)slot * p.st.entry;
            kr[i] = ld8(src);
            vr[i] = ld8(src + p.st.v_off);
        }
    };
    auto stash = [&](uint32_t buf, const half8(&kr)[4], const half8(&vr)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t off = img_off(srow + 16 * i, sch);
            *reinterpret_cast<half8*>(&s_img[buf][0][off]) = kr[i];
            *reinterpret_cast<half8*>(&s_img[buf][1][off]) = vr[i];
        }
    };

    f32x16 oacc[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[db][e] = 0.f;
    float m = -INFINITY, l = 0.f;
    const float c = p.scale_log2;

    // per-lane pieces of the transposed-read address: 16-lane group g, lane 4 q + pp of it
    const uint32_t tg = (lane >> 4) & 1u, tq = (lane & 15u) >> 2, tp = lane & 3u;

    half8 kr[4], vr[4];
    issue(0, kr, vr);
    stash(0, kr, vr);
    __syncthreads();
    if (n_tiles > 1) issue(1, kr, vr);

    for (uint32_t t = 0; t < n_tiles; ++t) {
        const uint32_t buf = t & 1u;
        const uint32_t key0 = t * kPfKeys;
        if (wave_live && key0 <= limit_hi) {
            const unsigned char* img_k = s_img[buf][0];
            const unsigned char* img_v = s_img[buf][1];
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
#pragma unroll
                for (int s = 0; s < D / 16; ++s) {
                    const half8 kf = *reinterpret_cast<const half8*>(img_k + img_off(32 * kb + r, 2 * s + h));
                    sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[s], sacc[kb], 0, 0, 0);
                }
            }
            // keys beyond this lane's limit (causal mask, tail of the last tile) never enter the softmax
            if (key0 + kPfKeys - 1 > limit_lo) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const uint32_t key = key0 + 32 * kb + (e & 3) + 8 * (e >> 2) + 4 * h;
                        sacc[kb][e] = key <= limit ? sacc[kb][e] : -INFINITY;
                    }
            }
            float smax = sacc[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) smax = __builtin_fmaxf(smax, sacc[kb][e]);
            smax = __builtin_fmaxf(smax, lane_xor<32>(smax, lane));
            // key 0 is visible to every query, so from tile 0 on m_new is finite (a NaN score: NaN in, NaN out)
            const float m_new = __builtin_fmaxf(m, smax);
            const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);
            m = m_new;
            const float mc = m_new * c;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][e], c, -mc));
                    sacc[kb][e] = pe;
                    psum += pe;
                }
            l = __builtin_fmaf(l, alpha, psum);
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                for (int db = 0; db < D / 32; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
            }
            // O^T += V^T P^T over the tile's four 16-key steps
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                half8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (half_t)sacc[s >> 1][8 * (s & 1) + j];
#pragma unroll
                for (int db = 0; db < D / 32; ++db) {
                    half8 vf;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const uint32_t row = 16 * s + 8 * u + 4 * h + tq;
                        const uint32_t off = img_off(row, 4 * db + 2 * tg + (tp >> 1)) + 8 * (tp & 1u);
                        const short4v tv = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4v*)(uintptr_t)(
                                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)img_v + off));
                        const half4 hv = __builtin_bit_cast(half4, tv);
                        vf[4 * u + 0] = hv[0];
                        vf[4 * u + 1] = hv[1];
                        vf[4 * u + 2] = hv[2];
                        vf[4 * u + 3] = hv[3];
                    }
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[db], 0, 0, 0);
                }
            }
        }
        if (t + 1 < n_tiles) stash(buf ^ 1u, kr, vr);
        __syncthreads();
        if (t + 2 < n_tiles) issue(t + 2, kr, vr);
    }

    if (wave_live && q0w + r < p.n_q) {
        const float inv = 1.0f / (l + lane_xor<32>(l, lane));
        half_t* orow = p.o + ((size_t)(q0w + r) * p.num_qo_heads + head) * D + 4 * h;
#pragma unroll
        for (int db = 0; db < D / 32; ++db)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                half4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[e] = (half_t)(oacc[db][4 * g4 + e] * inv);
                *reinterpret_cast<half4*>(orow + 32 * db + 8 * g4) = out;
            }
    }
}

}  // namespace quest

using namespace quest;

extern "C" int quest_prefill_with_paged_kv_cache(const void* q, void* o, uint32_t n_q, uint32_t num_qo_heads,
                                                 quest_paged_kv_t kv, uint32_t n_pages_host, int causal,
                                                 quest_stream_t stream) {
    if (!q || !o || !kv.data || !kv.indices) return QUEST_EINVAL;
    if (n_q == 0 || num_qo_heads == 0 || kv.num_heads == 0 || kv.page_size == 0 || n_pages_host == 0) return QUEST_EINVAL;
    if (kv.layout != QUEST_LAYOUT_NHD && kv.layout != QUEST_LAYOUT_HND) return QUEST_EINVAL;
    if (kv.last_page_len == 0 || kv.last_page_len > kv.page_size) return QUEST_EINVAL;
    if (num_qo_heads % kv.num_heads != 0) return QUEST_EINVAL;
    if (kv.head_dim != 128) return QUEST_EUNSUPPORTED;
    const uint64_t kv_len = (uint64_t)(n_pages_host - 1) * kv.page_size + kv.last_page_len;
    if (kv_len > 0x7fffffffull || n_q > kv_len) return QUEST_EINVAL;  // the reference assumes kv_len >= qo_len
    PrefillParams p;
    p.q = static_cast<const half_t*>(q);
    p.o = static_cast<half_t*>(o);
    p.kv = static_cast<const half_t*>(kv.data);
    p.table = kv.indices;
    p.n_q = n_q;
    p.kv_len = (uint32_t)kv_len;
    p.num_qo_heads = num_qo_heads;
    p.group = num_qo_heads / kv.num_heads;
    p.page_size = kv.page_size;
    p.q_blocks = (n_q + kPfRows - 1) / kPfRows;
    p.causal = causal ? 1u : 0u;
    p.st = pool_strides(kv);
    p.scale_log2 = 1.4426950408889634f / sqrtf((float)kv.head_dim);
    const uint64_t grid = (uint64_t)p.q_blocks * num_qo_heads;
    if (grid > 0x7fffffffull) return QUEST_ETOOLARGE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (kv.page_size == 16)
        hipLaunchKernelGGL(prefill_kernel<true>, dim3((uint32_t)grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(prefill_kernel<false>, dim3((uint32_t)grid), dim3(256), 0, s, p);
    QUEST_LAUNCH_CHECK();
    return 0;
}
