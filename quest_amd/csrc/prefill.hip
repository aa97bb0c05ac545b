// Prefill attention over the paged KV cache: prefill_with_paged_kv_cache (bsk_ops.h:78-86, batch_prefill.cu:27-117 ->
// BatchPrefillWithPagedKVCache, kernels/include/prefill/prefill.cuh:1008-1119, kernel :688-882) -- SURVEY 8(f)-4's
// "HIP flash-prefill", the one GEMM-shaped operator of the `_kernels` surface and so the one that belongs on MFMA.
//
// What the reference computes (restated, not translated): o[i] = softmax_j(q[i] . k[j] / sqrt(D)) v[j] over the keys
// j <= kv_len - n + i (causal; every key otherwise) of the sequence whose pages `indices` lists; the new tokens' K/V are
// already in the cache (utils/__init__.py:127-170), no rotary (RotaryMode::kNone, batch_prefill.cu:102), fp16 in/out.
//
// Shape of the kernel (gfx950; head_dim 64 / 128 / 256, the set of the reference's SWITCH_HEAD_DIM_PREFILL,
// prefill.cuh:1073 -- the macro itself lives in the absent flashinfer submodule; the text below is for 128): a
// workgroup = 4 waves = 128 query rows of one query head; a wave owns 32 of them.  Per
// 64-key tile a wave computes the TRANSPOSED scores S^T = K Q^T with v_mfma_f32_32x32x16_f16 (K rows from LDS as the A
// operand, its Q rows -- loaded once, 32 registers -- as B): a 32x32 result has its column, i.e. the QUERY, on the lane
// and 16 keys in the lane's registers, so the row maximum and sum of the online softmax are per-lane loops plus one
// exchange with lane ^ 32, the rescale factor of the output is one scalar per lane, and the fp16 probabilities are,
// register for register, the B operand of O^T += V^T P^T (guide: "an accumulator tile as the next MFMA's operand") --
// no LDS round trip for P.  V^T fragments come out of the row-major V tile with ds_read_b64_tr_b16.  K and V tiles are
// staged global -> registers -> LDS one tile ahead (two LDS buffers, one barrier per tile) in the guide's dual-use
// image (8-row x 32-column subtiles, the chunks of a subtile row XOR-swizzled by the row), conflict-free for both kinds
// of read and for the staging writes (SQ_LDS_BANK_CONFLICT = 0 measured).  The tile loop is unrolled by the two buffers so that every LDS address is a
// loop-invariant register plus an immediate, and the softmax keeps a deferred reference maximum (guide T13): in the
// steady state a tile costs a wave 32 MFMAs, 48 LDS reads and ~150 vector instructions (32 v_exp, 32 v_fma, 32 v_add,
// 16 v_max3, 16 v_cvt_pk) -- the first version's 220 bought 7 %.
// Measured (profiles/r05_prefill_*): 0.74-0.98 PFLOP/s at 4K-32K tokens, 32 query heads (1.7-2.6x torch's fused
// attention on contiguous K/V, 3.0-4.3x its masked path on a chunk, 20x with GQA); MFMA pipe busy 48 % of the cycles at
// the ~1.6 GHz the chip holds under this load, LDS port active 38 %, no bank conflicts.  head_dim 64: 0.56-0.82, head_dim
// 256: 0.58-0.73 PFLOP/s (1.2-2.5x torch's).
#include "quest_common.cuh"

namespace quest {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

constexpr uint32_t kPfRows = 128;  // query rows per workgroup
constexpr int kPfKeys = 64;        // keys per tile

struct PrefillParams {
    const half_t* q;
    half_t* o;
    const half_t* kv;
    const int32_t* table;
    uint32_t n_q, kv_len, num_qo_heads, group, page_size, q_blocks, causal;
    PoolStrides st;
    float scale_log2;  // log2(e) / sqrt(D)
};

// Byte offset of 16-byte chunk `ch` of row `row` in a [rows][D halves] tile image: the guide's dual-use image (a) (T10) --
// 8-row x 32-column subtiles of 512 B, the four chunks of a subtile row XOR-swizzled by the row -- which serves the K row
// reads (ds_read_b128) and the V transposed reads (ds_read_b64_tr_b16) of the 32x32x16 operands without bank conflicts and
// with two address registers each for any D (every other term is an immediate).
template <int D>
__device__ __forceinline__ constexpr uint32_t img_off(uint32_t row, uint32_t ch) {
    return 16u * D * (row >> 3) + 512u * (ch >> 2) + 64u * (row & 7u) + 16u * ((ch & 3u) ^ ((row >> 2) & 3u));
}

#define QUEST_LDS __attribute__((address_space(3)))

__device__ __forceinline__ half8 lds_read16(uint32_t addr) {
    return *reinterpret_cast<const QUEST_LDS half8*>((uintptr_t)addr);
}
__device__ __forceinline__ void lds_write16(uint32_t addr, half8 v) {
    *reinterpret_cast<QUEST_LDS half8*>((uintptr_t)addr) = v;
}
// ds_read_b64_tr_b16: the 16 lanes of a group read a 4-row x 16-column block and each receives one COLUMN of it
__device__ __forceinline__ half4 lds_read_tr(uint32_t addr) {
    return __builtin_bit_cast(half4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                         reinterpret_cast<QUEST_LDS short4v*>((uintptr_t)addr)));
}

constexpr float kPfDefer = 8.0f;  // log2 units a row maximum may run ahead of the exponent's reference before a rescale

// Two 32-row blocks per wave (256-row workgroups, one wave per SIMD, each K / V fragment feeding two MFMAs) was built and
// measured: 0.65-0.69 against 0.84-0.91 PFLOP/s -- hipcc keeps the second set of accumulators in AGPRs and copies them to
// and from VGPRs around every vector instruction (1250 v_accvgpr moves per two tiles).
//
// D = 64 / 128: two LDS buffers (32 / 64 KiB), two workgroups per CU.  D = 256: the output accumulators alone are 128
// registers, so one wave per SIMD and ONE 64 KiB buffer (a second barrier per tile instead of the second buffer).
template <int D, bool S16>
__global__ __launch_bounds__(256, D <= 128 ? 2 : 1) void prefill_kernel(const PrefillParams p) {
    constexpr uint32_t IMG = kPfKeys * 2 * D;        // bytes of one K (or V) tile image
    constexpr uint32_t BUFB = 2 * IMG;               // one buffer: K image then V image
    constexpr int NBUF = D <= 128 ? 2 : 1;
    constexpr uint32_t CPR = D / 8;                  // 16-byte chunks per row
    constexpr uint32_t RP = 256 / CPR;               // rows staged per pass of the 256 threads
    constexpr int NP = kPfKeys / RP;                 // passes per tile
    __shared__ __attribute__((aligned(16))) unsigned char s_img[NBUF * BUFB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;

    // workgroup -> (head, query block): the heads of an XCD are neighbours (a GQA group shares its K/V through one L2),
    // and the query blocks with the most keys under the causal mask are dispatched first
    uint32_t head, qblk;
    {
        const uint32_t id = blockIdx.x;
        if ((p.num_qo_heads & 7u) == 0) {
            const uint32_t hpx = p.num_qo_heads >> 3, slot = id >> 3;
            head = (id & 7u) * hpx + slot % hpx;
            qblk = slot / hpx;
        } else {
            head = id % p.num_qo_heads;
            qblk = id / p.num_qo_heads;
        }
        qblk = p.q_blocks - 1 - qblk;
    }
    const uint32_t kv_head = head / p.group;
    const uint32_t q0 = qblk * kPfRows, q0w = q0 + wave * 32;
    const uint32_t last_q = p.n_q - 1, last_key = p.kv_len - 1;
    const uint32_t shift = p.kv_len - p.n_q;  // query i sees keys <= shift + i (causal)
    const uint32_t limit = p.causal ? shift + min(q0w + r, last_q) : last_key;        // this lane's query
    const uint32_t limit_lo = p.causal ? shift + min(q0w, last_q) : last_key;         // first query of the wave (uniform)
    const uint32_t limit_hi = p.causal ? shift + min(q0w + 31, last_q) : last_key;    // last query of the wave
    const uint32_t limit_wg = p.causal ? shift + min(q0 + kPfRows - 1, last_q) : last_key;
    const uint32_t n_tiles = limit_wg / kPfKeys + 1;
    const bool wave_live = q0w < p.n_q;

    // Q^T fragments (B operand): lane (r, h) holds q[row r][16 s + 8 h ..] of k-step s.  Unscaled: the reference folds
    // log2(e) / sqrt(D) into q in fp16 (prefill.cuh:758), which rounds every score by up to 2^-11 of its terms -- 4e-3 of
    // an output once |score| reaches ~100; here the factor multiplies the fp32 score on its way into the exponent.
    // Rows past the end of q repeat the last row (computed, never stored).
    half8 qf[D / 16];
    {
        const half_t* qrow = p.q + ((size_t)min(q0w + r, last_q) * p.num_qo_heads + head) * D + 8 * h;
#pragma unroll
        for (int s = 0; s < D / 16; ++s) qf[s] = ld8(qrow + 16 * s);
    }

    // LDS addresses, loop-invariant up to immediates (buffer, key block, k-step pair, 16-key step, d-block): every read
    // below is `register + constant`, no address arithmetic in the loop
    const uint32_t lds0 = (uint32_t)(uintptr_t)(QUEST_LDS unsigned char*)s_img;
    // K row read of k-step s: row 32 kb + r, chunk 2 s + h = subtile s >> 1, chunk 2 (s & 1) + h of it
    const uint32_t ka[2] = {lds0 + img_off<D>(r, h), lds0 + img_off<D>(r, 2 + h)};
    // V transposed read of d-block db, half u of 16-key step s: rows 16 s + 8 u + 4 h + tq, 16-lane group tg = the d half,
    // lane 4 tq + tp of the group supplies row tq, columns 4 tp .. 4 tp + 3 of the block
    const uint32_t tg = (lane >> 4) & 1u, tq = (lane & 15u) >> 2, tp = lane & 3u;
    const uint32_t va[2] = {lds0 + IMG + img_off<D>(4 * h + tq, 2 * tg + (tp >> 1)) + 8 * (tp & 1u),
                            lds0 + IMG + img_off<D>(8 + 4 * h + tq, 2 * tg + (tp >> 1)) + 8 * (tp & 1u)};

    // staging: thread -> chunk sch of rows srow + RP i of a tile.  Four lanes take the four chunks of a subtile row, the
    // next lanes the next ROW of the same subtile: a wave writes whole 512-byte subtiles (no bank conflict on the
    // ds_write_b128) and still reads 128 contiguous bytes of each of its 8 rows from memory.
    constexpr uint32_t SUB = CPR / 4;  // subtiles per 8-row group
    const uint32_t srow = 8u * (tid / (32u * SUB)) + ((tid >> 2) & 7u), sch = 4u * ((tid >> 5) % SUB) + (tid & 3u);
    uint32_t wa[NP];  // (for RP = 16 / 32 the passes differ by a constant)
#pragma unroll
    for (int i = 0; i < NP; ++i) wa[i] = lds0 + img_off<D>(srow + RP * i, sch);
    // address of (page, entry, kv head) = pool + page * st.page + entry * st.entry + pool_slot(head, entry) * st.head: on the
    // row-rotated pool (QUEST_LAYOUT_NHD_ROT) the head's slot depends on the entry, so the head term is per lane
    const unsigned char* kv_base = reinterpret_cast<const unsigned char*>(p.kv);
    const uint32_t v_bytes = pool_v_off(p.st, kv_head) * 2u;
    auto issue = [&](uint32_t t, half8(&kr)[NP], half8(&vr)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const uint32_t row = srow + RP * i;
            if constexpr (S16) {
                // one page per 16-row group -- the same one for every lane of a wave --, its id a scalar load; rows past the
                // end of the sequence read stale slots of the last page (inside the pool): their scores are masked, their
                // V rows zeroed below
                const uint32_t pi = t * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(row >> 4));
                const uint32_t pg = (uint32_t)ld_uniform_i32(p.table + min(pi, last_key >> 4));
                const unsigned char* src = kv_base + (size_t)pg * p.st.page * 2u;
                const uint32_t lane_bytes = ((row & 15u) * p.st.entry + pool_slot(p.st, kv_head, row & 15u) * p.st.head + sch * 8u) * 2u;
                kr[i] = *reinterpret_cast<const half8*>(src + lane_bytes);
                vr[i] = *reinterpret_cast<const half8*>(src + v_bytes + lane_bytes);
            } else {
                const uint32_t key = min(t * kPfKeys + row, last_key);
                const uint32_t pi = key / p.page_size;
                const uint32_t pg = (uint32_t)p.table[pi], slot = key - pi * p.page_size;
                const unsigned char* src = kv_base + ((size_t)pg * p.st.page + (size_t)slot * p.st.entry +
                                                      (size_t)pool_slot(p.st, kv_head, slot) * p.st.head + sch * 8u) * 2u;
                kr[i] = *reinterpret_cast<const half8*>(src);
                vr[i] = *reinterpret_cast<const half8*>(src + v_bytes);
            }
        }
        if constexpr (S16) {
            if (t * kPfKeys + kPfKeys - 1 > last_key) {  // 0 x stale bits must stay 0
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    if (t * kPfKeys + srow + RP * i > last_key) vr[i] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    };
    auto stash = [&](uint32_t base, const half8(&kr)[NP], const half8(&vr)[NP]) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            lds_write16(wa[i] + base, kr[i]);
            lds_write16(wa[i] + base + IMG, vr[i]);
        }
    };

    f32x16 oacc[D / 32];
#pragma unroll
    for (int db = 0; db < D / 32; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[db][e] = 0.f;
    // online softmax with a deferred maximum (guide T13): m_ref, the reference the exponents are taken against, follows a
    // row's maximum only when that runs ahead by more than kPfDefer (so p <= 2^8: exact range for fp16 and fp32 sums), and
    // then everything still at the old reference (O and l) is moved exactly once, before the tile's exponents are taken
    float m_ref = 0.f, l = 0.f;
    const float c = p.scale_log2, defer = kPfDefer / c;  // scores are in raw units, the exponent in log2 units

    half8 kr[NP], vr[NP];

    auto tile = [&](auto buf_c, uint32_t t) {
        constexpr uint32_t base = decltype(buf_c)::value * BUFB;
        const uint32_t key0 = t * kPfKeys;
        if (wave_live && key0 <= limit_hi) {
            f32x16 sacc[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc[kb][e] = 0.f;
#pragma unroll
                for (int s = 0; s < D / 16; ++s) {
                    const half8 kf = lds_read16(ka[s & 1] + base + kb * (64 * D) + 512 * (s >> 1));
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
            float g = sacc[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) g = __builtin_fmaxf(g, sacc[kb][e]);
            g = __builtin_fmaxf(g, lane_xor<32>(g, lane));  // the row's maximum in this tile
            // key 0 is visible to every query, so tile 0 gives every row a finite reference (NaN in, NaN out)
            if (t == 0 || __builtin_amdgcn_ballot_w64(g - m_ref > defer)) {
                const float m_new = t == 0 ? g : __builtin_fmaxf(m_ref, g);
                // tile 0 has nothing to move (l and O are still zero): alpha = 1 -- exp2((0 - g) c) overflows to +inf for a row
                // whose first-tile maximum is below about -128 log2 units, and 0 x inf would turn the whole row into NaN
                const float alpha = t == 0 ? 1.f : __builtin_amdgcn_exp2f((m_ref - m_new) * c);
                m_ref = m_new;
                l *= alpha;
#pragma unroll
                for (int db = 0; db < D / 32; ++db)
#pragma unroll
                    for (int e = 0; e < 16; ++e) oacc[db][e] *= alpha;
            }
            const float mc = m_ref * c;
            float psum = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kb][e], c, -mc));
                    sacc[kb][e] = pe;
                    psum += pe;
                }
            l += psum;
            // O^T += V^T P^T over the tile's four 16-key steps
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                half8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (half_t)sacc[s >> 1][8 * (s & 1) + j];
#pragma unroll
                for (int db = 0; db < D / 32; ++db) {
                    const half4 lo = lds_read_tr(va[0] + base + s * (32 * D) + 512 * db);
                    const half4 hi = lds_read_tr(va[1] + base + s * (32 * D) + 512 * db);
                    const half8 vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[db], 0, 0, 0);
                }
            }
        }
        if constexpr (NBUF == 2) {
            // tile t + 1 into the other buffer: its last readers passed the previous barrier
            if (t + 1 < n_tiles) stash(base ^ BUFB, kr, vr);
            __syncthreads();
        } else {
            __syncthreads();  // every wave is done reading tile t
            if (t + 1 < n_tiles) stash(0, kr, vr);
            __syncthreads();
        }
        if (t + 2 < n_tiles) issue(t + 2, kr, vr);
    };

    issue(0, kr, vr);
    stash(0, kr, vr);
    __syncthreads();
    if (n_tiles > 1) issue(1, kr, vr);
    if constexpr (NBUF == 2) {
        for (uint32_t t = 0; t < n_tiles; t += 2) {
            tile(std::integral_constant<uint32_t, 0>{}, t);
            if (t + 1 < n_tiles) tile(std::integral_constant<uint32_t, 1>{}, t + 1);
        }
    } else {
        for (uint32_t t = 0; t < n_tiles; ++t) tile(std::integral_constant<uint32_t, 0>{}, t);
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

template <int D>
static void launch_prefill(const PrefillParams& p, uint32_t grid, bool s16, hipStream_t s) {
    if (s16)
        hipLaunchKernelGGL((prefill_kernel<D, true>), dim3(grid), dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((prefill_kernel<D, false>), dim3(grid), dim3(256), 0, s, p);
}

}  // namespace quest

using namespace quest;

extern "C" int quest_prefill_with_paged_kv_cache(const void* q, void* o, uint32_t n_q, uint32_t num_qo_heads,
                                                 quest_paged_kv_t kv, uint32_t n_pages_host, int causal,
                                                 quest_stream_t stream) {
    if (!q || !o || !kv.data || !kv.indices) return QUEST_EINVAL;
    if (n_q == 0 || num_qo_heads == 0 || kv.num_heads == 0 || kv.page_size == 0 || n_pages_host == 0) return QUEST_EINVAL;
    if (kv.layout > QUEST_LAYOUT_NHD_ROT) return QUEST_EINVAL;
    if (kv.last_page_len == 0 || kv.last_page_len > kv.page_size) return QUEST_EINVAL;
    if (num_qo_heads % kv.num_heads != 0) return QUEST_EINVAL;
    if (kv.head_dim != 64 && kv.head_dim != 128 && kv.head_dim != 256) return QUEST_EUNSUPPORTED;
    const uint64_t kv_len = (uint64_t)(n_pages_host - 1) * kv.page_size + kv.last_page_len;
    // causal rows are the LAST n_q tokens of the sequence: the reference assumes kv_len >= qo_len (test_prefill_attention.py:50)
    if (kv_len > 0x7fffffffull || (causal && n_q > kv_len)) return QUEST_EINVAL;
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
    p.causal = causal ? 1u : 0u;
    p.st = pool_strides(kv);
    p.scale_log2 = 1.4426950408889634f / sqrtf((float)kv.head_dim);
    p.q_blocks = (n_q + kPfRows - 1) / kPfRows;
    const uint64_t grid = (uint64_t)p.q_blocks * num_qo_heads;
    if (grid > 0x7fffffffull) return QUEST_ETOOLARGE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (kv.head_dim) {
        case 64: launch_prefill<64>(p, (uint32_t)grid, kv.page_size == 16, s); break;
        case 128: launch_prefill<128>(p, (uint32_t)grid, kv.page_size == 16, s); break;
        default: launch_prefill<256>(p, (uint32_t)grid, kv.page_size == 16, s); break;
    }
    QUEST_LAUNCH_CHECK();
    return 0;
}
