// Page-criticality estimate: o[h][p] = fp16( sum_d max(q[h][d]*Kmax[p][h][d], q[h][d]*Kmin[p][h][d]) )
// for every KV page p except the current (last) one.
//
// Reference behaviour restated (not translated): MaxPossibleSampleWithPagedKVCacheKernel,
// kernels/include/decode/decode_attn.cuh:245-401, arithmetic compute_max_possible :137-168.
// The reference launches one block per kv head (grid (1, H), :1131) which starves a 256-CU
// part; here the (entry, head) rows are tiled: one workgroup = 8 entries x 8 heads = 64 rows
// (1024 workgroups at cfg 3), each streaming 32 KiB with all of its loads in flight at once
// (8 x 16 B per lane); LDS holds the tile's query vectors and the score transpose.
//
// Semantics: the reference's per-feature max of the two fp32 products (decode_attn.cuh:152-156), for ANY
// caller-supplied metadata -- the K slot need not be >= the V slot (the reference's own gtest fills both with
// N(0,1), test_max_possible.cu:50-51), entries may be +-inf or NaN.  See the kernel comment for how.
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
#ifndef QUEST_EST_WAVES
#define QUEST_EST_WAVES 4
#endif
constexpr int kEstIter = QUEST_EST_ITER;    // load instructions per tensor per wave, all in flight together
constexpr int kEstWaves = QUEST_EST_WAVES;  // waves per workgroup
#ifndef QUEST_EST_ITER_GQA
#define QUEST_EST_ITER_GQA 4
#endif
#ifndef QUEST_EST_MIN_WAVES
#define QUEST_EST_MIN_WAVES 1
#endif
template <int G>
constexpr int est_iter() { return G >= 2 ? QUEST_EST_ITER_GQA : kEstIter; }

struct AppendTail {  // optional decode-append riding in the same launch (blocks >= est_blocks)
    quest_paged_kv_t kv;
    const uint16_t* key;
    const uint16_t* value;
    uint32_t est_blocks;
    uint32_t enabled;
    const quest_step_state_t* state;  // optional device-resident lengths / last-page ids (graph replay)
    uint32_t o_stride;                // row stride of o (== n_out unless state-driven)
    uint32_t tile_heads;              // kv heads per workgroup tile (power of two dividing num_heads, <= 8)
    uint32_t meta_table_stride;       // batched launches (blockIdx.y = sequence): entries between page tables
};

// Workgroup tile = EW entries x HW kv heads = 64 rows (D = 128): HW = tile_heads (8 for 8 or 32 kv heads), so
// a tile reads HW*256 B contiguous per entry (NHD) or EW*256 B contiguous per head (HND) for max and for
// min, and produces EW consecutive scores for each of its HW*G query heads.  Row order inside the tile
// follows memory: NHD head-fastest, HND entry-fastest.
//
//   1. the tile's query vectors are requested first and parked in LDS together with a 16-bit sign mask per
//      feature (0xffff where q < 0).  For a finite non-zero q, x -> q*x is strictly monotone on the extended
//      reals and exact in fp32 (fp16 x fp16 has 22 significant bits, |q*x| <= 65504^2), so
//      max(q*Kmax, q*Kmin) == q * (q > 0 ? max(Kmax,Kmin) : min(Kmax,Kmin)) bit for bit, NaN entries included
//      (maxNum/minNum drop a NaN operand exactly as fmaxf drops a NaN product).  hi/lo are two packed fp16
//      instructions per feature PAIR shared by the G query heads of the row, the select is one v_bfi_b32 per
//      pair and the product-accumulate one v_fma_mix_f32 (both operands read as fp16; the product is exact, so
//      the FMA rounds like `acc += q*x`).  A tile whose query vectors contain a zero or a non-finite element
//      (0*inf = NaN breaks the monotonicity argument) takes the literal form instead: two products, fmaxf, add
//      -- the workgroup-uniform flag is found while staging q;
//   2. page-table entries, then all 2*kEstIter metadata loads of the wave (streaming, nt) are issued;
//   3. scores are transposed through LDS and leave as EW-long contiguous runs per query head (the 2-byte
//      scattered stores of the first version cost 0.7 us at MHA and 1.5 us at GQA-4 in write amplification).
template <int D, int G, bool HND>
__global__ __launch_bounds__(kEstWaves* kWave, QUEST_EST_MIN_WAVES) void estimate_kernel(const half_t* __restrict__ q, half_t* __restrict__ o,
                                                                    quest_paged_kv_t meta, uint32_t n_out,
                                                                    AppendTail tail) {
    constexpr int LPR = D / kVec;   // lanes per row
    constexpr int R = kWave / LPR;  // rows per load instruction
    constexpr int kEstIter = est_iter<G>();  // (shadows the namespace constant: GQA instantiations may differ)
    constexpr int ROWS = kEstWaves * kEstIter * R;
    // n_out as passed bounds every address (state-driven launches pass the largest n_out the graph will
    // see; page tables and pools cover it); the live n_out comes from the state further down.
    const uint32_t n_cap = n_out;
    if (tail.state) {  // state-driven launches may be batched: blockIdx.y = sequence (0 for a single one)
        const uint32_t seq = blockIdx.y;
        tail.state += seq;
        q += (size_t)seq * meta.num_heads * G * D;
        o += (size_t)seq * meta.num_heads * G * tail.o_stride;
        meta.indices += (size_t)seq * tail.meta_table_stride;
        tail.key += (size_t)seq * meta.num_heads * D;
        tail.value += (size_t)seq * meta.num_heads * D;
    }
    if (tail.enabled && blockIdx.x >= tail.est_blocks) {
        if (tail.state) {
            const quest_step_state_t st = *tail.state;
            meta.last_page_len = (uint32_t)st.meta_last_page_len;
            meta.last_page_idx = st.meta_last_page_idx;
            tail.kv.last_page_len = (uint32_t)st.kv_last_page_len;
            tail.kv.last_page_idx = st.kv_last_page_idx;
        }
        // The appended token only touches the CURRENT page's KV entry and metadata entry (index n_out),
        // which the estimate excludes (e < n_out), so the two halves of the launch share no byte.
        append_decode_body(tail.kv, meta, tail.key, tail.value, (blockIdx.x - tail.est_blocks) * (kEstWaves * kWave) + threadIdx.x);
        return;
    }

    extern __shared__ __attribute__((aligned(16))) unsigned char est_smem[];
    const uint32_t HW = tail.tile_heads, EW = ROWS / HW;
    half_t* q_s = reinterpret_cast<half_t*>(est_smem);                              // [HW*G][D]  q
    uint16_t* neg_s = reinterpret_cast<uint16_t*>(q_s + (size_t)HW * G * D);        // [HW*G][D]  q < 0 ? 0xffff : 0
    half_t* out_s = reinterpret_cast<half_t*>(neg_s + (size_t)HW * G * D);          // [HW*G][EW] scores
    __shared__ uint32_t s_literal[kEstWaves];  // per wave: some staged q element is zero or non-finite

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane / LPR, col = lane % LPR;
    const uint32_t Hkv = meta.num_heads, S = meta.page_size;
    const uint32_t head_tiles = Hkv / HW;
    const uint32_t et = blockIdx.x / head_tiles, ht = blockIdx.x % head_tiles;
    const uint32_t e0 = et * EW, h0 = ht * HW;
    const PoolStrides ms = pool_strides(meta);
    const half_t* data = reinterpret_cast<const half_t*>(meta.data);
    const int32_t* idx = meta.indices;  // batch_size == 1: indptr[0] == 0 (estimate.cu:14)

    // (1) query vectors of this tile: HW*G contiguous heads starting at h0*G
    constexpr int QV_PER_THREAD = 2;  // covers HW*G*D/8 <= 512 16-byte vectors
    half8 qreg[QV_PER_THREAD];
    const uint32_t q_vecs = HW * G * LPR;
#pragma unroll
    for (int t = 0; t < QV_PER_THREAD; ++t) {
        const uint32_t vi = threadIdx.x + t * (kEstWaves * kWave);
        qreg[t] = ld8(q + (size_t)h0 * G * D + (size_t)(vi < q_vecs ? vi : 0) * kVec);
    }

    // (2) rows of this wave: page-table entries first, then the metadata itself
    uint32_t el[kEstIter], hl[kEstIter];
    size_t page[kEstIter];
    uint32_t ecl[kEstIter];
#pragma unroll
    for (int j = 0; j < kEstIter; ++j) {
        const uint32_t r = (wave * kEstIter + j) * R + row;  // row inside the tile
        el[j] = HND ? r % EW : r / HW;
        hl[j] = HND ? r / EW : r % HW;
        const uint32_t e = e0 + el[j];
        // Loads are unconditional from a clamped entry: a predicated load compiles to branch + load + wait
        // and serialises the round trips.  Clamped rows re-read the last entry (tail tiles only).
        ecl[j] = e < n_cap ? e : n_cap - 1;
    }
    if (S % EW == 0) {
        // the tile's EW consecutive entries (and the clamped ones of a tail tile) lie in ONE metadata page: a
        // single wave-uniform (scalar) table load instead of a vector load per row at the head of the
        // table -> metadata dependency chain
        const size_t pg = (size_t)idx[e0 / S];
#pragma unroll
        for (int j = 0; j < kEstIter; ++j) page[j] = pg;
    } else {
#pragma unroll
        for (int j = 0; j < kEstIter; ++j) page[j] = (size_t)idx[ecl[j] / S];
    }
    if (tail.state) {  // live length (<= n_cap); whole tiles past it have nothing to do
        n_out = (uint32_t)(tail.state->n_pages - 1);
        if (e0 >= n_out) return;
    }
    half8 mx[kEstIter], mn[kEstIter];
#pragma unroll
    for (int j = 0; j < kEstIter; ++j) {
        const half_t* p = data + page[j] * ms.page + (size_t)(h0 + hl[j]) * ms.head + (size_t)(ecl[j] % S) * ms.entry +
                          col * kVec;
        mx[j] = ld8_stream(p);
        mn[j] = ld8_stream(p + ms.v_off);
    }

    // q -> LDS (these loads are older than the metadata loads, so this does not wait for them)
    bool odd_q = false;  // a zero or non-finite query element among the ones this thread stages
#pragma unroll
    for (int t = 0; t < QV_PER_THREAD; ++t) {
        const uint32_t vi = threadIdx.x + t * (kEstWaves * kWave);
        if (vi < q_vecs) {
            const uint4 w = __builtin_bit_cast(uint4, qreg[t]);
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
            uint32_t m[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                m[i] = ((ww[i] >> 15) & 0x00010001u) * 0xffffu;
                // (|x| bits) - 1 >= 0x7bff  <=>  x is +-0, +-inf or NaN
                odd_q |= ((ww[i] & 0x7fffu) - 1u >= 0x7bffu) | (((ww[i] >> 16) & 0x7fffu) - 1u >= 0x7bffu);
            }
            st8(q_s + (size_t)vi * kVec, qreg[t]);
            *reinterpret_cast<uint4*>(neg_s + (size_t)vi * kVec) = make_uint4(m[0], m[1], m[2], m[3]);
        }
    }
    {
        const unsigned long long any = __ballot(odd_q);
        if (lane == 0) s_literal[wave] = any != 0ull;
    }
    __syncthreads();
    uint32_t literal = 0;
#pragma unroll
    for (int w = 0; w < kEstWaves; ++w) literal |= s_literal[w];
    literal = __builtin_amdgcn_readfirstlane(literal);

    // (3) scores
    if (!literal) {
#pragma unroll
        for (int j = 0; j < kEstIter; ++j) {
            const half8 hi = __builtin_elementwise_max(mx[j], mn[j]), lo = __builtin_elementwise_min(mx[j], mn[j]);
            const uint4 hb = __builtin_bit_cast(uint4, hi), lb = __builtin_bit_cast(uint4, lo);
            float accs[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t qh = hl[j] * G + g;  // query head inside the tile
                const float8 qf = to_f32(ld8(q_s + (size_t)qh * D + col * kVec));
                const uint4 nm = *reinterpret_cast<const uint4*>(neg_s + (size_t)qh * D + col * kVec);
                uint4 sb;  // per 16-bit feature: q < 0 ? lo : hi
                sb.x = (lb.x & nm.x) | (hb.x & ~nm.x);
                sb.y = (lb.y & nm.y) | (hb.y & ~nm.y);
                sb.z = (lb.z & nm.z) | (hb.z & ~nm.z);
                sb.w = (lb.w & nm.w) | (hb.w & ~nm.w);
                const float8 x = to_f32(__builtin_bit_cast(half8, sb));
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < kVec; ++i) acc = __builtin_fmaf(qf[i], x[i], acc);
                accs[g] = acc;
            }
            if constexpr (LPR == 16 && G > 1) {
                // the G row sums share one reduction tree walk (same association order as below, hence the same bits)
                int g_mine;
                const float total = row_segmented_sum16<G>(accs, col, g_mine);
                if ((col & (16 / G - 1)) == 0) out_s[(hl[j] * G + g_mine) * EW + el[j]] = (half_t)total;
            } else {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float total = row_allreduce_sum_fast<LPR>(accs[g]);
                    if (col == 0) out_s[(hl[j] * G + g) * EW + el[j]] = (half_t)total;
                }
            }
        }
    } else {  // literal form of decode_attn.cuh:152-156 (zero / non-finite query element in the tile)
#pragma unroll
        for (int j = 0; j < kEstIter; ++j) {
            const float8 a = to_f32(mx[j]), b = to_f32(mn[j]);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const uint32_t qh = hl[j] * G + g;
                const float8 qf = to_f32(ld8(q_s + (size_t)qh * D + col * kVec));
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < kVec; ++i) acc += __builtin_fmaxf(qf[i] * a[i], qf[i] * b[i]);
                acc = row_allreduce_sum_fast<LPR>(acc);
                if (col == 0) out_s[qh * EW + el[j]] = (half_t)acc;
            }
        }
    }
    __syncthreads();
    const uint32_t n_scores = HW * G * EW;
    for (uint32_t t = threadIdx.x; t < n_scores; t += kEstWaves * kWave) {
        const uint32_t qh = t / EW, e = e0 + t % EW;
#ifdef QUEST_EST_NOSTORE
        if (e < n_out && out_s[t] == (half_t)12345.f)
#else
        if (e < n_out)
#endif
            o[((size_t)h0 * G + qh) * tail.o_stride + e] = out_s[t];
    }
}

// kv heads per tile: the largest power of two <= 8 that divides num_heads and whose query vectors
// (hw * G * D/8 sixteen-byte vectors) fit the two-per-thread staging pass
static uint32_t pick_tile_heads(uint32_t num_heads, uint32_t G, uint32_t lpr) {
    for (uint32_t hw = 8; hw > 1; hw >>= 1)
        if (num_heads % hw == 0 && hw * G * lpr <= 2u * kEstWaves * kWave) return hw;
    return 1;
}

template <int D, int G>
static int launch_estimate(const void* q, void* o, uint32_t n_out, const quest_paged_kv_t& meta, AppendTail tail,
                           hipStream_t s, uint32_t n_seqs) {
    constexpr int R = kWave / (D / kVec);
    constexpr uint32_t ROWS = kEstWaves * est_iter<G>() * R;
    const bool hnd = meta.layout == QUEST_LAYOUT_HND;
    if (!tail.state) tail.o_stride = n_out;
    const uint32_t hw = pick_tile_heads(meta.num_heads, G, D / kVec), ew = ROWS / hw;
    tail.tile_heads = hw;
    tail.est_blocks = ((n_out + ew - 1) / ew) * (meta.num_heads / hw);
    uint32_t blocks = tail.est_blocks;
    if (tail.enabled) blocks += (meta.num_heads * (D / kVec) + kEstWaves * kWave - 1) / (kEstWaves * kWave);
    if (blocks == 0) return 0;
    dim3 grid(blocks, n_seqs);
    const size_t lds = (size_t)hw * G * (2 * D + ew) * sizeof(half_t);
    if (hw * G * (D / kVec) > 2 * kEstWaves * kWave) return QUEST_EUNSUPPORTED;  // q staging capacity
    if (hnd)
        hipLaunchKernelGGL((estimate_kernel<D, G, true>), grid, dim3(kEstWaves * kWave), lds, s, (const half_t*)q,
                           (half_t*)o, meta, n_out, tail);
    else
        hipLaunchKernelGGL((estimate_kernel<D, G, false>), grid, dim3(kEstWaves * kWave), lds, s, (const half_t*)q,
                           (half_t*)o, meta, n_out, tail);
    QUEST_LAUNCH_CHECK();
    return 0;
}

template <int D>
static int dispatch_group(uint32_t G, const void* q, void* o, uint32_t n_out, const quest_paged_kv_t& meta,
                          const AppendTail& tail, hipStream_t s, uint32_t n_seqs) {
    switch (G) {
        case 1: return launch_estimate<D, 1>(q, o, n_out, meta, tail, s, n_seqs);
        case 2: return launch_estimate<D, 2>(q, o, n_out, meta, tail, s, n_seqs);
        case 4: return launch_estimate<D, 4>(q, o, n_out, meta, tail, s, n_seqs);
        case 8: return launch_estimate<D, 8>(q, o, n_out, meta, tail, s, n_seqs);
        default: return QUEST_EUNSUPPORTED;
    }
}

}  // namespace quest

using namespace quest;

namespace quest {
int check_pool(const quest_paged_kv_t& p);  // append.hip
}

static int estimate_entry(const void* q, void* o, uint32_t num_qo_heads, uint32_t n_out, const quest_paged_kv_t& metadata,
                          const AppendTail& tail, hipStream_t s, uint32_t n_seqs = 1) {
    if (!q || !metadata.data || !metadata.indices) return QUEST_EINVAL;
    if (metadata.layout > QUEST_LAYOUT_HND || metadata.num_heads == 0 || metadata.page_size == 0) return QUEST_EINVAL;
    if (num_qo_heads == 0 || num_qo_heads % metadata.num_heads != 0) return QUEST_EINVAL;
    if (n_out > 0 && !o) return QUEST_EINVAL;
    if (n_out == 0 && !tail.enabled) return 0;  // nothing to score (single page)
    const uint32_t G = num_qo_heads / metadata.num_heads;
    switch (metadata.head_dim) {
        case 64: return dispatch_group<64>(G, q, o, n_out, metadata, tail, s, n_seqs);
        case 128: return dispatch_group<128>(G, q, o, n_out, metadata, tail, s, n_seqs);
        case 256: return dispatch_group<256>(G, q, o, n_out, metadata, tail, s, n_seqs);
        default: return QUEST_EUNSUPPORTED;
    }
}

extern "C" int quest_estimate_attn_score(const void* q, void* o, uint32_t num_qo_heads, uint32_t n_out,
                                         quest_paged_kv_t metadata, quest_stream_t stream) {
    AppendTail tail{};
    return estimate_entry(q, o, num_qo_heads, n_out, metadata, tail, (hipStream_t)stream);
}

static int append_estimate_state(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                 uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out,
                                 quest_paged_kv_t metadata, const quest_step_state_t* state, quest_batch_t batch,
                                 quest_stream_t stream) {
    if (!k || !v || !state || max_n_out == 0 || o_stride < max_n_out || batch.n_seqs == 0) return QUEST_EINVAL;
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
    tail.meta_table_stride = batch.meta_table_stride;
    return estimate_entry(q, o, num_qo_heads, max_n_out, metadata, tail, (hipStream_t)stream, batch.n_seqs);
}

extern "C" int quest_append_estimate_dyn(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                         uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out,
                                         quest_paged_kv_t metadata, const quest_step_state_t* state,
                                         quest_stream_t stream) {
    const quest_batch_t one = {1, 0, 0, 0};
    return append_estimate_state(k, v, kv, q, o, num_qo_heads, o_stride, max_n_out, metadata, state, one, stream);
}

extern "C" int quest_append_estimate_batched(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                             uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out,
                                             quest_paged_kv_t metadata, const quest_step_state_t* state,
                                             quest_batch_t batch, quest_stream_t stream) {
    // every sequence's metadata table must cover the entries the grid is sized for
    if (batch.n_seqs > 1 && (uint64_t)batch.meta_table_stride * metadata.page_size < max_n_out) return QUEST_EINVAL;
    return append_estimate_state(k, v, kv, q, o, num_qo_heads, o_stride, max_n_out, metadata, state, batch, stream);
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
