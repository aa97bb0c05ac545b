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
#include "estimate_device.cuh"

namespace quest {

// One tile per workgroup (device body: estimate_device.cuh); blocks >= tail.est_blocks run the decode append.
// The leading scalar arguments are the fields the first loads of a workgroup need (role, page-table entry, metadata
// rows, live length): scalar kernel arguments are preloaded into SGPRs at wave launch (14 dwords beside the kernarg
// pointer, build.py), a struct passed by value is not -- its remaining fields arrive by s_load while those loads fly.
template <int D, int G, bool HND, int EWV>
__global__ __launch_bounds__(EWV* kWave, QUEST_EST_MIN_WAVES) void estimate_kernel(
    const half_t* __restrict__ q, void* a_meta_data, const int32_t* a_meta_indices, const quest_step_state_t* a_state,
    uint32_t n_out, uint32_t a_num_heads, uint32_t a_page_size, uint32_t a_tile_log2, uint32_t a_append_from,
    uint32_t a_meta_table_stride, half_t* __restrict__ o, quest_paged_kv_t meta, AppendTail tail) {
    // a_append_from: first append block (= est_blocks), 0xffffffff when no append rides in this launch
    meta.data = a_meta_data, meta.indices = a_meta_indices, meta.num_heads = a_num_heads, meta.page_size = a_page_size;
    // a_tile_log2: log2(tile heads) | row-rotated pool << 8 (the layout decides the first loads' addresses, so it travels in
    // the preloaded block, not in the struct; NHD and NHD_ROT share the HND = false instantiation)
    meta.head_dim = D, meta.layout = HND ? QUEST_LAYOUT_HND : (a_tile_log2 >> 8) ? QUEST_LAYOUT_NHD_ROT : QUEST_LAYOUT_NHD;
    a_tile_log2 &= 255u;
    tail.state = a_state, tail.tile_log2 = a_tile_log2, tail.tile_heads = 1u << a_tile_log2, tail.est_blocks = a_append_from, tail.enabled = 1u;
    tail.meta_table_stride = a_meta_table_stride;
    uint32_t seq = 0;
    if (tail.state) {  // state-driven launches may be batched: blockIdx.y = sequence (0 for a single one)
        seq = blockIdx.y;
        tail.state += seq;
        q += (size_t)seq * meta.num_heads * G * D;
        meta.indices += (size_t)seq * tail.meta_table_stride;
    }
    if (QUEST_UNLIKELY(blockIdx.x >= tail.est_blocks)) {  // the few append blocks at the end of the grid
        tail.key += (size_t)seq * meta.num_heads * D;
        tail.value += (size_t)seq * meta.num_heads * D;
        if (tail.state) {
            const quest_step_state_t st = *tail.state;
            meta.last_page_len = (uint32_t)st.meta_last_page_len;
            meta.last_page_idx = st.meta_last_page_idx;
            tail.kv.last_page_len = (uint32_t)st.kv_last_page_len;
            tail.kv.last_page_idx = st.kv_last_page_idx;
        }
        // The appended token only touches the CURRENT page's KV entry and metadata entry (index n_out),
        // which the estimate excludes (e < n_out), so the two halves of the launch share no byte.
        append_decode_body(tail.kv, meta, tail.key, tail.value, (blockIdx.x - tail.est_blocks) * (EWV * kWave) + threadIdx.x);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char est_smem[];
    __shared__ uint32_t s_literal[EWV];
    const uint32_t head_tiles = meta.num_heads >> tail.tile_log2;
    estimate_tile<D, G, HND, EWV, 1>(q, o, meta, n_out, tail, blockIdx.x / head_tiles, blockIdx.x % head_tiles,
                                     threadIdx.x, est_smem, s_literal, seq * meta.num_heads * G);
}

// Tile shape of a launch with EWV-wave workgroups: kv heads per tile (`hw`) and whether the shape can be served.
template <int D, int G, int EWV>
static bool estimate_tile_shape(const quest_paged_kv_t& meta, const AppendTail& tail, const PoolStrides& strides, uint32_t& hw) {
    constexpr int R = kWave / (D / kVec);
    constexpr uint32_t ROWS = EWV * est_iter<G>() * R;
    hw = pick_tile_heads(meta.num_heads, G, D / kVec, EWV);
    // Tile maxima are one key per run of 8 consecutive columns of ONE query head, taken from 8 consecutive lanes of the
    // score store loop: a tile must be a multiple of 8 entries wide.  Narrow the tile's head range where it is not (head_dim
    // 256, two-wave workgroups), and refuse what still does not fit -- before anything is launched.
    while (tail.tile_off && hw > 1 && (ROWS / hw) % 8u != 0) hw >>= 1;
    if (tail.tile_off && (ROWS / hw) % 8u != 0) return false;
    // row-rotated pool: the heads of a tile's rows (slot ^ (entry & rot)) must be heads of the same tile
    if (strides.rot >= hw) return false;
    return hw * G * (D / kVec) <= 2u * EWV * kWave;  // q staging capacity
}

template <int D, int G, int EWV>
static int launch_estimate_w(const void* q, void* o, uint32_t n_out, const quest_paged_kv_t& meta, AppendTail tail,
                             hipStream_t s, uint32_t n_seqs, uint32_t hw) {
    constexpr int R = kWave / (D / kVec);
    constexpr uint32_t ROWS = EWV * est_iter<G>() * R;
    const bool hnd = meta.layout == QUEST_LAYOUT_HND;
    if (!tail.state && tail.o_stride == 0) tail.o_stride = n_out;  // contiguous rows unless the caller pads them
    const uint32_t ew = ROWS / hw;
    tail.tile_heads = hw;
    tail.tile_log2 = (uint32_t)__builtin_ctz(hw);
    tail.est_blocks = ((n_out + ew - 1) / ew) * (meta.num_heads / hw);
    uint32_t blocks = tail.est_blocks;
    if (tail.enabled) blocks += (meta.num_heads * (D / kVec) + EWV * kWave - 1) / (EWV * kWave);
    if (blocks == 0) return 0;
    dim3 grid(blocks, n_seqs);
    const size_t lds = est_tile_lds_bytes(hw, G, D, ew);
    const uint32_t append_from = tail.enabled ? tail.est_blocks : 0xffffffffu;
    if (hnd)
        hipLaunchKernelGGL((estimate_kernel<D, G, true, EWV>), grid, dim3(EWV * kWave), lds, s, (const half_t*)q, meta.data,
                           meta.indices, tail.state, n_out, meta.num_heads, meta.page_size, tail.tile_log2, append_from,
                           tail.meta_table_stride, (half_t*)o, meta, tail);
    else
        hipLaunchKernelGGL((estimate_kernel<D, G, false, EWV>), grid, dim3(EWV * kWave), lds, s, (const half_t*)q, meta.data,
                           meta.indices, tail.state, n_out, meta.num_heads, meta.page_size,
                           tail.tile_log2 | (meta.layout == QUEST_LAYOUT_NHD_ROT ? 1u << 8 : 0u), append_from,
                           tail.meta_table_stride, (half_t*)o, meta, tail);
    QUEST_LAUNCH_CHECK();
    return 0;
}

// Workgroup width: TWO waves (32-row tiles) where the tile shape allows -- smaller workgroups start and retire in finer
// steps, measured A B A B on one box (profiles/r06_ab_estimate_tile_shapes.txt, r06_ab_estimate_waves.txt): cfg 4 A+E 9.16 ->
// 8.73 us (layer 28.9 -> 28.4), cfg 3 7.97 -> 7.83 (21.82 -> 21.72), 8 x cfg 3 two-launch form 11.20 -> 11.11, cfg 5 no
// change; 8 waves and 8 load rounds per wave are slower -- else the four-wave tile of rounds 1-5 (it stages twice the query
// vectors: GQA shapes whose group does not fit two waves' staging, row-rotated pools whose rotation needs >= 4 heads).
// -DQUEST_EST_WAVES=4 forces the four-wave form (A/B).
template <int D, int G>
static int launch_estimate(const void* q, void* o, uint32_t n_out, const quest_paged_kv_t& meta, const AppendTail& tail,
                           hipStream_t s, uint32_t n_seqs) {
    const PoolStrides strides = pool_strides(meta);
    uint32_t hw = 0;
    if (kEstWaves == 2 && estimate_tile_shape<D, G, 2>(meta, tail, strides, hw))
        return launch_estimate_w<D, G, 2>(q, o, n_out, meta, tail, s, n_seqs, hw);
    if (estimate_tile_shape<D, G, 4>(meta, tail, strides, hw)) return launch_estimate_w<D, G, 4>(q, o, n_out, meta, tail, s, n_seqs, hw);
    return QUEST_EUNSUPPORTED;
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
    if (metadata.layout > QUEST_LAYOUT_NHD_ROT || metadata.num_heads == 0 || metadata.page_size == 0) return QUEST_EINVAL;
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
                                 quest_stream_t stream, uint32_t tile_off = 0) {
    if (!k || !v || !state || max_n_out == 0 || o_stride < max_n_out || batch.n_seqs == 0) return QUEST_EINVAL;
    if (tile_off && (tile_off % 8u != 0 || tile_off < ((max_n_out + 7u) & ~7u) ||
                     o_stride < tile_off + ((((max_n_out + 7u) / 8u) + 3u) & ~3u)))
        return QUEST_EINVAL;  // the maxima live behind the scores of the same row, readable in 8-byte pieces
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
    tail.tile_off = tile_off;
    return estimate_entry(q, o, num_qo_heads, max_n_out, metadata, tail, (hipStream_t)stream, batch.n_seqs);
}

extern "C" int quest_estimate_attn_score_batched(const void* q, void* o, uint32_t num_qo_heads, uint32_t o_stride,
                                                 uint32_t max_n_out, quest_paged_kv_t metadata,
                                                 const quest_step_state_t* state, quest_batch_t batch, quest_stream_t stream) {
    if (!state || max_n_out == 0 || o_stride < max_n_out || batch.n_seqs == 0) return QUEST_EINVAL;
    if (batch.n_seqs > 1 && (uint64_t)batch.meta_table_stride * metadata.page_size < max_n_out) return QUEST_EINVAL;
    metadata.last_page_len = 1;  // placeholder; the kernel reads the live length from `state`
    if (int e = check_pool(metadata)) return e;
    AppendTail tail{};
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

extern "C" int quest_append_estimate_tiles_dyn(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                               uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out,
                                               uint32_t tile_max_offset, quest_paged_kv_t metadata,
                                               const quest_step_state_t* state, quest_stream_t stream) {
    if (tile_max_offset == 0) return QUEST_EINVAL;
    const quest_batch_t one = {1, 0, 0, 0};
    return append_estimate_state(k, v, kv, q, o, num_qo_heads, o_stride, max_n_out, metadata, state, one, stream,
                                 tile_max_offset);
}

extern "C" int quest_append_estimate_batched(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                             uint32_t num_qo_heads, uint32_t o_stride, uint32_t max_n_out,
                                             quest_paged_kv_t metadata, const quest_step_state_t* state,
                                             quest_batch_t batch, quest_stream_t stream) {
    // every sequence's metadata table must cover the entries the grid is sized for
    if (batch.n_seqs > 1 && (uint64_t)batch.meta_table_stride * metadata.page_size < max_n_out) return QUEST_EINVAL;
    return append_estimate_state(k, v, kv, q, o, num_qo_heads, o_stride, max_n_out, metadata, state, batch, stream);
}

extern "C" int quest_append_estimate_strided(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                             uint32_t num_qo_heads, uint32_t n_out, uint32_t o_stride,
                                             quest_paged_kv_t metadata, quest_stream_t stream) {
    if (!k || !v || (o_stride != 0 && o_stride < n_out)) return QUEST_EINVAL;
    if (int e = check_pool(kv)) return e;
    if (int e = check_pool(metadata)) return e;
    if (kv.num_heads != metadata.num_heads || kv.head_dim != metadata.head_dim) return QUEST_EINVAL;
    AppendTail tail{};
    tail.kv = kv;
    tail.key = (const uint16_t*)k;
    tail.value = (const uint16_t*)v;
    tail.enabled = 1;
    tail.o_stride = o_stride;
    return estimate_entry(q, o, num_qo_heads, n_out, metadata, tail, (hipStream_t)stream);
}

extern "C" int quest_append_estimate(const void* k, const void* v, quest_paged_kv_t kv, const void* q, void* o,
                                     uint32_t num_qo_heads, uint32_t n_out, quest_paged_kv_t metadata,
                                     quest_stream_t stream) {
    return quest_append_estimate_strided(k, v, kv, q, o, num_qo_heads, n_out, 0, metadata, stream);
}
