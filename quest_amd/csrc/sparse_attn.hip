// Sparse paged single-token decode attention over the top-K selected pages + the current page,
// and the handler that plans its split over the chip.
//
// Reference behaviour restated (not translated):
//   kernel   BatchDecodeWithPagedKVCacheKernel        kernels/include/decode/decode_attn.cuh:440-646
//   indexing protective_get_k_ptr_heads              kernels/include/decode/decode_page.cuh:325-351
//   planner  BatchDecodeWithPagedKVCacheWorkEstimation / PartitionPagedKVCacheComputeAuxiliaryInfo
//                                                      decode_attn.cuh:675-893
//   merge    VariableLengthMergeStates (flashinfer cascade.cuh, un-vendored) call :992-1001
//   handler  BatchDecodeHandler                        kernels/include/decode/decode_handler.cuh:39-244
//
// gfx950 design.  The op is a gather of (page, head) K/V tiles -- 16 rows x 256 B each for
// D=128 -- and a batch-1 vector-matrix product: HBM-bound, no MFMA.  Per head the page list
// (n_sel selected pages + the current page as one extra slot, so no pseudo-batch) is cut into
// chunks of `pages_per_chunk`; one 4-wave workgroup per (chunk, head).  A wave owns one page at a
// time: each 16 B-per-lane load instruction fetches 4 token rows (1 KiB), all 8 loads of a page
// (4 K + 4 V) are issued before any is consumed and the next page's loads are issued before the
// current page is reduced (register double buffer, no LDS round trip -- K/V are used once).
// The 16 lanes of a row reduce q.k with an xor butterfly; every row keeps its own online-softmax
// state (m, d, acc[8]/lane), rows merge by shuffles, waves through LDS, chunks by a small second
// kernel that also normalises and casts (VariableLengthMergeStates' job).
#include <new>

#include "quest_common.cuh"

namespace quest {

constexpr int kDecWaves = 4;
constexpr float kNegFloor = -1.0e30f;  // finite "-inf": exp2(floor - floor) stays finite, weights it carries are 0

struct DecodeParams {
    const half_t* q;
    half_t* o;
    float* lse;
    const half_t* kv;
    const int32_t* indices;
    float* ws;  // [Hq][n_chunks][D + 2] fp32 partial (acc[D], m, d)
    PoolStrides st;
    uint32_t idx_stride;
    uint32_t n_sel;
    uint32_t last_page_len;
    int32_t last_page_idx;
    uint32_t page_size;
    uint32_t group;  // qo heads per kv head
    uint32_t pages_per_chunk;
    uint32_t n_chunks;
    float scale_log2;  // 1/sqrt(D) * log2(e)
};

template <int D>
struct RowState {
    float m = kNegFloor, d = 0.f;
    float8 acc = (float8)(0.f);
};

// Fold one group of R token rows (one load instruction's worth) into the row state.
template <int D, int T>
__device__ __forceinline__ void fold_page(RowState<D>& st, const float8& qv, const half8 (&k)[T], const half8 (&v)[T],
                                          int row, uint32_t len) {
    constexpr int LPR = D / kVec, R = kWave / LPR;
    float s[T];
    float m_new = st.m;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const float8 kf = to_f32(k[t]);
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < kVec; ++i) dot = __builtin_fmaf(qv[i], kf[i], dot);
        dot = row_allreduce_sum<LPR>(dot);
        const bool valid = (uint32_t)(t * R + row) < len;
        s[t] = valid ? dot : kNegFloor;
        m_new = __builtin_fmaxf(m_new, s[t]);
    }
    const float scale = __builtin_amdgcn_exp2f(st.m - m_new);
    st.d *= scale;
    st.acc *= scale;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const bool valid = (uint32_t)(t * R + row) < len;
        const float p = valid ? __builtin_amdgcn_exp2f(s[t] - m_new) : 0.f;
        st.d += p;
        const float8 vf = to_f32(v[t]);
#pragma unroll
        for (int i = 0; i < kVec; ++i) st.acc[i] = __builtin_fmaf(p, vf[i], st.acc[i]);
    }
    st.m = m_new;
}

template <int D, int T>
__device__ __forceinline__ void load_page(half8 (&k)[T], half8 (&v)[T], const half_t* pk, uint32_t entry_stride,
                                          uint32_t v_off, int row, uint32_t len) {
    constexpr int R = kWave / (D / kVec);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t tok = t * R + row;
        // rows past `len` (only possible in the sequence's last page) are not fetched
        if (tok < len) {
            const half_t* p = pk + (size_t)tok * entry_stride;
            k[t] = ld8(p);
            v[t] = ld8(p + v_off);
        } else {
            k[t] = (half8)(0);
            v[t] = (half8)(0);
        }
    }
}

// S_T = compile-time page size (16) or 0 for the generic run-time path.
template <int D, int S_T>
__global__ __launch_bounds__(kDecWaves* kWave) void sparse_decode_kernel(DecodeParams p) {
    constexpr int LPR = D / kVec, R = kWave / LPR;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane / LPR, col = lane % LPR;
    const uint32_t chunk = blockIdx.x, hq = blockIdx.y, hk = hq / p.group;
    const uint32_t S = S_T ? (uint32_t)S_T : p.page_size;
    const uint32_t n_slots = p.n_sel + 1;  // selected pages + the current page
    const uint32_t slot_begin = chunk * p.pages_per_chunk;
    const uint32_t slot_end = min(n_slots, slot_begin + p.pages_per_chunk);

    float8 qv = to_f32(ld8(p.q + (size_t)hq * D + col * kVec));
    qv *= p.scale_log2;

    const half_t* base = p.kv + (size_t)hk * p.st.head + col * kVec;
    const int32_t* idx_row = p.indices + (size_t)hq * p.idx_stride;
    RowState<D> st;

    auto page_ptr = [&](uint32_t slot, uint32_t& len) -> const half_t* {
        const bool sel = slot < p.n_sel;
        const int32_t page = sel ? idx_row[slot] : p.last_page_idx;
        len = sel ? S : p.last_page_len;
        return base + (size_t)page * p.st.page;
    };

    if constexpr (S_T > 0) {
        constexpr int T = (S_T + R - 1) / R;
        half8 ka[T], va[T], kb[T], vb[T];
        uint32_t len_a = 0, len_b = 0;
        uint32_t slot = slot_begin + wave;
        if (slot < slot_end) {
            const half_t* pk = page_ptr(slot, len_a);
            load_page<D, T>(ka, va, pk, p.st.entry, p.st.v_off, row, len_a);
        }
        while (slot < slot_end) {
            uint32_t nxt = slot + kDecWaves;
            if (nxt < slot_end) {
                const half_t* pk = page_ptr(nxt, len_b);
                load_page<D, T>(kb, vb, pk, p.st.entry, p.st.v_off, row, len_b);
            }
            fold_page<D, T>(st, qv, ka, va, row, len_a);
            slot = nxt;
            if (slot >= slot_end) break;
            nxt = slot + kDecWaves;
            if (nxt < slot_end) {
                const half_t* pk = page_ptr(nxt, len_a);
                load_page<D, T>(ka, va, pk, p.st.entry, p.st.v_off, row, len_a);
            }
            fold_page<D, T>(st, qv, kb, vb, row, len_b);
            slot = nxt;
        }
    } else {
        for (uint32_t slot = slot_begin + wave; slot < slot_end; slot += kDecWaves) {
            uint32_t len;
            const half_t* pk = page_ptr(slot, len);
            for (uint32_t t0 = 0; t0 < len; t0 += R) {
                half8 k1[1], v1[1];
                const uint32_t rem = len - t0;
                load_page<D, 1>(k1, v1, pk + (size_t)t0 * p.st.entry, p.st.entry, p.st.v_off, row, rem);
                fold_page<D, 1>(st, qv, k1, v1, row, rem);
            }
        }
    }

    // rows of the wave -> one state (xor butterfly across rows; both partners get the same bits)
#pragma unroll
    for (int off = LPR; off < kWave; off <<= 1) {
        const float m_o = __shfl_xor(st.m, off, kWave), d_o = __shfl_xor(st.d, off, kWave);
        const float m_n = __builtin_fmaxf(st.m, m_o);
        const float a = __builtin_amdgcn_exp2f(st.m - m_n), b = __builtin_amdgcn_exp2f(m_o - m_n);
        st.d = st.d * a + d_o * b;
#pragma unroll
        for (int i = 0; i < kVec; ++i) st.acc[i] = st.acc[i] * a + __shfl_xor(st.acc[i], off, kWave) * b;
        st.m = m_n;
    }

    // waves -> workgroup through LDS
    __shared__ float s_acc[kDecWaves][D];
    __shared__ float s_md[kDecWaves][2];
    if (row == 0) {
#pragma unroll
        for (int i = 0; i < kVec; ++i) s_acc[wave][col * kVec + i] = st.acc[i];
        if (col == 0) {
            s_md[wave][0] = st.m;
            s_md[wave][1] = st.d;
        }
    }
    __syncthreads();
    const int f = threadIdx.x;
    if (f < D) {
        float M = s_md[0][0];
#pragma unroll
        for (int w = 1; w < kDecWaves; ++w) M = __builtin_fmaxf(M, s_md[w][0]);
        float acc = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < kDecWaves; ++w) {
            const float e = __builtin_amdgcn_exp2f(s_md[w][0] - M);
            acc += e * s_acc[w][f];
            den += e * s_md[w][1];
        }
        if (p.n_chunks == 1) {
            p.o[(size_t)hq * D + f] = (half_t)(acc / den);
            if (p.lse && f == 0) p.lse[hq] = (M + __builtin_amdgcn_logf(den)) * 0.6931471805599453f;
        } else {
            float* w = p.ws + ((size_t)hq * p.n_chunks + chunk) * (D + 2);
            w[f] = acc;
            if (f == 0) {
                w[D] = M;
                w[D + 1] = den;
            }
        }
    }
}

// Merge the per-chunk partial states of a head, normalise, cast to fp16.
template <int D>
__global__ __launch_bounds__(D) void merge_states_kernel(const float* __restrict__ ws, half_t* __restrict__ o,
                                                         float* __restrict__ lse, uint32_t n_chunks) {
    const uint32_t hq = blockIdx.x, f = threadIdx.x;
    const float* w = ws + (size_t)hq * n_chunks * (D + 2);
    float M = kNegFloor;
    for (uint32_t c = 0; c < n_chunks; ++c) M = __builtin_fmaxf(M, w[(size_t)c * (D + 2) + D]);
    float acc = 0.f, den = 0.f;
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const float* wc = w + (size_t)c * (D + 2);
        const float e = __builtin_amdgcn_exp2f(wc[D] - M);
        acc += e * wc[f];
        den += e * wc[D + 1];
    }
    o[(size_t)hq * D + f] = (half_t)(acc / den);
    if (lse && f == 0) lse[hq] = (M + __builtin_amdgcn_logf(den)) * 0.6931471805599453f;
}

}  // namespace quest

using namespace quest;

struct quest_decode_handler {
    uint32_t layout = 0;
    bool started = false;
    uint32_t n_sel = 0, num_qo_heads = 0, num_kv_heads = 0, head_dim = 0, page_size = 0;
    uint32_t pages_per_chunk = 0, n_chunks = 0;
    uint32_t forced_ppc = 0;
    float* ws = nullptr;
    size_t ws_bytes = 0;
};

// Workgroups the planner aims for: 4 per CU x 256 CUs keeps ~128 KiB of loads in flight per CU.
static constexpr uint32_t kTargetWorkgroups = 1024;

extern "C" int quest_decode_handler_create(quest_decode_handler_t** out, uint32_t layout) {
    if (!out || layout > QUEST_LAYOUT_HND) return QUEST_EINVAL;
    quest_decode_handler* h = new (std::nothrow) quest_decode_handler();
    if (!h) return (int)hipErrorOutOfMemory;
    h->layout = layout;
    *out = h;
    return 0;
}

extern "C" void quest_decode_handler_destroy(quest_decode_handler_t* h) {
    if (!h) return;
    if (h->ws) (void)hipFree(h->ws);
    delete h;
}

extern "C" int quest_decode_set_pages_per_chunk(quest_decode_handler_t* h, uint32_t ppc) {
    if (!h) return QUEST_EINVAL;
    h->forced_ppc = ppc;
    return 0;
}

extern "C" int quest_decode_begin_forward(quest_decode_handler_t* h, uint32_t n_selected_pages, uint32_t num_qo_heads,
                                          uint32_t num_kv_heads, uint32_t head_dim, uint32_t page_size,
                                          quest_stream_t stream) {
    (void)stream;
    if (!h || num_qo_heads == 0 || num_kv_heads == 0 || page_size == 0) return QUEST_EINVAL;
    if (num_qo_heads % num_kv_heads != 0) return QUEST_EINVAL;  // decode_attn.cuh:1045-1050
    if (head_dim != 64 && head_dim != 128 && head_dim != 256) return QUEST_EUNSUPPORTED;
    h->n_sel = n_selected_pages;
    h->num_qo_heads = num_qo_heads;
    h->num_kv_heads = num_kv_heads;
    h->head_dim = head_dim;
    h->page_size = page_size;
    const uint32_t n_slots = n_selected_pages + 1;
    uint32_t ppc;
    if (h->forced_ppc) {
        ppc = h->forced_ppc;
    } else {
        uint32_t chunks = kTargetWorkgroups / num_qo_heads;
        if (chunks < 1) chunks = 1;
        if (chunks > n_slots) chunks = n_slots;
        ppc = (n_slots + chunks - 1) / chunks;
    }
    h->pages_per_chunk = ppc;
    h->n_chunks = (n_slots + ppc - 1) / ppc;
    const size_t need = (size_t)num_qo_heads * h->n_chunks * (head_dim + 2) * sizeof(float);
    if (h->n_chunks > 1 && need > h->ws_bytes) {  // grow-only; reused across begin/end cycles
        if (h->ws) (void)hipFree(h->ws);
        h->ws = nullptr;
        h->ws_bytes = 0;
        hipError_t e = hipMalloc((void**)&h->ws, need);
        if (e != hipSuccess) return (int)e;
        h->ws_bytes = need;
    }
    h->started = true;
    return 0;
}

extern "C" int quest_decode_end_forward(quest_decode_handler_t* h) {
    if (!h) return QUEST_EINVAL;
    h->started = false;  // workspace is kept for the next begin_forward (freed in destroy)
    return 0;
}

extern "C" int quest_decode_plan_info(const quest_decode_handler_t* h, uint32_t* pages_per_chunk,
                                      uint32_t* chunks_per_head) {
    if (!h || !h->started) return QUEST_ESTATE;
    if (pages_per_chunk) *pages_per_chunk = h->pages_per_chunk;
    if (chunks_per_head) *chunks_per_head = h->n_chunks;
    return 0;
}

template <int D>
static int launch_decode(const quest_decode_handler* h, const DecodeParams& p, uint32_t num_qo_heads, hipStream_t s) {
    dim3 grid(h->n_chunks, num_qo_heads), block(kDecWaves * kWave);
    if (p.page_size == 16)
        hipLaunchKernelGGL((sparse_decode_kernel<D, 16>), grid, block, 0, s, p);
    else
        hipLaunchKernelGGL((sparse_decode_kernel<D, 0>), grid, block, 0, s, p);
    QUEST_LAUNCH_CHECK();
    if (h->n_chunks > 1) {
        hipLaunchKernelGGL((merge_states_kernel<D>), dim3(num_qo_heads), dim3(D), 0, s, (const float*)p.ws, p.o, p.lse,
                           h->n_chunks);
        QUEST_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int quest_decode_forward(quest_decode_handler_t* h, const void* q, void* o, quest_paged_kv_t kv,
                                    uint32_t num_qo_heads, float* lse, quest_stream_t stream) {
    if (!h) return QUEST_EINVAL;
    if (!h->started) return QUEST_ESTATE;
    if (!q || !o || !kv.data) return QUEST_EINVAL;
    if (h->n_sel > 0 && (!kv.indices || kv.page_budget < h->n_sel)) return QUEST_EINVAL;
    if (kv.layout != h->layout || kv.head_dim != h->head_dim || kv.page_size != h->page_size ||
        kv.num_heads != h->num_kv_heads || num_qo_heads != h->num_qo_heads)
        return QUEST_EINVAL;
    if (kv.last_page_len == 0 || kv.last_page_len > kv.page_size) return QUEST_EINVAL;
    DecodeParams p;
    p.q = (const half_t*)q;
    p.o = (half_t*)o;
    p.lse = lse;
    p.kv = (const half_t*)kv.data;
    p.indices = kv.indices;
    p.ws = h->ws;
    p.st = pool_strides(kv);
    p.idx_stride = kv.page_budget;
    p.n_sel = h->n_sel;
    p.last_page_len = kv.last_page_len;
    p.last_page_idx = kv.last_page_idx;
    p.page_size = kv.page_size;
    p.group = num_qo_heads / kv.num_heads;
    p.pages_per_chunk = h->pages_per_chunk;
    p.n_chunks = h->n_chunks;
    p.scale_log2 = (float)(1.4426950408889634 / sqrt((double)kv.head_dim));
    hipStream_t s = (hipStream_t)stream;
    switch (kv.head_dim) {
        case 64: return launch_decode<64>(h, p, num_qo_heads, s);
        case 128: return launch_decode<128>(h, p, num_qo_heads, s);
        case 256: return launch_decode<256>(h, p, num_qo_heads, s);
        default: return QUEST_EUNSUPPORTED;
    }
}

extern "C" const char* quest_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case QUEST_EINVAL: return "quest: invalid argument";
        case QUEST_EUNSUPPORTED: return "quest: unsupported head_dim/page_size/group size";
        case QUEST_ESTATE: return "quest: begin_forward() must be called before forward()";
        case QUEST_ETOOLARGE: return "quest: top-k row exceeds QUEST_TOPK_MAX_ROW";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "quest: unknown error";
    }
}

extern "C" const char* quest_build_info(void) { return "quest_hip gfx950 r1"; }
