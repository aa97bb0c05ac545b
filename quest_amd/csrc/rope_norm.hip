// RoPE in place on q/k and RMSNorm -- the two remaining `_kernels` ops the decode layer calls
// around the sparse path (SURVEY.md 8f "next" rows, built so the whole operator surface is native).
//
// Reference behaviour restated (not translated):
//   rope     QKApplyRotaryInPlaceKernel   kernels/include/decode/decode_page.cuh:644-692
//            (rotate-half, freq_i = (1/scale) * theta^(-2*(i mod D/2)/D), position = past_len + row)
//   rmsnorm  rmsnorm_twoPassAlgo_e8       quest/ops/csrc/rms_norm.cu:82-158
// Both are elementwise / one-row reductions: HBM-bound, 16 B per lane accesses.
#include "quest_common.cuh"

namespace quest {

template <int D>
__global__ __launch_bounds__(256) void rope_kernel(half_t* __restrict__ q, half_t* __restrict__ k, uint32_t n,
                                                   uint32_t past_len, uint32_t hq, uint32_t hk, float rcp_scale,
                                                   float log2_rcp_theta, const quest_step_state_t* state) {
    // state-driven: row i is the decode token of sequence i (one row for a single sequence)
    constexpr int LPR = D / kVec;
    constexpr int RPB = 256 / LPR;  // vectors per block
    const uint32_t vec = blockIdx.x * RPB + threadIdx.x / LPR;
    const int col = threadIdx.x % LPR;
    const uint32_t per_tok = hq + hk;
    if (vec >= n * per_tok) return;
    const uint32_t tok = vec / per_tok, h = vec % per_tok;
    half_t* x = (h < hq) ? q + ((size_t)tok * hq + h) * D : k + ((size_t)tok * hk + (h - hq)) * D;
    x += col * kVec;
    const float8 self = to_f32(ld8(x));
    const float pos = state ? (float)(state[tok].seq_len - 1) : (float)(past_len + tok);
    float8 out;
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
        const int d = col * kVec + i;
        // partner element d +- D/2 lives in lane col +- LPR/2, same i
        const float other = __shfl_xor(self[i], LPR / 2, kWave);
        const float freq = rcp_scale * exp2f(log2_rcp_theta * (float)(2 * (d % (D / 2))) / (float)D);
        float s, c;
        sincosf(pos * freq, &s, &c);
        out[i] = self[i] * c + (d < D / 2 ? -other : other) * s;
    }
    st8(x, __builtin_convertvector(out, half8));
}

__global__ __launch_bounds__(1024) void rms_norm_kernel(const half_t* __restrict__ in, const half_t* __restrict__ w,
                                                        half_t* __restrict__ out, uint32_t cols, float eps) {
    __shared__ float s_part[16];
    __shared__ float s_inv;
    const size_t row = blockIdx.x;
    const half_t* x = in + row * cols;
    float ss = 0.f;
    for (uint32_t c = threadIdx.x * kVec; c < cols; c += blockDim.x * kVec) {
        const float8 v = to_f32(ld8(x + c));
#pragma unroll
        for (int i = 0; i < kVec; ++i) ss = __builtin_fmaf(v[i], v[i], ss);
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) ss += __shfl_xor(ss, off, kWave);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (uint32_t i = 0; i < (blockDim.x + 63) / 64; ++i) t += s_part[i];
        s_inv = 1.0f / sqrtf(t / (float)cols + eps);
    }
    __syncthreads();
    const float inv = s_inv;
    for (uint32_t c = threadIdx.x * kVec; c < cols; c += blockDim.x * kVec) {
        const float8 v = to_f32(ld8(x + c)), g = to_f32(ld8(w + c));
        float8 r;
#pragma unroll
        for (int i = 0; i < kVec; ++i) r[i] = v[i] * inv * g[i];
        st8(out + row * cols + c, __builtin_convertvector(r, half8));
    }
}

}  // namespace quest

using namespace quest;

static int rope_entry(void* q, void* k, uint32_t n, uint32_t past_kv_len, uint32_t num_qo_heads, uint32_t num_kv_heads,
                      uint32_t head_dim, float rope_scale, float rope_theta, const quest_step_state_t* state,
                      quest_stream_t stream) {
    if (!q || !k || num_qo_heads == 0 || num_kv_heads == 0) return QUEST_EINVAL;
    if (rope_scale == 0.f || rope_theta <= 0.f) return QUEST_EINVAL;
    if (n == 0) return 0;
    const float rcp_scale = 1.0f / rope_scale;
    const float log2_rcp_theta = -log2f(rope_theta);
    const uint32_t vecs = n * (num_qo_heads + num_kv_heads);
    hipStream_t s = (hipStream_t)stream;
#define QUEST_ROPE_CASE(DD)                                                                                       \
    case DD: {                                                                                                    \
        const uint32_t rpb = 256 / (DD / kVec);                                                                   \
        hipLaunchKernelGGL((rope_kernel<DD>), dim3((vecs + rpb - 1) / rpb), dim3(256), 0, s, (half_t*)q,          \
                           (half_t*)k, n, past_kv_len, num_qo_heads, num_kv_heads, rcp_scale, log2_rcp_theta,    \
                           state);                                                                               \
        break;                                                                                                    \
    }
    switch (head_dim) {
        QUEST_ROPE_CASE(64)
        QUEST_ROPE_CASE(128)
        QUEST_ROPE_CASE(256)
        default: return QUEST_EUNSUPPORTED;
    }
#undef QUEST_ROPE_CASE
    QUEST_LAUNCH_CHECK();
    return 0;
}

extern "C" int quest_apply_rope_in_place(void* q, void* k, uint32_t n, uint32_t past_kv_len, uint32_t num_qo_heads,
                                         uint32_t num_kv_heads, uint32_t head_dim, float rope_scale, float rope_theta,
                                         quest_stream_t stream) {
    return rope_entry(q, k, n, past_kv_len, num_qo_heads, num_kv_heads, head_dim, rope_scale, rope_theta, nullptr, stream);
}

extern "C" int quest_apply_rope_in_place_dyn(void* q, void* k, uint32_t num_qo_heads, uint32_t num_kv_heads,
                                             uint32_t head_dim, float rope_scale, float rope_theta,
                                             const quest_step_state_t* state, quest_stream_t stream) {
    if (!state) return QUEST_EINVAL;
    return rope_entry(q, k, 1, 0, num_qo_heads, num_kv_heads, head_dim, rope_scale, rope_theta, state, stream);
}

extern "C" int quest_apply_rope_in_place_batched(void* q, void* k, uint32_t num_qo_heads, uint32_t num_kv_heads,
                                                 uint32_t head_dim, float rope_scale, float rope_theta,
                                                 const quest_step_state_t* state, quest_batch_t batch,
                                                 quest_stream_t stream) {
    if (!state || batch.n_seqs == 0) return QUEST_EINVAL;
    return rope_entry(q, k, batch.n_seqs, 0, num_qo_heads, num_kv_heads, head_dim, rope_scale, rope_theta, state, stream);
}

extern "C" int quest_rms_norm_forward(const void* input, const void* weight, void* output, uint32_t rows, uint32_t cols,
                                      float epsilon, quest_stream_t stream) {
    if (!input || !weight || !output) return QUEST_EINVAL;
    if (cols == 0 || cols % 8 != 0) return QUEST_EUNSUPPORTED;  // rms_norm.cu:164-166
    if (rows == 0) return 0;
    uint32_t threads = (cols / 8 + 63) / 64 * 64;
    if (threads > 1024) threads = 1024;
    hipLaunchKernelGGL(rms_norm_kernel, dim3(rows), dim3(threads), 0, (hipStream_t)stream, (const half_t*)input,
                       (const half_t*)weight, (half_t*)output, cols, epsilon);
    QUEST_LAUNCH_CHECK();
    return 0;
}
