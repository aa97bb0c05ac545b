// Batch-1 decode-token projections around the sparse attention path, fused so that a decoder layer is 4 launches besides
// its attention (EXTENSION, SURVEY.md 8f-4 "drop into the HF pipeline" / VERDICT r2 item 8).
//
// The reference leaves these to cuBLAS + small PyTorch kernels: per layer RMSNorm (quest/ops/csrc/rms_norm.cu:82-213,
// called at quest/models/llama.py:72), q/k/v projections (QuestAttention.py:64-66), RoPE (decode_page.cuh:644-728 via
// QuestAttention.py:70), o_proj (:118), residual adds, gate/up projections, SiLU * up, down projection
// (llama.py LlamaMLP) -- 14 launches of which 7 are fp16 GEMVs (M = 1) that stream the layer's 386 MiB of weights.
// On MI355X a decode token of Llama-2-7B spent 4.5 of its 5.4 ms there (~450 launches, profiles/r02_e2e_*).  Here:
//
//   quest_decode_qkv_rope      h -> RMSNorm -> [Wq; Wk; Wv] x -> RoPE(q, k) -> q, k, v          (1 launch)
//   quest_decode_gemv_residual h += W x            (o_proj on the attention output, down_proj on the MLP activation)
//   quest_decode_mlp_gate_up   h -> RMSNorm -> SiLU(Wg x) * (Wu x)                               (1 launch)
//   quest_decode_norm_gemv     h -> RMSNorm -> W x                                               (final norm + lm_head)
//
// One kernel: 8-wave workgroups, a wave owns 1 / 2 / 4 output row-dots (the host picks the largest count that still gives
// two workgroups per CU), a lane every 64th 16-byte vector of a row, 8 loads of 16 bytes in flight per lane (80-90 VGPRs,
// 128 KiB in flight per CU), fp16 x fp16 -> fp32 by v_dot2_f32_f16; the input vector sits in LDS (normalised there when the
// launch carries a RMSNorm prologue -- every workgroup recomputes the 8 KiB reduction rather than pay a launch for it),
// the weight loads of the first round are issued BEFORE the prologue so its latency hides under them.  HBM-bound
// (weights read exactly once): 4.9-5.7 TB/s at Llama-2-7B shapes; no MFMA: M = 1.
#include <cstdio>
#include <cstdlib>

#include "quest_common.cuh"

namespace quest {

// Workgroup shape: NWV waves, RW row-dots per wave, U iterations (x 64 lanes x 16 B x RW rows) in flight -- template
// parameters picked by the host (launch_gemv): 8-wave workgroups so that the input vector is staged in LDS once per
// 8-32 rows (with 4-wave, 4-row workgroups the 22 KiB vector of down_proj was re-staged 1024 times: +26 % of L2 traffic),
// and the largest RW that still gives >= 2 workgroups per CU.  Round-3 history (profiles/r03_e2e_kernel_stats*.csv):
// 4 x 4 rows everywhere -> o_proj / down_proj at 3.9 TB/s (256 workgroups: one per CU); 1 row per wave there -> 4.5-4.9.
enum GemvMode { kGvPlain = 0, kGvResidual = 1, kGvSiluMul = 2, kGvQkvRope = 3 };

struct GemvArgs {
    const half_t* x;       // [in_dim] input vector
    const half_t* gamma;   // RMSNorm weight [in_dim] or nullptr (no prologue)
    float eps;
    uint32_t in_dim;       // multiple of 8
    const half_t* w[3];    // row-major [rows[i]][in_dim]
    uint32_t rows[3];
    half_t* out[3];
    const half_t* residual;  // kGvResidual: out[0][r] = residual[r] + W[r] . x  (may alias out[0])
    // kGvQkvRope: w = {Wq, Wk, Wv}; rotate-half RoPE on the first two (decode_page.cuh:644-692)
    uint32_t head_dim;
    float rcp_scale, log2_rcp_theta;
    const quest_step_state_t* state;  // position = state->seq_len - 1
};

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dot8(const half8& a, const half8& b, float acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const half2_t x = {a[2 * i], a[2 * i + 1]}, y = {b[2 * i], b[2 * i + 1]};
        acc = __builtin_amdgcn_fdot2(x, y, acc, false);
    }
    return acc;
}

template <int MODE, int kGvWaves, int kGvRows, int kGvUnroll>
__global__ __launch_bounds__(kGvWaves* kWave, kGvWaves / 2) void gemv_kernel(GemvArgs p) {  // 2 workgroups per CU
    static_assert(MODE == kGvPlain || MODE == kGvResidual || kGvRows % 2 == 0, "pairs of row-dots stay in one wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char gv_smem[];
    half_t* x_s = reinterpret_cast<half_t*>(gv_smem);  // in_dim halves, zero-padded to a multiple of 64 * 8 * kGvUnroll
    __shared__ float s_part[kGvWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_vec = p.in_dim / kVec;
    const uint32_t iters = (n_vec + kWave - 1) / kWave;
    const uint32_t outer = (iters + kGvUnroll - 1) / kGvUnroll;

    // ---- rows of this wave
    const uint32_t vr0 = (blockIdx.x * kGvWaves + wave) * kGvRows;  // first virtual row-dot of the wave
    const half_t* wrow[kGvRows];
    uint32_t mat[kGvRows], row[kGvRows];
    bool live[kGvRows];
#pragma unroll
    for (int r = 0; r < kGvRows; ++r) {
        uint32_t vr = vr0 + r, m = 0, rr;
        if constexpr (MODE == kGvSiluMul) {  // (gate, up) of the same output row side by side
            m = vr & 1u;
            rr = vr >> 1;
            live[r] = rr < p.rows[0];
        } else {
            while (m < 2 && vr >= p.rows[m]) vr -= p.rows[m], ++m;
            live[r] = vr < p.rows[m];
            rr = vr;
            if constexpr (MODE == kGvQkvRope) {
                // pair-major order inside a head: virtual rows 2j, 2j+1 are features j and j + head_dim/2, so that a
                // wave's 4 row-dots are two complete rotation pairs
                const uint32_t hd = p.head_dim, head = rr / hd, j = rr % hd;
                rr = head * hd + (j >> 1) + (j & 1u) * (hd / 2);
            }
        }
        mat[r] = m;
        row[r] = live[r] ? rr : 0u;
        wrow[r] = p.w[live[r] ? m : 0] + (size_t)row[r] * p.in_dim;
    }

    // ---- the first weight loads leave before the input vector is staged (they do not depend on it).  Addressing: the
    // row base is wave-uniform (an SGPR pair, advanced per round), the lane's part one 32-bit offset, the slot
    // an immediate -- except in the last round of a row whose length is not a multiple of a round, where the vector index
    // is clamped (kClamp).  A round = kGvUnroll slots x 64 lanes x 16 bytes per row.
    constexpr uint32_t kRoundHalves = kGvUnroll * kWave * kVec;
    const uint32_t full = n_vec / (kGvUnroll * kWave);  // rounds whose every slot lies inside the row for every lane
    half8 wv[kGvUnroll][kGvRows];
    const uint32_t lane_off = lane * kVec;
    const half_t* const* rbase = wrow;  // (already wave-uniform for the compiler: built from blockIdx and the SGPR wave index)
    // request round `o` of every row into slot u (kClamp: the round may reach past the row's end)
    auto issue = [&](uint32_t o, int u, auto clamp_c) {
        constexpr bool kClamp = decltype(clamp_c)::value;
        if constexpr (!kClamp) {
            const uint32_t off = lane_off + (uint32_t)u * (kWave * kVec);
#pragma unroll
            for (int r = 0; r < kGvRows; ++r) wv[u][r] = ld8_stream(rbase[r] + (size_t)o * kRoundHalves + off);
        } else {
            const uint32_t v = lane + (o * kGvUnroll + (uint32_t)u) * kWave, vc = v < n_vec ? v : n_vec - 1;
#pragma unroll
            for (int r = 0; r < kGvRows; ++r) wv[u][r] = ld8_stream(rbase[r] + (size_t)vc * kVec);
        }
    };
    if (full > 0) {  // block-uniform
#pragma unroll
        for (int u = 0; u < kGvUnroll; ++u) issue(0, u, std::false_type{});
    } else {
#pragma unroll
        for (int u = 0; u < kGvUnroll; ++u) issue(0, u, std::true_type{});
    }

    // ---- input vector -> LDS (RMSNorm prologue: x * rsqrt(mean(x^2) + eps) * gamma, rounded to fp16 like the
    // stand-alone rms_norm_forward's output)
    const uint32_t pad_vec = outer * kGvUnroll * kWave;  // vectors the loop touches
    if (p.gamma) {
        float ss = 0.f;
        for (uint32_t v = tid; v < n_vec; v += kGvWaves * kWave) {
            const float8 xf = to_f32(ld8(p.x + (size_t)v * kVec));
#pragma unroll
            for (int i = 0; i < kVec; ++i) ss = __builtin_fmaf(xf[i], xf[i], ss);
        }
        ss = wave_allreduce_sum(ss, (int)lane);
        if (lane == 0) s_part[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < kGvWaves; ++w) tot += s_part[w];
        const float inv = 1.0f / sqrtf(tot / (float)p.in_dim + p.eps);
        for (uint32_t v = tid; v < pad_vec; v += kGvWaves * kWave) {
            half8 o = (half8)(half_t)0;
            if (v < n_vec) {
                const float8 xf = to_f32(ld8(p.x + (size_t)v * kVec)), g = to_f32(ld8(p.gamma + (size_t)v * kVec));
                float8 rr;
#pragma unroll
                for (int i = 0; i < kVec; ++i) rr[i] = xf[i] * inv * g[i];
                o = __builtin_convertvector(rr, half8);
            }
            st8(x_s + (size_t)v * kVec, o);
        }
    } else {
        for (uint32_t v = tid; v < pad_vec; v += kGvWaves * kWave)
            st8(x_s + (size_t)v * kVec, v < n_vec ? ld8(p.x + (size_t)v * kVec) : (half8)(half_t)0);
    }
    __syncthreads();

    // ---- stream the rows: a slot is re-issued for the next round as soon as it has been consumed (loads complete in
    // order, so the wait for slot u leaves the younger ones in flight).  Three copies of the round body keep the
    // steady state free of per-slot branches: next round full / next round clamped / last round.
    float acc[kGvRows] = {};
    auto consume = [&](uint32_t o, int u) {
        const uint32_t v = lane + (o * kGvUnroll + (uint32_t)u) * kWave;
        const half8 xv = ld8(x_s + (size_t)v * kVec);  // zero beyond in_dim: clamped weight vectors contribute nothing
#pragma unroll
        for (int r = 0; r < kGvRows; ++r) acc[r] = dot8(wv[u][r], xv, acc[r]);
    };
    for (uint32_t o = 0; o < outer; ++o) {
        if (o + 1 < full) {
#pragma unroll
            for (int u = 0; u < kGvUnroll; ++u) {
                consume(o, u);
                issue(o + 1, u, std::false_type{});
            }
        } else if (o + 1 < outer) {
#pragma unroll
            for (int u = 0; u < kGvUnroll; ++u) {
                consume(o, u);
                issue(o + 1, u, std::true_type{});
            }
        } else {
#pragma unroll
            for (int u = 0; u < kGvUnroll; ++u) consume(o, u);
        }
    }
#pragma unroll
    for (int r = 0; r < kGvRows; ++r) acc[r] = wave_allreduce_sum(acc[r], (int)lane);

    // ---- epilogue (every lane holds the four totals; lane 0 writes)
    if (lane != 0) return;
    if constexpr (MODE == kGvPlain) {
#pragma unroll
        for (int r = 0; r < kGvRows; ++r)
            if (live[r]) p.out[mat[r]][row[r]] = (half_t)acc[r];
    } else if constexpr (MODE == kGvResidual) {
#pragma unroll
        for (int r = 0; r < kGvRows; ++r)
            if (live[r]) p.out[0][row[r]] = (half_t)((float)p.residual[row[r]] + acc[r]);
    } else if constexpr (MODE == kGvSiluMul) {
#pragma unroll
        for (int r = 0; r < kGvRows; r += 2)
            if (live[r]) {
                const float g = acc[r], u = acc[r + 1];
                p.out[0][row[r]] = (half_t)(g / (1.0f + __expf(-g)) * u);
            }
    } else {  // kGvQkvRope
        const float pos = (float)(p.state->seq_len - 1);
#pragma unroll
        for (int r = 0; r < kGvRows; r += 2) {
            if (!live[r]) continue;
            if (mat[r] < 2) {  // q / k: (acc[r], acc[r+1]) = features (d, d + D/2) of one head
                const uint32_t d = row[r] % p.head_dim;  // < D/2
                const float freq = p.rcp_scale * exp2f(p.log2_rcp_theta * (float)(2 * d) / (float)p.head_dim);
                float s, c;
                sincosf(pos * freq, &s, &c);
                p.out[mat[r]][row[r]] = (half_t)(acc[r] * c - acc[r + 1] * s);
                p.out[mat[r]][row[r + 1]] = (half_t)(acc[r + 1] * c + acc[r] * s);
            } else {
                p.out[2][row[r]] = (half_t)acc[r];
                p.out[2][row[r + 1]] = (half_t)acc[r + 1];
            }
        }
    }
}

template <int MODE, int NWV, int RW, int U>
static int launch_gemv_rw(const GemvArgs& p, uint32_t virtual_rows, hipStream_t s) {
    const uint32_t n_vec = p.in_dim / kVec, iters = (n_vec + kWave - 1) / kWave;
    const uint32_t outer = (iters + U - 1) / U;
    const size_t lds = (size_t)outer * U * kWave * kVec * sizeof(half_t);
    if (lds > 60 * 1024) return QUEST_EUNSUPPORTED;  // in_dim <= 30720
    const uint32_t per_wg = NWV * RW;
    hipLaunchKernelGGL((gemv_kernel<MODE, NWV, RW, U>), dim3((virtual_rows + per_wg - 1) / per_wg), dim3(NWV * kWave), lds, s, p);
    QUEST_LAUNCH_CHECK();
    return 0;
}

// 8-wave workgroups, 8 loads of 16 bytes in flight per lane (RW x U = 8: <= 100 VGPRs, two workgroups = 16 waves = 128 KiB
// in flight per CU, the attention kernel's shape; 16 per lane needed 146-176 VGPRs and spilled under the 128 cap);
// the largest rows-per-wave that still gives two workgroups per CU (512).  QUEST_GEMV_CFG="RW,U" forces one of the
// built shapes (tuning).
template <int MODE>
static int launch_gemv(const GemvArgs& p, uint32_t virtual_rows, hipStream_t s) {
    constexpr uint32_t kWant = 512;
    static const int forced = [] {
        const char* e = getenv("QUEST_GEMV_CFG");
        int rw = 0, u = 0;
        return e && sscanf(e, "%d,%d", &rw, &u) == 2 ? rw * 100 + u : 0;
    }();
    constexpr bool kPairs = !(MODE == kGvPlain || MODE == kGvResidual);  // row-dots come in pairs: RW even
    switch (forced) {
        case 402: return launch_gemv_rw<MODE, 8, 4, 2>(p, virtual_rows, s);
        case 204: return launch_gemv_rw<MODE, 8, 2, 4>(p, virtual_rows, s);
        case 401: return launch_gemv_rw<MODE, 8, 4, 1>(p, virtual_rows, s);
        case 202: return launch_gemv_rw<MODE, 8, 2, 2>(p, virtual_rows, s);
        case 201: return launch_gemv_rw<MODE, 8, 2, 1>(p, virtual_rows, s);
        case 104: if constexpr (!kPairs) return launch_gemv_rw<MODE, 8, 1, 4>(p, virtual_rows, s); break;
        case 102: if constexpr (!kPairs) return launch_gemv_rw<MODE, 8, 1, 2>(p, virtual_rows, s); break;
        case 108: if constexpr (!kPairs) return launch_gemv_rw<MODE, 8, 1, 8>(p, virtual_rows, s); break;
        default: break;
    }
    if (virtual_rows / (8 * 4) >= kWant) return launch_gemv_rw<MODE, 8, 4, 2>(p, virtual_rows, s);
    if constexpr (!kPairs) {
        if (virtual_rows / (8 * 2) >= kWant) return launch_gemv_rw<MODE, 8, 2, 4>(p, virtual_rows, s);
        return launch_gemv_rw<MODE, 8, 1, 8>(p, virtual_rows, s);
    } else {
        return launch_gemv_rw<MODE, 8, 2, 4>(p, virtual_rows, s);
    }
}

}  // namespace quest

using namespace quest;

extern "C" int quest_decode_norm_gemv(const void* x, const void* gamma, float eps, const void* w, void* out,
                                      uint32_t in_dim, uint32_t out_dim, quest_stream_t stream) {
    if (!x || !w || !out || in_dim == 0 || out_dim == 0) return QUEST_EINVAL;
    if (in_dim % 8 != 0) return QUEST_EUNSUPPORTED;
    GemvArgs p{};
    p.x = (const half_t*)x, p.gamma = (const half_t*)gamma, p.eps = eps, p.in_dim = in_dim;
    p.w[0] = (const half_t*)w, p.rows[0] = out_dim, p.out[0] = (half_t*)out;
    return launch_gemv<kGvPlain>(p, out_dim, (hipStream_t)stream);
}

extern "C" int quest_decode_gemv_residual(const void* x, const void* w, void* h, uint32_t in_dim, uint32_t out_dim,
                                          quest_stream_t stream) {
    if (!x || !w || !h || in_dim == 0 || out_dim == 0) return QUEST_EINVAL;
    if (in_dim % 8 != 0) return QUEST_EUNSUPPORTED;
    GemvArgs p{};
    p.x = (const half_t*)x, p.in_dim = in_dim;
    p.w[0] = (const half_t*)w, p.rows[0] = out_dim, p.out[0] = (half_t*)h, p.residual = (const half_t*)h;
    return launch_gemv<kGvResidual>(p, out_dim, (hipStream_t)stream);
}

extern "C" int quest_decode_mlp_gate_up(const void* h, const void* gamma, float eps, const void* w_gate, const void* w_up,
                                        void* act, uint32_t hidden, uint32_t intermediate, quest_stream_t stream) {
    if (!h || !gamma || !w_gate || !w_up || !act || hidden == 0 || intermediate == 0) return QUEST_EINVAL;
    if (hidden % 8 != 0) return QUEST_EUNSUPPORTED;
    GemvArgs p{};
    p.x = (const half_t*)h, p.gamma = (const half_t*)gamma, p.eps = eps, p.in_dim = hidden;
    p.w[0] = (const half_t*)w_gate, p.w[1] = (const half_t*)w_up, p.rows[0] = p.rows[1] = intermediate;
    p.out[0] = (half_t*)act;
    return launch_gemv<kGvSiluMul>(p, 2 * intermediate, (hipStream_t)stream);
}

extern "C" int quest_decode_qkv_rope(const void* h, const void* gamma, float eps, const void* wq, const void* wk,
                                     const void* wv, void* q, void* k, void* v, uint32_t hidden, uint32_t num_qo_heads,
                                     uint32_t num_kv_heads, uint32_t head_dim, float rope_scale, float rope_theta,
                                     const quest_step_state_t* state, quest_stream_t stream) {
    if (!h || !gamma || !wq || !wk || !wv || !q || !k || !v || !state) return QUEST_EINVAL;
    if (hidden == 0 || num_qo_heads == 0 || num_kv_heads == 0 || rope_scale == 0.f || rope_theta <= 0.f) return QUEST_EINVAL;
    if (hidden % 8 != 0 || head_dim % 4 != 0) return QUEST_EUNSUPPORTED;  // a wave's 4 row-dots = two rotation pairs of one head
    GemvArgs p{};
    p.x = (const half_t*)h, p.gamma = (const half_t*)gamma, p.eps = eps, p.in_dim = hidden;
    p.w[0] = (const half_t*)wq, p.w[1] = (const half_t*)wk, p.w[2] = (const half_t*)wv;
    p.rows[0] = num_qo_heads * head_dim, p.rows[1] = p.rows[2] = num_kv_heads * head_dim;
    p.out[0] = (half_t*)q, p.out[1] = (half_t*)k, p.out[2] = (half_t*)v;
    p.head_dim = head_dim, p.rcp_scale = 1.0f / rope_scale, p.log2_rcp_theta = -log2f(rope_theta), p.state = state;
    return launch_gemv<kGvQkvRope>(p, p.rows[0] + p.rows[1] + p.rows[2], (hipStream_t)stream);
}
