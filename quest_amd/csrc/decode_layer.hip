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
#include <atomic>
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
        const char* e = quest_tuning_env("QUEST_GEMV_CFG");
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

// ---------------------------------------------------------------------------------------------------------------------
// The same four launches for n <= 16 decode tokens at once (one per sequence of a batch: BatchedInferenceController).
// The weights are still read exactly once, so the launch is as HBM-bound as the batch-1 one, but 8 tokens x 2 flops per
// weight no longer fit the vector ALUs next to the loads (44 TFLOP/s of v_dot2 at 5.5 TB/s): the products go through the
// matrix cores -- in the one MFMA shape whose operand layout IS a coalesced row sweep.
//
//   v_mfma_f32_4x4x4_16B_f16 = 16 independent 4 x 4 x 4 products, lane 4b + i holding row i of block b's A, lane 4b + j
//   column j of its B (4 halves each), lane 4b + j column j of its D (4 floats).  With block = a k chunk, A = 4 tokens and
//   B = 4 weight rows, a lane (b, j) loads 16 contiguous bytes of weight row j at k = 128 s + 8 b: one wave instruction
//   reads 4 rows x 256 contiguous bytes -- full 128-byte lines, like the batch-1 kernel -- and feeds two MFMAs (k halves)
//   per token group with no staging and no shuffle.  (The 16 x 16 x 32 shape, first version of this kernel, wants 16 rows
//   x 64 bytes per instruction: twice the lines per byte through the texture addresser -- 3.1 TB/s against 5.1-5.6.)
//   The tokens' inputs are loaded the same way (lane (b, i): token i, same k chunk; L2 hits, 4 tokens x 256 bytes per
//   instruction), 0.5 input bytes per weight byte at 8 tokens x 16 rows; the RMSNorm prologue is applied to that
//   fragment in registers (x * inv_rms[token] * gamma, rounded to fp16 like rms_norm_forward), gamma from LDS.
//
//   workgroup = 16 consecutive (virtual) weight rows (4 quads) x the whole reduction, split over its NW waves by k: wave w
//   takes the 128-wide k steps w, w + NW, ...; U steps in flight per lane.  The 16 k chunks of a lane's accumulators meet
//   by DPP at the end of the wave, the waves' 16 x n tiles in LDS; 128 threads then own one (token, row pair) each and
//   run the same epilogues as the batch-1 kernel.  TG = token groups of 4 (n <= 4 TG).
struct SkinnyArgs {
    GemvArgs g;              // x: [n_tokens][x_stride], out[i]: [n_tokens][out_stride[i]], residual like out[0]
    uint32_t n_tokens;       // 1..16
    uint32_t x_stride;       // halves between the inputs of consecutive tokens
    uint32_t out_stride[3];  // halves between the outputs of consecutive tokens
    uint32_t state_stride;   // kGvQkvRope: quest_step_state_t records between consecutive tokens' states (batched state: 1)
};

typedef float float4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

// (matrix, row) of virtual row vr of a launch (the batch-1 kernel's order: gate / up interleaved; rotation pairs adjacent)
template <int MODE>
__device__ __forceinline__ bool skinny_row(const GemvArgs& p, uint32_t vr, uint32_t& m, uint32_t& rr) {
    bool live;
    m = 0;
    if constexpr (MODE == kGvSiluMul) {
        m = vr & 1u, rr = vr >> 1, live = rr < p.rows[0];
    } else {
        while (m < 2 && vr >= p.rows[m]) vr -= p.rows[m], ++m;
        live = vr < p.rows[m], rr = vr;
        if constexpr (MODE == kGvQkvRope) {
            const uint32_t hd = p.head_dim, head = rr / hd, j = rr % hd;
            rr = head * hd + (j >> 1) + (j & 1u) * (hd / 2);
        }
    }
    return live;
}

template <int MODE, int NW, int U, int TG>
__global__ __launch_bounds__(NW* kWave, 2) void skinny_kernel(SkinnyArgs a) {  // 8 waves per CU: <= 256 VGPRs
    const GemvArgs& p = a.g;
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
    half_t* gamma_s = reinterpret_cast<half_t*>(sk_smem);  // RMSNorm weight (in_dim halves; launches with a prologue only)
    __shared__ float s_ss[4 * TG][NW];                     // per token, per wave: sum of squares
    __shared__ float s_part[NW][16][4 * TG];               // the waves' partial tiles [row][token]
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t j = lane & 3u, blk = lane >> 2;  // B column (weight row of a quad) = A row (token of a group); k chunk

    // ---- the lane's four weight rows (quad q: virtual row 16 block + 4 q + j) and TG input rows (token 4 t + j)
    const half_t* wrow[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        uint32_t m, rr;
        const bool live = skinny_row<MODE>(p, blockIdx.x * 16u + 4u * (uint32_t)q + j, m, rr);
        wrow[q] = p.w[live ? m : 0] + (size_t)(live ? rr : 0u) * p.in_dim;  // dead rows read row 0 (never written out)
    }
    const half_t* xrow[TG];
    bool tok[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        tok[t] = 4u * (uint32_t)t + j < a.n_tokens;
        xrow[t] = p.x + (size_t)(tok[t] ? 4u * (uint32_t)t + j : 0u) * a.x_stride;
    }
    const bool norm = p.gamma != nullptr;  // block-uniform

    const uint32_t n_steps = (p.in_dim + 127u) / 128u;
    const uint32_t mine = wave < n_steps ? (n_steps - wave + NW - 1) / NW : 0u;  // this wave's steps
    half8 wv[U][4], xv[U][TG];
    auto issue = [&](uint32_t s, int u) {
        const uint32_t k0 = (wave + NW * s) * 128u + blk * 8u;
        const uint32_t kc = k0 < p.in_dim ? k0 : 0u;  // clamped; masked at use
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[u][q] = ld8_stream(wrow[q] + kc);
#pragma unroll
        for (int t = 0; t < TG; ++t) xv[u][t] = ld8(xrow[t] + kc);
    };
#pragma unroll
    for (int u = 0; u < U; ++u) issue((uint32_t)u, u);

    // ---- RMSNorm prologue: 1 / rms of every token (every workgroup recomputes them; their loads queue behind the first
    // weight fragments)
    float inv[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) inv[t] = 1.0f;
    if (norm) {
        const uint32_t n_vec = p.in_dim / kVec;
        for (uint32_t t = 0; t < a.n_tokens; ++t) {
            float ss = 0.f;
            for (uint32_t v = tid; v < n_vec; v += NW * kWave) {
                const float8 xf = to_f32(ld8(p.x + (size_t)t * a.x_stride + (size_t)v * kVec));
#pragma unroll
                for (int i = 0; i < kVec; ++i) ss = __builtin_fmaf(xf[i], xf[i], ss);
            }
            ss = wave_allreduce_sum(ss, (int)lane);
            if (lane == 0) s_ss[t][wave] = ss;
        }
        for (uint32_t v = tid; v < n_vec; v += NW * kWave) st8(gamma_s + (size_t)v * kVec, ld8(p.gamma + (size_t)v * kVec));
        __syncthreads();
#pragma unroll
        for (int t = 0; t < TG; ++t)
            if (tok[t]) {
                float tot = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) tot += s_ss[4 * t + j][w];
                inv[t] = 1.0f / sqrtf(tot / (float)p.in_dim + p.eps);
            }
    }

    // ---- stream
    float4_t acc[4][TG];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < TG; ++t) acc[q][t] = float4_t{0.f, 0.f, 0.f, 0.f};
    auto consume = [&](uint32_t s, int u) {
        const uint32_t k0 = (wave + NW * s) * 128u + blk * 8u;
        const bool in = k0 < p.in_dim;
        float8 gf;
        if (norm) gf = to_f32(ld8(gamma_s + (in ? k0 : 0u)));
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            half8 b = xv[u][t];
            if (norm) {
                const float8 xf = to_f32(b);
                float8 rr;
#pragma unroll
                for (int i = 0; i < kVec; ++i) rr[i] = xf[i] * inv[t] * gf[i];
                b = __builtin_convertvector(rr, half8);
            }
            if (!tok[t] || !in) b = (half8)(half_t)0;  // dead token / past the row's end: contributes nothing
            const half4_t b0 = {b[0], b[1], b[2], b[3]}, b1 = {b[4], b[5], b[6], b[7]};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const half8 w8 = wv[u][q];
                const half4_t w0 = {w8[0], w8[1], w8[2], w8[3]}, w1 = {w8[4], w8[5], w8[6], w8[7]};
                acc[q][t] = __builtin_amdgcn_mfma_f32_4x4x4f16(b0, w0, acc[q][t], 0, 0, 0);  // A = tokens, B = weight rows
                acc[q][t] = __builtin_amdgcn_mfma_f32_4x4x4f16(b1, w1, acc[q][t], 0, 0, 0);
            }
        }
    };
    const uint32_t rounds = (mine + U - 1) / U;
    for (uint32_t o = 0; o < rounds; ++o) {
        if (o + 1 < rounds) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                consume(o * U + (uint32_t)u, u);
                issue((o + 1) * U + (uint32_t)u, u);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) consume(o * U + (uint32_t)u, u);
        }
    }
    // ---- the 16 k chunks of the wave meet: lanes 4 b + j over b (distances 4, 8 inside a row of 16 lanes, then 16, 32);
    // D element e of acc[q][t] at lane (b, j) = token 4 t + e, row 4 q + j
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < TG; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[q][t][e];
                v += dpp_f<kDppRowRor + 8>(v);
                v += dpp_f<kDppRowRor + 4>(v);
                v += lane_xor<16>(v, (int)lane);
                v += lane_xor<32>(v, (int)lane);
                if (lane < 4u) s_part[wave][4 * q + lane][4 * t + e] = v;
            }
    __syncthreads();

    // ---- 128 threads: (token, row pair)
    if (tid >= 128u) return;
    const uint32_t token = tid & 15u, pair = tid >> 4;
    if (token >= a.n_tokens) return;
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v0 += s_part[w][2u * pair][token], v1 += s_part[w][2u * pair + 1u][token];
    // the pair's two virtual rows -> (matrix, row)
    uint32_t mat[2], row[2];
    bool live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) live[i] = skinny_row<MODE>(p, blockIdx.x * 16u + 2u * pair + (uint32_t)i, mat[i], row[i]);
    if constexpr (MODE == kGvPlain) {
        if (live[0]) p.out[mat[0]][(size_t)token * a.out_stride[mat[0]] + row[0]] = (half_t)v0;
        if (live[1]) p.out[mat[1]][(size_t)token * a.out_stride[mat[1]] + row[1]] = (half_t)v1;
    } else if constexpr (MODE == kGvResidual) {
        const size_t base = (size_t)token * a.out_stride[0];
        if (live[0]) p.out[0][base + row[0]] = (half_t)((float)p.residual[base + row[0]] + v0);
        if (live[1]) p.out[0][base + row[1]] = (half_t)((float)p.residual[base + row[1]] + v1);
    } else if constexpr (MODE == kGvSiluMul) {
        if (live[0]) p.out[0][(size_t)token * a.out_stride[0] + row[0]] = (half_t)(v0 / (1.0f + __expf(-v0)) * v1);
    } else {  // kGvQkvRope: rows come in (d, d + D/2) pairs of one head of one matrix
        if (!live[0]) return;
        half_t* o = p.out[mat[0]] + (size_t)token * a.out_stride[mat[0]];
        if (mat[0] < 2) {
            const float pos = (float)(p.state[(size_t)token * a.state_stride].seq_len - 1);
            const uint32_t d = row[0] % p.head_dim;
            const float freq = p.rcp_scale * exp2f(p.log2_rcp_theta * (float)(2 * d) / (float)p.head_dim);
            float sn, cs;
            sincosf(pos * freq, &sn, &cs);
            o[row[0]] = (half_t)(v0 * cs - v1 * sn);
            o[row[1]] = (half_t)(v1 * cs + v0 * sn);
        } else {
            o[row[0]] = (half_t)v0;
            o[row[1]] = (half_t)v1;
        }
    }
}

// One 8-wave workgroup per CU (two waves per SIMD: 256 VGPRs each -- U x (4 + TG) fragments of 16 bytes in flight per lane
// next to 16 TG accumulators; 128 KiB of weights in flight per CU at U = 4.  16 waves at 128 VGPRs spilled from TG = 2 on).
// The fallback of the persistent kernel below: any n x in_dim, inputs from L2 (3.0-3.8 TB/s at 8 tokens).
template <int MODE, int TG>
static int launch_skinny_tg(const SkinnyArgs& a, uint32_t blocks, size_t lds, hipStream_t s) {
    constexpr int U = TG <= 2 ? 4 : 2;
    hipLaunchKernelGGL((skinny_kernel<MODE, 8, U, TG>), dim3(blocks), dim3(8 * kWave), lds, s, a);
    QUEST_LAUNCH_CHECK();
    return 0;
}
template <int MODE>
static int launch_skinny(const SkinnyArgs& a, uint32_t virtual_rows, hipStream_t s) {
    const uint32_t blocks = (virtual_rows + 15u) / 16u;
    const size_t lds = a.g.gamma ? (size_t)a.g.in_dim * sizeof(half_t) : 0;
    if (lds > 48 * 1024) return QUEST_EUNSUPPORTED;  // in_dim <= 24576 with a RMSNorm prologue
    if (a.n_tokens <= 4) return launch_skinny_tg<MODE, 1>(a, blocks, lds, s);
    if (a.n_tokens <= 8) return launch_skinny_tg<MODE, 2>(a, blocks, lds, s);
    return launch_skinny_tg<MODE, 4>(a, blocks, lds, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// The n-token launches proper: ONE 16-wave workgroup per CU, resident for the whole launch, with the tokens' inputs --
// normalised once -- in ITS LDS (8 tokens x 4096 x 2 bytes = 64 KiB of the CU's 160), so that the weight stream is the
// only thing that crosses the memory system: the first n-token kernel above re-reads the inputs from L2 for every 16 rows
// (0.5 bytes per weight byte, and a RMSNorm per fragment); the batch-1 kernel with 8 accumulators per row-dot spends the
// vector ALUs (61 us against 32 for the gate/up launch).
//
//   task = one quad of weight rows (the B operand of the 4x4x4 MFMA: lane (b, j) = 16 bytes of row j at k chunk b, a
//   fully coalesced 4 x 256-byte sweep) x one of `ks` interleaved k slices (k steps slice, slice + ks, ...: at most CM
//   per task).  A wave holds TWO sets of CM fragments: it requests the next task's set, then consumes the current one
//   against the A fragments (lane (b, i) = token i, chunk b) read from LDS -- row stride = 64 bytes mod 256, conflict-free
//   for ds_read_b128's four 16-lane groups -- so a task's loads fly during the previous task's MFMAs, its k-chunk
//   reduction (8 TG values by DPP), and the round's barrier.  The 16 waves of the workgroup = 16 / ks quads x ks slices;
//   the slices of a quad meet in LDS (double-buffered, one barrier per round), (quad, row pair, token) threads run the
//   batch-1 kernel's epilogues.  Rounds: quad = (round x workgroups + workgroup) x (16 / ks) + wave / ks.
//
//   Inputs too large for LDS (down_proj: 8 x 11008 x 2 = 172 KiB) take the first kernel.  (Staging them in k ranges through
//   two LDS buffers, the accumulators living across the ranges, was built and measured: 32.8 us against that kernel's
//   23.7 at 8 tokens -- removed.)
//
//   Order of requests (in-kernel stamps, scripts/ps_timeline.py): a CU's memory pipeline holds about 64 KiB of requests;
//   a wave whose loads do not fit stalls at issue (up to 4 us at the start of a launch), and a wave's loads return in
//   order.  The prologue requests the inputs first (every wave its share, a bare barrier so that all 16 have, before
//   anybody's weights), normalises and stores them, and then starts the weights; with the first weight set requested
//   ahead of the inputs the launch takes the same time (37.3 vs 37.1 us).
constexpr int kPsWaves = 16, kCM = 4;  // kCM: k steps per task = fragments per set (two sets of 8 spill next to the inputs' registers)
// input vectors (16 bytes) per thread and staged range: 4096 per workgroup, 8192 for 9-16 tokens
constexpr int ps_input_vectors(int tg) { return (tg <= 2 ? 64 : 128) / kPsWaves; }
struct PersistPlan {
    uint32_t ks, ks_log2;     // k slices per quad (power of two <= 16)
    uint32_t rounds;
    uint32_t spp;             // 128-wide k steps of a row
    uint32_t x_row;           // halves between tokens in LDS (>= 128 spp, = 32 mod 128)
    uint32_t n_quads;
    uint32_t vpp_magic;       // ceil(2^32 / (16 spp)): e / (16 spp) = umulhi(e, magic) for e < 2^16
};

template <int MODE, int TG, int CM>
__global__ __launch_bounds__(kPsWaves* kWave, 4) void persist_kernel(SkinnyArgs a, PersistPlan pl) {
    const GemvArgs& p = a.g;
    constexpr int kXV = ps_input_vectors(TG);
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];
    half_t* x_s = reinterpret_cast<half_t*>(ps_smem);  // [4 TG][x_row] (normalised) inputs
    __shared__ float s_inv[4 * TG];
    __shared__ float s_part[2][kPsWaves][4][4 * TG];  // [buffer][wave = quad slot x slice][row of the quad][token]
    // kGvQkvRope: (cos, sin) of every (token, frequency), computed once per workgroup while the first weights are on their
    // way; a round's epilogue reads it (no sincosf, with its large-argument path, per round)
    constexpr int kRopeMaxHalfD = 128;
    __shared__ float2 s_rope[MODE == kGvQkvRope ? 4 * TG * kRopeMaxHalfD : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef QUEST_PS_TIMELINE  // tuning build: s_memtime stamps of (workgroup 0 | last, wave 0 | 15) -> p.out[2] (16 x 4 x int64)
    long long ps_t[16];
    int ps_n = 0;
#define PS_STAMP() do { if (ps_n < 16) ps_t[ps_n++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PS_STAMP() do {} while (0)
#endif
    PS_STAMP();
    const uint32_t j = lane & 3u, blk = lane >> 2;
    const uint32_t ks = pl.ks, slice = wave & (ks - 1u), qslot = wave >> pl.ks_log2, qw = kPsWaves >> pl.ks_log2;
    const bool norm = p.gamma != nullptr;
    const uint32_t n_steps = (p.in_dim + 127u) / 128u;
    // workgroup barrier that publishes LDS only: __syncthreads() also waits for the global loads in flight (vmcnt(0))
    // whenever global stores are pending -- the epilogue's -- which would drain the prefetched set every round
    auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    const uint32_t n_seg = pl.rounds;  // a segment = one round of tasks
    auto seg_quad = [&](uint32_t seg) { return (seg * gridDim.x + blockIdx.x) * qw + qslot; };
    auto row_ptr = [&](uint32_t quad) -> const half_t* {
        uint32_t m, rr;
        const bool live = quad < pl.n_quads && skinny_row<MODE>(p, 4u * quad + j, m, rr);
        return p.w[live ? m : 0] + (size_t)(live ? rr : 0u) * p.in_dim;  // dead rows read row 0 (never written out)
    };
    // Every call requests exactly CM fragments, whatever the task: steps past the slice's end and quads past the matrix
    // re-read cached bytes.  (A load under a branch makes the compiler's s_waitcnt bookkeeping assume it may not have been
    // issued: every later wait for an OLDER load became vmcnt(0) -- the consumption of a set waited for the set requested
    // just before it, i.e. nothing was in flight under the MFMAs.  For the same reason the requests past the last segment
    // are issued all the same -- one cached line -- instead of being guarded.)
    half8 wa[CM], wb[CM];
    auto issue = [&](half8 (&w)[CM], uint32_t seg) {
        const bool real = seg < n_seg;  // block-uniform; past the end: CM requests of ONE cached line, nobody waits for them
        const half_t* wrow = real ? row_ptr(seg_quad(seg)) : p.w[0];
        const uint32_t s1 = real ? n_steps : 0u;
#pragma unroll
        for (int i = 0; i < CM; ++i) {
            const uint32_t st = slice + ks * (uint32_t)i, k0 = st * 128u + blk * 8u;
            w[i] = ld8_stream(wrow + (st < s1 && k0 < p.in_dim ? k0 : 0u));
        }
    };

    // ---- the inputs: element e = tid + 1024 i -> (token e / vpp, vector e % vpp), vpp = 16 spp vectors per token (the
    // host guarantees tokens x vpp <= kXV x 1024); always kXV loads (clamped)
    const uint32_t vpp = pl.spp * 16u, x_tot = a.n_tokens * vpp;
    half8 xr[kXV];
    auto x_load = [&] {
#pragma unroll
        for (int i = 0; i < kXV; ++i) {
            const uint32_t e = tid + (uint32_t)i * kPsWaves * kWave, t = __umulhi(e, pl.vpp_magic), k0 = (e - t * vpp) * kVec;
            const bool in = e < x_tot && k0 < p.in_dim;
            xr[i] = ld8(p.x + (in ? (size_t)t * a.x_stride + k0 : (size_t)0));
        }
    };
    auto x_store = [&](half_t* dst) {  // zero for dead tokens and past the row's end
#pragma unroll
        for (int i = 0; i < kXV; ++i) {
            const uint32_t e = tid + (uint32_t)i * kPsWaves * kWave, t = __umulhi(e, pl.vpp_magic), v = e - t * vpp;
            if (e >= 4u * TG * vpp) continue;
            half8 o = xr[i];
            const bool in = e < x_tot && v * kVec < p.in_dim;
            if (norm && in) {
                const float8 xf = to_f32(o), g = to_f32(ld8(p.gamma + v * kVec));
                const float inv = s_inv[t];
                float8 rr;
#pragma unroll
                for (int c = 0; c < kVec; ++c) rr[c] = xf[c] * inv * g[c];
                o = __builtin_convertvector(rr, half8);
            }
            st8(dst + (size_t)t * pl.x_row + (size_t)v * kVec, in ? o : (half8)(half_t)0);
        }
    };

    // ---- prologue: inputs first (see above)
    x_load();
    if constexpr (MODE == kGvQkvRope) {  // under the inputs' latency; published by the staging barrier
        const uint32_t half_d = p.head_dim / 2u;
        for (uint32_t e = tid; e < a.n_tokens * half_d; e += kPsWaves * kWave) {
            const uint32_t t = e / half_d, d = e % half_d;
            const float pos = (float)(p.state[(size_t)t * a.state_stride].seq_len - 1);
            const float freq = p.rcp_scale * exp2f(p.log2_rcp_theta * (float)(2 * d) / (float)p.head_dim);
            float sn, cs;
            sincosf(pos * freq, &sn, &cs);
            s_rope[t * kRopeMaxHalfD + d] = make_float2(cs, sn);
        }
    }

    __builtin_amdgcn_s_barrier();
    PS_STAMP();
    if (norm) {  // 1 / rms per token: a thread's vector i lies in ONE token; per-token totals through LDS atomics
        if (tid < 4u * TG) s_inv[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kXV; ++i) {
            const uint32_t e = tid + (uint32_t)i * kPsWaves * kWave, t = __umulhi(e, pl.vpp_magic);
            float ss = 0.f;
            if (e < x_tot && (e - t * vpp) * kVec < p.in_dim) {
                const float8 xf = to_f32(xr[i]);
#pragma unroll
                for (int c = 0; c < kVec; ++c) ss = __builtin_fmaf(xf[c], xf[c], ss);
            }
            // lanes of a wave hold at most two tokens (vpp >= 64): reduce each side, one atomic per side
            const uint32_t t0 = __builtin_amdgcn_readfirstlane(t);
            const float s_lo = wave_allreduce_sum(t == t0 ? ss : 0.f, (int)lane), s_hi = wave_allreduce_sum(t == t0 ? 0.f : ss, (int)lane);
            if (lane == 0 && t0 < 4u * TG) atomicAdd(&s_inv[t0], s_lo);
            if (lane == 0 && t0 + 1u < 4u * TG) atomicAdd(&s_inv[t0 + 1u], s_hi);
        }
        __syncthreads();
        if (tid < 4u * TG) s_inv[tid] = 1.0f / sqrtf(s_inv[tid] / (float)p.in_dim + p.eps);
        __syncthreads();
    }
    x_store(x_s);
    PS_STAMP();
    lds_barrier();
    issue(wa, 0);
    PS_STAMP();
    float4_t acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t) acc[t] = float4_t{0.f, 0.f, 0.f, 0.f};
    auto consume = [&](const half8 (&w)[CM], uint32_t seg) {
        if (seg_quad(seg) >= pl.n_quads) return;  // wave-uniform
        const half_t* xb = x_s;
#pragma unroll
        for (int i = 0; i < CM; ++i) {
            const uint32_t st = slice + ks * (uint32_t)i;
            if (st < n_steps) {  // wave-uniform
                const uint32_t kl = st * 128u + blk * 8u;
                const half8 w8 = w[i];
                const half4_t w0 = {w8[0], w8[1], w8[2], w8[3]}, w1 = {w8[4], w8[5], w8[6], w8[7]};
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    const half8 x8 = ld8(xb + (size_t)(4u * (uint32_t)t + j) * pl.x_row + kl);
                    const half4_t x0 = {x8[0], x8[1], x8[2], x8[3]}, x1 = {x8[4], x8[5], x8[6], x8[7]};
                    acc[t] = __builtin_amdgcn_mfma_f32_4x4x4f16(x0, w0, acc[t], 0, 0, 0);  // A = tokens, B = weight rows
                    acc[t] = __builtin_amdgcn_mfma_f32_4x4x4f16(x1, w1, acc[t], 0, 0, 0);
                }
            }
        }
    };
    // the k slices of the quads meet; epilogue
    auto finish = [&](uint32_t seg) {
        const uint32_t buf = seg & 1u;
#pragma unroll
        for (int t = 0; t < TG; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // D element e at lane (b, j) = token 4 t + e, row j of the quad; sum over b
                float v = acc[t][e];
                v += dpp_f<kDppRowRor + 8>(v);
                v += dpp_f<kDppRowRor + 4>(v);
                v += lane_xor<16>(v, (int)lane);
                v += lane_xor<32>(v, (int)lane);
                if (lane < 4u) s_part[buf][wave][lane][4 * t + e] = v;
                acc[t][e] = 0.f;
            }
        lds_barrier();
        // (quad slot, row pair, token) threads
        const uint32_t token = tid & 15u, pair = (tid >> 4) & 1u, qs = tid >> 5;
        if (qs >= qw || token >= a.n_tokens) return;
        const uint32_t quad = (seg * gridDim.x + blockIdx.x) * qw + qs;
        if (quad >= pl.n_quads) return;
        float v0 = 0.f, v1 = 0.f;
        for (uint32_t sl = 0; sl < ks; ++sl) {
            v0 += s_part[buf][qs * ks + sl][2u * pair][token];
            v1 += s_part[buf][qs * ks + sl][2u * pair + 1u][token];
        }
        uint32_t mat[2], row[2];
        bool live[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) live[i] = skinny_row<MODE>(p, 4u * quad + 2u * pair + (uint32_t)i, mat[i], row[i]);
        if constexpr (MODE == kGvPlain) {
            if (live[0]) p.out[mat[0]][(size_t)token * a.out_stride[mat[0]] + row[0]] = (half_t)v0;
            if (live[1]) p.out[mat[1]][(size_t)token * a.out_stride[mat[1]] + row[1]] = (half_t)v1;
        } else if constexpr (MODE == kGvResidual) {
            const size_t base = (size_t)token * a.out_stride[0];
            if (live[0]) p.out[0][base + row[0]] = (half_t)((float)p.residual[base + row[0]] + v0);
            if (live[1]) p.out[0][base + row[1]] = (half_t)((float)p.residual[base + row[1]] + v1);
        } else if constexpr (MODE == kGvSiluMul) {
            if (live[0]) p.out[0][(size_t)token * a.out_stride[0] + row[0]] = (half_t)(v0 / (1.0f + __expf(-v0)) * v1);
        } else {  // kGvQkvRope: rows come in (d, d + D/2) pairs of one head of one matrix
            if (!live[0]) return;
            half_t* o = p.out[mat[0]] + (size_t)token * a.out_stride[mat[0]];
            if (mat[0] < 2) {
                const float2 r = s_rope[token * kRopeMaxHalfD + row[0] % p.head_dim];  // row[0] % head_dim < D / 2
                const float cs = r.x, sn = r.y;
                o[row[0]] = (half_t)(v0 * cs - v1 * sn);
                o[row[1]] = (half_t)(v1 * cs + v0 * sn);
            } else {
                o[row[0]] = (half_t)v0;
                o[row[1]] = (half_t)v1;
            }
        }
    };
    // Segment s: request segment s + 1, consume the set of s, let the slices meet.  A CU's memory
    // pipeline holds ~64 KiB of requests; a wave whose requests do not fit stalls at issue.  Requested HERE, after the
    // previous segment's barrier, a stalled wave holds nobody up; requested two segments ahead, right after the
    // consumption and before the barrier, the stall sat in front of the barrier: 40.3 us instead of 36.8 for the gate/up
    // launch (more in flight than the pipeline holds buys nothing).
    auto body = [&](const half8 (&cur)[CM], half8 (&nxt)[CM], uint32_t sg) {
        issue(nxt, sg + 1);
        PS_STAMP();
        consume(cur, sg);
        PS_STAMP();
        finish(sg);
        PS_STAMP();
    };
    for (uint32_t seg = 0;; seg += 2) {  // wa holds segment seg; leaves by break (no merge with loads behind it)
        body(wa, wb, seg);
        if (seg + 1 >= n_seg) break;
        body(wb, wa, seg + 1);
        if (seg + 2 >= n_seg) break;
    }
    PS_STAMP();
#ifdef QUEST_PS_TIMELINE
    if (lane == 0 && (wave == 0 || wave == kPsWaves - 1) && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && p.out[2]) {
        long long* o = reinterpret_cast<long long*>(p.out[2]) + ((blockIdx.x ? 2 : 0) + (wave ? 1 : 0)) * 16;
        for (int i = 0; i < 16; ++i) o[i] = i < ps_n ? ps_t[i] : 0;
    }
#endif
}

// dynamic LDS a workgroup may ask for: the CU's 160 KiB less the kernel's static arrays (s_part 2 TG KiB, the rotation table
// 4 TG KiB) and some slack
constexpr int kMaxDevices = 64;
static int current_device() {  // index into the per-device caches below; -1: unknown (no caching)
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    return dev;
}
static size_t persist_lds_budget(uint32_t tg, bool rope) { return (size_t)(158 - tg * (2 + (rope ? 4 : 0))) * 1024; }

template <int MODE, int TG>
static int launch_persist_tg(const SkinnyArgs& a, const PersistPlan& pl, uint32_t grid, size_t lds, hipStream_t s) {
    // more than 64 KiB of dynamic LDS needs the opt-in, once per DEVICE (a process may drive several GPUs) -- kept in a
    // per-device table of atomics so that two threads launching on different devices do not race on one flag
    static std::atomic<bool> attr[kMaxDevices];
    const int dev = current_device();
    if (dev < 0 || !attr[dev].load(std::memory_order_acquire)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&persist_kernel<MODE, TG, kCM>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)persist_lds_budget(TG, MODE == kGvQkvRope));
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) attr[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((persist_kernel<MODE, TG, kCM>), dim3(grid), dim3(kPsWaves * kWave), lds, s, a, pl);
    QUEST_LAUNCH_CHECK();
    return 0;
}

static int persist_cus() {  // compute units of the CURRENT device (cached per device)
    static std::atomic<int> cus[kMaxDevices];
    const int dev = current_device();
    int n = dev >= 0 ? cus[dev].load(std::memory_order_relaxed) : 0;
    if (n <= 0) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) n = 0;
        n = n > 0 ? n : 256;
        if (dev >= 0) cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// Plan of the persistent kernel, or false when the shape does not fit it (then: launch_skinny).
static bool plan_persist(const SkinnyArgs& a, uint32_t virtual_rows, uint32_t tg, bool rope, PersistPlan& pl, uint32_t& grid,
                         size_t& lds) {
    const uint32_t spp = (a.g.in_dim + 127u) / 128u, tokens = 4u * tg;  // 128-wide k steps of a row
    const size_t budget = persist_lds_budget(tg, rope);
    pl.n_quads = (virtual_rows + 3u) / 4u;
    grid = (uint32_t)persist_cus();
    const uint32_t x_row = spp * 128u + 32u;  // + 64 bytes: = 64 mod 256 bytes
    const size_t need = (size_t)tokens * x_row * sizeof(half_t);
    if (need > budget) return false;  // the inputs do not fit LDS (down_proj at 8 tokens)
    if ((size_t)tokens * spp * 16u > (size_t)ps_input_vectors((int)tg) * kPsWaves * kWave) return false;  // <= kXV vectors per thread
    if (spp * 16u < (uint32_t)kWave) return false;  // (the 1 / rms reduction assumes <= 2 tokens per wave sweep)
    uint32_t ks = 1, lg = 0;
    while (ks < (uint32_t)kPsWaves && (spp + ks - 1) / ks > (uint32_t)kCM) ks *= 2, ++lg;
    if ((spp + ks - 1) / ks > (uint32_t)kCM) return false;  // rows longer than 16 slices x 4 steps x 128
    // more slices per quad = fewer quads per workgroup and round = more, smaller rounds: take the split whose last round
    // is fullest, as long as a task keeps >= 2 loads per lane
    auto rounds_of = [&](uint32_t k) { return (pl.n_quads + grid * ((uint32_t)kPsWaves / k) - 1) / (grid * ((uint32_t)kPsWaves / k)); };
    auto eff_of = [&](uint32_t k) { return (double)pl.n_quads / ((double)rounds_of(k) * grid * ((uint32_t)kPsWaves / k)); };
    for (uint32_t k = ks * 2, l = lg + 1; k <= (uint32_t)kPsWaves && (spp + k - 1) / k >= 2; k *= 2, ++l)
        if (eff_of(k) > eff_of(ks) + 0.04) ks = k, lg = l;
    pl.ks = ks, pl.ks_log2 = lg, pl.rounds = rounds_of(ks), pl.spp = spp, pl.x_row = x_row;
    pl.vpp_magic = (uint32_t)(0xffffffffu / (spp * 16u)) + 1u;
    constexpr size_t kFloor = 81 * 1024;  // more than half of the CU's LDS: exactly one workgroup per CU
    lds = need > kFloor ? need : kFloor;
    return true;
}

// QUEST_BATCHED_GEMV=skinny forces the first kernel (tuning / tests).
template <int MODE>
static int launch_batched(const SkinnyArgs& a, uint32_t virtual_rows, hipStream_t s) {
    if (a.n_tokens == 0 || a.n_tokens > 16) return QUEST_EINVAL;
    if (a.g.in_dim % 8 != 0) return QUEST_EUNSUPPORTED;
    static const bool force_skinny = [] {
        const char* e = quest_tuning_env("QUEST_BATCHED_GEMV");
        return e && e[0] == 's';
    }();
    const uint32_t tg = a.n_tokens <= 4 ? 1 : (a.n_tokens <= 8 ? 2 : 4);
    PersistPlan pl{};
    uint32_t grid = 0;
    size_t lds = 0;
    if (MODE == kGvQkvRope && a.g.head_dim > 256) return launch_skinny<MODE>(a, virtual_rows, s);  // (the LDS rotation table)
    if (!force_skinny && plan_persist(a, virtual_rows, tg, MODE == kGvQkvRope, pl, grid, lds)) {
        if (tg == 1) return launch_persist_tg<MODE, 1>(a, pl, grid, lds, s);
        if (tg == 2) return launch_persist_tg<MODE, 2>(a, pl, grid, lds, s);
        return launch_persist_tg<MODE, 4>(a, pl, grid, lds, s);
    }
    return launch_skinny<MODE>(a, virtual_rows, s);
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

// ---- the same for n_tokens <= 16 decode tokens (x: [n_tokens][in_dim], outputs [n_tokens][out_dim], contiguous)
extern "C" int quest_decode_norm_gemv_batched(const void* x, const void* gamma, float eps, const void* w, void* out,
                                              uint32_t in_dim, uint32_t out_dim, uint32_t n_tokens, quest_stream_t stream) {
    if (!x || !w || !out || in_dim == 0 || out_dim == 0) return QUEST_EINVAL;
    SkinnyArgs a{};
    GemvArgs& p = a.g;
    p.x = (const half_t*)x, p.gamma = (const half_t*)gamma, p.eps = eps, p.in_dim = in_dim;
    p.w[0] = (const half_t*)w, p.rows[0] = out_dim, p.out[0] = (half_t*)out;
    a.n_tokens = n_tokens, a.x_stride = in_dim, a.out_stride[0] = out_dim;
    return launch_batched<kGvPlain>(a, out_dim, (hipStream_t)stream);
}

extern "C" int quest_decode_gemv_residual_batched(const void* x, const void* w, void* h, uint32_t in_dim, uint32_t out_dim,
                                                  uint32_t n_tokens, quest_stream_t stream) {
    if (!x || !w || !h || in_dim == 0 || out_dim == 0) return QUEST_EINVAL;
    SkinnyArgs a{};
    GemvArgs& p = a.g;
    p.x = (const half_t*)x, p.in_dim = in_dim;
    p.w[0] = (const half_t*)w, p.rows[0] = out_dim, p.out[0] = (half_t*)h, p.residual = (const half_t*)h;
    a.n_tokens = n_tokens, a.x_stride = in_dim, a.out_stride[0] = out_dim;
#ifdef QUEST_PS_TIMELINE
    if (const char* e = quest_tuning_env("QUEST_PS_DEBUG_PTR")) p.out[2] = reinterpret_cast<half_t*>(strtoull(e, nullptr, 0));
#endif
    return launch_batched<kGvResidual>(a, out_dim, (hipStream_t)stream);
}

extern "C" int quest_decode_mlp_gate_up_batched(const void* h, const void* gamma, float eps, const void* w_gate,
                                                const void* w_up, void* act, uint32_t hidden, uint32_t intermediate,
                                                uint32_t n_tokens, quest_stream_t stream) {
    if (!h || !gamma || !w_gate || !w_up || !act || hidden == 0 || intermediate == 0) return QUEST_EINVAL;
    SkinnyArgs a{};
    GemvArgs& p = a.g;
    p.x = (const half_t*)h, p.gamma = (const half_t*)gamma, p.eps = eps, p.in_dim = hidden;
    p.w[0] = (const half_t*)w_gate, p.w[1] = (const half_t*)w_up, p.rows[0] = p.rows[1] = intermediate;
    p.out[0] = (half_t*)act;
    a.n_tokens = n_tokens, a.x_stride = hidden, a.out_stride[0] = intermediate;
    return launch_batched<kGvSiluMul>(a, 2 * intermediate, (hipStream_t)stream);
}

extern "C" int quest_decode_qkv_rope_batched(const void* h, const void* gamma, float eps, const void* wq, const void* wk,
                                             const void* wv, void* q, void* k, void* v, uint32_t hidden,
                                             uint32_t num_qo_heads, uint32_t num_kv_heads, uint32_t head_dim,
                                             float rope_scale, float rope_theta, const quest_step_state_t* states,
                                             uint32_t n_tokens, quest_stream_t stream) {
    if (!h || !gamma || !wq || !wk || !wv || !q || !k || !v || !states) return QUEST_EINVAL;
    if (hidden == 0 || num_qo_heads == 0 || num_kv_heads == 0 || rope_scale == 0.f || rope_theta <= 0.f) return QUEST_EINVAL;
    if (head_dim % 16 != 0) return QUEST_EUNSUPPORTED;  // a 16-row block = 8 rotation pairs of ONE head of ONE matrix
    SkinnyArgs a{};
    GemvArgs& p = a.g;
    p.x = (const half_t*)h, p.gamma = (const half_t*)gamma, p.eps = eps, p.in_dim = hidden;
    p.w[0] = (const half_t*)wq, p.w[1] = (const half_t*)wk, p.w[2] = (const half_t*)wv;
    p.rows[0] = num_qo_heads * head_dim, p.rows[1] = p.rows[2] = num_kv_heads * head_dim;
    p.out[0] = (half_t*)q, p.out[1] = (half_t*)k, p.out[2] = (half_t*)v;
    p.head_dim = head_dim, p.rcp_scale = 1.0f / rope_scale, p.log2_rcp_theta = -log2f(rope_theta), p.state = states;
    a.n_tokens = n_tokens, a.x_stride = hidden;
    a.out_stride[0] = p.rows[0], a.out_stride[1] = p.rows[1], a.out_stride[2] = p.rows[2];
    a.state_stride = 1;
    return launch_batched<kGvQkvRope>(a, p.rows[0] + p.rows[1] + p.rows[2], (hipStream_t)stream);
}

// Which kernel an n-token launch of this shape takes (tests / tuning; no launch): returns 1 and fills
// info = {k slices per quad, rounds, dynamic LDS bytes, workgroups, 128-wide k steps per row, fragments per set} for the
// persistent kernel, 0 for the first (inputs-from-L2) kernel, < 0 for argument errors as the launches return them.
extern "C" int quest_decode_batched_plan(uint32_t in_dim, uint32_t virtual_rows, uint32_t n_tokens, int rope, uint32_t info[6]) {
    if (in_dim == 0 || virtual_rows == 0 || n_tokens == 0 || n_tokens > 16 || !info) return QUEST_EINVAL;
    if (in_dim % 8 != 0) return QUEST_EUNSUPPORTED;
    SkinnyArgs a{};
    a.g.in_dim = in_dim, a.n_tokens = n_tokens;
    const uint32_t tg = n_tokens <= 4 ? 1 : (n_tokens <= 8 ? 2 : 4);
    PersistPlan pl{};
    uint32_t grid = 0;
    size_t lds = 0;
    if (!plan_persist(a, virtual_rows, tg, rope != 0, pl, grid, lds)) return 0;
    info[0] = pl.ks, info[1] = pl.rounds, info[2] = (uint32_t)lds, info[3] = grid, info[4] = pl.spp, info[5] = (uint32_t)kCM;
    return 1;
}
