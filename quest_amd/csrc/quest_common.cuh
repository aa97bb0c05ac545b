// Shared device helpers for the gfx950 kernels.  wave = 64 lanes; a "row" is the group of
// head_dim/8 lanes (16 for D=128) that together hold one fp16 vector, 8 halves (16 B) per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/quest_hip.h"

namespace quest {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;
constexpr int kVec = 8;  // halves per lane per 16-byte access

// Strides (in halves) of one pool layer; the layouts of decode_page.cuh:196-239 reduce to these.  `rot` / `vflip` are the
// row rotation of QUEST_LAYOUT_NHD_ROT (include/quest_hip.h), 0 for the reference's two layouts: the K / max vector of
// (head h, entry e) lives in head slot  h ^ (e & rot)  of the entry's row of heads, its V / min vector in that slot ^ vflip.
struct PoolStrides {
    uint32_t page;    // 2 * S * H * D
    uint32_t v_off;   // S * H * D
    uint32_t head;    // NHD: D        HND: S * D
    uint32_t entry;   // NHD: H * D    HND: D
    uint32_t rot;     // NHD_ROT: min(H & -H, 4) - 1, else 0
    uint32_t vflip;   // NHD_ROT: (min(H & -H, 32) - 1) & ~3, else 0
};

__host__ __device__ inline PoolStrides pool_strides(const quest_paged_kv_t& p) {
    PoolStrides s;
    s.v_off = p.page_size * p.num_heads * p.head_dim;
    s.page = 2u * s.v_off;
    s.rot = s.vflip = 0;
    if (p.layout == QUEST_LAYOUT_HND) {
        s.head = p.page_size * p.head_dim;
        s.entry = p.head_dim;
    } else {
        s.head = p.head_dim;
        s.entry = p.num_heads * p.head_dim;
        if (p.layout == QUEST_LAYOUT_NHD_ROT) {
            const uint32_t low = p.num_heads & (0u - p.num_heads);  // largest power of two dividing the head count
            s.rot = (low < 4u ? low : 4u) - 1u;
            s.vflip = ((low < 32u ? low : 32u) - 1u) & ~3u;
        }
    }
    return s;
}
// head slot of head h's K / max vector in entry e (e = index INSIDE the page)
__host__ __device__ inline uint32_t pool_slot(const PoolStrides& s, uint32_t h, uint32_t e) { return h ^ (e & s.rot); }
// halves from the K / max vector in head slot `slot` to its V / min vector (slot ^ vflip, one tensor further).  The flip
// only touches slot bits >= 2, which the rotation (bits 0-1) leaves alone: the distance is a per-HEAD constant,
// pool_v_off(s, h).  (unsigned wrap-around: the true result lies in (0, page))
__host__ __device__ inline uint32_t pool_v_off(const PoolStrides& s, uint32_t slot) {
    return s.v_off + (s.vflip - 2u * (slot & s.vflip)) * s.head;
}
// The R-row walk of one head's page (attention gather, one-launch layer): entry e = t R + row (R = rows per wave load
// instruction, a power of two; row < R).  slot(e) = hk ^ (e & rot) splits into a per-LANE part (the bits below R) and a
// wave-UNIFORM part per round t, so the address stays  page base + uniform[t] + lane offset  whatever the layout:
//   lane offset  = row * entry + ((hk & (R-1)) ^ (row & rot)) * head + col * 8
//   uniform[t]   = t R * entry + ((hk ^ (t R & rot)) & ~(R-1)) * head
__host__ __device__ inline uint32_t walk_lane_off(const PoolStrides& s, uint32_t hk, uint32_t R, uint32_t row, uint32_t col8) {
    return row * s.entry + (((hk & (R - 1u)) ^ (row & s.rot)) * s.head) + col8;
}
__host__ __device__ inline uint32_t walk_uniform(const PoolStrides& s, uint32_t hk, uint32_t R, uint32_t t) {
    return t * R * s.entry + ((hk ^ (t * R & s.rot)) & ~(R - 1u)) * s.head;
}

__device__ __forceinline__ half8 ld8(const half_t* p) { return *reinterpret_cast<const half8*>(p); }
// Load of a WAVE-UNIFORM address from memory that nobody writes during the kernel (page tables, the step state): through
// the constant address space, i.e. an s_load (scalar cache) into an SGPR -- the compiler otherwise emits a 64-lane
// vector load of one address, with the vector path's latency, at the head of the table -> data dependency chain.
__device__ __forceinline__ int32_t ld_uniform_i32(const int32_t* p) {
    typedef const int32_t __attribute__((address_space(4))) * cptr;
    return *(cptr)(uintptr_t)p;
}
// Streaming (read-once) 16 B load: `nt` cache policy, for K/V pages and metadata that no other
// workgroup re-reads (MI355X guide, nt-weights row: issued->landed -18 % on once-read streams).
__device__ __forceinline__ half8 ld8_stream(const half_t* p) {
#ifdef QUEST_NO_NT
    return *reinterpret_cast<const half8*>(p);
#else
    return __builtin_nontemporal_load(reinterpret_cast<const half8*>(p));
#endif
}
__device__ __forceinline__ void st8(half_t* p, half8 v) { *reinterpret_cast<half8*>(p) = v; }

__device__ __forceinline__ float8 to_f32(half8 h) { return __builtin_convertvector(h, float8); }

// Butterfly sum over the W lanes of a row, offsets W/2 .. 1, every lane ends with the same bits: the
// reduction order of the reference kernels (decode_attn.cuh:101-104, :157-160).  Each step is a
// ds_bpermute (LDS crossbar) round trip on gfx9; the hot kernels use row_allreduce_sum_fast below.
template <int W>
__device__ __forceinline__ float row_allreduce_sum(float x) {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    return x;
}

// ---- DPP cross-lane helpers (gfx9 encodings).  One VALU instruction each, no LDS crossbar trip.
// masked-off / out-of-range lanes contribute `old` = 0.
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, BANK_MASK, true);
}
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float x) {
    return __builtin_bit_cast(float, dpp_i<CTRL, ROW_MASK, BANK_MASK>(__builtin_bit_cast(int, x)));
}
constexpr int kDppRowShr = 0x110, kDppRowRor = 0x120, kDppHalfMirror = 0x141, kDppBcast15 = 0x142, kDppBcast31 = 0x143;
constexpr int kDppQuadXor1 = 0xB1, kDppQuadXor2 = 0x4E;  // quad_perm [1,0,3,2], [2,3,0,1]

// The value of the lane OFF lanes away (lane ^ OFF) without an LDS-crossbar trip: OFF = 8 is a DPP rotation inside
// the 16-lane row; 16 and 32 use gfx950's row-swap instructions (v_permlane16_swap / v_permlane32_swap exchange the
// odd 16- / 32-lane rows of one register with the even rows of another: with both operands = x, one of the two
// results holds the partner's value in every lane).  ds_bpermute (what __shfl_xor compiles to) costs an LDS round
// trip per call; these are one or two VALU instructions.
typedef unsigned uint2v __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ float lane_xor(float x, int lane) {
    static_assert(OFF == 8 || OFF == 16 || OFF == 32, "lane distance");
    if constexpr (OFF == 8) {
        return dpp_f<kDppRowRor + 8>(x);
    } else {
        const unsigned xb = __builtin_bit_cast(unsigned, x);
        uint2v r;
        if constexpr (OFF == 16) r = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);
        else r = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
        // r[0]: own even rows kept, odd rows replaced by the even rows; r[1]: the odd rows everywhere
        return __builtin_bit_cast(float, (lane & OFF) ? r[0] : r[1]);
    }
}
// Apply f(integral_constant<int, OFF>) for OFF = FROM, 2 FROM, ... < 64 (compile-time lane distances).
template <int FROM, typename F>
__device__ __forceinline__ void for_each_row_distance(F&& f) {
    if constexpr (FROM < 64) {
        f(std::integral_constant<int, FROM>{});
        for_each_row_distance<FROM * 2>(f);
    }
}

// All-reduce over the 64 lanes of a wave (every lane gets the result) in six VALU steps: quad permutations for the
// distances 1 and 2, half-row mirror for 4 (the quads are uniform by then), row rotation for 8, row swaps for 16, 32.
template <typename Op>
__device__ __forceinline__ float wave_allreduce(float x, int lane, Op op) {
    x = op(x, dpp_f<kDppQuadXor1>(x));
    x = op(x, dpp_f<kDppQuadXor2>(x));
    x = op(x, dpp_f<kDppHalfMirror>(x));
    x = op(x, dpp_f<kDppRowRor + 8>(x));
    x = op(x, lane_xor<16>(x, lane));
    x = op(x, lane_xor<32>(x, lane));
    return x;
}
__device__ __forceinline__ float wave_allreduce_max(float x, int lane) {
    return wave_allreduce(x, lane, [](float a, float b) { return __builtin_fmaxf(a, b); });
}
__device__ __forceinline__ float wave_allreduce_sum(float x, int lane) {
    return wave_allreduce(x, lane, [](float a, float b) { return a + b; });
}

// Sum over the W lanes of a row, every lane gets the total, by DPP rotations (row_ror 8,4,2,1: one VALU
// instruction per step).  The association order differs from the xor butterfly above.  The estimate kernel
// uses THIS tree and the CPU oracle (qo_row_reduce) restates it, so their fp32 sums agree bit for bit; the
// attention kernels use it too (their parity bar is a tolerance).
template <int W>
__device__ __forceinline__ float row_allreduce_sum_fast(float x) {
    static_assert(W == 8 || W == 16 || W == 32, "row width");
    if constexpr (W == 8) {
        x += dpp_f<kDppQuadXor1>(x);
        x += dpp_f<kDppQuadXor2>(x);
        x += dpp_f<kDppHalfMirror>(x);
    } else {
        x += dpp_f<kDppRowRor + 8>(x);
        x += dpp_f<kDppRowRor + 4>(x);
        x += dpp_f<kDppRowRor + 2>(x);
        x += dpp_f<kDppRowRor + 1>(x);
        if constexpr (W == 32) x += lane_xor<16>(x, (int)(threadIdx.x & 63));
    }
    return x;
}

// G per-lane values, each to be summed over the 16 lanes of a row (D = 128), in the SAME association order as
// row_allreduce_sum_fast<16> (pairs at lane distance 8, then 4, 2, 1 -- fp32 addition is commutative bit for bit, so
// only the tree matters), but computing every node of the G trees once instead of on all 16 lanes: at the first
// log2(G) levels a lane keeps half of its values and hands the other half to its partner.  Returns the finished
// sum of value `which` in the lanes with (lane & (16/G - 1)) == 0, where `which` is made of the lane's high bits.
// 12 VALU instructions for G = 4 instead of 16.
constexpr int kDppRowShl = 0x100;
template <int G>
__device__ __forceinline__ float row_segmented_sum16(const float (&v)[G], int col, int& which) {
    static_assert(G == 2 || G == 4 || G == 8, "group size");
    const bool b3 = (col & 8) != 0, b2 = (col & 4) != 0, b1 = (col & 2) != 0;
    // level 8: partner = col ^ 8 (row_ror 8 is symmetric)
    float w[G / 2];
#pragma unroll
    for (int i = 0; i < G / 2; ++i) {
        const float keep = b3 ? v[G / 2 + i] : v[i], send = b3 ? v[i] : v[G / 2 + i];
        w[i] = keep + dpp_f<kDppRowRor + 8>(send);
    }
    // level 4: partner = col ^ 4 = col - 4 where bit 2 is set (row_shr 4), col + 4 where it is not (row_shl 4).  (After
    // the first split the two 8-lane halves of the row hold different values: a rotation would mix them.)
    float x;
    if constexpr (G == 2) {
        x = w[0];
        const float from_lo = dpp_f<kDppRowShr + 4>(x), from_hi = dpp_f<kDppRowShl + 4>(x);
        x += b2 ? from_lo : from_hi;
        x += dpp_f<kDppQuadXor2>(x);
        x += dpp_f<kDppQuadXor1>(x);
        which = b3 ? 1 : 0;
        return x;
    } else {
        float y[G / 4];
#pragma unroll
        for (int i = 0; i < G / 4; ++i) {
            const float keep = b2 ? w[G / 4 + i] : w[i], send = b2 ? w[i] : w[G / 4 + i];
            const float from_lo = dpp_f<kDppRowShr + 4>(send), from_hi = dpp_f<kDppRowShl + 4>(send);
            y[i] = keep + (b2 ? from_lo : from_hi);
        }
        if constexpr (G == 4) {
            x = y[0];
            x += dpp_f<kDppQuadXor2>(x);
            x += dpp_f<kDppQuadXor1>(x);
            which = (b3 ? 2 : 0) + (b2 ? 1 : 0);
            return x;
        } else {
            // level 2: partner = col ^ 2 (quad permutation)
            const float keep = b1 ? y[1] : y[0], send = b1 ? y[0] : y[1];
            x = keep + dpp_f<kDppQuadXor2>(send);
            x += dpp_f<kDppQuadXor1>(x);
            which = (b3 ? 4 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);
            return x;
        }
    }
}

// Inclusive prefix sum over the 64 lanes of a wave in 7 DPP adds.
__device__ __forceinline__ uint32_t wave_scan_incl_dpp(uint32_t x) {
    int v = (int)x;
    const int s1 = dpp_i<kDppRowShr + 1>(v), s2 = dpp_i<kDppRowShr + 2>(v), s3 = dpp_i<kDppRowShr + 3>(v);
    v += s1;
    v += s2;
    v += s3;
    v += dpp_i<kDppRowShr + 4, 0xF, 0xE>(v);
    v += dpp_i<kDppRowShr + 8, 0xF, 0xC>(v);
    v += dpp_i<kDppBcast15, 0xA, 0xF>(v);
    v += dpp_i<kDppBcast31, 0xC, 0xF>(v);
    return (uint32_t)v;
}

__device__ __forceinline__ uint16_t half_bits(half_t h) { return __builtin_bit_cast(uint16_t, h); }
__device__ __forceinline__ half_t bits_half(uint16_t b) { return __builtin_bit_cast(half_t, b); }

// Order-preserving 16-bit key of an fp16 bit pattern (RAFT radix-select twiddle).
__device__ __forceinline__ uint32_t half_key(uint16_t b) {
    return (b & 0x8000u) ? (uint32_t)(uint16_t)~b : (uint32_t)(b | 0x8000u);
}

}  // namespace quest

// Branch-weight hints for rarely taken paths (the append role of the estimate grid, the literal-form estimate).  They
// are kept OUT of the attention kernel: marking its second-slot branch as likely cost 3.5 us per batched launch and 1.2
// at cfg 4 (gpurun_out/r3j_*), and hints on its cold front-end fallbacks bought nothing.  What does pay there is code
// SIZE: the 12 KiB instantiation that holds only the front-end variant in use is 0.57 us per launch faster than the
// generic 27 KiB kernel (sparse_decode_kernel's VF parameter).  -DQUEST_NO_LAYOUT_HINTS turns the hints off (A/B).
#ifdef QUEST_NO_LAYOUT_HINTS
#define QUEST_UNLIKELY(x) (x)
#define QUEST_LIKELY(x) (x)
#else
#define QUEST_UNLIKELY(x) __builtin_expect(!!(x), 0)
#define QUEST_LIKELY(x) __builtin_expect(!!(x), 1)
#endif

// Tuning knobs (environment variables read by the launch paths) are honoured only when QUEST_TUNING=1 is set as well:
// an inherited environment must not silently change which kernel a product launch takes.
#include <cstdlib>
static inline const char* quest_tuning_env(const char* name) {
    static const bool on = [] { const char* e = getenv("QUEST_TUNING"); return e && atoi(e) != 0; }();
    return on ? getenv(name) : nullptr;
}

#define QUEST_LAUNCH_CHECK()                  \
    do {                                      \
        hipError_t e__ = hipGetLastError();   \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)
