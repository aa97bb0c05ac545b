// Shared device helpers for the gfx950 kernels.  wave = 64 lanes; a "row" is the group of
// head_dim/8 lanes (16 for D=128) that together hold one fp16 vector, 8 halves (16 B) per lane.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/quest_hip.h"

namespace quest {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float float8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;
constexpr int kVec = 8;  // halves per lane per 16-byte access

// Strides (in halves) of one pool layer; both layouts of decode_page.cuh:196-239 reduce to these.
struct PoolStrides {
    uint32_t page;    // 2 * S * H * D
    uint32_t v_off;   // S * H * D
    uint32_t head;    // NHD: D        HND: S * D
    uint32_t entry;   // NHD: H * D    HND: D
};

__host__ __device__ inline PoolStrides pool_strides(const quest_paged_kv_t& p) {
    PoolStrides s;
    s.v_off = p.page_size * p.num_heads * p.head_dim;
    s.page = 2u * s.v_off;
    if (p.layout == QUEST_LAYOUT_HND) {
        s.head = p.page_size * p.head_dim;
        s.entry = p.head_dim;
    } else {
        s.head = p.head_dim;
        s.entry = p.num_heads * p.head_dim;
    }
    return s;
}

__device__ __forceinline__ half8 ld8(const half_t* p) { return *reinterpret_cast<const half8*>(p); }
__device__ __forceinline__ void st8(half_t* p, half8 v) { *reinterpret_cast<half8*>(p) = v; }

__device__ __forceinline__ float8 to_f32(half8 h) { return __builtin_convertvector(h, float8); }

// Butterfly sum over the W lanes of a row, offsets W/2 .. 1, every lane ends with the same bits.
// This is the reduction order of the reference kernels (decode_attn.cuh:101-104, :157-160) and of
// the CPU oracle, so fp32 results agree bit for bit.
template <int W>
__device__ __forceinline__ float row_allreduce_sum(float x) {
#pragma unroll
    for (int off = W / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    return x;
}

__device__ __forceinline__ uint16_t half_bits(half_t h) { return __builtin_bit_cast(uint16_t, h); }
__device__ __forceinline__ half_t bits_half(uint16_t b) { return __builtin_bit_cast(half_t, b); }

// Order-preserving 16-bit key of an fp16 bit pattern (RAFT radix-select twiddle).
__device__ __forceinline__ uint32_t half_key(uint16_t b) {
    return (b & 0x8000u) ? (uint32_t)(uint16_t)~b : (uint32_t)(b | 0x8000u);
}

}  // namespace quest

#define QUEST_LAUNCH_CHECK()                  \
    do {                                      \
        hipError_t e__ = hipGetLastError();   \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)
