// Why do the (sequence, head) workgroups of a batched launch on the NHD pool finish at different times although they move
// the same bytes?  (profiles/r05_wallstamps_layer_*: heads 1, 5, 9, ... and head 31 end 13 us after the others.)
// This probe reproduces the access shape without any of the kernels' logic: workgroup (class c, replica r) of 8 waves reads
// 128 "pages" (random 256 KiB regions of a 16 GiB buffer); of each page 32 pieces of 256 bytes at a stride of 8 KiB (16 K
// rows, 16 V rows 128 KiB further) at byte offset c * 256 inside the 8 KiB row -- what head c of an NHD pool reads.
// Every workgroup stamps its start and end with the 100 MHz wall clock; printed: mean duration per class.
//   mode 0: class = workgroup id mod 32 (XCD = class mod 8, the batched launches' plain order)
//   mode 1: class = (id + 3 * (id / 32)) mod 32 (every XCD serves all classes)
//   shift : the buffer base is moved by shift * 256 bytes (is the class a property of the address or of the head index?)
//   mask  : only classes with (mask >> (class mod 4)) & 1 run (do the slow classes slow each other down, or are they slow alone?)
//   rot   : (round 6) the piece of token row e of head c sits in head SLOT  1: c ^ (e & 3)   2: (c + e) mod 32   3: c ^ (e & 15)
//           of its 8 KiB row instead of slot c -- a row-rotated NHD pool: every workgroup then reads all four values of address
//           bits 8-9 in equal parts.  Does every workgroup see the mix's mean?
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probe/addr_class_probe.hip -o scripts/probe/addr_class_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                     \
        }                                                                                 \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

// NW waves per workgroup (128 pages per workgroup whatever NW); KIND 0: nontemporal loads (the kernels'), 1: plain loads
template <int NW, int KIND>
__global__ __launch_bounds__(NW * 64) void probe(const char* __restrict__ base, uint32_t n_regions, uint32_t mode, uint32_t mask,
                                                 uint32_t salt, uint32_t hnd, uint32_t rot, unsigned* stamps, uint32_t* sink) {
    const uint32_t id = blockIdx.x, r = id / 32;
    uint32_t c = mode == 0 ? id % 32 : (id + 3 * r) % 32;
    if (mode >= 100 && mode < 200) c = mode - 100;  // every workgroup reads the SAME class: what that class of addresses can deliver
    if (mode >= 200) {  // two classes, A = (mode - 200) / 32 and B = (mode - 200) % 32, alternating by workgroup PAIR (both on every XCD)
        c = ((id >> 3) & 1u) ? (mode - 200) % 32 : (mode - 200) / 32;
    }
    if (!((mask >> (c % 4)) & 1u)) {
        if (threadIdx.x == 0) stamps[2 * id] = stamps[2 * id + 1] = 0;
        return;
    }
    const unsigned t0 = (unsigned)wall_clock64();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, row = lane >> 4, col = lane & 15;
    u32x4 acc = (u32x4)(0u);
    // NHD: piece (token row t, head c) at t * 8 KiB + c * 256 B; HND: at c * 4 KiB + t * 256 B
    const size_t lane_off = (hnd & 1u) ? (size_t)c * 4096 + row * 256 + col * 16 : (size_t)row * 8192 + c * 256 + col * 16;
    const size_t step = (hnd & 1u) ? 4 * 256 : 4 * 8192;
    // (round 6) geometry of the pool row: H heads of 256 B; hnd >> 8 = 8: GQA pool (8 kv heads, 2 KiB rows, 64 KiB pages; class =
    // kv head of query head id mod 32)
    const uint32_t H = hnd >> 8 ? (hnd >> 8) : 32u;
    const bool is_hnd = (hnd & 1u) != 0;
    if (H != 32u && mode < 100) c = (mode == 0 ? id % 32 : (id + 3 * r) % 32) / (32u / H);
    const size_t row_bytes = (size_t)H * 256, v_off = 16 * row_bytes;
    size_t offs[4], offs_v[4];  // byte offset of this lane's piece in round t of a tensor (token row e = 4 t + row)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint32_t e = 4 * t + row, hm = H - 1;
        uint32_t slot = c, slot_v;
        if (rot == 1) slot = c ^ (e & 3u);
        if (rot == 2) slot = (c + e) & hm;
        if (rot == 3) slot = c ^ (e & 15u & hm);
        if (rot == 4 || rot == 5) slot = c ^ (e & 15u & hm);
        if (rot == 7) slot = c ^ ((2 * e) & hm);
        if (rot >= 64) slot = c ^ (e & ((rot - 64) & 15u) & hm);   // general form: rot = 64 + mask + (V flip << 4)
        slot_v = slot;
        if (rot >= 64) slot_v = slot ^ (((rot - 64) >> 4) & hm);
        if (rot == 4) slot_v = slot ^ (16u & hm);   // V pieces in the other half of the row
        if (rot == 7) slot_v = slot ^ 1u;
        offs[t] = is_hnd ? (size_t)c * 4096 + e * 256 + col * 16 : (size_t)e * row_bytes + slot * 256 + col * 16;
        offs_v[t] = is_hnd ? offs[t] + 131072 : (size_t)e * row_bytes + slot_v * 256 + col * 16 + v_off;
    }
    const size_t region_bytes = 2 * v_off;   // one page
    const uint32_t nreg = n_regions * (uint32_t)(262144 / region_bytes);  // the same 16 GiB whatever the page size
    const uint32_t page_flip = rot == 5 ? (16u & (H - 1)) * 256u : 0u;  // rot 5: odd physical pages use the other half of the row
    if (KIND == 4) {  // every class with 4 loads in flight per lane (half a page per round)
        for (uint32_t it = 0; it < 256 / NW; ++it) {
            u32x4 v[4];
            const uint32_t region = mix(salt + (id * NW + wave) * 16 + it / 2) % n_regions;
            const char* p = base + (size_t)region * 262144 + lane_off + (it & 1) * 131072;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + t * step));
#pragma unroll
            for (int i = 0; i < 4; ++i) acc ^= v[i];
        }
    } else
    if ((KIND == 2 && c % 4 != 1 && c != 31) || KIND == 3) {  // (the fast) classes with HALF the loads in flight (one page per round)
        for (uint32_t it = 0; it < 128 / NW; ++it) {
            u32x4 v[8];
            const uint32_t region = mix(salt + (id * NW + wave) * 16 + it) % nreg;
            const char* p = base + (size_t)region * region_bytes;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v[t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (offs[t] ^ ((region & 1u) * page_flip))));
                v[4 + t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (offs_v[t] ^ ((region & 1u) * page_flip))));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc ^= v[i];
        }
    } else
    for (uint32_t it = 0; it < 64 / NW; ++it) {
        u32x4 v[16];
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            const uint32_t region = mix(salt + (id * NW + wave) * 16 + it * 2 + pg) % nreg;
            const char* p = base + (size_t)region * region_bytes;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (KIND != 1) {
                    v[pg * 8 + t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (offs[t] ^ ((region & 1u) * page_flip))));
                    v[pg * 8 + 4 + t] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + (offs_v[t] ^ ((region & 1u) * page_flip))));
                } else {
                    v[pg * 8 + t] = *reinterpret_cast<const u32x4*>(p + offs[t]);
                    v[pg * 8 + 4 + t] = *reinterpret_cast<const u32x4*>(p + offs_v[t]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[2 * id] = t0;
        stamps[2 * id + 1] = (unsigned)wall_clock64();
    }
}

int main(int argc, char** argv) {
    CK(hipSetDevice(0));
    const size_t bytes = (size_t)16 << 30;
    char* buf;
    uint32_t* sink;
    unsigned* stamps;
    CK(hipMalloc(&buf, bytes + (1 << 20)));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&stamps, 256 * 2 * sizeof(unsigned)));
    CK(hipMemset(buf, 1, bytes + (1 << 20)));
    const uint32_t n_regions = (uint32_t)(bytes / 262144);
    struct Case { uint32_t mode, shift, mask, hnd; const char* what; int nw = 8, kind = 0; uint32_t rot = 0; };
    const bool rot_only = argc > 1 && argv[1][0] == 'r';
    const bool rot2_only = argc > 1 && argv[1][0] == 'r' && (argv[1][1] == '2' || argv[1][1] == '3');  // `addr_class_probe r`: only the round-6 rotated-layout cases
    const Case cases[] = {
        {0, 0, 15, 0, "[rot] NHD pieces, class = id mod 32 (the batched launches' order), plain layout"},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, slot = c ^ (e & 3)", 8, 0, 1},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, slot = (c + e) mod 32", 8, 0, 2},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, slot = c ^ (e & 15)", 8, 0, 3},
        {1, 0, 15, 0, "[rot] NHD, classes rotated over the XCDs, plain layout"},
        {1, 0, 15, 0, "[rot] NHD, XCD-rotated, slot = c ^ (e & 3)", 8, 0, 1},
        {1, 0, 15, 0, "[rot] NHD, XCD-rotated, slot = (c + e) mod 32", 8, 0, 2},
        {1, 0, 15, 0, "[rot] NHD, XCD-rotated, slot = c ^ (e & 15)", 8, 0, 3},
        {0, 1, 15, 0, "[rot] NHD, id mod 32, slot = c ^ (e & 3), base + 256 B", 8, 0, 1},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, 8 loads in flight, plain layout", 8, 3, 0},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, 8 loads in flight, slot = c ^ (e & 3)", 8, 3, 1},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, 8 loads in flight, slot = (c + e) mod 32", 8, 3, 2},
        {0, 0, 15, 1, "[rot] HND tiles, class = id mod 32 (for comparison)"},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, K slot = c ^ e, V slot = c ^ e ^ 16", 8, 0, 4},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, slot = c ^ e ^ 16 * (page & 1)", 8, 0, 5},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, K slot = c ^ 2e, V slot = c ^ 2e ^ 1", 8, 0, 7},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, 8 loads in flight, slot = c ^ (e & 15)", 8, 3, 3},
        {0, 0, 15, 0, "[rot] NHD, id mod 32, 8 loads in flight, K slot = c ^ e, V slot = c ^ e ^ 16", 8, 3, 4},
        {0, 0, 15, 0, "[rot2] NHD, id mod 32, K slot = c ^ (e & 7), V the same", 8, 0, 64 + 7},
        {0, 0, 15, 0, "[rot2] NHD, id mod 32, K slot = c ^ (e & 7), V slot ^ 8", 8, 0, 64 + 7 + (8 << 4)},
        {0, 0, 15, 0, "[rot2] NHD, id mod 32, K slot = c ^ (e & 7), V slot ^ 16", 8, 0, 64 + 7 + (16 << 4)},
        {0, 0, 15, 0, "[rot2] NHD, id mod 32, K slot = c ^ (e & 7), V slot ^ 24", 8, 0, 64 + 7 + (24 << 4)},
        {0, 0, 15, 0, "[rot2] NHD, id mod 32, K slot = c ^ (e & 15), V slot ^ 16", 8, 0, 64 + 15 + (16 << 4)},
        {0, 0, 15, 0, "[rot2] NHD, id mod 32, K slot = c ^ (e & 3), V slot ^ 4", 8, 0, 64 + 3 + (4 << 4)},
        {0, 0, 15, 0, "[rot2] NHD, id mod 32, K slot = c ^ (e & 3), V slot ^ 28", 8, 0, 64 + 3 + (28 << 4)},
        {0, 0, 15, 0, "[rot2] NHD, id mod 32, plain K, V slot ^ 16", 8, 0, 64 + 0 + (16 << 4)},
        {0, 0, 15, 0, "[rot2] NHD, id mod 32, plain layout (again)", 8, 0, 0},
        {1, 0, 15, 0, "[rot2] NHD, XCD-rotated, K slot = c ^ (e & 15), V slot ^ 16", 8, 0, 64 + 15 + (16 << 4)},
        {1, 0, 15, 0, "[rot2] NHD, XCD-rotated, K slot = c ^ (e & 7), V slot ^ 8", 8, 0, 64 + 7 + (8 << 4)},
        {0, 0, 15, 8 << 8, "[rot2] GQA pool, K slot = c ^ (e & 3), V slot ^ 4", 8, 0, 64 + 3 + (4 << 4)},
        {0, 0, 15, 8 << 8, "[rot2] GQA pool, K slot = c ^ (e & 7), V the same", 8, 0, 64 + 7},
        {0, 0, 15, 0, "[rot3] round 0 id mod 32: plain layout", 8, 0, 0},
        {0, 0, 15, 0, "[rot3] round 0 id mod 32: K slot = c ^ (e & 15), V slot ^ 16", 8, 0, 64 + 15 + (16 << 4)},
        {0, 0, 15, 0, "[rot3] round 0 id mod 32: K slot = c ^ (e & 3), V slot ^ 28", 8, 0, 64 + 3 + (28 << 4)},
        {0, 0, 15, 0, "[rot3] round 0 id mod 32: K slot = c ^ (e & 7), V slot ^ 24", 8, 0, 64 + 7 + (24 << 4)},
        {1, 0, 15, 0, "[rot3] round 0 XCD-rotated: plain layout", 8, 0, 0},
        {1, 0, 15, 0, "[rot3] round 0 XCD-rotated: K slot = c ^ (e & 15), V slot ^ 16", 8, 0, 64 + 15 + (16 << 4)},
        {1, 0, 15, 0, "[rot3] round 0 XCD-rotated: K slot = c ^ (e & 3), V slot ^ 28", 8, 0, 64 + 3 + (28 << 4)},
        {1, 0, 15, 0, "[rot3] round 0 XCD-rotated: K slot = c ^ (e & 7), V slot ^ 24", 8, 0, 64 + 7 + (24 << 4)},
        {0, 0, 15, 8 << 8, "[rot3] round 0 GQA pool: plain layout", 8, 0, 0},
        {0, 0, 15, 8 << 8, "[rot3] round 0 GQA pool: K slot = c ^ (e & 7)", 8, 0, 64 + 7},
        {0, 0, 15, 8 << 8, "[rot3] round 0 GQA pool: K slot = c ^ (e & 3), V slot ^ 4", 8, 0, 64 + 3 + (4 << 4)},
        {0, 0, 15, 0, "[rot3] round 1 id mod 32: plain layout", 8, 0, 0},
        {0, 0, 15, 0, "[rot3] round 1 id mod 32: K slot = c ^ (e & 15), V slot ^ 16", 8, 0, 64 + 15 + (16 << 4)},
        {0, 0, 15, 0, "[rot3] round 1 id mod 32: K slot = c ^ (e & 3), V slot ^ 28", 8, 0, 64 + 3 + (28 << 4)},
        {0, 0, 15, 0, "[rot3] round 1 id mod 32: K slot = c ^ (e & 7), V slot ^ 24", 8, 0, 64 + 7 + (24 << 4)},
        {1, 0, 15, 0, "[rot3] round 1 XCD-rotated: plain layout", 8, 0, 0},
        {1, 0, 15, 0, "[rot3] round 1 XCD-rotated: K slot = c ^ (e & 15), V slot ^ 16", 8, 0, 64 + 15 + (16 << 4)},
        {1, 0, 15, 0, "[rot3] round 1 XCD-rotated: K slot = c ^ (e & 3), V slot ^ 28", 8, 0, 64 + 3 + (28 << 4)},
        {1, 0, 15, 0, "[rot3] round 1 XCD-rotated: K slot = c ^ (e & 7), V slot ^ 24", 8, 0, 64 + 7 + (24 << 4)},
        {0, 0, 15, 8 << 8, "[rot3] round 1 GQA pool: plain layout", 8, 0, 0},
        {0, 0, 15, 8 << 8, "[rot3] round 1 GQA pool: K slot = c ^ (e & 7)", 8, 0, 64 + 7},
        {0, 0, 15, 8 << 8, "[rot3] round 1 GQA pool: K slot = c ^ (e & 3), V slot ^ 4", 8, 0, 64 + 3 + (4 << 4)},
        {0, 0, 15, 0, "[rot3] round 2 id mod 32: plain layout", 8, 0, 0},
        {0, 0, 15, 0, "[rot3] round 2 id mod 32: K slot = c ^ (e & 15), V slot ^ 16", 8, 0, 64 + 15 + (16 << 4)},
        {0, 0, 15, 0, "[rot3] round 2 id mod 32: K slot = c ^ (e & 3), V slot ^ 28", 8, 0, 64 + 3 + (28 << 4)},
        {0, 0, 15, 0, "[rot3] round 2 id mod 32: K slot = c ^ (e & 7), V slot ^ 24", 8, 0, 64 + 7 + (24 << 4)},
        {1, 0, 15, 0, "[rot3] round 2 XCD-rotated: plain layout", 8, 0, 0},
        {1, 0, 15, 0, "[rot3] round 2 XCD-rotated: K slot = c ^ (e & 15), V slot ^ 16", 8, 0, 64 + 15 + (16 << 4)},
        {1, 0, 15, 0, "[rot3] round 2 XCD-rotated: K slot = c ^ (e & 3), V slot ^ 28", 8, 0, 64 + 3 + (28 << 4)},
        {1, 0, 15, 0, "[rot3] round 2 XCD-rotated: K slot = c ^ (e & 7), V slot ^ 24", 8, 0, 64 + 7 + (24 << 4)},
        {0, 0, 15, 8 << 8, "[rot3] round 2 GQA pool: plain layout", 8, 0, 0},
        {0, 0, 15, 8 << 8, "[rot3] round 2 GQA pool: K slot = c ^ (e & 7)", 8, 0, 64 + 7},
        {0, 0, 15, 8 << 8, "[rot3] round 2 GQA pool: K slot = c ^ (e & 3), V slot ^ 4", 8, 0, 64 + 3 + (4 << 4)},
        {0, 0, 15, 8 << 8, "[rot] GQA pool (8 kv heads, 2 KiB rows): class = kv head of query head id mod 32, plain layout"},
        {0, 0, 15, 8 << 8, "[rot] GQA pool, slot = c ^ (e & 3)", 8, 0, 1},
        {0, 0, 15, 8 << 8, "[rot] GQA pool, slot = c ^ (e & 7)", 8, 0, 3},
        {0, 0, 15, 8 << 8, "[rot] GQA pool, K slot = c ^ 2e, V slot = c ^ 2e ^ 1", 8, 0, 7},
        {1, 0, 15, 8 << 8, "[rot] GQA pool, query heads rotated over the XCDs, plain layout"},
        {1, 0, 15, 8 << 8, "[rot] GQA pool, query heads rotated over the XCDs, slot = c ^ (e & 7)", 8, 0, 3},
        {0, 0, 15, 0, "NHD pieces, class = id mod 32"},
        {1, 0, 15, 0, "NHD pieces, classes rotated over the XCDs"},
        {1, 1, 15, 0, "  the same, base + 256 B"},
        {1, 2, 15, 0, "  the same, base + 512 B"},
        {1, 0, 2, 0, "  only classes = 1 mod 4"},
        {1, 0, 13, 0, "  only classes != 1 mod 4"},
        {0, 0, 15, 1, "HND tiles, class = id mod 32"},
        {1, 0, 15, 1, "HND tiles, classes rotated over the XCDs"},
        {100, 0, 15, 0, "NHD pieces, ALL workgroups on class 0"},
        {101, 0, 15, 0, "NHD pieces, ALL workgroups on class 1"},
        {102, 0, 15, 0, "NHD pieces, ALL workgroups on class 2"},
        {103, 0, 15, 0, "NHD pieces, ALL workgroups on class 3"},
        {105, 0, 15, 0, "NHD pieces, ALL workgroups on class 5"},
        {131, 0, 15, 0, "NHD pieces, ALL workgroups on class 31"},
        {200 + 0 * 32 + 1, 0, 15, 0, "two classes: 0 and 1"},
        {200 + 0 * 32 + 2, 0, 15, 0, "two classes: 0 and 2"},
        {200 + 2 * 32 + 3, 0, 15, 0, "two classes: 2 and 3"},
        {200 + 1 * 32 + 2, 0, 15, 0, "two classes: 1 and 2"},
        {200 + 1 * 32 + 3, 0, 15, 0, "two classes: 1 and 3"},
        {200 + 1 * 32 + 5, 0, 15, 0, "two classes: 1 and 5"},
        {200 + 1 * 32 + 31, 0, 15, 0, "two classes: 1 and 31"},
        {200 + 4 * 32 + 8, 0, 15, 0, "two classes: 4 and 8"},
        {1, 0, 15, 0, "NHD pieces, rotated, fast classes with 8 loads in flight per lane, slow ones with 16", 8, 2},
        {1, 0, 15, 0, "NHD pieces, rotated, every class with 8 loads in flight per lane", 8, 3},
        {1, 0, 15, 0, "NHD pieces, rotated, every class with 4 loads in flight per lane", 8, 4},
        {1, 0, 15, 0, "NHD pieces, rotated, 16 waves x 8 loads in flight per lane", 16, 3},
        {1, 0, 15, 0, "NHD pieces, rotated, 16 waves x 4 loads in flight per lane", 16, 4},
        {1, 0, 15, 1, "HND tiles, rotated, every class with 8 loads in flight per lane", 8, 3},
        {1, 0, 15, 0, "NHD pieces, rotated, PLAIN loads instead of nontemporal", 8, 1},
        {1, 0, 15, 0, "NHD pieces, rotated, 16-wave workgroups", 16, 0},
        {1, 0, 15, 0, "NHD pieces, rotated, 16-wave workgroups, plain loads", 16, 1},
        {1, 0, 15, 0, "NHD pieces, rotated, 4-wave workgroups", 4, 0},
    };
    unsigned host[512];
    for (const Case& cs : cases) {
        if (rot_only && cs.what[0] != '[') continue;
        if (rot2_only && cs.what[4] != argv[1][1]) continue;
        double dur[32] = {0}, cnt[32] = {0}, span = 0;
        double xcd_dur[8] = {0}, xcd_cnt[8] = {0};
        const int reps = 10;
        for (int rep = 0; rep < reps + 2; ++rep) {
#define LAUNCH(NW, KIND)                                                                                                   \
    hipLaunchKernelGGL((probe<NW, KIND>), dim3(256), dim3(NW * 64), 0, 0, buf + (size_t)cs.shift * 256, n_regions, cs.mode, \
                       cs.mask, (uint32_t)rep * 7919u, cs.hnd, cs.rot, stamps, sink)
            if (cs.nw == 8 && cs.kind == 0) LAUNCH(8, 0);
            else if (cs.nw == 8 && cs.kind == 2) LAUNCH(8, 2);
            else if (cs.nw == 8 && cs.kind == 3) LAUNCH(8, 3);
            else if (cs.nw == 8 && cs.kind == 4) LAUNCH(8, 4);
            else if (cs.nw == 16 && cs.kind == 3) LAUNCH(16, 3);
            else if (cs.nw == 16 && cs.kind == 4) LAUNCH(16, 4);
            else if (cs.nw == 8) LAUNCH(8, 1);
            else if (cs.nw == 16 && cs.kind == 0) LAUNCH(16, 0);
            else if (cs.nw == 16) LAUNCH(16, 1);
            else LAUNCH(4, 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(host, stamps, sizeof(host), hipMemcpyDeviceToHost));
            if (rep < 2) continue;
            unsigned t_min = 0xffffffffu, t_max = 0;
            for (int id = 0; id < 256; ++id) {
                if (host[2 * id] == 0 && host[2 * id + 1] == 0) continue;
                const uint32_t r = id / 32, c = cs.mode >= 200 ? ((id >> 3) & 1u) * 16 + id % 8 : cs.mode >= 100 ? id % 32 : cs.mode == 0 ? id % 32 : (id + 3 * r) % 32;
                const double d = (host[2 * id + 1] - host[2 * id]) * 0.01;
                dur[c] += d, cnt[c] += 1;
                xcd_dur[id % 8] += d, xcd_cnt[id % 8] += 1;
                t_min = host[2 * id] < t_min ? host[2 * id] : t_min;
                t_max = host[2 * id + 1] > t_max ? host[2 * id + 1] : t_max;
            }
            span += (t_max - t_min) * 0.01;
        }
        printf("%s: launch span %.1f us; mean workgroup duration (us) per class:\n   ", cs.what, span / reps);
        for (int c = 0; c < 32; ++c) printf(" %5.1f", cnt[c] ? dur[c] / cnt[c] : 0.0);
        printf("\n    per XCD:");
        for (int x = 0; x < 8; ++x) printf(" %5.1f", xcd_cnt[x] ? xcd_dur[x] / xcd_cnt[x] : 0.0);
        printf("\n");
    }
    return 0;
}
