// Probe of two hardware facts the kernels rely on, run on the GPU box:
//   1. workgroup -> XCD placement: which XCC (HW_REG_XCC_ID) runs linear workgroup id i, for 1-D / 2-D / 3-D grids;
//   2. the latency of a producer -> consumer hand-off through memory between two workgroups on the SAME XCD and on
//      DIFFERENT XCDs, for the recipes of the chained launch (release/acquire fences; write-through stores + relaxed
//      counter) and for an XCD-local recipe (plain stores, waited for, + counter; consumer invalidates its L1 only).
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probe/xcd_probe.hip -o scripts/probe/xcd_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xfu;
}

__global__ void where_kernel(uint32_t* xcc) {
    if (threadIdx.x == 0) {
        const uint32_t lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        xcc[lin] = xcc_id();
    }
}

// mode 0: release fetch-add / acquire fence (agent scope)      -- the textbook hand-off
// mode 1: write-through stores (sc1) + s_waitcnt + relaxed add / relaxed poll + acquire fence  -- chained launch
// mode 2: plain stores + s_waitcnt + relaxed add / relaxed poll + L1-only invalidate (no L2 write-back, no L2
//         invalidate)  -- valid on the SAME XCD only (its L2 is shared)
// Workgroup `prod` writes 1 KiB of payload then bumps the flag; workgroup `cons` stamps the time, polls, reads the
// payload, stamps again.  out[0] = consumer wait in 10 ns ticks measured from the PRODUCER's start stamp (shared
// clock: s_memrealtime), out[1] = checksum error count, out[2..3] = the two XCC ids.
__global__ void handoff_kernel(int mode, uint32_t prod, uint32_t cons, uint32_t* flag, uint32_t* payload, uint32_t token,
                               long long* out, const u32x4* stream, uint32_t stream_vecs, uint32_t* sink) {
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    if (b >= 16) {  // background: every other workgroup streams 64 KiB x 4 with 16 loads in flight per lane (HBM under load)
        u32x4 acc = (u32x4)(0u);
        for (uint32_t r = 0; r < 4; ++r) {
            u32x4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                v[i] = __builtin_nontemporal_load(stream + ((size_t)(b - 16) * 4 + r) * 4096 % stream_vecs + i * 256 + t);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc ^= v[i];
        }
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
        return;
    }
    if (b == prod) {
        // let the consumer get into its poll loop first
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 300) {}
        const long long start = wall_clock64();
        if (mode == 1)
            __hip_atomic_store(payload + t, token + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
            payload[t] = token + t;
        if (mode != 0) __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (t == 0) {
            if (mode == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else if (mode == 1) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            out[4] = start;
            out[2] = xcc_id();
        }
    } else if (b == cons) {
        __shared__ int bad;
        if (t == 0) {
            bad = 0;
            const long long t0 = wall_clock64();
            if (mode == 2) {  // (a workgroup-scope poll is served by this CU's L1 and never sees the flag: measured)
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < 1000000)
                    __builtin_amdgcn_s_sleep(1);
                asm volatile("buffer_inv sc0" ::: "memory");  // this CU's L1 only; the payload sits in the shared L2
            } else {
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < 1000000)
                    __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
        const uint32_t v = payload[t];
        if (v != token + t) atomicAdd(&bad, 1);
        __syncthreads();
        if (t == 0) {
            out[0] = wall_clock64();
            out[1] = bad;
            out[3] = xcc_id();
        }
    }
}

// 3. does a page that ANOTHER workgroup (same or different XCD) read a few microseconds ago come back faster -- from that
// XCD's L2 or from the memory-side Infinity Cache -- while the HBM is saturated?  Workgroup `first` reads 64 KiB (8 KiB
// per wave, the shape of an attention workgroup's gather) at t0; workgroup `second` reads 64 KiB `delay` ticks later:
// the same bytes (warm) or bytes nobody touched (cold).  out[0] = the second read's duration in ticks.
__global__ void reread_kernel(uint32_t first, uint32_t second, const u32x4* region, const u32x4* other, long long delay, long long* out,
                              const u32x4* stream, uint32_t stream_vecs, uint32_t* sink, long long* t0_shared) {
    const uint32_t b = blockIdx.x, t = threadIdx.x;
    if (b >= 16) {
        u32x4 acc = (u32x4)(0u);
        for (uint32_t r = 0; r < 4; ++r) {
            u32x4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                v[i] = __builtin_nontemporal_load(stream + ((size_t)(b - 16) * 4 + r) * 4096 % stream_vecs + i * 256 + t);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc ^= v[i];
        }
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
        return;
    }
    if (b != first && b != second) return;
    // both wait until the background is streaming
    const long long start = wall_clock64();
    while (wall_clock64() - start < (b == first ? 300 : 300 + delay)) {}
    const u32x4* src = (b == first) ? region : other;  // `other` == region for the warm case
    const long long a = wall_clock64();
    u32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(src + i * 256 + t);
    u32x4 acc = (u32x4)(0u);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc ^= v[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[1] = 1;
    __syncthreads();
    if (t == 0) out[b == first ? 1 : 0] = wall_clock64() - a;
}

int main() {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    printf("device: %s, %d CUs\n", prop.name, prop.multiProcessorCount);

    // ---- 1. placement
    uint32_t* d_x;
    CK(hipMalloc(&d_x, 65536 * sizeof(uint32_t)));
    struct G { dim3 g; const char* what; } grids[] = {
        {dim3(64), "1-D 64"}, {dim3(16, 32), "2-D 16 x 32 (chunks x heads, one sequence)"},
        {dim3(1, 32, 8), "3-D 1 x 32 x 8 (batched: heads x sequences)"}, {dim3(2, 32, 8), "3-D 2 x 32 x 8"}, {dim3(1061), "1-D 1061"}};
    for (auto& gr : grids) {
        const uint32_t n = gr.g.x * gr.g.y * gr.g.z;
        CK(hipMemset(d_x, 0xff, n * sizeof(uint32_t)));
        hipLaunchKernelGGL(where_kernel, gr.g, dim3(512), 0, 0, d_x);
        CK(hipDeviceSynchronize());
        std::vector<uint32_t> h(n);
        CK(hipMemcpy(h.data(), d_x, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint32_t base = h[0], mism = 0;
        for (uint32_t i = 0; i < n; ++i) mism += h[i] != (base + i) % 8;
        printf("grid %-48s first 16 XCC ids:", gr.what);
        for (uint32_t i = 0; i < 16 && i < n; ++i) printf(" %u", h[i]);
        printf("   | workgroups off the (first + linear id) mod 8 rule: %u of %u\n", mism, n);
    }

    // ---- 2. hand-off latency
    uint32_t *d_flag, *d_pay;
    long long* d_out;
    CK(hipMalloc(&d_flag, 4096));
    CK(hipMalloc(&d_pay, 4096));
    CK(hipMalloc(&d_out, 64));
    const char* names[] = {"release add / acquire (agent)", "write-through stores + relaxed counter (chained launch)",
                           "plain stores + counter, L1-only invalidate (no L2 write-back)"};
    u32x4* d_stream;
    const uint32_t stream_vecs = 64u << 20;  // 1 GiB of uint4: larger than the Infinity Cache
    CK(hipMalloc(&d_stream, (size_t)stream_vecs * 16));
    CK(hipMemset(d_stream, 1, (size_t)stream_vecs * 16));
    for (int loaded = 0; loaded < 2; ++loaded)
    for (int mode = 0; mode < 3; ++mode)
        for (int same = 1; same >= 0; --same) {
            const uint32_t prod = 0, cons = same ? 8 : 1;  // linear ids 0 and 8 share an XCD, 0 and 1 do not
            std::vector<double> lat;
            long long errs = 0, x0 = -1, x1 = -1;
            for (int rep = 0; rep < 40; ++rep) {
                CK(hipMemset(d_flag, 0, 4096));
                CK(hipMemset(d_pay, 0, 4096));
                CK(hipMemset(d_out, 0, 64));
                hipLaunchKernelGGL(handoff_kernel, dim3(16 + (loaded ? 2048 : 0)), dim3(256), 0, 0, mode, prod, cons, d_flag, d_pay,
                                   1000u * (rep + 1), d_out, d_stream, stream_vecs, d_flag + 512);
                CK(hipDeviceSynchronize());
                long long o[8];
                CK(hipMemcpy(o, d_out, 64, hipMemcpyDeviceToHost));
                if (rep >= 8) lat.push_back((double)(o[0] - o[4]) / 100.0);
                errs += o[1];
                x0 = o[2], x1 = o[3];
            }
            double s = 0, mn = 1e9;
            for (double v : lat) s += v, mn = v < mn ? v : mn;
            printf("%s hand-off %-58s %s XCD (xcc %lld -> %lld): producer's first store -> consumer has the data: mean %.2f us, min %.2f us, payload errors %lld%s\n",
                   loaded ? "[HBM loaded]" : "[idle]      ", names[mode], same ? "same     " : "different", x0, x1, s / lat.size(), mn, errs,
                   (mode == 2 && !same) ? "  (not a valid recipe across XCDs: shown for contrast)" : "");
        }
    // ---- 3. re-read latency
    long long* d_t0;
    CK(hipMalloc(&d_t0, 64));
    for (int loaded = 0; loaded < 2; ++loaded)
        for (int warm = 1; warm >= 0; --warm)
            for (int same = 1; same >= 0; --same) {
                const uint32_t first = 0, second = same ? 8 : 1;
                double s1 = 0, s2 = 0;
                int n = 0;
                for (int rep = 0; rep < 24; ++rep) {
                    // a fresh region every repetition, far from the background stream's blocks of this launch
                    const u32x4* region = d_stream + ((size_t)(40u << 20) + (size_t)rep * 65536);
                    const u32x4* other = warm ? region : region + 32768;
                    CK(hipMemset(d_out, 0, 64));
                    hipLaunchKernelGGL(reread_kernel, dim3(16 + (loaded ? 2048 : 0)), dim3(256), 0, 0, first, second, region, other, 300LL, d_out,
                                       d_stream, stream_vecs, d_flag + 512, d_t0);
                    CK(hipDeviceSynchronize());
                    long long o[8];
                    CK(hipMemcpy(o, d_out, 64, hipMemcpyDeviceToHost));
                    if (rep >= 4) s1 += o[1] / 100.0, s2 += o[0] / 100.0, ++n;
                }
                printf("%s 64 KiB read, 3 us after %s workgroup on %s XCD read %s: first read %.2f us, second read %.2f us\n",
                       loaded ? "[HBM loaded]" : "[idle]      ", "a", same ? "the same     " : "a different  ", warm ? "the SAME bytes " : "other bytes    ", s1 / n, s2 / n);
            }
    return 0;
}
