// Measured HBM ceilings beside the 8 TB/s peak: a read-only stream (16-byte nontemporal loads, 16 in flight per
// lane -- the access shape of the estimate and gather kernels) over buffers far larger than the 256 MiB Infinity Cache,
// and a device-to-device hipMemcpy (read + write).
// Build: hipcc --offload-arch=gfx950 -O2 scripts/probe/hbm_ceiling.hip -o scripts/probe/hbm_ceiling
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            return 1;                                                                           \
        }                                                                                       \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every workgroup reads `rounds` x 64 KiB (256 threads x 16 loads x 16 B), workgroups interleaved over the buffer
__global__ __launch_bounds__(256) void read_stream(const u32x4* __restrict__ src, uint32_t rounds, uint32_t* sink) {
    u32x4 acc = (u32x4)(0u);
    for (uint32_t r = 0; r < rounds; ++r) {
        const u32x4* p = src + ((size_t)r * gridDim.x + blockIdx.x) * 4096 + threadIdx.x;
        u32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(p + i * 256);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
    CK(hipSetDevice(0));
    const size_t bytes = (size_t)8 << 30;  // 8 GiB
    u32x4* buf;
    uint32_t* sink;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const uint32_t grids[] = {512, 1024, 2048, 4096};
    for (uint32_t g : grids) {
        const uint32_t rounds = (uint32_t)(bytes / ((size_t)g * 65536));
        hipLaunchKernelGGL(read_stream, dim3(g), dim3(256), 0, 0, buf, rounds, sink);  // warm-up
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(read_stream, dim3(g), dim3(256), 0, 0, buf, rounds, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double gb = (double)g * rounds * 65536 / 1e9;
        printf("read stream, %4u workgroups x 256 threads, 16 x 16 B in flight per lane, %.2f GB: %.3f ms -> %.0f GB/s (%.0f %% of 8 TB/s)\n",
               g, gb, best, gb / (best * 1e-3), gb / (best * 1e-3) / 80.0);
    }
    // short bursts: what ONE launch of the size of the chain's kernels can reach (32 MiB and 256 MiB, cold: each repetition
    // reads a different part of the 8 GiB buffer)
    for (uint32_t mib : {32u, 64u, 256u}) {
        const uint32_t g = mib * 16;  // 64 KiB per workgroup, one round
        float tot = 0;
        const int reps = 16;
        for (int rep = 0; rep < reps; ++rep) {
            const u32x4* p = buf + (size_t)rep * (bytes / 16 / reps);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(read_stream, dim3(g), dim3(256), 0, 0, p, 1u, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            tot += ms;
        }
        const double gb = (double)mib * 1048576 / 1e9, ms = tot / reps;
        printf("one launch reading %3u MiB once (%u workgroups): %.2f us by events -> %.0f GB/s (%.0f %% of 8 TB/s)\n", mib, g,
               ms * 1e3, gb / (ms * 1e-3), gb / (ms * 1e-3) / 80.0);
    }
    // the bench's shape: 32 dependent launches back to back (one per layer, each on its own region), time per launch
    for (uint32_t mib : {32u, 64u, 256u}) {
        const uint32_t g = mib * 16, n = 32;
        const size_t step_vecs = (size_t)mib * 65536;  // region per launch, in 16-byte vectors
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0));
            for (uint32_t l = 0; l < n; ++l)
                hipLaunchKernelGGL(read_stream, dim3(g), dim3(256), 0, 0, buf + ((size_t)l * step_vecs) % (bytes / 16 - step_vecs), 1u, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double gb = (double)mib * 1048576 / 1e9, us = best * 1e3 / n;
        printf("32 launches back to back, %3u MiB each (%u workgroups): %.2f us per launch -> %.0f GB/s (%.0f %% of 8 TB/s)\n", mib, g, us,
               gb / (us * 1e-6), gb / (us * 1e-6) / 80.0);
    }
    // device-to-device copy
    u32x4* dst;
    const size_t half = bytes / 4;
    CK(hipMalloc(&dst, half));
    CK(hipMemcpy(dst, buf, half, hipMemcpyDeviceToDevice));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        CK(hipMemcpyAsync(dst, buf + (size_t)rep * (half / 16 / 8), half, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("hipMemcpy device-to-device %.2f GB: %.3f ms -> %.0f GB/s read + %.0f GB/s write\n", half / 1e9, best,
           half / 1e9 / (best * 1e-3), half / 1e9 / (best * 1e-3));
    return 0;
}
