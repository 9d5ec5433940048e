// hbm_stream_probe.hip — what a pure streaming READ of a buffer far larger than the 256 MB last-level cache achieves on this box, for the two
// ways a kernel can deal a tensor to its blocks: CHUNKED (block b streams its own contiguous 1/nblocks of the buffer: what the persistent conv /
// filter-gradient kernels do) and INTERLEAVED (all blocks walk the buffer together, block b takes every nblocks-th 16 KB piece).
// 16-byte loads, 8 per lane in flight, nothing else in the kernel.   hipcc --offload-arch=gfx950 -O3 hbm_stream_probe.hip -o hbm_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const f32x4* __restrict__ a, float* sink, size_t n16) {       // n16: 16-byte elements
    const size_t nb = gridDim.x, per = n16 / nb;                                                          // elements per block
    f32x4 acc = {0, 0, 0, 0};
    if (MODE == 0) {
        const f32x4* p = a + blockIdx.x * per;
        for (size_t i = threadIdx.x; i + 7 * 512 < per; i += 8 * 512) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += p[i + j * 512];
        }
    } else {
        const size_t piece = 1024;                                                                          // 16 KB
        for (size_t q = blockIdx.x; (q + 1) * piece <= n16; q += nb) {
            const f32x4* p = a + q * piece;
            acc += p[threadIdx.x]; acc += p[threadIdx.x + 512];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = 1.f;
}

int main() {
    const size_t bytes = 3ull << 30;
    f32x4* d; float* sink;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, bytes); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024, 2048}) {
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d, sink, bytes / 16);
                else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d, sink, bytes / 16);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%-11s %4d blocks x 512 threads: 3 GiB in %.3f ms = %.2f TB/s\n", mode == 0 ? "chunked" : "interleaved", blocks, best, bytes / best / 1e9);
        }
    }
    return 0;
}
