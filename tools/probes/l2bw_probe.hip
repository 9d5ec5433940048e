// l2bw_probe.hip — per-CU L2-hit load bandwidth on gfx950 for the access shapes the conv kernels use:
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), mode 1: plain buffer_load_dwordx4 to VGPRs.
//   Each block re-reads its own private `foot` KB region (L2 resident), NJ 1-KB wave-loads in flight per wave per round.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NJ>
__global__ __launch_bounds__(512) void k(const unsigned char* a, float* sink, int foot_bytes, int rounds, int line_stride) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[8 * NJ * 1024];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0x3fffffff, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // 8 lanes read one 128-B line; the 8 lines of a wave-load are line_stride lines apart (1 = contiguous KB, 2 = stride-2 pixels ...)
    const uint32_t base = (uint32_t)blockIdx.x * (uint32_t)foot_bytes;
    uint32_t pos = (uint32_t)(wave * NJ) * 1024u;
    f32x4 accv = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            uint32_t off = (pos + j * 1024u + (uint32_t)(lane >> 3) * 128u * line_stride + (lane & 7) * 16u) % (uint32_t)foot_bytes;
            if (MODE == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr)(lds + (wave * NJ + j) * 1024), 16, (int)(base + off), 0, 0, 0);
            else { f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(base + off), 0, 0)); accv += v; }
        }
        pos += (uint32_t)(nw * NJ) * 1024u;
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 0) { __syncthreads(); accv[0] = ((float*)lds)[threadIdx.x]; }
    if (accv[0] + accv[1] + accv[2] + accv[3] == 123.456f) sink[0] = 1.f;
}

template <int MODE, int NJ>
void run(const unsigned char* d, float* sink, int blocks, int threads, int foot_kb, int stride) {
    const int rounds = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NJ>), dim3(blocks), dim3(threads), 0, 0, d, sink, foot_kb * 1024, 20, stride);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NJ>), dim3(blocks), dim3(threads), 0, 0, d, sink, foot_kb * 1024, rounds, stride);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * (threads / 64) * NJ * 1024.0 * rounds;
    printf("mode=%d NJ=%d blocks=%d threads=%d foot=%dKB stride=%d : %.2f TB/s total, %.1f GB/s per CU (256), %.1f B/clk/CU @2.4GHz\n", MODE, NJ, blocks, threads, foot_kb,
           stride, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4);
}

int main() {
    unsigned char* d; float* sink;
    const size_t total = 1024ull * 256 * 1024;
    hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&sink, 64);
    for (int stride = 1; stride <= 2; ++stride) {
        run<0, 2>(d, sink, 256, 256, 64, stride);
        run<0, 4>(d, sink, 256, 256, 64, stride);
        run<0, 8>(d, sink, 256, 256, 64, stride);
        run<0, 8>(d, sink, 512, 256, 64, stride);
        run<0, 4>(d, sink, 256, 512, 64, stride);
        run<0, 8>(d, sink, 256, 512, 64, stride);
        run<0, 16>(d, sink, 256, 512, 64, stride);
        run<1, 4>(d, sink, 256, 256, 64, stride);
        run<1, 8>(d, sink, 256, 256, 64, stride);
        run<1, 8>(d, sink, 512, 256, 64, stride);
        run<1, 8>(d, sink, 256, 512, 64, stride);
        run<1, 8>(d, sink, 1024, 256, 64, stride);
    }
    // L1-resident (16 KB footprint) and streaming from beyond L2 (big footprint)
    run<0, 8>(d, sink, 256, 512, 16, 1);
    run<1, 8>(d, sink, 256, 512, 16, 1);
    run<0, 8>(d, sink, 256, 512, 1024, 1);
    run<1, 8>(d, sink, 256, 512, 1024, 1);
    return 0;
}
