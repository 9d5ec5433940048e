// mfma_probe.hip — issue rate of v_mfma_f32_32x32x16_bf16 on one SIMD: NACC independent accumulators per wave, WPS waves per SIMD, operands in VGPRs.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_probe.hip -o tools/probes/mfma_probe ; prints ns and cycles (at 2.4 GHz) per MFMA per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(512) void probe(float* out, int iters, const unsigned short* in) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = __builtin_bit_cast(__bf16, in[(threadIdx.x * 8 + e) & 1023]); b[e] = __builtin_bit_cast(__bf16, in[(threadIdx.x * 8 + e + 512) & 1023]); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC> void run(int threads, const unsigned short* in, float* out, const char* what) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NACC><<<256, threads>>>(out, 10, in);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NACC><<<256, threads>>>(out, iters, in);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * NACC * (threads / 256);
    printf("%-28s NACC=%d waves/SIMD=%d: %.1f us, %.2f ns per MFMA per SIMD = %.1f cycles at 2.4 GHz\n", what, NACC, threads / 256, ms * 1e3, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
}

int main() {
    unsigned short h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (unsigned short)(0x3f80 + (i * 37 % 64));      // bf16 values in [1, 1.5)
    unsigned short* in; float* out;
    hipMalloc(&in, sizeof(h)); hipMalloc(&out, 256 * 512 * 4);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<1>(256, in, out, "dependent chain");
    run<2>(256, in, out, "2 accumulators");
    run<3>(256, in, out, "3 accumulators");
    run<4>(256, in, out, "4 accumulators");
    run<6>(256, in, out, "6 accumulators");
    run<9>(256, in, out, "9 accumulators");
    run<3>(512, in, out, "3 accumulators, 2 waves");
    run<1>(512, in, out, "dependent chain, 2 waves");
    return 0;
}
