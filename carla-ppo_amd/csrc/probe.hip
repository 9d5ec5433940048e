// probe.hip — box calibration behind the C ABI (round 5): what THIS GPU sustains right now on the two resources the rooflines are priced
// against, measured by the library itself so that a bench line can state next to every datasheet fraction the fraction of the box that produced it.
//   MFMA:  nothing but v_mfma_f32_32x32x16_bf16 on four independent accumulators, two waves per SIMD, one block per CU (the instruction every bf16 kernel of the
//          ConvVAE path issues); the shader clock it ran at = s_memtime ticks (shader cycles) / s_memrealtime ticks (100 MHz) of one wave.
//   HBM:   a pure 16-byte-per-lane streaming read of a buffer far larger than the 256 MB Infinity Cache, and a copy of its first half onto its second half.
// Synchronous (HIP events + hipEventSynchronize): a diagnostic for bench.py / tools, never on a training path.  No reference counterpart ("confirm on the box").
#include "common.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"

namespace {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe_mfma_kernel(float* sink, unsigned long long* ticks, int iters) {
    pbf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        const unsigned short ua = (unsigned short)(0x3f80 + ((threadIdx.x * 8 + e) * 37 % 64)), ub = (unsigned short)(0x3f80 + ((threadIdx.x * 8 + e + 5) * 29 % 64));
        a[e] = __builtin_bit_cast(__bf16, ua); b[e] = __builtin_bit_cast(__bf16, ub);
    }
    pf32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 123.456f) sink[0] = s;                       // (keeps the chain alive; never true)
    if (blockIdx.x == 0 && threadIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = r1 - r0; }
}

// every block walks the buffer together: block b takes every gridDim-th 16 KB piece (the pattern that reached the highest rate in tools/probes/hbm_stream_probe.hip)
__global__ __launch_bounds__(512) void probe_read_kernel(const pf32x4* __restrict__ src, float* sink, long long n16) {
    const long long piece = 1024;
    pf32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long q = blockIdx.x; (q + 1) * piece <= n16; q += gridDim.x) {
        const pf32x4* p = src + q * piece;
        acc += __builtin_nontemporal_load(p + threadIdx.x);
        acc += __builtin_nontemporal_load(p + threadIdx.x + 512);
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = 1.f;
}

__global__ __launch_bounds__(512) void probe_copy_kernel(const pf32x4* __restrict__ src, pf32x4* __restrict__ dst, long long n16) {
    const long long piece = 1024;
    for (long long q = blockIdx.x; (q + 1) * piece <= n16; q += gridDim.x) {
        const pf32x4 v0 = src[q * piece + threadIdx.x], v1 = src[q * piece + threadIdx.x + 512];
        dst[q * piece + threadIdx.x] = v0; dst[q * piece + threadIdx.x + 512] = v1;
    }
}

struct EvPair {
    hipEvent_t a = nullptr, b = nullptr;
    bool ok() { return hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess; }
    ~EvPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};

}  // namespace

extern "C" {

long long mi_device_probe_scratch_bytes(void) { return 1536ll << 20; }     // what mi_device_probe would like (6 x the Infinity Cache); it works from 64 MiB upwards

// out[8]: 0 sustained bf16 MFMA rate, TFLOP/s (mean of the last two of three `millis`-long launches)      1 shader clock during it, MHz
//         2 the FIRST launch's rate (clocks as the caller left them)                                      3 shader clock during the first launch, MHz
//         4 HBM streaming read, TB/s (best of three)      5 HBM copy, TB/s of read + written bytes (best of three)      6 compute units      7 bytes the HBM probes walked
int mi_device_probe(void* stream, void* scratch, long long scratch_bytes, int millis, float* out8) {
    if (!out8 || !scratch || scratch_bytes < (64ll << 20)) return mi_fail(MI_ERR_ARG, "mi_device_probe: needs out8 and a device scratch buffer of at least 64 MiB");
    if (((uintptr_t)scratch) & 255) return mi_fail(MI_ERR_ARG, "mi_device_probe: scratch must be 256-byte aligned");
    if (millis < 1) millis = 1;
    if (millis > 50) millis = 50;
    hipStream_t st = (hipStream_t)stream;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return mi_fail(MI_ERR_STATE, "mi_device_probe: no device");
    const int cus = prop.multiProcessorCount;
    EvPair ev;
    if (!ev.ok()) return mi_fail(MI_ERR_STATE, "mi_device_probe: hipEventCreate failed");
    float* sink = (float*)scratch;
    unsigned long long* ticks = (unsigned long long*)((char*)scratch + 256);
    for (int i = 0; i < 8; ++i) out8[i] = 0.f;
    out8[6] = (float)cus;
    // ---- MFMA: one block per CU, 2 waves per SIMD, 4 accumulators; one MFMA per SIMD per 32 cycles at full rate: iterations for `millis` at 2.4 GHz ----
    const int iters = (int)((double)millis * 1e-3 * 2.4e9 / 32.0 / 8.0);
    double rate[3] = {0, 0, 0}, mhz[3] = {0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(ev.a, st);
        hipLaunchKernelGGL(probe_mfma_kernel, dim3(cus), dim3(512), 0, st, sink, ticks, iters);
        (void)hipEventRecord(ev.b, st);
        if (hipEventSynchronize(ev.b) != hipSuccess) return mi_check_launch("mi_device_probe: mfma probe");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev.a, ev.b);
        unsigned long long t[2] = {0, 0};
        if (hipMemcpy(t, ticks, sizeof(t), hipMemcpyDeviceToHost) != hipSuccess) return mi_fail(MI_ERR_STATE, "mi_device_probe: copy failed");
        const double flops = (double)cus * 8.0 /* waves */ * 4.0 /* accumulators */ * (double)iters * 32768.0;
        rate[rep] = ms > 0.f ? flops / (ms * 1e-3) / 1e12 : 0.0;
        mhz[rep] = t[1] > 0 ? (double)t[0] / (double)t[1] * 100.0 : 0.0;
    }
    out8[0] = (float)(0.5 * (rate[1] + rate[2])); out8[1] = (float)(0.5 * (mhz[1] + mhz[2]));
    out8[2] = (float)rate[0]; out8[3] = (float)mhz[0];
    // ---- HBM ----
    const long long usable = (scratch_bytes - 4096) / (32ll << 10) * (32ll << 10);         // (the first 4 KB hold the sink and the tick words)
    const pf32x4* base = (const pf32x4*)((char*)scratch + 4096);
    const long long n16 = usable / 16, half16 = n16 / 2 / 1024 * 1024;
    out8[7] = (float)usable;
    double best_r = 0.0, best_c = 0.0;
    for (int rep = 0; rep < 3; ++rep) {
        float ms = 0.f;
        (void)hipEventRecord(ev.a, st);
        hipLaunchKernelGGL(probe_read_kernel, dim3(cus * 8), dim3(512), 0, st, base, sink, n16);
        (void)hipEventRecord(ev.b, st);
        if (hipEventSynchronize(ev.b) != hipSuccess) return mi_check_launch("mi_device_probe: read probe");
        (void)hipEventElapsedTime(&ms, ev.a, ev.b);
        if (ms > 0.f && (double)n16 * 16.0 / (ms * 1e-3) / 1e12 > best_r) best_r = (double)n16 * 16.0 / (ms * 1e-3) / 1e12;
        (void)hipEventRecord(ev.a, st);
        hipLaunchKernelGGL(probe_copy_kernel, dim3(cus * 8), dim3(512), 0, st, base, (pf32x4*)base + half16, half16);
        (void)hipEventRecord(ev.b, st);
        if (hipEventSynchronize(ev.b) != hipSuccess) return mi_check_launch("mi_device_probe: copy probe");
        (void)hipEventElapsedTime(&ms, ev.a, ev.b);
        if (ms > 0.f && (double)half16 * 32.0 / (ms * 1e-3) / 1e12 > best_c) best_c = (double)half16 * 32.0 / (ms * 1e-3) / 1e12;
    }
    out8[4] = (float)best_r; out8[5] = (float)best_c;
    return mi_check_launch("mi_device_probe");
}

}  // extern "C"
