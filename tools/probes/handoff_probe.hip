// handoff_probe.hip — what one cross-queue hand-over costs the PRODUCING queue and the CONSUMING queue, by mechanism (round 6, VERDICT r05 weak 3:
// the two-queue backward pass of the ConvVAE step shows ~5 us of bubble behind every kernel of the caller's queue that hands a tensor to the filter-gradient queue).
//   mode 0  no hand-over at all (both queues run their kernels back to back)
//   mode 1  hipEventRecord behind the producer, hipStreamWaitEvent in front of the consumer                   (round 3 form)
//   mode 2  the producer's own dispatch packet carries the event (hipExtLaunchKernelGGL stop event)             (round 4/5 form, MI_LAUNCH)
//   mode 3  hipStreamWriteValue32 behind the producer, hipStreamWaitValue32 (>=) in front of the consumer
//   mode 4  the producer KERNEL writes the flag (last block out: fence, counter, flag), hipStreamWaitValue32 in front of the consumer: nothing on the producing queue
//   mode 5  as 4, the flag in ordinary device memory instead of signal memory
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/handoff_probe.hip -o /tmp/handoff_probe ; run: /tmp/handoff_probe [n] [producer_us] [consumer_us]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void spin_kernel(long long ticks, unsigned* flag, unsigned* counter, unsigned val, float* sink) {
    const long long t0 = wall_clock64();
    long long t = t0;
    float s = 0.f;
    while (t - t0 < ticks) { s += 1.0f; t = wall_clock64(); }
    if (sink && s < 0.f) sink[threadIdx.x] = s;
    if (flag) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned done = atomicAdd(counter, 1u);
            if (done == gridDim.x - 1) {
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 40;
    const double pus = argc > 2 ? atof(argv[2]) : 30.0, cus = argc > 3 ? atof(argv[3]) : 15.0;
    int can = 0; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned *sig = nullptr, *sig2 = nullptr, *plain = nullptr, *counter = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory)); CK(hipExtMallocWithFlags((void**)&sig2, 8, hipMallocSignalMemory));
    CK(hipMalloc(&plain, 256)); CK(hipMalloc(&counter, 256)); CK(hipMemset(plain, 0, 256)); CK(hipMemset(counter, 0, 256));
    CK(hipMemset(sig, 0, 8)); CK(hipMemset(sig2, 0, 8));
    // ticks of wall_clock64() per microsecond (s_memrealtime: constant 100 MHz)
    const double tick_per_us = 100.0;
    const long long pt = (long long)(pus * tick_per_us), ct = (long long)(cus * tick_per_us);
    hipEvent_t e0, e1, e2, hand[2][64], done;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    for (int k = 0; k < 2; ++k) for (int i = 0; i < 64; ++i) CK(hipEventCreateWithFlags(&hand[k][i], hipEventDisableTiming));
    const dim3 grid(256), block(256);
    unsigned seq = 0;
    auto run = [&](int mode, bool consumer_long) -> std::pair<double, double> {
        const long long P = consumer_long ? ct : pt, C = consumer_long ? pt : ct;      // consumer_long: the consuming queue is the bottleneck
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, a));
        CK(hipEventRecord(done, a)); CK(hipStreamWaitEvent(b, done, 0));               // both queues start together
        for (int i = 0; i < n; ++i) {
            ++seq;
            unsigned* f = mode == 5 ? plain : sig;
            if (mode == 2) hipExtLaunchKernelGGL(spin_kernel, grid, block, 0, a, nullptr, hand[0][i % 64], 0, P, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (float*)nullptr);
            else if (mode == 4 || mode == 5) hipLaunchKernelGGL(spin_kernel, grid, block, 0, a, P, f, counter, seq, (float*)nullptr);
            else hipLaunchKernelGGL(spin_kernel, grid, block, 0, a, P, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (float*)nullptr);
            if (mode == 1) CK(hipEventRecord(hand[0][i % 64], a));
            if (mode == 3) CK(hipStreamWriteValue32(a, sig, seq, 0));
            if (mode == 1 || mode == 2) CK(hipStreamWaitEvent(b, hand[0][i % 64], 0));
            if (mode >= 3) CK(hipStreamWaitValue32(b, f, seq, hipStreamWaitValueGte, 0xffffffffu));
            hipLaunchKernelGGL(spin_kernel, grid, block, 0, b, C, (unsigned*)nullptr, (unsigned*)nullptr, 0u, (float*)nullptr);
        }
        CK(hipEventRecord(e1, a));
        CK(hipEventRecord(e2, b));
        CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(e2));
        float ma, mb; CK(hipEventElapsedTime(&ma, e0, e1)); CK(hipEventElapsedTime(&mb, e0, e2));
        return {ma * 1e3 / n, mb * 1e3 / n};
    };
    const char* names[6] = {"no hand-over", "hipEventRecord + hipStreamWaitEvent", "stop event on the producer's packet + WaitEvent", "hipStreamWriteValue32 + hipStreamWaitValue32",
                            "flag written by the producer kernel + WaitValue32 (signal memory)", "flag written by the producer kernel + WaitValue32 (device memory)"};
    for (int cl = 0; cl < 2; ++cl) {
        printf("\n%s: producer kernels %.0f us, consumer kernels %.0f us, %d hand-overs, us per kernel pair on the producing / consuming queue (medians of 7)\n",
               cl ? "CONSUMING queue is the long one" : "PRODUCING queue is the long one", cl ? cus : pus, cl ? pus : cus, n);
        for (int mode = 0; mode < 6; ++mode) {
            if (mode >= 3 && !can) continue;
            std::vector<double> va, vb;
            run(mode, cl);
            for (int r = 0; r < 7; ++r) { auto p = run(mode, cl); va.push_back(p.first); vb.push_back(p.second); }
            std::sort(va.begin(), va.end()); std::sort(vb.begin(), vb.end());
            printf("  mode %d %-82s producing %.2f  consuming %.2f\n", mode, names[mode], va[3], vb[3]);
        }
    }
    return 0;
}
