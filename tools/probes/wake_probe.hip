// wake_probe.hip — how long after the PRODUCING kernel retires does a kernel on ANOTHER, idle queue start, by hand-over mechanism (round 6: in the ConvVAE step's timeline a
// wait that is already satisfied when the queue reaches it costs ~6 us, one that is reached BEFORE its event fires ~13 us: the filter-gradient queue at the start of the backward
// pass, the optimiser launch behind the join).  The consumer queue is idle and waiting; producer kernel = 256 blocks spinning `pus` microseconds; every block's last wave stamps
// wall_clock64() (100 MHz) at its end, the consumer's first block at its start: latency = consumer start - max producer end.
//   mode 1  hipEventRecord + hipStreamWaitEvent      mode 2  stop event on the producer's dispatch packet + hipStreamWaitEvent
//   mode 3  hipStreamWriteValue32 + hipStreamWaitValue32      mode 4 / 5  flag written by the producer kernel's last block (signal / device memory) + hipStreamWaitValue32
//   mode 6  same queue (no hand-over): the floor
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/wake_probe.hip -o /tmp/wake_probe ; run: /tmp/wake_probe [producer_us]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(256) void producer_kernel(long long ticks, unsigned long long* end_max, unsigned* flag, unsigned* counter, unsigned val) {
    const long long t0 = wall_clock64();
    long long t = t0;
    while (t - t0 < ticks) t = wall_clock64();
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(end_max, (unsigned long long)wall_clock64());
        if (flag) {
            __threadfence();
            const unsigned done = atomicAdd(counter, 1u);
            if (done == gridDim.x - 1) {
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
__global__ __launch_bounds__(256) void consumer_kernel(unsigned long long* start_min) {
    if (threadIdx.x == 0) atomicMin(start_min, (unsigned long long)wall_clock64());
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const double pus = argc > 1 ? atof(argv[1]) : 60.0;
    int can = 0; hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    unsigned *sig = nullptr, *plain = nullptr, *counter = nullptr; unsigned long long* stamps = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory)); CK(hipMalloc(&plain, 256)); CK(hipMalloc(&counter, 256)); CK(hipMalloc(&stamps, 256));
    CK(hipMemset(plain, 0, 256)); CK(hipMemset(counter, 0, 256)); CK(hipMemset(sig, 0, 8));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const long long pt = (long long)(pus * 100.0);
    unsigned seq = 0;
    const char* names[7] = {"", "hipEventRecord + hipStreamWaitEvent", "stop event on the producer's packet + hipStreamWaitEvent", "hipStreamWriteValue32 + hipStreamWaitValue32",
                            "flag from the producer kernel + WaitValue32 (signal memory)", "flag from the producer kernel + WaitValue32 (device memory)", "same queue (floor)"};
    printf("producer %.0f us on queue a, consumer queue idle and waiting; latency = first consumer block's start - last producer block's end (us, 100 MHz clock), 15 runs\n", pus);
    for (int mode = 1; mode <= 6; ++mode) {
        if ((mode >= 3 && mode <= 5) && !can) continue;
        std::vector<double> v;
        for (int r = 0; r < 16; ++r) {
            ++seq;
            const unsigned long long init[2] = {0ull, ~0ull};
            CK(hipMemcpy(stamps, init, 16, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            unsigned* f = mode == 5 ? plain : sig;
            if (mode == 2) hipExtLaunchKernelGGL(producer_kernel, dim3(256), dim3(256), 0, a, nullptr, ev, 0, pt, stamps, (unsigned*)nullptr, counter, 0u);
            else if (mode == 4 || mode == 5) hipLaunchKernelGGL(producer_kernel, dim3(256), dim3(256), 0, a, pt, stamps, f, counter, seq);
            else hipLaunchKernelGGL(producer_kernel, dim3(256), dim3(256), 0, a, pt, stamps, (unsigned*)nullptr, counter, 0u);
            if (mode == 1) CK(hipEventRecord(ev, a));
            if (mode == 3) CK(hipStreamWriteValue32(a, sig, seq, 0));
            hipStream_t cq = mode == 6 ? a : b;
            if (mode == 1 || mode == 2) CK(hipStreamWaitEvent(b, ev, 0));
            if (mode >= 3 && mode <= 5) CK(hipStreamWaitValue32(b, f, seq, hipStreamWaitValueGte, 0xffffffffu));
            hipLaunchKernelGGL(consumer_kernel, dim3(256), dim3(256), 0, cq, stamps + 1);
            CK(hipDeviceSynchronize());
            unsigned long long out[2]; CK(hipMemcpy(out, stamps, 16, hipMemcpyDeviceToHost));
            if (r > 0) v.push_back(((double)out[1] - (double)out[0]) / 100.0);
        }
        std::sort(v.begin(), v.end());
        printf("  mode %d %-62s median %6.2f  min %6.2f  max %6.2f\n", mode, names[mode], v[v.size() / 2], v.front(), v.back());
    }
    return 0;
}
