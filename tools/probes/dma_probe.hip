// dma_probe.hip — checks the buffer_load ... lds (LDS-DMA) semantics gemm2 relies on, on real gfx950:
//  (1) lane i of a wave lands at M0-base + 16*i; (2) out-of-range voffset writes ZEROS to LDS; (3) soffset is added, not range-checked.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/dma_probe tools/probes/dma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
__global__ void k(const uint32_t* a, uint32_t* out, uint32_t nbytes, int soff) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 1024 + 64];
    for (int i = threadIdx.x; i < (4 * 1024 + 64) / 4; i += blockDim.x) ((uint32_t*)lds)[i] = 0xDEADBEEFu;
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, nbytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave w: lanes load 16 B at byte offset (w*64 + (63-lane))*16 (reversed within the wave) ; odd lanes of wave 1 are forced OOB
    uint32_t voff = (uint32_t)((wave * 64 + (63 - lane)) * 16);
    if (wave == 1 && (lane & 1)) voff = 0x40000000u;
    if (wave == 2 && (lane & 1)) voff = 0x80000000u + voff;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(lds + wave * 1024), 16, voff, wave == 3 ? soff : 0, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = ((uint32_t*)lds)[i];
}
int main() {
    const int N = 8192;
    std::vector<uint32_t> h(N);
    for (int i = 0; i < N; ++i) h[i] = i;
    uint32_t *d, *o;
    hipMalloc(&d, N * 4); hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    // buffer covers only the first 4096 bytes (+ nothing): wave 3 uses soffset = 8192 bytes beyond that range -> tells whether soffset is range-checked
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, o, 4096u, 8192);
    std::vector<uint32_t> r(1024);
    hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
    int bad_place = 0, bad_zero = 0, soff_checked = 0, soff_added = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                uint32_t got = r[w * 256 + l * 4 + e];
                uint32_t src = (uint32_t)((w * 64 + (63 - l)) * 4 + e);
                if (w == 0 && got != src) ++bad_place;
                if (w == 1) { if (l & 1) { if (got != 0) ++bad_zero; } else if (got != src) ++bad_place; }
                if (w == 2) { if (l & 1) { if (got != 0) ++bad_zero; } else if (got != src) ++bad_place; }
                if (w == 3) { if (got == src + 2048) ++soff_added; else if (got == 0) ++soff_checked; }
            }
    printf("dma_probe: bad_place=%d bad_zero=%d (oob sample: 0x%08x 0x%08x) soffset: added=%d zeroed=%d sample=0x%08x\n", bad_place, bad_zero, r[256 + 4], r[512 + 4], soff_added, soff_checked, r[768]);
    return 0;
}
