// Probe: exact lane/element semantics of ds_read_b64_tr_b16 on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int mode) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x;
  int addr;
  if (mode == 0) addr = l * 4;                                  // canonical packed
  else addr = ((l & 15) >> 2) * 100 + (l & 3) * 4 + (l >> 4) * 1000;  // row stride 100 elements, group stride 1000
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]); printf("\n"); }
  }
  return 0;
}
