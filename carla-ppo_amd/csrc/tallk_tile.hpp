// tallk_tile.hpp — split-K dense layers with a long reduction and a narrow output (bf16): the latent-side GEMMs of the ConvVAE step
//   heads forward      [B, 6144] x [6144, 128]   (vae/models.py:97-98)        dense1 input gradient   [B, 6144] x [6144, 64]^T   (vae/models.py:259)
// and MlpVAE's 38400-long ones.  Under a millisecond of work, but ON the step's critical path between the encoder and the decoder, where the general
// kernel (32-deep K steps staged through LDS, one barrier each) took 21 / 16 us for 0.8 / 0.4 GFLOP.  Both operands are K-contiguous here, so an
// MFMA fragment (8 consecutive k of one row / column) IS a 16-byte global load: no LDS, no barrier, every load of a K-slice requested before its
// first MFMA.  A block owns 32 rows x all N columns (N / 32 output tiles) x one K slice; its four waves take (tile, K sub-slice) pairs and each
// writes its own fp32 slab [nsplit][M][N], summed by the consumer (reparameterisation kernels / mi_splitk_finish).
#pragma once
#include "common.hpp"

namespace mi {

struct TallKParams {
    const void* a; const void* wt; float* out;           // a [M][K], wt [N][K] (bf16), out [nsplit][M][N]
    int M, N, K, len;                                     // len: k per (blockIdx.y, sub-slice) = per slab
    unsigned a_bytes, w_bytes;
};

// NT = N / 32 output tiles (1, 2, 4); the four waves of a block: tile w % NT, K sub-slice w / NT; slab = blockIdx.y * (4 / NT) + sub-slice
template <int NT>
__global__ __launch_bounds__(256) void tallk_kernel(const TallKParams p) {
    constexpr int KP = 4 / NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lgrp = lane >> 5;
    const int nt = wave % NT, kp = wave / NT;
    const int m0 = blockIdx.x * 32, slab = blockIdx.y * KP + kp;
    const int k0 = slab * p.len;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wt, 0, (int)p.w_bytes, 0x00020000);
    const unsigned arow = (unsigned)min(m0 + lrow, p.M - 1) * (unsigned)p.K, wrow = (unsigned)(nt * 32 + lrow) * (unsigned)p.K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int U = 6;                                  // k16-steps requested together (12 x 16-byte loads per lane in flight)
    const int steps = p.len >> 4;
    for (int s0 = 0; s0 < steps; s0 += U) {
        u16x8 fa[U], fb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned k = (unsigned)(k0 + (s0 + u) * 16 + lgrp * 8);
            const bool on = s0 + u < steps;               // wave-uniform
            fa[u] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rsA, on ? (int)((arow + k) * 2u) : (int)0x40000000, 0, 0));
            fb[u] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, on ? (int)((wrow + k) * 2u) : (int)0x40000000, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[u]), __builtin_bit_cast(bf16x8, fb[u]), acc, 0, 0, 0);
    }
    // D[row = m][col = n]: register r of a lane = row (r & 3) + 8 (r >> 2) + 4 lgrp, column lrow: 32 lanes store 128 contiguous bytes
    float* const o = p.out + ((long long)slab * p.M) * p.N + nt * 32 + lrow;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lgrp;
        if (m < p.M) o[(long long)m * p.N] = acc[r];
    }
}

}  // namespace mi
