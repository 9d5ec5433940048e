// rollout.hip — the B = 1 inference path of the rollout loop as ONE C-ABI call: raw uint8 camera frame + measurements ->
// ConvVAE encoder mean z (vae/models.py:199-202, 249-256) -> state = [z, measurements] (vae_common.py:45-59) -> policy / value heads
// (ppo.py:231-251) -> (action, value, z) in one device buffer, one D2H copy.  Exact fp32 (v_mfma_f32_32x32x2_f32).
//
// At one frame the conv layers are tiny GEMMs with long K (conv4: 24 x 256 x 2048): the training kernels give them a handful of blocks that
// walk K serially (20-70 us each).  Here every layer is split over K as well: a wave owns a (32 pixels x 32 channels x 64 k) unit, the four
// waves of a block meet in LDS and add their tile to the (zeroed) raw output with fp32 atomics; bias + ReLU of a layer are applied by the
// NEXT layer's operand loader, so no layer needs a finishing pass.  9 launches of a few microseconds.
#include <stdlib.h>
#include "common.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"
#include "ppo_fused.hpp"

namespace mi {

typedef float f32x16_r __attribute__((ext_vector_type(16)));

struct RollConvParams {
    const float* x; const float* x_bias;                 // input [IH,IW,C] (x_bias != NULL: raw sums of the previous layer: relu(x + x_bias[c]) on load)
    const float* w; int ldw;                              // weights [K][ldw] (TF HWIO flattened: k = (kh KW + kw) C + ci), N <= ldw columns used
    float* out;                                           // raw output [M][N] (zeroed; fp32 atomics)
    int IH, IW, C, OW, M, N, K, KW, flat;                 // flat = 1: x is a flat vector of K values (dense head): k indexes it directly
};

// grid (ceil(N / 32), ceil(M / 32), ceil(K / 256)); wave w of a block: k in [256 z + 64 w, + 64)
__global__ __launch_bounds__(256) void rollout_conv_kernel(const RollConvParams p) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lgrp = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, kb = blockIdx.z * 256 + wave * 64;
    const int m = m0 + lrow, n = n0 + lrow;
    const bool mok = m < p.M, nok = n < p.N;
    const int oy = mok ? m / p.OW : 0, ox = mok ? m - oy * p.OW : 0;
    f32x4 a[8], b[8], bi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = kb + u * 8 + lgrp * 4;               // 4 consecutive k: one (kh, kw), 4 consecutive input channels (C % 4 == 0)
        const bool kok = k < p.K;
        a[u] = f32x4{0.f, 0.f, 0.f, 0.f}; bi[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (mok && kok) {
            int ci; long long off;
            if (p.flat) { off = k; ci = k % p.C; }
            else { const int tap = k / p.C; ci = k - tap * p.C; const int kh = tap / p.KW, kw = tap - kh * p.KW; off = ((long long)(2 * oy + kh) * p.IW + 2 * ox + kw) * p.C + ci; }
            a[u] = *(const f32x4*)(p.x + off);
            if (p.x_bias) bi[u] = *(const f32x4*)(p.x_bias + ci);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) b[u][e] = (nok && k + e < p.K) ? p.w[(long long)(k + e) * p.ldw + n] : 0.f;
    }
    f32x16_r acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool act = p.x_bias != nullptr;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        f32x4 av = a[u];
        if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = (mok && kb + u * 8 + lgrp * 4 < p.K) ? fmaxf(av[e] + bi[u][e], 0.f) : 0.f;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b[u][s], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0 && nok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = ((acc[r] + red[0][r][lane]) + (red[1][r][lane] + red[2][r][lane]));
            const int mm = m0 + (r & 3) + 8 * (r >> 2) + 4 * lgrp;
            if (mm < p.M) atomicAdd(p.out + (long long)mm * p.N + n, v);
        }
    }
}

// conv1 from the raw uint8 frame: out[m, n] = relu(sum_k (frame[..] / 255) W[k, n] + b[n]), K = KH KW 3 = 48; one wave per 32 pixels x 32 channels.
// The byte -> float32(k) / float32(255) conversion is the exact in-register form of common.hpp.  grid ceil(M / 128) blocks of 4 waves.
__global__ __launch_bounds__(256) void rollout_conv1_kernel(const unsigned char* __restrict__ frame, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ out, int IH, int IW, int Cs, int OW, int M, int N, int KW, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lgrp = lane >> 5;
    const int m0 = (blockIdx.x * 4 + wave) * 32;
    if (m0 >= M) return;
    const int m = m0 + lrow, n = lrow;
    const bool mok = m < M, nok = n < N;
    const int oy = mok ? m / OW : 0, ox = mok ? m - oy * OW : 0;
    f32x16_r acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int ksteps = (K + 7) / 8;                       // 6
    f32x4 a[6], b[6];
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = u * 8 + lgrp * 4 + e;
            const bool kok = u < ksteps && k < K;
            const int tap = k / Cs, ci = k - tap * Cs, kh = tap / KW, kw = tap - kh * KW;
            a[u][e] = (mok && kok) ? u8_to_unit_exact((float)frame[((long long)(2 * oy + kh) * IW + 2 * ox + kw) * Cs + ci]) : 0.f;
            b[u][e] = (nok && kok) ? w[(long long)k * N + n] : 0.f;
        }
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][s], b[u][s], acc, 0, 0, 0);
    if (nok) {
        const float bn = bias[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = m0 + (r & 3) + 8 * (r >> 2) + 4 * lgrp;
            if (mm < M) out[(long long)mm * N + n] = fmaxf(acc[r] + bn, 0.f);
        }
    }
}

}  // namespace mi

using namespace mi;

int mi_rollout_conv1(hipStream_t st, const unsigned char* frame, const float* w, const float* bias, float* out, int IH, int IW, int Cs, int KH, int KW, int N) {
    const int OH = (IH - KH) / 2 + 1, OW = (IW - KW) / 2 + 1, M = OH * OW, K = KH * KW * Cs;
    if (N > 32 || K > 48) return mi_fail(MI_ERR_SHAPE, "rollout conv1: at most 32 output channels and 48 patch values");
    hipLaunchKernelGGL(rollout_conv1_kernel, dim3((M + 127) / 128), dim3(256), 0, st, frame, w, bias, out, IH, IW, Cs, OW, M, N, KW, K);
    return mi_check_launch("rollout_conv1_kernel");
}

int mi_rollout_conv(hipStream_t st, const float* x, const float* x_bias, int IH, int IW, int C, const float* w, int ldw, int N, int KH, int KW, float* out_raw, int flat_k) {
    RollConvParams p = {};
    p.x = x; p.x_bias = x_bias; p.w = w; p.ldw = ldw; p.out = out_raw; p.IH = IH; p.IW = IW; p.C = C; p.N = N; p.KW = KW; p.flat = flat_k > 0 ? 1 : 0;
    if (C % 4 != 0) return mi_fail(MI_ERR_SHAPE, "rollout conv: channels must be a multiple of 4");
    if (p.flat) { p.M = 1; p.OW = 1; p.K = flat_k; }
    else { const int OH = (IH - KH) / 2 + 1; p.OW = (IW - KW) / 2 + 1; p.M = OH * p.OW; p.K = KH * KW * C; }
    const dim3 g((N + 31) / 32, (p.M + 31) / 32, (p.K + 255) / 256);
    hipLaunchKernelGGL(rollout_conv_kernel, g, dim3(256), 0, st, p);
    return mi_check_launch("rollout_conv_kernel");
}
