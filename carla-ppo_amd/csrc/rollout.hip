// rollout.hip — the B = 1 inference path of the rollout loop as ONE C-ABI call: raw uint8 camera frame + measurements ->
// ConvVAE encoder mean z (vae/models.py:199-202, 249-256) -> state = [z, measurements] (vae_common.py:45-59) -> policy / value heads
// (ppo.py:231-251) -> (action, value, z) in one buffer (device memory or pinned host memory).  Exact fp32 (v_mfma_f32_32x32x2_f32).
//
// At one frame every layer is a tiny GEMM with a long K (conv4: 24 x 256 x 2048; the trunks: 1 x 500 x 67, 1 x 300 x 500): the training
// kernels give such a shape a handful of blocks that walk K serially (10-70 us each).  Here every layer is split over K as well: a wave owns
// a (32 rows x 32 columns x 64 k) unit, the four waves of a block meet in LDS and add their tile to the zeroed raw output with fp32 atomics;
// bias + ReLU of a layer are applied by the NEXT layer's operand loader, so no layer needs a finishing pass, and the step is a chain of
// eight launches whose cost is their latency, not their work:
//     conv1 (also clears the raw buffers) -> conv2 -> conv3 -> conv4 -> mean -> trunk layer 1 (both nets) -> trunk layer 2 (both nets) -> heads
#include <stdlib.h>
#include "common.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"
#include "ppo_fused.hpp"

namespace mi {

typedef float f32x16_r __attribute__((ext_vector_type(16)));

// Every operand goes through a buffer descriptor whose extent is the operand's size: rows past M are clamped (their results are dropped by the
// output descriptor), k past K and columns past the matrix read 0.0 from the hardware range check -- no per-lane branches, so the kernels are
// a few hundred instructions long; at one wave per SIMD the instruction count IS the kernel time (the first form, with a guard around every
// load, was 3,500 instructions and 6-7 us per layer).
struct RollConvParams {
    const float* x; const float* x_bias;                 // input [IH,IW,C]; x_bias != NULL: raw sums of the previous layer: x + x_bias[c] on load, then ReLU if `relu`
    const float* x_tail; int split;                       // state-vector form (vec = 0): elements k >= split come from x_tail[k - split] as they are (the measurements)
    const float* w; int ldw;                              // weights [K][ldw] (TF HWIO flattened: k = (kh KW + kw) C + ci), N <= ldw columns used
    float* out;                                           // raw output [M][N] (zeroed; fp32 atomics)
    long long x_net, xb_net, w_net, out_net;              // flat only: blockIdx.y = net (policy / value trunk): per-net strides in floats
    unsigned x_bytes, xb_bytes, tail_bytes, w_bytes, out_bytes;   // descriptor extents
    int IW, C, OW, M, N, K, KW, flat, relu, c_shift, vec; // flat = 1: x is a vector of K values, M = 1; c_shift = log2(C) or -1
};

#define ROLL_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
#define ROLL_OOB 0x40000000u                              // a byte offset past every descriptor of this file

__device__ __forceinline__ float roll_ld(const __amdgpu_buffer_rsrc_t r, unsigned byte_off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0)); }
__device__ __forceinline__ f32x4 roll_ld4(const __amdgpu_buffer_rsrc_t r, unsigned byte_off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0)); }

// one block unit (bx, by, bz) of the grid (ceil(N / 32), ceil(M / 32) | nets, ceil(K / 256)); wave w of the block: k in [256 bz + 64 w, + 64)
// MODE: 0 = conv, power-of-two C and KW = 4 (shifts); 1 = flattened input, power-of-two C (the mean head); 2 = flattened, bias indexed by k
// (trunk layer 2); 3 = the assembled state vector (trunk layer 1); 4 = conv, any C / KW
template <int MODE>
__device__ __forceinline__ void roll_conv_unit(const RollConvParams& p, int bx, int by, int bz, f32x4 (*red)[4][64]) {
    constexpr bool FLAT = MODE == 1 || MODE == 2 || MODE == 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lgrp = lane >> 5;
    const int net = FLAT ? by : 0;
    const int n0 = bx * 32, m0 = FLAT ? 0 : by * 32, kb = bz * 256 + wave * 64;
    const int n = n0 + lrow, m = min(m0 + lrow, p.M - 1);
    const __amdgpu_buffer_rsrc_t rsX = ROLL_RSRC(p.x + net * p.x_net, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rsB = ROLL_RSRC(p.x_bias ? p.x_bias + net * p.xb_net : p.x, p.xb_bytes);
    const __amdgpu_buffer_rsrc_t rsT = ROLL_RSRC(p.x_tail ? p.x_tail : p.x, p.tail_bytes);
    const __amdgpu_buffer_rsrc_t rsW = ROLL_RSRC(p.w + net * p.w_net, p.w_bytes);
    const int oy = FLAT ? 0 : m / p.OW, ox = m - oy * p.OW;
    const unsigned pix = FLAT ? 0u : (unsigned)((2 * oy * p.IW + 2 * ox) * p.C);
    const unsigned ldw = (unsigned)p.ldw;
    f32x4 a[8], b[8], bi[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int k = kb + u * 8 + lgrp * 4;               // 4 consecutive k: one (kh, kw), 4 consecutive input channels (C % 4 == 0)
        if (MODE != 3) {
            unsigned off, ci;
            if (FLAT) { off = (unsigned)k; ci = MODE == 1 ? (unsigned)(k & (p.C - 1)) : (unsigned)k; }
            else {
                int tap, kh, kw;
                if (MODE == 0) { tap = k >> p.c_shift; ci = (unsigned)(k & (p.C - 1)); kh = tap >> 2; kw = tap & 3; }
                else { tap = k / p.C; ci = (unsigned)(k - tap * p.C); kh = tap / p.KW; kw = tap - kh * p.KW; }
                off = k < p.K ? pix + (unsigned)((kh * p.IW + kw) * p.C) + ci : (ROLL_OOB >> 2);
            }
            a[u] = roll_ld4(rsX, off * 4u);
            bi[u] = roll_ld4(rsB, ci * 4u);
        } else {                                           // the assembled state vector [x + bias | tail], any K: x / bias descriptors end at `split`
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned ke = (unsigned)(k + e);
                a[u][e] = roll_ld(rsX, ke * 4u) + roll_ld(rsT, (ke - (unsigned)p.split) * 4u);
                bi[u][e] = roll_ld(rsB, ke * 4u);
            }
        }
        const unsigned wo = ((unsigned)k * ldw + (unsigned)n) * 4u;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[u][e] = roll_ld(rsW, wo + (unsigned)e * ldw * 4u);
    }
    f32x16_r acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float floor_v = p.relu ? 0.f : -__builtin_inff();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        f32x4 av;
#pragma unroll
        for (int e = 0; e < 4; ++e) av[e] = fmaxf(a[u][e] + bi[u][e], floor_v);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b[u][s], acc, 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave - 1][q][lane] = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    }
    __syncthreads();
    if (wave == 0) {
        const __amdgpu_buffer_rsrc_t rsO = ROLL_RSRC(p.out + net * p.out_net, p.out_bytes);
        const unsigned col = n < p.N ? (unsigned)n * 4u : ROLL_OOB, rowb = (unsigned)p.N * 4u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 r0 = red[0][q][lane], r1 = red[1][q][lane], r2 = red[2][q][lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = ((acc[4 * q + e] + r0[e]) + (r1[e] + r2[e]));
                const unsigned mm = (unsigned)(m0 + e + 8 * q + 4 * lgrp);
                __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, rsO, (int)(mm * rowb + col), 0, 0);     // rows past M fall outside the descriptor
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void rollout_conv_kernel(const RollConvParams p) {
    __shared__ f32x4 red[3][4][64];
    roll_conv_unit<MODE>(p, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

static void launch_conv(hipStream_t st, const dim3& g, const RollConvParams& p) {
    const int mode = !p.flat ? ((p.c_shift >= 0 && p.KW == 4) ? 0 : 4) : (!p.vec ? 3 : (p.c_shift >= 0 ? 1 : 2));
    switch (mode) {
        case 0: hipLaunchKernelGGL(rollout_conv_kernel<0>, g, dim3(256), 0, st, p); break;
        case 1: hipLaunchKernelGGL(rollout_conv_kernel<1>, g, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL(rollout_conv_kernel<2>, g, dim3(256), 0, st, p); break;
        case 3: hipLaunchKernelGGL(rollout_conv_kernel<3>, g, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL(rollout_conv_kernel<4>, g, dim3(256), 0, st, p); break;
    }
}

// conv1 from the raw uint8 frame: out[m, n] = relu(sum_k (frame[..] / 255) W[k, n] + b[n]), K = KH KW 3 = 48; one wave per 32 pixels x 32 channels.
// The byte -> float32(k) / float32(255) conversion is the exact in-register form of common.hpp.  grid ceil(M / 128) blocks of 4 waves.
// ROW = KW * Cs when that is 12 (the reference's 4 x 4 x RGB kernel): a patch row is 12 contiguous bytes, 4 consecutive k never straddle two
// rows and the byte address is even, so a lane fetches its 4 patch values as two 16-bit loads; ROW = 0: any shape, one byte load per value.
// The launch also clears the raw-sum buffers of the layers behind it.
struct RollConv1Params {
    const unsigned char* frame; const float* w; const float* bias; float* out;
    int IH, IW, Cs, OW, M, N, KW, K;
    MiZeroList z;
};

__device__ __forceinline__ void roll_zero(const MiZeroList& z, long long t, long long nt) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float* zp = z.p[r]; const long long zn = z.n[r];
        if (!zp) continue;
        if ((((uintptr_t)zp) & 15) == 0) {
            const long long n4 = zn >> 2;
            for (long long i = t; i < n4; i += nt) ((f32x4*)zp)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (long long i = (n4 << 2) + t; i < zn; i += nt) zp[i] = 0.f;
        } else {
            for (long long i = t; i < zn; i += nt) zp[i] = 0.f;
        }
    }
}

// 32 pixels x 32 channels of conv1 by one wave; wunit = index of the 32-pixel group.  Same descriptor discipline as roll_conv_unit.
template <int ROW>
__device__ __forceinline__ void roll_conv1_unit(const RollConv1Params& c, int wunit) {
    const int IW = c.IW, Cs = c.Cs, OW = c.OW, M = c.M, N = c.N, KW = c.KW, K = c.K;
    const int lane = threadIdx.x & 63, lrow = lane & 31, lgrp = lane >> 5;
    const int m0 = wunit * 32;
    if (m0 >= M) return;
    const int m = min(m0 + lrow, M - 1), n = lrow;
    const __amdgpu_buffer_rsrc_t rsF = ROLL_RSRC(c.frame, c.IH * IW * Cs);
    const __amdgpu_buffer_rsrc_t rsW = ROLL_RSRC(c.w, K * N * 4);
    const __amdgpu_buffer_rsrc_t rsO = ROLL_RSRC(c.out, M * N * 4);
    const int oy = m / OW, ox = m - oy * OW;
    f32x16_r acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 a[6], b[6];
    const unsigned base = (unsigned)((2 * oy * IW + 2 * ox) * Cs), rowbytes = (unsigned)(IW * Cs);
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int k0 = u * 8 + lgrp * 4;
        if (ROW == 12) {
            const int kh = k0 / 12, kr = k0 - kh * 12;
            const unsigned o = base + (unsigned)kh * rowbytes + (unsigned)kr;
            const unsigned v = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsF, (int)o, 0, 0) | ((unsigned)__builtin_amdgcn_raw_buffer_load_b16(rsF, (int)o + 2, 0, 0) << 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[u][e] = u8_to_unit_exact((float)((v >> (8 * e)) & 255u));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + e;
                const int tap = k / Cs, ci = k - tap * Cs, kh = tap / KW, kw = tap - kh * KW;
                a[u][e] = u8_to_unit_exact((float)__builtin_amdgcn_raw_buffer_load_b8(rsF, k < K ? (int)(base + (unsigned)kh * rowbytes + (unsigned)(kw * Cs + ci)) : (int)ROLL_OOB, 0, 0));
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) b[u][e] = roll_ld(rsW, (unsigned)((k0 + e) * N + n) * 4u);      // k >= K: past the descriptor, 0.0
    }
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][s], b[u][s], acc, 0, 0, 0);
    const float bn = roll_ld(ROLL_RSRC(c.bias, N * 4), (unsigned)n * 4u);
    const unsigned col = n < N ? (unsigned)n * 4u : ROLL_OOB, rowb = (unsigned)N * 4u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned mm = (unsigned)(m0 + (r & 3) + 8 * (r >> 2) + 4 * lgrp);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(acc[r] + bn, 0.f)), rsO, (int)(mm * rowb + col), 0, 0);
    }
}

template <int ROW>
__global__ __launch_bounds__(256) void rollout_conv1_kernel(const RollConv1Params c) {
    roll_zero(c.z, (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256);
    roll_conv1_unit<ROW>(c, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// heads of the rollout step: h2 = relu(raw layer-2 sums + bias) of both trunks -> action mean (tanh, scaled to the action range, ppo.py:56-67),
// the sampled / greedy action (ppo.py:81-88, clipped), the value (ppo.py:70-71), and the encoder mean z beside them: out = [action | value | z].
// One block; the arithmetic of the finishing thread is that of ppo_predict_head_kernel.
struct RollHeadParams {
    const float *Wm, *bm, *logstd, *b2p, *b2v, *Wv, *bv, *low, *high;
    const float *h2raw, *mean_raw, *mean_bias, *noise;
    float *mean_out, *out;
    int A, H2, z_dim, greedy;
};

template <int NA>
__device__ __forceinline__ void roll_head(const RollHeadParams& q, float (*red)[NA + 1]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, A = q.A, H2 = q.H2;
    float au[NA], av = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) au[a] = 0.f;
    for (int j = tid; j < H2; j += 256) {
        const float hp = fmaxf(q.h2raw[j] + q.b2p[j], 0.f), hv = fmaxf(q.h2raw[H2 + j] + q.b2v[j], 0.f);
        av += hv * q.Wv[j];
#pragma unroll
        for (int a = 0; a < NA; ++a) if (a < A) au[a] += hp * q.Wm[(long long)j * A + a];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        av += __shfl_xor(av, o, 64);
#pragma unroll
        for (int a = 0; a < NA; ++a) au[a] += __shfl_xor(au[a], o, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < NA; ++a) red[wave][a] = au[a];
        red[wave][NA] = av;
    }
    float* out = q.out;
    for (int i = tid; i < q.z_dim; i += 256) out[A + 1 + i] = q.mean_raw[i] + q.mean_bias[i];
    __syncthreads();
    if (tid > A) return;
    if (tid == A) { out[A] = ((red[0][NA] + red[1][NA]) + (red[2][NA] + red[3][NA])) + q.bv[0]; return; }
    float u = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) if (a == tid) u = (red[0][a] + red[1][a]) + (red[2][a] + red[3][a]);
    const float lo = q.low[tid], hi = q.high[tid];
    const float mean = lo + ((tanhf(u + q.bm[tid]) + 1.0f) * 0.5f) * (hi - lo);
    if (q.mean_out) q.mean_out[tid] = mean;
    float act = mean;
    if (!q.greedy) act = fminf(fmaxf(mean + expf(q.logstd[tid]) * q.noise[tid], lo), hi);
    out[tid] = act;
}

template <int NA>
__global__ __launch_bounds__(256) void rollout_head_kernel(const RollHeadParams q) {
    __shared__ float red[4][NA + 1];
    roll_head<NA>(q, red);
}

}  // namespace mi

using namespace mi;

static int fill_conv1(RollConv1Params& c, const unsigned char* frame, const float* w, const float* bias, float* out, int IH, int IW, int Cs, int KH, int KW, int N, const MiZeroList* zero) {
    const int OH = (IH - KH) / 2 + 1, OW = (IW - KW) / 2 + 1;
    c = RollConv1Params{};
    c.frame = frame; c.w = w; c.bias = bias; c.out = out; c.IH = IH; c.IW = IW; c.Cs = Cs; c.OW = OW; c.M = OH * OW; c.N = N; c.KW = KW; c.K = KH * KW * Cs;
    if (zero) c.z = *zero;
    if (N > 32 || c.K > 48) return mi_fail(MI_ERR_SHAPE, "rollout conv1: at most 32 output channels and 48 patch values");
    return MI_OK;
}

int mi_rollout_conv1(hipStream_t st, const unsigned char* frame, const float* w, const float* bias, float* out, int IH, int IW, int Cs, int KH, int KW, int N, const MiZeroList* zero) {
    RollConv1Params c;
    int rc = fill_conv1(c, frame, w, bias, out, IH, IW, Cs, KH, KW, N, zero);
    if (rc != MI_OK) return rc;
    if (KW * Cs == 12) hipLaunchKernelGGL(rollout_conv1_kernel<12>, dim3((c.M + 127) / 128), dim3(256), 0, st, c);
    else hipLaunchKernelGGL(rollout_conv1_kernel<0>, dim3((c.M + 127) / 128), dim3(256), 0, st, c);
    return mi_check_launch("rollout_conv1_kernel");
}

static int log2_exact(int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; }

static int fill_conv(RollConvParams& p, dim3& g, const float* x, const float* x_bias, int IH, int IW, int C, const float* w, int ldw, int N, int KH, int KW, float* out_raw, int flat_k) {
    p = RollConvParams{};
    p.x = x; p.x_bias = x_bias; p.w = w; p.ldw = ldw; p.out = out_raw; p.IW = IW; p.C = C; p.N = N; p.KW = KW; p.flat = flat_k > 0 ? 1 : 0;
    p.relu = x_bias ? 1 : 0; p.c_shift = log2_exact(C);
    if (C % 4 != 0) return mi_fail(MI_ERR_SHAPE, "rollout conv: channels must be a multiple of 4");
    int ny;
    if (p.flat) {
        p.M = 1; p.OW = 1; p.K = flat_k; ny = 1;
        if (p.c_shift < 0 && C < flat_k) return mi_fail(MI_ERR_SHAPE, "rollout conv: a flattened input needs a power-of-two channel count");
    } else { const int OH = (IH - KH) / 2 + 1; p.OW = (IW - KW) / 2 + 1; p.M = OH * p.OW; p.K = KH * KW * C; ny = (p.M + 31) / 32; }
    p.vec = 1;
    if (p.flat && (flat_k & 3)) return mi_fail(MI_ERR_SHAPE, "rollout conv: the flattened input must be a multiple of 4 long");
    const long long xb = (p.flat ? (long long)flat_k : (long long)IH * IW * C) * 4, wb = (long long)p.K * ldw * 4, ob = (long long)p.M * N * 4;
    if (xb >= 0x40000000ll || wb >= 0x40000000ll || ob >= 0x40000000ll) return mi_fail(MI_ERR_SHAPE, "rollout conv: operand beyond 1 GiB");
    p.x_bytes = (unsigned)xb; p.xb_bytes = x_bias ? (unsigned)C * 4u : 0u; p.tail_bytes = 0; p.w_bytes = (unsigned)wb; p.out_bytes = (unsigned)ob;
    g = dim3((N + 31) / 32, ny, (p.K + 255) / 256);
    return MI_OK;
}

int mi_rollout_conv(hipStream_t st, const float* x, const float* x_bias, int IH, int IW, int C, const float* w, int ldw, int N, int KH, int KW, float* out_raw, int flat_k) {
    RollConvParams p; dim3 g;
    int rc = fill_conv(p, g, x, x_bias, IH, IW, C, w, ldw, N, KH, KW, out_raw, flat_k);
    if (rc != MI_OK) return rc;
    launch_conv(st, g, p);
    return mi_check_launch("rollout_conv_kernel");
}

// the two trunk layers as split-K stages: state = [mean_raw + mean_bias | measurements] -> layer 1 -> layer 2, raw sums into the zeroed q.h1 / q.h2
static void fill_trunks(RollConvParams& l1, dim3& g1, RollConvParams& l2, dim3& g2, const PpoFusedParams& q, const float* mean_raw, const float* mean_bias, int z_dim, const float* measurements) {
    RollConvParams p = {};
    p.flat = 1; p.M = 1; p.OW = 1; p.KW = 1; p.c_shift = -1;
    p.x = mean_raw; p.x_bias = mean_bias; p.x_tail = measurements; p.split = z_dim; p.relu = 0; p.C = q.din; p.K = q.din;
    p.w = q.theta + q.off[0]; p.ldw = q.H1; p.N = q.H1; p.out = q.h1;
    p.x_net = 0; p.xb_net = 0; p.w_net = q.off[7] - q.off[0]; p.out_net = q.H1;
    p.vec = 0; p.x_bytes = (unsigned)z_dim * 4u; p.xb_bytes = (unsigned)z_dim * 4u; p.tail_bytes = (unsigned)(q.din - z_dim) * 4u;
    p.w_bytes = (unsigned)q.din * (unsigned)q.H1 * 4u; p.out_bytes = (unsigned)q.H1 * 4u;
    l1 = p; g1 = dim3((q.H1 + 31) / 32, 2, (p.K + 255) / 256);
    p.x = q.h1; p.x_bias = q.theta + q.off[1]; p.x_tail = nullptr; p.split = 0; p.relu = 1; p.C = q.H1; p.K = q.H1;
    p.w = q.theta + q.off[2]; p.ldw = q.H2; p.N = q.H2; p.out = q.h2;
    p.x_net = q.H1; p.xb_net = q.off[8] - q.off[1]; p.w_net = q.off[9] - q.off[2]; p.out_net = q.H2;
    p.vec = 1; p.x_bytes = (unsigned)q.H1 * 4u; p.xb_bytes = (unsigned)q.H1 * 4u; p.tail_bytes = 0;
    p.w_bytes = (unsigned)q.H1 * (unsigned)q.H2 * 4u; p.out_bytes = (unsigned)q.H2 * 4u;
    l2 = p; g2 = dim3((q.H2 + 31) / 32, 2, (p.K + 255) / 256);
}

static void fill_head(RollHeadParams& h, const PpoFusedParams& q, const float* mean_raw, const float* mean_bias, int z_dim, const float* noise, int greedy, float* out) {
    h = RollHeadParams{};
    h.Wm = q.theta + q.off[4]; h.bm = q.theta + q.off[5]; h.logstd = q.theta + q.off[6]; h.b2p = q.theta + q.off[3]; h.b2v = q.theta + q.off[10];
    h.Wv = q.theta + q.off[11]; h.bv = q.theta + q.off[12]; h.low = q.low; h.high = q.high;
    h.h2raw = q.h2; h.mean_raw = mean_raw; h.mean_bias = mean_bias; h.noise = noise; h.mean_out = q.mean_out; h.out = out;
    h.A = q.A; h.H2 = q.H2; h.z_dim = z_dim; h.greedy = greedy;
}

int mi_rollout_policy(hipStream_t st, const PpoFusedParams& q, const float* mean_raw, const float* mean_bias, int z_dim, const float* measurements,
                      const float* noise, int greedy, float* out) {
    if (q.A < 1 || q.A > 8) return mi_fail(MI_ERR_ARG, "rollout step: 1 <= num_actions <= 8");
    if (q.H1 % 4 != 0) return mi_fail(MI_ERR_SHAPE, "rollout step: hidden sizes must be multiples of 4");
    RollConvParams l1, l2; dim3 g1, g2; RollHeadParams h;
    fill_trunks(l1, g1, l2, g2, q, mean_raw, mean_bias, z_dim, measurements);
    fill_head(h, q, mean_raw, mean_bias, z_dim, noise, greedy, out);
    launch_conv(st, g1, l1);
    launch_conv(st, g2, l2);
    if (q.A <= 2) hipLaunchKernelGGL(rollout_head_kernel<2>, dim3(1), dim3(256), 0, st, h);
    else hipLaunchKernelGGL(rollout_head_kernel<8>, dim3(1), dim3(256), 0, st, h);
    return mi_check_launch("rollout_policy");
}

