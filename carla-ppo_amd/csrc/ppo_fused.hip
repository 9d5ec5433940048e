// ppo_fused.hip — the PPO minibatch step (reference ppo.py:42-66,112-147,218-229) as FIVE launches instead of ~28:
//
//   ppo_l1_kernel         h1 = relu(s W1 + b1)          for the policy, the value net and (unless log pi_old is cached) the old policy
//   ppo_l2_kernel         h2 = relu(h1 W2 + b2)         same nets; the four waves of a block split K = 500
//   ppo_head_loss_kernel  heads (tanh-squashed mean, value), log-probabilities, ratio / clip / value / entropy terms, their gradients
//                         wrt the head pre-activations, per-block loss partial sums, and dh2 = head input gradients masked by relu'(h2)
//   ppo_dh1_kernel        dh1 = (dh2 W2^T) * relu'(h1)
//   ppo_wgrad_kernel      every weight / bias gradient of both nets as independent 32 x 32 tiles, one wave each (dW2 = h1^T dh2, dW1 = s^T dh1,
//                         head kernels, biases as all-ones MFMA rows) and -- single GPU -- its TF-Adam update straight from the accumulators;
//                         one spare wave finalises the loss scalars and logstd
//
// All arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32 = fmaf chains; 1e-4 parity with the oracle).  The step is latency bound (78 MFLOP,
// 1.5 MB of weights): what matters is the number of dependent launches and that none of them serialises on one CU.  With FUSE_ADAM the
// gradient never goes through HBM as a separate buffer: the block that finishes a weight tile's gradient (it sums over ALL minibatch rows)
// applies tf.train.AdamOptimizer to that tile at once.  Data parallel (more than one rank) uses the same kernels with FUSE_ADAM = false: they
// write the flat gradient buffer, the host all-reduces it and mi_adam_tf_flat applies the update.
#include <stdio.h>
#include <stdlib.h>
#include "common.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"
#include "ppo_fused.hpp"

namespace mi {

constexpr float PF_HALF_LOG_2PI = 0.918938533204672741780329736406f;
constexpr int PF_MAX_ACT = 8;
constexpr int PF_NPART = 3 + 2 * PF_MAX_ACT;             // per loss block: policy sum, value sq sum, ratio sum, dlogstd[A], action-mean sums [A]

typedef float f32x16_t __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void pf_mma4(const f32x4& a, const f32x4& b, f32x16_t& c) {
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], c, 0, 0, 0);
}
// D register r of a lane: row 8 (r >> 2) + 4 (lane >> 5) + (r & 3), column lane & 31
__device__ __forceinline__ int pf_row(int r, int lgrp) { return (r & 3) + 8 * (r >> 2) + 4 * lgrp; }

// U k-steps at a time: every operand of the chunk is requested before its first MFMA (a plain loop pays one L2 / HBM latency per step: the
// compiler does not software-pipeline it, and at 32 rows the whole step is nothing but such latencies)
template <int U, class FA, class FB>
__device__ __forceinline__ void pf_mma_chunked(int sb, int se, FA load_a, FB load_b, f32x16_t& acc) {
    for (int s0 = sb; s0 < se; s0 += U) {
        f32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s0 + u < se) { a[u] = load_a(s0 + u); b[u] = load_b(s0 + u); }
            else { a[u] = f32x4{0.f, 0.f, 0.f, 0.f}; b[u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) pf_mma4(a[u], b[u], acc);
    }
}

// operands through buffer descriptors: rows are clamped, reads past a tensor return 0.0 from the hardware range check, stores past it are dropped --
// no per-lane guard branches (at one wave per SIMD the instruction count is the kernel time: rollout.hip)
#define PF_RSRC(ptr, bytes) __builtin_amdgcn_make_buffer_rsrc((void*)(ptr), 0, (int)(bytes), 0x00020000)
#define PF_OOB 0x40000000u
__device__ __forceinline__ float pf_ld(const __amdgpu_buffer_rsrc_t r, unsigned byte_off) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0)); }
__device__ __forceinline__ f32x4 pf_ld4(const __amdgpu_buffer_rsrc_t r, unsigned byte_off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0)); }
__device__ __forceinline__ void pf_st(const __amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)byte_off, 0, 0); }

// TF ApplyAdam on one element
__device__ __forceinline__ void pf_adam(float* p, float* m, float* v, float g, const PpoFusedParams& q) {
    float mm = *m, vv = *v;
    mm += (g - mm) * q.omb1;
    vv += (g * g - vv) * q.omb2;
    *m = mm; *v = vv;
    *p -= (mm * q.alpha) / (sqrtf(vv) + q.epsilon);
}
template <bool FUSE> __device__ __forceinline__ void pf_emit(const PpoFusedParams& q, long long idx, float g, float* gb) {      // gb: q.grads, or this row chunk's slab
    if constexpr (FUSE) pf_adam(q.theta + idx, q.adam_m + idx, q.adam_v + idx, g, q);
    else gb[idx] = g;
}

__device__ __forceinline__ const float* pf_theta(const PpoFusedParams& q, int net) { return net == 2 ? q.theta_old : q.theta; }
// tensor index of (net, which): which 0 W1, 1 b1, 2 W2, 3 b2, 4 head kernel, 5 head bias
__device__ __forceinline__ long long pf_off(const PpoFusedParams& q, int net, int which) { return q.off[(net == 1 ? 7 : 0) + which]; }

// ---------------------------------------------------------------------------------------------------------------------
// layer 1: grid (ceil(H1 / 128), n_nets, ceil(M / 32)); wave w of a block owns the 32-column tile 4 blockIdx.x + w.  K = kin (72): 9 steps.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ppo_l1_kernel(const PpoFusedParams q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lgrp = lane >> 5;
    const int net = blockIdx.y, m0 = blockIdx.z * 32, n0 = (blockIdx.x * 4 + wave) * 32;
    // the gathered minibatch for the layer-1 filter gradient (rows m0 .. m0 + 31, columns dealt to the 8 half-waves of block column 0) is written BEFORE the
    // tile-range return: with H1 < 128 the waves whose column tile lies past H1 still own their share of the columns (a hidden size of 64 left
    // three quarters of s_gath unwritten)
    if (q.row_idx && q.s_gath && net == 0 && blockIdx.x == 0) {
        const int m = m0 + lrow;
        if (m < q.M) {
            const long long srow_g = min(max(q.row_idx[m], 0), q.n_rows - 1);      // (rows are clamped into the table: a bad index reads a wrong row, never past the buffers)
            for (int k = wave * 2 + lgrp; k < q.din; k += 8) q.s_gath[(long long)m * q.din + k] = q.states[srow_g * q.din + k];
        }
    }
    if (n0 >= q.H1) return;
    const float* th = pf_theta(q, net);
    const float* W = th + pf_off(q, net, 0);
    const float* bias = th + pf_off(q, net, 1);
    const int n = n0 + lrow;
    // states rows are din long, W1 has kin >= din rows of which the last kin - din are zero (and stay zero: their gradient is never formed): a k past
    // din reads the next row's finite values against a zero weight row; past the tensors the range check returns 0.0
    // (with a row index: the state table has n_rows rows and sample m is its row row_idx[m] -- one dependent load per lane in front of the operand loads)
    const int mrow = min(m0 + lrow, q.M - 1);
    const int srow = q.row_idx ? min(max(q.row_idx[mrow], 0), q.n_rows - 1) : mrow;
    const __amdgpu_buffer_rsrc_t rsS = PF_RSRC(q.states, (long long)(q.row_idx ? q.n_rows : q.M) * q.din * 4), rsW = PF_RSRC(W, (long long)q.kin * q.H1 * 4);
    const unsigned arow = (unsigned)srow * (unsigned)q.din, H1u = (unsigned)q.H1;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    pf_mma_chunked<9>(0, (q.kin + 7) / 8,
        [&](int s_) { const unsigned k = (unsigned)(s_ * 8 + lgrp * 4); f32x4 a;
#pragma unroll
                      for (int e = 0; e < 4; ++e) a[e] = pf_ld(rsS, (arow + k + e) * 4u);
                      return a; },
        [&](int s_) { const unsigned k = (unsigned)(s_ * 8 + lgrp * 4); f32x4 b;
#pragma unroll
                      for (int e = 0; e < 4; ++e) b[e] = pf_ld(rsW, ((k + e) * H1u + (unsigned)n) * 4u);
                      return b; }, acc);
    {
        const float bn = pf_ld(PF_RSRC(bias, (long long)q.H1 * 4), (unsigned)n * 4u);
        const __amdgpu_buffer_rsrc_t rsO = PF_RSRC(q.h1 + (long long)net * q.M * q.H1, (long long)q.M * q.H1 * 4);
        const unsigned col = n < q.H1 ? (unsigned)n * 4u : PF_OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) pf_st(rsO, (unsigned)(m0 + pf_row(r, lgrp)) * H1u * 4u + col, fmaxf(acc[r] + bn, 0.f));      // rows past M fall outside the descriptor
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// layer 2: grid (ceil(H2 / 32), n_nets, ceil(M / 32)); the four waves split K = H1, partial tiles meet in LDS.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ppo_l2_kernel(const PpoFusedParams q) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lgrp = lane >> 5;
    const int net = blockIdx.y, m0 = blockIdx.z * 32, n0 = blockIdx.x * 32;
    const float* th = pf_theta(q, net);
    const float* W = th + pf_off(q, net, 2);
    const float* bias = th + pf_off(q, net, 3);
    const float* x = q.h1 + (long long)net * q.M * q.H1;
    const int K = q.H1;
    const int ksteps = (K + 7) / 8, per = (ksteps + 3) / 4;
    const int sb = wave * per, se = min(ksteps, sb + per);
    const int n = n0 + lrow;
    const __amdgpu_buffer_rsrc_t rsX = PF_RSRC(x, (long long)q.M * K * 4), rsW = PF_RSRC(W, (long long)K * q.H2 * 4);
    const unsigned arow = (unsigned)min(m0 + lrow, q.M - 1) * (unsigned)K, H2u = (unsigned)q.H2;
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    pf_mma_chunked<16>(sb, se,                            // K % 4 == 0; a k past K reads the next row of x against weights past the tensor (0.0)
        [&](int s_) { return pf_ld4(rsX, (arow + (unsigned)(s_ * 8 + lgrp * 4)) * 4u); },
        [&](int s_) { const unsigned k = (unsigned)(s_ * 8 + lgrp * 4); f32x4 b;
#pragma unroll
                      for (int e = 0; e < 4; ++e) b[e] = pf_ld(rsW, ((k + e) * H2u + (unsigned)n) * 4u);
                      return b; }, acc);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const float bn = pf_ld(PF_RSRC(bias, (long long)q.H2 * 4), (unsigned)n * 4u);
        const __amdgpu_buffer_rsrc_t rsO = PF_RSRC(q.h2 + (long long)net * q.M * q.H2, (long long)q.M * q.H2 * 4);
        const unsigned col = n < q.H2 ? (unsigned)n * 4u : PF_OOB;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = ((acc[r] + red[0][r][lane]) + (red[1][r][lane] + red[2][r][lane]));
            pf_st(rsO, (unsigned)(m0 + pf_row(r, lgrp)) * H2u * 4u + col, fmaxf(v + bn, 0.f));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// heads + loss + head input gradients: grid ceil(M / 32) blocks of 256 threads; a block owns 32 samples, 8 threads per sample.
//   u[m,a] = h2_pi[m,:] Wm[:,a] + bm[a] (same for the old policy), vraw[m] = h2_v[m,:] Wv + bv
//   mean = low + (tanh(u)+1)/2 (high-low) ; logp = sum_a -.5 z^2 - (.5 log 2pi + log sigma) ; ratio = exp(logp - logp_old)
//   loss = -mean(min(r A, clip(r) A)) + vs mean((V-R)^2) - es sum_a(entropy)        (ppo.py:58-66,112-132; tf.minimum's tie rule)
//   du, dv = d loss / d head pre-activations ; dh2_pi[m,j] = relu'(h2) sum_a du[m,a] Wm[j,a] ; dh2_v[m,j] = relu'(h2) dv[m] Wv[j]
// Every thread requests its whole share of the three h2 rows (<= 10 float4 each) before anything is used, the head kernels are staged in
// LDS meanwhile, and the same registers produce dh2 at the end: one memory latency for the whole kernel.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PF_H2MAX = 320;                             // head kernels staged in LDS: H2 <= 320 (the reference: 300)
template <int NA>
__global__ __launch_bounds__(256) void ppo_head_loss_kernel(const PpoFusedParams q) {
    __shared__ __attribute__((aligned(16))) float sWm[PF_H2MAX * PF_MAX_ACT], sWo[PF_H2MAX * PF_MAX_ACT], sWv[PF_H2MAX];
    __shared__ float su[32][NA], suo[32][NA], sv[32], sdu[32][NA], sdv[32], spart[32][PF_NPART];
    const int tid = threadIdx.x, m0 = blockIdx.x * 32, A = q.A, H2 = q.H2;
    const float* __restrict__ Wm = q.theta + q.off[4]; const float* __restrict__ bm = q.theta + q.off[5];
    const float* __restrict__ Wmo = q.theta_old + q.off[4]; const float* __restrict__ bmo = q.theta_old + q.off[5];
    const float* __restrict__ Wv = q.theta + q.off[11]; const float* __restrict__ bv = q.theta + q.off[12];
    const float* __restrict__ h2p = q.h2; const float* __restrict__ h2v = q.h2 + (long long)q.M * H2; const float* __restrict__ h2o = q.h2 + 2ll * q.M * H2;
    const bool old_net = q.logp_old == nullptr;
    const int sm = tid >> 3, part = tid & 7, m = m0 + sm;
    const bool mok = m < q.M;
    constexpr int NV = (PF_H2MAX / 4 + 7) / 8;            // float4 pieces per thread and row (10)
    const int nf = H2 >> 2;                               // float4 per row (H2 % 4 == 0)
    f32x4 hp[NV], hv[NV], ho[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int f = part + 8 * i;
        const bool ok = mok && f < nf;
        hp[i] = ok ? *(const f32x4*)(h2p + (long long)m * H2 + 4 * f) : f32x4{0.f, 0.f, 0.f, 0.f};
        hv[i] = ok ? *(const f32x4*)(h2v + (long long)m * H2 + 4 * f) : f32x4{0.f, 0.f, 0.f, 0.f};
        ho[i] = (ok && old_net) ? *(const f32x4*)(h2o + (long long)m * H2 + 4 * f) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // head kernels -> LDS through registers (all requests in flight together), and the per-sample scalars the loss thread will need
    constexpr int NST = (PF_H2MAX * NA + 255) / 256;
    float stm[NST], sto[NST], stv[(PF_H2MAX + 255) / 256];
#pragma unroll
    for (int i = 0; i < NST; ++i) { const int x = tid + 256 * i; stm[i] = x < H2 * A ? Wm[x] : 0.f; sto[i] = (old_net && x < H2 * A) ? Wmo[x] : 0.f; }
#pragma unroll
    for (int i = 0; i < (PF_H2MAX + 255) / 256; ++i) { const int x = tid + 256 * i; stv[i] = x < H2 ? Wv[x] : 0.f; }
    float p_act[NA], p_adv_s = 0.f, p_ret_s = 0.f, p_lpo_s = 0.f, p_ls[NA], p_lso[NA], p_lo[NA], p_hi[NA];
    const int mr = (q.row_idx && mok) ? min(max(q.row_idx[m], 0), q.n_rows - 1) : m;     // the sample's row in the horizon-batch tables (actions / returns / advantages / cached log pi_old)
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const bool aok = a < A;
        p_act[a] = (mok && aok) ? q.actions[(long long)mr * A + a] : 0.f;
        p_ls[a] = aok ? q.theta[q.off[6] + a] : 0.f; p_lso[a] = (aok && old_net) ? q.theta_old[q.off[6] + a] : 0.f;
        p_lo[a] = aok ? q.low[a] : 0.f; p_hi[a] = aok ? q.high[a] : 0.f;
    }
    if (mok) { p_adv_s = q.adv[mr]; p_ret_s = q.returns[mr]; if (!old_net) p_lpo_s = q.logp_old[mr]; }
#pragma unroll
    for (int i = 0; i < NST; ++i) { const int x = tid + 256 * i; if (x < H2 * A) { sWm[x] = stm[i]; sWo[x] = sto[i]; } }
#pragma unroll
    for (int i = 0; i < (PF_H2MAX + 255) / 256; ++i) { const int x = tid + 256 * i; if (x < H2) sWv[x] = stv[i]; }
    __syncthreads();
    {
        float au[NA], ao[NA], av = 0.f;
#pragma unroll
        for (int a = 0; a < NA; ++a) { au[a] = 0.f; ao[a] = 0.f; }
        if constexpr (NA == 2) {                          // exactly two actions (the reference; the host picks this instantiation only then): head-kernel rows of 4 consecutive j are two 16-byte LDS reads
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int f = min(part + 8 * i, nf - 1);  // (threads past the row re-read its last piece: their h2 registers are zero)
                const f32x4 wv = *(const f32x4*)(sWv + 4 * f);
                const f32x4 wa = *(const f32x4*)(sWm + 8 * f), wb = *(const f32x4*)(sWm + 8 * f + 4);
                const f32x4 oa = *(const f32x4*)(sWo + 8 * f), ob = *(const f32x4*)(sWo + 8 * f + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    av += hv[i][e] * wv[e];
                    const float w0 = e < 2 ? wa[2 * e] : wb[2 * e - 4], w1 = e < 2 ? wa[2 * e + 1] : wb[2 * e - 3];
                    const float o0 = e < 2 ? oa[2 * e] : ob[2 * e - 4], o1 = e < 2 ? oa[2 * e + 1] : ob[2 * e - 3];
                    au[0] += hp[i][e] * w0; au[NA > 1 ? 1 : 0] += hp[i][e] * w1;
                    ao[0] += ho[i][e] * o0; ao[NA > 1 ? 1 : 0] += ho[i][e] * o1;
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = part + 8 * i;
            if (f >= nf) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * f + e;
                av += hv[i][e] * sWv[j];
                _Pragma("unroll") for (int a = 0; a < NA; ++a) if (a < A) { au[a] += hp[i][e] * sWm[j * A + a]; ao[a] += ho[i][e] * sWo[j * A + a]; }
            }
        }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) {
            av += __shfl_xor(av, o, 64);
#pragma unroll
            for (int a = 0; a < NA; ++a) { au[a] += __shfl_xor(au[a], o, 64); ao[a] += __shfl_xor(ao[a], o, 64); }
        }
        if (part == 0) {
            sv[sm] = av + bv[0];
            _Pragma("unroll") for (int a = 0; a < NA; ++a) if (a < A) { su[sm][a] = au[a] + bm[a]; suo[sm][a] = old_net ? ao[a] + bmo[a] : 0.f; }
        }
    }
    __syncthreads();
    // ---- per-sample loss terms: lane `part` of a sample's 8 threads owns action `part` (new and old policy), the terms meet by shuffles.
    //      exp / log / tanh through the hardware transcendental units (v_exp_f32 / v_log_f32: ~1 ulp; tanh(u) = 1 - 2 / (e^2u + 1)): the
    //      libm sequences are ~100-instruction dependent chains, which at one wave per block was most of this kernel ----
    {
        auto fast_tanh = [](float x) { const float xc = fminf(fmaxf(x, -15.f), 15.f); return 1.0f - 2.0f / (__expf(2.0f * xc) + 1.0f); };
        float lp_n = 0.f, lp_o = 0.f, dl = 0.f, zsq = 0.f, mean = 0.f;
        float act = 0.f, ls = 0.f, lso = 0.f, lo = 0.f, hi = 0.f;
#pragma unroll
        for (int a = 0; a < NA; ++a) if (a == part) { act = p_act[a]; ls = p_ls[a]; lso = p_lso[a]; lo = p_lo[a]; hi = p_hi[a]; }
        const bool aok = part < A && mok;
        if (aok) {
            const float t = fast_tanh(su[sm][part < NA ? part : 0]);
            mean = lo + ((t + 1.0f) * 0.5f) * (hi - lo);
            const float sigma = __expf(ls);
            const float z = (act - mean) / sigma;
            lp_n = -0.5f * z * z - (PF_HALF_LOG_2PI + __logf(sigma));
            dl = (z / sigma) * (0.5f * (hi - lo)) * (1.0f - t * t);
            zsq = z * z;
            if (q.mean_out) q.mean_out[(long long)m * A + part] = mean;
            if (old_net) {
                const float to = fast_tanh(suo[sm][part < NA ? part : 0]);
                const float mo = lo + ((to + 1.0f) * 0.5f) * (hi - lo);
                const float so = __expf(lso);
                const float zo = (act - mo) / so;
                lp_o = -0.5f * zo * zo - (PF_HALF_LOG_2PI + __logf(so));
            }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { lp_n += __shfl_xor(lp_n, o, 64); lp_o += __shfl_xor(lp_o, o, 64); }
        if (!old_net) lp_o = p_lpo_s;
        const float r = __expf(lp_n - lp_o);
        const float ad = p_adv_s;
        const float rc = fminf(fmaxf(r, 1.0f - q.clip_eps), 1.0f + q.clip_eps);
        const float s1 = r * ad, s2 = rc * ad;
        const float dr = (s1 <= s2) ? ad : 0.f;           // tf.minimum: gradient to the first argument on ties; the clipped branch has zero slope
        const float coef = -dr * r * q.inv_m;
        const float d = coef * dl;
        if (part < NA) sdu[sm][part < NA ? part : 0] = aok ? d : 0.f;
        if (aok) q.du[(long long)m * A + part] = d;
        const float dvv = sv[sm] - p_ret_s;
        const float dvm = mok ? 2.0f * q.value_scale * dvv * q.inv_m : 0.f;
#pragma unroll
        for (int k = part; k < PF_NPART; k += 8) spart[sm][k] = 0.f;
        __builtin_amdgcn_wave_barrier();
        if (part == 0) {
            sdv[sm] = dvm;
            if (mok) { q.dv[m] = dvm; if (q.logp_out) q.logp_out[m] = lp_n; spart[sm][0] = fminf(s1, s2); spart[sm][1] = dvv * dvv; spart[sm][2] = r; }
        }
        if (aok) { spart[sm][3 + part] = coef * (zsq - 1.0f); spart[sm][3 + PF_MAX_ACT + part] = mean; }
    }
    __syncthreads();
    if (tid < PF_NPART) {                                 // fixed-order block partial sums
        float sum = 0.f;
        for (int i = 0; i < 32; ++i) sum += spart[i][tid];
        q.partial[(long long)blockIdx.x * PF_NPART + tid] = sum;
    }
    // ---- head input gradients of this thread's columns of its sample, masked by relu'(h2): from the rows still held in registers ----
    if (mok) {
        float* dh2p = q.dh2 + (long long)m * H2; float* dh2v = q.dh2 + (long long)q.M * H2 + (long long)m * H2;
        float dus[NA];
        _Pragma("unroll") for (int a = 0; a < NA; ++a) if (a < A) dus[a] = sdu[sm][a];
        const float dvs = sdv[sm];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = part + 8 * i;
            if (f >= nf) continue;
            f32x4 gp, gv;
            if constexpr (NA == 2) {
                const f32x4 wv = *(const f32x4*)(sWv + 4 * f);
                const f32x4 wa = *(const f32x4*)(sWm + 8 * f), wb = *(const f32x4*)(sWm + 8 * f + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float w0 = e < 2 ? wa[2 * e] : wb[2 * e - 4], w1 = e < 2 ? wa[2 * e + 1] : wb[2 * e - 3];
                    float sa = dus[0] * w0;
                    sa += dus[NA > 1 ? 1 : 0] * w1;
                    gp[e] = hp[i][e] > 0.f ? sa : 0.f;
                    gv[e] = hv[i][e] > 0.f ? dvs * wv[e] : 0.f;
                }
            } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * f + e;
                float sa = 0.f;
                _Pragma("unroll") for (int a = 0; a < NA; ++a) if (a < A) sa += dus[a] * sWm[j * A + a];
                gp[e] = hp[i][e] > 0.f ? sa : 0.f;
                gv[e] = hv[i][e] > 0.f ? dvs * sWv[j] : 0.f;
            }
            }
            *(f32x4*)(dh2p + 4 * f) = gp;
            *(f32x4*)(dh2v + 4 * f) = gv;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// input gradient of layer 2: grid (ceil(H1 / 32), 2 nets, ceil(M / 32)); block = a 32 x 32 tile of
//   dh1[m, k] = (sum_n dh2[m, n] W2[k, n]) * relu'(h1[m, k])      the four waves split the 300 columns, partial tiles meet in LDS
// (its own launch: the weight-gradient kernel below updates W2 in place, this one still reads it)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ppo_dh1_kernel(const PpoFusedParams q) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lgrp = lane >> 5;
    const int net = blockIdx.y, H1 = q.H1, H2 = q.H2, M = q.M, k0 = blockIdx.x * 32, m0 = blockIdx.z * 32;
    const float* __restrict__ dh2 = q.dh2 + (long long)net * M * H2;
    const float* __restrict__ W2 = q.theta + pf_off(q, net, 2);
    const float* __restrict__ h1 = q.h1 + (long long)net * M * H1;
    float* dh1 = q.dh1 + (long long)net * M * H1;
    const int nsteps = (H2 + 7) / 8, per = (nsteps + 3) / 4;
    const int sb = wave * per, se = min(nsteps, sb + per);
    const int kk = k0 + lrow;
    const __amdgpu_buffer_rsrc_t rsG = PF_RSRC(dh2, (long long)M * H2 * 4), rsW = PF_RSRC(W2, (long long)H1 * H2 * 4), rsH = PF_RSRC(h1, (long long)M * H1 * 4);
    const unsigned grow = (unsigned)min(m0 + lrow, M - 1) * (unsigned)H2, wrow = (unsigned)min(kk, H1 - 1) * (unsigned)H2, H1u = (unsigned)H1;
    const unsigned col = kk < H1 ? (unsigned)kk * 4u : PF_OOB;
    float hmask[16];                                      // relu'(h1) of this lane's outputs, requested with the operands
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hmask[r] = pf_ld(rsH, (unsigned)(m0 + pf_row(r, lgrp)) * H1u * 4u + col);
    }
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    pf_mma_chunked<10>(sb, se,                            // H2 % 4 == 0; the reduction index past H2 must read zeros on one side: the W2 row is cut off there
        [&](int s_) { return pf_ld4(rsG, (grow + (unsigned)(s_ * 8 + lgrp * 4)) * 4u); },
        [&](int s_) { const int n = s_ * 8 + lgrp * 4; return pf_ld4(rsW, n < H2 ? (wrow + (unsigned)n) * 4u : PF_OOB); }, acc);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const __amdgpu_buffer_rsrc_t rsO = PF_RSRC(q.dh1 + (long long)net * M * H1, (long long)M * H1 * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = ((acc[r] + red[0][r][lane]) + (red[1][r][lane] + red[2][r][lane]));
            pf_st(rsO, (unsigned)(m0 + pf_row(r, lgrp)) * H1u * 4u + col, hmask[r] > 0.f ? v : 0.f);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// every weight / bias gradient of both nets (+ single rank: its TF-Adam update) as independent 32 x 32 tiles, one WAVE per tile:
//   dW[k, n] = sum_m X[m, k] G[m, n]        X / G = h1 / dh2 (layer 2), states / dh1 (layer 1), h2 / dhead (head: n < A)
//   tiles of a layer's first 32 rows also produce the bias gradient (column sums of G: one more MFMA per step against all-ones rows)
// grid ceil(tiles / 4) blocks of 4 waves; the last block's spare wave finalises the loss scalars and logstd.  All rows of the minibatch are
// summed inside the wave (M <= 256 on this path), so the update is applied to the tile straight from the accumulators.
// ---------------------------------------------------------------------------------------------------------------------
template <bool FUSE>
__global__ __launch_bounds__(256) void ppo_wgrad_kernel(const PpoFusedParams q) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lrow = lane & 31, lgrp = lane >> 5;
    const int H1 = q.H1, H2 = q.H2, M = q.M;
    const int KT1 = (q.kin + 31) / 32, NT1 = (H1 + 31) / 32, KT2 = NT1, NT2 = (H2 + 31) / 32, KTH = NT2;
    const int per_net = KT2 * NT2 + KT1 * NT1 + KTH;
    int t = blockIdx.x * 4 + wave;
    const bool split = !FUSE && q.m_chunk > 0;            // large minibatches: blockIdx.y = row chunk
    float* const gb = split ? q.gslab + (long long)blockIdx.y * q.gslab_stride : q.grads;      // where this wave's gradients go: the chunk's slab (every element of a slab
                                                                                               // is written by exactly one wave; an ordered pass adds the slabs) or the buffer
    if (t == 2 * per_net && blockIdx.y != 0) return;
    if (t == 2 * per_net) {                               // spare wave: loss scalars + logstd (fixed block order: deterministic)
        // lane k sums column k of the per-block partials (one load chain per lane instead of PF_NPART chains on lane 0: this wave's latency was
        // the kernel's duration), lane 0 collects them with shuffles; exp / log through the hardware units (arguments O(1), ~1e-7 relative)
        float col = 0.f;
        if (lane < PF_NPART) for (int b = 0; b < q.n_loss_blocks; ++b) col += q.partial[(long long)b * PF_NPART + lane];
        float sum[PF_NPART];
#pragma unroll
        for (int k = 0; k < PF_NPART; ++k) sum[k] = __shfl(col, k, 64);
        const float ls_l = lane < q.A ? q.theta[q.off[6] + lane] : 0.f;      // logstd[a] on lane a
        const float sd_l = __expf(ls_l);
        float ent = 0.f, sd[PF_MAX_ACT];
#pragma unroll
        for (int a = 0; a < PF_MAX_ACT; ++a) { sd[a] = __shfl(sd_l, a, 64); if (a < q.A) ent += 0.5f + PF_HALF_LOG_2PI + __logf(sd[a]); }
        if (lane != 0) return;
        const float pl = sum[0] * q.inv_m, vl = sum[1] * q.inv_m * q.value_scale, el = ent * q.entropy_scale;
        float* L = q.losses;                               // [0..4] policy, value, entropy, total, mean ratio ; [5..5+A) mean action_mean ; [5+A..5+2A) std
        L[0] = pl; L[1] = vl; L[2] = el; L[3] = -pl + vl - el; L[4] = sum[2] * q.inv_m;
#pragma unroll
        for (int a = 0; a < PF_MAX_ACT; ++a) if (a < q.A) { L[5 + a] = sum[3 + PF_MAX_ACT + a] * q.inv_m; L[5 + q.A + a] = sd[a]; }
        // the entropy term is state independent: under data parallelism grad_scale = local_M / global_M shares it across the ranks
#pragma unroll
        for (int a = 0; a < PF_MAX_ACT; ++a) if (a < q.A) pf_emit<FUSE>(q, q.off[6] + a, sum[3 + a] - q.entropy_scale * q.grad_scale, gb);
        return;
    }
    if (t > 2 * per_net) return;
    const int net = t >= per_net ? 1 : 0;
    t -= net * per_net;
    // ---- which tile ----
    const float* __restrict__ X; const float* __restrict__ G; int ldx, ldg, Kx, Ng, ldw, kt, nt, kvalid; long long oW, ob;
    if (t < KT2 * NT2) {                                  // layer 2
        kt = t / NT2; nt = t - kt * NT2;
        X = q.h1 + (long long)net * M * H1; ldx = H1; Kx = H1; G = q.dh2 + (long long)net * M * H2; ldg = H2; Ng = H2;
        oW = pf_off(q, net, 2); ob = pf_off(q, net, 3); ldw = H2; kvalid = H1;
    } else if (t < KT2 * NT2 + KT1 * NT1) {               // layer 1 (rows din .. kin-1 of W1 are the zero padding: their gradient stays 0)
        t -= KT2 * NT2; kt = t / NT1; nt = t - kt * NT1;
        X = q.row_idx ? q.s_gath : q.states; ldx = q.din; Kx = q.din; G = q.dh1 + (long long)net * M * H1; ldg = H1; Ng = H1;
        oW = pf_off(q, net, 0); ob = pf_off(q, net, 1); ldw = H1; kvalid = q.kin;
    } else {                                              // head kernel [H2, A]
        kt = t - KT2 * NT2 - KT1 * NT1; nt = 0;
        X = q.h2 + (long long)net * M * H2; ldx = H2; Kx = H2; const int A = net == 0 ? q.A : 1;
        G = net == 0 ? q.du : q.dv; ldg = A; Ng = A;
        oW = pf_off(q, net, 4); ob = pf_off(q, net, 5); ldw = A; kvalid = H2;
    }
    const int k = kt * 32 + lrow, n = nt * 32 + lrow;
    const bool nok = n < Ng;
    // descriptors: X / G end with the minibatch (rows past M read 0.0), a column past the tensor is sent past the descriptor; the weight tile's
    // descriptor ends at row Kx, so the optimiser state of rows past it reads 0.0 and their stores are dropped
    const __amdgpu_buffer_rsrc_t rsX = PF_RSRC(X, (long long)M * ldx * 4), rsG = PF_RSRC(G, (long long)M * ldg * 4);
    const unsigned xcol = k < Kx ? (unsigned)k * 4u : PF_OOB, gcol = nok ? (unsigned)n * 4u : PF_OOB;
    const unsigned ldxb = (unsigned)ldx * 4u, ldgb = (unsigned)ldg * 4u, ldwb = (unsigned)ldw * 4u;
    const long long wbytes = (long long)Kx * ldw * 4;
    const __amdgpu_buffer_rsrc_t rsP = PF_RSRC(q.theta + oW, wbytes), rsM = PF_RSRC(q.adam_m + oW, wbytes), rsV = PF_RSRC(q.adam_v + oW, wbytes);
    // the tile's optimiser state is requested together with the operands (FUSE): one memory round trip for the whole wave
    float pw[16], pm[16], pv[16];
    if constexpr (FUSE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned off = (unsigned)(kt * 32 + pf_row(r, lgrp)) * ldwb + gcol;
            pw[r] = pf_ld(rsP, off); pm[r] = pf_ld(rsM, off); pv[r] = pf_ld(rsV, off);
        }
    }
    f32x16_t acc, accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }
    const int m_beg = split ? (int)blockIdx.y * q.m_chunk : 0, m_end = split ? min(M, m_beg + q.m_chunk) : M;
    const int msteps = (m_end + 7) / 8;
    for (int s0 = m_beg / 8; s0 < msteps; s0 += 4) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned mm = (unsigned)((s0 + u) * 8 + lgrp * 4 + e);
                a[u][e] = pf_ld(rsX, mm * ldxb + xcol);
                b[u][e] = pf_ld(rsG, mm * ldgb + gcol);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            pf_mma4(a[u], b[u], acc);
            if (kt == 0) pf_mma4(f32x4{1.f, 1.f, 1.f, 1.f}, b[u], accb);          // all-ones rows: every output row = column sums of G
        }
    }
    if (!nok) return;
    if constexpr (FUSE) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned off = (unsigned)(kt * 32 + pf_row(r, lgrp)) * ldwb + gcol;
            float mm_ = pm[r], vv_ = pv[r];
            mm_ += (acc[r] - mm_) * q.omb1; vv_ += (acc[r] * acc[r] - vv_) * q.omb2;
            pf_st(rsM, off, mm_); pf_st(rsV, off, vv_);
            pf_st(rsP, off, pw[r] - (mm_ * q.alpha) / (sqrtf(vv_) + q.epsilon));
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = kt * 32 + pf_row(r, lgrp);
            const long long idx = oW + (long long)kr * ldw + n;
            if (kr < Kx) gb[idx] = acc[r];
            else if (kr < kvalid) gb[idx] = 0.f;
        }
    }
    if (kt == 0 && lgrp == 0) pf_emit<FUSE>(q, ob + n, accb[0], gb);
}

// ---------------------------------------------------------------------------------------------------------------------
// heads of PPO.predict (ppo.py:47,58-62,231-251) on the trunks of nets 0 / 1: action = clip(mean + exp(logstd) noise, low, high) or the mean,
// value = h2_v Wv + bv.  with logp_out: log pi(a | s) of GIVEN actions under net `lp_net` (0 policy, 2 old policy) instead (the cache of
// log pi_old for a whole horizon batch).  grid ceil(M / 32) blocks; 8 threads per sample.
// ---------------------------------------------------------------------------------------------------------------------
// LG: log2 of the threads per sample: 3 (32 samples per block) for batches, 6 (a wave per sample: 5 instead of 38 dependent steps over the 300 columns)
// for the handful of samples of an interactive predict
template <int NA, int LG>
__global__ __launch_bounds__(256) void ppo_predict_head_kernel(const PpoFusedParams q, const float* __restrict__ noise, int greedy,
                                                               float* __restrict__ action, float* __restrict__ value, float* __restrict__ logp_out, int lp_net) {
    constexpr int TPS = 1 << LG, SPB = 256 / TPS;
    const int tid = threadIdx.x, m0 = blockIdx.x * SPB, A = q.A, H2 = q.H2;
    const int sm = tid >> LG, part = tid & (TPS - 1), m = m0 + sm;
    const float* th = lp_net == 2 ? q.theta_old : q.theta;
    const float* Wm = th + q.off[4]; const float* bm = th + q.off[5]; const float* logstd = th + q.off[6];
    const float* Wv = q.theta + q.off[11]; const float* bv = q.theta + q.off[12];
    const float* h2p = q.h2 + (long long)(logp_out ? (lp_net == 2 ? 2 : 0) : 0) * q.M * H2; const float* h2v = q.h2 + (long long)q.M * H2;
    float au[NA], av = 0.f;
#pragma unroll
    for (int a = 0; a < NA; ++a) au[a] = 0.f;
    if (m < q.M) {
        for (int j = part; j < H2; j += TPS) {
            const float hp = h2p[(long long)m * H2 + j];
            if (!logp_out) av += h2v[(long long)m * H2 + j] * Wv[j];
            _Pragma("unroll") for (int a = 0; a < NA; ++a) if (a < A) au[a] += hp * Wm[(long long)j * A + a];
        }
    }
#pragma unroll
    for (int o = TPS / 2; o > 0; o >>= 1) {
        av += __shfl_xor(av, o, 64);
#pragma unroll
        for (int a = 0; a < NA; ++a) au[a] += __shfl_xor(au[a], o, 64);
    }
    if (part != 0 || m >= q.M) return;
    float lp = 0.f;
    _Pragma("unroll") for (int a = 0; a < NA; ++a) if (a < A) {
        const float lo = q.low[a], hi = q.high[a];
        const float mean = lo + ((tanhf(au[a] + bm[a]) + 1.0f) * 0.5f) * (hi - lo);
        if (logp_out) {
            const float sigma = expf(logstd[a]);
            const float z = (q.actions[(long long)m * A + a] - mean) / sigma;
            lp += -0.5f * z * z - (PF_HALF_LOG_2PI + logf(sigma));
        } else {
            if (q.mean_out) q.mean_out[(long long)m * A + a] = mean;
            float act = mean;
            if (!greedy) act = fminf(fmaxf(mean + expf(logstd[a]) * noise[(long long)m * A + a], lo), hi);
            action[(long long)m * A + a] = act;
        }
    }
    if (logp_out) logp_out[m] = lp; else value[m] = av + bv[0];
}

}  // namespace mi

using namespace mi;

// shapes the fused kernels are built for (checked BEFORE anything is launched; the engine falls back to the per-layer path otherwise)
bool mi_ppo_fused_shape_in_range(int A, int H2, int kin) { return A >= 1 && A <= PF_MAX_ACT && H2 <= PF_H2MAX && H2 % 4 == 0 && kin <= 96; }
static int pf_check_shape(const PpoFusedParams& q, const char* who) {
    if (mi_ppo_fused_shape_in_range(q.A, q.H2, q.kin)) return MI_OK;
    static thread_local char msg[160];
    snprintf(msg, sizeof(msg), "%s: shape outside the fused kernels' range (1 <= num_actions <= 8, H2 <= 320 and a multiple of 4, inputs <= 96)", who);
    return mi_fail(MI_ERR_SHAPE, msg);
}

int mi_ppo_fused_predict(hipStream_t st, PpoFusedParams& q, const float* noise, int greedy, float* action, float* value) {
    { const int rc0 = pf_check_shape(q, "ppo fused predict"); if (rc0 != MI_OK) return rc0; }
    q.n_nets = 2;
    int rc = mi_ppo_fused_trunks(st, q);
    if (rc != MI_OK) return rc;
    if (q.M <= 8) {                                       // a wave per sample
        if (q.A <= 2) hipLaunchKernelGGL((ppo_predict_head_kernel<2, 6>), dim3((q.M + 3) / 4), dim3(256), 0, st, q, noise, greedy, action, value, (float*)nullptr, 0);
        else hipLaunchKernelGGL((ppo_predict_head_kernel<PF_MAX_ACT, 6>), dim3((q.M + 3) / 4), dim3(256), 0, st, q, noise, greedy, action, value, (float*)nullptr, 0);
    } else if (q.A <= 2) hipLaunchKernelGGL((ppo_predict_head_kernel<2, 3>), dim3((q.M + 31) / 32), dim3(256), 0, st, q, noise, greedy, action, value, (float*)nullptr, 0);
    else hipLaunchKernelGGL((ppo_predict_head_kernel<PF_MAX_ACT, 3>), dim3((q.M + 31) / 32), dim3(256), 0, st, q, noise, greedy, action, value, (float*)nullptr, 0);
    return mi_check_launch("ppo_predict_head");
}

// log pi_old(a | s) of M samples (theta_old's trunk in net slot 2)
int mi_ppo_fused_logp_old(hipStream_t st, PpoFusedParams& q, float* out) {
    // only net 2 is needed: run the trunk kernels over the grid's net range [2, 3) by offsetting nothing -- the kernels index nets by blockIdx.y,
    // so all three are computed (M x 1.2 MFLOP, once per horizon batch)
    { const int rc0 = pf_check_shape(q, "ppo fused logp_old"); if (rc0 != MI_OK) return rc0; }
    q.n_nets = 3;
    int rc = mi_ppo_fused_trunks(st, q);
    if (rc != MI_OK) return rc;
    if (q.A <= 2) hipLaunchKernelGGL((ppo_predict_head_kernel<2, 3>), dim3((q.M + 31) / 32), dim3(256), 0, st, q, (const float*)nullptr, 1, (float*)nullptr, (float*)nullptr, out, 2);
    else hipLaunchKernelGGL((ppo_predict_head_kernel<PF_MAX_ACT, 3>), dim3((q.M + 31) / 32), dim3(256), 0, st, q, (const float*)nullptr, 1, (float*)nullptr, (float*)nullptr, out, 2);
    return mi_check_launch("ppo_logp_old");
}

int mi_ppo_fused_partial_floats(int M) { return ((M + 31) / 32) * PF_NPART; }

// forward only (PPO.predict, the cache of log pi_old): layers 1-2 of `n_nets` nets, then the caller's head kernel
int mi_ppo_fused_trunks(hipStream_t st, const PpoFusedParams& q) {
    const dim3 g1((q.H1 + 127) / 128, q.n_nets, (q.M + 31) / 32), g2((q.H2 + 31) / 32, q.n_nets, (q.M + 31) / 32);
    hipLaunchKernelGGL(ppo_l1_kernel, g1, dim3(256), 0, st, q);
    hipLaunchKernelGGL(ppo_l2_kernel, g2, dim3(256), 0, st, q);
    return mi_check_launch("ppo_fused_trunks");
}

// Measurement aid (round 5; profiles/r05_ppo.md): MI355_PPO_PAD=n issues n trivial one-wave launches behind layer 2 and n behind the layer-1
// input gradient -- 2 n extra dependent kernel boundaries on the step's stream.  (wall(n) - wall(0)) / 2 n is what ONE boundary costs inside THIS launch chain on this
// box, i.e. the most a fusion that removes a boundary can win before it has paid for any recomputation.  Default 0: nothing is launched.
// Compiled only with -DMI355_PPO_PAD_PROBE (HIPCC_EXTRA=-DMI355_PPO_PAD_PROBE python -m mi355.build --force): the production step carries no measurement code.
#ifdef MI355_PPO_PAD_PROBE
__global__ void ppo_boundary_probe_kernel(float* sink) { if (threadIdx.x == 1024) sink[0] = 0.f; }
static int ppo_pad_launches() { static int n = -1; if (n < 0) { const char* e = getenv("MI355_PPO_PAD"); n = e ? atoi(e) : 0; if (n < 0 || n > 64) n = 0; } return n; }
static int ppo_pad(hipStream_t st, float* sink) {
    const int pad = ppo_pad_launches();
    for (int i = 0; i < pad; ++i) hipLaunchKernelGGL(ppo_boundary_probe_kernel, dim3(1), dim3(64), 0, st, sink);
    return pad ? mi_check_launch("ppo_boundary_probe") : MI_OK;
}
#else
static inline int ppo_pad(hipStream_t, float*) { return MI_OK; }
#endif

// the whole minibatch step; fuse_adam = 0: gradients to q.grads instead of the in-place optimiser update
int mi_ppo_fused_step(hipStream_t st, PpoFusedParams& q, int fuse_adam) {
    { const int rc0 = pf_check_shape(q, "ppo fused step"); if (rc0 != MI_OK) return rc0; }         // before the first launch
    if (fuse_adam && q.M > 256) return mi_fail(MI_ERR_ARG, "ppo fused step: the in-kernel Adam update needs the whole minibatch in one wave (M <= 256)");
    q.n_loss_blocks = (q.M + 31) / 32;
    int rc = mi_ppo_fused_trunks(st, q);
    if (rc != MI_OK) return rc;
    rc = ppo_pad(st, q.losses);
    if (rc != MI_OK) return rc;
    if (q.A == 2) hipLaunchKernelGGL(ppo_head_loss_kernel<2>, dim3(q.n_loss_blocks), dim3(256), 0, st, q);     // (action loops are compile-time unrolled)
    else hipLaunchKernelGGL(ppo_head_loss_kernel<PF_MAX_ACT>, dim3(q.n_loss_blocks), dim3(256), 0, st, q);
    hipLaunchKernelGGL(ppo_dh1_kernel, dim3((q.H1 + 31) / 32, 2, (q.M + 31) / 32), dim3(256), 0, st, q);
    rc = ppo_pad(st, q.losses);
    if (rc != MI_OK) return rc;
    const int nt1 = (q.H1 + 31) / 32, nt2 = (q.H2 + 31) / 32, kt1 = (q.kin + 31) / 32;
    const int tiles = 2 * (nt1 * nt2 + kt1 * nt1 + nt2) + 1;         // + the wave that finalises the loss scalars
    q.m_chunk = 0;
    if (fuse_adam) {
        hipLaunchKernelGGL(ppo_wgrad_kernel<true>, dim3((tiles + 3) / 4), dim3(256), 0, st, q);
    } else if (q.M > 256) {                               // row chunks of 256: every chunk stores its own gradient slab, one ordered pass adds them (no atomics: round 4)
        const int chunks = (q.M + 255) / 256;
        if (!q.gslab) return mi_fail(MI_ERR_STATE, "ppo fused step: the engine's workspace has no gradient slabs for minibatches above 256 rows");
        q.m_chunk = 256;
        // (the alignment gaps between the tensors are written by nobody: zeroed, so that the sum leaves zeros there like the memset of the gradient buffer did)
        if (hipMemsetAsync(q.gslab, 0, (size_t)chunks * q.gslab_stride * 4, st) != hipSuccess) return mi_fail(MI_ERR_LAUNCH, "ppo fused step: memset failed");
        hipLaunchKernelGGL(ppo_wgrad_kernel<false>, dim3((tiles + 3) / 4, chunks), dim3(256), 0, st, q);
        const int rc2 = mi_check_launch("ppo_fused_step");
        if (rc2 != MI_OK) return rc2;
        return mi_reduce_slabs(st, q.gslab, q.gslab_stride, chunks, q.n_params, q.grads, 1);
    } else hipLaunchKernelGGL(ppo_wgrad_kernel<false>, dim3((tiles + 3) / 4), dim3(256), 0, st, q);
    return mi_check_launch("ppo_fused_step");
}
