// elementwise.hip — wavefront-fused HBM-bound kernels of the VAE path (gfx950, wave64):
//   reparameterise + KL (fwd, bwd), BCE-with-logits loss + dlogits + deterministic row reduction,
//   TF-form Adam over the flat parameter buffer (+ bf16 shadow weights, + grad clear), column sums (bias grads),
//   sigmoid, range check, loss finalisation / on-device epoch metric accumulators.
#include "common.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"

using namespace mi;

// one launch statement instantiated for the storage type the dtype code names (TT): fp32, bf16 or split (MI_BF16X3)
#define BY_DTYPE(dtype, ...) do { \
        if ((dtype) == MI_F32) { typedef float TT; __VA_ARGS__; } \
        else if ((dtype) == MI_BF16X3) { typedef split_t TT; __VA_ARGS__; } \
        else { typedef bf16_t TT; __VA_ARGS__; } } while (0)

namespace {

// ---------------------------------------------------------------------------------------------------
// reparam + KL forward (reference vae/models.py:7-9,97-105,131-134). One wave per batch row.
//   heads : [nsplit][B][2Z] fp32 split-K partial sums of flat*[W_mean | W_logvar]
//   mean/logvar = sum_s heads + bias ; z = mean + exp(.5 logvar) * eps (sample) or mean ; kl_b = -.5 sum(1+lv-mu^2-e^lv)
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// Counter-based N(0,1) source for the reparameterisation noise and the exploration noise (the reference draws both from TensorFlow's
// unseeded stateful RNG through tfp Normal.sample, vae/models.py:101-105, ppo.py:58-60: the stream itself cannot be reproduced, only its
// distribution).  Philox4x32-10 (Salmon et al. 2011; known-answer vectors in tests/test_oracle_golden.py) keyed by the seed, counter =
// running element index; Box-Muller on the first two words.  Element i of a stream depends on (seed, offset + i) only: any launch
// geometry, any number of ranks, and a hipGraph replay (the offset lives in device memory) give the same numbers.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned long long index) {
    uint32_t r[4];
    philox4x32_10((uint32_t)index, (uint32_t)(index >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    const float u1 = ((float)(r[0] >> 8) + 1.0f) * 5.9604644775390625e-08f;       // (0, 1]: 24 random bits
    const float u2 = (float)(r[1] >> 8) * 5.9604644775390625e-08f;                // [0, 1)
    return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

// rng (device, 4 x uint64): [0] seed, [1] element offset of the next draw, [2] block ticket, [3] unused.  When eps == nullptr and sample != 0
// the kernel draws eps itself (and stores it in eps_out for the backward pass); the LAST block to finish advances the offset by B * Z, after
// every block has read it (each block takes its ticket after its reads), so the next launch -- or the next replay of a captured graph --
// continues the stream.
template <typename T, int U = 8>
__global__ void reparam_kl_fwd_kernel(const float* __restrict__ heads, int nsplit, const float* __restrict__ bias_mean,
                                      const float* __restrict__ bias_lv, const float* __restrict__ eps, int sample,
                                      int B, int Z, float* __restrict__ mean, float* __restrict__ logvar,
                                      T* __restrict__ z, float* __restrict__ kl_row,
                                      unsigned long long* rng, float* __restrict__ eps_out) {
    const int row = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    const int lane = threadIdx.x & 63;
    const bool draw = sample && !eps && rng;
    unsigned long long seed = 0, off = 0;
    if (draw) { seed = rng[0]; off = __hip_atomic_load(&rng[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (row < B) {
        float klacc = 0.f;
        for (int j = lane; j < Z; j += WAVE) {
            float mu = bias_mean[j], lv = bias_lv[j];
            for (int s0 = 0; s0 < nsplit; s0 += U) {       // U (eight) slabs requested together, added in slab order (a plain loop pays one latency per slab)
                float a[U], b[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool ok = s0 + u < nsplit;
                    const float* h = heads + ((long long)(ok ? s0 + u : 0) * B + row) * (2 * Z);
                    a[u] = ok ? h[j] : 0.f; b[u] = ok ? h[Z + j] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) { mu += a[u]; lv += b[u]; }
            }
            mean[(long long)row * Z + j] = mu;
            logvar[(long long)row * Z + j] = lv;
            float zz = mu;
            if (sample) {
                float e;
                if (draw) { e = philox_normal(seed, off + (unsigned long long)row * Z + j); eps_out[(long long)row * Z + j] = e; }
                else e = eps[(long long)row * Z + j];
                zz = mu + expf(0.5f * lv) * e;
            }
            z[(long long)row * Z + j] = Elem<T>::from_f32(zz);
            klacc += 1.0f + lv - mu * mu - expf(lv);
        }
        klacc = wave_sum(klacc);
        if (lane == 0) kl_row[row] = -0.5f * klacc;
    }
    if (draw) {
        __syncthreads();                                  // every wave of this block has read the offset
        if (threadIdx.x == 0) {
            const unsigned long long t = atomicAdd(&rng[2], 1ull);
            if (t + 1 == gridDim.x) {                     // last block: nobody reads the offset of this launch any more
                __hip_atomic_store(&rng[1], off + (unsigned long long)B * Z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&rng[2], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// standalone draw (PPO exploration noise, tests): out[i] = N(0,1) element (offset + i) of stream `seed`
__global__ __launch_bounds__(256) void normal_philox_kernel(unsigned long long seed, unsigned long long offset, float* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = philox_normal(seed, offset + (unsigned long long)i);
}

// reparam + KL backward: dz = sum_s dzs ; dmu = dz + beta*mu*inv_b*act ; dlv = dz*eps*.5*exp(.5lv) + beta*.5*(e^lv-1)*inv_b*act
// act = 1 unless kl_tolerance clamps the row (tf.maximum passes the gradient to kl_b when kl_b >= tol*Z).
template <typename T, int U = 8>
__global__ void reparam_kl_bwd_kernel(const float* __restrict__ dzs, int nsplit, const float* __restrict__ mean,
                                      const float* __restrict__ logvar, const float* __restrict__ eps,
                                      const float* __restrict__ kl_row, float beta, float kl_floor, float inv_b,
                                      int B, int Z, T* __restrict__ dheads) {
    const int row = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float act = (kl_floor > 0.f && kl_row[row] < kl_floor) ? 0.f : 1.f;
    for (int j = lane; j < Z; j += WAVE) {
        float dz = 0.f;
        for (int s0 = 0; s0 < nsplit; s0 += U) {           // U (eight) slabs requested together, added in slab order
            float a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = s0 + u < nsplit ? dzs[((long long)(s0 + u) * B + row) * Z + j] : 0.f;
#pragma unroll
            for (int u = 0; u < U; ++u) dz += a[u];
        }
        const float mu = mean[(long long)row * Z + j], lv = logvar[(long long)row * Z + j];
        const float e = eps[(long long)row * Z + j];
        const float dmu = dz + beta * mu * inv_b * act;
        const float dlv = dz * e * 0.5f * expf(0.5f * lv) + beta * 0.5f * (expf(lv) - 1.0f) * inv_b * act;
        dheads[(long long)row * (2 * Z) + j] = Elem<T>::from_f32(dmu);
        dheads[(long long)row * (2 * Z) + Z + j] = Elem<T>::from_f32(dlv);
    }
}

// ---------------------------------------------------------------------------------------------------
// reconstruction loss + dlogits (reference vae/models.py:11-22,123-128).  grid = (chunks, B), 256 threads x 8 px.
//   kind 0: bce  max(x,0) - x*y + log1p(exp(-|x|))          d/dx = sigmoid(x) - y
//   kind 1: bce_v2  -(y log(1e-10+s) + (1-y) log(1e-10+1-s))  d/dx = (-y/(1e-10+s) + (1-y)/(1e-10+1-s)) s(1-s)
//   kind 2: mse  (y - s)^2                                    d/dx = 2 (s-y) s (1-s)
// Row partial sums go to partial[b][chunk] (fixed order -> deterministic), dlogits are pre-scaled by inv_b.
// ---------------------------------------------------------------------------------------------------
// One thread owns 24 consecutive pixels-times-channels of one frame (three 16-byte logit vectors in bf16): 24 is a multiple of
// every target depth on this path (3 = rgb, 1 = segmentation), so element j of a thread always belongs to channel j % Ct and
// the BiasAddGrad of the last transposed conv (sum of dlogits per channel) folds into the same pass.  exp / log go to the
// hardware transcendental units (v_exp_f32 / v_log_f32): the per-pixel terms are O(1), their absolute error ~1e-7.
constexpr int BCE_PER_THREAD = 24;
constexpr int BCE_CHUNK = 256 * BCE_PER_THREAD;

// TL = float: labels in [0, 1]; TL = unsigned char (round 5): raw camera bytes, float32(k) / float32(255) formed exactly in registers (u8_to_unit_exact)
template <typename T, typename TL>
__global__ __launch_bounds__(256) void recon_loss_kernel(const T* __restrict__ logits, const TL* __restrict__ labels,
                                                         const int* __restrict__ frame_idx, long long label_stride, int P,
                                                         int kind, float inv_b, T* __restrict__ dlogits,
                                                         float* __restrict__ partial, int nchunks, int Ct, float* __restrict__ dbias) {
    constexpr int VE = 16 / (int)sizeof(T);               // logits per 16-byte vector
    constexpr int NV = BCE_PER_THREAD / VE;
    const int b = blockIdx.y, chunk = blockIdx.x;
    const long long fr = frame_idx ? (long long)frame_idx[b] : (long long)b;
    const T* x = logits + (long long)b * P;
    const TL* y = labels + fr * label_stride;
    T* dx = dlogits ? dlogits + (long long)b * P : nullptr;
    const int e0 = chunk * BCE_CHUNK + threadIdx.x * BCE_PER_THREAD;
    const bool vec = (P % BCE_PER_THREAD) == 0 && ((((uintptr_t)x) | ((uintptr_t)dx)) & 15) == 0 && (((uintptr_t)y) & (sizeof(TL) == 1 ? 7 : 15)) == 0;   // whole groups, aligned rows
    float acc = 0.f;
    float gs[BCE_PER_THREAD];
#pragma unroll
    for (int j = 0; j < BCE_PER_THREAD; ++j) gs[j] = 0.f;
    if (e0 < P) {
        float xv[BCE_PER_THREAD], yv[BCE_PER_THREAD];
        if (vec) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const PackN<T, VE> t = *(const PackN<T, VE>*)(x + e0 + v * VE);
#pragma unroll
                for (int k = 0; k < VE; ++k) xv[v * VE + k] = Elem<T>::to_f32(t.v[k]);
            }
            if constexpr (sizeof(TL) == 1) {                // 24 label bytes = three 8-byte loads (e0 is a multiple of 24, the row 8-byte aligned)
#pragma unroll
                for (int v = 0; v < BCE_PER_THREAD / 8; ++v) {
                    const unsigned long long w = *(const unsigned long long*)(y + e0 + v * 8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) yv[v * 8 + k] = u8_to_unit_exact((float)((w >> (8 * k)) & 255ull));
                }
            } else {
#pragma unroll
                for (int v = 0; v < BCE_PER_THREAD / 4; ++v) {
                    const f32x4 t = *(const f32x4*)(y + e0 + v * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) yv[v * 4 + k] = t[k];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < BCE_PER_THREAD; ++j) {
                const bool in = e0 + j < P;
                xv[j] = in ? Elem<T>::to_f32(x[e0 + j]) : 0.f;
                if constexpr (sizeof(TL) == 1) yv[j] = in ? u8_to_unit_exact((float)y[e0 + j]) : 0.f; else yv[j] = in ? y[e0 + j] : 0.f;
            }
        }
        T gq[BCE_PER_THREAD];
#pragma unroll
        for (int j = 0; j < BCE_PER_THREAD; ++j) {
            float l, g;
            const float e = __expf(-fabsf(xv[j]));
            const float r = __frcp_rn(1.0f + e);
            const float s = xv[j] >= 0.f ? r : e * r;
            if (kind == 0) {
                l = fmaxf(xv[j], 0.f) - xv[j] * yv[j] + __logf(1.0f + e);
                g = s - yv[j];
            } else if (kind == 1) {
                l = -(yv[j] * __logf(1e-10f + s) + (1.0f - yv[j]) * __logf(1e-10f + 1.0f - s));
                g = (-yv[j] / (1e-10f + s) + (1.0f - yv[j]) / (1e-10f + 1.0f - s)) * s * (1.0f - s);
            } else {
                const float d = yv[j] - s;
                l = d * d;
                g = -2.0f * d * s * (1.0f - s);
            }
            const bool in = vec || e0 + j < P;
            acc += in ? l : 0.f;
            gq[j] = Elem<T>::from_f32(g * inv_b);
            gs[j] = in ? Elem<T>::to_f32(gq[j]) : 0.f;       // the bias gradient sums the STORED (rounded) values, like BiasAddGrad of dlogits
        }
        if (dx) {
            if (vec) {
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    PackN<T, VE> t;
#pragma unroll
                    for (int k = 0; k < VE; ++k) t.v[k] = gq[v * VE + k];
                    *(PackN<T, VE>*)(dx + e0 + v * VE) = t;
                }
            } else {
#pragma unroll
                for (int j = 0; j < BCE_PER_THREAD; ++j) if (e0 + j < P) dx[e0 + j] = gq[j];
            }
        }
    }
    __shared__ float red[4][4];
    acc = wave_sum(acc);
    // channel sums: Ct divides 24 and e0 is a multiple of 24, so element j is channel j % Ct (Ct <= 3 handled here)
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    if (dbias) {
#pragma unroll
        for (int j = 0; j < BCE_PER_THREAD; ++j) {
            const int c = Ct == 1 ? 0 : (Ct == 2 ? j % 2 : j % 3);
            if (c == 0) c0 += gs[j]; else if (c == 1) c1 += gs[j]; else c2 += gs[j];
        }
        c0 = wave_sum(c0); c1 = wave_sum(c1); c2 = wave_sum(c2);
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = acc; red[threadIdx.x >> 6][1] = c0; red[threadIdx.x >> 6][2] = c1; red[threadIdx.x >> 6][3] = c2; }
    __syncthreads();
    if (threadIdx.x == 0) partial[(long long)b * nchunks + chunk] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    if (dbias && threadIdx.x >= 1 && threadIdx.x <= Ct) {
        const int c = threadIdx.x;
        atomicAdd(&dbias[c - 1], (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]));
    }
}

// loss finalisation: recon = mean_b sum_chunk partial ; kl = mean_b max(kl_b, floor) ; fixed summation order.
// out[0]=recon, out[1]=kl ; metrics[0..2] += (recon, kl, 1)  (the on-device tf.metrics.mean accumulators).
// With data parallelism each rank calls this on its local rows with inv_b = 1/B_global and all-reduces out/metrics.
constexpr int FIN_NT = 1024;
__global__ __launch_bounds__(FIN_NT) void finalize_losses_kernel(const float* __restrict__ partial, int n_partial, const float* __restrict__ kl_row,
                                       float kl_floor, int B, float inv_b, float* __restrict__ out,
                                       float* __restrict__ metrics, float metric_weight,
                                       const float* __restrict__ bpart, int n_bpart, int channels, float* __restrict__ dbias) {
    // one block, fixed assignment and fixed tree: deterministic.  Up to ~13k block partials of the fused decoder tail: 1024 threads and
    // 4 independent loads in flight per thread keep this at a few microseconds.
    __shared__ float sm[5][FIN_NT];
    const int t = threadIdx.x;
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, k = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
    int i = t;
    for (; i + 3 * FIN_NT < n_partial; i += 4 * FIN_NT) { r0 += partial[i]; r1 += partial[i + FIN_NT]; r2 += partial[i + 2 * FIN_NT]; r3 += partial[i + 3 * FIN_NT]; }
    for (; i < n_partial; i += FIN_NT) r0 += partial[i];
    for (int b = t; b < B; b += FIN_NT) k += fmaxf(kl_row[b], kl_floor > 0.f ? kl_floor : -3.0e38f);
    if (bpart) {
        const f32x4* bp = (const f32x4*)bpart;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b2 = {0.f, 0.f, 0.f, 0.f};
        int j = t;
        for (; j + FIN_NT < n_bpart; j += 2 * FIN_NT) { a += bp[j]; b2 += bp[j + FIN_NT]; }
        for (; j < n_bpart; j += FIN_NT) a += bp[j];
        a += b2; c0 = a[0]; c1 = a[1]; c2 = a[2];
    }
    sm[0][t] = (r0 + r1) + (r2 + r3); sm[1][t] = k; sm[2][t] = c0; sm[3][t] = c1; sm[4][t] = c2;
    __syncthreads();
    for (int o = FIN_NT / 2; o > 0; o >>= 1) {
        if (t < o) {
#pragma unroll
            for (int q = 0; q < 5; ++q) sm[q][t] += sm[q][t + o];
        }
        __syncthreads();
    }
    if (t == 0) {
        const float recon = sm[0][0] * inv_b, kl = sm[1][0] * inv_b;
        out[0] = recon; out[1] = kl;
        if (metrics) { metrics[0] += recon; metrics[1] += kl; metrics[2] += metric_weight; }
        if (bpart && dbias) for (int c = 0; c < channels && c < 3; ++c) dbias[c] += sm[2 + c][0];
    }
}

// ---------------------------------------------------------------------------------------------------
// TF ApplyAdam over a flat fp32 buffer:  m += (g-m)(1-b1); v += (g*g-v)(1-b2); p -= m*alpha/(sqrt(v)+eps)
// alpha = lr*sqrt(1-b2^t)/(1-b1^t) comes from the host (fp32).  Optionally refreshes the bf16 shadow weights the
// MFMA kernels read and clears the gradient for the next step's atomics (saves a memset + a cast pass).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_tf_update(float& pp, float& mm, float& vv, const float gg, const float alpha, const float omb1, const float omb2, const float epsilon) {
    mm += (gg - mm) * omb1;
    vv += (gg * gg - vv) * omb2;
    pp -= (mm * alpha) / (sqrtf(vv) + epsilon);
}

__global__ __launch_bounds__(256) void adam_tf_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                      float* __restrict__ g, long long n, float alpha_arg, const float* __restrict__ alpha_dev, float omb1,
                                                      float omb2, float epsilon, void* __restrict__ shadow_any, int shadow_split, int clear_grad) {
    bf16_t* __restrict__ shadow = shadow_split ? nullptr : (bf16_t*)shadow_any;           // the weight copy the MFMA kernels read: bf16, or
    split_t* __restrict__ shadow_s = shadow_split ? (split_t*)shadow_any : nullptr;       // split storage (hi | lo halves, 4 bytes per weight)
    const float alpha = alpha_dev ? alpha_dev[0] : alpha_arg;      // device-resident step size: a captured step is replayed with a new value
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = ((f32x4*)p)[i], mv = ((f32x4*)m)[i], vv = ((f32x4*)v)[i], gv = ((f32x4*)g)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pp = pv[e], mm = mv[e], v1 = vv[e];
            adam_tf_update(pp, mm, v1, gv[e], alpha, omb1, omb2, epsilon);
            pv[e] = pp; mv[e] = mm; vv[e] = v1;
        }
        ((f32x4*)p)[i] = pv; ((f32x4*)m)[i] = mv; ((f32x4*)v)[i] = vv;
        if (clear_grad) { f32x4 zz = {0.f, 0.f, 0.f, 0.f}; ((f32x4*)g)[i] = zz; }
        if (shadow) {
            u16x4 s;
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = f32_to_bf16(pv[e]);
            ((u16x4*)shadow)[i] = s;
        }
        if (shadow_s) {
            const float pf[4] = {pv[0], pv[1], pv[2], pv[3]};
            *(PackN<split_t, 4>*)(shadow_s + 4 * i) = pack4<split_t>(pf);
        }
    }
    // tail (n not a multiple of 4)
    for (long long i = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gg = g[i], mm = m[i], vv = v[i], pp = p[i];
        adam_tf_update(pp, mm, vv, gg, alpha, omb1, omb2, epsilon);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (clear_grad) g[i] = 0.f;
        if (shadow) shadow[i] = f32_to_bf16(pp);
        if (shadow_s) shadow_s[i] = split_from_f32(pp);
    }
}

__global__ __launch_bounds__(256) void cast_f32_split_kernel(const float* __restrict__ src, split_t* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = split_from_f32(src[i]);
}
__global__ __launch_bounds__(256) void cast_split_f32_kernel(const split_t* __restrict__ src, float* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = split_to_f32(src[i]);
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = f32_to_bf16(src[i]);
}

// ---------------------------------------------------------------------------------------------------
// TF ApplyAdam that also writes BOTH weight layouts the MFMA kernels read (round 4, the MlpVAE engine: 39.5 M weights -- the separate transpose pass behind Adam
// re-read 158 MB and wrote 79 MB per step at 2.8 TB/s): every [K, N] kernel of the flat buffer is walked in 64 x 64 tiles -- p / m / v / g as 16-byte vectors along
// N (256 contiguous bytes per tile row), the storage-type copy in the same layout (`shadow`, bf16 engines), and, through an LDS transpose, the K-contiguous copy
// wt[n][k] as 16-byte vectors along K.  Everything outside the listed kernels (biases) is covered by `flat` ranges handled by the blocks behind the tile blocks.
// The arithmetic is adam_tf_update, the same inline function as adam_tf_kernel: bit-identical parameters.
// ---------------------------------------------------------------------------------------------------
constexpr int AL_MAX = 16;
struct AdamLayoutJobs {
    long long off[AL_MAX]; int K[AL_MAX], N[AL_MAX], tile0[AL_MAX + 1]; int count;          // the 2-D kernels: tiles tile0[t] .. tile0[t + 1]
    unsigned no_shadow, no_wt;                                                              // bit t: kernel t has no storage-type copy / no K-contiguous copy (nobody reads it)
    // round 5: up to two FRAGMENT-ORDERED bf16 copies per kernel for the activation-resident convolutions (ares_tile.hpp: form 0 conv form / 1 gather form of a
    // [16 x 128][256] kernel, 2 gather form of a [16 x 64][128] kernel) -- 16-byte granules of the tile this block holds in LDS anyway, written from there: the separate
    // ares_pack launch (5.6 us + a kernel boundary at the head of every step) is gone.  form < 0: none
    bf16_t* frag[AL_MAX][2]; signed char fform[AL_MAX][2];
    long long foff[AL_MAX], fn[AL_MAX]; int fblk0[AL_MAX + 1]; int fcount;                 // flat ranges: 1,024 elements per block
};

template <typename TT>
__global__ __launch_bounds__(256) void adam_tf_layouts_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, float* __restrict__ g, const AdamLayoutJobs jb,
                                                              float alpha_arg, const float* __restrict__ alpha_dev, float omb1, float omb2, float epsilon,
                                                              TT* __restrict__ shadow, TT* __restrict__ wt, int clear_grad) {
    constexpr int PITCH = sizeof(TT) == 2 ? 66 : 65;     // bf16: 33 dwords per row; fp32: 65 -- the transposed reads of a wave hit distinct banks
    __shared__ TT tile[64 * PITCH];
    const float alpha = alpha_dev ? alpha_dev[0] : alpha_arg;
    const int tid = threadIdx.x, b = (int)blockIdx.x;
    const int ntile = jb.tile0[jb.count];
    if (b >= ntile) {                                     // (block-uniform) flat ranges: biases
        const int fb = b - ntile;
        int t = 0;
#pragma unroll
        for (int i = 1; i < AL_MAX; ++i) t += (i < jb.fcount && fb >= jb.fblk0[i]) ? 1 : 0;
        const long long base = jb.foff[t], n = jb.fn[t];
        for (int j = 0; j < 4; ++j) {
            const long long i = (long long)(fb - jb.fblk0[t]) * 1024 + j * 256 + tid;
            if (i < n) {
                float gg = g[base + i], mm = m[base + i], vv = v[base + i], pp = p[base + i];
                adam_tf_update(pp, mm, vv, gg, alpha, omb1, omb2, epsilon);
                p[base + i] = pp; m[base + i] = mm; v[base + i] = vv;
                if (clear_grad) g[base + i] = 0.f;
                if (shadow) shadow[base + i] = Elem<TT>::from_f32(pp);
            }
        }
        return;
    }
    int t = 0;
#pragma unroll
    for (int i = 1; i < AL_MAX; ++i) t += (i < jb.count && b >= jb.tile0[i]) ? 1 : 0;
    const int K = jb.K[t], N = jb.N[t];
    const long long off = jb.off[t];
    const int tiles_n = (N + 63) >> 6;
    const int local = b - jb.tile0[t];
    const int k0 = (local / tiles_n) << 6, n0 = (local % tiles_n) << 6;
    const int c4 = (tid & 15) << 2, r0 = tid >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 16 * i, k = k0 + r, n = n0 + c4;
        f32x4 pv = {0.f, 0.f, 0.f, 0.f};
        if (k < K && n < N) {                             // (N % 4 == 0: a vector is inside or outside as a whole)
            const long long e = off + (long long)k * N + n;
            pv = *(const f32x4*)(p + e);
            f32x4 mv = *(const f32x4*)(m + e), vv = *(const f32x4*)(v + e);
            const f32x4 gv = *(const f32x4*)(g + e);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float pp = pv[q], mm = mv[q], v1 = vv[q];
                adam_tf_update(pp, mm, v1, gv[q], alpha, omb1, omb2, epsilon);
                pv[q] = pp; mv[q] = mm; vv[q] = v1;
            }
            *(f32x4*)(p + e) = pv; *(f32x4*)(m + e) = mv; *(f32x4*)(v + e) = vv;
            if (clear_grad) { const f32x4 zz = {0.f, 0.f, 0.f, 0.f}; *(f32x4*)(g + e) = zz; }
            if (shadow && !((jb.no_shadow >> t) & 1u)) {
                if constexpr (sizeof(TT) == 2) {
                    u16x4 sv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) sv[q] = f32_to_bf16(pv[q]);
                    *(u16x4*)(shadow + e) = sv;
                } else *(f32x4*)(shadow + e) = pv;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) tile[r * PITCH + c4 + q] = Elem<TT>::from_f32(pv[q]);
    }
    __syncthreads();
    if constexpr (sizeof(TT) == 2) {
        // fragment-ordered copies (see AdamLayoutJobs::frag; index maps = ares_pack_kernel's, inverted): 512 granules of 8 values per tile and copy, two per thread
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16_t* const fr = jb.frag[t][q];
            const int form = jb.fform[t][q];
            if (!fr || form < 0) continue;                // (block-uniform)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int gi = tid + 256 * it;
                u16x8 o;
                long long G;
                if (form == 6) {                           // (round 6) deconv3's [800][64] kernel for its input gradient (conv form, k = 5): 8 consecutive k = channels of one kernel position, column n
                    const int nl = gi & 63, kl = (gi >> 6) << 3;
                    const int n = n0 + nl, k = k0 + kl;
                    if (k >= K) continue;                  // (K = 800: the last tile is 32 rows)
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = tile[(kl + e) * PITCH + nl];
                    const int kq = k >> 5, kh = kq / 5, kw = kq - kh * 5, c = k & 31;
                    const int ta = kh >> 1, tb = kw >> 1, nph = ta < 2 ? 2 : 1, npw = tb < 2 ? 2 : 1;
                    const int ord = (ta < 2 ? 20 * ta : 40) + tb * nph * 4 + (kh & 1) * npw * 2 + (kw & 1) * 2 + (c >> 4);       // ares_pack_kernel form 6, inverted
                    G = ((long long)((n >> 5) * 50 + ord)) * 64 + ((c >> 3) & 1) * 32 + (n & 31);
                } else
                if (form == 5) {                           // (round 6) conv2's [512][64] kernel for the fused encoder head: 8 consecutive k of one column n
                    const int nl = gi & 63, kl = (gi >> 6) << 3;
                    const int n = n0 + nl, k = k0 + kl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = tile[(kl + e) * PITCH + nl];
                    G = ((long long)((n >> 5) * 32 + (k >> 4))) * 64 + ((k >> 3) & 1) * 32 + (n & 31);
                } else
                if (form == 4) {                           // (round 6) deconv3's [5][5][32][64] kernel, gather form of the register-weight kernel: 8 consecutive c of one row k = (kh * 5 + kw) * 32 + n
                    const int r = gi >> 3, nl = (gi & 7) << 3;
                    const int k = k0 + r;
                    if (k >= K) continue;                  // (K = 800: the last tile is 32 rows)
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = tile[r * PITCH + nl + e];
                    const int kq = k >> 5, kh = kq / 5, kw = kq - kh * 5, c = n0 + nl;
                    const int cls = (kh & 1) * 2 + (kw & 1), tap = (2 - (kh >> 1)) * 3 + (2 - (kw >> 1));
                    G = ((long long)((cls * 9 + tap) * 4 + (c >> 4))) * 64 + ((c >> 3) & 1) * 32 + (k & 31);
                } else
                if (form == 3) {                           // (round 6) conv form of the register-weight kernel, [1024][128]: 8 consecutive k = channels of one kernel position, column n
                    const int nl = gi & 63, kl = (gi >> 6) << 3;
                    const int n = n0 + nl, k = k0 + kl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = tile[(kl + e) * PITCH + nl];
                    const int kh = k >> 8, kw = (k >> 6) & 3, c = k & 63;
                    const int kidx = ((kh >> 1) * 2 + (kw >> 1)) * 16 + (kh & 1) * 8 + (kw & 1) * 4 + (c >> 4);
                    G = ((long long)((n >> 5) * 64 + kidx)) * 64 + ((c >> 3) & 1) * 32 + (n & 31);
                } else
                if (form == 0) {                           // 8 consecutive k of one column n: dst granule ((n >> 5) * 128 + (k >> 4)) * 64 + ((k >> 3) & 1) * 32 + (n & 31)
                    const int nl = gi & 63, kl = (gi >> 6) << 3;
                    const int n = n0 + nl, k = k0 + kl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = tile[(kl + e) * PITCH + nl];
                    G = ((long long)((n >> 5) * 128 + (k >> 4))) * 64 + ((k >> 3) & 1) * 32 + (n & 31);
                } else {                                   // 8 consecutive n of one row k = (kh * 4 + kw) * C + nt * 32 + lo
                    const int r = gi >> 3, nl = (gi & 7) << 3;
                    const int k = k0 + r, n = n0 + nl;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = tile[r * PITCH + nl + e];
                    const int csh = form == 1 ? 7 : 6;     // C = 128 | 64 rows per tap
                    const int tapi = k >> csh, cc = k & ((1 << csh) - 1);
                    const int kh = tapi >> 2, kw = tapi & 3;
                    const int cls = (kh & 1) * 2 + (kw & 1), tap = (kh >> 1) * 2 + (kw >> 1);
                    const int nt = cc >> 5, lo = cc & 31, l = ((n >> 3) & 1) * 32 + lo;
                    if (form == 1) G = ((long long)((cls * 4 + nt) * 64 + tap * 16 + (n >> 4))) * 64 + l;
                    else G = ((long long)(cls * 64 + nt * 32 + tap * 8 + (n >> 4))) * 64 + l;
                }
                *(u16x8*)(fr + G * 8) = o;
            }
        }
    }
    if (!wt || ((jb.no_wt >> t) & 1u)) return;
    // wt[off + n * K + k]: thread (n = tid >> 2, quarter = tid & 3) writes 16 consecutive k of its row
    const int nn = tid >> 2, kq = (tid & 3) << 4;
    const int n = n0 + nn;
    if (n >= N) return;
    TT* const dst = wt + off + (long long)n * K + k0 + kq;
    constexpr int VE = 16 / (int)sizeof(TT);             // elements per 16-byte store
    const bool vec_ok = ((K % VE) | (int)(off % VE)) == 0 && k0 + kq + 16 <= K;
    if (vec_ok) {
#pragma unroll
        for (int j = 0; j < 16 / VE; ++j) {
            PackN<TT, VE> o;
#pragma unroll
            for (int q = 0; q < VE; ++q) o.v[q] = tile[(kq + j * VE + q) * PITCH + nn];
            *(PackN<TT, VE>*)(dst + j * VE) = o;
        }
    } else {
        for (int q = 0; q < 16; ++q) if (k0 + kq + q < K) dst[q] = tile[(kq + q) * PITCH + nn];
    }
}

// rows idx[b] (or b) of a float32 table -> a dense [B, row_len] tensor of the engine's storage type: the MlpVAE engine's frame staging in ONE launch
// (index_select + contiguous + cast were three passes over the 79 MB minibatch)
// TS = float: a float32 table; TS = unsigned char (round 5): raw camera bytes k -> float32(k) / float32(255), correctly rounded (u8_to_unit_exact: the value the reference's
// host preprocessing produces, vae/train_vae.py:15-18; rounded once more to the storage type for bf16)
template <typename T, typename TS>
__global__ __launch_bounds__(256) void gather_rows_cast_kernel(const TS* __restrict__ src, const int* __restrict__ idx, long long row_len, T* __restrict__ out, unsigned chunks) {
    const int b = (int)(blockIdx.x / chunks);            // (row, chunk) folded into grid.x: any batch size (grid.y stops at 65535)
    const TS* s = src + (idx ? (long long)idx[b] : (long long)b) * row_len;
    T* d = out + (long long)b * row_len;
    const long long i = ((long long)(blockIdx.x % chunks) * 256 + threadIdx.x) * 8;
    if constexpr (sizeof(TS) == 1) {
        if (i + 8 <= row_len && (((uintptr_t)(s + i)) & 7) == 0 && (((uintptr_t)d) & 15) == 0) {
            const unsigned long long w = *(const unsigned long long*)(s + i);
            float fa[4], fc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { fa[q] = u8_to_unit_exact((float)((w >> (8 * q)) & 255ull)); fc[q] = u8_to_unit_exact((float)((w >> (8 * q + 32)) & 255ull)); }
            *(PackN<T, 4>*)(d + i) = pack4<T>(fa);
            *(PackN<T, 4>*)(d + i + 4) = pack4<T>(fc);
        } else {
            for (int q = 0; q < 8; ++q) if (i + q < row_len) d[i + q] = Elem<T>::from_f32(u8_to_unit_exact((float)s[i + q]));
        }
    } else {
        if (i + 8 <= row_len && ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
            const f32x4 a = *(const f32x4*)(s + i), c = *(const f32x4*)(s + i + 4);
            const float fa[4] = {a[0], a[1], a[2], a[3]}, fc[4] = {c[0], c[1], c[2], c[3]};
            *(PackN<T, 4>*)(d + i) = pack4<T>(fa);
            *(PackN<T, 4>*)(d + i + 4) = pack4<T>(fc);
        } else {
            for (int q = 0; q < 8; ++q) if (i + q < row_len) d[i + q] = Elem<T>::from_f32(s[i + q]);
        }
    }
}

// raw camera bytes -> [0, 1] frames: float32(k) / float32(255), correctly rounded (the value the reference's host preprocessing
// `frame.astype(np.float32) / 255.0` produces -- vae/train_vae.py:15-18); 16 bytes in, four 16-byte vectors out per thread step
__global__ __launch_bounds__(256) void u8_to_unit_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x * 16;
    for (long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += stride) {
        if (i + 16 <= n && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            const u32x4_t v = *(const u32x4_t*)(src + i);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = __fdiv_rn((float)((v[q] >> (8 * e)) & 0xffu), 255.0f);
                *(f32x4*)(dst + i + 4 * q) = o;
            }
        } else {
            for (long long j = i; j < n && j < i + 16; ++j) dst[j] = __fdiv_rn((float)src[j], 255.0f);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// column sums (BiasAddGrad): out[n] += sum_m x[m,n].  Two regimes: N <= 256 (flat, stride a multiple of N) and N > 256.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_small_kernel(const T* __restrict__ x, long long M, int N, int rows_per_block,
                                                           float* __restrict__ out, float* __restrict__ part, int part_stride) {
    __shared__ float cs[256];
    const int ntu = (256 / N) * N;                       // active threads: stride is a multiple of N -> fixed column per thread
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(M, r0 + (long long)rows_per_block);
    float acc = 0.f;
    if ((int)threadIdx.x < ntu) {
        for (long long i = r0 * N + threadIdx.x; i < r1 * N; i += ntu) acc += Elem<T>::to_f32(x[i]);
    }
    cs[threadIdx.x] = acc;
    __syncthreads();
    if ((int)threadIdx.x < N) {
        float s = 0.f;
        for (int t = threadIdx.x; t < ntu; t += N) s += cs[t];
        if (part) part[(long long)blockIdx.x * part_stride + threadIdx.x] = s;      // per-block partial sums, added up in a fixed order by the caller's reduce
        else atomicAdd(&out[threadIdx.x], s);
    }
}

// vectorised variant for N % VEC == 0 (VEC = 16 B of T): each thread owns a fixed group of VEC columns
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ x, long long M, int N, int rows_per_block,
                                                         float* __restrict__ out, float* __restrict__ part, int part_stride) {
    constexpr int VEC = 16 / (int)sizeof(T);
    __shared__ float cs[256][VEC + 1];
    const int gpr = N / VEC;                             // vector groups per row
    const int ntu = (256 / gpr) * gpr;                   // active threads: stride is a whole number of rows
    const long long v0 = (long long)blockIdx.x * rows_per_block * gpr;
    const long long v1 = min(M, (long long)(blockIdx.x + 1) * rows_per_block) * gpr;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if ((int)threadIdx.x < ntu) {
        // four loads in flight per thread (one dependent load per iteration ran at 1.6 TB/s on the 202 MB tensors of the split-storage engine: round 3)
        long long v = v0 + threadIdx.x;
        for (; v + 3ll * ntu < v1; v += 4ll * ntu) {
            PackN<T, VEC> t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *(const PackN<T, VEC>*)(x + (v + (long long)u * ntu) * VEC);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += Elem<T>::to_f32(t[u].v[e]);
        }
        for (; v < v1; v += ntu) {
            const PackN<T, VEC> t = *(const PackN<T, VEC>*)(x + v * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += Elem<T>::to_f32(t.v[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) cs[threadIdx.x][e] = acc[e];
    __syncthreads();
    if ((int)threadIdx.x < N) {
        const int g = threadIdx.x / VEC, e = threadIdx.x % VEC;
        float s = 0.f;
        for (int t = g; t < ntu; t += gpr) s += cs[t][e];
        if (part) part[(long long)blockIdx.x * part_stride + threadIdx.x] = s;
        else atomicAdd(&out[threadIdx.x], s);
    }
}

// K-contiguous ("transposed") copies of the [K,N] kernels: dst[off + n*K + k] = (T) src[off + k*N + n]
struct TransposeBatch { long long off[16]; int K[16]; int N[16]; int tile0[17]; int count; };

template <typename T>
__global__ __launch_bounds__(256) void transpose_weights_kernel(const float* __restrict__ src, T* __restrict__ dst, const TransposeBatch tb) {
    __shared__ float tile[32][33];
    int t = 0;
    while (t + 1 < tb.count && (int)blockIdx.x >= tb.tile0[t + 1]) ++t;
    const int K = tb.K[t], N = tb.N[t];
    const int tiles_n = (N + 31) / 32;
    const int local = blockIdx.x - tb.tile0[t];
    const int k0 = (local / tiles_n) * 32, n0 = (local % tiles_n) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    const float* s = src + tb.off[t];
    T* d = dst + tb.off[t];
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, n = n0 + tx;
        tile[r][tx] = (k < K && n < N) ? s[(long long)k * N + n] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, k = k0 + tx;
        if (n < N && k < K) d[(long long)n * K + k] = Elem<T>::from_f32(tile[tx][r]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_wide_kernel(const T* __restrict__ x, long long M, int N, int rows_per_block,
                                                          float* __restrict__ out, float* __restrict__ part, int part_stride) {
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min(M, r0 + (long long)rows_per_block);
    for (int c = blockIdx.y * 256 + threadIdx.x; c < N; c += gridDim.y * 256) {
        float acc = 0.f;
        for (long long r = r0; r < r1; ++r) acc += Elem<T>::to_f32(x[r * N + c]);
        if (part) part[(long long)blockIdx.x * part_stride + c] = acc;
        else atomicAdd(&out[c], acc);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sigmoid_kernel(const T* __restrict__ x, float* __restrict__ out, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float xv = Elem<T>::to_f32(x[i]);
        out[i] = 1.0f / (1.0f + expf(-xv));
    }
}

// verify_range (reference vae/models.py:24-30): flag[0] |= 1 if any element outside [lo, hi] or NaN
__global__ __launch_bounds__(256) void range_check_kernel(const float* __restrict__ x, long long n, float lo, float hi, int* __restrict__ flag) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    int bad = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float v = x[i];
        bad |= !(v >= lo && v <= hi);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

inline int grid_for(long long n, int per_block, int cap = 2048) {
    long long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

namespace mi {
// out[m, n] = mask(act(sum_s slabs[s][m][n] + bias[n])): the finishing pass of a split-K dense layer (MlpVAE's 38400-long reductions)
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slabs, int nsplit, long long mn, int N, const float* __restrict__ bias, int relu,
                                                            const T* __restrict__ mask, T* __restrict__ out_t, float* __restrict__ out_f) {
    const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= mn) return;
    f32x4 s = *(const f32x4*)(slabs + i4);
    for (int k = 1; k < nsplit; ++k) s += *(const f32x4*)(slabs + (long long)k * mn + i4);
    const int n = (int)(i4 % N);                          // N % 4 == 0: the four values share a row
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = s[e] + (bias ? bias[n + e] : 0.f);
        if (relu) x = fmaxf(x, 0.f);
        if (mask && !(Elem<T>::to_f32(mask[i4 + e]) > 0.f)) x = 0.f;
        v[e] = x;
    }
    if (out_f) *(f32x4*)(out_f + i4) = f32x4{v[0], v[1], v[2], v[3]};
    else *(PackN<T, 4>*)(out_t + i4) = pack4<T>(v);
}

}  // namespace mi

extern "C" {

int mi_vae_reparam_kl_fwd(void* stream, int dtype, const float* heads, int nsplit, const float* bias_mean, const float* bias_lv,
                          const float* eps, int sample, int B, int Z, float* mean, float* logvar, void* z, float* kl_row) {
    return mi_vae_reparam_kl_fwd_rng(stream, dtype, heads, nsplit, bias_mean, bias_lv, eps, sample, B, Z, mean, logvar, z, kl_row, nullptr, nullptr);
}

static bool reparam_wide() {                             // MI355_REPARAM_WIDE=1 (A/B knob, default off): the reparameterisation kernels request up to 32 slabs per element at once
    static int on = -1;
    if (on < 0) { const char* e = getenv("MI355_REPARAM_WIDE"); on = (e && e[0] == '1') ? 1 : 0; }
    return on != 0;
}

// same; eps == NULL with sample != 0: the kernel draws the noise from the Philox stream in rng_state (4 x uint64 on the device: seed, next
// element offset, 2 words of kernel bookkeeping -- zero them once) and stores what it drew in eps_out [B,Z] for the backward pass
int mi_vae_reparam_kl_fwd_rng(void* stream, int dtype, const float* heads, int nsplit, const float* bias_mean, const float* bias_lv,
                              const float* eps, int sample, int B, int Z, float* mean, float* logvar, void* z, float* kl_row,
                              unsigned long long* rng_state, float* eps_out) {
    if (sample && !eps && !(rng_state && eps_out)) return mi_fail(MI_ERR_ARG, "mi_vae_reparam_kl_fwd: sampling needs eps or a generator state + eps_out");
    dim3 g((B + 3) / 4), b(256);
    if (reparam_wide() && nsplit > 8) {                   // MI355_REPARAM_WIDE=1: all (up to 32) slabs of an element requested before the first add -- same order, same sums
        BY_DTYPE(dtype, hipLaunchKernelGGL((reparam_kl_fwd_kernel<TT, 32>), g, b, 0, (hipStream_t)stream, heads, nsplit, bias_mean, bias_lv, eps, sample, B, Z, mean, logvar, (TT*)z, kl_row, rng_state, eps_out));
        return mi_check_launch("reparam_kl_fwd");
    }
    BY_DTYPE(dtype, hipLaunchKernelGGL(reparam_kl_fwd_kernel<TT>, g, b, 0, (hipStream_t)stream, heads, nsplit, bias_mean, bias_lv, eps, sample, B, Z, mean, logvar, (TT*)z, kl_row, rng_state, eps_out));
    return mi_check_launch("reparam_kl_fwd");
}

int mi_normal_philox(void* stream, unsigned long long seed, unsigned long long offset, float* out, long long n) {
    if (n < 0 || (n > 0 && !out)) return mi_fail(MI_ERR_ARG, "mi_normal_philox: bad buffer");
    if (n == 0) return MI_OK;
    hipLaunchKernelGGL(normal_philox_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, seed, offset, out, n);
    return mi_check_launch("normal_philox");
}

int mi_vae_reparam_kl_bwd(void* stream, int dtype, const float* dz_slabs, int nsplit, const float* mean, const float* logvar,
                          const float* eps, const float* kl_row, float beta, float kl_floor, float inv_batch, int B, int Z, void* dheads) {
    dim3 g((B + 3) / 4), b(256);
    if (reparam_wide() && nsplit > 8) {
        BY_DTYPE(dtype, hipLaunchKernelGGL((reparam_kl_bwd_kernel<TT, 32>), g, b, 0, (hipStream_t)stream, dz_slabs, nsplit, mean, logvar, eps, kl_row, beta, kl_floor, inv_batch, B, Z, (TT*)dheads));
        return mi_check_launch("reparam_kl_bwd");
    }
    BY_DTYPE(dtype, hipLaunchKernelGGL(reparam_kl_bwd_kernel<TT>, g, b, 0, (hipStream_t)stream, dz_slabs, nsplit, mean, logvar, eps, kl_row, beta, kl_floor, inv_batch, B, Z, (TT*)dheads));
    return mi_check_launch("reparam_kl_bwd");
}

// SURVEY 8b's name for the pair: forward half when heads != NULL, backward half when dz_slabs != NULL (both: forward, then backward on what it wrote)
int mi_vae_reparam_kl_fwd_bwd(void* stream, int dtype, const float* heads, int nsplit, const float* bias_mean, const float* bias_lv, const float* eps, int sample, int B, int Z,
                              float* mean, float* logvar, void* z, float* kl_row, const float* dz_slabs, int dz_nsplit, float beta, float kl_floor, float inv_batch, void* dheads) {
    if (!heads && !dz_slabs) return mi_fail(MI_ERR_ARG, "mi_vae_reparam_kl_fwd_bwd: neither half requested (heads and dz_slabs are both NULL)");
    if (!mean || !logvar || !kl_row || B < 1 || Z < 1) return mi_fail(MI_ERR_ARG, "mi_vae_reparam_kl_fwd_bwd: missing mean / logvar / kl_row buffers or empty shape");
    if (heads) {
        if (!z || (sample && !eps)) return mi_fail(MI_ERR_ARG, "mi_vae_reparam_kl_fwd_bwd: the forward half needs z (and eps when sampling)");
        const int rc = mi_vae_reparam_kl_fwd(stream, dtype, heads, nsplit, bias_mean, bias_lv, eps, sample, B, Z, mean, logvar, z, kl_row);
        if (rc != MI_OK) return rc;
    }
    if (dz_slabs) {
        if (!dheads || !eps) return mi_fail(MI_ERR_ARG, "mi_vae_reparam_kl_fwd_bwd: the backward half needs dheads and the noise of the forward pass");
        return mi_vae_reparam_kl_bwd(stream, dtype, dz_slabs, dz_nsplit, mean, logvar, eps, kl_row, beta, kl_floor, inv_batch, B, Z, dheads);
    }
    return MI_OK;
}

int mi_recon_loss_chunks(int P) { return (P + BCE_CHUNK - 1) / BCE_CHUNK; }

// logits [B,P] (T) vs labels (fp32 frames, optionally gathered through frame_idx) -> partial[B][chunks], dlogits [B,P] (T, may be null)
int mi_bce_logits_fwd_bwd(void* stream, int dtype, const void* logits, const float* labels, const int* frame_idx, long long label_stride,
                          int B, int P, int loss_kind, float inv_batch, void* dlogits, float* partial) {
    return mi_bce_logits_fwd_bwd_bias(stream, dtype, logits, labels, frame_idx, label_stride, B, P, loss_kind, inv_batch, dlogits, partial, 1, nullptr);
}

// same + BiasAddGrad of the producing layer fused: dbias[c] += sum over frames and pixels of the stored dlogits, c = element % channels
int mi_bce_logits_fwd_bwd_bias(void* stream, int dtype, const void* logits, const float* labels, const int* frame_idx, long long label_stride,
                               int B, int P, int loss_kind, float inv_batch, void* dlogits, float* partial, int channels, float* dbias) {
    const int nch = mi_recon_loss_chunks(P);
    dim3 g(nch, B), b(256);
    if (loss_kind < 0 || loss_kind > 2) return mi_fail(MI_ERR_ARG, "mi_bce_logits_fwd_bwd: loss_kind must be 0 (bce), 1 (bce_v2) or 2 (mse)");
    if (dbias && (channels < 1 || channels > 3 || !dlogits)) return mi_fail(MI_ERR_ARG, "mi_bce_logits_fwd_bwd_bias: fused bias gradient needs 1..3 channels and dlogits");
    BY_DTYPE(dtype, hipLaunchKernelGGL((recon_loss_kernel<TT, float>), g, b, 0, (hipStream_t)stream, (const TT*)logits, labels, frame_idx, label_stride, P, loss_kind, inv_batch, (TT*)dlogits, partial, nch, channels, dbias));
    return mi_check_launch("recon_loss");
}

// the same with the labels as raw uint8 camera bytes (label_stride in bytes = values): float32(k) / float32(255) exactly, in registers
int mi_bce_logits_fwd_bwd_u8(void* stream, int dtype, const void* logits, const unsigned char* labels, const int* frame_idx, long long label_stride,
                             int B, int P, int loss_kind, float inv_batch, void* dlogits, float* partial) {
    if (!logits || !labels || !partial || B < 1 || P < 1) return mi_fail(MI_ERR_ARG, "mi_bce_logits_fwd_bwd_u8: bad arguments");
    if (loss_kind < 0 || loss_kind > 2) return mi_fail(MI_ERR_ARG, "mi_bce_logits_fwd_bwd_u8: loss_kind must be 0 (bce), 1 (bce_v2) or 2 (mse)");
    const int nch = mi_recon_loss_chunks(P);
    dim3 g(nch, B), b(256);
    BY_DTYPE(dtype, hipLaunchKernelGGL((recon_loss_kernel<TT, unsigned char>), g, b, 0, (hipStream_t)stream, (const TT*)logits, labels, frame_idx, label_stride, P, loss_kind, inv_batch, (TT*)dlogits, partial, nch, 1, (float*)nullptr));
    return mi_check_launch("recon_loss_u8");
}

int mi_vae_finalize_losses(void* stream, const float* partial, int nchunks, const float* kl_row, float kl_floor, int B, float inv_batch,
                           float* out2, float* metrics3, float metric_weight) {
    return mi_vae_finalize_losses_flat(stream, partial, B * nchunks, kl_row, kl_floor, B, inv_batch, out2, metrics3, metric_weight, nullptr, 0, 0, nullptr);
}

// same over a flat list of n_partial loss partial sums (any producer); optionally adds the per-block channel sums bias_partial[n][4]
// of a fused loss pass into dbias[0..channels)
int mi_vae_finalize_losses_flat(void* stream, const float* partial, int n_partial, const float* kl_row, float kl_floor, int B, float inv_batch,
                                float* out2, float* metrics3, float metric_weight, const float* bias_partial, int n_bias_partial, int channels, float* dbias) {
    hipLaunchKernelGGL(finalize_losses_kernel, dim3(1), dim3(FIN_NT), 0, (hipStream_t)stream, partial, n_partial, kl_row, kl_floor, B, inv_batch, out2, metrics3,
                       metric_weight, bias_partial, n_bias_partial, channels, dbias);
    return mi_check_launch("finalize_losses");
}

int mi_adam_tf_flat(void* stream, float* param, float* m, float* v, float* grad, long long n, float alpha, float beta1, float beta2,
                    float epsilon, void* bf16_shadow, int clear_grad) {
    return mi_adam_tf_flat_dev(stream, param, m, v, grad, n, alpha, nullptr, beta1, beta2, epsilon, bf16_shadow, clear_grad);
}

// same; alpha_dev != NULL: the step size is read from device memory at run time (one float) instead of the argument
int mi_adam_tf_flat_dev(void* stream, float* param, float* m, float* v, float* grad, long long n, float alpha, const float* alpha_dev, float beta1, float beta2,
                        float epsilon, void* bf16_shadow, int clear_grad) {
    return mi_adam_tf_flat_shadow(stream, param, m, v, grad, n, alpha, alpha_dev, beta1, beta2, epsilon, bf16_shadow, MI_BF16, clear_grad);
}

// same; shadow_dtype names the storage type of the shadow weight copy the MFMA kernels read: MI_BF16 (2 bytes per weight) or MI_BF16X3 (split, 4 bytes)
int mi_adam_tf_flat_shadow(void* stream, float* param, float* m, float* v, float* grad, long long n, float alpha, const float* alpha_dev, float beta1, float beta2,
                           float epsilon, void* shadow, int shadow_dtype, int clear_grad) {
    if ((((uintptr_t)param) | ((uintptr_t)m) | ((uintptr_t)v) | ((uintptr_t)grad)) & 15) return mi_fail(MI_ERR_ARG, "mi_adam_tf_flat: buffers must be 16-byte aligned");
    if (shadow && shadow_dtype != MI_BF16 && shadow_dtype != MI_BF16X3) return mi_fail(MI_ERR_ARG, "mi_adam_tf_flat: the shadow copy is bf16 or split storage");
    if (shadow && (((uintptr_t)shadow) & (shadow_dtype == MI_BF16X3 ? 15 : 7))) return mi_fail(MI_ERR_ARG, "mi_adam_tf_flat: shadow copy must be 8-byte (bf16) / 16-byte (split) aligned");
    hipLaunchKernelGGL(adam_tf_kernel, dim3(grid_for(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, param, m, v, grad, n, alpha, alpha_dev,
                       1.0f - beta1, 1.0f - beta2, epsilon, shadow, shadow_dtype == MI_BF16X3 ? 1 : 0, clear_grad);
    return mi_check_launch("adam_tf");
}

int mi_splitk_finish(void* stream, int dtype, const float* slabs, int nsplit, int M, int N, const float* bias, int relu, const void* mask, void* out, int out_f32) {
    if (!slabs || !out || nsplit < 1 || M < 1 || N < 4 || N % 4 != 0) return mi_fail(MI_ERR_ARG, "mi_splitk_finish: bad arguments (N must be a multiple of 4)");
    const long long mn = (long long)M * N;
    const dim3 g((unsigned)((mn / 4 + 255) / 256));
    BY_DTYPE(dtype, hipLaunchKernelGGL(splitk_finish_kernel<TT>, g, dim3(256), 0, (hipStream_t)stream, slabs, nsplit, mn, N, bias, relu, (const TT*)mask, out_f32 ? nullptr : (TT*)out, out_f32 ? (float*)out : nullptr));
    return mi_check_launch("splitk_finish");
}

int mi_cast_f32_to_bf16(void* stream, const float* src, void* dst, long long n) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    return mi_check_launch("cast_f32_bf16");
}

// fp32 <-> split storage (MI_BF16X3): x -> (bf16(x) << 16 | bf16(x - bf16(x))) and back (hi + lo)
int mi_cast_f32_to_split(void* stream, const float* src, void* dst, long long n) {
    hipLaunchKernelGGL(cast_f32_split_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, src, (split_t*)dst, n);
    return mi_check_launch("cast_f32_split");
}
int mi_cast_split_to_f32(void* stream, const void* src, float* dst, long long n) {
    hipLaunchKernelGGL(cast_split_f32_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, (const split_t*)src, dst, n);
    return mi_check_launch("cast_split_f32");
}

int mi_u8_to_unit_f32(void* stream, const unsigned char* src, float* dst, long long n) {
    if (n < 0) return mi_fail(MI_ERR_ARG, "mi_u8_to_unit_f32: negative length");
    if (n == 0) return MI_OK;
    hipLaunchKernelGGL(u8_to_unit_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return mi_check_launch("u8_to_unit");
}

// TF ApplyAdam over the flat buffer [0, n) that also refreshes both weight copies of the `count` listed [K, N] kernels (offsets in floats, N % 4 == 0, offsets % 4 == 0):
// shadow (storage-type copy, same layout; NULL for fp32 engines) and wt (K-contiguous copy wt[off + n * K + k]; may be NULL).  dtype MI_F32 / MI_BF16 = element type of
// both copies.  Everything between / behind the listed kernels (the biases) gets the plain update (+ shadow).  The kernels must be listed in ascending offset order
// and must not overlap.  skip (may be NULL): per kernel, bit 0 = leave its shadow copy alone, bit 1 = leave its K-contiguous copy alone (copies nobody reads:
// the first layer of an encoder has no input gradient).  Same arithmetic as mi_adam_tf_flat: bit-identical p / m / v.
int mi_adam_tf_layouts(void* stream, int dtype, float* param, float* m, float* v, float* grad, long long n, const long long* offsets, const int* K, const int* N, const int* skip, int count,
                       float alpha, const float* alpha_dev, float beta1, float beta2, float epsilon, void* shadow, void* wt, int clear_grad) {
    return mi_adam_tf_layouts_frag(stream, dtype, param, m, v, grad, n, offsets, K, N, skip, count, alpha, alpha_dev, beta1, beta2, epsilon, shadow, wt, clear_grad, nullptr, nullptr);
}

// same + fragment-ordered bf16 copies of some of the kernels (frag_ptrs[2 * i + q], frag_forms[2 * i + q] for kernel i, q = 0, 1; NULL / -1 = none; bf16 storage only): what
// mi_ares_pack_weights writes, emitted by the optimiser launch itself (round 5)
int mi_adam_tf_layouts_frag(void* stream, int dtype, float* param, float* m, float* v, float* grad, long long n, const long long* offsets, const int* K, const int* N, const int* skip, int count,
                            float alpha, const float* alpha_dev, float beta1, float beta2, float epsilon, void* shadow, void* wt, int clear_grad, void* const* frag_ptrs, const int* frag_forms) {
    if (dtype != MI_F32 && dtype != MI_BF16) return mi_fail(MI_ERR_ARG, "mi_adam_tf_layouts: dtype must be MI_F32 or MI_BF16");
    if (count < 0 || count > AL_MAX) return mi_fail(MI_ERR_ARG, "mi_adam_tf_layouts: 0 <= count <= 16");
    if ((((uintptr_t)param) | ((uintptr_t)m) | ((uintptr_t)v) | ((uintptr_t)grad)) & 15) return mi_fail(MI_ERR_ARG, "mi_adam_tf_layouts: buffers must be 16-byte aligned");
    if ((shadow && (((uintptr_t)shadow) & 15)) || (wt && (((uintptr_t)wt) & 15))) return mi_fail(MI_ERR_ARG, "mi_adam_tf_layouts: weight copies must be 16-byte aligned");
    AdamLayoutJobs jb = {};
    long long pos = 0;
    int tiles = 0, fblk = 0;
    auto add_flat = [&](long long lo, long long hi) -> bool {
        if (hi <= lo) return true;
        if (jb.fcount == AL_MAX) return false;
        jb.foff[jb.fcount] = lo; jb.fn[jb.fcount] = hi - lo; jb.fblk0[jb.fcount] = fblk;
        fblk += (int)((hi - lo + 1023) / 1024); ++jb.fcount;
        return true;
    };
    for (int i = 0; i < count; ++i) {
        const long long sz = (long long)K[i] * N[i];
        if (K[i] < 1 || N[i] < 4 || N[i] % 4 != 0 || offsets[i] % 4 != 0) return mi_fail(MI_ERR_SHAPE, "mi_adam_tf_layouts: kernels need N % 4 == 0 and an offset that is a multiple of 4");
        if (offsets[i] < pos || offsets[i] + sz > n) return mi_fail(MI_ERR_ARG, "mi_adam_tf_layouts: kernels must be listed in ascending, non-overlapping order inside [0, n)");
        if (!add_flat(pos, offsets[i])) return mi_fail(MI_ERR_ARG, "mi_adam_tf_layouts: too many ranges between the kernels");
        jb.off[i] = offsets[i]; jb.K[i] = K[i]; jb.N[i] = N[i]; jb.tile0[i] = tiles;
        if (skip && (skip[i] & 1)) jb.no_shadow |= 1u << i;
        if (skip && (skip[i] & 2)) jb.no_wt |= 1u << i;
        for (int q = 0; q < 2; ++q) {
            jb.frag[i][q] = nullptr; jb.fform[i][q] = -1;
            if (!frag_ptrs || !frag_forms || !frag_ptrs[2 * i + q]) continue;
            const int form = frag_forms[2 * i + q];
            const bool shape_ok = (form == 0 || form == 1) ? (K[i] == 2048 && N[i] == 256) : (form == 2 || form == 3) ? (K[i] == 1024 && N[i] == 128) : (form == 4 || form == 6) ? (K[i] == 800 && N[i] == 64) : (form == 5 && K[i] == 512 && N[i] == 64);
            if (dtype != MI_BF16 || !shape_ok || (((uintptr_t)frag_ptrs[2 * i + q]) & 15)) return mi_fail(MI_ERR_ARG, "mi_adam_tf_layouts_frag: fragment copies are bf16, 16-byte aligned, form 0 | 1 of a [2048, 256] kernel, 2 | 3 of a [1024, 128], 4 of a [800, 64], 5 of a [512, 64] kernel");
            jb.frag[i][q] = (bf16_t*)frag_ptrs[2 * i + q]; jb.fform[i][q] = (signed char)form;
        }
        tiles += ((K[i] + 63) / 64) * ((N[i] + 63) / 64);
        pos = offsets[i] + sz;
    }
    if (!add_flat(pos, n)) return mi_fail(MI_ERR_ARG, "mi_adam_tf_layouts: too many ranges between the kernels");
    jb.count = count;
    for (int i = count; i <= AL_MAX; ++i) jb.tile0[i] = tiles;
    for (int i = jb.fcount; i <= AL_MAX; ++i) jb.fblk0[i] = fblk;
    if (tiles + fblk == 0) return MI_OK;
    if (dtype == MI_BF16) hipLaunchKernelGGL(adam_tf_layouts_kernel<bf16_t>, dim3(tiles + fblk), dim3(256), 0, (hipStream_t)stream, param, m, v, grad, jb, alpha, alpha_dev,
                                             1.0f - beta1, 1.0f - beta2, epsilon, (bf16_t*)shadow, (bf16_t*)wt, clear_grad);
    else hipLaunchKernelGGL(adam_tf_layouts_kernel<float>, dim3(tiles + fblk), dim3(256), 0, (hipStream_t)stream, param, m, v, grad, jb, alpha, alpha_dev,
                            1.0f - beta1, 1.0f - beta2, epsilon, (float*)shadow, (float*)wt, clear_grad);
    return mi_check_launch("adam_tf_layouts");
}

// out[b, :] = storage_type(src[idx[b], :])  (idx NULL: rows 0 .. B-1); dtype MI_F32 (a gathered copy) or MI_BF16
int mi_gather_rows_cast(void* stream, int dtype, const float* src, const int* idx, int B, long long row_len, void* out) {
    if (dtype != MI_F32 && dtype != MI_BF16) return mi_fail(MI_ERR_ARG, "mi_gather_rows_cast: dtype must be MI_F32 or MI_BF16");
    if (!src || !out || B < 0 || row_len < 1) return mi_fail(MI_ERR_ARG, "mi_gather_rows_cast: bad arguments");
    if (B == 0) return MI_OK;
    const long long chunks = (row_len + 2047) / 2048;
    if (chunks * B > 0x7fffffffll) return mi_fail(MI_ERR_SHAPE, "mi_gather_rows_cast: B x row_len beyond one launch");
    const dim3 g((unsigned)(chunks * B));
    if (dtype == MI_BF16) hipLaunchKernelGGL((gather_rows_cast_kernel<bf16_t, float>), g, dim3(256), 0, (hipStream_t)stream, src, idx, row_len, (bf16_t*)out, (unsigned)chunks);
    else hipLaunchKernelGGL((gather_rows_cast_kernel<float, float>), g, dim3(256), 0, (hipStream_t)stream, src, idx, row_len, (float*)out, (unsigned)chunks);
    return mi_check_launch("gather_rows_cast");
}

// the same from a uint8 table of raw camera bytes: out[b, :] = storage_type(float32(src[idx[b], :]) / float32(255)) -- the reference's host preprocessing (vae/train_vae.py:15-18)
// done where the minibatch is staged: a quarter of the table in HBM, a quarter of the gather's read traffic
int mi_gather_rows_cast_u8(void* stream, int dtype, const unsigned char* src, const int* idx, int B, long long row_len, void* out) {
    if (dtype != MI_F32 && dtype != MI_BF16) return mi_fail(MI_ERR_ARG, "mi_gather_rows_cast_u8: dtype must be MI_F32 or MI_BF16");
    if (!src || !out || B < 0 || row_len < 1) return mi_fail(MI_ERR_ARG, "mi_gather_rows_cast_u8: bad arguments");
    if (B == 0) return MI_OK;
    const long long chunks = (row_len + 2047) / 2048;
    if (chunks * B > 0x7fffffffll) return mi_fail(MI_ERR_SHAPE, "mi_gather_rows_cast_u8: B x row_len beyond one launch");
    const dim3 g((unsigned)(chunks * B));
    if (dtype == MI_BF16) hipLaunchKernelGGL((gather_rows_cast_kernel<bf16_t, unsigned char>), g, dim3(256), 0, (hipStream_t)stream, src, idx, row_len, (bf16_t*)out, (unsigned)chunks);
    else hipLaunchKernelGGL((gather_rows_cast_kernel<float, unsigned char>), g, dim3(256), 0, (hipStream_t)stream, src, idx, row_len, (float*)out, (unsigned)chunks);
    return mi_check_launch("gather_rows_cast_u8");
}

int mi_transpose_weights(void* stream, int dtype, const float* src, void* dst, const long long* offsets, const int* K, const int* N, int count) {
    if (count < 1 || count > 16) return mi_fail(MI_ERR_ARG, "mi_transpose_weights: 1 <= count <= 16");
    TransposeBatch tb;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        if (K[i] < 1 || N[i] < 1) return mi_fail(MI_ERR_ARG, "mi_transpose_weights: empty tensor");
        tb.off[i] = offsets[i]; tb.K[i] = K[i]; tb.N[i] = N[i]; tb.tile0[i] = tiles;
        tiles += ((K[i] + 31) / 32) * ((N[i] + 31) / 32);
    }
    tb.tile0[count] = tiles; tb.count = count;
    BY_DTYPE(dtype, hipLaunchKernelGGL(transpose_weights_kernel<TT>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, src, (TT*)dst, tb));
    return mi_check_launch("transpose_weights");
}

// out[N] += column sums of x[M,N]   (BiasAddGrad)
int mi_colsum(void* stream, int dtype, const void* x, long long M, int N, float* out) {
    return mi_colsum_ws(stream, dtype, x, M, N, out, nullptr, 0);
}

// row blocks (and rows per block) of a column-sum launch: a function of the shape and storage type only
static int colsum_plan(int dtype, const void* x, long long M, int N, int* kind, long long* rows_out) {
    const int vec = dtype == MI_BF16 ? 8 : 4;
    long long rows;
    if (N <= 256 && N % vec == 0 && ((((uintptr_t)x) & 15) == 0)) {
        // <= 512 blocks (2,048 for tensors beyond 32 MB): block count (not bytes) sets the floor of the small ones
        const long long nblk = (long long)M * N * (dtype == MI_BF16 ? 2 : 4) > (32ll << 20) ? 2048 : 512;
        rows = (M + nblk - 1) / nblk;
        if (rows * N < 32768) rows = (32768 + N - 1) / N;
        *kind = 0;
    } else if (N <= 256) {
        rows = (M + 1023) / 1024;
        if (rows * N < 4096) rows = (4096 + N - 1) / N;
        *kind = 1;
    } else {
        rows = (M + 63) / 64;
        if (rows < 8) rows = 8;
        *kind = 2;
    }
    *rows_out = rows;
    return (int)((M + rows - 1) / rows);
}

// scratch of the deterministic form: one row of N (rounded up to 4) fp32 partial sums per row block of the launch
long long mi_colsum_scratch_bytes(int dtype, long long M, int N) {
    if (M <= 0 || N <= 0) return 0;
    int kind; long long rows;
    int gx = colsum_plan(dtype, nullptr, M, N, &kind, &rows);
    if (kind == 0) {                                      // a tensor of this shape that is not 16-byte aligned takes the scalar kernel and its row blocks
        long long r1 = (M + 1023) / 1024;
        if (r1 * N < 4096) r1 = (4096 + N - 1) / N;
        const int g1 = (int)((M + r1 - 1) / r1);
        if (g1 > gx) gx = g1;
    }
    return (long long)gx * ((N + 3) / 4 * 4) * 4;
}

// out[n] += sum_m x[m, n].  With scratch (>= mi_colsum_scratch_bytes) every row block stores its column sums and one ordered pass adds them to out:
// two runs are bitwise equal (round 4).  Without it the row blocks meet in fp32 atomics on out.
int mi_colsum_ws(void* stream, int dtype, const void* x, long long M, int N, float* out, void* scratch, long long scratch_bytes) {
    if (M <= 0 || N <= 0) return MI_OK;
    const int pstride = (N + 3) / 4 * 4;
    int kind; long long rows;
    const int gx = colsum_plan(dtype, x, M, N, &kind, &rows);
    float* part = (gx > 1 && scratch && (((uintptr_t)scratch) & 15) == 0 && scratch_bytes >= (long long)gx * pstride * 4) ? (float*)scratch : nullptr;
    if (kind == 0) {
        BY_DTYPE(dtype, hipLaunchKernelGGL(colsum_vec_kernel<TT>, dim3(gx), dim3(256), 0, (hipStream_t)stream, (const TT*)x, M, N, (int)rows, out, part, pstride));
    } else if (kind == 1) {
        BY_DTYPE(dtype, hipLaunchKernelGGL(colsum_small_kernel<TT>, dim3(gx), dim3(256), 0, (hipStream_t)stream, (const TT*)x, M, N, (int)rows, out, part, pstride));
    } else {
        const int gy = (N + 255) / 256;
        BY_DTYPE(dtype, hipLaunchKernelGGL(colsum_wide_kernel<TT>, dim3(gx, gy), dim3(256), 0, (hipStream_t)stream, (const TT*)x, M, N, (int)rows, out, part, pstride));
    }
    int rc = mi_check_launch("colsum");
    if (rc == MI_OK && part) rc = mi_reduce_slabs((hipStream_t)stream, part, pstride, gx, N, out);
    return rc;
}

int mi_sigmoid(void* stream, int dtype, const void* x, float* out, long long n) {
    BY_DTYPE(dtype, hipLaunchKernelGGL(sigmoid_kernel<TT>, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, (const TT*)x, out, n));
    return mi_check_launch("sigmoid");
}

int mi_range_check(void* stream, const float* x, long long n, float lo, float hi, int* flag) {
    hipLaunchKernelGGL(range_check_kernel, dim3(grid_for(n, 2048)), dim3(256), 0, (hipStream_t)stream, x, n, lo, hi, flag);
    return mi_check_launch("range_check");
}

}  // extern "C"
