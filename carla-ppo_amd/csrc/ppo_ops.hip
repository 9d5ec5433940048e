// ppo_ops.hip — wavefront-fused PPO kernels (gfx950, wave64):
//   clipped-surrogate / value / entropy losses with analytic gradients wrt the network heads,
//   Gaussian policy head (tanh squash, sample, clip), GAE reverse scan and advantage normalisation in fp64.
#include "common.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"

using namespace mi;

namespace {

constexpr float HALF_LOG_2PI = 0.918938533204672741780329736406f;
constexpr int MAX_ACT = 8;

// ---------------------------------------------------------------------------------------------------
// PPO loss + head gradients (reference ppo.py:47,58-66,112-132).  One thread per sample.
//   u, u_old : [M,A] pre-tanh outputs of action_mean (policy / policy_old)     vraw : [M] value head output
//   mean = low + (tanh(u)+1)/2*(high-low) ; logp = sum_a -.5*((a-mean)/sigma)^2 - (.5log2pi + log sigma), sigma = exp(logstd)
//   ratio = exp(logp - logp_old) ; L_clip = mean(min(r*A, clip(r,1-e,1+e)*A)) ; L_v = vs*mean((V-R)^2) ; L_ent = es*sum_a(1.4189+log sigma)
//   loss = -L_clip + L_v - L_ent.   Outputs du [M,A], dv [M] (d loss / d head pre-activations, already / M),
//   per-block partials: [policy_sum, value_sq_sum, ratio_sum, dlogstd_0..A-1] -> finalised in ppo_loss_finalize_kernel.
// ---------------------------------------------------------------------------------------------------
constexpr int PPO_NPART = 3 + MAX_ACT;

__global__ __launch_bounds__(256) void ppo_loss_kernel(const float* __restrict__ u, const float* __restrict__ u_old,
                                                       const float* __restrict__ logstd, const float* __restrict__ logstd_old,
                                                       const float* __restrict__ vraw, const float* __restrict__ actions,
                                                       const float* __restrict__ returns, const float* __restrict__ adv,
                                                       const float* __restrict__ low, const float* __restrict__ high,
                                                       int M, int A, float clip_eps, float value_scale, float inv_m,
                                                       float* __restrict__ du, float* __restrict__ dv, float* __restrict__ partial) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float vals[PPO_NPART];
#pragma unroll
    for (int k = 0; k < PPO_NPART; ++k) vals[k] = 0.f;
    if (i < M) {
        float logp = 0.f, logp_old = 0.f;
        float dlogp_dmean_scaled[MAX_ACT], zsq[MAX_ACT];
        for (int a = 0; a < A; ++a) {
            const float lo = low[a], hi = high[a];
            const float act = actions[(long long)i * A + a];
            const float t = tanhf(u[(long long)i * A + a]);
            const float mean = lo + ((t + 1.0f) * 0.5f) * (hi - lo);
            const float sigma = expf(logstd[a]);
            const float z = (act - mean) / sigma;
            logp += -0.5f * z * z - (HALF_LOG_2PI + logf(sigma));
            // d logp / d u = (act-mean)/sigma^2 * (hi-lo)/2 * (1 - t^2)
            dlogp_dmean_scaled[a] = (z / sigma) * (0.5f * (hi - lo)) * (1.0f - t * t);
            zsq[a] = z * z;
            const float to = tanhf(u_old[(long long)i * A + a]);
            const float mo = lo + ((to + 1.0f) * 0.5f) * (hi - lo);
            const float so = expf(logstd_old[a]);
            const float zo = (act - mo) / so;
            logp_old += -0.5f * zo * zo - (HALF_LOG_2PI + logf(so));
        }
        const float r = expf(logp - logp_old);
        const float ad = adv[i];
        const float rc = fminf(fmaxf(r, 1.0f - clip_eps), 1.0f + clip_eps);
        const float s1 = r * ad, s2 = rc * ad;
        vals[0] = fminf(s1, s2);
        // tf.minimum sends the gradient to the first argument when s1 <= s2 (ties included); the clipped branch has zero slope
        const float dr = (s1 <= s2) ? ad : 0.f;
        const float coef = -dr * r * inv_m;                       // d(-L_clip)/d logp
        for (int a = 0; a < A; ++a) {
            du[(long long)i * A + a] = coef * dlogp_dmean_scaled[a];
            vals[3 + a] = coef * (zsq[a] - 1.0f);                   // d(-L_clip)/d logstd_a   (d log sigma/d logstd = 1)
        }
        const float dvv = vraw[i] - returns[i];
        vals[1] = dvv * dvv;
        dv[i] = 2.0f * value_scale * dvv * inv_m;
        vals[2] = r;
    }
    __shared__ float red[4][PPO_NPART];
#pragma unroll
    for (int k = 0; k < PPO_NPART; ++k) {
        const float s = wave_sum(vals[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < PPO_NPART)
        partial[(long long)blockIdx.x * PPO_NPART + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// losses[0..4] = policy_loss, value_loss, entropy_loss, loss, mean ratio ; dlogstd[a] (+)= grad.  Fixed block order.
__global__ void ppo_loss_finalize_kernel(const float* __restrict__ partial, int nblocks, const float* __restrict__ logstd, int A,
                                         float inv_m, float value_scale, float entropy_scale, float grad_scale,
                                         float* __restrict__ losses, float* __restrict__ dlogstd) {
    if (threadIdx.x != 0) return;
    float s[PPO_NPART];
    for (int k = 0; k < PPO_NPART; ++k) s[k] = 0.f;
    for (int b = 0; b < nblocks; ++b)
        for (int k = 0; k < 3 + A; ++k) s[k] += partial[(long long)b * PPO_NPART + k];
    float ent = 0.f;
    for (int a = 0; a < A; ++a) ent += 0.5f + HALF_LOG_2PI + logf(expf(logstd[a]));
    const float pl = s[0] * inv_m, vl = s[1] * inv_m * value_scale, el = ent * entropy_scale;
    losses[0] = pl; losses[1] = vl; losses[2] = el; losses[3] = -pl + vl - el; losses[4] = s[2] * inv_m;
    // entropy term is state independent: under data parallelism grad_scale = local_M / global_M shares it across ranks
    for (int a = 0; a < A; ++a) dlogstd[a] += s[3 + a] - entropy_scale * grad_scale;
}

// Gaussian head for predict() (reference ppo.py:47,58-62): mean from u; action = clip(mean + exp(logstd)*noise, low, high) or mean.
__global__ void policy_head_kernel(const float* __restrict__ u, const float* __restrict__ logstd, const float* __restrict__ noise,
                                   const float* __restrict__ low, const float* __restrict__ high, int M, int A, int greedy,
                                   float* __restrict__ action, float* __restrict__ mean_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * A) return;
    const int a = idx % A;
    const float lo = low[a], hi = high[a];
    const float mean = lo + ((tanhf(u[idx]) + 1.0f) * 0.5f) * (hi - lo);
    if (mean_out) mean_out[idx] = mean;
    float act = mean;
    if (!greedy) act = fminf(fmaxf(mean + expf(logstd[a]) * noise[idx], lo), hi);
    action[idx] = act;
}

// ---------------------------------------------------------------------------------------------------
// GAE (reference utils.py:45-50) in fp64 with the exact rounding sequence of numpy + scipy.signal.lfilter:
//   delta_t = r_t + ((1-done_t)*gamma)*V_{t+1} - V_t ;  A_t = delta_t + (gamma*lam)*A_{t+1}   (no FMA contraction)
// One thread per trajectory row; rows are independent (config C5 shards them across GPUs with no exchange).
// ---------------------------------------------------------------------------------------------------
__global__ void gae_scan_f64_kernel(const double* __restrict__ rewards, const double* __restrict__ values, const double* __restrict__ terminals,
                                    int R, int T, double gamma, double gl, double* __restrict__ adv) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    const double* r = rewards + (long long)row * T;
    const double* v = values + (long long)row * (T + 1);
    const double* d = terminals + (long long)row * T;
    double* o = adv + (long long)row * T;
    double carry = 0.0;
    for (int t = T - 1; t >= 0; --t) {
        const double nonterm = __dsub_rn(1.0, d[t]);
        const double delta = __dsub_rn(__dadd_rn(r[t], __dmul_rn(__dmul_rn(nonterm, gamma), v[t + 1])), v[t]);
        const double y = __dadd_rn(carry, delta);
        carry = __dmul_rn(gl, y);
        o[t] = y;
    }
}

// returns = A + V ; A = (A - mean(A)) / (std(A) + 1e-8), population std, per row (reference train.py:176-177). One wave per row.
__global__ void adv_normalize_f64_kernel(double* __restrict__ adv, const double* __restrict__ values, int R, int T, double* __restrict__ returns) {
    const int row = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    double* a = adv + (long long)row * T;
    const double* v = values + (long long)row * (T + 1);
    double s = 0.0;
    for (int t = lane; t < T; t += WAVE) { s += a[t]; if (returns) returns[(long long)row * T + t] = a[t] + v[t]; }
    s = wave_sum_f64(s);
    const double mean = s / (double)T;
    double ss = 0.0;
    for (int t = lane; t < T; t += WAVE) { const double dd = a[t] - mean; ss += dd * dd; }
    ss = wave_sum_f64(ss);
    const double sd = sqrt(ss / (double)T);
    for (int t = lane; t < T; t += WAVE) a[t] = (a[t] - mean) / (sd + 1e-8);
}

}  // namespace

namespace {
__global__ __launch_bounds__(256) void relu_grad_kernel(const float* __restrict__ g, const float* __restrict__ h, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = h[i] > 0.f ? g[i] : 0.f;
}
}  // namespace

extern "C" {

int mi_ppo_loss_blocks(int M) { return (M + 255) / 256; }
int mi_ppo_loss_partial_floats(int M) { return mi_ppo_loss_blocks(M) * PPO_NPART; }

int mi_ppo_loss_fwd_bwd(void* stream, const float* u, const float* u_old, const float* logstd, const float* logstd_old, const float* vraw,
                        const float* actions, const float* returns, const float* advantage, const float* low, const float* high,
                        int M, int A, float clip_eps, float value_scale, float entropy_scale, float inv_m, float grad_scale,
                        float* du, float* dv, float* partial, float* losses5, float* dlogstd) {
    if (A < 1 || A > MAX_ACT) return mi_fail(MI_ERR_ARG, "mi_ppo_loss_fwd_bwd: 1 <= num_actions <= 8");
    if (M < 1) return mi_fail(MI_ERR_ARG, "mi_ppo_loss_fwd_bwd: empty minibatch");
    const int nb = mi_ppo_loss_blocks(M);
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, u, u_old, logstd, logstd_old, vraw, actions, returns,
                       advantage, low, high, M, A, clip_eps, value_scale, inv_m, du, dv, partial);
    hipLaunchKernelGGL(ppo_loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, nb, logstd, A, inv_m, value_scale,
                       entropy_scale, grad_scale, losses5, dlogstd);
    return mi_check_launch("ppo_loss");
}

int mi_policy_head(void* stream, const float* u, const float* logstd, const float* noise, const float* low, const float* high,
                   int M, int A, int greedy, float* action, float* mean_out) {
    if (!greedy && !noise) return mi_fail(MI_ERR_ARG, "mi_policy_head: sampling needs noise");
    hipLaunchKernelGGL(policy_head_kernel, dim3((M * A + 255) / 256), dim3(256), 0, (hipStream_t)stream, u, logstd, noise, low, high, M, A, greedy, action, mean_out);
    return mi_check_launch("policy_head");
}

// rewards [R,T], values [R,T+1] (last column = bootstrap), terminals [R,T] (0/1), all fp64 -> adv [R,T]
// build_mlp trunk of the policy / value network as OP-level calls (SURVEY 8b names; the engines use the fused forms of ppo_fused.hip): utils.py:25-28 with
// hidden sizes (H1, H2) and ReLU on both layers (ppo.py:42-44,51-53).  Exact fp32 (the same dense kernels the engine's unfused path runs).
//   fwd: h1 = relu(x W1 + b1) [M, H1], h2 = relu(h1 W2 + b2) [M, H2]                       W1 [din, H1], W2 [H1, H2] (tf.layers.dense kernels)
//   bwd: given g2 = dL/dh2 (post-activation) [M, H2]: dW2 += h1^T (g2 . relu'(h2)), db2 += column sums, dW1 += x^T dh1, db1 += ..., with
//        dh1 = ((g2 . relu'(h2)) W2^T) . relu'(h1); scratch: M * (H1 + H2) floats.  The gradient wrt x is not produced (nothing upstream of the state trains).
int mi_mlp_policy_fwd(void* stream, const float* x, int M, int din, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, float* h1, float* h2) {
    if (!x || !W1 || !b1 || !W2 || !b2 || !h1 || !h2 || M < 1 || din < 1 || H1 < 1 || H2 < 1) return mi_fail(MI_ERR_ARG, "mi_mlp_policy_fwd: missing buffers or empty shape");
    if (din % 4 != 0 || H1 % 4 != 0 || H2 % 4 != 0) return mi_fail(MI_ERR_SHAPE, "mi_mlp_policy_fwd: din, H1, H2 must be multiples of 4 (16-byte rows; pad the state with zero columns as the engine does)");
    int rc = mi_gemm_bias_act(stream, MI_F32, x, M, din, W1, 0, H1, b1, 1, nullptr, h1, 1, 1);
    if (rc != MI_OK) return rc;
    return mi_gemm_bias_act(stream, MI_F32, h1, M, H1, W2, 0, H2, b2, 1, nullptr, h2, 1, 1);
}
int mi_mlp_policy_bwd(void* stream, const float* x, int M, int din, const float* W2, int H1, int H2, const float* h1, const float* h2, const float* g2,
                      float* dW1, float* db1, float* dW2, float* db2, float* scratch) {
    if (!x || !W2 || !h1 || !h2 || !g2 || !dW1 || !db1 || !dW2 || !db2 || !scratch || M < 1) return mi_fail(MI_ERR_ARG, "mi_mlp_policy_bwd: missing buffers or empty shape");
    float* g2m = scratch; float* dh1 = scratch + (long long)M * H2;
    if (din % 4 != 0 || H1 % 4 != 0 || H2 % 4 != 0) return mi_fail(MI_ERR_SHAPE, "mi_mlp_policy_bwd: din, H1, H2 must be multiples of 4 (16-byte rows; pad the state with zero columns as the engine does)");
    {
        const long long n = (long long)M * H2;
        hipLaunchKernelGGL(relu_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g2, h2, n, g2m);
        const int rc0 = mi_check_launch("relu_grad_kernel");
        if (rc0 != MI_OK) return rc0;
    }
    int rc = MI_OK;
    rc = mi_colsum(stream, MI_F32, g2m, M, H2, db2);
    if (rc == MI_OK) rc = mi_gemm_wgrad(stream, MI_F32, h1, g2m, M, H1, H2, dW2);
    if (rc == MI_OK) rc = mi_gemm_bias_act(stream, MI_F32, g2m, M, H2, W2, 1, H1, nullptr, 0, h1, dh1, 1, 1);      // x W^T with the ReluGrad mask of h1
    if (rc == MI_OK) rc = mi_colsum(stream, MI_F32, dh1, M, H1, db1);
    if (rc == MI_OK) rc = mi_gemm_wgrad(stream, MI_F32, x, dh1, M, din, H1, dW1);
    return rc;
}

int mi_gae_scan(void* stream, const double* rewards, const double* values, const double* terminals, int R, int T, double gamma, double lam, double* adv) {
    if (R < 1 || T < 1) return mi_fail(MI_ERR_ARG, "mi_gae_scan: empty input");
    hipLaunchKernelGGL(gae_scan_f64_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, rewards, values, terminals, R, T, gamma, gamma * lam, adv);
    return mi_check_launch("gae_scan");
}

int mi_adv_normalize(void* stream, double* adv, const double* values, int R, int T, double* returns) {
    if (R < 1 || T < 1) return mi_fail(MI_ERR_ARG, "mi_adv_normalize: empty input");
    hipLaunchKernelGGL(adv_normalize_f64_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, adv, values, R, T, returns);
    return mi_check_launch("adv_normalize");
}

}  // extern "C"
