/* oracle/gae_ref.c — plain-C restatement of the scalar recurrences on the PPO/VAE hot path.
 *
 * TEST INFRASTRUCTURE (see oracle/__init__.py): an independent second statement of
 *   - utils.py:45-50      compute_gae  (delta, then the order-1 IIR scipy.signal.lfilter([1],[1,-g*l]) runs)
 *   - train.py:176-177    returns = A + V ; A = (A - mean(A)) / (std(A) + 1e-8)   (population std, f64)
 *   - TF 1.13 ApplyAdam   (vae/models.py:141-142, ppo.py:143-144): m,v,var update in fp32, "epsilon hat" form
 * used by tests/ to cross-check oracle/ppo_oracle.py (scipy) and the HIP kernels.  Never linked into the product.
 *
 * Built by oracle/Makefile with -ffp-contract=off so no FMA contraction changes the rounding sequence:
 * scipy's lfilter (direct form II transposed) evaluates  y[n] = z + 1*x[n];  z = 0*x[n] - a1*y[n]
 * i.e. one f64 multiply and one f64 add per element — gae_f64 below is bit-identical to it.
 */
#include <math.h>
#include <stddef.h>

/* rewards[T], values[T+1] (last = bootstrap), terminals[T] (0/1) -> adv[T]   (all f64) */
void gae_f64(const double *rewards, const double *values, const double *terminals, int T,
             double gamma, double lam, double *adv)
{
    double gl = gamma * lam;            /* python evaluates -gamma*lam once: a1 = -(gamma*lam) */
    double carry = 0.0;
    for (int t = T - 1; t >= 0; --t) {
        double nonterm = 1.0 - terminals[t];
        double delta = rewards[t] + (nonterm * gamma) * values[t + 1] - values[t];
        double y = carry + delta;        /* y[n] = z[n-1] + b0*x[n], b0 = 1 */
        carry = gl * y;                  /* z[n]  = b1*x[n] - a1*y[n] = 0 + (gamma*lam)*y[n] */
        adv[t] = y;
    }
}

/* returns[t] = adv[t] + values[t]; adv normalised in place with population std (naive two-pass sums). */
void returns_and_normalize_f64(double *adv, const double *values, int T, double *returns)
{
    double s = 0.0;
    for (int t = 0; t < T; ++t) { returns[t] = adv[t] + values[t]; s += adv[t]; }
    double mean = s / T, ss = 0.0;
    for (int t = 0; t < T; ++t) { double d = adv[t] - mean; ss += d * d; }
    double sd = sqrt(ss / T);
    for (int t = 0; t < T; ++t) adv[t] = (adv[t] - mean) / (sd + 1e-8);
}

/* One TF ApplyAdam over a flat fp32 buffer.  alpha = lr*sqrt(1-b2p)/(1-b1p) is computed by the caller in fp32. */
void adam_tf_f32(float *var, float *m, float *v, const float *g, size_t n,
                 float alpha, float beta1, float beta2, float epsilon)
{
    float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    for (size_t i = 0; i < n; ++i) {
        m[i] += (g[i] - m[i]) * omb1;
        v[i] += (g[i] * g[i] - v[i]) * omb2;
        var[i] -= (m[i] * alpha) / (sqrtf(v[i]) + epsilon);
    }
}
