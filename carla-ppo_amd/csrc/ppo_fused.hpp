// ppo_fused.hpp — parameter block and host entry points of the fused PPO kernels (ppo_fused.hip), shared with ppo_engine.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace mi {

struct PpoFusedParams {
    // flat parameter buffers (engine layout, offsets in floats) and optimiser state
    float *theta, *adam_m, *adam_v, *grads; const float* theta_old;
    long long off[13];
    int kin, din, H1, H2, A, M;
    long long n_params;                                   // floats in each flat buffer
    // activations (workspace): [net][M][H] with net 0 = policy, 1 = value, 2 = old policy
    const float *states, *actions, *returns, *adv, *low, *high;
    float *h1, *h2, *dh1, *dh2;                           // h1/h2: 3 nets; dh1/dh2: 2 nets
    float *du, *dv, *partial, *losses, *mean_out;
    // minibatch gather fused into the step (mi_ppo_train_step_idx): sample m of the minibatch is row row_idx[m] of the horizon-batch tables
    // states / actions / returns / adv / logp_old (n_rows rows each); layer 1 also leaves the gathered states in s_gath [M, din] for the filter gradient
    const int* row_idx; int n_rows; float* s_gath;
    const float* logp_old;                                // cached log pi_old(a|s) per sample (nullptr: recomputed from net 2)
    float* logp_out;                                      // when set: the loss kernel also stores log pi(a|s) (used to fill the cache)
    int n_nets;                                           // 3, or 2 with the cache
    float clip_eps, value_scale, entropy_scale, inv_m, grad_scale;
    float alpha, omb1, omb2, epsilon;                     // Adam: alpha = lr * sqrt(1 - b2^t) / (1 - b1^t); omb = 1 - beta
    int n_loss_blocks;
    int m_chunk;                                          // weight-gradient launch: 0 = one wave sums the whole minibatch of its tile (M <= 256); > 0: rows per
                                                          // blockIdx.y (multiple of 32): chunk c STORES its partial gradient into gslab + c * gslab_stride and one ordered
                                                          // pass adds the chunks (round 4: no atomics, two runs bitwise equal; no fused Adam)
    float* gslab; long long gslab_stride;                 // per-chunk gradient slabs of the large-minibatch form (engine workspace; nullptr when max_batch <= 256)
};

}  // namespace mi

int mi_ppo_fused_partial_floats(int M);
int mi_ppo_fused_trunks(hipStream_t st, const mi::PpoFusedParams& q);
bool mi_ppo_fused_shape_in_range(int A, int H2, int kin);
int mi_ppo_fused_step(hipStream_t st, mi::PpoFusedParams& q, int fuse_adam);
int mi_ppo_fused_predict(hipStream_t st, mi::PpoFusedParams& q, const float* noise, int greedy, float* action, float* value);
int mi_ppo_fused_logp_old(hipStream_t st, mi::PpoFusedParams& q, float* out);
// internal accessors of the two engines for the rollout step (rollout path only)
int mi_ppo_internal_fill(void* ppo_handle, mi::PpoFusedParams* q, const float* states, int M);
struct MiZeroList { float* p[3]; long long n[3]; };     // raw-sum buffers conv1's launch clears for the split-K layers behind it
int mi_rollout_conv(hipStream_t st, const float* x, const float* x_bias, int IH, int IW, int C, const float* w, int ldw, int N, int KH, int KW, float* out_raw, int flat_k);
int mi_rollout_conv1(hipStream_t st, const unsigned char* frame, const float* w, const float* bias, float* out, int IH, int IW, int Cs, int KH, int KW, int N, const MiZeroList* zero);
int mi_rollout_policy(hipStream_t st, const mi::PpoFusedParams& q, const float* mean_raw, const float* mean_bias, int z_dim, const float* measurements, const float* noise, int greedy, float* out);
