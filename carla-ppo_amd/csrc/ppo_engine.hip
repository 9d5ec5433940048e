// ppo_engine.hip — PPO policy/value network: predict and one minibatch-SGD step behind the C ABI.
// Mirrors PolicyGraph + PPO (reference ppo.py:16-66,112-147,218-251,275-276) with utils.build_mlp (utils.py:25-28):
//   pi: s -> dense 500 relu -> dense 300 relu -> action_mean (tanh, rescaled to [low,high]) ; free action_logstd
//   V : s -> dense 500 relu -> dense 300 relu -> value
// fp32 storage + exact-fp32 MFMA (the net is 369 505 parameters; the step is launch-latency bound, not MFMA bound).
// The state dimension is padded to a multiple of 8 on the device (zero rows in the first-layer kernels).
#include <stdlib.h>
#include "common.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"
#include "ppo_fused.hpp"

using namespace mi;

namespace {

constexpr int PPO_TENSORS = 13;   // dense{k,b} dense_1{k,b} action_mean{k,b} action_logstd dense_2{k,b} dense_3{k,b} value{k,b}

inline long long pad8(long long n) { return (n + 7) / 8 * 8; }

struct PpoEngine {
    MiPpoDesc d;
    int kin;                                   // padded input dim
    long long off[PPO_TENSORS], size[PPO_TENSORS], total;
    float *params, *params_old, *grads, *m, *v;
    char* ws;
    // workspace offsets (bytes)
    long long s_pad, h1, h2, g1, g2, u, vraw, h1o, h2o, uo, du, dv, dh2, dh1, dg2, dg1, partial, losses, mean, low, high, ws_total;
    long long f_gslab;                            // per-chunk gradient slabs of the large-minibatch step (max_batch > 256): [ceil(max_batch / 256)][total]
    long long f_h1, f_h2, f_dh1, f_dh2, f_part;   // fused step (ppo_fused.hip): [3][M][H1], [3][M][H2], [2][M][H1], [2][M][H2], loss-block partials
    int last_M;
    hipStream_t side; hipEvent_t ev_fork, ev_join; int side_ok;     // second stream of the forked minibatch step (0 = not tried, 1 = ok, -1 = unavailable)
    float* P(int t) const { return params + off[t]; }
    float* PO(int t) const { return params_old + off[t]; }
    float* G(int t) const { return grads + off[t]; }
    void* at(long long o) const { return ws + o; }
};

void layout(PpoEngine& e) {
    const MiPpoDesc& d = e.d;
    e.kin = (d.input_dim + 7) / 8 * 8;
    long long o = 0; int t = 0;
    auto add = [&](long long n) { e.off[t] = o; e.size[t] = n; o += pad8(n); ++t; };
    add((long long)e.kin * d.h1); add(d.h1); add((long long)d.h1 * d.h2); add(d.h2); add((long long)d.h2 * d.num_actions); add(d.num_actions);
    add(d.num_actions);
    add((long long)e.kin * d.h1); add(d.h1); add((long long)d.h1 * d.h2); add(d.h2); add(d.h2); add(1);
    e.total = o;
    const long long M = d.max_batch;
    long long w = 0;
    auto wa = [&](long long bytes) { long long r = w; w += (bytes + 255) / 256 * 256; return r; };
    e.s_pad = wa(M * e.kin * 4);
    e.h1 = wa(M * d.h1 * 4); e.h2 = wa(M * d.h2 * 4); e.g1 = wa(M * d.h1 * 4); e.g2 = wa(M * d.h2 * 4);
    e.u = wa(M * d.num_actions * 4); e.vraw = wa(M * 4);
    e.h1o = wa(M * d.h1 * 4); e.h2o = wa(M * d.h2 * 4); e.uo = wa(M * d.num_actions * 4);
    e.du = wa(M * d.num_actions * 4); e.dv = wa(M * 4);
    e.dh2 = wa(M * d.h2 * 4); e.dh1 = wa(M * d.h1 * 4); e.dg2 = wa(M * d.h2 * 4); e.dg1 = wa(M * d.h1 * 4);
    e.partial = wa((long long)mi_ppo_loss_partial_floats((int)M) * 4);
    e.losses = wa(256); e.mean = wa(M * d.num_actions * 4); e.low = wa(256); e.high = wa(256);
    e.f_h1 = wa(3 * M * d.h1 * 4); e.f_h2 = wa(3 * M * d.h2 * 4); e.f_dh1 = wa(2 * M * d.h1 * 4); e.f_dh2 = wa(2 * M * d.h2 * 4);
    e.f_part = wa((long long)mi_ppo_fused_partial_floats((int)M) * 4);
    e.f_gslab = wa(M > 256 ? ((M + 255) / 256) * e.total * 4 : 0);
    e.ws_total = w;
}

bool valid_desc(const MiPpoDesc* d) {
    return d && d->max_batch >= 1 && d->input_dim >= 1 && d->num_actions >= 1 && d->num_actions <= 8 && d->h1 % 4 == 0 && d->h2 % 4 == 0 && d->h1 > 0 && d->h2 > 0;
}

// states [M, input_dim] -> zero-padded [M, kin]
__global__ void pad_states_kernel(const float* __restrict__ s, int M, int din, int kin, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * kin) return;
    const int r = i / kin, c = i - r * kin;
    out[i] = c < din ? s[(long long)r * din + c] : 0.f;
}

// gradient through the two tiny heads back into the trunks (K = num_actions / 1 is below the MFMA vector width):
//   dh2[m,j] = (h2>0) * sum_a du[m,a] * Wm[j,a] ;  dg2[m,j] = (g2>0) * dv[m] * Wv[j]
__global__ void head_dgrad_kernel(const float* __restrict__ du, const float* __restrict__ dv, const float* __restrict__ Wm,
                                  const float* __restrict__ Wv, const float* __restrict__ h2, const float* __restrict__ g2,
                                  int M, int H, int A, float* __restrict__ dh2, float* __restrict__ dg2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * H) return;
    const int m = i / H, j = i - m * H;
    float s = 0.f;
    for (int a = 0; a < A; ++a) s += du[(long long)m * A + a] * Wm[(long long)j * A + a];
    dh2[i] = h2[i] > 0.f ? s : 0.f;
    dg2[i] = g2[i] > 0.f ? dv[m] * Wv[j] : 0.f;
}

#define CK(call) do { int rc__ = (call); if (rc__ != MI_OK) return rc__; } while (0)

bool fused_switch() {                                    // MI355_PPO_FUSED=0: the first-generation step (one launch per layer op), for A/B runs
    static int on = -1;
    if (on < 0) { const char* ev = getenv("MI355_PPO_FUSED"); on = (ev && ev[0] == '0') ? 0 : 1; }
    return on != 0;
}
// the fused kernels serve the shapes they were built for (the reference's 500 / 300 trunk, <= 8 actions); anything else takes the per-layer path
bool fused_enabled(const PpoEngine* e);

bool fused_enabled(const PpoEngine* e) { return fused_switch() && mi_ppo_fused_shape_in_range(e->d.num_actions, e->d.h2, e->kin); }

void fill_fused(const PpoEngine* e, PpoFusedParams& q, const float* states, int M) {
    const MiPpoDesc& d = e->d;
    q = PpoFusedParams{};
    q.theta = e->params; q.theta_old = e->params_old; q.adam_m = e->m; q.adam_v = e->v; q.grads = e->grads;
    for (int i = 0; i < 13; ++i) q.off[i] = e->off[i];
    q.kin = e->kin; q.din = d.input_dim; q.H1 = d.h1; q.H2 = d.h2; q.A = d.num_actions; q.M = M; q.n_params = e->total;
    q.states = states; q.low = (const float*)e->at(e->low); q.high = (const float*)e->at(e->high);
    q.h1 = (float*)e->at(e->f_h1); q.h2 = (float*)e->at(e->f_h2); q.dh1 = (float*)e->at(e->f_dh1); q.dh2 = (float*)e->at(e->f_dh2);
    q.du = (float*)e->at(e->du); q.dv = (float*)e->at(e->dv); q.partial = (float*)e->at(e->f_part); q.losses = (float*)e->at(e->losses);
    q.mean_out = (float*)e->at(e->mean);
    q.gslab = d.max_batch > 256 ? (float*)e->at(e->f_gslab) : nullptr; q.gslab_stride = e->total;      // (total is a multiple of 8 floats: pad8 per tensor)
    q.n_nets = 3;
    q.clip_eps = d.clip_eps; q.value_scale = d.value_scale; q.entropy_scale = d.entropy_scale;
}

int policy_fwd(PpoEngine* e, void* st, const float* prm, const long long* off, int M, void* h1, void* h2, void* u) {
    const MiPpoDesc& d = e->d;
    CK(mi_gemm_bias_act(st, MI_F32, e->at(e->s_pad), M, e->kin, prm + off[0], 0, d.h1, prm + off[1], 1, nullptr, h1, 1, 1));
    CK(mi_gemm_bias_act(st, MI_F32, h1, M, d.h1, prm + off[2], 0, d.h2, prm + off[3], 1, nullptr, h2, 1, 1));
    return mi_gemm_bias_act(st, MI_F32, h2, M, d.h2, prm + off[4], 0, d.num_actions, prm + off[5], 0, nullptr, u, 1, 1);
}
int value_fwd(PpoEngine* e, void* st, const float* prm, const long long* off, int M, void* vraw) {
    const MiPpoDesc& d = e->d;
    CK(mi_gemm_bias_act(st, MI_F32, e->at(e->s_pad), M, e->kin, prm + off[7], 0, d.h1, prm + off[8], 1, nullptr, e->at(e->g1), 1, 1));
    CK(mi_gemm_bias_act(st, MI_F32, e->at(e->g1), M, d.h1, prm + off[9], 0, d.h2, prm + off[10], 1, nullptr, e->at(e->g2), 1, 1));
    return mi_gemm_bias_act(st, MI_F32, e->at(e->g2), M, d.h2, prm + off[11], 0, 1, prm + off[12], 0, nullptr, vraw, 1, 1);
}
int trunk_fwd(PpoEngine* e, void* st, const float* prm, const long long* off, int M, bool value_branch, void* h1, void* h2, void* u, void* vraw) {
    CK(policy_fwd(e, st, prm, off, M, h1, h2, u));
    return value_branch ? value_fwd(e, st, prm, off, M, vraw) : MI_OK;
}

int stage_states(PpoEngine* e, void* st, const float* states, int M) {
    const int n = M * e->kin;
    hipLaunchKernelGGL(pad_states_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, states, M, e->d.input_dim, e->kin, (float*)e->at(e->s_pad));
    return mi_check_launch("pad_states");
}

}  // namespace

int mi_ppo_internal_fill(void* h, PpoFusedParams* q, const float* states, int M) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e || M < 1 || M > e->d.max_batch) return mi_fail(MI_ERR_ARG, "ppo engine: bad handle or batch");
    fill_fused(e, *q, states, M);
    return MI_OK;
}

extern "C" {

int mi_ppo_desc_size(void) { return (int)sizeof(MiPpoDesc); }
int mi_ppo_tensor_count(void) { return PPO_TENSORS; }

long long mi_ppo_param_floats(const MiPpoDesc* d) {
    if (!valid_desc(d)) { mi_fail(MI_ERR_SHAPE, "mi_ppo_param_floats: bad descriptor (hidden sizes must be multiples of 4, 1..8 actions)"); return -1; }
    PpoEngine e; e.d = *d; layout(e); return e.total;
}

// device order: dense{kernel [kin,h1], bias} dense_1{k,b} action_mean{k,b} action_logstd dense_2{k,b} dense_3{k,b} value{k,b}
// (first-layer kernels carry kin - input_dim zero rows at the end)
int mi_ppo_param_layout(const MiPpoDesc* d, long long* offsets, long long* sizes, int n) {
    if (!valid_desc(d)) return mi_fail(MI_ERR_SHAPE, "mi_ppo_param_layout: bad descriptor");
    if (n != PPO_TENSORS) return mi_fail(MI_ERR_ARG, "mi_ppo_param_layout: expected 13 entries");
    PpoEngine e; e.d = *d; layout(e);
    for (int i = 0; i < PPO_TENSORS; ++i) { offsets[i] = e.off[i]; sizes[i] = e.size[i]; }
    return MI_OK;
}

long long mi_ppo_workspace_bytes(const MiPpoDesc* d) {
    if (!valid_desc(d)) { mi_fail(MI_ERR_SHAPE, "mi_ppo_workspace_bytes: bad descriptor"); return -1; }
    PpoEngine e; e.d = *d; layout(e); return e.ws_total;
}

void* mi_ppo_create(const MiPpoDesc* d, float* params, float* params_old, float* grads, float* adam_m, float* adam_v,
                    void* workspace, long long workspace_bytes, const float* action_low, const float* action_high) {
    if (!valid_desc(d)) { mi_fail(MI_ERR_SHAPE, "mi_ppo_create: bad descriptor"); return nullptr; }
    PpoEngine* e = (PpoEngine*)calloc(1, sizeof(PpoEngine));
    if (!e) { mi_fail(MI_ERR_STATE, "mi_ppo_create: out of host memory"); return nullptr; }
    e->d = *d; layout(*e);
    if (!params || !params_old || !workspace || workspace_bytes < e->ws_total || !action_low || !action_high) { free(e); mi_fail(MI_ERR_ARG, "mi_ppo_create: missing buffers or workspace too small"); return nullptr; }
    if ((((uintptr_t)params) | ((uintptr_t)params_old) | ((uintptr_t)workspace) | ((uintptr_t)grads)) & 255) { free(e); mi_fail(MI_ERR_ARG, "mi_ppo_create: buffers must be 256-byte aligned"); return nullptr; }
    e->params = params; e->params_old = params_old; e->grads = grads; e->m = adam_m; e->v = adam_v; e->ws = (char*)workspace;
    // action bounds are host arrays (gym Box.low/high); keep a device copy in the workspace
    if (hipMemcpy(e->at(e->low), action_low, d->num_actions * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->at(e->high), action_high, d->num_actions * 4, hipMemcpyHostToDevice) != hipSuccess) {
        free(e); mi_fail(MI_ERR_STATE, "mi_ppo_create: copying action bounds failed"); return nullptr;
    }
    return e;
}

void mi_ppo_destroy(void* h) {
    PpoEngine* e = (PpoEngine*)h;
    if (e && e->side_ok == 1) { hipStreamDestroy(e->side); hipEventDestroy(e->ev_fork); hipEventDestroy(e->ev_join); }
    free(h);
}

// 0 losses[5] (policy, value, entropy, total, mean ratio)   1 action_mean [M,A] of the last predict
void* mi_ppo_buffer(void* h, int which) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e) return nullptr;
    return which == 0 ? e->at(e->losses) : which == 1 ? e->at(e->mean) : nullptr;
}

// PPO.update_old_policy (ppo.py:275-276): theta_old <- theta, one device copy of the flat buffer
int mi_ppo_update_old(void* h, void* stream) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "ppo engine: null handle");
    if (hipMemcpyAsync(e->params_old, e->params, (size_t)e->total * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return mi_fail(MI_ERR_LAUNCH, "mi_ppo_update_old: copy failed");
    return MI_OK;
}

// PPO.predict (ppo.py:231-251): states [M,input_dim] -> action [M,A] (sampled+clipped, or greedy mean), value [M]
int mi_ppo_predict(void* h, void* stream, const float* states, int M, const float* noise, int greedy, float* action, float* value) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "ppo engine: null handle");
    if (M < 1 || M > e->d.max_batch) return mi_fail(MI_ERR_ARG, "mi_ppo_predict: batch outside [1, max_batch]");
    if (!greedy && !noise) return mi_fail(MI_ERR_ARG, "mi_ppo_predict: sampling needs noise");
    if (fused_enabled(e)) {                                // 3 launches: layer 1, layer 2 (policy + value), heads
        PpoFusedParams q; fill_fused(e, q, states, M);
        return mi_ppo_fused_predict((hipStream_t)stream, q, noise, greedy, action, value);
    }
    CK(stage_states(e, stream, states, M));
    CK(trunk_fwd(e, stream, e->params, e->off, M, true, e->at(e->h1), e->at(e->h2), e->at(e->u), value));
    return mi_policy_head(stream, (const float*)e->at(e->u), e->P(6), noise, (const float*)e->at(e->low), (const float*)e->at(e->high),
                          M, e->d.num_actions, greedy, action, (float*)e->at(e->mean));
}

// Forward + loss + backward of one minibatch (the gradient half of PPO.train, ppo.py:218-229) into the flat gradient
// buffer (zero on entry).  inv_m = 1/M_global, grad_scale = M_local/M_global (data parallel; 1 for a single GPU).
int mi_ppo_forward_backward(void* h, void* stream, const float* states, const float* actions, const float* returns, const float* advantage,
                            int M, float inv_m, float grad_scale) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "ppo engine: null handle");
    if (M < 1 || M > e->d.max_batch) return mi_fail(MI_ERR_ARG, "mi_ppo_forward_backward: batch outside [1, max_batch]");
    if (!e->grads) return mi_fail(MI_ERR_STATE, "mi_ppo_forward_backward: engine created without a gradient buffer");
    if (fused_enabled(e)) {                                // 5 launches; gradients written (not accumulated) into the flat buffer (M > 256: the row chunks' partial sums are added in a fixed order, round 4)
        PpoFusedParams q; fill_fused(e, q, states, M);
        q.actions = actions; q.returns = returns; q.adv = advantage; q.inv_m = inv_m; q.grad_scale = grad_scale;
        e->last_M = M;
        return mi_ppo_fused_step((hipStream_t)stream, q, 0);
    }
    const MiPpoDesc& d = e->d;
    void* st = stream;
    const int A = d.num_actions;
    // The step is ~28 launches of 5-9 us kernels that leave the chip almost empty, and the policy and value networks only meet in the loss:
    // the value side (and the old policy's forward) runs on a second stream.  Forward: policy(theta) | value(theta) + policy(theta_old);
    // backward: policy head + trunk | value head + trunk.  OFF by default (MI355_PPO_STREAMS=1 enables it): with device-resident minibatches
    // the step drops from 181 to 172 us, but the four event operations raise the host cost per step from 90 to 135 us, and the reference's
    // calling pattern (one host minibatch per train() call) is host-bound: 3.25 -> 3.6 ms per update of 16 steps.
    static int two_streams = -1;
    if (two_streams < 0) { const char* ev = getenv("MI355_PPO_STREAMS"); two_streams = (ev && ev[0] == '1') ? 1 : 0; }
    if (two_streams && !e->side_ok) {
        if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming) == hipSuccess) e->side_ok = 1;
        else e->side_ok = -1;
    }
    const bool fork = two_streams && e->side_ok == 1;
    void* sw = fork ? (void*)e->side : st;
    auto release = [&]() { if (fork) { hipEventRecord(e->ev_fork, (hipStream_t)st); hipStreamWaitEvent(e->side, e->ev_fork, 0); } };
    auto join = [&]() { if (fork) { hipEventRecord(e->ev_join, e->side); hipStreamWaitEvent((hipStream_t)st, e->ev_join, 0); } };
    CK(stage_states(e, st, states, M));
    release();
    CK(policy_fwd(e, st, e->params, e->off, M, e->at(e->h1), e->at(e->h2), e->at(e->u)));
    CK(value_fwd(e, sw, e->params, e->off, M, e->at(e->vraw)));
    CK(policy_fwd(e, sw, e->params_old, e->off, M, e->at(e->h1o), e->at(e->h2o), e->at(e->uo)));
    join();
    CK(mi_ppo_loss_fwd_bwd(st, (const float*)e->at(e->u), (const float*)e->at(e->uo), e->P(6), e->PO(6), (const float*)e->at(e->vraw), actions, returns, advantage,
                           (const float*)e->at(e->low), (const float*)e->at(e->high), M, A, d.clip_eps, d.value_scale, d.entropy_scale, inv_m, grad_scale,
                           (float*)e->at(e->du), (float*)e->at(e->dv), (float*)e->at(e->partial), (float*)e->at(e->losses), e->G(6)));
    {   // gradient through the two heads into both trunks (needs du and dv: before the fork)
        const int n = M * d.h2;
        hipLaunchKernelGGL(head_dgrad_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)st, (const float*)e->at(e->du), (const float*)e->at(e->dv), e->P(4), e->P(11),
                           (const float*)e->at(e->h2), (const float*)e->at(e->g2), M, d.h2, A, (float*)e->at(e->dh2), (float*)e->at(e->dg2));
        CK(mi_check_launch("head_dgrad"));
    }
    release();
    // policy side (caller's stream): head, trunk
    CK(mi_colsum(st, MI_F32, e->at(e->du), M, A, e->G(5)));
    CK(mi_gemm_wgrad(st, MI_F32, e->at(e->h2), e->at(e->du), M, d.h2, A, e->G(4)));
    CK(mi_colsum(st, MI_F32, e->at(e->dh2), M, d.h2, e->G(3)));
    CK(mi_gemm_wgrad(st, MI_F32, e->at(e->h1), e->at(e->dh2), M, d.h1, d.h2, e->G(2)));
    CK(mi_gemm_bias_act(st, MI_F32, e->at(e->dh2), M, d.h2, e->P(2), 1, d.h1, nullptr, 0, e->at(e->h1), e->at(e->dh1), 1, 1));
    CK(mi_colsum(st, MI_F32, e->at(e->dh1), M, d.h1, e->G(1)));
    CK(mi_gemm_wgrad(st, MI_F32, e->at(e->s_pad), e->at(e->dh1), M, e->kin, d.h1, e->G(0)));
    // value side (second stream): head, trunk
    CK(mi_colsum(sw, MI_F32, e->at(e->dv), M, 1, e->G(12)));
    CK(mi_gemm_wgrad(sw, MI_F32, e->at(e->g2), e->at(e->dv), M, d.h2, 1, e->G(11)));
    CK(mi_colsum(sw, MI_F32, e->at(e->dg2), M, d.h2, e->G(10)));
    CK(mi_gemm_wgrad(sw, MI_F32, e->at(e->g1), e->at(e->dg2), M, d.h1, d.h2, e->G(9)));
    CK(mi_gemm_bias_act(sw, MI_F32, e->at(e->dg2), M, d.h2, e->P(9), 1, d.h1, nullptr, 0, e->at(e->g1), e->at(e->dg1), 1, 1));
    CK(mi_colsum(sw, MI_F32, e->at(e->dg1), M, d.h1, e->G(8)));
    CK(mi_gemm_wgrad(sw, MI_F32, e->at(e->s_pad), e->at(e->dg1), M, e->kin, d.h1, e->G(7)));
    join();
    e->last_M = M;
    return MI_OK;
}

// The whole of PPO.train's device work (ppo.py:218-229) in one call, single rank: forward of policy / value / old policy, losses, backward
// and tf.train.AdamOptimizer, as five launches (ppo_fused.hip); the optimiser update is applied by the blocks that produce each gradient
// tile, so no gradient buffer is involved.  logp_old != NULL: log pi_old(a|s) of these samples, computed once per horizon batch with
// mi_ppo_logp_old (theta_old is constant between two update_old_policy() calls): the old policy's forward pass is skipped.
int mi_ppo_train_step(void* h, void* stream, const float* states, const float* actions, const float* returns, const float* advantage, const float* logp_old,
                      int M, float inv_m, float grad_scale, float alpha, float beta1, float beta2, float epsilon) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "ppo engine: null handle");
    if (M < 1 || M > e->d.max_batch) return mi_fail(MI_ERR_ARG, "mi_ppo_train_step: batch outside [1, max_batch]");
    if (!e->grads || !e->m || !e->v) return mi_fail(MI_ERR_STATE, "mi_ppo_train_step: engine created without optimiser buffers");
    if (M > 256 || !fused_enabled(e)) {
        // large minibatches (the synthetic replay: 2048 rows per GPU): the weight-gradient launch splits the rows into chunks of 256 whose partial
        // sums are added in a fixed order (round 4: no atomics), so the Adam update is its own (flat) launch; everything before it is the same five-kernel chain
        if (!fused_enabled(e)) { CK(mi_ppo_forward_backward(h, stream, states, actions, returns, advantage, M, inv_m, grad_scale)); }
        else {
            PpoFusedParams q; fill_fused(e, q, states, M);
            q.actions = actions; q.returns = returns; q.adv = advantage; q.inv_m = inv_m; q.grad_scale = grad_scale;
            q.logp_old = logp_old; q.n_nets = logp_old ? 2 : 3;
            e->last_M = M;
            CK(mi_ppo_fused_step((hipStream_t)stream, q, 0));
        }
        return mi_ppo_apply_adam(h, stream, alpha, beta1, beta2, epsilon);
    }
    PpoFusedParams q; fill_fused(e, q, states, M);
    q.actions = actions; q.returns = returns; q.adv = advantage; q.inv_m = inv_m; q.grad_scale = grad_scale;
    q.logp_old = logp_old; q.n_nets = logp_old ? 2 : 3;
    q.alpha = alpha; q.omb1 = 1.0f - beta1; q.omb2 = 1.0f - beta2; q.epsilon = epsilon;
    e->last_M = M;
    return mi_ppo_fused_step((hipStream_t)stream, q, 1);
}

// 1: this engine's shape (num_actions, hidden sizes, input width) is inside the range of the fused kernels (csrc/ppo_fused.hip) and they are switched on, i.e.
// mi_ppo_train_step_idx / mi_ppo_logp_old will run; 0: only the per-layer path (mi_ppo_train_step / mi_ppo_forward_backward fall back to it by themselves;
// the _idx form has no per-layer equivalent: gather on the host side).  The host mirror asks before it picks the in-kernel gather.
int mi_ppo_fused_shape_ok(void* h) {
    PpoEngine* e = (PpoEngine*)h;
    return (e && fused_enabled(e)) ? 1 : 0;
}

// mi_ppo_train_step with the minibatch GATHER fused in: states / actions / returns / advantage / logp_old are the horizon-batch tables (n_rows rows;
// device resident for the whole update, train.py:175-207) and row_idx [M] (int32, device) names the rows of this minibatch -- the reference's
// `states[mb_idx]` fancy-indexing (train.py:199-204) happens inside the kernels that read the operands instead of as five gather launches per step.
int mi_ppo_train_step_idx(void* h, void* stream, const float* states, const float* actions, const float* returns, const float* advantage, const float* logp_old,
                          const int* row_idx, int n_rows, int M, float inv_m, float grad_scale, float alpha, float beta1, float beta2, float epsilon) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "ppo engine: null handle");
    if (M < 1 || M > e->d.max_batch || n_rows < 1) return mi_fail(MI_ERR_ARG, "mi_ppo_train_step_idx: batch outside [1, max_batch] or empty tables");
    if (!row_idx) return mi_fail(MI_ERR_ARG, "mi_ppo_train_step_idx: missing row index (use mi_ppo_train_step for contiguous minibatches)");
    if (!e->grads || !e->m || !e->v) return mi_fail(MI_ERR_STATE, "mi_ppo_train_step_idx: engine created without optimiser buffers");
    if (!fused_enabled(e)) return mi_fail(MI_ERR_SHAPE, "mi_ppo_train_step_idx: needs the fused kernels (shape outside their range or MI355_PPO_FUSED=0): gather on the host side and call mi_ppo_train_step");
    PpoFusedParams q; fill_fused(e, q, states, M);
    q.actions = actions; q.returns = returns; q.adv = advantage; q.inv_m = inv_m; q.grad_scale = grad_scale;
    q.logp_old = logp_old; q.n_nets = logp_old ? 2 : 3;
    q.row_idx = row_idx; q.n_rows = n_rows; q.s_gath = (float*)e->at(e->s_pad);
    e->last_M = M;
    if (M > 256) {
        CK(mi_ppo_fused_step((hipStream_t)stream, q, 0));
        return mi_ppo_apply_adam(h, stream, alpha, beta1, beta2, epsilon);
    }
    q.alpha = alpha; q.omb1 = 1.0f - beta1; q.omb2 = 1.0f - beta2; q.epsilon = epsilon;
    return mi_ppo_fused_step((hipStream_t)stream, q, 1);
}

// One DATA-PARALLEL SGD step of PPO.train (ppo.py:218-229 behind train.py:193-207) in ONE call (round 6): the fused five-launch chain of
// mi_ppo_train_step[_idx] with the gradients left in the flat buffer (sums over this rank's M rows / M_global: inv_m = 1 / M_global, grad_scale = M / M_global), ONE
// all-reduce of that buffer (1.48 MB: one bucket, in stream order -- nothing of this step is left to overlap it with), tf.train.AdamOptimizer.  row_idx != NULL: the
// minibatch gather of train.py:199-204 stays inside the kernels (states / actions / returns / advantage / logp_old are this rank's horizon-batch tables of n_rows rows);
// row_idx == NULL: contiguous minibatch tensors (n_rows ignored).  comm: mi_comm_init (or a recording communicator).  Before round 6 the host issued
// forward_backward -> a blocking Python-side all-reduce -> apply_adam and gathered the rows itself whenever world_size > 1.
int mi_ppo_train_step_dp(void* h, void* comm, void* stream, const float* states, const float* actions, const float* returns, const float* advantage, const float* logp_old,
                         const int* row_idx, int n_rows, int M, float inv_m, float grad_scale, float alpha, float beta1, float beta2, float epsilon) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "ppo engine: null handle");
    if (!comm) return mi_fail(MI_ERR_ARG, "mi_ppo_train_step_dp: null communicator (single rank: mi_ppo_train_step)");
    if (M < 1 || M > e->d.max_batch || (row_idx && n_rows < 1)) return mi_fail(MI_ERR_ARG, "mi_ppo_train_step_dp: batch outside [1, max_batch] or empty tables");
    if (!e->grads || !e->m || !e->v) return mi_fail(MI_ERR_STATE, "mi_ppo_train_step_dp: engine created without optimiser buffers");
    if (fused_enabled(e)) {
        PpoFusedParams q; fill_fused(e, q, states, M);
        q.actions = actions; q.returns = returns; q.adv = advantage; q.inv_m = inv_m; q.grad_scale = grad_scale;
        q.logp_old = logp_old; q.n_nets = logp_old ? 2 : 3;
        if (row_idx) { q.row_idx = row_idx; q.n_rows = n_rows; q.s_gath = (float*)e->at(e->s_pad); }
        e->last_M = M;
        CK(mi_ppo_fused_step((hipStream_t)stream, q, 0));      // gradients written into the flat buffer (M > 256: ordered row chunks)
    } else {
        if (row_idx) return mi_fail(MI_ERR_SHAPE, "mi_ppo_train_step_dp: the in-kernel gather needs the fused kernels (shape outside their range or MI355_PPO_FUSED=0): gather on the host side and pass row_idx = NULL");
        CK(mi_ppo_forward_backward(h, stream, states, actions, returns, advantage, M, inv_m, grad_scale));
    }
    CK(mi_allreduce_sum_f32(comm, stream, e->grads, e->total));
    return mi_ppo_apply_adam(h, stream, alpha, beta1, beta2, epsilon);
}

// log pi_old(a | s) of M samples under theta_old -> out [M] (the per-horizon cache for mi_ppo_train_step)
int mi_ppo_logp_old(void* h, void* stream, const float* states, const float* actions, int M, float* out) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "ppo engine: null handle");
    if (M < 1 || M > e->d.max_batch) return mi_fail(MI_ERR_ARG, "mi_ppo_logp_old: batch outside [1, max_batch]");
    PpoFusedParams q; fill_fused(e, q, states, M);
    q.actions = actions;
    return mi_ppo_fused_logp_old((hipStream_t)stream, q, out);
}

// tf.train.AdamOptimizer over the 13 policy/ variables (ppo.py:143-144); alpha folds lr*lr_decay^episode and the bias correction
int mi_ppo_apply_adam(void* h, void* stream, float alpha, float beta1, float beta2, float epsilon) {
    PpoEngine* e = (PpoEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "ppo engine: null handle");
    if (!e->grads || !e->m || !e->v) return mi_fail(MI_ERR_STATE, "mi_ppo_apply_adam: engine created without optimiser buffers");
    return mi_adam_tf_flat(stream, e->params, e->m, e->v, e->grads, e->total, alpha, beta1, beta2, epsilon, nullptr, 1);
}

}  // extern "C"
