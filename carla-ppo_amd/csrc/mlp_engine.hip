// mlp_engine.hip — MlpVAE training / inference step behind the C ABI (round 4): the dense encoder / decoder variant of the reference's VAE
// (vae/models.py:271-299 on top of the base graph :85-142) as an engine like vae_engine.hip -- one C call per SGD step, host C++ only, every FLOP in the kernels of
// gemm_core.hpp / elementwise.hip.  The engine owns NO device memory: parameter / optimiser / workspace buffers are caller-provided (torch tensors).
//
//   x [B, S] -> dense + relu (encoder_sizes) -> [mean | logstd_sq] dense -> z = mean + exp(.5 lv) eps -> dense + relu (decoder_sizes) -> dense -> logits [B, P]; ELBO; TF-Adam
//
// What the engine form buys over the host-sequenced op calls of rounds 1-3 (39.5 M parameters at the reference's sizes, 1.0 ms per step at batch 512):
//   * the frame rows of a minibatch are gathered AND converted in one launch (mi_gather_rows_cast: index_select + contiguous + cast were three passes over 79 MB);
//   * the bias gradients are a ones row of the filter gradients, and the filter gradients STORE their result (mi_gemm_wgrad_bias_set: no atomics on 79 MB per big
//     layer, no zeroing of the 158 MB gradient buffer by the optimiser);
//   * Adam writes both weight layouts the MFMA kernels read (mi_adam_tf_layouts: no transpose pass over the master weights behind every step).
#include <stdlib.h>
#include <string.h>
#include "common.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"

namespace {

constexpr int ML_MAX = MI_MLP_MAX_HIDDEN;                 // hidden layers per side
constexpr long long SPLIT_K = 4096;                       // reductions at least this long are split over the chip (38400 -> 30 slabs) and finished by mi_splitk_finish

struct Dense { long long wo, bo; int K, N; };             // kernel [K, N] at wo, bias [N] at bo (floats into the flat buffers)

struct MlpEngine {
    MiMlpVaeDesc d;
    int ne, nd;                                           // encoder hidden layers; decoder layers INCLUDING the output layer
    Dense enc[ML_MAX], heads, dec[ML_MAX + 1];
    long long total, decoder_offset;
    float *params, *grads, *m, *v;
    void *shadow, *wt;
    char* ws; long long ws_bytes;
    long long o_x, o_h[ML_MAX], o_heads, o_mean, o_logvar, o_klrow, o_z, o_d[ML_MAX + 1], o_partial, o_gd[ML_MAX + 1], o_dz, o_dheads, o_gh[ML_MAX], o_slab, o_scratch, o_out2, ws_total;
    long long slab_bytes, scratch_bytes;
    int nchunks, esz;
    int last_B;
    const void* last_x;
    hipStream_t side; hipEvent_t ev_fork, ev_chain, ev_done; int side_ok;      // second stream of a full backward pass (created on first use)
    char* at(long long o) const { return ws + o; }
    const void* w(const Dense& l) const { return d.dtype == MI_BF16 ? (const void*)((const unsigned short*)shadow + l.wo) : (const void*)(params + l.wo); }
    const void* w_t(const Dense& l) const { return d.dtype == MI_BF16 ? (const void*)((const unsigned short*)wt + l.wo) : (const void*)((const float*)wt + l.wo); }
    const float* b(const Dense& l) const { return params + l.bo; }
};

inline long long al256(long long n) { return (n + 255) / 256 * 256; }

int splitk_slabs(long long K) {
    if (K < SPLIT_K || K % 128 != 0) return 1;
    int ns = (int)(K / 1280);
    if (ns > 32) ns = 32;
    if (ns < 1) ns = 1;
    while (ns > 1 && K % ((long long)ns * 128) != 0) --ns;
    return ns;
}

bool init_engine(MlpEngine& e, const MiMlpVaeDesc* dp, int max_batch_override = -1) {
    memset(&e, 0, sizeof(e));
    e.d = *dp;
    const MiMlpVaeDesc& d = e.d;
    if (d.dtype != MI_F32 && d.dtype != MI_BF16) return false;        // (split storage is the ConvVAE engine's)
    if (d.n_enc < 1 || d.n_enc > ML_MAX || d.n_dec < 1 || d.n_dec > ML_MAX || d.source_size < 1 || d.target_size < 1 || d.z_dim < 1) return false;
    const int vec = d.dtype == MI_BF16 ? 8 : 4;
    if (d.source_size % vec || d.target_size % vec || d.z_dim % vec) return false;
    for (int i = 0; i < d.n_enc; ++i) if (d.enc[i] < 1 || d.enc[i] % vec) return false;
    for (int i = 0; i < d.n_dec; ++i) if (d.dec[i] < 1 || d.dec[i] % vec) return false;
    e.esz = d.dtype == MI_BF16 ? 2 : 4;
    e.ne = d.n_enc; e.nd = d.n_dec + 1;
    long long o = 0;
    auto add = [&](Dense& l, int K, int N) { l.K = K; l.N = N; l.wo = o; o += (long long)K * N; l.bo = o; o += N; };
    for (int i = 0; i < e.ne; ++i) add(e.enc[i], i ? d.enc[i - 1] : d.source_size, d.enc[i]);
    add(e.heads, d.enc[e.ne - 1], 2 * d.z_dim);
    e.decoder_offset = o;
    for (int i = 0; i < e.nd; ++i) add(e.dec[i], i ? d.dec[i - 1] : d.z_dim, i + 1 < e.nd ? d.dec[i] : d.target_size);
    e.total = o;
    e.nchunks = mi_recon_loss_chunks(d.target_size);
    // workspace
    const long long B = max_batch_override > 0 ? max_batch_override : d.max_batch;
    if (B < 1) return false;
    long long w = 0;
    auto reg = [&](long long bytes) { const long long at = w; w += al256(bytes); return at; };
    e.o_x = reg(B * d.source_size * e.esz);
    for (int i = 0; i < e.ne; ++i) e.o_h[i] = reg(B * e.enc[i].N * e.esz);
    e.o_heads = reg(B * 2 * d.z_dim * 4);
    e.o_mean = reg(B * d.z_dim * 4); e.o_logvar = reg(B * d.z_dim * 4); e.o_klrow = reg(B * 4);
    e.o_z = reg(B * d.z_dim * e.esz);
    for (int i = 0; i < e.nd; ++i) e.o_d[i] = reg(B * e.dec[i].N * e.esz);
    e.o_partial = reg(B * e.nchunks * 4);
    e.o_out2 = reg(16);
    long long slab = 0, scr = 0;
    auto use = [&](long long K, long long N) { const int ns = splitk_slabs(K); if (ns > 1 && ns * B * N * 4 > slab) slab = ns * B * N * 4; };
    for (int i = 0; i < e.ne; ++i) use(e.enc[i].K, e.enc[i].N);
    use(e.heads.K, e.heads.N);
    for (int i = 0; i < e.nd; ++i) use(e.dec[i].K, e.dec[i].N);
    if (d.with_optimizer) {
        for (int i = 0; i < e.nd; ++i) e.o_gd[i] = reg(B * e.dec[i].N * e.esz);
        e.o_dz = reg(B * d.z_dim * 4);
        e.o_dheads = reg(B * 2 * d.z_dim * e.esz);
        for (int i = 0; i < e.ne; ++i) e.o_gh[i] = reg(B * e.enc[i].N * e.esz);
        auto sc = [&](const Dense& l) { scr += al256(mi_gemm_wgrad_scratch_bytes(d.dtype, 1 << 20, l.K, l.N)); };   // (the bound over every row count; every layer its own piece: the slab sums of a pass are ONE launch at its end)
        for (int i = 0; i < e.ne; ++i) { sc(e.enc[i]); if (i) use(e.enc[i].N, e.enc[i].K); }       // (input gradients: x * W^T reduces over N)
        sc(e.heads); use(e.heads.N, e.heads.K);
        for (int i = 0; i < e.nd; ++i) { sc(e.dec[i]); use(e.dec[i].N, e.dec[i].K); }
    }
    e.slab_bytes = slab; e.scratch_bytes = scr;
    e.o_slab = reg(slab ? slab : 16);
    e.o_scratch = reg(scr ? scr : 16);
    e.ws_total = w;
    return true;
}

#define CK(x) do { const int rc__ = (x); if (rc__ != MI_OK) return rc__; } while (0)

int check_batch(const MlpEngine* e, int B) {
    if (!e) return mi_fail(MI_ERR_STATE, "mlp vae engine: null handle");
    if (B < 1 || B > e->d.max_batch) return mi_fail(MI_ERR_ARG, "mlp vae engine: batch outside 1 .. max_batch");
    return MI_OK;
}

// out = mask(act(a W + bias)).  layout 0: forward, through the K-contiguous copy wt[N][K]; 1: input gradient x * W^T on the [K_layer, N_layer] original (which IS
// K-contiguous for that product: rows = the output index).  Long reductions are split over the chip and finished by mi_splitk_finish.
int dense(MlpEngine* e, void* st, const void* a, int M, int K, const void* w, int N, const float* bias, int relu, const void* mask, void* out, int out_f32) {
    const int ns = splitk_slabs(K);
    if (ns > 1 && (long long)ns * M * N * 4 <= e->slab_bytes) {
        float* slab = (float*)e->at(e->o_slab);
        CK(mi_gemm_bias_act(st, e->d.dtype, a, M, K, w, 1, N, nullptr, 0, nullptr, slab, 1, ns));
        return mi_splitk_finish(st, e->d.dtype, slab, ns, M, N, bias, relu, mask, out, out_f32);
    }
    return mi_gemm_bias_act(st, e->d.dtype, a, M, K, w, 1, N, bias, relu, mask, out, out_f32, 1);
}

// rows idx[0 .. B) (or the first B rows) of the float32 frame table in the engine's storage type
// (src_u8: the table holds raw uint8 camera bytes, normalised to float32(k) / float32(255) while the rows are staged -- round 5)
int stage_input(MlpEngine* e, void* st, const void* src, int src_u8, const int* idx, int B, const void** x) {
    if (!src) return mi_fail(MI_ERR_ARG, "mlp vae engine: missing frame table");
    if (e->d.dtype == MI_F32 && !idx && !src_u8) { *x = src; return MI_OK; }
    if (src_u8) CK(mi_gather_rows_cast_u8(st, e->d.dtype, (const unsigned char*)src, idx, B, e->d.source_size, e->at(e->o_x)));
    else CK(mi_gather_rows_cast(st, e->d.dtype, (const float*)src, idx, B, e->d.source_size, e->at(e->o_x)));
    *x = e->at(e->o_x);
    return MI_OK;
}

int run_encoder(MlpEngine* e, void* st, const void* x, int B) {
    const void* a = x;
    for (int i = 0; i < e->ne; ++i) {
        const Dense& l = e->enc[i];
        CK(dense(e, st, a, B, l.K, e->w_t(l), l.N, e->b(l), 1, nullptr, e->at(e->o_h[i]), 0));
        a = e->at(e->o_h[i]);
    }
    // both heads as one [K, 2Z] product; their biases are added by the reparameterisation kernel
    return dense(e, st, a, B, e->heads.K, e->w_t(e->heads), e->heads.N, nullptr, 0, nullptr, e->at(e->o_heads), 1);
}

int reparam(MlpEngine* e, void* st, int B, const float* eps, int sample) {
    const int Z = e->d.z_dim;
    if (sample && !eps) return mi_fail(MI_ERR_ARG, "mlp vae engine: a sampling pass needs the noise tensor eps [B, z_dim]");
    return mi_vae_reparam_kl_fwd(st, e->d.dtype, (const float*)e->at(e->o_heads), 1, e->params + e->heads.bo, e->params + e->heads.bo + Z, eps, sample, B, Z,
                                 (float*)e->at(e->o_mean), (float*)e->at(e->o_logvar), e->at(e->o_z), (float*)e->at(e->o_klrow));
}

int run_decoder(MlpEngine* e, void* st, const void* z, int B) {
    const void* a = z;
    for (int i = 0; i < e->nd; ++i) {
        const Dense& l = e->dec[i];
        CK(dense(e, st, a, B, l.K, e->w_t(l), l.N, e->b(l), i + 1 < e->nd ? 1 : 0, nullptr, e->at(e->o_d[i]), 0));
        a = e->at(e->o_d[i]);
    }
    return MI_OK;
}

void kernel_table(const MlpEngine* e, long long* off, int* K, int* N, int* n) {
    int c = 0;
    for (int i = 0; i < e->ne; ++i) { off[c] = e->enc[i].wo; K[c] = e->enc[i].K; N[c] = e->enc[i].N; ++c; }
    off[c] = e->heads.wo; K[c] = e->heads.K; N[c] = e->heads.N; ++c;
    for (int i = 0; i < e->nd; ++i) { off[c] = e->dec[i].wo; K[c] = e->dec[i].K; N[c] = e->dec[i].N; ++c; }
    *n = c;
}

}  // namespace

extern "C" {

int mi_mlpvae_desc_size(void) { return (int)sizeof(MiMlpVaeDesc); }

long long mi_mlpvae_param_floats(const MiMlpVaeDesc* d) {
    MlpEngine e;
    if (!d || !init_engine(e, d, 1)) { mi_fail(MI_ERR_SHAPE, "mi_mlpvae_param_floats: unsupported sizes (1-4 hidden layers per side, every width a multiple of the 16-byte vector; fp32 or bf16)"); return -1; }
    return e.total;
}

// tensors in device order: encoder layers {kernel [K, N], bias [N]}, heads {kernel [K, 2Z] = [mean | logstd_sqare], bias [2Z]}, decoder layers {kernel, bias} incl. the output layer
int mi_mlpvae_tensor_count(const MiMlpVaeDesc* d) {
    MlpEngine e;
    if (!d || !init_engine(e, d, 1)) return mi_fail(MI_ERR_SHAPE, "mi_mlpvae_tensor_count: unsupported sizes");
    return 2 * (e.ne + 1 + e.nd);
}

int mi_mlpvae_param_layout(const MiMlpVaeDesc* d, long long* offsets, long long* sizes, int n) {
    MlpEngine e;
    if (!d || !init_engine(e, d, 1)) return mi_fail(MI_ERR_SHAPE, "mi_mlpvae_param_layout: unsupported sizes");
    if (!offsets || !sizes || n != 2 * (e.ne + 1 + e.nd)) return mi_fail(MI_ERR_ARG, "mi_mlpvae_param_layout: n must be mi_mlpvae_tensor_count()");
    int t = 0;
    auto put = [&](const Dense& l) { offsets[t] = l.wo; sizes[t] = (long long)l.K * l.N; ++t; offsets[t] = l.bo; sizes[t] = l.N; ++t; };
    for (int i = 0; i < e.ne; ++i) put(e.enc[i]);
    put(e.heads);
    for (int i = 0; i < e.nd; ++i) put(e.dec[i]);
    return MI_OK;
}

long long mi_mlpvae_workspace_bytes(const MiMlpVaeDesc* d) {
    MlpEngine e;
    if (!d || !init_engine(e, d)) { mi_fail(MI_ERR_SHAPE, "mi_mlpvae_workspace_bytes: unsupported sizes"); return -1; }
    return e.ws_total;
}

// weights_t: the K-contiguous kernel copies (storage type, n_flat elements); shadow: the bf16 copy of the flat buffer (bf16 engines; NULL for fp32).
// grads / adam_m / adam_v may be NULL for an inference-only engine (with_optimizer = 0).
void* mi_mlpvae_create(const MiMlpVaeDesc* d, float* params, float* grads, float* adam_m, float* adam_v, void* shadow, void* weights_t, void* workspace, long long workspace_bytes) {
    MlpEngine* e = new MlpEngine;
    if (!d || !init_engine(*e, d)) { delete e; mi_fail(MI_ERR_SHAPE, "mi_mlpvae_create: unsupported sizes"); return nullptr; }
    if (!params || !weights_t || !workspace || (d->dtype == MI_BF16 && !shadow) || (d->with_optimizer && (!grads || !adam_m || !adam_v))) {
        delete e; mi_fail(MI_ERR_ARG, "mi_mlpvae_create: missing buffers"); return nullptr;
    }
    if (workspace_bytes < e->ws_total || (((uintptr_t)workspace) & 255)) { delete e; mi_fail(MI_ERR_ARG, "mi_mlpvae_create: workspace too small or not 256-byte aligned"); return nullptr; }
    e->params = params; e->grads = grads; e->m = adam_m; e->v = adam_v; e->shadow = shadow; e->wt = weights_t; e->ws = (char*)workspace; e->ws_bytes = workspace_bytes;
    return e;
}

void mi_mlpvae_destroy(void* h) {
    MlpEngine* e = (MlpEngine*)h;
    if (!e) return;
    if (e->side_ok == 1) { hipStreamSynchronize(e->side); hipEventDestroy(e->ev_fork); hipEventDestroy(e->ev_chain); hipEventDestroy(e->ev_done); hipStreamDestroy(e->side); }
    delete e;
}

// refresh the derived weight copies after the master weights were written from outside (load / broadcast)
int mi_mlpvae_sync_shadow(void* h, void* stream) {
    MlpEngine* e = (MlpEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "mlp vae engine: null handle");
    if (e->d.dtype == MI_BF16) CK(mi_cast_f32_to_bf16(stream, e->params, e->shadow, e->total));
    long long off[16]; int K[16], N[16], n = 0;
    kernel_table(e, off, K, N, &n);
    return mi_transpose_weights(stream, e->d.dtype, e->params, e->wt, off, K, N, n);
}

// device pointers into the workspace (fp32 unless noted): 0 losses[2] (recon, kl), 1 mean [B,Z], 2 logvar [B,Z], 3 kl_row [B], 4 logits [B,P] (storage type), 5 z [B,Z] (storage type)
void* mi_mlpvae_buffer(void* h, int which) {
    MlpEngine* e = (MlpEngine*)h;
    if (!e) return nullptr;
    switch (which) {
        case 0: return e->at(e->o_out2);
        case 1: return e->at(e->o_mean);
        case 2: return e->at(e->o_logvar);
        case 3: return e->at(e->o_klrow);
        case 4: return e->at(e->o_d[e->nd - 1]);
        case 5: return e->at(e->o_z);
        default: return nullptr;
    }
}

// forward + ELBO terms of one minibatch (vae/models.py:226-229 / the forward half of :213-216).  src / tgt: float32 frame tables [n_frames, S] / [n_frames, P] on the
// device; idx: int32 [B] rows (NULL: the first B); inv_batch = 1 / B_global; eps [B, Z] (sample != 0); want_grad: also leave dlogits for mi_mlpvae_backward.
int mi_mlpvae_forward(void* h, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps, int sample, int want_grad,
                      float* metrics3, float metric_weight) {
    MlpEngine* e = (MlpEngine*)h;
    CK(check_batch(e, B));
    if (!tgt) return mi_fail(MI_ERR_ARG, "mi_mlpvae_forward: missing target table");
    const MiMlpVaeDesc& d = e->d;
    const void* x = nullptr;
    CK(stage_input(e, stream, src, frames_u8 & 1, idx, B, &x));
    CK(run_encoder(e, stream, x, B));
    CK(reparam(e, stream, B, eps, sample));
    CK(run_decoder(e, stream, e->at(e->o_z), B));
    const float kl_floor = d.kl_tolerance > 0.f ? d.kl_tolerance * d.z_dim : 0.f;
    const bool grad = want_grad && d.with_optimizer;
    if (frames_u8 & 2)
        CK(mi_bce_logits_fwd_bwd_u8(stream, d.dtype, e->at(e->o_d[e->nd - 1]), (const unsigned char*)tgt, idx, d.target_size, B, d.target_size, d.loss_kind, inv_batch,
                                    grad ? e->at(e->o_gd[e->nd - 1]) : nullptr, (float*)e->at(e->o_partial)));
    else
        CK(mi_bce_logits_fwd_bwd(stream, d.dtype, e->at(e->o_d[e->nd - 1]), (const float*)tgt, idx, d.target_size, B, d.target_size, d.loss_kind, inv_batch,
                                 grad ? e->at(e->o_gd[e->nd - 1]) : nullptr, (float*)e->at(e->o_partial)));
    CK(mi_vae_finalize_losses(stream, (const float*)e->at(e->o_partial), e->nchunks, (const float*)e->at(e->o_klrow), kl_floor, B, inv_batch, (float*)e->at(e->o_out2),
                              metrics3, metric_weight));
    e->last_B = B; e->last_x = x;
    return MI_OK;
}

// gradients of the last forward(want_grad = 1) into the gradient buffer (STORED, not accumulated).  part 0 = everything, 1 = decoder half (+ dz), 2 = encoder half:
// the data-parallel host all-reduces grads[decoder_offset:] in between (mi_mlpvae_decoder_offset).  eps: the noise of that forward pass.
int mi_mlpvae_backward(void* h, void* stream, const float* eps, float inv_batch, int part) {
    MlpEngine* e = (MlpEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "mlp vae engine: null handle");
    if (!e->grads) return mi_fail(MI_ERR_STATE, "mi_mlpvae_backward: engine created without optimiser buffers");
    if (e->last_B < 1) return mi_fail(MI_ERR_STATE, "mi_mlpvae_backward: no forward pass recorded");
    if (part < 0 || part > 2) return mi_fail(MI_ERR_ARG, "mi_mlpvae_backward: part must be 0, 1 or 2");
    const MiMlpVaeDesc& d = e->d;
    const int B = e->last_B, Z = d.z_dim, dt = d.dtype;
    // the ordered slab sums of the layers whose filter gradient is split over rows (the small ones) are recorded and issued as ONE launch at the end of the part
    // (eight 4.5 us launches per step otherwise); every layer owns a piece of the scratch until then
    struct SrGuard { int prev; SrGuard() : prev(mi_small_reduce_defer(1)) {} ~SrGuard() { mi_small_reduce_defer(prev); } } sr_guard;
    // A full pass (part 0) runs on two streams: the input-gradient chain is a row of dependent kernels, eight of them a few microseconds of work on an empty GPU, and no
    // filter gradient is on it.  The output layer's filter gradient (needs only the loss gradient) runs UNDER the chain on the second stream; behind the chain the small
    // layers' filter gradients run there next to the first layer's on the caller's stream.  Two events on the caller's stream, one join.  MI355_MLP_STREAMS=0: one stream.
    static int streams_on = -1;
    if (streams_on < 0) { const char* ev = getenv("MI355_MLP_STREAMS"); streams_on = (ev && ev[0] == '0') ? 0 : 1; }
    if (part == 0 && streams_on && e->side_ok == 0) {
        if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&e->ev_chain, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming) == hipSuccess) e->side_ok = 1;
        else e->side_ok = -1;
    }
    const bool fork = part == 0 && streams_on && e->side_ok == 1;
    hipStream_t sm = (hipStream_t)stream;
    long long scr_used = 0;
    auto wgrad_on = [&](void* s_, const Dense& l, const void* a, const void* gy) {
        const long long need = al256(mi_gemm_wgrad_scratch_bytes(dt, 1 << 20, l.K, l.N));
        if (scr_used + need > e->scratch_bytes) return mi_fail(MI_ERR_STATE, "mi_mlpvae_backward: scratch exhausted");
        void* scr = e->at(e->o_scratch + scr_used);
        scr_used += need;
        return mi_gemm_wgrad_bias_set(s_, dt, a, gy, B, l.K, l.N, e->grads + l.wo, e->grads + l.bo, scr, need, 1);
    };
    struct Later { const Dense* l; const void* a; const void* gy; } later[2 * ML_MAX + 2];
    int nlater = 0;
    // where a layer's filter gradient goes: one stream -> right here; two streams -> the output layer's at once on the second stream, the others behind the chain
    auto wgrad = [&](const Dense& l, const void* a, const void* gy, bool output_layer) -> int {
        if (!fork) return wgrad_on(stream, l, a, gy);
        if (output_layer) return wgrad_on((void*)e->side, l, a, gy);
        later[nlater].l = &l; later[nlater].a = a; later[nlater].gy = gy; ++nlater;
        return MI_OK;
    };
    // every return path below joins the second stream back (an error return between fork and join must not leave work of this pass pending on the caller's buffers
    // with nothing ordered behind it: ADVICE r04)
    struct JoinGuard {
        MlpEngine* e; hipStream_t sm; bool armed;
        int join() { armed = false; return (hipEventRecord(e->ev_done, e->side) == hipSuccess && hipStreamWaitEvent(sm, e->ev_done, 0) == hipSuccess) ? MI_OK : mi_fail(MI_ERR_LAUNCH, "mi_mlpvae_backward: stream join failed"); }
        ~JoinGuard() { if (armed) (void)join(); }
    } join_guard{e, sm, false};
    if (fork) {      // (the loss gradient and every activation are complete on the caller's stream here)
        if (hipEventRecord(e->ev_fork, sm) != hipSuccess || hipStreamWaitEvent(e->side, e->ev_fork, 0) != hipSuccess) return mi_fail(MI_ERR_LAUNCH, "mi_mlpvae_backward: stream fork failed");
        join_guard.armed = true;
    }
    if (part == 0 || part == 1) {
        for (int i = e->nd - 1; i >= 0; --i) {
            const Dense& l = e->dec[i];
            const void* gy = e->at(e->o_gd[i]);
            const void* a = i > 0 ? e->at(e->o_d[i - 1]) : e->at(e->o_z);
            CK(wgrad(l, a, gy, i == e->nd - 1));
            // dx = gy W^T, ReluGrad mask = the layer's input (a ReLU output): W[k, n] read as [N_out = k][K_in = n]
            if (i > 0) CK(dense(e, stream, gy, B, l.N, e->w(l), l.K, nullptr, 0, e->at(e->o_d[i - 1]), e->at(e->o_gd[i - 1]), 0));
            else CK(dense(e, stream, gy, B, l.N, e->w(l), l.K, nullptr, 0, nullptr, e->at(e->o_dz), 1));
        }
    }
    if (part == 0 || part == 2) {
        const float kl_floor = d.kl_tolerance > 0.f ? d.kl_tolerance * Z : 0.f;
        CK(mi_vae_reparam_kl_bwd(stream, dt, (const float*)e->at(e->o_dz), 1, (const float*)e->at(e->o_mean), (const float*)e->at(e->o_logvar), eps, (const float*)e->at(e->o_klrow),
                                 d.beta, kl_floor, inv_batch, B, Z, e->at(e->o_dheads)));
        CK(wgrad(e->heads, e->at(e->o_h[e->ne - 1]), e->at(e->o_dheads), false));
        CK(dense(e, stream, e->at(e->o_dheads), B, e->heads.N, e->w(e->heads), e->heads.K, nullptr, 0, e->at(e->o_h[e->ne - 1]), e->at(e->o_gh[e->ne - 1]), 0));
        for (int i = e->ne - 1; i >= 0; --i) {
            const Dense& l = e->enc[i];
            const void* gy = e->at(e->o_gh[i]);
            const void* a = i > 0 ? (const void*)e->at(e->o_h[i - 1]) : e->last_x;
            CK(wgrad(l, a, gy, false));
            if (i > 0) CK(dense(e, stream, gy, B, l.N, e->w(l), l.K, nullptr, 0, e->at(e->o_h[i - 1]), e->at(e->o_gh[i - 1]), 0));
        }
    }
    if (fork) {
        // behind the chain: every gradient of an activation exists.  All but the LAST recorded layer (the first encoder layer: the big one) on the second stream, their
        // slab sums as one launch there; the first encoder layer's on the caller's stream next to them (its own slab sum, if it has one, by the flush below)
        if (hipEventRecord(e->ev_chain, sm) != hipSuccess || hipStreamWaitEvent(e->side, e->ev_chain, 0) != hipSuccess) return mi_fail(MI_ERR_LAUNCH, "mi_mlpvae_backward: stream hand-over failed");
        for (int j = 0; j + 1 < nlater; ++j) CK(wgrad_on((void*)e->side, *later[j].l, later[j].a, later[j].gy));
        CK(mi_small_reduce_flush((void*)e->side));
        if (nlater > 0) CK(wgrad_on(stream, *later[nlater - 1].l, later[nlater - 1].a, later[nlater - 1].gy));
        const int rc = mi_small_reduce_flush(stream);
        const int rcj = join_guard.join();
        return rc != MI_OK ? rc : rcj;
    }
    return mi_small_reduce_flush(stream);
}

long long mi_mlpvae_decoder_offset(void* h) {
    MlpEngine* e = (MlpEngine*)h;
    return e ? e->decoder_offset : -1;
}

// TF ApplyAdam on the whole flat buffer (vae/models.py:140-142); refreshes the bf16 copy and the K-contiguous kernel copies in the same launch
int mi_mlpvae_apply_adam(void* h, void* stream, float alpha, float beta1, float beta2, float epsilon) {
    MlpEngine* e = (MlpEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "mlp vae engine: null handle");
    if (!e->grads || !e->m || !e->v) return mi_fail(MI_ERR_STATE, "mi_mlpvae_apply_adam: engine created without optimiser buffers");
    long long off[16]; int K[16], N[16], n = 0;
    kernel_table(e, off, K, N, &n);
    int skip[16] = {0};
    skip[0] = 1;                                          // the first encoder layer has no input gradient: nobody reads its [K, N] storage-type copy (39 MB of writes)
    // INVARIANT: from the first optimiser step on, shadow[enc[0].wo ...] is STALE (it holds the weights of the last mi_mlpvae_sync_shadow); the engine
    // reads layer 0 only through its K-contiguous copy wt (forward).  Anything that wants the storage-type weights of layer 0 in [K, N] order must call
    // mi_mlpvae_sync_shadow first.
    return mi_adam_tf_layouts(stream, e->d.dtype, e->params, e->m, e->v, e->grads, e->total, off, K, N, skip, n, alpha, nullptr, beta1, beta2, epsilon,
                              e->d.dtype == MI_BF16 ? e->shadow : nullptr, e->wt, 0);
}

// One whole SGD step (the reference's sess.run([train_step, ...]), vae/models.py:213-216) in ONE call; nothing synchronises the host
int mi_mlpvae_train_step(void* h, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps,
                         float alpha, float beta1, float beta2, float epsilon, float* metrics3, float metric_weight) {
    CK(mi_mlpvae_forward(h, stream, src, tgt, frames_u8, idx, B, inv_batch, eps, 1, 1, metrics3, metric_weight));
    CK(mi_mlpvae_backward(h, stream, eps, inv_batch, 0));
    return mi_mlpvae_apply_adam(h, stream, alpha, beta1, beta2, epsilon);
}

// One whole DATA-PARALLEL SGD step of the MlpVAE in ONE call (round 6): forward + ELBO of this rank's rows (inv_batch = 1 / B_global), the
// decoder half of the backward pass, its gradients' all-reduce queued on the communicator's own stream (it runs under the encoder half), the encoder half, its all-reduce,
// the join, TF-Adam -- what vae/models.py's host loop issued as two backward calls + two Python-side collectives.  Buckets (mi_mlpvae_dp_buckets): {part 1: floats
// [decoder_offset, total)} then {part 2: [0, decoder_offset)} -- the order the backward pass completes them.  comm: mi_comm_init (or a recording communicator).
int mi_mlpvae_dp_buckets(void* h, long long* out6) {
    MlpEngine* e = (MlpEngine*)h;
    if (!e || !out6) return mi_fail(MI_ERR_ARG, "mi_mlpvae_dp_buckets: null handle or output");
    const long long b[6] = {1, e->decoder_offset, e->total, 2, 0, e->decoder_offset};
    for (int i = 0; i < 6; ++i) out6[i] = b[i];
    return MI_OK;
}

int mi_mlpvae_train_step_dp(void* h, void* comm, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps,
                            float alpha, float beta1, float beta2, float epsilon, float* metrics3, float metric_weight) {
    MlpEngine* e = (MlpEngine*)h;
    CK(check_batch(e, B));
    if (!comm) return mi_fail(MI_ERR_ARG, "mi_mlpvae_train_step_dp: null communicator (single rank: mi_mlpvae_train_step)");
    if (!e->grads) return mi_fail(MI_ERR_STATE, "mi_mlpvae_train_step_dp: engine created without a gradient buffer");
    long long bk[6];
    CK(mi_mlpvae_dp_buckets(h, bk));
    CK(mi_mlpvae_forward(h, stream, src, tgt, frames_u8, idx, B, inv_batch, eps, 1, 1, metrics3, metric_weight));
    int rc = MI_OK;                                       // (a local failure still joins what was queued; it is fatal for the job all the same: mi_vae_train_step_dp)
    for (int i = 0; i < 2 && rc == MI_OK; ++i) {
        rc = mi_mlpvae_backward(h, stream, eps, inv_batch, (int)bk[3 * i]);
        if (rc == MI_OK) rc = mi_allreduce_sum_f32_async(comm, stream, e->grads + bk[3 * i + 1], bk[3 * i + 2] - bk[3 * i + 1]);
    }
    const int rcw = mi_comm_wait(comm, stream);
    if (rc != MI_OK) return rc;
    CK(rcw);
    return mi_mlpvae_apply_adam(h, stream, alpha, beta1, beta2, epsilon);
}

// VAE.encode (vae/models.py:199-202): frames -> mean [B, Z] fp32
int mi_mlpvae_encode(void* h, void* stream, const void* src, int frames_u8, const int* idx, int B, float* mean_out) {
    MlpEngine* e = (MlpEngine*)h;
    CK(check_batch(e, B));
    if (!mean_out) return mi_fail(MI_ERR_ARG, "mi_mlpvae_encode: missing output");
    const void* x = nullptr;
    CK(stage_input(e, stream, src, frames_u8 & 1, idx, B, &x));
    CK(run_encoder(e, stream, x, B));
    CK(reparam(e, stream, B, nullptr, 0));
    if (hipMemcpyAsync(mean_out, e->at(e->o_mean), (size_t)B * e->d.z_dim * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return mi_fail(MI_ERR_LAUNCH, "mi_mlpvae_encode: copy failed");
    return MI_OK;
}

// VAE.generate_from_latent (vae/models.py:204-205): z [B, Z] fp32 -> sigmoid(logits) [B, P] fp32
int mi_mlpvae_decode(void* h, void* stream, const float* z, int B, float* recon_out) {
    MlpEngine* e = (MlpEngine*)h;
    CK(check_batch(e, B));
    if (!z || !recon_out) return mi_fail(MI_ERR_ARG, "mi_mlpvae_decode: missing buffers");
    const void* zin = z;
    if (e->d.dtype == MI_BF16) { CK(mi_cast_f32_to_bf16(stream, z, e->at(e->o_z), (long long)B * e->d.z_dim)); zin = e->at(e->o_z); }
    CK(run_decoder(e, stream, zin, B));
    return mi_sigmoid(stream, e->d.dtype, e->at(e->o_d[e->nd - 1]), recon_out, (long long)B * e->d.target_size);
}

// VAE.reconstruct (vae/models.py:196-197)
int mi_mlpvae_reconstruct(void* h, void* stream, const void* src, int frames_u8, const int* idx, int B, const float* eps, int sample, float* recon_out) {
    MlpEngine* e = (MlpEngine*)h;
    CK(check_batch(e, B));
    if (!recon_out) return mi_fail(MI_ERR_ARG, "mi_mlpvae_reconstruct: missing output");
    const void* x = nullptr;
    CK(stage_input(e, stream, src, frames_u8 & 1, idx, B, &x));
    CK(run_encoder(e, stream, x, B));
    CK(reparam(e, stream, B, eps, sample));
    CK(run_decoder(e, stream, e->at(e->o_z), B));
    return mi_sigmoid(stream, e->d.dtype, e->at(e->o_d[e->nd - 1]), recon_out, (long long)B * e->d.target_size);
}

}  // extern "C"
