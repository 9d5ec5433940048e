// vae_engine.hip — ConvVAE training / inference step orchestration behind the C ABI (host C++ only; every
// FLOP is in the kernels of gemm_core.hpp / elementwise.hip).  One engine = one model replica on one GPU.
// The engine owns NO device memory: parameter / optimiser / workspace buffers are caller-provided (torch tensors).
//
// Mirrors the reference graph built in VAE.__init__ + ConvVAE (vae/models.py:85-142,249-266):
//   4x conv k4 s2 relu -> flatten(H,W,C) -> [mean | logstd_sq] dense -> z = mean + exp(.5 lv) eps
//   -> dense1 -> reshape(3,8,256) -> deconv k4,k4,k5 relu -> deconv4 k4 -> logits ; ELBO ; TF-Adam.
#include <stdlib.h>
#include <string.h>
#include "common.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"
#include "ppo_fused.hpp"

namespace {

constexpr int NCONV = 4;
constexpr int ENC_F[NCONV] = {32, 64, 128, 256};
constexpr int DEC_F[3] = {128, 64, 32};
constexpr int DEC_K[4] = {4, 4, 5, 4};
constexpr int N_TENSORS = 20;   // conv1-4 (k,b) heads (k,b) dense1 (k,b) deconv1-4 (k,b)

inline long long pad8(long long n) { return (n + 7) / 8 * 8; }

struct Layout {
    long long off[N_TENSORS], size[N_TENSORS], total;
};

struct Geom {
    int ih[NCONV + 1], iw[NCONV + 1], c[NCONV + 1];      // encoder maps: index 0 = input, i = output of conv i
    int dh[5], dw[5], dc[5];                              // decoder maps: index 0 = dense1 reshape, i = output of deconv i
    int flat;
};

bool make_geom(const MiVaeDesc& d, Geom& g) {
    g.ih[0] = d.ih; g.iw[0] = d.iw; g.c[0] = d.cin;
    for (int i = 0; i < NCONV; ++i) {
        if (g.ih[i] < 4 || g.iw[i] < 4) return false;
        g.ih[i + 1] = (g.ih[i] - 4) / 2 + 1; g.iw[i + 1] = (g.iw[i] - 4) / 2 + 1; g.c[i + 1] = ENC_F[i];
    }
    g.flat = g.ih[4] * g.iw[4] * g.c[4];
    g.dh[0] = g.ih[4]; g.dw[0] = g.iw[4]; g.dc[0] = g.c[4];
    for (int i = 0; i < 4; ++i) {
        g.dh[i + 1] = (g.dh[i] - 1) * 2 + DEC_K[i]; g.dw[i + 1] = (g.dw[i] - 1) * 2 + DEC_K[i];
        g.dc[i + 1] = i < 3 ? DEC_F[i] : d.ct;
    }
    return true;
}

void make_layout(const MiVaeDesc& d, const Geom& g, Layout& L) {
    long long o = 0;
    int t = 0;
    auto add = [&](long long n) { L.off[t] = o; L.size[t] = n; o += pad8(n); ++t; };
    for (int i = 0; i < NCONV; ++i) { add(16LL * g.c[i] * g.c[i + 1]); add(g.c[i + 1]); }
    add((long long)g.flat * 2 * d.z_dim); add(2 * d.z_dim);
    add((long long)d.z_dim * g.flat); add(g.flat);
    for (int i = 0; i < 4; ++i) { add((long long)DEC_K[i] * DEC_K[i] * g.dc[i + 1] * g.dc[i]); add(g.dc[i + 1]); }
    L.total = o;
}

struct Workspace {
    // byte offsets into the caller's workspace
    long long act[NCONV + 1];          // act[i] = output of conv i (T)         (act[0] unused: frames stay in the dataset)
    long long gact[NCONV + 1];         // gradient wrt pre-activation of conv i output (T)
    long long dec[5];                  // dec[0] = dense1 output, dec[i] = output of deconv i (T); dec[4] = logits
    long long gdec[5];
    long long z, dheads;               // T
    long long heads_slab, dz_slab, mean, logvar, kl_row, partial, bpart, out2, zf32;   // fp32
    long long tail_slabs, tail_slab_bytes;   // per-block partial filter gradients of the fused decoder tail (dectail_tile.hpp)
    long long enc_slabs, enc_slab_bytes;     // per-block partial sums of the fused encoder-head backward kernel (enchead_tile.hpp)
    long long scratch, scratch_bytes;  // split-reduction slabs of the filter-gradient kernels on the filter-gradient stream (SCRATCH_REGIONS rotating regions)
    long long scratch_main, scratch_main_bytes;    // ... of the filter / bias gradients issued on the caller's stream (one region, reused in stream order)
    long long scratch_third, scratch_third_bytes;  // ... of the latent layers' gradients on the optional third stream
    long long scratch_tail, scratch_tail_bytes;    // ... of the small reductions at the end of a full backward pass that run as ONE deferred launch (bump-allocated: every job keeps its slabs until the flush)
    long long scratch_side, scratch_side_bytes;    // ... of the latent layers' gradients when they run on the filter-gradient stream (outside the rotating regions: those may hold deferred slabs)
    long long bits_act1, bits_dec3;    // ReLU bit words of conv1's / deconv3's output (bf16 engine: 8 bytes per pixel; read by conv2's / deconv4's input gradient)
    long long roll, roll_bytes;        // rollout step (B = 1): act1 | raw sums of conv2..4 and of the mean head (zeroed per step)
    long long eps_buf, rng;   // noise drawn by the engine [B,Z] fp32; generator state (4 x uint64)
    long long wfrag[11];       // fragment-ordered bf16 copies of conv4's / deconv1's kernels for the activation-resident kernels (ares_tile.hpp): conv4 forward, conv4 input
                              // gradient, deconv1 forward, deconv1 input gradient (1 MB each), conv3 input gradient, deconv2 forward (256 KB each); rewritten behind every optimiser step with the K-contiguous copies
    long long total;
    // debug (MI355_DEBUG_GUARDS=1 at mi_vae_workspace_bytes AND mi_vae_create time): 256 bytes of a known pattern behind every region; mi_vae_debug_check_guards
    // finds the region a kernel wrote past (the bounds-checking debug mode of the new build)
    int n_guards; long long guard_off[96];
};
constexpr uint32_t GUARD_WORD = 0xC0DEFA11u;

// ---- per-op HIP-event timing (bench.py's live roofline numbers; off by default) ----
enum {
    OP_CONV_FWD = 0, OP_HEADS_FWD = 4, OP_REPARAM_FWD = 5, OP_DENSE1_FWD = 6, OP_DECONV_FWD = 7, OP_RECON_LOSS = 11, OP_FINALIZE = 12,
    OP_DECONV_BIAS = 13, OP_DECONV_WGRAD = 17, OP_DECONV_DGRAD = 21, OP_DENSE1_BIAS = 25, OP_DENSE1_WGRAD = 26, OP_DENSE1_DGRAD = 27,
    OP_REPARAM_BWD = 28, OP_HEADS_BIAS = 29, OP_HEADS_WGRAD = 30, OP_HEADS_DGRAD = 31, OP_CONV_BIAS = 32, OP_CONV_WGRAD = 36,
    OP_CONV_DGRAD = 40, OP_ADAM = 44, N_OPS = 45
};
constexpr int SCRATCH_REGIONS = 6;  // deconv3..1 and conv4..2 keep their partial-sum slabs until the deferred reduce (conv_ops.hip)
const char* const OP_NAMES[N_OPS] = {
    "conv1.fwd", "conv2.fwd", "conv3.fwd", "conv4.fwd", "heads.fwd", "reparam_kl.fwd", "dense1.fwd",
    "deconv1.fwd", "deconv2.fwd", "deconv3.fwd", "deconv4.fwd", "recon_loss", "finalize_losses",
    "deconv1.bias_grad", "deconv2.bias_grad", "deconv3.bias_grad", "deconv4.bias_grad",
    "deconv1.wgrad", "deconv2.wgrad", "deconv3.wgrad", "deconv4.wgrad",
    "deconv1.dgrad", "deconv2.dgrad", "deconv3.dgrad", "deconv4.dgrad",
    "dense1.bias_grad", "dense1.wgrad", "dense1.dgrad", "reparam_kl.bwd", "heads.bias_grad", "heads.wgrad", "heads.dgrad",
    "conv1.bias_grad", "conv2.bias_grad", "conv3.bias_grad", "conv4.bias_grad",
    "conv1.wgrad", "conv2.wgrad", "conv3.wgrad", "conv4.wgrad",
    "conv1.dgrad", "conv2.dgrad", "conv3.dgrad", "conv4.dgrad", "adam"};

struct Timing {
    int mode;            // 0 off, 1 every op, 2 only op_filter
    int op_filter, cap, n;
    hipEvent_t* ev;      // 2 per record
    int* op;
};

struct VaeEngine {
    Timing tm;
    MiVaeDesc d;
    Geom g;
    Layout L;
    Workspace W;
    float *params, *grads, *m, *v;
    void* shadow;
    void* wt;                           // K-contiguous ("transposed") copies of the 10 kernels, type T, same offsets
    char* ws;
    int esz;                            // bytes per T
    int ns_heads, ns_dz, nchunks, partial_cap;
    int last_B;
    int b4_fused;                       // the last forward already accumulated deconv4's bias gradient
    int tail_nblk;                      // partial filter gradients of the fused decoder tail waiting in the workspace (reduced inside the backward pass)
    // finalize_losses of the last forward, deferred (mi_vae_train_step only): nothing in the backward pass reads the loss scalars, so the one-block
    // kernel runs where the caller's stream would otherwise wait for the filter-gradient stream instead of between forward and backward
    struct { int pending, nblk, B; float kl_floor, inv_batch, metric_weight; float* metrics3; float* dbias; } fin;
    int defer_fin;
    int tail_sched;                     // the current backward pass started behind a fused decoder tail (stream placement of the encoder's filter gradients)
    int tail_fused;                     // ... and (decoder tail in one launch, dectail_tile.hpp) deconv4's filter gradient and deconv3's output gradient
    hipStream_t side;                   // filter-gradient stream of the backward pass (created on first use; host object only)
    hipEvent_t ev_ready, ev_done;
    int side_ok;
    int dp_open_join, dp_side_ready;    // mi_vae_train_step_dp (round 6): a backward PART ends with the filter-gradient stream waiting for the caller's instead of the other way round; the bucket's
                                        // all-reduce is then chained to the filter-gradient stream and the caller's stream goes straight on with the next part's input-gradient chain
    int ares_mid;                       // ... and the mid-layer copies (conv3's input gradient, deconv2 forward)
    int ares_ok;                        // the fragment-ordered weight copies exist (bf16 engine, the model's geometry): the four small-grid layers run on the activation-resident kernels
    int fwd_produced;                   // the last forward's final kernel (the fused decoder tail) carries ev_ready on its own dispatch packet (MI355_KEVENT; consumed by the backward pass's first hand-over)
    hipStream_t main2;                  // MI355_CU_SPLIT=N (measurement aid, round 6): the caller-side half of the backward pass on an engine stream bound to compute units [0, N), the filter-gradient stream to [N, 256)
    hipEvent_t ev_in, ev_out;
    int main2_ok;
    hipStream_t third;                  // latent-layer gradients + loss finalisation of a full two-stream backward (small launches with early operands)
    hipEvent_t ev_lat, ev_third;
    int third_ok;
    int bits1_ok, bits3_ok;             // the last forward pass wrote the ReLU bit words of act1 / dec3
    int rng_ready;                      // generator state in the workspace has been initialised (mi_vae_set_seed)
    const float* last_eps;              // the noise the last sampling forward used (caller's buffer or the engine's own draw)
    int last_u8;                        // frame-table format of the last forward (backward reads the same source table)
    // weights as the MFMA kernels read them: the fp32 masters (MI_F32) or the shadow copy Adam keeps in the engine's storage type (bf16 / split)
    const void* wptr(int t) const { return d.dtype == MI_F32 ? (const void*)(params + L.off[t]) : (const void*)((const char*)shadow + L.off[t] * esz); }
    const void* wtptr(int t) const { return (const void*)((const char*)wt + L.off[t] * esz); }
    const float* bptr(int t) const { return params + L.off[t]; }
    float* gptr(int t) const { return grads + L.off[t]; }
    void* at(long long off) const { return ws + off; }
};

int pick_split(int M, int N, int K, int bk) {
    const int tiles = ((M + 127) / 128) * (N <= 64 ? 1 : (N + 127) / 128);
    int ns = 256 / (tiles > 0 ? tiles : 1);
    if (ns < 1) ns = 1;
    // MI355_LATENT_SPLIT=<n>: at most n K slices in the latent layers' split-K sums.  16 since late round 5 (32 before): half the slab traffic between the tall-K kernels and the
    // reparameterisation kernels that sum them, one block per CU instead of two; step 0.8225 -> 0.8179 / 0.8153 -> 0.8125 ms on two boxes (12: the same, 8 and 24: slower)
    static int ns_cap = -1;
    if (ns_cap < 0) { const char* ev = getenv("MI355_LATENT_SPLIT"); ns_cap = ev ? atoi(ev) : 16; if (ns_cap < 1 || ns_cap > 32) ns_cap = 16; }
    if (ns > ns_cap) ns = ns_cap;
    while (ns > 1) {                    // every slab must own at least one K block
        int len = (K + ns - 1) / ns; len = (len + bk - 1) / bk * bk;
        if ((long long)len * (ns - 1) < K) break;
        --ns;
    }
    return ns;
}

void make_workspace(VaeEngine& e) {
    const MiVaeDesc& d = e.d; const Geom& g = e.g;
    const long long B = d.max_batch;
    long long o = 0;
    Workspace& W = e.W;
    const char* ge = getenv("MI355_DEBUG_GUARDS");
    const bool guards = ge && ge[0] == '1';
    W.n_guards = 0;
    auto add = [&](long long bytes) {
        long long r = o; o += (bytes + 255) / 256 * 256;
        if (guards && W.n_guards < 96) { W.guard_off[W.n_guards++] = o; o += 256; }
        return r;
    };
    W.act[0] = W.gact[0] = 0;
    for (int i = 1; i <= NCONV; ++i) {
        const long long n = B * g.ih[i] * g.iw[i] * g.c[i];
        W.act[i] = add(n * e.esz); W.gact[i] = add(n * e.esz);
    }
    for (int i = 0; i <= 4; ++i) {
        const long long n = B * g.dh[i] * g.dw[i] * g.dc[i];
        W.dec[i] = add(n * e.esz); W.gdec[i] = add(n * e.esz);
    }
    W.z = add(B * d.z_dim * e.esz); W.dheads = add(B * 2 * d.z_dim * e.esz);
    W.heads_slab = add((long long)e.ns_heads * B * 2 * d.z_dim * 4);
    W.dz_slab = add((long long)e.ns_dz * B * d.z_dim * 4);
    W.mean = add(B * d.z_dim * 4); W.logvar = add(B * d.z_dim * 4); W.kl_row = add(B * 4);
    e.partial_cap = (int)(B * 64 > B * e.nchunks ? B * 64 : B * e.nchunks);   // loss partial sums: per (frame, chunk) or per block of the fused decoder tail
    W.partial = add((long long)e.partial_cap * 4); W.bpart = add((long long)e.partial_cap * 16); W.out2 = add(256); W.zf32 = add(B * d.z_dim * 4);
    // 256 position splits x the largest per-split slab (deconv3: 25 taps x 64 x 32 floats), rounded up
    // one region per raw-staged filter gradient of a backward pass (split storage: + the unfolded dW').  Round 4: the fp32 engine has them too -- every
    // filter / bias gradient of every engine reduces its position splits through slabs in a fixed order (two runs of a step are bitwise equal)
    // An inference-only engine (MiVaeDesc::inference_only: VAE(training=False) -- rollout, evaluation, encode) never runs a backward pass: no slab scratch at all
    // (every B = 1 rollout engine paid ~0.5 GB of HBM for it).  The slab sizes do not depend on the batch: 256 position splits x the largest per-split slab.
    const bool train = d.inference_only == 0;
    W.scratch_bytes = train ? SCRATCH_REGIONS * (64ll << 20) : 0;
    W.scratch = add(train ? W.scratch_bytes : 256);
    W.scratch_main_bytes = train ? 64ll << 20 : 0; W.scratch_main = add(train ? W.scratch_main_bytes : 256);
    W.scratch_third_bytes = train ? 16ll << 20 : 0; W.scratch_third = add(train ? W.scratch_third_bytes : 256);
    W.scratch_side_bytes = train ? 16ll << 20 : 0; W.scratch_side = add(train ? W.scratch_side_bytes : 256);
    W.scratch_tail_bytes = train ? 32ll << 20 : 0; W.scratch_tail = add(train ? W.scratch_tail_bytes : 256);
    W.tail_slab_bytes = (train && d.dtype == MI_BF16) ? 2048ll * 6144 : 0;      // up to 8 resident blocks per CU x 6 KB
    W.tail_slabs = add(W.tail_slab_bytes > 0 ? W.tail_slab_bytes : 256);
    W.enc_slab_bytes = (train && d.dtype == MI_BF16) ? 2048ll * 8320 : 0;       // up to 8 resident blocks per CU x (64 x 32 + 32) floats
    W.enc_slabs = add(W.enc_slab_bytes > 0 ? W.enc_slab_bytes : 256);
    W.bits_act1 = add(B * g.ih[1] * g.iw[1] * (g.c[1] / 16) * 4); W.bits_dec3 = add(B * g.dh[3] * g.dw[3] * (g.dc[3] / 16) * 4);
    {
        long long n = 0;
        for (int i = 1; i <= NCONV; ++i) n += (long long)g.ih[i] * g.iw[i] * g.c[i];
        W.roll_bytes = (n + d.z_dim + 64) * 4;
        W.roll = add(W.roll_bytes);
    }
    W.eps_buf = add(B * d.z_dim * 4); W.rng = add(256);
    for (int i = 0; i < 11; ++i) W.wfrag[i] = add(mi_ares_weight_bytes());     // (round 6 -- 6, 7: the conv-form copies of conv3 / deconv2 for the register-weight kernel, 256 KB used; 8: deconv3 for its gather form, 144 KB; 9: conv2 for the fused encoder head, 64 KB; 10: deconv3 for its input gradient, conv form k = 5, 100 KB)
    W.total = o;
}

bool init_engine(VaeEngine& e, const MiVaeDesc* desc) {
    e.d = *desc;
    if (!make_geom(e.d, e.g)) return false;
    make_layout(e.d, e.g, e.L);
    e.esz = e.d.dtype == MI_BF16 ? 2 : 4;                  // MI_F32 and MI_BF16X3 (split storage) are 4-byte elements
    const int bk = e.d.dtype == MI_BF16 ? 32 : 16;
    e.ns_heads = pick_split(e.d.max_batch, 2 * e.d.z_dim, e.g.flat, bk);
    e.ns_dz = pick_split(e.d.max_batch, e.d.z_dim, e.g.flat, bk);
    e.nchunks = mi_recon_loss_chunks(e.g.dh[4] * e.g.dw[4] * e.g.dc[4]);
    make_workspace(e);
    return true;
}

#define CK(call) do { int rc__ = (call); if (rc__ != MI_OK) return rc__; } while (0)

inline bool tm_on(const VaeEngine* e, int id) {
    return e->tm.mode && (e->tm.mode == 1 || e->tm.op_filter == id) && e->tm.n < e->tm.cap;
}
// timed op: HIP events recorded on the SAME stream the kernel is launched on
#define TOP(e, st, id, call) do { \
        const bool t__ = tm_on(e, id); \
        if (t__) hipEventRecord(e->tm.ev[2 * e->tm.n], (hipStream_t)(st)); \
        int rc__ = (call); \
        if (t__) { hipEventRecord(e->tm.ev[2 * e->tm.n + 1], (hipStream_t)(st)); e->tm.op[e->tm.n] = (id); ++e->tm.n; } \
        if (rc__ != MI_OK) return rc__; } while (0)

int check_batch(const VaeEngine* e, int B) {
    if (!e) return mi_fail(MI_ERR_STATE, "vae engine: null handle");
    if (B < 1 || B > e->d.max_batch) return mi_fail(MI_ERR_ARG, "vae engine: batch outside [1, max_batch]");
    return MI_OK;
}

static int cu_split_env() {                            // MI355_CU_SPLIT=N, 8 <= N <= 248: compute units of the caller-side backward queue (0 / unset: no masks)
    static int n = -1;
    if (n < 0) { const char* ev = getenv("MI355_CU_SPLIT"); n = ev ? atoi(ev) : 0; if (n < 8 || n > 248) n = 0; }
    return n;
}

unsigned ready_event_flags() {                         // MI355_KEVENT=2: the hand-over event with timing enabled (A/B: what hipExtLaunchKernelGGL's stop event wants)
    const char* ev = getenv("MI355_KEVENT");
    unsigned f = (ev && atoi(ev) == 2) ? hipEventDefault : hipEventDisableTiming;
    // the hand-over events are consumed on this device only: MI355_EVENT_SCOPE=1 device-scope release, =2 no system-scope fence (A/B; default 0 = the runtime's default)
    const char* sc = getenv("MI355_EVENT_SCOPE");
    if (sc && atoi(sc) == 1) f |= hipEventReleaseToDevice;
    if (sc && atoi(sc) == 2) f |= hipEventDisableSystemFence;
    return f;
}

bool rc_wfrag_enabled() {                              // MI355_RC_WFRAG=0: the conv-form register-weight kernels of the mid layers read the K-contiguous weight copy (A/B runs)
    static int on = -1;
    if (on < 0) { const char* ev = getenv("MI355_RC_WFRAG"); on = (ev && ev[0] == '0') ? 0 : 1; }
    return on != 0;
}

bool rc_wfrag6_enabled() {                             // MI355_RC_WFRAG6=0: deconv3's input gradient alone back on the K-contiguous copy (A/B runs)
    static int on = -1;
    if (on < 0) { const char* ev = getenv("MI355_RC_WFRAG6"); on = (ev && ev[0] == '0') ? 0 : 1; }
    return on != 0;
}
// conv2 is 32 -> 64 channels k = 4 and deconv3 64 -> 32 channels k = 5 (the reference's geometry): their fragment-ordered copies (pack forms 5 / 4, 6) exist next to the activation-resident ones
bool rc_small_frag_ok(const VaeEngine* e) {
    const Geom& g = e->g;
    return g.c[1] == 32 && g.c[2] == 64 && g.dc[2] == 64 && g.dc[3] == 32 && DEC_K[2] == 5;
}

bool relu_bits_enabled() {                             // MI355_RELU_BITS=0: the input gradients read the activation tensors as ReluGrad masks (A/B runs)
    static int on = -1;
    if (on < 0) { const char* ev = getenv("MI355_RELU_BITS"); on = (ev && ev[0] == '0') ? 0 : 1; }
    return on != 0;
}

// encoder: frames (fp32, optional gather) -> act[1..4] -> heads slabs -> mean/logvar/z/kl
int run_encoder(VaeEngine* e, void* st, const void* frames, int frames_u8, const int* idx, int B, const float* eps, int sample, int want_bits = 0) {
    const MiVaeDesc& d = e->d; const Geom& g = e->g;
    if (frames_u8 && d.dtype != MI_BF16) return mi_fail(MI_ERR_ARG, "vae engine: uint8 frame tables are read by the bf16 engine only (fp32 mode takes float frames)");
    e->bits1_ok = 0;
    for (int i = 0; i < NCONV; ++i) {
        const void* x = i == 0 ? frames : e->at(e->W.act[i]);
        if (i == 0 && d.dtype == MI_BF16 && g.c[0] == 3 && g.c[1] == 32 && g.c[2] == 64 && e->tm.mode != 1) {      // (the fused kernel reads 3-channel frames: frame stride FH FW 3, w1 as [32][48])
            // round 5: conv1 + conv2 as ONE launch (enc12_tile.hpp): conv1's activation stays in LDS for conv2 (it is still written for the backward pass, never read back here).
            // (per-op timing keeps the two layer launches: they are what the profile names)
            const bool bits12 = want_bits && relu_bits_enabled();
            int launched = 0;
            if (e->ares_ok && rc_small_frag_ok(e) && rc_wfrag_enabled()) mi_tl_rc_wfrag = e->at(e->W.wfrag[9]);      // conv2's kernel in fragment order (consumed by the launch below)
            struct WfragGuard0 { ~WfragGuard0() { mi_tl_rc_wfrag = nullptr; } } wfrag_guard0;
            TOP(e, st, OP_CONV_FWD + 1, mi_conv2d_enc12_fwd(st, d.dtype, frames, frames_u8 ? 2 : 1, idx, B, g.ih[0], g.iw[0], e->wtptr(0), e->bptr(1), e->wtptr(2), e->bptr(3),
                                                         e->at(e->W.act[1]), bits12 ? e->at(e->W.bits_act1) : nullptr, e->at(e->W.act[2]), &launched));
            if (launched) { if (bits12) e->bits1_ok = 1; i = 1; continue; }
        }
        // conv1 (training pass, bf16): also writes the ReLU bit words conv2's input gradient reads instead of the 101 MB activation tensor
        const bool bits = i == 0 && want_bits && relu_bits_enabled() && d.dtype == MI_BF16 && g.c[1] == 32;
        if (i == 3 && e->ares_ok) {                           // conv4: activation-resident kernel (frames of the group in LDS, fragment-ordered weights streamed)
            int launched = 0;
            TOP(e, st, OP_CONV_FWD + i, mi_ares_conv(st, d.dtype, 0, x, B, e->at(e->W.wfrag[0]), e->bptr(7), 1, nullptr, e->at(e->W.act[4]), &launched));
            if (launched) continue;
        }
        // conv3 (64 -> 128 channels, k = 4): the register-weight kernel loads its weights from their fragment-ordered copy (round 6; MI355_RC_WFRAG=0: from the K-contiguous one)
        if (i == 2 && e->ares_ok && e->ares_mid && rc_wfrag_enabled()) mi_tl_rc_wfrag = e->at(e->W.wfrag[6]);
        struct WfragGuard { ~WfragGuard() { mi_tl_rc_wfrag = nullptr; } } wfrag_guard;
        TOP(e, st, OP_CONV_FWD + i, mi_conv2d_nhwc_fwd_bits(st, d.dtype, x, i == 0 ? idx : nullptr, i == 0 ? (frames_u8 ? 2 : 1) : 0, B, g.ih[i], g.iw[i], g.c[i],
                              e->wtptr(2 * i), 1, e->bptr(2 * i + 1), 4, 4, g.c[i + 1], 1, e->at(e->W.act[i + 1]), bits ? e->at(e->W.bits_act1) : nullptr, bits ? &e->bits1_ok : nullptr));
    }
    TOP(e, st, OP_HEADS_FWD, mi_gemm_bias_act(st, d.dtype, e->at(e->W.act[4]), B, g.flat, e->wtptr(8), 1, 2 * d.z_dim, nullptr, 0, nullptr,
                        e->at(e->W.heads_slab), 1, e->ns_heads));
    // split-K slabs are laid out [ns][B][2Z] with the CURRENT batch as the middle dimension
    if (sample && !eps && !e->rng_ready)
        return mi_fail(MI_ERR_STATE, "vae engine: sampling without injected noise needs mi_vae_set_seed() first");
    TOP(e, st, OP_REPARAM_FWD, mi_vae_reparam_kl_fwd_rng(st, d.dtype, (const float*)e->at(e->W.heads_slab), e->ns_heads, e->bptr(9), e->bptr(9) + d.z_dim, eps, sample,
                             B, d.z_dim, (float*)e->at(e->W.mean), (float*)e->at(e->W.logvar), e->at(e->W.z), (float*)e->at(e->W.kl_row),
                             (unsigned long long*)e->at(e->W.rng), (float*)e->at(e->W.eps_buf)));
    if (sample) e->last_eps = eps ? eps : (const float*)e->at(e->W.eps_buf);
    return MI_OK;
}

// decoder: z (T) -> dense1 -> deconv1..4 -> logits
int run_decoder(VaeEngine* e, void* st, int B, int last = 4, int want_bits = 0) {
    const MiVaeDesc& d = e->d; const Geom& g = e->g;
    e->bits3_ok = 0;
    TOP(e, st, OP_DENSE1_FWD, mi_gemm_bias_act(st, d.dtype, e->at(e->W.z), B, d.z_dim, e->wtptr(10), 1, g.flat, e->bptr(11), 0, nullptr, e->at(e->W.dec[0]), 0, 1));
    for (int i = 0; i < last; ++i) {
        const bool bits = i == 2 && want_bits && relu_bits_enabled() && d.dtype == MI_BF16 && g.dc[3] == 32;   // deconv3: ReLU bit words for deconv4's input gradient
        if (i == 0 && e->ares_ok) {
            int launched = 0;
            TOP(e, st, OP_DECONV_FWD + i, mi_ares_conv(st, d.dtype, 1, e->at(e->W.dec[0]), B, e->at(e->W.wfrag[2]), e->bptr(13), 1, nullptr, e->at(e->W.dec[1]), &launched));
            if (launched) continue;
        }
        if (i == 1 && e->ares_ok && e->ares_mid) {
            int launched = 0;
            TOP(e, st, OP_DECONV_FWD + i, mi_ares_conv(st, d.dtype, 2, e->at(e->W.dec[1]), B, e->at(e->W.wfrag[5]), e->bptr(15), 1, nullptr, e->at(e->W.dec[2]), &launched));
            if (launched) continue;
        }
        if (i == 2 && e->ares_ok && rc_small_frag_ok(e) && rc_wfrag_enabled()) mi_tl_rc_wfrag = e->at(e->W.wfrag[8]);      // deconv3: its kernel in the gather form's fragment order (round 6)
        struct WfragGuard3 { ~WfragGuard3() { mi_tl_rc_wfrag = nullptr; } } wfrag_guard3;
        TOP(e, st, OP_DECONV_FWD + i, mi_deconv2d_nhwc_fwd_bits(st, d.dtype, e->at(e->W.dec[i]), B, g.dh[i], g.dw[i], g.dc[i], e->wptr(12 + 2 * i), e->bptr(13 + 2 * i),
                                DEC_K[i], DEC_K[i], g.dc[i + 1], i < 3 ? 1 : 0, e->at(e->W.dec[i + 1]), bits ? e->at(e->W.bits_dec3) : nullptr, bits ? &e->bits3_ok : nullptr));
    }
    return MI_OK;
}

// K-contiguous copies of the 10 kernels (conv fwd, deconv dgrad, heads/dense1 fwd read them as the MFMA B operand)
// the 10 kernels as [K, N] matrices inside the flat buffer (ascending offsets)
int kernel_table(const VaeEngine* e, long long* off, int* K, int* N) {
    const Geom& g = e->g; const MiVaeDesc& d = e->d;
    int n = 0;
    for (int i = 0; i < NCONV; ++i) { off[n] = e->L.off[2 * i]; K[n] = 16 * g.c[i]; N[n] = g.c[i + 1]; ++n; }
    off[n] = e->L.off[8]; K[n] = g.flat; N[n] = 2 * d.z_dim; ++n;
    off[n] = e->L.off[10]; K[n] = d.z_dim; N[n] = g.flat; ++n;
    for (int i = 0; i < 4; ++i) { off[n] = e->L.off[12 + 2 * i]; K[n] = DEC_K[i] * DEC_K[i] * g.dc[i + 1]; N[n] = g.dc[i]; ++n; }
    return n;
}

// the activation-resident kernels take this model (bf16 storage, the reference's geometry); *mid: also the mid-layer gather form (conv3's input gradient, deconv2 forward)
bool ares_eligible(const VaeEngine* e, bool* mid) {
    const Geom& g = e->g; const MiVaeDesc& d = e->d;
    static int ares_on = -1;
    if (ares_on < 0) { const char* ev = getenv("MI355_ARES"); ares_on = (ev && ev[0] == '0') ? 0 : 1; }
    const bool ok = ares_on && d.dtype == MI_BF16 && g.ih[3] == 8 && g.iw[3] == 18 && g.c[3] == 128 && g.c[4] == 256 && g.dh[0] == 3 && g.dw[0] == 8 && g.dc[0] == 256 && g.dc[1] == 128 && DEC_K[0] == 4;
    if (mid) *mid = ok && g.c[2] == 64 && g.c[3] == 128 && g.dc[1] == 128 && g.dc[2] == 64 && DEC_K[1] == 4;
    return ok;
}

// have_wt: the K-contiguous copies were already written (by the optimiser launch itself, mi_adam_tf_layouts); have_frag: ... and the fragment-ordered copies as well (round 5)
int refresh_transposed(VaeEngine* e, void* st, bool have_wt = false, bool have_frag = false) {
    const MiVaeDesc& d = e->d;
    long long off[10]; int K[10], N[10];
    const int n = kernel_table(e, off, K, N);
    if (!have_wt) CK(mi_transpose_weights(st, d.dtype, e->params, e->wt, off, K, N, n));
    e->ares_ok = 0; e->ares_mid = 0;
    bool mid = false;
    if (ares_eligible(e, &mid)) {
        // conv4's kernel: HWIO [4][4][128][256]; deconv1's kernel: [kh][kw][out = 128][in = 256] -- the same [16][128][256] shape, read either way (ares.hip)
        // wfrag: 0 conv4 forward, 1 conv4 input gradient, 2 deconv1 forward, 3 deconv1 input gradient -- one launch
        if (!have_frag)
            CK(mi_ares_pack_weights8(st, e->params + e->L.off[6], e->params + e->L.off[12], mid ? e->params + e->L.off[4] : nullptr, mid ? e->params + e->L.off[14] : nullptr,
                                     e->at(e->W.wfrag[0]), e->at(e->W.wfrag[1]), e->at(e->W.wfrag[2]), e->at(e->W.wfrag[3]), mid ? e->at(e->W.wfrag[4]) : nullptr, mid ? e->at(e->W.wfrag[5]) : nullptr,
                                     mid ? e->at(e->W.wfrag[6]) : nullptr, mid ? e->at(e->W.wfrag[7]) : nullptr));
        if (!have_frag && rc_small_frag_ok(e)) {           // (round 6) deconv3's kernel for the gather-form register-weight kernel, conv2's for the fused encoder head
            CK(mi_ares_pack_weights(st, 4, e->params + e->L.off[16], e->at(e->W.wfrag[8])));
            CK(mi_ares_pack_weights(st, 5, e->params + e->L.off[2], e->at(e->W.wfrag[9])));
            CK(mi_ares_pack_weights(st, 6, e->params + e->L.off[16], e->at(e->W.wfrag[10])));     // deconv3's input gradient: conv form, k = 5
        }
        e->ares_mid = mid ? 1 : 0;
        e->ares_ok = 1;
    }
    return MI_OK;
}

}  // namespace

extern "C" {

int mi_vae_desc_size(void) { return (int)sizeof(MiVaeDesc); }
int mi_vae_tensor_count(void) { return N_TENSORS; }

long long mi_vae_param_floats(const MiVaeDesc* d) {
    VaeEngine e;
    if (!d || !init_engine(e, d)) { mi_fail(MI_ERR_SHAPE, "mi_vae_param_floats: unsupported geometry"); return -1; }
    return e.L.total;
}

// offsets/sizes (in floats) of the 20 device tensors inside the flat buffer, device order:
// conv1..4 {kernel,bias}, heads {kernel [flat,2Z] = [mean|logstd_sqare], bias [2Z]}, dense1 {kernel,bias}, deconv1..4 {kernel,bias}
int mi_vae_param_layout(const MiVaeDesc* d, long long* offsets, long long* sizes, int n) {
    VaeEngine e;
    if (!d || !init_engine(e, d)) return mi_fail(MI_ERR_SHAPE, "mi_vae_param_layout: unsupported geometry");
    if (n != N_TENSORS) return mi_fail(MI_ERR_ARG, "mi_vae_param_layout: expected 20 entries");
    for (int i = 0; i < N_TENSORS; ++i) { offsets[i] = e.L.off[i]; sizes[i] = e.L.size[i]; }
    return MI_OK;
}

long long mi_vae_workspace_bytes(const MiVaeDesc* d) {
    VaeEngine e;
    if (!d || !init_engine(e, d)) { mi_fail(MI_ERR_SHAPE, "mi_vae_workspace_bytes: unsupported geometry"); return -1; }
    return e.W.total;
}

void* mi_vae_create(const MiVaeDesc* d, float* params, float* grads, float* adam_m, float* adam_v, void* bf16_shadow,
                    void* weights_t, void* workspace, long long workspace_bytes) {
    VaeEngine* e = (VaeEngine*)calloc(1, sizeof(VaeEngine));
    if (!e) { mi_fail(MI_ERR_STATE, "mi_vae_create: out of host memory"); return nullptr; }
    if (!d || !init_engine(*e, d)) { free(e); mi_fail(MI_ERR_SHAPE, "mi_vae_create: unsupported geometry"); return nullptr; }
    if (d->dtype != MI_F32 && d->dtype != MI_BF16 && d->dtype != MI_BF16X3) { free(e); mi_fail(MI_ERR_ARG, "mi_vae_create: dtype must be 0 (f32), 1 (bf16) or 2 (split storage, bf16x3)"); return nullptr; }
    if (d->dtype != MI_F32 && !bf16_shadow) { free(e); mi_fail(MI_ERR_ARG, "mi_vae_create: bf16 / split mode needs the shadow weight buffer (2 / 4 bytes per parameter)"); return nullptr; }
    if (d->inference_only && grads) { free(e); mi_fail(MI_ERR_ARG, "mi_vae_create: an inference_only engine takes no gradient buffer (its workspace has no gradient scratch)"); return nullptr; }
    if (!params || !weights_t || !workspace || workspace_bytes < e->W.total) { free(e); mi_fail(MI_ERR_ARG, "mi_vae_create: missing buffers or workspace too small"); return nullptr; }
    if ((((uintptr_t)params) | ((uintptr_t)workspace) | ((uintptr_t)bf16_shadow) | ((uintptr_t)grads)) & 255) { free(e); mi_fail(MI_ERR_ARG, "mi_vae_create: buffers must be 256-byte aligned"); return nullptr; }
    e->params = params; e->grads = grads; e->m = adam_m; e->v = adam_v; e->shadow = bf16_shadow; e->wt = weights_t; e->ws = (char*)workspace;
    e->last_B = 0;
    if (e->W.n_guards > 0) {                              // debug mode: arm the guard words (synchronous copies; never on the production path)
        uint32_t pat[64];
        for (int i = 0; i < 64; ++i) pat[i] = GUARD_WORD ^ (uint32_t)i;
        for (int k = 0; k < e->W.n_guards; ++k)
            if (hipMemcpy(e->ws + e->W.guard_off[k], pat, 256, hipMemcpyHostToDevice) != hipSuccess) { free(e); mi_fail(MI_ERR_STATE, "mi_vae_create: arming the debug guards failed"); return nullptr; }
    }
    return e;
}

// Debug (MI355_DEBUG_GUARDS=1): *n_regions = guarded workspace regions (0: the mode is off), *n_bad = guards whose 256 bytes no longer hold the pattern,
// i.e. regions some kernel wrote past; mi_last_error() names the first one.  Synchronises the device.  guard_index >= 0: also returns that guard's byte
// offset in *guard_offset (tests corrupt one on purpose).
int mi_vae_debug_check_guards(void* h, int* n_regions, int* n_bad, int guard_index, long long* guard_offset) {
    VaeEngine* e = (VaeEngine*)h;
    if (!e || !n_regions || !n_bad) return mi_fail(MI_ERR_ARG, "mi_vae_debug_check_guards: missing arguments");
    *n_regions = e->W.n_guards; *n_bad = 0;
    if (guard_offset) *guard_offset = (guard_index >= 0 && guard_index < e->W.n_guards) ? e->W.guard_off[guard_index] : -1;
    if (e->W.n_guards == 0) return MI_OK;
    if (hipDeviceSynchronize() != hipSuccess) return mi_fail(MI_ERR_STATE, "mi_vae_debug_check_guards: device error before the check");
    int first = -1;
    for (int k = 0; k < e->W.n_guards; ++k) {
        uint32_t got[64];
        if (hipMemcpy(got, e->ws + e->W.guard_off[k], 256, hipMemcpyDeviceToHost) != hipSuccess) return mi_fail(MI_ERR_STATE, "mi_vae_debug_check_guards: copy failed");
        bool bad = false;
        for (int i = 0; i < 64; ++i) bad |= got[i] != (GUARD_WORD ^ (uint32_t)i);
        if (bad) { ++*n_bad; if (first < 0) first = k; }
    }
    if (first >= 0) {
        char msg[160];
        snprintf(msg, sizeof(msg), "workspace guard %d (byte offset %lld) overwritten: a kernel wrote past the region in front of it", first, e->W.guard_off[first]);
        mi_fail(MI_OK, msg);
    }
    return MI_OK;
}

void mi_vae_destroy(void* h) {
    VaeEngine* e = (VaeEngine*)h;
    if (e && e->side_ok == 1) { hipStreamDestroy(e->side); hipEventDestroy(e->ev_ready); hipEventDestroy(e->ev_done); }
    if (e && e->third_ok == 1) { hipStreamDestroy(e->third); hipEventDestroy(e->ev_lat); hipEventDestroy(e->ev_third); }
    if (e && e->main2_ok == 1) { hipStreamDestroy(e->main2); hipEventDestroy(e->ev_in); hipEventDestroy(e->ev_out); }
    free(h);
}

// bf16 mode: refresh the shadow weights from the fp32 masters (after load / init; Adam keeps them in sync afterwards)
int mi_vae_sync_shadow(void* h, void* stream) {
    VaeEngine* e = (VaeEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "vae engine: null handle");
    if (e->d.dtype == MI_BF16) CK(mi_cast_f32_to_bf16(stream, e->params, e->shadow, e->L.total));
    if (e->d.dtype == MI_BF16X3) CK(mi_cast_f32_to_split(stream, e->params, e->shadow, e->L.total));
    return refresh_transposed(e, stream);
}

// device pointers into the workspace (valid after the corresponding call; fp32): 0 losses[2] (recon, kl), 1 mean [B,Z],
// 2 logvar [B,Z], 3 kl_row [B];  4 logits [B,P] (T), 5 z [B,Z] (T)
void* mi_vae_buffer(void* h, int which) {
    VaeEngine* e = (VaeEngine*)h;
    if (!e) return nullptr;
    switch (which) {
        case 0: return e->at(e->W.out2);
        case 1: return e->at(e->W.mean);
        case 2: return e->at(e->W.logvar);
        case 3: return e->at(e->W.kl_row);
        case 4: return e->at(e->W.dec[4]);
        case 5: return e->at(e->W.z);
        case 6: case 7: case 8: case 9: case 10: case 11: case 12: case 13: case 14: case 15: case 16: return e->ares_ok ? e->at(e->W.wfrag[which - 6]) : nullptr;      // the fragment-ordered weight copies (tests)
        default: return nullptr;
    }
}

// Forward pass + ELBO terms of one minibatch: VAE.evaluate's per-batch sess.run (vae/models.py:226-229) and the
// forward half of train_step (:213-216).  src/tgt: fp32 frame tables [n_frames, ...] on the device; idx: int32 [B] or null.
// inv_batch = 1/B_global (data parallel: each rank passes its local rows).  want_grad: also write dlogits for backward.
int mi_vae_forward(void* h, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch,
                   const float* eps, int sample, int want_grad, float* metrics3, float metric_weight) {
    VaeEngine* e = (VaeEngine*)h;
    CK(check_batch(e, B));
    const MiVaeDesc& d = e->d; const Geom& g = e->g;
    struct StopEventGuard { ~StopEventGuard() { mi_tl_stop_event = nullptr; } } stop_guard;      // an error return between arming and the launch must not leave the event armed for this thread's next, unrelated launch
    CK(run_encoder(e, stream, src, frames_u8, idx, B, eps, sample, want_grad));
    e->last_u8 = frames_u8 ? 1 : 0;
    const int P = g.dh[4] * g.dw[4] * g.dc[4];
    const float kl_floor = d.kl_tolerance > 0.f ? d.kl_tolerance * d.z_dim : 0.f;
    // the BiasAddGrad of deconv4 (sum of dlogits per target channel) rides on the loss pass when the gradient is wanted
    const bool fuse_b4 = want_grad && e->grads && d.ct <= 3;
    if (fuse_b4 && e->b4_fused)
        return mi_fail(MI_ERR_STATE, "mi_vae_forward(want_grad=1): the previous forward's gradient was never consumed by mi_vae_backward "
                                     "(its deconv4 bias gradient is already in the gradient buffer)");
    e->b4_fused = fuse_b4 ? 1 : 0;
    // decoder tail.  Training pass, bf16, rgb target: ONE launch computes deconv4, the loss, deconv4's filter gradient (straight into the gradient
    // buffer) and the gradient of deconv3's output (dectail_tile.hpp): dlogits never exist in HBM, deconv3's output is read once instead of three
    // times, and the backward pass starts at deconv3.  Otherwise: deconv4 with the loss fused into its epilogue (dlogits written), or the plain ops.
    static int tail_on = -1;
    if (tail_on < 0) { const char* ev = getenv("MI355_DECTAIL"); tail_on = (ev && ev[0] == '0') ? 0 : 1; }
    const bool tail_try = tail_on && want_grad && e->grads && d.dtype == MI_BF16 && d.ct == 3 && g.dc[3] == 32 && DEC_K[3] == 4 && e->W.scratch_bytes > 0;
    e->tail_fused = 0;
    CK(run_decoder(e, stream, B, 3, want_grad && !tail_try));
    int nblk = 0;
    e->fwd_produced = 0;
    if (tail_try) {
        static int kev = -1;
        if (kev < 0) { const char* ev = getenv("MI355_KEVENT"); kev = ev ? atoi(ev) : 1; }
        const bool carry = kev && kev != 3 && e->side_ok == 1 && e->defer_fin && e->tm.mode != 1;      // (MI355_KEVENT=3: in the backward pass only)      // (mi_vae_train_step: nothing else is issued between this kernel and the backward pass)
        if (carry) mi_tl_stop_event = e->ev_ready;
        TOP(e, stream, OP_DECONV_FWD + 3, mi_deconv2d_tail_fused(stream, d.dtype, e->at(e->W.dec[3]), B, g.dh[3], g.dw[3], g.dc[3], e->wptr(18), e->wtptr(18), e->bptr(19), DEC_K[3], DEC_K[3], g.dc[4],
                                           tgt, frames_u8, idx, (long long)P, d.loss_kind, inv_batch, e->at(e->W.gdec[3]), e->gptr(18),
                                           (float*)e->at(e->W.partial), (float*)e->at(e->W.bpart), e->partial_cap, &nblk, e->at(e->W.tail_slabs), e->W.tail_slab_bytes, 0));
        if (carry) { e->fwd_produced = (nblk > 0 && mi_tl_stop_event == nullptr) ? 1 : 0; mi_tl_stop_event = nullptr; }
        if (nblk > 0) { e->tail_fused = 1; e->tail_nblk = nblk; }      // (its slab reduce runs inside mi_vae_backward)
        {
            static int defer_on = -1;
            if (defer_on < 0) { const char* ev = getenv("MI355_DEFER"); defer_on = (ev && ev[0] == '0') ? 0 : 1; }
            if (nblk > 0 && !defer_on) { CK(mi_deconv2d_tail_reduce(stream, e->at(e->W.tail_slabs), nblk, e->gptr(18))); e->tail_nblk = 0; }
        }
    }
    // (the fused forms never store the logits: only the loss partial sums and dlogits / the gradients leave the kernel)
    if (nblk == 0) TOP(e, stream, OP_DECONV_FWD + 3, mi_deconv2d_nhwc_fwd_bce_u8(stream, d.dtype, e->at(e->W.dec[3]), B, g.dh[3], g.dw[3], g.dc[3], e->wptr(18), e->bptr(19), DEC_K[3], DEC_K[3], g.dc[4],
                                            nullptr, tgt, frames_u8, idx, (long long)P, d.loss_kind, inv_batch, want_grad ? e->at(e->W.gdec[4]) : nullptr,
                                            (float*)e->at(e->W.partial), (float*)e->at(e->W.bpart), e->partial_cap, &nblk));
    e->fin.pending = 0;
    if (nblk > 0 && e->defer_fin && want_grad) {
        e->fin.pending = 1; e->fin.nblk = nblk; e->fin.B = B; e->fin.kl_floor = kl_floor; e->fin.inv_batch = inv_batch; e->fin.metric_weight = metric_weight;
        e->fin.metrics3 = metrics3; e->fin.dbias = fuse_b4 ? e->gptr(19) : nullptr;
    } else if (nblk > 0) {
        TOP(e, stream, OP_FINALIZE, mi_vae_finalize_losses_flat(stream, (const float*)e->at(e->W.partial), nblk, (const float*)e->at(e->W.kl_row), kl_floor, B, inv_batch,
                                       (float*)e->at(e->W.out2), metrics3, metric_weight, (const float*)e->at(e->W.bpart), nblk, d.ct, fuse_b4 ? e->gptr(19) : nullptr));
    } else {
        if (frames_u8) return mi_fail(MI_ERR_ARG, "vae engine: uint8 target frames need the fused decoder-tail loss kernel (1- or 3-channel target, narrow kernels enabled)");
        TOP(e, stream, OP_DECONV_FWD + 3, mi_deconv2d_nhwc_fwd(stream, d.dtype, e->at(e->W.dec[3]), B, g.dh[3], g.dw[3], g.dc[3], e->wptr(18), e->bptr(19), DEC_K[3], DEC_K[3], g.dc[4], 0, e->at(e->W.dec[4])));
        TOP(e, stream, OP_RECON_LOSS, mi_bce_logits_fwd_bwd_bias(stream, d.dtype, e->at(e->W.dec[4]), (const float*)tgt, idx, (long long)P, B, P, d.loss_kind, inv_batch,
                                 want_grad ? e->at(e->W.gdec[4]) : nullptr, (float*)e->at(e->W.partial), d.ct, fuse_b4 ? e->gptr(19) : nullptr));
        TOP(e, stream, OP_FINALIZE, mi_vae_finalize_losses(stream, (const float*)e->at(e->W.partial), e->nchunks, (const float*)e->at(e->W.kl_row), kl_floor, B,
                                  inv_batch, (float*)e->at(e->W.out2), metrics3, metric_weight));
    }
    e->last_B = B;
    return MI_OK;
}

// Backward of the last mi_vae_forward(want_grad=1) into the flat fp32 gradient buffer (which must be zero on entry;
// mi_vae_apply_adam clears it again).  part: 0 = everything, 1 = decoder half (deconv4..dense1 + dz), 2 = encoder half = 3 (heads + conv4:
// 88 % of the encoder's parameters) followed by 4 (conv3..conv1).  The parts exist so the data-parallel host can all-reduce the gradient
// bucket of a finished part while the next part runs; every separately called part ends with the stream join that completes its bucket.
int mi_vae_backward(void* h, void* stream, const void* src, const int* idx, const float* eps, float inv_batch, int part) {
    VaeEngine* e = (VaeEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "vae engine: null handle");
    if (!eps) eps = e->last_eps;                          // the noise the forward pass drew itself (or was given)
    if (!eps) return mi_fail(MI_ERR_STATE, "mi_vae_backward: no sampling forward pass recorded");
    const int B = e->last_B;
    if (B < 1) return mi_fail(MI_ERR_STATE, "mi_vae_backward: no forward pass recorded");
    if (part == 0 || part == 1) e->tail_sched = e->tail_fused;      // (remembered for the encoder parts of a split backward)
    const bool tail_was_fused = e->tail_sched != 0;
    if (!e->grads) return mi_fail(MI_ERR_STATE, "mi_vae_backward: engine created without a gradient buffer");
    const MiVaeDesc& d = e->d; const Geom& g = e->g; const Workspace& W = e->W;
    void* st = stream;
    // The filter (+bias) gradient of a layer and its input gradient only share their INPUT (the gradient of the layer's output), so the
    // filter gradients run on a second stream, each one released by an event recorded on the caller's stream when its operand is ready.
    // Every kernel on this path fills a whole CU (150 KB of LDS), so the gain is in the tails: a 342-block launch leaves 2/3 of the
    // chip idle in its second round, which the neighbouring launch now fills.  MI355_BWD_STREAMS=0 serialises everything again.
    static int two_streams = -1;
    if (two_streams < 0) { const char* ev = getenv("MI355_BWD_STREAMS"); two_streams = (ev && ev[0] == '0') ? 0 : 1; }
    if (two_streams && !e->side_ok) {
        // MI355_SIDE_PRIO=-1 / 1: the filter-gradient queue above / below the caller's queue in the hardware scheduler's priority order (default 0: equal; A/B knob)
        static int side_prio = -99;
        if (side_prio == -99) { const char* ev = getenv("MI355_SIDE_PRIO"); side_prio = ev ? atoi(ev) : 0; }
        int prio_lo = 0, prio_hi = 0;
        if (side_prio != 0) hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);        // (numerically: lowest priority = largest value)
        const int prio = side_prio < 0 ? prio_hi : prio_lo;
        // MI355_CU_SPLIT=N (measurement aid): the two queues of the backward pass on DISJOINT compute units -- mask bits [0, N) for the caller-side queue (an
        // engine stream the pass forks to and joins from), [N, 256) for the filter-gradient queue; KFD deals the mask bits round-robin over the eight XCDs, so both halves
        // span all of them.  Tells time-slicing of whole CUs (147 KB-LDS blocks) from HBM / L2 interference: DESIGN 3.16.
        const int cu_split = cu_split_env();
        uint32_t mask_side[8], mask_main[8];
        for (int w = 0; w < 8; ++w) { mask_side[w] = 0; mask_main[w] = 0; }
        for (int b = 0; b < 256; ++b) { if (b < cu_split) mask_main[b >> 5] |= 1u << (b & 31); else mask_side[b >> 5] |= 1u << (b & 31); }
        if (cu_split > 0 && !e->main2_ok) {
            if (hipExtStreamCreateWithCUMask(&e->main2, 8, mask_main) == hipSuccess && hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming) == hipSuccess) e->main2_ok = 1;
            else e->main2_ok = -1;
        }
        if ((cu_split > 0 ? hipExtStreamCreateWithCUMask(&e->side, 8, mask_side) : side_prio == 0 ? hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) : hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, prio)) == hipSuccess &&
            hipEventCreateWithFlags(&e->ev_ready, ready_event_flags()) == hipSuccess &&
            hipEventCreateWithFlags(&e->ev_done, ready_event_flags()) == hipSuccess) e->side_ok = 1;
        else e->side_ok = -1;
    }
    static int third_on = -1;                             // MI355_THIRD=1: the latent layers' filter / bias gradients, the tail's slab sum and the loss finalisation on a third stream
    if (third_on < 0) { const char* ev = getenv("MI355_THIRD"); third_on = (ev && ev[0] == '1') ? 1 : 0; }
    if (third_on && two_streams && !e->third_ok) {
        if (hipStreamCreateWithFlags(&e->third, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&e->ev_lat, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&e->ev_third, hipEventDisableTiming) == hipSuccess) e->third_ok = 1;
        else e->third_ok = -1;
    }
    const bool fork = two_streams && e->side_ok == 1 && e->tm.mode != 1;      // per-op timing (mode 1) wants one op at a time
    const bool split_cus = fork && cu_split_env() > 0 && e->main2_ok == 1 && part == 0;
    if (split_cus) { hipEventRecord(e->ev_in, (hipStream_t)stream); hipStreamWaitEvent(e->main2, e->ev_in, 0); st = (void*)e->main2; }
    struct SplitGuard { bool on; VaeEngine* e; void* caller; ~SplitGuard() { if (on) { hipEventRecord(e->ev_out, e->main2); hipStreamWaitEvent((hipStream_t)caller, e->ev_out, 0); } } } split_guard{split_cus, e, stream};
    struct StopEventGuard { ~StopEventGuard() { mi_tl_stop_event = nullptr; } } stop_guard;      // (see mi_vae_forward)
    void* sw = fork ? (void*)e->side : st;                                     // stream of the filter gradients
    // Hand-over of a gradient tensor from the caller's stream to the filter-gradient stream.  Round 3 form: hipEventRecord behind the producing kernel -- a marker
    // packet of its own that costs the PRODUCING queue a 6-8 us bubble each time (six per step on the critical queue, profiles/r03_d).  Round 4 (MI355_KEVENT=0: the record form; measured 0.913 -> 0.896 ms per step, two interleaved pairs on one box):
    // the producing kernel itself carries the event as the completion signal of its dispatch packet (hipExtLaunchKernelGGL's stop event, mi_internal.hpp MI_LAUNCH):
    // no marker on the caller's queue, and the other queue's wait resolves the moment that kernel retires.
    static int kev_env = -1;
    if (kev_env < 0) { const char* ev = getenv("MI355_KEVENT"); kev_env = ev ? atoi(ev) : 1; }
    const int kev = kev_env;
    bool produced = fork && kev && e->fwd_produced && (part == 0 || part == 1);      // ev_ready already rides on the last kernel issued on st
    e->fwd_produced = 0;
    bool mid_flush_pending = false;
    auto release = [&]() {                                                     // "everything issued on st so far is an input of the next sw op"
        if (fork) { if (!produced) hipEventRecord(e->ev_ready, (hipStream_t)st); hipStreamWaitEvent(e->side, e->ev_ready, 0); }
        produced = false;
    };
    // call(): a layer op whose single kernel produces the tensor the NEXT release() hands over
    auto arm = [&]() { produced = false; if (fork && kev) mi_tl_stop_event = e->ev_ready; };
    auto armed_ok = [&]() { if (fork && kev) { produced = mi_tl_stop_event == nullptr; mi_tl_stop_event = nullptr; } };
    // In two-stream mode the split reductions of the filter gradients are deferred to the end of the side stream's work (they are tiny, but
    // next to a big input-gradient kernel each takes 15-30 us instead of ~7): every layer writes its slabs into its own scratch region.
    const bool defer = fork && W.scratch_bytes > 0;
    const long long region = W.scratch_bytes / SCRATCH_REGIONS;
    int next_region = 0;
    auto scratch_of = [&]() -> void* {
        void* ptr = (char*)e->at(W.scratch) + (defer ? (next_region % SCRATCH_REGIONS) * region : 0);
        ++next_region;
        return ptr;
    };
    const long long scratch_sz = defer ? region : W.scratch_bytes;
    // latent-layer filter / bias gradients (dense1, heads): slabs of their row splits in the region that belongs to the stream they are issued on
    // (reused in stream order: each call's ordered reduce is issued right behind its kernel)
    // End of a full two-stream pass (round 4, MI355_TAIL_FUSE=0 switches it off): the slab sums of the encoder head, the decoder tail and the latent layers' filter /
    // bias gradients -- seven to eleven latency-bound launches in a row on the caller's stream -- are recorded and issued as ONE launch (mi_small_reduce_flush);
    // every job then keeps its own piece of the tail scratch until that launch.
    static int tail_fuse_on = -1;
    if (tail_fuse_on < 0) { const char* ev = getenv("MI355_TAIL_FUSE"); tail_fuse_on = (ev && ev[0] == '0') ? 0 : 1; }
    bool tail_defer = false;
    long long tail_bump = 0;
    // A backward PART on two streams (the data-parallel step; round 6): the small slab sums of the part's filter-gradient stream -- the latent layer's filter + bias gradient, the bias
    // rows of a lone raw-staged filter gradient -- are recorded the same way and issued as ONE launch in front of the part's join (they were two or three ~5 us launches per part on the
    // stream that ends the data-parallel step); MI355_PART_FUSE=0: a launch each.  (tail_defer belongs to the full pass, part 0: the two never meet.)
    static int part_fuse_on = -1;
    if (part_fuse_on < 0) { const char* ev = getenv("MI355_PART_FUSE"); part_fuse_on = (ev && ev[0] == '0') ? 0 : 1; }
    bool side_defer = false;
    if (part_fuse_on && fork && part != 0 && W.scratch_tail_bytes > 0 && e->tm.mode != 1) { mi_small_reduce_defer(1); mi_small_reduce_bind(sw); side_defer = true; }
    struct SrGuard { bool* on; bool* on2; ~SrGuard() { if (*on || *on2) mi_small_reduce_defer(0); } } sr_guard{&tail_defer, &side_defer};
    auto small_ws = [&](void* s_, long long need, long long* bytes) -> void* {
        if ((tail_defer && s_ == st) || (side_defer && s_ == sw)) {
            need = (need + 255) / 256 * 256;
            if (need <= W.scratch_tail_bytes) {
                if (tail_bump + need > W.scratch_tail_bytes) { mi_small_reduce_flush(s_); tail_bump = 0; }      // (the jobs recorded so far read their slabs before anything below overwrites them: same stream)
                void* ptr = (char*)e->at(W.scratch_tail) + tail_bump;
                tail_bump += need; *bytes = need;
                return ptr;
            }
            mi_small_reduce_flush(s_);
        }
        if (fork && s_ == (void*)e->side) { *bytes = W.scratch_side_bytes; return e->at(W.scratch_side); }
        if (e->third_ok == 1 && s_ == (void*)e->third) { *bytes = W.scratch_third_bytes; return e->at(W.scratch_third); }
        *bytes = W.scratch_main_bytes; return e->at(W.scratch_main);
    };
    auto bias_grad = [&](void* s_, const void* x, long long M, int N, float* out) -> int {
        long long nb = 0; void* ws_ = small_ws(s_, mi_colsum_scratch_bytes(d.dtype, M, N), &nb);
        return mi_colsum_ws(s_, d.dtype, x, M, N, out, ws_, nb);
    };
    // round 4: the latent layers' BiasAddGrad rides on their filter gradient as one more row of the same product (mi_gemm_wgrad_bias_ws: a column of ones in the
    // loader) -- two launches less at the end of the pass; MI355_DENSE_BIAS_FUSED=0: the separate column sums
    static int bias_fused = -1;
    if (bias_fused < 0) { const char* ev = getenv("MI355_DENSE_BIAS_FUSED"); bias_fused = (ev && ev[0] == '0') ? 0 : 1; }
    auto dense_wgrad = [&](void* s_, const void* a, const void* dy, int M, int K, int N, float* dw, float* db) -> int {
        long long nb = 0; void* ws_ = small_ws(s_, mi_gemm_wgrad_scratch_bytes(d.dtype, M, K, N), &nb);
        return mi_gemm_wgrad_bias_ws(s_, d.dtype, a, dy, M, K, N, dw, bias_fused ? db : nullptr, ws_, nb);
    };
    if (defer) mi_tapwgrad_defer(1);
    auto join = [&]() {
        if (defer) mi_tapwgrad_flush(sw);
        if (side_defer) { mi_small_reduce_flush(sw); mi_small_reduce_defer(0); side_defer = false; }
        if (fork && e->dp_open_join) {                      // (one-call data-parallel step, parts 1 and 3) everything of this part that ran on the caller's stream is complete on the other one
            hipEventRecord(e->ev_done, (hipStream_t)st); hipStreamWaitEvent(e->side, e->ev_done, 0);
            e->dp_side_ready = 1;
            return;
        }
        if (fork) { hipEventRecord(e->ev_done, e->side); hipStreamWaitEvent((hipStream_t)st, e->ev_done, 0); }
    };
    struct DeferGuard { bool on; ~DeferGuard() { if (on) mi_tapwgrad_defer(0); } } guard{defer};
    // The position-split partial sums of the six raw-staged filter gradients are stored ROUNDED TO BF16 (round 3): 211 MB written + 214 MB read per step
    // become half of that.  ~250 slabs per element, each within 2^-9 of ITS OWN value (RMS 2^-9 / sqrt 3) with independent rounding errors: the error of the
    // total is that fraction of sqrt(sum s_i^2), i.e. ~1e-3 of the RMS element of dW when the partial sums have independent signs (the usual case for a
    // gradient; 7e-5 only if they all agree) -- measured at batch 512 on the ops by tests/test_ops_gpu.py::test_bf16_partial_sum_slabs_error_bound_at_batch_512,
    // and inside the 1.25 e_emul + 2e-3 gradient criterion of the bf16 engine with bench.py's parity.bf16.grad_worst = 0.79 (the operands were bf16 to begin
    // with: their own rounding contributes 3e-3 ... 6e-2 of the tensor max).  The layer-op entry points keep exact fp32 slabs.
    // Measured, three interleaved pairs on one box: 0.966 -> 0.943 ms per step.  MI355_SLAB_BF16=0: fp32 slabs.
    static int slab16 = -1;
    if (slab16 < 0) { const char* ev = getenv("MI355_SLAB_BF16"); slab16 = (ev && ev[0] == '0') ? 0 : 1; }
    struct SlabGuard { int prev; bool on; ~SlabGuard() { if (on) mi_tapwgrad_slab_bf16(prev); } } slab_guard{-1, false};
    if (slab16 && d.dtype == MI_BF16) { slab_guard.prev = mi_tapwgrad_slab_bf16(1); slab_guard.on = true; }
    static int late_on = -1;                              // MI355_LATE_DENSE=0: dense1 / heads filter gradients on the filter-gradient stream as in round 2
    if (late_on < 0) { const char* ev = getenv("MI355_LATE_DENSE"); late_on = (ev && ev[0] == '0') ? 0 : 1; }
    const bool late_dense = late_on && fork && part == 0;
    const bool use_third = late_dense && third_on && e->third_ok == 1;
    static int heads_main = -1;                           // MI355_HEADS_MAIN=0: the heads' filter / bias gradient stay on the filter-gradient stream behind a fused encoder head
    if (heads_main < 0) { const char* ev = getenv("MI355_HEADS_MAIN"); heads_main = (ev && ev[0] == '0') ? 0 : 1; }
    if (part == 0 || part == 1) {
        for (int i = 3; i >= 0; --i) {                       // deconv(i+1): input dec[i] -> output dec[i+1]
            if (i == 3 && e->tail_fused) continue;           // deconv4's two gradients were computed by the forward pass's decoder-tail kernel
            const void* gy = e->at(W.gdec[i + 1]);
            release();                                       // gy is complete on st (loss pass / previous input gradient)
            // BiasAddGrad is fused into the filter-gradient call
            TOP(e, sw, OP_DECONV_WGRAD + i, mi_deconv2d_nhwc_wgrad_ws(sw, d.dtype, gy, B, g.dh[i + 1], g.dw[i + 1], g.dc[i + 1], e->at(W.dec[i]), DEC_K[i], DEC_K[i], g.dc[i], e->gptr(12 + 2 * i), scratch_of(), scratch_sz, (i == 3 && e->b4_fused) ? nullptr : e->gptr(13 + 2 * i)));
            if (i == 0 && e->ares_ok) {                      // deconv1's input gradient: conv form on the activation-resident kernel (no mask: dense1 has no ReLU)
                int launched = 0;
                TOP(e, st, OP_DECONV_DGRAD + i, mi_ares_conv(st, d.dtype, 0, gy, B, e->at(W.wfrag[3]), nullptr, 0, nullptr, e->at(W.gdec[0]), &launched));
                if (launched) continue;
            }
            if (i > 0) arm();                                // its output is the next layer's filter-gradient operand
            if (i == 1 && e->ares_ok && e->ares_mid && rc_wfrag_enabled()) mi_tl_rc_wfrag = e->at(W.wfrag[7]);      // deconv2's input gradient: conv form, fragment-ordered weights
            if (i == 2 && e->ares_ok && rc_small_frag_ok(e) && rc_wfrag_enabled() && rc_wfrag6_enabled()) mi_tl_rc_wfrag = e->at(W.wfrag[10]);   // deconv3's: conv form, k = 5 (pack form 6)
            struct WfragGuard2 { ~WfragGuard2() { mi_tl_rc_wfrag = nullptr; } } wfrag_guard2;
            TOP(e, st, OP_DECONV_DGRAD + i, mi_deconv2d_nhwc_dgrad_bits(st, d.dtype, gy, B, g.dh[i + 1], g.dw[i + 1], g.dc[i + 1], e->wtptr(12 + 2 * i), 1, DEC_K[i], DEC_K[i], g.dc[i],
                                      i > 0 ? e->at(W.dec[i]) : nullptr, (i == 3 && e->bits3_ok) ? e->at(W.bits_dec3) : nullptr, e->at(W.gdec[i])));
            if (i > 0) armed_ok();
        }
        e->b4_fused = 0; e->tail_fused = 0;
        {
            // Round 5: the slab sums of the three DECODER filter gradients are issued here, behind deconv1's filter gradient, instead of with the encoder's at the very end:
            // the filter-gradient stream idles ~20 us at this point (conv4's filter gradient waits for the latent chain dense1.dgrad -> reparam.bwd -> heads.dgrad on the
            // caller's stream, six small launches that leave the chip nearly empty), and the reduce at the end of the pass -- which ends the longer of the two queues --
            // shrinks from six layers to three.  MI355_MID_FLUSH=0: one reduce at the end (A/B runs).
            static int mid_flush = -1;
            if (mid_flush < 0) { const char* ev = getenv("MI355_MID_FLUSH"); mid_flush = (ev && ev[0] == '0') ? 0 : 1; }
            // MI355_MID_FLUSH_LATE=1 (round 6 A/B): the same launch on the same queue behind the same kernel, but SUBMITTED behind the latent chain's launches (dense1.dgrad,
            // reparam.bwd, heads.dgrad) -- on boxes whose dispatcher serves the older submission first the chain's 128-block kernels otherwise queue behind the reduce's 1,864 blocks
            static int mid_late = -1;
            if (mid_late < 0) { const char* ev = getenv("MI355_MID_FLUSH_LATE"); mid_late = (ev && ev[0] == '1') ? 1 : 0; }
            mid_flush_pending = defer && part == 0 && mid_flush && mid_late;
            if (defer && part == 0 && mid_flush && !mid_late) CK(mi_tapwgrad_flush(sw));
        }
        if (e->tail_nblk > 0 && !late_dense) {               // deconv4's filter gradient: the fused tail's per-block sums -> the gradient buffer
            CK(mi_deconv2d_tail_reduce(st, e->at(W.tail_slabs), e->tail_nblk, e->gptr(18)));      // (full two-stream backward: at the tail of the caller's stream, below)
            e->tail_nblk = 0;
        }
        // dense1: h = z W1 + b1.  Full two-stream backward (round 3): the four latent-side filter / bias gradients (dense1, heads: ~60 us of the
        // filter-gradient stream, which is the longer one) are issued on the caller's stream at the very END of the pass, where that stream would
        // otherwise wait ~100 us for the other one; their operands (gdec0, z, dheads, act4) stay intact until then and no event is needed for them.
        if (!late_dense) {
            release();
            if (!bias_fused) TOP(e, sw, OP_DENSE1_BIAS, bias_grad(sw, e->at(W.gdec[0]), B, g.flat, e->gptr(11)));
            TOP(e, sw, OP_DENSE1_WGRAD, dense_wgrad(sw, e->at(W.z), e->at(W.gdec[0]), B, d.z_dim, g.flat, e->gptr(10), e->gptr(11)));
        }
        TOP(e, st, OP_DENSE1_DGRAD, mi_gemm_bias_act(st, d.dtype, e->at(W.gdec[0]), B, g.flat, e->wptr(10), 1, d.z_dim, nullptr, 0, nullptr, e->at(W.dz_slab), 1, e->ns_dz));
        if (part == 1) join();                               // a full backward joins once, at its end: nothing below reads a filter gradient
    }
    if (part < 0 || part > 4) return mi_fail(MI_ERR_SHAPE, "mi_vae_backward: part must be 0..4");
    const bool upper = part == 0 || part == 2 || part == 3, lower = part == 0 || part == 2 || part == 4;
    if (upper) {
        const float kl_floor = d.kl_tolerance > 0.f ? d.kl_tolerance * d.z_dim : 0.f;
        TOP(e, st, OP_REPARAM_BWD, mi_vae_reparam_kl_bwd(st, d.dtype, (const float*)e->at(W.dz_slab), e->ns_dz, (const float*)e->at(W.mean), (const float*)e->at(W.logvar),
                                 eps, (const float*)e->at(W.kl_row), d.beta, kl_floor, inv_batch, B, d.z_dim, e->at(W.dheads)));
        if (!late_dense) {
            release();
            if (!bias_fused) TOP(e, sw, OP_HEADS_BIAS, bias_grad(sw, e->at(W.dheads), B, 2 * d.z_dim, e->gptr(9)));
            TOP(e, sw, OP_HEADS_WGRAD, dense_wgrad(sw, e->at(W.act[4]), e->at(W.dheads), B, g.flat, 2 * d.z_dim, e->gptr(8), e->gptr(9)));
        }
        if (!use_third) arm();                               // gact4: conv4's filter-gradient operand
        TOP(e, st, OP_HEADS_DGRAD, mi_gemm_bias_act(st, d.dtype, e->at(W.dheads), B, 2 * d.z_dim, e->wptr(8), 1, g.flat, nullptr, 0, e->at(W.act[4]), e->at(W.gact[4]), 0, 1));
        if (!use_third) armed_ok();
        if (mid_flush_pending) { mid_flush_pending = false; CK(mi_tapwgrad_flush(sw)); }
        if (use_third) {
            // everything the latent layers' gradients read exists from here on (gdec0, z, dheads, act4; the tail's slabs and loss partials since the forward pass): they run on
            // their own stream under the encoder half instead of serialising ~70 us of small launches at the end of the caller's stream
            hipStream_t s3 = e->third;
            hipEventRecord(e->ev_lat, (hipStream_t)st); hipStreamWaitEvent(s3, e->ev_lat, 0);
            if (!bias_fused) TOP(e, s3, OP_DENSE1_BIAS, bias_grad(s3, e->at(W.gdec[0]), B, g.flat, e->gptr(11)));
            TOP(e, s3, OP_DENSE1_WGRAD, dense_wgrad(s3, e->at(W.z), e->at(W.gdec[0]), B, d.z_dim, g.flat, e->gptr(10), e->gptr(11)));
            if (!bias_fused) TOP(e, s3, OP_HEADS_BIAS, bias_grad(s3, e->at(W.dheads), B, 2 * d.z_dim, e->gptr(9)));
            TOP(e, s3, OP_HEADS_WGRAD, dense_wgrad(s3, e->at(W.act[4]), e->at(W.dheads), B, g.flat, 2 * d.z_dim, e->gptr(8), e->gptr(9)));
            if (e->tail_nblk > 0) { CK(mi_deconv2d_tail_reduce(s3, e->at(W.tail_slabs), e->tail_nblk, e->gptr(18))); e->tail_nblk = 0; }
            if (e->fin.pending) {
                e->fin.pending = 0;
                TOP(e, s3, OP_FINALIZE, mi_vae_finalize_losses_flat(s3, (const float*)e->at(W.partial), e->fin.nblk, (const float*)e->at(W.kl_row), e->fin.kl_floor, e->fin.B, e->fin.inv_batch,
                                               (float*)e->at(W.out2), e->fin.metrics3, e->fin.metric_weight, (const float*)e->at(W.bpart), e->fin.nblk, d.ct, e->fin.dbias));
            }
            hipEventRecord(e->ev_third, s3);
        }
    }
    if (upper || lower) {
        bool enc_fused = false, dense_done = false, pair_used = false;
        for (int i = NCONV - 1; i >= 0; --i) {               // conv(i+1): input act[i] -> output act[i+1]
            if (i == NCONV - 1 ? !upper : !lower) continue;
            const void* gy = e->at(W.gact[i + 1]);
            const void* x = i == 0 ? (const void*)src : e->at(W.act[i]);
            // The filter-gradient stream is the longer one of the two: conv1 has no input gradient, and since the register-weight kernels shortened
            // conv2's / conv3's input gradients the caller's stream has room for one more -- the filter gradients of conv1 AND conv3 run there (mask 5,
            // measured against 1 / 3 / 9 / 13 interleaved on one box: 1.068 vs 1.090 / 1.111 / 1.078 / 1.079 ms per step; moving a decoder layer's
            // filter gradient instead: no gain), without the shared split scratch, which the other stream may still be using (fp32 atomics into dW).
            // MI355_WGRAD_MAIN_MASK: bit i = conv(i+1).
            // Round 3: with the decoder tail fused into the forward pass (its two gradient launches were the head of both streams) the filter-gradient
            // stream is the shorter one again and takes conv3's filter gradient back: mask 1 (0.969 vs 0.984 / 0.988 / 0.992 / 1.014 ms for 5 / 3 / 0 / 4).
            static int main_mask = -2;
            if (main_mask == -2) { const char* ev = getenv("MI355_WGRAD_MAIN_MASK"); main_mask = ev ? atoi(ev) : -1; }
            const int mm = main_mask >= 0 ? main_mask : (tail_was_fused ? 1 : 5);
            const bool on_main = ((mm >> i) & 1) != 0;
            void* sg = on_main ? st : sw;
            if (i == 0 && enc_fused) continue;               // conv1's filter / bias gradient came out of the fused encoder-head kernel below
            if (!on_main) release();
            // on the caller's stream next to a live filter-gradient stream: its own scratch region, and its slab reduce is issued right behind it on THIS stream
            // (the deferred list is flushed on the other one)
            const bool own = on_main && fork;
            if (own) mi_tapwgrad_defer_pause(1);
            const int rcw = [&]() -> int {
                TOP(e, sg, OP_CONV_WGRAD + i, mi_conv2d_nhwc_wgrad_ws(sg, d.dtype, x, i == 0 ? idx : nullptr, i == 0 ? (e->last_u8 ? 2 : 1) : 0, B, g.ih[i], g.iw[i], g.c[i], gy, 4, 4, g.c[i + 1], e->gptr(2 * i),
                                                                       own ? e->at(W.scratch_main) : scratch_of(), own ? W.scratch_main_bytes : scratch_sz, e->gptr(2 * i + 1)));
                return MI_OK;
            }();
            if (own) mi_tapwgrad_defer_pause(0);
            CK(rcw);
            if (i == 1 && late_dense && !use_third && tail_fuse_on && !tail_defer) { mi_small_reduce_defer(1); mi_small_reduce_bind(st); tail_defer = true; }      // from here to the end of the pass every small slab sum on st is one job of the fused launch
            // MI355_DENSE_EARLY=1 (round 4 A/B): the latent layers' filter gradients in FRONT of the encoder-head kernel instead of behind it: behind it they only
            // start when conv2's filter gradient on the other stream releases its compute units (64 KB of LDS next to 147 KB: no co-residence) and end the pass late
            static int dense_early = -1;
            if (dense_early < 0) { const char* ev = getenv("MI355_DENSE_EARLY"); dense_early = (ev && ev[0] == '1') ? 1 : 0; }
            if (i == 1 && dense_early && late_dense && !use_third && tail_defer && !dense_done) {
                TOP(e, st, OP_DENSE1_WGRAD, dense_wgrad(st, e->at(W.z), e->at(W.gdec[0]), B, d.z_dim, g.flat, e->gptr(10), bias_fused ? e->gptr(11) : nullptr));
                if (!bias_fused) TOP(e, st, OP_DENSE1_BIAS, bias_grad(st, e->at(W.gdec[0]), B, g.flat, e->gptr(11)));
                TOP(e, st, OP_HEADS_WGRAD, dense_wgrad(st, e->at(W.act[4]), e->at(W.dheads), B, g.flat, 2 * d.z_dim, e->gptr(8), bias_fused ? e->gptr(9) : nullptr));
                if (!bias_fused) TOP(e, st, OP_HEADS_BIAS, bias_grad(st, e->at(W.dheads), B, 2 * d.z_dim, e->gptr(9)));
                dense_done = true;
            }
            if (i == 1 && d.dtype == MI_BF16 && e->bits1_ok && e->W.enc_slab_bytes > 0 && g.c[0] == 3 && g.c[1] == 32 && g.c[2] == 64) {
                // conv2's input gradient feeds nothing but conv1's filter gradient: both in one launch, the 99 MB tensor between them never exists (enchead_tile.hpp)
                int nblk = 0;
                TOP(e, st, OP_CONV_DGRAD + 1, mi_conv2d_head_bwd_fused(st, d.dtype, src, e->last_u8 ? 2 : 1, idx, B, d.ih, d.iw, gy, e->wptr(2), e->at(W.bits_act1),
                                                                      e->gptr(0), e->gptr(1), e->at(W.enc_slabs), W.enc_slab_bytes, &nblk));
                if (nblk > 0) { enc_fused = true; continue; }
            }
            if (i > 1) arm();
            if (i == 2 && e->ares_ok && e->ares_mid) {       // conv3's input gradient: the mid-layer gather form, ReluGrad mask = conv2's output
                int launched = 0;
                TOP(e, st, OP_CONV_DGRAD + i, mi_ares_conv(st, d.dtype, 2, gy, B, e->at(W.wfrag[4]), nullptr, 0, e->at(W.act[2]), e->at(W.gact[2]), &launched));
                if (launched) { armed_ok(); continue; }
            }
            if (i == 3 && e->ares_ok) {                      // conv4's input gradient: gather form on the activation-resident kernel, ReluGrad mask = conv3's output
                int launched = 0;
                TOP(e, st, OP_CONV_DGRAD + i, mi_ares_conv(st, d.dtype, 1, gy, B, e->at(W.wfrag[1]), nullptr, 0, e->at(W.act[3]), e->at(W.gact[3]), &launched));
                if (launched) { armed_ok(); continue; }
            }
            if (i > 0)                                       // conv1's input gradient is never used
                TOP(e, st, OP_CONV_DGRAD + i, mi_conv2d_nhwc_dgrad_bits(st, d.dtype, gy, B, g.ih[i + 1], g.iw[i + 1], g.c[i + 1], e->wptr(2 * i), 4, 4, g.c[i], g.ih[i], g.iw[i],
                                        e->at(W.act[i]), (i == 1 && e->bits1_ok) ? e->at(W.bits_act1) : nullptr, e->at(W.gact[i])));
            if (i > 1) armed_ok();
        }
        static int dbg_skip_tail = -1;                       // MI355_DBG_SKIP_TAIL=1 (wrong results): how much of the step the small launches at the end of the caller's stream are
        if (dbg_skip_tail < 0) { const char* ev = getenv("MI355_DBG_SKIP_TAIL"); dbg_skip_tail = (ev && ev[0] == '1') ? 1 : 0; }
        if (late_dense && dbg_skip_tail) { if (defer) mi_tapwgrad_flush(sw); e->tail_nblk = 0; e->fin.pending = 0; }
        else
        if (late_dense && !use_third) {                      // the tails of both streams: dense1 + the decoder tail's slab sums here, the heads on the other one
            if (tail_fuse_on && !tail_defer) { mi_small_reduce_defer(1); mi_small_reduce_bind(st); tail_defer = true; }
            // round 6: dense1's and the heads' filter + bias gradients as ONE launch when both are issued here on the caller's stream (a fused encoder head in front): the two
            // ~190-block grids of a latency-bound kernel ran back to back on the step's critical tail (MI355_DENSE_PAIR=0: two launches; per-op timing keeps them apart)
            const bool pair = !dense_done && bias_fused && enc_fused && heads_main && e->tm.mode != 1;
            // ... and the one-block loss finalisation IN FRONT of that launch (MI355_FIN_EARLY=0: behind the slab sums): the pair's blocks wait for conv2's filter gradient to release
            // the CUs anyway, so the 4 us kernel and its boundary leave the serial tail (slab sums -> Adam)
            static int fin_early = -1;
            if (fin_early < 0) { const char* ev = getenv("MI355_FIN_EARLY"); fin_early = (ev && ev[0] == '0') ? 0 : 1; }
            // ... and the small slab sums that are READY (the fused encoder head's, the decoder tail's) as their own launch in front of both (MI355_TAIL_EARLY_FLUSH=1; default: one launch with the
            // pair's at the very end -- measured 0.7360 against 0.7380 ms): it runs while the pair waits for compute units, and the launch between the pair and Adam sums the pair's slabs only
            static int early_flush = -1;
            if (early_flush < 0) { const char* ev = getenv("MI355_TAIL_EARLY_FLUSH"); early_flush = (ev && ev[0] == '1') ? 1 : 0; }
            if (pair && early_flush && tail_defer) {
                if (e->tail_nblk > 0) { CK(mi_deconv2d_tail_reduce(st, e->at(W.tail_slabs), e->tail_nblk, e->gptr(18))); e->tail_nblk = 0; }
                CK(mi_small_reduce_flush(st));
            }
            if (pair && fin_early && e->fin.pending && part == 0) {
                e->fin.pending = 0;
                TOP(e, st, OP_FINALIZE, mi_vae_finalize_losses_flat(st, (const float*)e->at(W.partial), e->fin.nblk, (const float*)e->at(W.kl_row), e->fin.kl_floor, e->fin.B, e->fin.inv_batch,
                                               (float*)e->at(W.out2), e->fin.metrics3, e->fin.metric_weight, (const float*)e->at(W.bpart), e->fin.nblk, d.ct, e->fin.dbias));
            }
            long long nb0 = 0, nb1 = 0;
            void* ws0 = pair ? small_ws(st, mi_gemm_wgrad_scratch_bytes(d.dtype, B, d.z_dim, g.flat), &nb0) : nullptr;
            void* ws1 = pair ? small_ws(st, mi_gemm_wgrad_scratch_bytes(d.dtype, B, g.flat, 2 * d.z_dim), &nb1) : nullptr;
            // (both launches' slabs are live at once: two DISJOINT pieces of the tail scratch, which its bump allocator hands out while the slab sums are deferred)
            if (pair && ws0 && ws1 && ((char*)ws0 + nb0 <= (char*)ws1 || (char*)ws1 + nb1 <= (char*)ws0)) {
                TOP(e, st, OP_DENSE1_WGRAD, mi_gemm_wgrad_bias_pair_ws(st, d.dtype, e->at(W.z), e->at(W.gdec[0]), B, d.z_dim, g.flat, e->gptr(10), e->gptr(11), ws0, nb0,
                                                                       e->at(W.act[4]), e->at(W.dheads), B, g.flat, 2 * d.z_dim, e->gptr(8), e->gptr(9), ws1, nb1));
                dense_done = true; pair_used = true;
            }
            if (!dense_done) {
                if (!bias_fused) TOP(e, st, OP_DENSE1_BIAS, bias_grad(st, e->at(W.gdec[0]), B, g.flat, e->gptr(11)));
                TOP(e, st, OP_DENSE1_WGRAD, dense_wgrad(st, e->at(W.z), e->at(W.gdec[0]), B, d.z_dim, g.flat, e->gptr(10), e->gptr(11)));
            }
            if (e->tail_nblk > 0) { CK(mi_deconv2d_tail_reduce(st, e->at(W.tail_slabs), e->tail_nblk, e->gptr(18))); e->tail_nblk = 0; }
            if (defer) { mi_tapwgrad_flush(sw); }            // (the deferred slab reductions first: they end the other stream's real work; join() then finds the list empty)
            // the heads' gradients: behind a fused encoder head the caller's stream is the one that ends early (conv1's filter gradient is no longer a launch of its own)
            void* sh = (enc_fused && heads_main) ? st : sw;
            if (tail_defer && sh != st && !dense_done) { CK(mi_small_reduce_flush(st)); mi_small_reduce_defer(0); tail_defer = false; }      // (what follows is issued on the other stream: nothing of it may land in st's list)
            if (!dense_done) {
                if (!bias_fused) TOP(e, sh, OP_HEADS_BIAS, bias_grad(sh, e->at(W.dheads), B, 2 * d.z_dim, e->gptr(9)));
                TOP(e, sh, OP_HEADS_WGRAD, dense_wgrad(sh, e->at(W.act[4]), e->at(W.dheads), B, g.flat, 2 * d.z_dim, e->gptr(8), e->gptr(9)));
            }
            if (tail_defer) { CK(mi_small_reduce_flush(st)); mi_small_reduce_defer(0); tail_defer = false; }
        }
        if (use_third) { if (defer) mi_tapwgrad_flush(sw); hipStreamWaitEvent((hipStream_t)st, e->ev_third, 0); }
        if (e->fin.pending && (part == 0 || part == 2 || part == 4)) {      // the deferred loss scalars: on the caller's stream, in front of its wait for the other one
            e->fin.pending = 0;
            // Round 5: behind the filter-gradient queue's last reduce instead (it ends ~20 us before the caller's queue does: the one-block kernel and its boundary leave
            // the critical tail; everything it reads exists since the forward pass, what it writes -- the loss scalars, deconv4's bias gradient -- is read behind the join).
            // MI355_FIN_SIDE=0: on the caller's stream as before.  Round 6: with dense1's and the heads' filter gradients in ONE launch the caller's queue ends first again
            // (its last slab sum ~10 us before the other queue's): the kernel is back on the caller's stream unless MI355_FIN_SIDE=1 (0.7468 -> 0.7446, 0.7818 -> 0.7792 ms on two boxes).
            static int fin_side = -1;
            if (fin_side < 0) { const char* ev = getenv("MI355_FIN_SIDE"); fin_side = !ev ? 2 : ev[0] == '0' ? 0 : 1; }      // 2: by the tail's shape
            void* sf = (fork && (fin_side == 1 || (fin_side == 2 && !pair_used)) && part == 0) ? sw : st;
            TOP(e, sf, OP_FINALIZE, mi_vae_finalize_losses_flat(sf, (const float*)e->at(W.partial), e->fin.nblk, (const float*)e->at(W.kl_row), e->fin.kl_floor, e->fin.B, e->fin.inv_batch,
                                           (float*)e->at(W.out2), e->fin.metrics3, e->fin.metric_weight, (const float*)e->at(W.bpart), e->fin.nblk, d.ct, e->fin.dbias));
        }
        join();
    }
    return MI_OK;
}

// tf.train.AdamOptimizer step over all 22 reference variables at once; alpha = lr*sqrt(1-b2^t)/(1-b1^t) from the host.
static int apply_adam(VaeEngine* e, void* stream, float alpha, const float* alpha_dev, float beta1, float beta2, float epsilon) {
    if (!e->grads || !e->m || !e->v) return mi_fail(MI_ERR_STATE, "mi_vae_apply_adam: engine created without optimiser buffers");
    // fp32 / bf16 storage: the optimiser launch writes the K-contiguous kernel copies as well (mi_adam_tf_layouts, round 4: one launch and one pass over the master
    // weights less per step; same arithmetic, bit-identical parameters).  MI355_ADAM_LAYOUTS=0: Adam, then the transpose launch (split storage always).
    static int layouts_on = -1;
    if (layouts_on < 0) { const char* ev = getenv("MI355_ADAM_LAYOUTS"); layouts_on = (ev && ev[0] == '0') ? 0 : 1; }
    if (layouts_on && e->d.dtype != MI_BF16X3) {
        long long off[10]; int K[10], N[10];
        const int n = kernel_table(e, off, K, N);
        // round 5: the fragment-ordered copies of conv4 / deconv1 (/ conv3 / deconv2) for the activation-resident kernels come out of the same launch (MI355_ADAM_FRAG=0: the
        // separate ares_pack launch behind it, 5.6 us + a boundary at the head of the next step)
        static int frag_on = -1;
        if (frag_on < 0) { const char* ev = getenv("MI355_ADAM_FRAG"); frag_on = (ev && ev[0] == '0') ? 0 : 1; }
        void* fp[20]; int ff[20];
        for (int i = 0; i < 20; ++i) { fp[i] = nullptr; ff[i] = -1; }
        bool mid = false;
        const bool frag = frag_on && ares_eligible(e, &mid) && n == 10;
        if (frag) {      // kernel table: 0-3 conv1-4, 4 heads, 5 dense1, 6-9 deconv1-4
            fp[2 * 3] = e->at(e->W.wfrag[0]); ff[2 * 3] = 0; fp[2 * 3 + 1] = e->at(e->W.wfrag[1]); ff[2 * 3 + 1] = 1;      // conv4: forward (conv form), input gradient (gather form)
            fp[2 * 6] = e->at(e->W.wfrag[2]); ff[2 * 6] = 1; fp[2 * 6 + 1] = e->at(e->W.wfrag[3]); ff[2 * 6 + 1] = 0;      // deconv1: forward (gather form), input gradient (conv form)
            if (mid) { fp[2 * 2] = e->at(e->W.wfrag[4]); ff[2 * 2] = 2; fp[2 * 7] = e->at(e->W.wfrag[5]); ff[2 * 7] = 2; }  // conv3 input gradient, deconv2 forward
            if (mid) { fp[2 * 2 + 1] = e->at(e->W.wfrag[6]); ff[2 * 2 + 1] = 3; fp[2 * 7 + 1] = e->at(e->W.wfrag[7]); ff[2 * 7 + 1] = 3; }      // (round 6) conv3 forward, deconv2 input gradient: conv form of the register-weight kernel
            if (rc_small_frag_ok(e)) { fp[2 * 8] = e->at(e->W.wfrag[8]); ff[2 * 8] = 4; fp[2 * 1] = e->at(e->W.wfrag[9]); ff[2 * 1] = 5; fp[2 * 8 + 1] = e->at(e->W.wfrag[10]); ff[2 * 8 + 1] = 6; }      // (round 6) deconv3 forward (gather form), conv2 inside the fused encoder head
        }
        TOP(e, stream, OP_ADAM, mi_adam_tf_layouts_frag(stream, e->d.dtype, e->params, e->m, e->v, e->grads, e->L.total, off, K, N, nullptr, n, alpha, alpha_dev, beta1, beta2, epsilon,
                                                        e->d.dtype == MI_BF16 ? e->shadow : nullptr, e->wt, 1, frag ? fp : nullptr, frag ? ff : nullptr));
        return refresh_transposed(e, stream, true, frag);
    }
    TOP(e, stream, OP_ADAM, mi_adam_tf_flat_shadow(stream, e->params, e->m, e->v, e->grads, e->L.total, alpha, alpha_dev, beta1, beta2, epsilon,
                                                   e->d.dtype != MI_F32 ? e->shadow : nullptr, e->d.dtype == MI_BF16X3 ? MI_BF16X3 : MI_BF16, 1));
    return refresh_transposed(e, stream);
}

int mi_vae_apply_adam(void* h, void* stream, float alpha, float beta1, float beta2, float epsilon) {
    VaeEngine* e = (VaeEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "vae engine: null handle");
    return apply_adam(e, stream, alpha, nullptr, beta1, beta2, epsilon);
}

// Seeds the engine's own noise source (Philox stream `seed`, element 0 next): used whenever a sampling pass is given eps == NULL.
int mi_vae_set_seed(void* h, unsigned long long seed) {
    VaeEngine* e = (VaeEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "vae engine: null handle");
    const unsigned long long st[4] = {seed, 0ull, 0ull, 0ull};
    if (hipMemcpy(e->at(e->W.rng), st, sizeof(st), hipMemcpyHostToDevice) != hipSuccess) return mi_fail(MI_ERR_STATE, "mi_vae_set_seed: copy failed");
    e->rng_ready = 1;
    return MI_OK;
}


// One whole SGD step (the reference's sess.run([train_step, ...]), vae/models.py:213-216): forward + ELBO, backward, TF-Adam, in ONE call; nothing synchronises the
// host.  Single-rank path: the data-parallel host drives mi_vae_forward / _backward(parts) / _apply_adam around its bucket all-reduces.
// (Rounds 2-3 could also capture this launch sequence into a hipGraph and replay it; replay measured slower than the eager launches every time -- 1.26 vs 1.19 ms in
//  round 2, 0.991 vs 0.878 ms in round 3: ROCm 7.2 executes the captured fork / join of the two backward streams without their overlap, and the host is nowhere near
//  launch-bound at ~36 launches per 0.9 ms -- so round 4 removed the path instead of shipping a flag that loses 13 %: DESIGN finding 15.)
int mi_vae_train_step(void* h, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps,
                      float alpha, float beta1, float beta2, float epsilon, float* metrics3, float metric_weight) {
    VaeEngine* e = (VaeEngine*)h;
    CK(check_batch(e, B));
    struct FinGuard { VaeEngine* e; ~FinGuard() { e->defer_fin = 0; } } fin_guard{e};
    static int defer_on = -1;                           // MI355_DEFER=0: loss finalisation and the tail's slab reduce right behind the forward pass (A/B runs)
    if (defer_on < 0) { const char* ev = getenv("MI355_DEFER"); defer_on = (ev && ev[0] == '0') ? 0 : 1; }
    e->defer_fin = defer_on;                            // forward + backward are issued together here: the loss scalars are finalised inside the backward pass
    CK(mi_vae_forward(h, stream, src, tgt, frames_u8, idx, B, inv_batch, eps, 1, 1, metrics3, metric_weight));
    CK(mi_vae_backward(h, stream, src, idx, eps, inv_batch, 0));
    return apply_adam(e, stream, alpha, nullptr, beta1, beta2, epsilon);
}

// Gradient buckets of the data-parallel step in the order the backward pass completes them: 3 x {engine part, first float, one past the last float} of the flat
// gradient buffer -- decoder (part 1: dense1 .. deconv4, 43 % of the parameters) | heads + conv4 (part 3, 51 %) | conv3 .. conv1 (part 4, 6 %).  Pure function of the
// descriptor (no GPU); the host mirror (mi355/vae_device.py grad_buckets) and mi_vae_train_step_dp both follow it.
int mi_vae_dp_buckets(const MiVaeDesc* d, long long* out9) {
    VaeEngine e;
    if (!d || !out9 || !init_engine(e, d)) return mi_fail(MI_ERR_SHAPE, "mi_vae_dp_buckets: unsupported geometry or missing output");
    const long long dec = e.L.off[10], c4 = e.L.off[6];
    const long long b[9] = {1, dec, e.L.total, 3, c4, dec, 4, 0, c4};
    for (int i = 0; i < 9; ++i) out9[i] = b[i];
    return MI_OK;
}

// One whole DATA-PARALLEL SGD step in ONE call (round 5; SURVEY 8e, north_star "RCCL all-reduce of gradients over xGMI"): forward + ELBO of this
// rank's rows (inv_batch = 1 / B_global), the backward pass in the three parts of mi_vae_dp_buckets with each finished bucket's all-reduce queued on the communicator's
// own stream (mi_allreduce_sum_f32_async: it runs under the next part), mi_comm_wait, TF-Adam -- the sequence vae/models.py's host loop used to issue as three
// backward calls + three Python-side collectives per step.  comm: a communicator of mi_comm_init (or a recording one).  Nothing synchronises the host.
int mi_vae_train_step_dp(void* h, void* comm, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps,
                         float alpha, float beta1, float beta2, float epsilon, float* metrics3, float metric_weight) {
    VaeEngine* e = (VaeEngine*)h;
    CK(check_batch(e, B));
    if (!comm) return mi_fail(MI_ERR_ARG, "mi_vae_train_step_dp: null communicator (single rank: mi_vae_train_step)");
    if (!e->grads) return mi_fail(MI_ERR_STATE, "mi_vae_train_step_dp: engine created without a gradient buffer");
    long long bk[9];
    CK(mi_vae_dp_buckets(&e->d, bk));
    CK(mi_vae_forward(h, stream, src, tgt, frames_u8, idx, B, inv_batch, eps, 1, 1, metrics3, metric_weight));
    // A LOCAL failure between two buckets must not leave all-reduces queued with nothing joined behind them: whatever was queued is joined to `stream` before
    // the error is returned, so the caller's buffers are quiescent once `stream` drains.  The collective schedule of THIS rank is broken all the same -- its peers are
    // inside (or about to enter) collectives it will never issue: a non-OK return is FATAL for the job (the host mirror raises; the launcher tears every rank down).
    // Round 6: the first two parts do NOT make the caller's stream wait for the filter-gradient stream (that join idled the input-gradient chain ~50 us per part: the
    // data-parallel step without its collectives ran 0.863 ms against the single-rank step's 0.744).  Nothing the later parts issue on the caller's stream reads what the
    // filter-gradient stream produces; the part's bucket is complete when THAT stream has also seen the caller's work of the part, so the all-reduce is chained to it.
    // MI355_DP_OPEN_JOIN=0: the joins of round 5.
    static int open_join = -1;
    if (open_join < 0) { const char* ev = getenv("MI355_DP_OPEN_JOIN"); open_join = (ev && ev[0] == '0') ? 0 : 1; }
    int rc = MI_OK;
    for (int i = 0; i < 3 && rc == MI_OK; ++i) {
        e->dp_open_join = (open_join && i < 2) ? 1 : 0; e->dp_side_ready = 0;
        rc = mi_vae_backward(h, stream, src, idx, eps, inv_batch, (int)bk[3 * i]);
        void* producer = (e->dp_side_ready && e->side_ok == 1) ? (void*)e->side : stream;
        e->dp_open_join = 0; e->dp_side_ready = 0;
        if (rc == MI_OK) rc = mi_allreduce_sum_f32_async(comm, producer, e->grads + bk[3 * i + 1], bk[3 * i + 2] - bk[3 * i + 1]);
    }
    const int rcw = mi_comm_wait(comm, stream);
    if (rc != MI_OK) return rc;
    CK(rcw);
    return apply_adam(e, stream, alpha, nullptr, beta1, beta2, epsilon);
}

// One environment step of the rollout loop in ONE call (callers vae_common.py:45-61, train.py:142, run_eval.py:54):
//   frame_u8 [IH,IW,3] raw camera bytes (device) -> /255 -> conv x 4 -> mean z -> state = [z, measurements] -> PPO.predict
//   out (device, num_actions + 1 + z_dim floats) = [action | value | z]; noise [num_actions] (device) for sampling, NULL with greedy.
// Exact fp32 on the fp32 master weights whatever the engine's storage type; 8 launches (rollout.hip), no host synchronisation inside.
// `out` may be device memory or pinned (device-mapped) host memory: the head kernel stores its 67 floats straight into it.
int mi_rollout_step(void* vae_h, void* ppo_h, void* stream, const unsigned char* frame_u8, const float* measurements, int n_meas, const float* noise, int greedy, float* out) {
    VaeEngine* e = (VaeEngine*)vae_h;
    if (!e || !ppo_h) return mi_fail(MI_ERR_STATE, "mi_rollout_step: null handle");
    if (!frame_u8 || !out || (n_meas > 0 && !measurements) || (!greedy && !noise)) return mi_fail(MI_ERR_ARG, "mi_rollout_step: missing buffers");
    const MiVaeDesc& d = e->d; const Geom& g = e->g;
    hipStream_t st = (hipStream_t)stream;
    float* roll = (float*)e->at(e->W.roll);
    float* act[NCONV + 1]; act[0] = nullptr;
    long long o = 0;
    for (int i = 1; i <= NCONV; ++i) { act[i] = roll + o; o += (long long)g.ih[i] * g.iw[i] * g.c[i]; }
    float* mean_raw = roll + o;
    mi::PpoFusedParams q;
    CK(mi_ppo_internal_fill(ppo_h, &q, mean_raw, 1));
    if (q.din != d.z_dim + n_meas) return mi_fail(MI_ERR_SHAPE, "mi_rollout_step: z_dim + measurements must equal the policy's input size");
    // the raw-sum buffers of the split-K layers start at zero: conv1 (which writes its own output directly) clears them in the same launch
    MiZeroList zl = {};
    zl.p[0] = act[2]; zl.n[0] = (mean_raw + d.z_dim) - act[2];
    zl.p[1] = q.h1; zl.n[1] = 2LL * q.H1;
    zl.p[2] = q.h2; zl.n[2] = 2LL * q.H2;
    struct MiRolloutConv { const float* x; const float* x_bias; int IH, IW, C; const float* w; int ldw, N, KH, KW; float* out_raw; int flat_k; } cv[4];
    for (int i = 1; i < NCONV; ++i)                      // conv(i+1): input act[i] (conv2 reads conv1's finished output, the others raw sums + bias + ReLU on load)
        cv[i - 1] = MiRolloutConv{act[i], i == 1 ? nullptr : e->bptr(2 * (i - 1) + 1), g.ih[i], g.iw[i], g.c[i], e->params + e->L.off[2 * i], g.c[i + 1], g.c[i + 1], 4, 4, act[i + 1], 0};
    // mean head: the first z_dim columns of the fused [flat, 2 z] kernel; input = relu(conv4 raw + bias) flattened in (H, W, C) order
    cv[3] = MiRolloutConv{act[NCONV], e->bptr(2 * (NCONV - 1) + 1), 1, 1, g.c[NCONV], e->params + e->L.off[8], 2 * d.z_dim, d.z_dim, 1, 1, mean_raw, g.flat};
    CK(mi_rollout_conv1(st, frame_u8, e->params + e->L.off[0], e->bptr(1), act[1], g.ih[0], g.iw[0], g.c[0], 4, 4, g.c[1], &zl));
    for (int i = 0; i < 4; ++i)
        CK(mi_rollout_conv(st, cv[i].x, cv[i].x_bias, cv[i].IH, cv[i].IW, cv[i].C, cv[i].w, cv[i].ldw, cv[i].N, cv[i].KH, cv[i].KW, cv[i].out_raw, cv[i].flat_k));
    return mi_rollout_policy(st, q, mean_raw, e->bptr(9), d.z_dim, measurements, noise, greedy, out);
}

// VAE.encode (vae/models.py:199-202): frames -> mean [B,Z] fp32
int mi_vae_encode(void* h, void* stream, const void* src, int frames_u8, const int* idx, int B, float* mean_out) {
    VaeEngine* e = (VaeEngine*)h;
    CK(check_batch(e, B));
    CK(run_encoder(e, stream, src, frames_u8, idx, B, nullptr, 0));
    if (mean_out && hipMemcpyAsync(mean_out, e->at(e->W.mean), (size_t)B * e->d.z_dim * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
        return mi_fail(MI_ERR_LAUNCH, "mi_vae_encode: copy failed");
    return MI_OK;
}

// VAE.generate_from_latent (vae/models.py:188-191): z fp32 [B,Z] fed in place of the sample -> sigmoid(logits) [B,P] fp32
int mi_vae_decode(void* h, void* stream, const float* z, int B, float* recon_out) {
    VaeEngine* e = (VaeEngine*)h;
    CK(check_batch(e, B));
    const MiVaeDesc& d = e->d; const Geom& g = e->g;
    const long long n = (long long)B * d.z_dim;
    if (d.dtype == MI_F32) {
        if (hipMemcpyAsync(e->at(e->W.z), z, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return mi_fail(MI_ERR_LAUNCH, "mi_vae_decode: copy failed");
    } else if (d.dtype == MI_BF16X3) {
        CK(mi_cast_f32_to_split(stream, z, e->at(e->W.z), n));
    } else {
        CK(mi_cast_f32_to_bf16(stream, z, e->at(e->W.z), n));
    }
    CK(run_decoder(e, stream, B));
    return mi_sigmoid(stream, d.dtype, e->at(e->W.dec[4]), recon_out, (long long)B * g.dh[4] * g.dw[4] * g.dc[4]);
}

// VAE.reconstruct (vae/models.py:193-197): frames -> sigmoid(logits); samples z when sample=1 (training graph)
int mi_vae_reconstruct(void* h, void* stream, const void* src, int frames_u8, const int* idx, int B, const float* eps, int sample, float* recon_out) {
    VaeEngine* e = (VaeEngine*)h;
    CK(check_batch(e, B));
    CK(run_encoder(e, stream, src, frames_u8, idx, B, eps, sample));
    CK(run_decoder(e, stream, B));
    const Geom& g = e->g;
    return mi_sigmoid(stream, e->d.dtype, e->at(e->W.dec[4]), recon_out, (long long)B * g.dh[4] * g.dw[4] * g.dc[4]);
}

}  // extern "C"

// ---- per-op timing API (HIP events on the launch stream) ----
extern "C" {

int mi_vae_op_count(void) { return N_OPS; }
const char* mi_vae_op_name(int op) { return (op >= 0 && op < N_OPS) ? OP_NAMES[op] : ""; }

// mode 1: time every op of the following steps; mode 2: only `op_filter`.  max_records bounds the event pool.
int mi_vae_timing_begin(void* h, int mode, int op_filter, int max_records) {
    VaeEngine* e = (VaeEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "vae engine: null handle");
    if (e->tm.ev) return mi_fail(MI_ERR_STATE, "mi_vae_timing_begin: timing already active");
    if (mode < 1 || mode > 2 || max_records < 1) return mi_fail(MI_ERR_ARG, "mi_vae_timing_begin: mode 1|2, max_records >= 1");
    e->tm.ev = (hipEvent_t*)calloc(2 * (size_t)max_records, sizeof(hipEvent_t));
    e->tm.op = (int*)calloc((size_t)max_records, sizeof(int));
    if (!e->tm.ev || !e->tm.op) return mi_fail(MI_ERR_STATE, "mi_vae_timing_begin: out of host memory");
    for (int i = 0; i < 2 * max_records; ++i)
        if (hipEventCreate(&e->tm.ev[i]) != hipSuccess) return mi_fail(MI_ERR_STATE, "mi_vae_timing_begin: hipEventCreate failed");
    e->tm.cap = max_records; e->tm.n = 0; e->tm.op_filter = op_filter; e->tm.mode = mode;
    return MI_OK;
}

// stops timing, waits for the recorded events and accumulates elapsed milliseconds / launch counts per op id
int mi_vae_timing_collect(void* h, float* ms_sum, int* count, int n_ops) {
    VaeEngine* e = (VaeEngine*)h;
    if (!e) return mi_fail(MI_ERR_STATE, "vae engine: null handle");
    if (!e->tm.ev) return mi_fail(MI_ERR_STATE, "mi_vae_timing_collect: timing not active");
    if (n_ops != N_OPS) return mi_fail(MI_ERR_ARG, "mi_vae_timing_collect: n_ops mismatch");
    e->tm.mode = 0;
    for (int i = 0; i < N_OPS; ++i) { ms_sum[i] = 0.f; count[i] = 0; }
    int rc = MI_OK;
    for (int r = 0; r < e->tm.n; ++r) {
        float ms = 0.f;
        if (hipEventSynchronize(e->tm.ev[2 * r + 1]) != hipSuccess || hipEventElapsedTime(&ms, e->tm.ev[2 * r], e->tm.ev[2 * r + 1]) != hipSuccess) {
            rc = mi_fail(MI_ERR_STATE, "mi_vae_timing_collect: event query failed");
            break;
        }
        ms_sum[e->tm.op[r]] += ms; count[e->tm.op[r]] += 1;
    }
    for (int i = 0; i < 2 * e->tm.cap; ++i) hipEventDestroy(e->tm.ev[i]);
    free(e->tm.ev); free(e->tm.op);
    e->tm.ev = nullptr; e->tm.op = nullptr; e->tm.cap = e->tm.n = 0;
    return rc;
}

}  // extern "C"
