// ares.hip — host side of the activation-resident kernels (ares_tile.hpp) for the four small-grid layers of the ConvVAE: conv4 forward / deconv1 input
// gradient (conv form) and deconv1 forward / conv4 input gradient (gather form).  bf16 storage, the model's geometry only; anything else returns
// "not launched" and the caller takes the general kernels (gemm2 / tapconv).
#include <stdlib.h>
#include "ares_tile.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"

using namespace mi;

static bool ares_on() {                                     // MI355_ARES=0: the general tile kernels (A/B runs)
    static int on = -1;
    if (on < 0) { const char* e = getenv("MI355_ARES"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}

static int ares_cfg() {                                    // MI355_ARES_CFG (A/B runs): bit 0 conv form with 2 frames per block (two blocks per CU), bit 1 gather form with 8 (one block per CU)
    static int c = -1;
    if (c < 0) { const char* e = getenv("MI355_ARES_CFG"); c = e ? atoi(e) : 0; }
    return c;
}

extern "C" {

// bytes of one fragment-ordered weight copy (either form: 16 x 128 x 256 bf16)
long long mi_ares_weight_bytes(void) { return 16ll * 128 * 256 * 2; }

// fp32 master kernel -> fragment order (bf16).  form 0 (conv form): w is [kh][kw][128][256] -- conv4's HWIO kernel for its forward pass, or deconv1's
// [kh,kw,out = 128,in = 256] kernel for deconv1's INPUT gradient.  form 1 (gather form): w is [kh][kw][128][256] read as [kh][kw][n][c] -- deconv1's kernel
// for its forward pass, or conv4's HWIO kernel for conv4's INPUT gradient.  (vae/models.py:253,261 and their gradients behind :142)
int mi_ares_pack_weights(void* stream, int form, const float* w_fp32, void* wf_out) {
    // round 6: forms 3 ([1024][128] mid-layer kernel, conv form of the register-weight kernel), 4 (deconv3's [5][5][32][64] kernel, gather form of the register-weight kernel: 144 KB),
    // 5 (conv2's [512][64] kernel for the fused encoder head: 64 KB) -- the orders those kernels' prologues load their weight registers in
    if (!w_fp32 || !wf_out || form < 0 || form > 6) return mi_fail(MI_ERR_ARG, "mi_ares_pack_weights: bad arguments");
    if ((((uintptr_t)w_fp32) | ((uintptr_t)wf_out)) & 15) return mi_fail(MI_ERR_ARG, "mi_ares_pack_weights: buffers must be 16-byte aligned");
    AresPackJobs j = {};
    j.src[0] = w_fp32; j.dst[0] = (bf16_t*)wf_out; j.form[0] = form; j.n = 1;
    hipLaunchKernelGGL(ares_pack_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, j);
    return mi_check_launch("ares_pack_kernel");
}

// the four copies an engine keeps (conv4 forward, conv4 input gradient, deconv1 forward, deconv1 input gradient) in ONE launch: conv4_w / deconv1_w are the two
// [4][4][128][256] fp32 master kernels, wf_out[0..3] the four 1 MB destinations in that order
int mi_ares_pack_weights4(void* stream, const float* conv4_w, const float* deconv1_w, void* wf0, void* wf1, void* wf2, void* wf3) {
    return mi_ares_pack_weights6(stream, conv4_w, deconv1_w, nullptr, nullptr, wf0, wf1, wf2, wf3, nullptr, nullptr);
}

// ... plus (optional) the two mid-layer gather-form copies: conv3_w (HWIO [4][4][64][128]) -> wf4 for conv3's input gradient, deconv2_w ([kh,kw,out = 64,in = 128]) -> wf5
// for deconv2's forward pass (form 2, 256 KB each)
int mi_ares_pack_weights6(void* stream, const float* conv4_w, const float* deconv1_w, const float* conv3_w, const float* deconv2_w,
                          void* wf0, void* wf1, void* wf2, void* wf3, void* wf4, void* wf5) {
    return mi_ares_pack_weights8(stream, conv4_w, deconv1_w, conv3_w, deconv2_w, wf0, wf1, wf2, wf3, wf4, wf5, nullptr, nullptr);
}

// ... plus (optional, round 6) the two conv-form copies of the SAME mid-layer kernels for the register-weight kernel (form 3, 256 KB each): conv3_w -> wf6 for conv3's forward
// pass, deconv2_w -> wf7 for deconv2's input gradient -- the order rwconv_conv_kernel<4, 2>'s prologue loads its 64 weight fragments in, 1 KB contiguous per wave load
int mi_ares_pack_weights8(void* stream, const float* conv4_w, const float* deconv1_w, const float* conv3_w, const float* deconv2_w,
                          void* wf0, void* wf1, void* wf2, void* wf3, void* wf4, void* wf5, void* wf6, void* wf7) {
    if (!conv4_w || !deconv1_w || !wf0 || !wf1 || !wf2 || !wf3) return mi_fail(MI_ERR_ARG, "mi_ares_pack_weights: bad arguments");
    if ((((uintptr_t)conv4_w) | ((uintptr_t)deconv1_w) | ((uintptr_t)conv3_w) | ((uintptr_t)deconv2_w) | ((uintptr_t)wf0) | ((uintptr_t)wf1) | ((uintptr_t)wf2) | ((uintptr_t)wf3) |
         ((uintptr_t)wf4) | ((uintptr_t)wf5)) & 15) return mi_fail(MI_ERR_ARG, "mi_ares_pack_weights: buffers must be 16-byte aligned");
    AresPackJobs j = {};
    j.n = 4;
    j.src[0] = conv4_w; j.dst[0] = (bf16_t*)wf0; j.form[0] = 0;
    j.src[1] = conv4_w; j.dst[1] = (bf16_t*)wf1; j.form[1] = 1;
    j.src[2] = deconv1_w; j.dst[2] = (bf16_t*)wf2; j.form[2] = 1;
    j.src[3] = deconv1_w; j.dst[3] = (bf16_t*)wf3; j.form[3] = 0;
    if (conv3_w && wf4) { j.src[j.n] = conv3_w; j.dst[j.n] = (bf16_t*)wf4; j.form[j.n] = 2; ++j.n; }
    if (deconv2_w && wf5) { j.src[j.n] = deconv2_w; j.dst[j.n] = (bf16_t*)wf5; j.form[j.n] = 2; ++j.n; }
    if ((((uintptr_t)wf6) | ((uintptr_t)wf7)) & 15) return mi_fail(MI_ERR_ARG, "mi_ares_pack_weights: buffers must be 16-byte aligned");
    if (conv3_w && wf6) { j.src[j.n] = conv3_w; j.dst[j.n] = (bf16_t*)wf6; j.form[j.n] = 3; ++j.n; }
    if (deconv2_w && wf7) { j.src[j.n] = deconv2_w; j.dst[j.n] = (bf16_t*)wf7; j.form[j.n] = 3; ++j.n; }
    hipLaunchKernelGGL(ares_pack_kernel, dim3(256 * j.n), dim3(256), 0, (hipStream_t)stream, j);
    return mi_check_launch("ares_pack_kernel");
}

// form 0: x [B,8,18,128] -> out [B,3,8,256] (k4 s2 conv: out = relu?(conv(x) + bias), masked by `mask` [B,3,8,256] when given);
// form 1: x [B,3,8,256] -> out [B,8,18,128] (k4 s2 transposed conv in gather form, same epilogue, mask [B,8,18,128]).
// wf: the fragment-ordered weights of mi_ares_pack_weights(form).  *launched = 0: not eligible (nothing was launched; use the general ops).
int mi_ares_conv(void* stream, int dtype, int form, const void* x, int B, const void* wf, const float* bias, int relu, const void* mask, void* out, int* launched) {
    if (!launched) return mi_fail(MI_ERR_ARG, "mi_ares_conv: missing arguments");
    *launched = 0;
    if (!ares_on() || dtype != MI_BF16 || form < 0 || form > 2 || !x || !wf || !out || B < 1) return MI_OK;
    if ((((uintptr_t)x) | ((uintptr_t)wf) | ((uintptr_t)out) | ((uintptr_t)mask) | ((uintptr_t)bias)) & 15) return MI_OK;
    const long long xb = (long long)B * (form == 1 ? 3 * 8 * 256 : 8 * 18 * 128) * 2;      // the descriptor's range check supplies the zeros of a ragged last frame group
    if (xb >= (long long)G2_OOB) return MI_OK;
    AresParams p = {};
    p.x = x; p.x_bytes = (uint32_t)xb; p.wf = wf; p.B = B;
    p.out = out; p.bias = bias; p.mask = mask; p.relu = relu; p.out_f32 = 0;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("MI355_ARES_DBG"); dbg = e ? atoi(e) : 0; } p.dbg = dbg; }
    hipStream_t st = (hipStream_t)stream;
    if (form == 0) {
        p.M = B * 24; p.N = AC_N; p.OH = AC_OH; p.OW = AC_OW;
        const int F = ares_cfg() & 1 ? 2 : 4;               // frames per block (MI355_ARES_CFG bit 0: the two-blocks-per-CU form, measured slower)
        const int groups = (B + F - 1) / F;
        const int nb = (groups + 7) / 8 * 16;                // block b: frame group (b & 7) + 8 (b >> 4), column half (b >> 3) & 1
        if (F == 4) MI_LAUNCH((ares_conv_kernel<4, 1>), dim3(nb), dim3(256), 0, st, p);
        else MI_LAUNCH((ares_conv_kernel<2, 2>), dim3(nb), dim3(256), 0, st, p);
        const int rc = mi_check_launch("ares_conv_kernel");
        if (rc != MI_OK) return rc;
    } else if (form == 2) {
        static int mid_on = -1;                             // MI355_ARES_MID=0: the mid layers stay on the register-weight kernels (A/B runs)
        if (mid_on < 0) { const char* e = getenv("MI355_ARES_MID"); mid_on = (e && e[0] == '0') ? 0 : 1; }
        if (!mid_on) return MI_OK;
        p.M = B * G2_RPF; p.N = G2_N; p.OH = G2_OH; p.OW = G2_OW;
        for (int c = 0; c < 4; ++c) { p.dc_ohw[c] = make_fastdiv(G2_RPF); p.dc_ow[c] = make_fastdiv(19); }
        MI_LAUNCH(ares_gather2_kernel, dim3(B), dim3(256), 0, st, p);
        const int rc = mi_check_launch("ares_gather2_kernel");
        if (rc != MI_OK) return rc;
    } else {
        p.M = B * AG_RPF; p.N = AG_N; p.OH = AG_OH; p.OW = AG_OW;
        for (int c = 0; c < 4; ++c) { p.dc_ohw[c] = make_fastdiv(AG_RPF); p.dc_ow[c] = make_fastdiv(9); }
        const int F = ares_cfg() & 2 ? 8 : 4;               // (bit 1: the one-block-per-CU form of the gather kernel)
        const int groups = (B + F - 1) / F;
        const int nb = (groups + 7) / 8 * 32;                // block b: parity class (b >> 3) & 3, frame group (b & 7) + 8 (b >> 5)
        if (F == 8) MI_LAUNCH((ares_gather_kernel<8, 1>), dim3(nb), dim3(256), 0, st, p);
        else MI_LAUNCH((ares_gather_kernel<4, 2>), dim3(nb), dim3(256), 0, st, p);
        const int rc = mi_check_launch("ares_gather_kernel");
        if (rc != MI_OK) return rc;
    }
    *launched = 1;
    return MI_OK;
}

}  // extern "C"
