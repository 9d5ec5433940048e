// dectail_tile.hpp — the decoder tail of a training step as ONE kernel (round 3): deconv4 forward (tf.layers.conv2d_transpose 32 -> 3 channels, k4 s2,
// vae/models.py:264), the reconstruction loss of vae/models.py:11-22,123-128 on its logits, and BOTH gradients of that layer that the backward pass
// needs -- its input gradient (Conv2D of dlogits, ReluGrad-masked: the gradient of deconv3's output) and its filter gradient (Conv2DBackpropFilter) --
// without the logits or dlogits ever leaving the chip.
//
// Unfused (round 2) the three launches move: deconv4.fwd + loss  reads dec3 (101 MB at batch 512) + labels, writes dlogits (39 MB);
//                                            deconv4.dgrad       reads dlogits (2x through L1) + ReLU bit words, writes g_dec3 (101 MB);
//                                            deconv4.wgrad       reads dec3 (101 MB) AGAIN + dlogits (39 MB).
// Here dec3 is read once, g_dec3 written once, the labels read once: ~220 MB instead of ~480 MB, one launch instead of three (the two gradient
// launches were the head of both backward streams).
//
// Geometry: output pixel (oy, ox) = (2 gy + ph, 2 gx + pw) of "slot" (gy, gx) in a (IH + 1) x (IW + 1) slot grid, built from the input pixels
// (gy - 1 + ta, gx - 1 + tb), ta, tb in {0, 1}, through kernel rows kh = ph + 2 (1 - ta) (gather form, as gather_narrow_kernel).  The input gradient of
// pixel (y, x) is a 4 x 4 x 3 patch of dlogits at output rows 2 y .. 2 y + 3 = slots (y .. y + 1, x .. x + 1).  A block owns a TY x TX tile of pixels AND
// of slots with the same origin: it stages the (TY + 2) x (TX + 2) input pixels around it, computes the (TY + 1) x (TX + 1) slots its pixels' patches
// touch (the extra row / column is recomputed by the neighbour that OWNS it: only owned slots enter the loss and the bias gradient), keeps their
// dlogits in LDS as a [2 TY + 2][2 TX + 2][3] tile, and then runs the two gradient contractions on that tile.  All matrix products on
// v_mfma_f32_32x32x16_bf16; the summation orders of the logits and of the input gradient are those of gather_narrow_kernel / narrow_conv48_kernel, so
// both are bit-identical to the unfused path (tests/test_ops_gpu.py::test_decoder_tail_fused_equals_the_three_ops).
#pragma once
#include "narrow_tile.hpp"

namespace mi {

constexpr int DT_TY = 8, DT_TX = 16;                     // owned tile (pixels and slots)
constexpr int DT_PR = DT_TY + 2, DT_PC = DT_TX + 2;       // staged input pixels: 10 x 18
constexpr int DT_NPIX = DT_PR * DT_PC;                    // 180
constexpr int DT_SY = DT_TY + 1, DT_SX = DT_TX + 1;       // computed slots: 9 x 17
constexpr int DT_NSLOT = DT_SY * DT_SX;                   // 153
constexpr int DT_DLC = 2 * DT_TX + 2;                     // dlogits tile: 18 rows x 34 pixels x 3 channels, bf16
constexpr int DT_DLPITCH = DT_DLC * 6;                    // 204 bytes per row (4-byte aligned)
constexpr int DT_DLBYTES = (2 * DT_TY + 2) * DT_DLPITCH;  // 3672
constexpr int DT_XS = DT_NPIX * 64;                       // 11520: staged input pixels, 64 bytes each (32 bf16 channels), chunk-swizzled
constexpr int DT_PT = 32 * 128;                           // per wave: transposed-read tile of its 32 patches, 64 columns (48 used)
constexpr int DT_SLAB = 48 * 32;                          // floats of dW per block

struct DecTailParams {
    const bf16_t* x; int B, IH, IW;                      // deconv3's output [B, IH, IW, 32] (post-ReLU)
    const bf16_t* w;                                     // deconv4 kernel [4][4][3][32]  (kh, kw, out, in)
    const bf16_t* wt;                                    // the same K-contiguous: [32][48], k = (kh * 4 + kw) * 3 + out
    const float* bias;                                   // [3]
    const void* labels; int lab_u8; const int* lab_idx; long long lab_stride;   // target frames [*, OH * OW * 3] fp32 (or raw camera bytes)
    int loss_kind; float inv_b;
    bf16_t* dx;                                          // gradient of the loss wrt deconv3's PRE-activation [B, IH, IW, 32]
    float* slabs;                                        // [gridDim.x][DT_SLAB] partial filter gradients (reduce_slabs_kernel adds them to dW)
    float* lpart; float* bpart;                          // per block: loss partial sum; 4 floats of per-channel dlogits sums
    int OH, OW, GH, GW, tiles_x, tiles_per_frame, ntiles;
    FastDiv div_tpf, div_tx;
};

__device__ __forceinline__ int dt_swz(int q) { return (q >> 2) & 3; }      // 64-byte rows read by 32 consecutive rows (gn_swz<4>)

template <bool FASTBCE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void dectail_kernel(const DecTailParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[DT_XS + 4 * DT_PT + 3840 + 4 * 12 * 64];
    unsigned char* const xs = lds;                        // staged input pixels
    unsigned char* const pt = lds + DT_XS;                // 4 wave-private patch tiles (the cross-wave reduction at the very end reuses them)
    unsigned char* const dl = lds + DT_XS + 4 * DT_PT;    // dlogits tile (+ 16 floats of block reduction behind it)
    static_assert(DT_DLBYTES + 16 + 64 <= 3840, "dlogits tile + reduction words");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lgrp = lane >> 5;
    typedef __attribute__((address_space(3))) s16x4* lds_v4;

    // ---- per block, once: the forward weights into LDS as [tap][row ne = class * 3 + channel (12 live rows)][32 channels] (3 KB: in registers they
    // cost 32 VGPRs, which at three waves per SIMD spilled); the input-gradient weights (12 VGPRs) stay in registers ----
    unsigned char* const wl = dl + 3840;
    if (tid < 4 * 12 * 4) {
        const int tap = tid / 48, rem = tid - tap * 48, ne = rem >> 2, ch = rem & 3;
        const int cls = ne / 3, n = ne - cls * 3;
        const int kh = (cls >> 1) + 2 * (1 - (tap >> 1)), kw = (cls & 1) + 2 * (1 - (tap & 1));
        *(f32x4*)(wl + (tap * 12 + ne) * 64 + ch * 16) = *(const f32x4*)(p.w + ((kh * 4 + kw) * 3 + n) * 32 + ch * 8);
    }
    const bool wrow_ok = lrow < 12;
    const int wrow = wrow_ok ? lrow : 0;
    // input gradient: MFMA step s, row ci = lrow, this half-wave's 8 k (narrow_conv48_kernel's layout)
    u16x8 wtf[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) wtf[s] = *(const u16x8*)(p.wt + lrow * 48 + s * 16 + lgrp * 8);
    const float bias0 = p.bias ? p.bias[0] : 0.f, bias1 = p.bias ? p.bias[1] : 0.f, bias2 = p.bias ? p.bias[2] : 0.f;

    // patch tile columns 48 .. 63 stay zero for the whole kernel (physical chunk = logical chunk ^ (row & 7))
    unsigned char* const ptw = pt + wave * DT_PT;
    *(f32x4*)(ptw + lrow * 128 + (((6 + lgrp) ^ (lrow & 7)) << 4)) = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x16 accw[2];                                       // this wave's share of dW: rows k = mt * 32 + .., columns ci
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[mt][r] = 0.f;
    float lsum = 0.f, gs0 = 0.f, gs1 = 0.f, gs2 = 0.f;

    // transpose-read lane roles (tr_fragment, wgrad_tile.hpp): 16-lane group g supplies rows (g >> 1) * 8 + (c >> 2) [+ 4], columns (g & 1) * 16 + (c & 3) * 4
    const int tg = lane >> 4, tc = lane & 15;
    const int trow = (tg >> 1) * 8 + (tc >> 2), tcol = (tg & 1) * 16 + (tc & 3) * 4;

    for (int tile = (int)blockIdx.x; tile < p.ntiles; tile += (int)gridDim.x) {
        uint32_t b, rem, ty, tx;
        p.div_tpf.divmod((uint32_t)tile, b, rem);
        p.div_tx.divmod(rem, ty, tx);
        const int y0 = (int)ty * DT_TY, x0 = (int)tx * DT_TX;
        const bf16_t* xb = p.x + (long long)b * p.IH * p.IW * 32;

        // ---- stage the (TY + 2) x (TX + 2) input pixels around the tile: 720 sixteen-byte chunks, zero outside the image ----
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int id = tid + 256 * i;
            if (id < DT_NPIX * 4) {
                const int q = id >> 2, pc = id & 3;       // tile pixel, PHYSICAL chunk
                const int r = q / DT_PC, c = q - r * DT_PC;
                const int y = y0 - 1 + r, x = x0 - 1 + c;
                const bool in = y >= 0 && y < p.IH && x >= 0 && x < p.IW;
                const int lc = pc ^ dt_swz(q);            // the logical chunk that lives there
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (in) v = *(const f32x4*)(xb + ((long long)y * p.IW + x) * 32 + lc * 8);
                *(f32x4*)(xs + q * 64 + pc * 16) = v;
            }
        }
        __syncthreads();

        // ---- phase 1: logits of the 9 x 17 slots, loss, dlogits into the LDS tile ----
        const long long fr = p.lab_idx ? (long long)p.lab_idx[b] : (long long)b;
        for (int g = wave; g * 32 < DT_NSLOT; g += 4) {
            const int sidx = min(g * 32 + lrow, DT_NSLOT - 1);
            const bool sv = g * 32 + lrow < DT_NSLOT;
            const int sy = sidx / DT_SX, sx = sidx - sy * DT_SX;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const int q = (sy + (tap >> 1)) * DT_PC + sx + (tap & 1);
                const int sw = dt_swz(q);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const u16x8 af = *(const u16x8*)(xs + q * 64 + (((2 * kk + lgrp) ^ sw) << 4));
                    u32x4 wv = *(const u32x4*)(wl + (tap * 12 + wrow) * 64 + (2 * kk + lgrp) * 16);      // rows >= 12 of the MFMA's A operand are zero
#pragma unroll
                    for (int e = 0; e < 4; ++e) wv[e] = wrow_ok ? wv[e] : 0u;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, af), acc, 0, 0, 0);
                }
            }
            // D rows: register r of half-wave h is row (r & 3) + 8 (r >> 2) + 4 h; rows 0 .. 11 = class * 3 + channel are live:
            // h = 0: registers 0..3 (rows 0..3) and 4..7 (rows 8..11); h = 1: registers 0..3 (rows 4..7)
            const bool slot_in = sv && y0 + sy < p.GH && x0 + sx < p.GW;
            const bool owned = slot_in && sy < DT_TY && sx < DT_TX;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = lgrp ? 4 + (i & 3) : (i < 4 ? i : i + 4);
                const bool act = lgrp == 0 || i < 4;
                const int cls = row / 3, n = row - cls * 3;
                const int ph = cls >> 1, pw = cls & 1;
                const float xraw = acc[i] + (n == 0 ? bias0 : (n == 1 ? bias1 : bias2));
                const bf16_t xq = f32_to_bf16(xraw);        // the loss reads the logits as they would have been STORED (same values as the unfused kernel)
                const float xv = bf16_to_f32(xq);
                const int oy = 2 * (y0 + sy) + ph, ox = 2 * (x0 + sx) + pw;
                const long long li = ((long long)oy * p.OW + ox) * 3 + n;
                float yv = 0.f;
                if (slot_in && act) {
                    if (p.lab_u8) yv = u8_to_unit_exact((float)((const unsigned char*)p.labels)[fr * p.lab_stride + li]);
                    else yv = ((const float*)p.labels)[fr * p.lab_stride + li];
                }
                float l, gr;
                if constexpr (FASTBCE) {                   // loss_kind 0 on the hardware transcendentals (gather_narrow_kernel, FASTBCE)
                    const float e = __builtin_amdgcn_exp2f(-fabsf(xv) * 1.44269504f);
                    const float s1 = 1.0f + e;
                    const float rr = __builtin_amdgcn_rcpf(s1);
                    const float sg = xv >= 0.f ? rr : e * rr;
                    l = fmaf(__builtin_amdgcn_logf(s1), 0.69314718f, fmaf(-xv, yv, fmaxf(xv, 0.f)));
                    gr = sg - yv;
                } else {
                    const float e = __expf(-fabsf(xv));
                    const float rr = __frcp_rn(1.0f + e);
                    const float sg = xv >= 0.f ? rr : e * rr;
                    if (p.loss_kind == 0) { l = fmaxf(xv, 0.f) - xv * yv + __logf(1.0f + e); gr = sg - yv; }
                    else if (p.loss_kind == 1) {
                        l = -(yv * __logf(1e-10f + sg) + (1.0f - yv) * __logf(1e-10f + 1.0f - sg));
                        gr = (-yv / (1e-10f + sg) + (1.0f - yv) / (1e-10f + 1.0f - sg)) * sg * (1.0f - sg);
                    } else { const float dd = yv - sg; l = dd * dd; gr = -2.0f * dd * sg * (1.0f - sg); }
                }
                const bf16_t gq = slot_in ? f32_to_bf16(gr * p.inv_b) : (bf16_t)0;
                if (owned && act) {
                    lsum += l;
                    const float gst = bf16_to_f32(gq);      // the bias gradient sums the STORED (rounded) values, like BiasAddGrad of dlogits
                    if (n == 0) gs0 += gst; else if (n == 1) gs1 += gst; else gs2 += gst;
                }
                if (sv && act) *(bf16_t*)(dl + (2 * sy + ph) * DT_DLPITCH + ((2 * sx + pw) * 3 + n) * 2) = gq;
            }
        }
        __syncthreads();

        // ---- phase 2: input gradient of this wave's 32 pixels: rows yl = 2 wave, 2 wave + 1, columns xl = 0 .. 15 ----
        const int yl = 2 * wave + (lrow >> 4), xl = lrow & 15;
        u16x8 xf[3];
#pragma unroll
        for (int j = 0; j < 6; ++j) {                      // group j = 2 s + gi: q4 = 4 s + gi (+ 2 for the upper half-wave) -> kernel row q4 / 3, values (q4 % 3) * 4 ..
            const int q4 = 4 * (j >> 1) + (j & 1) + 2 * lgrp;
            const int kh = q4 / 3, o4 = q4 - kh * 3;
            const unsigned char* a = dl + (2 * yl + kh) * DT_DLPITCH + xl * 12 + o4 * 8;
            uint32_t* dst = (uint32_t*)&xf[j >> 1] + 2 * (j & 1);
            dst[0] = *(const uint32_t*)a; dst[1] = *(const uint32_t*)(a + 4);
        }
        f32x16 acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 3; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wtf[s]), __builtin_bit_cast(bf16x8, xf[s]), acc2, 0, 0, 0);
        {
            uint32_t R[4][2];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float v[4] = {acc2[4 * qd], acc2[4 * qd + 1], acc2[4 * qd + 2], acc2[4 * qd + 3]};
                const PackN<uint32_t, 2> w2 = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(v));
                R[qd][0] = w2.v[0]; R[qd][1] = w2.v[1];
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {                  // half-wave exchange: lane (pixel, h) then owns channels 16 h .. 16 h + 15 in the order R0 R2 R1 R3
                auto r0 = __builtin_amdgcn_permlane32_swap(R[0][d], R[2][d], false, false); R[0][d] = r0[0]; R[2][d] = r0[1];
                auto r1 = __builtin_amdgcn_permlane32_swap(R[1][d], R[3][d], false, false); R[1][d] = r1[0]; R[3][d] = r1[1];
            }
            uint32_t o[8] = {R[0][0], R[0][1], R[2][0], R[2][1], R[1][0], R[1][1], R[3][0], R[3][1]};
            // ReluGrad: the staged activation row itself is the mask
            const int qm = (yl + 1) * DT_PC + xl + 1;
            const PackN<uint32_t, 4> m0 = *(const PackN<uint32_t, 4>*)(xs + qm * 64 + (((2 * lgrp) ^ dt_swz(qm)) << 4));
            const PackN<uint32_t, 4> m1 = *(const PackN<uint32_t, 4>*)(xs + qm * 64 + (((2 * lgrp + 1) ^ dt_swz(qm)) << 4));
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                uint32_t pos, nz;                          // per 16-bit half: x > 0 as a signed integer (negative floats are negative int16) -> 1, else 0
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(pos) : "v"(d < 4 ? m0.v[d] : m1.v[d - 4]));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(pos), "v"(0x00010001u));
                o[d] &= nz * 0xffffu;
            }
            const int y = y0 + yl, x = x0 + xl;
            if (y < p.IH && x < p.IW) {
                bf16_t* out = p.dx + (((long long)b * p.IH + y) * p.IW + x) * 32 + lgrp * 16;
                *(PackN<uint32_t, 4>*)out = PackN<uint32_t, 4>{{o[0], o[1], o[2], o[3]}};
                *(PackN<uint32_t, 4>*)(out + 8) = PackN<uint32_t, 4>{{o[4], o[5], o[6], o[7]}};
            }
        }

        // ---- phase 3: filter gradient  dW[k][ci] += sum over this wave's pixels of patch[pixel][k] * x[pixel][ci] ----
        // patches: this lane's 24 values -> row lrow of the wave's transposed-read tile; the activations are read transposed from the staged tile
#pragma unroll
        for (int s = 0; s < 3; ++s) *(u16x8*)(ptw + lrow * 128 + (((2 * s + lgrp) ^ (lrow & 7)) << 4)) = xf[s];
        __builtin_amdgcn_wave_barrier();                  // same wave, in-order LDS queue
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                   // k-step = the 16 pixels of tile row 2 wave + ks
            u16x8 bfr;
            {
                const int q0 = (2 * wave + ks + 1) * DT_PC + 1 + trow;      // staged-tile pixel of row trow (and trow + 4)
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(xs + q0 * 64 + (((tcol >> 3) ^ dt_swz(q0)) << 4) + (tcol & 7) * 2));
                const int q1 = q0 + 4;
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(xs + q1 * 64 + (((tcol >> 3) ^ dt_swz(q1)) << 4) + (tcol & 7) * 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) { bfr[e] = (unsigned short)lo[e]; bfr[4 + e] = (unsigned short)hi[e]; }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int col = mt * 32 + tcol;
                const int r0 = ks * 16 + trow, r1 = r0 + 4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(ptw + r0 * 128 + (((col >> 3) ^ (r0 & 7)) << 4) + (col & 7) * 2));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(ptw + r1 * 128 + (((col >> 3) ^ (r1 & 7)) << 4) + (col & 7) * 2));
                u16x8 afr;
#pragma unroll
                for (int e = 0; e < 4; ++e) { afr[e] = (unsigned short)lo[e]; afr[4 + e] = (unsigned short)hi[e]; }
                accw[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr), __builtin_bit_cast(bf16x8, bfr), accw[mt], 0, 0, 0);
            }
        }
        __syncthreads();                                  // every wave is done with the staged tiles before the next tile overwrites them
    }

    // ---- block totals: dW (4 waves take turns on one 8 KB buffer) -> this block's slab; loss / bias partials ----
    float* red = (float*)pt;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* q = &red[(mt * 16 + r) * 64 + lane];
                    *q = w == 0 ? accw[mt][r] : *q + accw[mt][r];
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < 2 * 16 * 64; i += 256) {
        const int mt = i >> 10, r = (i >> 6) & 15, l = i & 63;
        const int k = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = l & 31;
        if (k < 48) p.slabs[(long long)blockIdx.x * DT_SLAB + k * 32 + n] = red[i];
    }
    lsum = wave_sum(lsum); gs0 = wave_sum(gs0); gs1 = wave_sum(gs1); gs2 = wave_sum(gs2);
    float* red2 = (float*)(dl + DT_DLBYTES + ((16 - (DT_DLBYTES & 15)) & 15));
    if (lane == 0) { red2[wave * 4 + 0] = lsum; red2[wave * 4 + 1] = gs0; red2[wave * 4 + 2] = gs1; red2[wave * 4 + 3] = gs2; }
    __syncthreads();
    if (tid < 4) {
        const float t = (red2[tid] + red2[4 + tid]) + (red2[8 + tid] + red2[12 + tid]);
        if (tid == 0) p.lpart[blockIdx.x] = t; else p.bpart[(long long)blockIdx.x * 4 + tid - 1] = t;
    }
}

}  // namespace mi
