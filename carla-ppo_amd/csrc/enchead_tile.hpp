// enchead_tile.hpp — the encoder head of a backward pass as ONE kernel (round 3): conv2's input gradient (Conv2DBackpropInput of vae/models.py:251, ReluGrad-masked
// by conv1's activation: the gradient of conv1's output) and conv1's filter + bias gradient (Conv2DBackpropFilter / BiasAddGrad of vae/models.py:250), without that
// gradient ever leaving the chip.  conv1 has no input gradient (its input is the camera frame), so conv1's filter gradient is the ONLY consumer of that tensor.
//
// Unfused the two launches move: conv2.dgrad   reads dy2 (44 MB at batch 512) + ReLU bit words (6 MB), writes g1 (99 MB);
//                                conv1.wgrad   reads g1 (101 MB) + the frames (20 MB as camera bytes).
// Here: dy2 + bit words + frames in, 2 x 6 KB of partial sums per block out: ~70 MB instead of ~270 MB, one launch instead of two.
//
// Geometry (k = 4, s = 2, VALID, both layers).  g1[y, x, ci] = sum over (ta, tb) in {0, 1}^2, co of dy2[gy - ta, gx - tb, co] * W2[ph + 2 ta, pw + 2 tb, ci, co]
// with (ph, pw) = (y & 1, x & 1) the pixel's parity class and (gy, gx) = (y >> 1, x >> 1): K = 4 taps x 64 channels per pixel, and the 4 x 64 x 32 weights of ONE
// class fit a wave's registers (64 VGPRs, as in rwconv.hip).  A block owns an 8 x 16 tile of g1 pixels; its four waves are the four parity classes, 32 pixels each
// (rows y0 + ph + 2 i, columns x0 + pw + 2 j, lane = 8 i + j).  Per tile: the 5 x 9 dy2 pixels around it are staged in LDS (requested one tile ahead), 16 MFMAs per
// wave give g1 of its 32 pixels, the epilogue of narrow_conv48_kernel<., 1> (bf16 rounding, half-wave exchange, ReLU bit words) leaves every lane 16 channels of its
// pixel, which go to a wave-private LDS tile; the lane's half of its pixel's 48-value frame patch (the loader of narrow_conv48_kernel) goes to a second one with a
// constant 1.0 in column 48; both are read back transposed (ds_read_b64_tr_b16) for 4 MFMAs into two persistent 32 x 32 accumulators: rows 0 .. 47 = dW1, row 48 =
// the bias gradient.  v_mfma_f32_32x32x16_bf16 throughout; g1 is rounded to bf16 exactly where the unfused path stores it.
#pragma once
#include "wgrad_tile.hpp"

namespace mi {

constexpr int EH_TY = 8, EH_TX = 16;                      // owned tile of g1 pixels
constexpr int EH_SR = EH_TY / 2 + 1, EH_SC = EH_TX / 2 + 1;   // staged dy2 pixels: 5 x 9
constexpr int EH_NS = EH_SR * EH_SC;                      // 45
constexpr int EH_SP = 144;                                // their LDS pitch: 128 bytes (64 bf16 channels) + 16 (consecutive pixels start in different bank quads)
constexpr int EH_DYS = (EH_NS * EH_SP + 255) & ~255;      // 6656
constexpr int EH_GP = 80;                                 // per wave: g1 tile [32 pixels][32 channels] bf16 at an 80-byte pitch (transposed reads, cf. DT_XP)
constexpr int EH_GT = 32 * EH_GP;                         // 2560
constexpr int EH_SLAB = 64 * 32 + 32;                     // floats per block: dW1 rows padded to 64 x 32, then the 32 bias sums (= NW_SLAB of narrow_tile.hpp: same reduce)
constexpr int EH_PT = 32 * 128;                           // per wave: patch tile [32 pixels][64 columns] bf16, chunk ^ (row & 7)

struct EncHeadParams {
    const bf16_t* dy; unsigned dy_bytes; int B, OH, OW;  // conv2's output gradient [B, OH, OW, 64]
    const bf16_t* w;                                     // conv2 kernel [4][4][32][64] (kh, kw, in, out): the layer's own HWIO storage
    const uint32_t* bits; unsigned bits_bytes;           // ReLU bit words of conv1's output: 2 x uint32 per pixel (narrow_tile.hpp)
    const void* frames; const int* frame_idx; long long frame_stride;   // camera frames [*, FH, FW, 3] (uint8 or fp32), elements per frame
    int IH, IW, FW;                                      // g1 / conv1 output grid; frame width
    float* slabs;                                        // [gridDim.x][EH_SLAB] (reduce_slabs_kernel adds them up)
    int tiles_x, tiles_per_frame, ntiles;
    FastDiv div_tpf, div_tx;
};

template <typename TS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void enchead_bwd_kernel(const EncHeadParams p) {
    constexpr int SSZ = (int)sizeof(TS), GSZ = 4 * SSZ, GDW = GSZ / 4;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[EH_DYS + 4 * EH_PT + 4 * EH_GT + 4 * 4096];
    unsigned char* const dys = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lgrp = lane >> 5;
    unsigned char* const ptw = lds + EH_DYS + wave * EH_PT;             // this wave's patch tile (the final cross-wave reduction reuses the four of them)
    unsigned char* const gtw = lds + EH_DYS + 4 * EH_PT + wave * EH_GT; // this wave's g1 tile
    const int ph = wave >> 1, pw = wave & 1;              // parity class of this wave's pixels
    const int li = lrow >> 3, lj = lrow & 7;              // the lane's pixel inside the tile: (ph + 2 li, pw + 2 lj)
    typedef __attribute__((address_space(3))) s16x4* lds_v4;

    // ---- per wave, once: the class's weights: tap (ta, tb) -> kernel (ph + 2 ta, pw + 2 tb); MFMA row = ci = lrow, this half-wave's 8 co of k-step kk ----
    // taps 0 .. 2 in registers (48 VGPRs); the fourth tap's four fragments live in a wave-private 4 KB LDS tile, lane-linear (one conflict-free ds_read_b128 each): with all
    // 64 weight registers the kernel sat at the three-waves-per-SIMD limit with two fragments spilled, and their scratch reloads inside the tile loop drained every prefetch
    unsigned char* const wlw = lds + EH_DYS + 4 * EH_PT + 4 * EH_GT + wave * 4096;
    u16x8 wf[3][4];
    {
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 16 * 32 * 64 * 2, 0x00020000);
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            const int kh = ph + 2 * (tap >> 1), kw = pw + 2 * (tap & 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const u16x8 f = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, (((kh * 4 + kw) * 32 + lrow) * 64 + kk * 16 + lgrp * 8) * 2, 0, 0));
                if (tap < 3) wf[tap][kk] = f; else *(u16x8*)(wlw + kk * 1024 + lane * 16) = f;
            }
        }
    }
    // patch tile: column 48 = 1.0 in every row (the bias gradient rides along as row 48 of the filter gradient), columns 49 .. 63 = 0: chunks 6, 7 of the 128-byte row
    {
        const f32x4 one_then_zero = {__builtin_bit_cast(float, 0x00003F80u), 0.f, 0.f, 0.f};
        *(f32x4*)(ptw + lrow * 128 + (((6 + lgrp) ^ (lrow & 7)) << 4)) = lgrp ? f32x4{0.f, 0.f, 0.f, 0.f} : one_then_zero;
    }
    f32x16 accw[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) accw[mt][r] = 0.f;

    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.bits, 0, (int)p.bits_bytes, 0x00020000);
    constexpr int EH_OOB = 0x7ffffff0;

    // staging roles: 45 pixels x 8 sixteen-byte chunks = 360 items, two per thread (the second only for tid < 104); recomputed where they are used (held in
    // registers they were the difference between 168 VGPRs with two weight fragments spilled -- reloaded by scratch loads INSIDE the tile loop, whose s_waitcnt
    // vmcnt drained every prefetch in front of them -- and no spill)
    auto role = [&](int i, int& spix, int& schk, int& srow, int& scl, bool& sin) {
        const int id = tid + 256 * i; sin = id < EH_NS * 8; spix = sin ? id >> 3 : 0; schk = id & 7;
        srow = (spix * 57) >> 9; scl = spix - srow * EH_SC;            // spix / 9 for spix < 45
    };
    struct Staged { f32x4 d[2]; };
    auto tile_origin = [&](int tile, int& b, int& y0, int& x0) {
        uint32_t bb, rem, ty, tx;
        p.div_tpf.divmod((uint32_t)tile, bb, rem);
        p.div_tx.divmod(rem, ty, tx);
        b = (int)bb; y0 = (int)ty * EH_TY; x0 = (int)tx * EH_TX;
    };
    auto request = [&](int tile, Staged& R) {
        int b, y0, x0;
        tile_origin(tile, b, y0, x0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int spix, schk, srow, scl; bool sin;
            role(i, spix, schk, srow, scl, sin);
            const int oy = (y0 >> 1) - 1 + srow, ox = (x0 >> 1) - 1 + scl;
            const bool in = sin && (unsigned)oy < (unsigned)p.OH && (unsigned)ox < (unsigned)p.OW;
            const int off = in ? (((b * p.OH + oy) * p.OW + ox) * 64 + schk * 8) * 2 : EH_OOB;
            R.d[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsD, off, 0, 0));
        }
    };
    auto commit = [&](const Staged& R) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int spix, schk, srow, scl; bool sin;
            role(i, spix, schk, srow, scl, sin);
            if (sin) *(f32x4*)(dys + spix * EH_SP + schk * 16) = R.d[i];
        }
    };
    // frame patch of the lane's pixel: group j = 2 s + gi of this lane: q = 4 s + gi (+ 2 for the upper half-wave) -> kernel row q / 3, value offset (q % 3) * 4
    const uint32_t rowb = (uint32_t)(p.FW * 3 * SSZ);
    auto goff = [&](int j) -> uint32_t {                  // (a select between two wave-uniform values per use instead of six registers held across the kernel)
        const int qa = 4 * (j >> 1) + (j & 1), qb = qa + 2;
        const uint32_t oa = (uint32_t)(qa / 3) * rowb + (uint32_t)((qa % 3) * GSZ), ob = (uint32_t)(qb / 3) * rowb + (uint32_t)((qb % 3) * GSZ);
        return lgrp ? ob : oa;
    };
    struct Raw { uint32_t d[6][GDW]; uint32_t mw; };
    auto frame_of = [&](int b) -> int {                   // scalar load + its own wait (dectail_tile.hpp: a vector load of a uniform value drains the prefetch)
        int fr = b;
        if (p.frame_idx) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(fr) : "s"(p.frame_idx + b) : "memory");
        return fr;
    };
    auto request_patch = [&](int tile, Raw& r) {
        int b, y0, x0;
        tile_origin(tile, b, y0, x0);
        const int y = y0 + ph + 2 * li, x = x0 + pw + 2 * lj;
        // (a pixel outside the image has g1 == 0 -- all its taps fall outside dy2 -- so its patch only has to be readable: the coordinates are clamped)
        const int yc = min(y, p.IH - 1), xc = min(x, p.IW - 1);
        const unsigned char* pix = (const unsigned char*)p.frames + ((long long)frame_of(b) * p.frame_stride + (2ll * yc * p.FW + 2 * xc) * 3) * SSZ;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const PackU<uint32_t, GDW, 2> v = *(const PackU<uint32_t, GDW, 2>*)(pix + goff(j));
#pragma unroll
            for (int e = 0; e < GDW; ++e) r.d[j][e] = v.v[e];
        }
        const bool in = y < p.IH && x < p.IW;
        r.mw = __builtin_amdgcn_raw_buffer_load_b32(rsB, in ? (((b * p.IH + y) * p.IW + x) * 2 + lgrp) * 4 : EH_OOB, 0, 0);
    };

    // tile order: the blocks of one XCD (blockIdx % 8) walk one contiguous range of tiles (dectail_tile.hpp, DESIGN finding 24)
    const int G = (int)gridDim.x;
    const int vb = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    Staged cur;
    int t_nxt = min(vb + G, p.ntiles - 1);
    request(min(vb, p.ntiles - 1), cur);
    // LDS byte addresses of the lane's dy2 fragment at tap (1, 1) (the other taps are immediate offsets) and of its transposed-read roles (wgrad_tile.hpp)
    const uint32_t dya = (uint32_t)((li * EH_SC + lj) * EH_SP + lgrp * 16);
    const int tg = lane >> 4, tc = lane & 15;
    const int trow = (tg >> 1) * 8 + (tc >> 2), tcol = (tg & 1) * 16 + (tc & 3) * 4;

    // the frame patch + bit word of a tile are CONSUMED at the top of the tile (converted into the wave's patch tile, the bit word kept in one register) and the next
    // tile's are requested right behind, into the same registers: a whole tile (~9 k cycles per block at three blocks per CU) to land.  Requested at the top of their own
    // tile they had 16 MFMAs = 0.2 us; requested behind the conversion in the middle of the tile, half a tile: 57 % of the kernel's wave cycles were waits.
    Raw raw;
    request_patch(min(vb, p.ntiles - 1), raw);
    for (int tile = vb; tile < p.ntiles; tile += G) {
        commit(cur);
        // ---- the lane's half of its pixel's frame patch -> row lrow of the wave's patch tile (k order (kh * 4 + kw) * 3 + c = conv1's HWIO rows) ----
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            u16x8 xf;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const int j = 2 * s + gi;
                float f[4];
                if constexpr (SSZ == 1) {                 // camera bytes: k * (1 / 255), which rounds to the same bf16 as the exact quotient (common.hpp)
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = (float)((raw.d[j][0] >> (8 * e)) & 255u) * U8_RCP255;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = __builtin_bit_cast(float, raw.d[j][e < GDW ? e : 0]);
                }
                const PackN<uint32_t, 2> h = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(f));
                uint32_t* dst = (uint32_t*)&xf + 2 * gi;
                dst[0] = h.v[0]; dst[1] = h.v[1];
            }
            *(u16x8*)(ptw + lrow * 128 + (((2 * s + lgrp) ^ (lrow & 7)) << 4)) = xf;
        }
        const uint32_t mw = raw.mw;
        request_patch(t_nxt, raw);                         // (past the end: a valid tile again -- no branch around the loads)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the staged dy2 pixels are published; LDS-only: __syncthreads() would drain the requests just issued
        request(t_nxt, cur);                               // the next tile's dy2 pixels: land under this tile's work (LDS-only barriers from here on)
        t_nxt = min(t_nxt + G, p.ntiles - 1);

        // ---- g1 of this wave's 32 pixels: D[ci][pixel] = sum over taps, co ----
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 4; ++tap)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const u16x8 bf = *(const u16x8*)(dys + dya + ((1 - (tap >> 1)) * EH_SC + (1 - (tap & 1))) * EH_SP + kk * 32);
                u16x8 wa;
                if (tap < 3) wa = wf[tap][kk]; else wa = *(const u16x8*)(wlw + kk * 1024 + lane * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa), __builtin_bit_cast(bf16x8, bf), acc, 0, 0, 0);
            }
        // ---- epilogue of narrow_conv48_kernel<., 1>: bf16, half-wave exchange (lane (pixel, g) then owns channels 16 g .. 16 g + 15), ReluGrad from the bit word ----
        {
            uint32_t R[4][2];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float v[4] = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
                const PackN<uint32_t, 2> w2 = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(v));
                R[qd][0] = w2.v[0]; R[qd][1] = w2.v[1];
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                auto r0 = __builtin_amdgcn_permlane32_swap(R[0][d], R[2][d], false, false); R[0][d] = r0[0]; R[2][d] = r0[1];
                auto r1 = __builtin_amdgcn_permlane32_swap(R[1][d], R[3][d], false, false); R[1][d] = r1[0]; R[3][d] = r1[1];
            }
            uint32_t o[8] = {R[0][0], R[0][1], R[2][0], R[2][1], R[1][0], R[1][1], R[3][0], R[3][1]};       // dword d: channels 16 g + 2 d, + 1
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] &= ((mw >> d) & 0x00010001u) * 0xffffu;
            *(PackN<uint32_t, 4>*)(gtw + lrow * EH_GP + lgrp * 32) = PackN<uint32_t, 4>{{o[0], o[1], o[2], o[3]}};
            *(PackN<uint32_t, 4>*)(gtw + lrow * EH_GP + lgrp * 32 + 16) = PackN<uint32_t, 4>{{o[4], o[5], o[6], o[7]}};
        }
        __builtin_amdgcn_wave_barrier();                  // same wave, in-order LDS queue (g1 tile written above, patch tile at the top of the tile)
        // ---- filter (+ bias) gradient: dW[k][ci] += sum over this wave's pixels of patch[pixel][k] * g1[pixel][ci]; k = 48 is the all-ones column ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                   // k-step = 16 pixels
            u16x8 bfr;
            {
                const int q0 = ks * 16 + trow, q1 = q0 + 4;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(gtw + q0 * EH_GP + tcol * 2));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(gtw + q1 * EH_GP + tcol * 2));
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
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier (the next tile's loads stay in flight): every wave is done with the staged dy2 pixels
    }

    // ---- block totals: the four waves take turns on one 8 KB buffer -> this block's slab: rows 0 .. 47 of dW1 (row stride 32), then the 32 bias sums ----
    __syncthreads();
    float* red = (float*)(lds + EH_DYS);
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
        float* slab = p.slabs + (long long)blockIdx.x * EH_SLAB;
        if (k < 48) slab[k * 32 + n] = red[i];
        else if (k == 48) slab[64 * 32 + n] = red[i];
    }
}

}  // namespace mi
