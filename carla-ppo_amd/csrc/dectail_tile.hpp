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
#include <type_traits>
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
constexpr int DT_XP = 80;                                 // staged input pixels: 64 bytes (32 bf16 channels) at an 80-byte pitch -- 16 consecutive pixels start in 16
                                                          // different bank quads, so the 16-byte fragment reads, the mask reads and the transposed reads need no swizzle
                                                          // (and every tap of a slot is ONE base address + an immediate offset)
constexpr int DT_XS = (DT_NPIX * DT_XP + 255) & ~255;      // 14592
constexpr int DT_PT = 32 * 128;                           // per wave: transposed-read tile of its 32 patches, 64 columns (48 used)
constexpr int DT_SLAB = 48 * 32;                          // floats of dW per block
constexpr int DT_LBC = 104;                               // label tile: 18 rows x 102 values (34 pixels x 3) as fp32, 104-float pitch
constexpr int DT_LBBYTES = (2 * DT_TY + 2) * DT_LBC * 4;  // 7488

struct DecTailParams {
    const bf16_t* x; int B, IH, IW;                      // deconv3's output [B, IH, IW, 32] (post-ReLU)
    unsigned x_bytes;                                    // its size (< 2^31: read through a buffer descriptor with 32-bit offsets)
    const bf16_t* w;                                     // deconv4 kernel [4][4][3][32]  (kh, kw, out, in)
    const bf16_t* wt;                                    // the same K-contiguous: [32][48], k = (kh * 4 + kw) * 3 + out
    const float* bias;                                   // [3]
    const void* labels; int lab_u8; const int* lab_idx; long long lab_stride;   // target frames [*, OH * OW * 3] fp32 (or raw camera bytes)
    int loss_kind; float inv_b;
    bf16_t* dx;                                          // gradient of the loss wrt deconv3's PRE-activation [B, IH, IW, 32]
    float* slabs;                                        // [gridDim.x][DT_SLAB] partial filter gradients (reduce_slabs_kernel adds them to dW)
    float* lpart; float* bpart;                          // per block: loss partial sum; 4 floats of per-channel dlogits sums
    int OH, OW, GH, GW, tiles_x, tiles_per_frame, ntiles;
    int edge_own;                                        // 1: tiles cover the PIXEL grid; the last slot row / column (index IH / IW) is owned by the last tile row / column
    FastDiv div_tpf, div_tx;
    int dbg;                                             // TIMING INSTANTIATION only (dectail_kernel<.., true>, mi_set_tuning key 25; wrong results, honest durations -- tools/dectail_ablate.py):
                                                         // 1 no transcendentals in the loss, 2 no input-gradient phase, 4 no filter-gradient phase, 8 no gradient stores,
                                                         // 16 no loads (the first tile's data stay), 32 no slot groups at all (phase 1 off), 64 only ONE slot group per wave (no fifth group)
};

// two fp32 values -> two bf16 in one dword (low half = a): one v_cvt_pk_bf16_f32, round-to-nearest-even like f32_to_bf16
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_));
}

// block barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory queue (s_waitcnt vmcnt(0)), which would make the
// software prefetch of the next tile wait at the first barrier it meets
__device__ __forceinline__ void dt_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// (Round 6, measured and dropped -- DESIGN 3.16: the same body at TWO waves per SIMD with the forward weights in registers and all eight activation fragments of a slot group
//  requested before its first MFMA -- the generated code here is "two ds_read_b128, s_waitcnt lgkmcnt(0), MFMA" eight times per group, 165 of 170 registers leave no room to read
//  ahead -- 197 registers, two blocks per CU: 78.2 / 76.9 / 77.3 us against 74.7 / 75.6 / 74.9, step 0.8220 against 0.8199 ms.)
template <bool FASTBCE, bool DBG = false, bool SPLIT5 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void dectail_kernel(const DecTailParams p) {
    const int dbg = DBG ? p.dbg : 0;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[DT_XS + 4 * DT_PT + 3840 + 4 * 13 * 64 + DT_LBBYTES];
    unsigned char* const xs = lds;                        // staged input pixels
    unsigned char* const pt = lds + DT_XS;                // 4 wave-private patch tiles (the cross-wave reduction at the very end reuses them)
    unsigned char* const dl = lds + DT_XS + 4 * DT_PT;    // dlogits tile (+ 16 floats of block reduction behind it)
    static_assert(DT_DLBYTES + 16 + 64 <= 3840, "dlogits tile + reduction words");
    static_assert(DT_NSLOT > 128 && DT_NSLOT <= 160, "phase 1: four groups of 32 slots + one rotating fifth");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lgrp = lane >> 5;
    typedef __attribute__((address_space(3))) s16x4* lds_v4;

    // ---- per block, once: the forward weights into LDS as [tap][row ne = class * 3 + channel (12 live rows)][32 channels] (3 KB: in registers they
    // cost 32 VGPRs, which at three waves per SIMD spilled); the input-gradient weights (12 VGPRs) stay in registers ----
    unsigned char* const wl = dl + 3840;
    if (tid < 4 * 13 * 4) {                               // row 12 of every tap is all zero: lanes 12 .. 31 of the MFMA's A operand read it
        const int tap = tid / 52, rem = tid - tap * 52, ne = rem >> 2, ch = rem & 3;
        const int cls = ne / 3, n = ne - cls * 3;
        const int kh = (cls >> 1) + 2 * (1 - (tap >> 1)), kw = (cls & 1) + 2 * (1 - (tap & 1));
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ne < 12) v = *(const f32x4*)(p.w + ((kh * 4 + kw) * 3 + n) * 32 + ch * 8);
        *(f32x4*)(wl + (tap * 13 + ne) * 64 + ch * 16) = v;
    }
    float* const lab = (float*)(wl + 4 * 13 * 64);        // the tile's labels as fp32 (a global byte load per logit made the loss loop a chain of memory latencies)
    const int wrow = lrow < 12 ? lrow : 12;
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

    // The staged data of a tile -- 720 sixteen-byte chunks of input pixels (3 per thread), 468 four-value label items (2 per thread) -- is REQUESTED one
    // tile ahead: the loads of tile t + 1 are issued right after the barrier that publishes tile t and land under its three phases (a block that stages,
    // waits, computes was a chain of one HBM latency per tile: with three blocks per CU that was half the kernel).  Per-thread roles are fixed:
    int sq[3], spc[3]; bool sin3[3];                       // input chunk i: tile pixel, 16-byte chunk
    int lr[2], ld4[2]; bool lin2[2];                       // label item i: tile row, item
    int xrel[3], xr_[3], xc_[3];                           // input chunk i: byte offset relative to the tile origin's pixel, tile row / column
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int id = tid + 256 * i; sin3[i] = id < DT_NPIX * 4; sq[i] = sin3[i] ? id >> 2 : 0; spc[i] = id & 3;
        xr_[i] = sq[i] / DT_PC; xc_[i] = sq[i] - xr_[i] * DT_PC;
        xrel[i] = ((xr_[i] - 1) * p.IW + xc_[i] - 1) * 64 + spc[i] * 16;
    }
    int lrel[2];                                          // label item i: value offset relative to the tile's first label value
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int id = tid + 256 * i; lin2[i] = id < (2 * DT_TY + 2) * (DT_LBC / 4); lr[i] = lin2[i] ? id / (DT_LBC / 4) : 0; ld4[i] = lin2[i] ? id - lr[i] * (DT_LBC / 4) : 0;
        lrel[i] = lr[i] * (3 * p.OW) + 4 * ld4[i];
    }
    // Both tensors are read through buffer descriptors with 32-bit offsets (the host checks the sizes): an item outside the image / the frame is requested at
    // an offset past the end and comes back as zeros from the range check -- no branch around a load (with exec-masked loads the compiler loses count of what is
    // in flight and falls back to s_waitcnt vmcnt(0), which drains the prefetch), no select when it is committed, no 64-bit address arithmetic.
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    const int lesz = p.lab_u8 ? 1 : 4;
    constexpr int DT_OOB = 0x7ffffff0;
    // the label frame of a tile (minibatch gather) is a SCALAR load (s_load_dword + its own lgkmcnt wait: the 2 KB index vector lives in the scalar cache).  As a
    // vector load it was followed by v_readfirstlane behind an s_waitcnt vmcnt(0) -- which drains the prefetch just issued.  The scalar frame also gives the label
    // descriptor an exact base and extent per tile: offsets inside one frame, range check = the frame.
    auto frame_of = [&](int tile) -> int {
        const int b = (int)p.div_tpf.div((uint32_t)tile);
        int fr = b;
        if (p.lab_idx) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(fr) : "s"(p.lab_idx + b) : "memory");
        return fr;
    };
    struct Staged { f32x4 x[3]; f32x4 l[2]; };             // (uint8 labels: the raw dword travels in l[i][0])
    auto request = [&](int tile, int frame, Staged& R) {
        uint32_t b, rem, ty, tx;
        p.div_tpf.divmod((uint32_t)tile, b, rem);
        p.div_tx.divmod(rem, ty, tx);
        const int y0 = (int)ty * DT_TY, x0 = (int)tx * DT_TX;
        const int xbase = (((int)b * p.IH + y0) * p.IW + x0) * 64;                  // bytes; the host guarantees B IH IW 64 < 2^31
        const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned char*)p.labels + (long long)frame * p.lab_stride * lesz), 0,
                                                                             p.OH * p.OW * 3 * lesz, 0x00020000);
        const int lbase = 2 * y0 * (3 * p.OW) + 6 * x0;                             // values inside the frame
        // labels of the 18 x 34 output pixels the tile's slots cover: row r = output row 2 y0 + r, 102 consecutive values from column 2 x0 (4 values per
        // item; whole items only: the host guarantees 3 OW % 4 == 0, so a row's valid part ends on an item boundary; the rest reads 0)
        int loff[2], xoff[3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool in = lin2[i] && 2 * y0 + lr[i] < p.OH && 6 * x0 + 4 * ld4[i] + 4 <= 3 * p.OW;
            loff[i] = in ? (lbase + lrel[i]) * lesz : DT_OOB;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool in = sin3[i] && (unsigned)(y0 - 1 + xr_[i]) < (unsigned)p.IH && (unsigned)(x0 - 1 + xc_[i]) < (unsigned)p.IW;
            xoff[i] = in ? xbase + xrel[i] : DT_OOB;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) R.x[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, xoff[i], 0, 0));
        if (p.lab_u8) {
#pragma unroll
            for (int i = 0; i < 2; ++i) R.l[i][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsL, loff[i], 0, 0));
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) R.l[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsL, loff[i], 0, 0));
        }
    };
    auto commit = [&](const Staged& R) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (sin3[i]) *(f32x4*)(xs + sq[i] * DT_XP + spc[i] * 16) = R.x[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 v = R.l[i];
            if (p.lab_u8) {
                const uint32_t u = __builtin_bit_cast(uint32_t, (float)R.l[i][0]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = u8_to_unit_exact((float)((u >> (8 * e)) & 255u));
            }
            if (lin2[i]) *(f32x4*)(lab + lr[i] * DT_LBC + 4 * ld4[i]) = v;
        }
    };
    Staged cur;
    // Tile order: block b runs on XCD b % 8 (observed placement; only speed depends on it), and neighbouring tiles share a third of their staged pixels.
    // In every round of gridDim.x tiles XCD x takes the contiguous run [x G / 8, (x + 1) G / 8): the halo re-reads then hit that XCD's L2 (with tile =
    // b + k G the eight neighbours of a tile sat on eight different L2s: 186 MB fetched from HBM for 121 MB of tensors).  The host launches G % 8 == 0.
    const int G = (int)gridDim.x;
    const int vb = (G & 7) == 0 ? ((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    int t_nxt = min(vb + G, p.ntiles - 1);                   // (past the end: a valid tile is requested again -- no branch around the loads)
    request(min(vb, p.ntiles - 1), frame_of(min(vb, p.ntiles - 1)), cur);
    // Phase 1 has FIVE groups of 32 slots for four waves (9 x 17 = 153 slots): wave w takes group w, wave 0 also the fifth.  (Rotating the fifth group over the
    // waves from tile to tile changed nothing: 88.6 vs 90.3 us.)  The slot of a lane in its wave's own group never changes: its LDS addresses are set up once.
    struct SlotLane { int sy, sx; uint32_t xa, la, da; bool sv; };       // slot; byte addresses of its first tap's fragment / its label row / its dlogits row
    auto slot_lane = [&](int g) -> SlotLane {
        SlotLane L;
        const int sidx = min(g * 32 + lrow, DT_NSLOT - 1);
        L.sv = g * 32 + lrow < DT_NSLOT;
        L.sy = sidx / DT_SX; L.sx = sidx - L.sy * DT_SX;
        L.xa = (uint32_t)((L.sy * DT_PC + L.sx) * DT_XP + lgrp * 16);
        L.la = (uint32_t)(((2 * L.sy + lgrp) * DT_LBC + L.sx * 6) * 4);
        L.da = (uint32_t)((2 * L.sy + lgrp) * DT_DLPITCH + L.sx * 12);
        return L;
    };
    const SlotLane own = slot_lane(wave);
    // the fifth group: slots 128 .. 152 = slot row 7, columns 9 .. 16 and slot row 8 (set up once: the generic decode inside the tile loop was 49 VALU instructions per tile)
    SlotLane fifth;
    {
        const int sidx = min(128 + lrow, DT_NSLOT - 1);
        fifth.sv = 128 + lrow < DT_NSLOT;
        fifth.sy = sidx >= 8 * DT_SX ? 8 : 7; fifth.sx = sidx - fifth.sy * DT_SX;
        fifth.xa = (uint32_t)((fifth.sy * DT_PC + fifth.sx) * DT_XP + lgrp * 16);
        fifth.la = (uint32_t)(((2 * fifth.sy + lgrp) * DT_LBC + fifth.sx * 6) * 4);
        fifth.da = (uint32_t)((2 * fifth.sy + lgrp) * DT_DLPITCH + fifth.sx * 12);
    }
    static_assert(DT_SY == 9 && DT_SX == 17, "fifth group: slot rows 7 and 8");
    const uint32_t wa = (uint32_t)(wrow * 64 + lgrp * 16);
    for (int tile = vb; tile < p.ntiles; tile += G) {
        uint32_t b, rem, ty, tx;
        p.div_tpf.divmod((uint32_t)tile, b, rem);
        p.div_tx.divmod(rem, ty, tx);
        const int y0 = (int)ty * DT_TY, x0 = (int)tx * DT_TX;
        commit(cur);
        dt_lds_barrier();                                 // LDS only: __syncthreads() also waited for the ACKNOWLEDGEMENTS of the previous tile's gradient stores (vmcnt counts stores too)
        if (!(dbg & 16)) request(t_nxt, frame_of(t_nxt), cur);
        t_nxt = min(t_nxt + G, p.ntiles - 1);

        // ---- phase 1: logits of the 9 x 17 slots, loss, dlogits into the LDS tile ----
        // PAIR < 0: the whole group (all six logits of the lane's output row).  PAIR = 0 | 1 | 2 (round 6): only logit pair PAIR -- the FIFTH group (25 of its 32 slots live) used to
        // be wave 0's second group while three waves waited at the barrier: 12.9 of the kernel's 76.7 us (tools/dectail_ablate.py).  Its eight MFMAs and four half-wave swaps are
        // cheap; the ~110 VALU instructions + 18 transcendentals of the loss are not: in the SPLIT5 instantiation (MI355_DECTAIL_SPLIT5=1, mi_set_tuning key 26) waves 0, 1, 2 each run the MFMAs of the fifth group and the loss of ONE of
        // its three logit pairs (same values, same dlogits tile; the loss / bias partial sums are regrouped: fp32 summation order).  MEASURED NEUTRAL (one box, three interleaved
        // rounds: 74.4 / 77.4 / 74.9 us one wave, 77.2 / 76.2 / 77.6 us shared; step 0.8440 = 0.8440 ms): what the fifth group costs its wave is not the loss arithmetic but the
        // LATENCY CHAIN in front of it -- eight times two LDS reads, a wait, an MFMA -- and three waves now pay that chain instead of one.  Kept behind the knob (default off)
        // with its test.
        auto slot_group = [&](const SlotLane& L, auto pair_c) {
            constexpr int PAIR = decltype(pair_c)::value;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 4; ++tap)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const u16x8 af = *(const u16x8*)(xs + L.xa + ((tap >> 1) * DT_PC + (tap & 1)) * DT_XP + kk * 32);
                    const u32x4 wv = *(const u32x4*)(wl + wa + tap * 13 * 64 + kk * 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, af), acc, 0, 0, 0);
                }
            // (round 6, measured and dropped: two accumulators -- two chains of four MFMAs instead of one of eight -- 77.3 / 77.8 / 76.2 us against 76.0 / 75.5 / 74.8; the
            //  generated code is "two ds_read_b128, s_waitcnt lgkmcnt(0), MFMA" eight times: what a lone group waits for is the LDS latency in front of every MFMA, and with
            //  three waves per SIMD at 165 of 170 registers there is no room to read further ahead: DESIGN 3.16)
            // D rows: register r of half-wave h is row (r & 3) + 8 (r >> 2) + 4 h; rows 0 .. 11 = class * 3 + channel are live: h = 0 holds rows 0..3
            // (registers 0..3) and 8..11 (registers 4..7), h = 1 rows 4..7 (registers 0..3).  Four half-wave swaps give every lane ONE output row of its
            // slot -- h = 0: rows 0..5 = output row 2 sy (pixels 2 sx, 2 sx + 1 x 3 channels), h = 1: rows 6..11 = output row 2 sy + 1 -- six consecutive
            // values in the label tile and in the dlogits tile, with the channel of value j a literal (j % 3): no per-value address arithmetic or selects.
            float v0 = acc[0], v1 = acc[1], v2 = acc[2], v3 = acc[3], v4 = acc[4], v5 = acc[5], v6 = acc[6], v7 = acc[7];
            {
                auto sw = [](float& x, float& y) {            // x.upper <- y.lower, y.lower <- x.upper
                    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, x), __builtin_bit_cast(uint32_t, y), false, false);
                    x = __builtin_bit_cast(float, (uint32_t)r[0]); y = __builtin_bit_cast(float, (uint32_t)r[1]);
                };
                sw(v0, v4); sw(v1, v5);                       // v0 = [row 0 | row 8], v4 = [row 4 | -];  v1 = [row 1 | row 9], v5 = [row 5 | -]
                sw(v4, v6); sw(v5, v7);                       // v4 = [row 4 | row 10],                   v5 = [row 5 | row 11]
            }
            // lower half: rows 0..5 = v0 v1 v2 v3 v4 v5; upper half: rows 6..11 = v2 v3 v0 v1 v4 v5
            const float xr[6] = {lgrp ? v2 : v0, lgrp ? v3 : v1, lgrp ? v0 : v2, lgrp ? v1 : v3, v4, v5};
            // a tile owns the slots of its own 8 x 16 pixels; the slot grid is one row and one column larger than the pixel grid, and that last row /
            // column -- the ninth / seventeenth of the last tiles, computed anyway for their pixels' patches -- belongs to them
            // (bitwise on purpose: with && / || hipcc turned these into five branches per group)
            const int ylast = p.GH - 1 - y0, xlast = p.GW - 1 - x0;
            const bool eo = p.edge_own != 0;
            const bool slot_in = L.sv & (L.sy <= ylast) & (L.sx <= xlast);
            const bool owned = slot_in & ((L.sy < DT_TY) | (eo & (L.sy == ylast))) & ((L.sx < DT_TX) | (eo & (L.sx == xlast)));
            // masks as DATA: a select per value made hipcc branch around each logit's gradient (six s_and_saveexec / s_cbranch_execz pairs per group)
            const uint32_t inm = slot_in ? 0xffffffffu : 0u;
            const float ownf = owned ? 1.0f : 0.0f;
            float yv[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) { const PackN<float, 2> t = *(const PackN<float, 2>*)((const unsigned char*)lab + L.la + 8 * j); yv[2 * j] = t.v[0]; yv[2 * j + 1] = t.v[1]; }
            uint32_t gw[3];
#pragma unroll
            for (int jp = 0; jp < 3; ++jp) {                // two logits at a time: one v_cvt_pk_bf16_f32 rounds both (logits as STORED, dlogits as STORED)
                if (PAIR >= 0 && jp != PAIR) continue;
                const int ja = 2 * jp, jb = 2 * jp + 1;
                const float xa_ = xr[ja] + (ja % 3 == 0 ? bias0 : (ja % 3 == 1 ? bias1 : bias2));
                const float xb_ = xr[jb] + (jb % 3 == 0 ? bias0 : (jb % 3 == 1 ? bias1 : bias2));
                const uint32_t xpk = pack2_bf16(xa_, xb_);
                const float xv2[2] = {__builtin_bit_cast(float, xpk << 16), __builtin_bit_cast(float, xpk & 0xffff0000u)};
                float gr2[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int j = 2 * jp + h;
                    const float xv = xv2[h];
                    float l, gr;
                    if (DBG && (dbg & 1)) { l = xv * yv[j]; gr = xv - yv[j]; }
                    else
                    if constexpr (FASTBCE) {               // loss_kind 0 on the hardware transcendentals (gather_narrow_kernel, FASTBCE)
                        const float e = __builtin_amdgcn_exp2f(-fabsf(xv) * 1.44269504f);
                        const float s1 = 1.0f + e;
                        const float rr = __builtin_amdgcn_rcpf(s1);
                        const float sg = xv >= 0.f ? rr : e * rr;
                        l = fmaf(__builtin_amdgcn_logf(s1), 0.69314718f, fmaf(-xv, yv[j], fmaxf(xv, 0.f)));
                        gr = sg - yv[j];
                    } else {
                        const float e = __expf(-fabsf(xv));
                        const float rr = __frcp_rn(1.0f + e);
                        const float sg = xv >= 0.f ? rr : e * rr;
                        if (p.loss_kind == 0) { l = fmaxf(xv, 0.f) - xv * yv[j] + __logf(1.0f + e); gr = sg - yv[j]; }
                        else if (p.loss_kind == 1) {
                            l = -(yv[j] * __logf(1e-10f + sg) + (1.0f - yv[j]) * __logf(1e-10f + 1.0f - sg));
                            gr = (-yv[j] / (1e-10f + sg) + (1.0f - yv[j]) / (1e-10f + 1.0f - sg)) * sg * (1.0f - sg);
                        } else { const float dd = yv[j] - sg; l = dd * dd; gr = -2.0f * dd * sg * (1.0f - sg); }
                    }
                    gr2[h] = gr * p.inv_b;
                    lsum = fmaf(l, ownf, lsum);
                }
                gw[jp] = pack2_bf16(gr2[0], gr2[1]) & inm;
                // the bias gradient sums the STORED (rounded) values, like BiasAddGrad of dlogits
                const float ga = __builtin_bit_cast(float, gw[jp] << 16), gb = __builtin_bit_cast(float, gw[jp] & 0xffff0000u);
                if (ja % 3 == 0) gs0 = fmaf(ga, ownf, gs0); else if (ja % 3 == 1) gs1 = fmaf(ga, ownf, gs1); else gs2 = fmaf(ga, ownf, gs2);
                if (jb % 3 == 0) gs0 = fmaf(gb, ownf, gs0); else if (jb % 3 == 1) gs1 = fmaf(gb, ownf, gs1); else gs2 = fmaf(gb, ownf, gs2);
            }
            if (L.sv) {
                uint32_t* drow = (uint32_t*)(dl + L.da);
#pragma unroll
                for (int j = 0; j < 3; ++j) if (PAIR < 0 || j == PAIR) drow[j] = gw[j];
            }
        };
        if (!(dbg & 32)) {
            slot_group(own, std::integral_constant<int, -1>{});
            if (!(dbg & 64)) {
                if constexpr (SPLIT5) {                       // (its own instantiation: no run-time switch inside the product kernel's tile loop)
                    if (wave == 0) slot_group(fifth, std::integral_constant<int, 0>{});
                    else if (wave == 1) slot_group(fifth, std::integral_constant<int, 1>{});
                    else if (wave == 2) slot_group(fifth, std::integral_constant<int, 2>{});
                } else if (wave == 0) slot_group(fifth, std::integral_constant<int, -1>{});
            }
        }
        dt_lds_barrier();                                 // dlogits tile complete (LDS only: the next tile's global loads stay in flight)

        // ---- phase 2: input gradient of this wave's 32 pixels: rows yl = 2 wave, 2 wave + 1, columns xl = 0 .. 15 ----
        const int yl = 2 * wave + (lrow >> 4), xl = lrow & 15;
        if (DBG && (dbg & 2) && (dbg & 4)) { dt_lds_barrier(); continue; }
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
        if (!(DBG && (dbg & 2))) {
#pragma unroll
        for (int s = 0; s < 3; ++s) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wtf[s]), __builtin_bit_cast(bf16x8, xf[s]), acc2, 0, 0, 0);
        }
        if (!(DBG && (dbg & 2))) {
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
            const PackN<uint32_t, 4> m0 = *(const PackN<uint32_t, 4>*)(xs + qm * DT_XP + lgrp * 32);
            const PackN<uint32_t, 4> m1 = *(const PackN<uint32_t, 4>*)(xs + qm * DT_XP + lgrp * 32 + 16);
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                uint32_t pos, nz;                          // per 16-bit half: x > 0 as a signed integer (negative floats are negative int16) -> 1, else 0
                asm("v_pk_max_i16 %0, %1, 0" : "=v"(pos) : "v"(d < 4 ? m0.v[d] : m1.v[d - 4]));
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(pos), "v"(0x00010001u));
                o[d] &= nz * 0xffffu;
            }
            const int y = y0 + yl, x = x0 + xl;
            if (y < p.IH && x < p.IW && !(DBG && (dbg & 8))) {
                bf16_t* out = p.dx + (((long long)b * p.IH + y) * p.IW + x) * 32 + lgrp * 16;
                *(PackN<uint32_t, 4>*)out = PackN<uint32_t, 4>{{o[0], o[1], o[2], o[3]}};
                *(PackN<uint32_t, 4>*)(out + 8) = PackN<uint32_t, 4>{{o[4], o[5], o[6], o[7]}};
            }
        }

        // ---- phase 3: filter gradient  dW[k][ci] += sum over this wave's pixels of patch[pixel][k] * x[pixel][ci] ----
        // patches: this lane's 24 values -> row lrow of the wave's transposed-read tile; the activations are read transposed from the staged tile
        if (DBG && (dbg & 4)) { dt_lds_barrier(); continue; }
#pragma unroll
        for (int s = 0; s < 3; ++s) *(u16x8*)(ptw + lrow * 128 + (((2 * s + lgrp) ^ (lrow & 7)) << 4)) = xf[s];
        __builtin_amdgcn_wave_barrier();                  // same wave, in-order LDS queue
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                   // k-step = the 16 pixels of tile row 2 wave + ks
            u16x8 bfr;
            {
                const int q0 = (2 * wave + ks + 1) * DT_PC + 1 + trow;      // staged-tile pixel of row trow (and trow + 4)
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(xs + q0 * DT_XP + tcol * 2));
                const int q1 = q0 + 4;
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(xs + q1 * DT_XP + tcol * 2));
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
        dt_lds_barrier();                                 // every wave is done with the staged tiles before the next tile overwrites them
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

// dW[1536] += sum over the n per-block slabs: a block owns 32 consecutive outputs (128 bytes per slab row), its 32 thread rows take every 32nd slab,
// eight loads in flight per thread (768 slabs: three batches); fixed summation order (deterministic).  (256-thread blocks with 8 thread rows: 12 batches
// of dependent-latency loads, 16 us at the very end of the step's critical path.)
__global__ __launch_bounds__(1024) void dectail_reduce_kernel(const float* __restrict__ slabs, int n, float* __restrict__ dw) {
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const int col = (int)blockIdx.x * 32 + c;
    float s = 0.f;
    for (int k0 = r0; k0 < n; k0 += 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + 32 * u; v[u] = k < n ? __builtin_nontemporal_load(&slabs[(long long)k * DT_SLAB + col]) : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    red[r0][c] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
        dw[col] += t;
    }
}

}  // namespace mi
