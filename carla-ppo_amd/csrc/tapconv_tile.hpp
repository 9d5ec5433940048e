// tapconv_tile.hpp — raw-staged MFMA kernel for the stride-2 conv / transposed-conv layers (gfx950, wave64).
//
// Both stride-2 forms on this path are stride-1 TAP convolutions on a half-resolution "slot" grid:
//   conv form   (conv fwd, deconv dgrad):  out[b,oy,ox,n]       = sum_{ta,tb} sum_{ph,pw,c} x[b,2(oy+ta)+ph,2(ox+tb)+pw,c] W[2ta+ph,2tb+pw,c,n]
//                                          slot (gy,gx) = the 2x2 pixel block (2gy+ph, 2gx+pw): 4C channels, N outputs
//   gather form (deconv fwd, conv dgrad):  out[b,2y+ph,2x+pw,n] = sum_{th,tw} sum_c x[b,y-th,x-tw,c] W[ph+2th,pw+2tw,n,c]
//                                          slot (gy,gx) = pixel (gy-HY, gx-HX): C channels, 4N outputs (the 4 output parities)
// With slots numbered P = (b*GH + gy)*GW + gx, output position P needs slots P + ta*GW + tb, ta<TH, tb<TW (2x2 taps for k=4,
// 3x3 with some all-zero weight blocks for k=5).  So a block
//   * stages the slot range [P0, P0 + 256 + halo) of one 128-byte channel slice ONCE per slice (LDS-DMA, zero fill of
//     everything outside the image by the buffer range check) instead of once per tap  -> 4-9x fewer bytes through the
//     per-CU load path (~52 B/clk/CU from L2, the measured bound of the im2col-gather kernels),
//   * streams one weight tile [BNE x 128 B] per (slice, tap) step,
//   * computes a 256-position x BNE-output tile with 4 waves (128 x BNE/2 each, one wave per SIMD: 32 MFMAs per barrier
//     cover the ~150 address/issue instructions of a step); tap shifts are plain LDS address offsets.
// LDS rows are 128 B, XOR-swizzled on the DMA source side (chunk ^= (row >> 1) & 7) -> conflict-free ds_read_b128.
// Rows of dummy positions (gy >= OH etc.) are computed and dropped in the epilogue.
#pragma once
#include <type_traits>
#include "gemm2_tile.hpp"

namespace mi {

enum { TC_CONV = 0, TC_GATHER = 1 };
constexpr int TC_BMT = 256;          // positions per block
constexpr int TC_MAXHALO = 96;       // (TH-1)*GW + TW-1 must not exceed this
constexpr int TC_NT = 512;           // threads per block: 8 waves (one wave per SIMD measured 1.4x slower: nothing hides its LDS latency)
constexpr int TC_NBUF = 2;           // weight stages (double buffer)

struct TapParams {
    const void* a; uint32_t a_bytes;
    const void* b; uint32_t b_bytes;
    int B, IH, IW, C;                // input tensor
    int OH, OW, N;                   // output tensor
    int KH, KW;
    int GH, GW, TH, TW, HY, HX;      // slot grid per image, taps, gather-form halos
    int KC, NE, MP;                  // channels per slot (4C | C), effective outputs (N | 4N), B*GH*GW
    int ldb;                         // conv form: row pitch (elements) of the K-contiguous weight copy [N][KH*KW*C]
    const void* bfrag;               // rwconv_conv_kernel<4, 2, .., WFRAG>: the same weights in the kernel's own fragment order (mi_ares_pack_weights8 form 3), 1 KB contiguous per wave load
    FastDiv div_g, div_gw, div_n, div_2c, div_c;
    void* out; const float* bias; const void* mask; int relu;
    const uint32_t* mask_bits; uint32_t* bits_out;    // rwconv.hip only: ReLU bit words read instead of `mask` / written next to `out` (mi355_carla.h)
    int direct_epilogue;             // 1: registers -> 16-byte stores (half-wave swap), 0: LDS-staged coalesced stores
    long long* trace; int trace_cap;   // debug: per-wave s_memtime stamps (mi_debug_set_trace), nullptr in production
    int dbg;                           // debug (mi_set_tuning key 2): 3 = direct epilogue without its stores
    int mask_prefetch;                 // touch the ReluGrad-mask lines in the last main-loop step (mi_set_tuning key 12)
    // gather_narrow_kernel only: reconstruction loss fused into the epilogue (labels == nullptr: plain transposed conv)
    const float* labels; const int* lab_idx; long long lab_stride;   // target frames [*, OH*OW*N] fp32 (or uint8 bytes, lab_u8), optional gather
    int lab_u8;
    int loss_kind; float inv_b;
    void* dlogits;                     // d loss / d logits * inv_b, same layout as out (nullptr: loss only)
    float* lpart; float* bpart;        // per block: loss partial sum; 4 floats of per-channel dlogits sums
};

// TAPS: taps per axis (2 for k <= 4, 3 for k = 5,6); the tap loop is unrolled so tap offsets / issue slots are literals.
// BMT x MAXHALO: positions per block and the largest tap reach it can stage.  256 x 96: 8 waves, 155 KB of LDS, one block per CU.
// 128 x 48: 4 waves, 77 KB, TWO independent blocks per CU: one block's prologue / epilogue / barrier waits are covered by the
// other block's MFMAs (the big tile leaves 25-45 % of a block's life MFMA-idle with nothing to overlap it).
template <typename T, int MODE, int BNE, int TAPS, int BMT, int MAXHALO>
__global__ __launch_bounds__(BMT * 2) void tapconv_kernel(const TapParams p) {
    constexpr int TC_NT = BMT * 2;                        // threads: one wave per 32 positions x 2 output halves
    constexpr int RB = 128;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int CHS = RB / ESZ;                         // channels per stage
    constexpr int VE = 16 / ESZ;
    constexpr int MAXSLOT = BMT + MAXHALO;                // 352 | 176, multiple of 8
    constexpr int NWAVE = TC_NT / 64;
    constexpr int NIA = (MAXSLOT / 8 + NWAVE - 1) / NWAVE; // A-tile DMA instructions per wave (upper bound: ceil(44 / 8) = 6)
    constexpr int WN = 2, WM = NWAVE / WN;
    constexpr int TM = BMT / WM / 32;                     // 2
    constexpr int TN = BNE / WN / 32;                     // 2 | 1
    constexpr int TPS = BMT / BNE;                        // taps per barrier step: 32 (16 for the small tile) MFMAs per wave between barriers
    constexpr int NSS = (TAPS * TAPS + TPS - 1) / TPS;    // steps per channel slice
    constexpr int NJB = BNE / 8 / NWAVE;                  // B-tile DMA instructions per wave (8 rows each)
    constexpr int ASTAGE = MAXSLOT * RB, BTILE = BNE * RB, BSTAGE = TPS * BTILE;
    static_assert(TN >= 1 && NJB >= 1, "tile config");
    typedef typename Frag<T>::reg freg;

    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * ASTAGE + TC_NBUF * BSTAGE];
    unsigned char* const Abase = lds;
    unsigned char* const Bbase = lds + 2 * ASTAGE;

    constexpr int NT = TAPS * TAPS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: LDS-DMA bases stay in SGPRs
    const int wm = wave / WN, wn = wave % WN;
    // 32-output tile of accumulator column j of this wave.  k = 5 gather form with one parity class per tile (N = 32, four tiles): the classes
    // reach 9 / 6 / 6 / 4 of the 3 x 3 taps (zero tiles are skipped), so the natural split {0,1} | {2,3} gives one wave column 15 tile-taps and
    // the other 10; {0,3} | {1,2} gives 13 and 12.
    const bool pair_classes = TAPS == 3 && MODE == TC_GATHER && TN == 2 && p.N == 32 && p.NE == 128;
    auto tile_of = [&](int j) { const int t = wn * TN + j; return pair_classes ? ((0x9C >> (2 * t)) & 3) : t; };   // 0x9C: {0,3,1,2}
    const int lrow = lane & 31, lgrp = lane >> 5;
    const int r8 = lane >> 3;

    int tr_n = 0;
    long long* const tr = p.trace ? p.trace + ((long long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (tid >> 6)) * 32 : nullptr;   // 8 wave slots per block in the trace layout
    const bool tr_on = tr && ((long long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 8) * 32 <= p.trace_cap && lane == 0;
#define TC_STAMP() do { if (tr_on && tr_n < 32) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    TC_STAMP();
    const int P0 = xcd_remap(blockIdx.x, gridDim.x) * BMT;
    const int n0 = blockIdx.y * BNE;
    // bias of this lane's output channels (register r of a lane: channel 4 lgrp + 8 (r >> 2) + (r & 3) of a 32-output tile), requested
    // first so that it is back long before the accumulators are initialised with it: no bias loads between the stores of the epilogue
    // (a load there makes the wave wait for the stores before it) and no load latency in front of the first step either
    f32x4 bias4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bias4[j][q] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int ne = n0 + tile_of(j) * 32 + 4 * lgrp + 8 * q;
            if (p.bias && ne < p.NE) {
                int nb = ne;
                if constexpr (MODE == TC_GATHER) nb -= (int)p.div_n.div((uint32_t)ne) * p.N;
                bias4[j][q] = *(const f32x4*)(p.bias + nb);
            }
        }
    const int halo = (TAPS - 1) * p.GW + TAPS - 1;
    const int ninstrA = (BMT + halo + 7) >> 3;            // 8-slot DMA instructions covering the staged range
    const int NCC = (p.KC + CHS - 1) / CHS;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, (int)p.b_bytes, 0x00020000);

    // ---------------- A-tile DMA roles: instruction t = wave + NWAVE i fills slots 8t .. 8t+7 ----------------
    // logical chunk of this thread: (lane & 7) ^ ((slot >> 1) & 7), slot = 8 (wave + NWAVE i) + r8  ->  independent of i
    const int cchA = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
    uint32_t offA[NIA];                                   // byte offset of the slot's first pixel (G2_OOB: slot outside every image)
    uint32_t vmA[NIA];                                    // conv form: validity of the (ph,pw) sub-pixels, bit ph*2+pw
    auto slotA = [&](int i) {                              // slot -> pixel decode of this thread's row of DMA instruction i
        const int t = wave + NWAVE * i;
        const int P = P0 + 8 * t + r8;
        const bool ok = t < ninstrA && P < p.MP;
        uint32_t g, gx, b, gy;
        p.div_gw.divmod((uint32_t)(ok ? P : 0), g, gx);
        p.div_g.divmod(g, b, gy);
        if constexpr (MODE == TC_CONV) {
            const int y0 = 2 * (int)gy, x0 = 2 * (int)gx;
            uint32_t vm = 0;
            if (ok) {
                if (y0 < p.IH && x0 < p.IW) vm |= 1u;
                if (y0 < p.IH && x0 + 1 < p.IW) vm |= 2u;
                if (y0 + 1 < p.IH && x0 < p.IW) vm |= 4u;
                if (y0 + 1 < p.IH && x0 + 1 < p.IW) vm |= 8u;
            }
            vmA[i] = vm;
            offA[i] = (((b * p.IH + y0) * p.IW + x0) * p.C) * ESZ;
        } else {
            const int iy = (int)gy - p.HY, ix = (int)gx - p.HX;
            const bool in = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            offA[i] = in ? (((b * p.IH + iy) * p.IW + ix) * p.C) * ESZ : G2_OOB;
            vmA[i] = 0;
        }
    };
    // per channel slice: this thread's chunk -> byte offset inside the slot (+ which sub-pixel it belongs to, conv form)
    uint32_t sl_koff = 0, sl_bit = 0;
    auto sliceA = [&](int cc) {
        const int kc = cc * CHS + cchA * VE;
        if constexpr (MODE == TC_CONV) {
            uint32_t phh, r, pww, c;
            p.div_2c.divmod((uint32_t)kc, phh, r);        // kc = ph*2C + pw*C + c ; (pw,c) is contiguous in memory
            p.div_c.divmod(r, pww, c);
            sl_koff = (phh * p.IW * p.C + r) * ESZ;
            sl_bit = kc < p.KC ? 1u << (phh * 2 + pww) : 0u;
        } else {
            sl_koff = kc < p.KC ? (uint32_t)kc * ESZ : G2_OOB;
        }
    };
    auto issueA = [&](int buf, int i) -> int {            // one DMA instruction (8 slots x 128 B) of the slice set up by sliceA
        const int t = wave + NWAVE * i;
        if (t >= ninstrA) return 0;                       // wave-uniform
        uint32_t vo;
        if constexpr (MODE == TC_CONV) vo = (vmA[i] & sl_bit) ? offA[i] + sl_koff : G2_OOB;
        else vo = offA[i] + sl_koff;                      // G2_OOB + anything below 2^30 stays out of range
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(Abase + buf * ASTAGE + t * 1024), 16, (int)vo, 0, 0, 0);
        return 1;
    };

    // slice 0 of the slot range goes out instruction by instruction as soon as its rows are decoded: the first loads are in flight ~2k
    // cycles earlier than with "decode everything, then issue everything", under the rest of the set-up (weight roles, LDS addresses)
    sliceA(0);
#pragma unroll
    for (int i = 0; i < NIA; ++i) { slotA(i); issueA(0, i); }

    // ---------------- B-tile DMA roles: wave fills rows 8 (wave*NJB + j) .. +7 ----------------
    int cchB[NJB];
    uint32_t offB[NJB];                                   // row part of the weight offset (elements) or G2_OOB
    int clsB[NJB];
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
        const int row = 8 * (wave * NJB + j) + r8;
        cchB[j] = (lane & 7) ^ ((row >> 1) & 7);
        const int ne = n0 + row;
        if constexpr (MODE == TC_CONV) {
            clsB[j] = 0;
            offB[j] = ne < p.N ? (uint32_t)ne * (uint32_t)p.ldb : G2_OOB;
        } else {
            uint32_t cls, n;
            p.div_n.divmod((uint32_t)ne, cls, n);
            clsB[j] = (int)cls;
            offB[j] = ne < p.NE ? n * (uint32_t)p.C : G2_OOB;
        }
    }
    // weight chunk address = per-thread part (row, channel chunk; fixed within a slice) + wave-uniform tap part
    uint32_t sb_off[NJB];                                 // element offset without the tap term, or G2_OOB
    bool sb_h0[NJB], sb_w0[NJB];                          // may this chunk be used with the LAST tap row / column (k = 5: only even kh / kw exist there)
    auto sliceB = [&](int cc) {
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const int kc = cc * CHS + cchB[j] * VE;
            const bool ok = kc < p.KC && offB[j] != G2_OOB;
            if constexpr (MODE == TC_CONV) {
                uint32_t phh, r, pww, c;
                p.div_2c.divmod((uint32_t)kc, phh, r);
                p.div_c.divmod(r, pww, c);
                sb_off[j] = ok ? offB[j] + (phh * p.KW + pww) * p.C + c : G2_OOB;
                sb_h0[j] = 2 * (TAPS - 1) + (int)phh < p.KH; sb_w0[j] = 2 * (TAPS - 1) + (int)pww < p.KW;
            } else {
                const int ch = clsB[j] >> 1, cw = clsB[j] & 1;
                sb_off[j] = ok ? (uint32_t)((ch * p.KW + cw) * p.N) * (uint32_t)p.C + offB[j] + (uint32_t)kc : G2_OOB;
                sb_h0[j] = ch + 2 * p.HY < p.KH; sb_w0[j] = cw + 2 * p.HX < p.KW;
            }
        }
    };
    auto issueB = [&](int tap, int buf) {                 // weight tile of (slice set up by sliceB, tap) -> LDS byte offset buf
        const int ta = tap / TAPS, tb = tap - ta * TAPS;
        // conv form: kh = 2 ta + ph ; gather form: kh = ph + 2 (HY - ta)   (same for kw)
        const int th2 = MODE == TC_CONV ? 2 * ta : 2 * (p.HY - ta), tw2 = MODE == TC_CONV ? 2 * tb : 2 * (p.HX - tb);
        const uint32_t tapoff = MODE == TC_CONV ? (uint32_t)((th2 * p.KW + tw2) * p.C) : (uint32_t)((th2 * p.KW + tw2) * p.N) * (uint32_t)p.C;
        const bool last_h = MODE == TC_CONV ? ta == TAPS - 1 : ta == 0, last_w = MODE == TC_CONV ? tb == TAPS - 1 : tb == 0;
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
            const bool v = sb_off[j] != G2_OOB && (!last_h || sb_h0[j]) && (!last_w || sb_w0[j]);
            const uint32_t vo = v ? (sb_off[j] + tapoff) * ESZ : G2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vptr)(Bbase + buf + (wave * NJB + j) * 1024), 16, (int)vo, 0, 0, 0);
        }
    };

    // ---------------- main loop over (channel slice, tap) steps ----------------
    f32x16 acc[TM][TN];
    // Double-buffered pipeline, one barrier per step of TPS taps (32 MFMAs per wave): at the top of a step everything issued
    // during the previous step has had >= 2k cycles to land; the next step's weight tiles and a share of the next channel
    // slice of the slot range are issued right after the barrier.
    auto issue_step_B = [&](int ss, int stage) {          // the TPS weight tiles of step ss of the slice set up by sliceB
#pragma unroll
        for (int u = 0; u < TPS; ++u)
            if (ss * TPS + u < NT) issueB(ss * TPS + u, stage * BSTAGE + u * BTILE);
    };
    sliceB(0);
    issue_step_B(0, 0);
    TC_STAMP();

    // fragment-read offsets that do not depend on the tap
    int boff[4][TN];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < TN; ++j) boff[kk][j] = (tile_of(j) * 32 + lrow) * RB + (((kk * 2 + lgrp) ^ ((lrow >> 1) & 7)) << 4);
    int q0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) q0[i] = (wm * TM + i) * 32 + lrow;

    // LDS read addresses are computed ONCE (the address arithmetic of 48-80 fragment reads per step was ~2/3 of the step's VALU
    // instructions and the kernel is VALU-issue bound, not MFMA bound): a[tap][i][kk] for 2x2 taps (32 VGPRs), per (tap, i) row /
    // swizzle terms otherwise.  Slice / stage parities are template arguments (the slice loop is unrolled by two) so that the
    // buffer bases fold into the 16-bit offset field of ds_read_b128.
    constexpr bool PRE = TAPS == 2;
    uint32_t aaddr[PRE ? NT : 1][TM][4];
    if constexpr (PRE) {
#pragma unroll
        for (int tap = 0; tap < NT; ++tap)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int q = q0[i] + (tap / TAPS) * p.GW + (tap % TAPS);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) aaddr[tap][i][kk] = (uint32_t)(q * RB + ((((kk * 2 + lgrp) ^ ((q >> 1) & 7))) << 4));
            }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < TN; ++j) boff[kk][j] += 2 * ASTAGE;          // weight stages follow the two slot stages

    // ---------------- direct-epilogue store addresses + ReluGrad-mask prefetch, run in the LAST step of the main loop ----------------
    // The address arithmetic (one slot decode per subtile row) moves under the last step's MFMAs, and the 128-byte lines of the mask
    // the epilogue will read (128 KB per block from HBM: ~3.6k cycles in front of the stores of the input-gradient layers) are touched
    // there with one dword load each, so that the epilogue's 16-byte mask loads hit the cache.  (Loading the mask vectors themselves
    // that early would hold 32 registers across the whole slice loop.)
    const T* __restrict__ maskp = (const T*)p.mask;
    constexpr bool PK = ESZ == 2;                          // 16-byte units of 8 bf16; fp32: a 4-channel group already is 16 bytes
    constexpr int NU = PK ? 2 : 4;                         // store units per subtile and lane
    const int lo = p.relu ? 0 : (int)0x80000000;          // ReLU as an integer max: negative floats are negative integers
    uint32_t uoff[TM][TN][NU]; bool uok[TM][TN];
    uint32_t pf[TM][TN];
    auto last_step_prep = [&]() {
        if (!p.direct_epilogue) return;
    #pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int P = P0 + (wm * TM + i) * 32 + lrow;
            const bool pin = P < p.MP;
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(pin ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            // element offset of the slot's first output pixel, and which of its (up to 4) pixels exist
            const int oy0 = MODE == TC_CONV ? (int)gy : 2 * (int)gy, ox0 = MODE == TC_CONV ? (int)gx : 2 * (int)gx;
            const uint32_t pbase = ((b * p.OH + oy0) * p.OW + ox0) * p.N;        // < 2^31 elements (host check)
            const bool vy0 = pin && oy0 < p.OH, vx0 = ox0 < p.OW, vy1 = pin && oy0 + 1 < p.OH, vx1 = ox0 + 1 < p.OW;
    #pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int ne0 = n0 + tile_of(j) * 32;     // wave-uniform; a 32-wide output tile never straddles a parity class
                int sub = ne0;                             // wave-uniform element offset of the tile inside the slot's pixels
                bool ok = vy0 && vx0;
                if constexpr (MODE == TC_GATHER) {
                    const int cls = (int)p.div_n.div((uint32_t)ne0);
                    sub = ((cls >> 1) * p.OW + (cls & 1)) * p.N + ne0 - cls * p.N;
                    ok = ((cls >> 1) ? vy1 : vy0) && ((cls & 1) ? vx1 : vx0);
                }
                uok[i][j] = ok && ne0 < p.NE;
    #pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const uint32_t ch = PK ? 16 * u + 8 * lgrp : 4 * lgrp + 8 * u;
                    uoff[i][j][u] = uok[i][j] ? pbase + (uint32_t)sub + ch : 0u;
                }
            }
        }
        if (maskp && p.mask_prefetch) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) pf[i][j] = *(const uint32_t*)(maskp + uoff[i][j][0]);     // one touch per 128-byte line
        }
    };
    auto slice_body = [&](int cc, auto par_c) {
        constexpr int PAR = decltype(par_c)::value;        // cc & 1
        const bool more_a = cc + 1 < NCC;
        if (more_a) sliceA(cc + 1);
#pragma unroll
        for (int ss = 0; ss < NSS; ++ss) {
            const int stage = (PAR * NSS + ss) & 1;       // literal after unrolling: (global step index) & 1
            __syncthreads();                              // this step's tiles have landed (vmcnt(0) + barrier); the other stage is free
            if (cc == 0) TC_STAMP();
            if (more_a) {
#pragma unroll
                for (int i = 0; i < NIA; ++i)
                    if (i % NSS == ss) issueA(PAR ^ 1, i);
            }
            if (ss + 1 < NSS) issue_step_B(ss + 1, stage ^ 1);
            else if (more_a) { sliceB(cc + 1); issue_step_B(0, stage ^ 1); }
            else last_step_prep();                        // last step of the tile
#pragma unroll
            for (int u = 0; u < TPS; ++u) {
                const int tap = ss * TPS + u;
                if (tap >= NT) continue;                  // literal after unrolling: padded tap of a 3x3 tap set
                const int ta = tap / TAPS, tb = tap % TAPS;
                const int aconst = PAR * ASTAGE, bconst = stage * BSTAGE + u * BTILE;      // literals
                int qrow[TM], qx[TM];
                if constexpr (!PRE) {
                    const int delta = ta * p.GW + tb;
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const int q = q0[i] + delta;
                        qrow[i] = q * RB; qx[i] = (q >> 1) & 7;
                    }
                }
                // k = 5 (3x3 taps): the last tap row / column only reaches the even kernel rows / columns, so a 32-output tile
                // whose rows all have an odd kh (kw) there meets an all-zero weight tile: skip it (wave-uniform, per tile)
                bool live[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) live[j] = true;
                if constexpr (TAPS == 3 && MODE == TC_GATHER) {
                    if (p.N >= 32) {                      // a 32-wide tile lies inside one parity class
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const int cls = (int)p.div_n.div((uint32_t)(n0 + tile_of(j) * 32));
                            live[j] = !((ta == 0 && (cls >> 1) + 2 * p.HY >= p.KH) || (tb == 0 && (cls & 1) + 2 * p.HX >= p.KW));
                        }
                    }
                }
                bool any_live = false;
#pragma unroll
                for (int j = 0; j < TN; ++j) any_live = any_live || live[j];
                if (!any_live) continue;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    freg af[TM], bf[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        if constexpr (PRE) af[i] = *(const freg*)(lds + aaddr[tap][i][kk] + aconst);
                        else af[i] = *(const freg*)(lds + qrow[i] + ((((kk * 2 + lgrp) ^ qx[i])) << 4) + aconst);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[j] = *(const freg*)(lds + boff[kk][j] + bconst);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            if (live[j]) Frag<T>::mma(bf[j], af[i], acc[i][j]);   // D[row = output channel][col = position]
                }
            }
        }
    };
    // the accumulators start at the bias of their output channel (loaded at the top of the kernel)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        f32x16 b16;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < 4; ++t) b16[4 * q + t] = bias4[j][q][t];
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[i][j] = b16;
    }
    for (int cc = 0; cc < NCC; cc += 2) {
        slice_body(cc, std::integral_constant<int, 0>());
        if (cc + 1 < NCC) slice_body(cc + 1, std::integral_constant<int, 1>());
    }

    if (p.direct_epilogue) {
        // ---------------- direct epilogue (bias is already in the accumulators) ----------------
        // Per lane one position (col lrow of each 32-position subtile); the 4-channel groups g of a 32-output tile alternate between
        // the two half-waves (lane: channels 8g + 4 lgrp ..+3).  One v_permlane32_swap per dword turns each PAIR of groups into 16
        // contiguous bytes per lane (lower half: channels 16m .. 16m+7, upper half: 16m+8 .. 16m+15): two 16-byte stores per subtile.
        // (Measured: regrouping the lanes further so that a quad writes one whole 64-byte line changes nothing -- with the stores
        // removed altogether the epilogue is 10 % shorter; it is bound by its own VALU work at two waves per SIMD.  Hence: no runtime
        // selects per value (ReLU is a max against 0 or -inf), one branch for the whole ReluGrad mask, addresses = one pixel base per
        // subtile row + wave-uniform class / channel terms.)
        // Phase 1 computes every store address and issues ALL ReluGrad-mask loads; phase 2 only converts and stores.  (A load between the
        // stores makes the wave wait for the acknowledgement of the stores before it -- vmcnt counts both.)
        TC_STAMP();
        PackN<uint32_t, 4> umk[TM][TN][NU];
        if (maskp && p.mask_prefetch) {
            uint32_t sink = 0;                            // the prefetch results are consumed (and dropped) here so that their registers die
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) sink ^= pf[i][j];
            if (sink == 0x7fc00001u && p.dbg == 99) ((volatile uint32_t*)p.out)[0] = sink;      // never true: keeps the loads alive
        }
        if (maskp) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int u = 0; u < NU; ++u) umk[i][j][u] = *(const PackN<uint32_t, 4>*)(maskp + uoff[i][j][u]);   // offset 0 is always readable
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v[4][4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {               // ReLU as ONE v_max_i32 on the bit pattern (fmaxf costs a canonicalising max on top)
                        const float a = acc[i][j][4 * g + t];   // (a copy: __builtin_bit_cast applied to the vector element itself reads element 0)
                        const int bits = __builtin_bit_cast(int, a);
                        v[g][t] = __builtin_bit_cast(float, bits > lo ? bits : lo);
                    }
                if constexpr (PK) {
                    uint32_t w[4][2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const PackN<T, 4> pk = pack4<T>(v[g]);
                        w[g][0] = (uint32_t)pk.v[0] | ((uint32_t)pk.v[1] << 16); w[g][1] = (uint32_t)pk.v[2] | ((uint32_t)pk.v[3] << 16);
                    }
#pragma unroll
                    for (int x = 0; x < 2; ++x)                 // register bit g0 <-> half-wave
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            auto r = __builtin_amdgcn_permlane32_swap(w[2 * x][d], w[2 * x + 1][d], false, false);
                            w[2 * x][d] = r[0]; w[2 * x + 1][d] = r[1];
                        }
                    if (maskp) {                                // bf16 > 0  <=>  signed 16-bit pattern > 0: 0 / 0xffff per half by packed integer ops
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int d = 0; d < 2; ++d) {
                                typedef short s16x2 __attribute__((ext_vector_type(2)));
                                const s16x2 mk = __builtin_bit_cast(s16x2, umk[i][j][g >> 1].v[2 * (g & 1) + d]);
                                const s16x2 one = __builtin_elementwise_min(__builtin_elementwise_max(mk, (s16x2){0, 0}), (s16x2){1, 1});   // v_pk_max_i16, v_pk_min_i16
                                w[g][d] &= __builtin_bit_cast(uint32_t, (s16x2){0, 0} - one);                                           // 0xffff where the mask is positive
                            }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u)                 // unit u = registers (2u, 2u + 1)
                        if (uok[i][j] && p.dbg != 3) *(PackN<uint32_t, 4>*)((T*)p.out + uoff[i][j][u]) = PackN<uint32_t, 4>{{w[2 * u][0], w[2 * u][1], w[2 * u + 1][0], w[2 * u + 1][1]}};
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        PackN<T, 4> o = pack4<T>(v[u]);
                        if (maskp) {
                            const PackN<T, 4> mk = __builtin_bit_cast(PackN<T, 4>, umk[i][j][u]);
#pragma unroll
                            for (int t = 0; t < 4; ++t) o.v[t] = Elem<T>::to_f32(mk.v[t]) > 0.f ? o.v[t] : zero_of<T>();
                        }
                        if (uok[i][j]) *(PackN<T, 4>*)((T*)p.out + uoff[i][j][u]) = o;
                    }
                }
            }
        }
        TC_STAMP();
        return;
    }
    TC_STAMP();
    // ---------------- epilogue: accumulators -> LDS [position][output] -> full-line coalesced 16-byte stores ----------------
    // (a row-per-lane store of 8 bytes per lane at a 64..256-byte stride is store-ISSUE bound: 7-17k cycles per block measured
    //  against ~1.8k per MFMA step; staged through LDS every store instruction writes 1 KB of whole 128-byte lines)
    constexpr int PITCH = BNE * ESZ + 16;                 // +16 B: the 32 positions of a fragment column hit 16 distinct bank groups
    constexpr int CP = BNE * ESZ / 16;                    // 16-byte chunks per position
    unsigned char* const stg = lds;
    int2* const pinfo = (int2*)(lds + BMT * PITCH);        // per position: element offset of its first output pixel, validity flags
    static_assert(BMT * PITCH + BMT * 8 <= 2 * ASTAGE + TC_NBUF * BSTAGE, "epilogue staging fits the pipeline buffers");
    __syncthreads();                                      // every wave is done reading the last step's tiles
    if (tid < BMT) {
        const int P = P0 + tid;
        uint32_t g, gx, b, gy;
        p.div_gw.divmod((uint32_t)(P < p.MP ? P : 0), g, gx);
        p.div_g.divmod(g, b, gy);
        int2 pi;
        if constexpr (MODE == TC_CONV) {
            pi.x = (int)(((b * p.OH + gy) * p.OW + gx) * p.N);
            pi.y = (P < p.MP && (int)gy < p.OH && (int)gx < p.OW) ? 15 : 0;
        } else {
            const int oy = 2 * (int)gy, ox = 2 * (int)gx;
            pi.x = (int)(((b * p.OH + oy) * p.OW + ox) * p.N);
            const int vh0 = oy < p.OH, vh1 = oy + 1 < p.OH, vw0 = ox < p.OW, vw1 = ox + 1 < p.OW;
            pi.y = P < p.MP ? ((vh0 & vw0) | ((vh0 & vw1) << 1) | ((vh1 & vw0) << 2) | ((vh1 & vw1) << 3)) : 0;   // bit = class ph*2+pw
        }
        pinfo[tid] = pi;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pos = (wm * TM + i) * 32 + lrow;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cn = tile_of(j) * 32 + 4 * lgrp + 8 * q;          // output column inside the tile
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = acc[i][j][4 * q + t];
                if (p.relu) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
                }
                *(PackN<T, 4>*)(stg + pos * PITCH + cn * ESZ) = pack4<T>(v);
            }
        }
    }
    __syncthreads();
    TC_STAMP();
    constexpr int NCH = BMT * CP / TC_NT;                 // chunks per thread
    constexpr int GRP = NCH >= 8 ? 8 : NCH;               // chunks whose mask loads are in flight together
#pragma unroll
    for (int k0 = 0; k0 < NCH; k0 += GRP) {
        long long off[GRP];
        bool ok[GRP];
        PackN<T, VE> mk[GRP];
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int id = tid + (k0 + u) * TC_NT;
            const int pos = id / CP, c16 = id % CP;
            const int2 pi = pinfo[pos];
            const int ne = n0 + c16 * VE;
            int n = ne, sub = 0, cls = 0;
            if constexpr (MODE == TC_GATHER) {
                cls = (int)p.div_n.div((uint32_t)ne);
                n = ne - cls * p.N;
                sub = ((cls >> 1) * p.OW + (cls & 1)) * p.N;
            }
            ok[u] = ne < p.NE && ((pi.y >> cls) & 1);
            off[u] = ok[u] ? (long long)pi.x + sub + n : 0;
            if (maskp) mk[u] = *(const PackN<T, VE>*)(maskp + off[u]);       // offset 0 is always readable
        }
#pragma unroll
        for (int u = 0; u < GRP; ++u) {
            const int id = tid + (k0 + u) * TC_NT;
            const int pos = id / CP, c16 = id % CP;
            PackN<T, VE> o = *(const PackN<T, VE>*)(stg + pos * PITCH + c16 * 16);
            if (maskp) {
#pragma unroll
                for (int t = 0; t < VE; ++t) o.v[t] = Elem<T>::to_f32(mk[u].v[t]) > 0.f ? o.v[t] : zero_of<T>();
            }
            if (ok[u]) *(PackN<T, VE>*)((T*)p.out + off[u]) = o;
        }
    }
    TC_STAMP();
#undef TC_STAMP
}

}  // namespace mi
