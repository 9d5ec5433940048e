// tapconv_persist.hpp — persistent-block form of tapconv_kernel (tapconv_tile.hpp), direct epilogue only.  EXPERIMENTAL, off by
// default (mi_set_tuning key 8 / MI355_TAP_PERSIST=1).
//
// One tapconv block spends more than half of its life on serial latencies (address set-up, the first tiles landing, the store tail) and with
// ~150 KB of LDS there is no second block on the CU to hide them (finding 7).  Here a block walks a contiguous range of
// tiles: the first DMA loads of tile t+1 are issued before the stores of tile t, and the tile-invariant set-up (LDS fragment addresses,
// weight roles) is paid once.  A first prototype of this loop was 8-20 % faster per layer than the same code run one tile per block, but lost
// the gain to register pressure: with the LDS-staged epilogue inside the loop, hipcc hoisted ~110 VGPRs of tile-invariant address terms and
// spilled 240-290 SGPRs.  This version keeps only what the main loop holds anyway across tiles: the bias is re-read per tile, the staged
// epilogue is not in the loop (launches that need it use tapconv_kernel), the trace stamps are gone.
// Roles, main loop and conversion code are tapconv_kernel's; see tapconv_tile.hpp for the layout comments.
#pragma once
#include "tapconv_tile.hpp"

namespace mi {

template <typename T, int MODE, int BNE, int TAPS, int BMT, int MAXHALO>
__global__ __launch_bounds__(BMT * 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void tapconv_persist_kernel(const TapParams p) {
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
    // per-tile opaque copies for everything OUTSIDE the main loop (bias loads, store addresses, conversion): the tile loop must not keep
    // their tile-invariant address terms in registers across the main loop
    int lrow_e = lrow, lgrp_e = lgrp;

    // Block x walks the contiguous tile range [tile_lo, tile_hi) of the slot grid (the halo slots two neighbouring tiles share are
    // re-read from this XCD's L2).  The grid is sized by the host to what is resident at once.
    const int ntile = (p.MP + BMT - 1) / BMT;
    const int tile_lo = (int)(((long long)blockIdx.x * ntile) / gridDim.x), tile_hi = (int)(((long long)(blockIdx.x + 1) * ntile) / gridDim.x);
    if (tile_lo >= tile_hi) return;                       // block-uniform
    int P0 = tile_lo * BMT;
    const int n0 = blockIdx.y * BNE;
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
    auto issue_first = [&]() {                            // the tile at P0: slice 0 of its slot range, instruction by instruction as the rows are
        sliceA(0);                                        // decoded, and the weight tiles of its step 0
#pragma unroll
        for (int i = 0; i < NIA; ++i) { slotA(i); issueA(0, i); }
        sliceB(0);
        issue_step_B(0, 0);
    };
    issue_first();

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
    auto last_step_prep = [&]() {
    #pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int P = P0 + (wm * TM + i) * 32 + lrow_e;
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
                    const uint32_t ch = PK ? 16 * u + 8 * lgrp_e : 4 * lgrp_e + 8 * u;
                    uoff[i][j][u] = uok[i][j] ? pbase + (uint32_t)sub + ch : 0u;
                }
            }
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
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        asm volatile("" : "+v"(lrow_e), "+v"(lgrp_e));
        // the accumulators start at the bias of their output channel (register r of a lane: channel 4 lgrp + 8 (r >> 2) + (r & 3) of a 32-output
        // tile).  Re-read per tile (L2-hot) instead of held in 16 TN registers across the loop; the loads queue behind the DMA loads of this
        // tile, which the first barrier waits for anyway.
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x16 b16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 bb = f32x4{0.f, 0.f, 0.f, 0.f};
                const int ne = n0 + tile_of(j) * 32 + 4 * lgrp_e + 8 * q;
                if (p.bias && ne < p.NE) {
                    int nb = ne;
                    if constexpr (MODE == TC_GATHER) nb -= (int)p.div_n.div((uint32_t)ne) * p.N;
                    bb = *(const f32x4*)(p.bias + nb);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) b16[4 * q + t] = bb[t];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = b16;
        }
        for (int cc = 0; cc < NCC; cc += 2) {
            slice_body(cc, std::integral_constant<int, 0>());
            if (cc + 1 < NCC) slice_body(cc + 1, std::integral_constant<int, 1>());
        }
        // ---------------- direct epilogue of this tile, with the next tile's first loads issued in front of its stores ----------------
        // Order: barrier (every wave is done reading the stages) -> ReluGrad-mask loads of this tile -> DMA loads of the next tile (slice 0
        // of its slot range, weight tiles of its step 0) -> convert + store.  Loads return in order, so the mask loads must be IN FRONT of
        // the DMA loads: the conversion then waits for the masks only, while the next tile's ~50 KB are in flight under the stores.
        const bool more = tile + 1 < tile_hi;             // block-uniform
        PackN<uint32_t, 4> umk[TM][TN][NU];
        if (more) __syncthreads();
        if (maskp) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int u = 0; u < NU; ++u) umk[i][j][u] = *(const PackN<uint32_t, 4>*)(maskp + uoff[i][j][u]);   // offset 0 is always readable
        }
        if (more) { P0 += BMT; issue_first(); }
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
                        if (uok[i][j]) *(PackN<uint32_t, 4>*)((T*)p.out + uoff[i][j][u]) = PackN<uint32_t, 4>{{w[2 * u][0], w[2 * u][1], w[2 * u + 1][0], w[2 * u + 1][1]}};
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        PackN<T, 4> o = pack4<T>(v[u]);
                        if (maskp) {
                            const PackN<T, 4> mk = __builtin_bit_cast(PackN<T, 4>, umk[i][j][u]);
#pragma unroll
                            for (int t = 0; t < 4; ++t) o.v[t] = Elem<T>::to_f32(mk.v[t]) > 0.f ? o.v[t] : (T)0;
                        }
                        if (uok[i][j]) *(PackN<T, 4>*)((T*)p.out + uoff[i][j][u]) = o;
                    }
                }
            }
        }
    }
}

}  // namespace mi
