// tapwgrad_tile.hpp — weight gradients of the stride-2 conv / transposed-conv layers on raw-staged slot tiles (bf16, gfx950).
//
// Same slot formulation as tapconv_tile.hpp.  With A[P][kc] the slot stream (conv form: the 2x2 pixel block, 4C channels;
// gather form: the pixel, C channels) and D[P][ne] the output-side gradient of position P (conv form: dy[b,gy,gx,:], N
// columns; gather form: dy of the four output parities, 4N columns):
//     dW'[tap][kc][ne] = sum_P A[P + ta*GW + tb][kc] * D[P][ne]
// i.e. a GEMM whose reduction index is the POSITION.  A block owns a [KT*32 channels] x [NTB*32 outputs] x [all taps] slab of
// dW' (4..8 accumulator tiles per wave) and walks a contiguous range of positions, 128 per step:
//   * the slot range [P, P+128+halo) x KCB channels and the 128 x NEB gradient rows are staged once per step by LDS-DMA
//     (zero fill outside the images / past the range through the buffer range check); every tap reuses the staged slots,
//   * MFMA operands need 8 consecutive POSITIONS per lane: they are read from the position-major LDS tiles with the hardware
//     transpose read (ds_read_b64_tr_b16); the tap shift is a row offset; all per-lane offsets are precomputed once,
//   * rows are unpadded (DMA writes lane-linear); the 4 rows x 64 B a half-wave transpose read touches are spread over the
//     four bank quarters by an XOR on the 16-byte chunk index applied on the DMA source side,
//   * one block per CU: the reduction over positions is split ~256 ways in total, 4x fewer fp32 atomics than wgrad_kernel.
// fp32 (parity mode) stays on wgrad_kernel: the transpose read is a 16-bit instruction.
#pragma once
#include "tapconv_tile.hpp"
#include "wgrad_tile.hpp"

namespace mi {

constexpr int TW_BP = 128;           // positions per step
constexpr int TW_NT = 512;           // threads (8 waves)
constexpr int TW_MAXPAIR = 32;

struct TapWgradParams {
    const void* a; uint32_t a_bytes;
    const void* d; uint32_t d_bytes;
    int B, IH, IW, C;                // slot-side tensor
    int OH, OW, N;                   // gradient tensor dy[B,OH,OW,N]
    int KH, KW;
    int GH, GW, HY, HX;
    int KC, NE, MP;
    int nkb;                         // channel blocks (blockIdx.y = nb * nkb + kb)
    int pos_per_split;               // multiple of TW_BP
    int npairs;                      // (tap, output tile) pairs of a block, dealt round-robin to the 8 waves
    unsigned char pair_tap[TW_MAXPAIR], pair_nt[TW_MAXPAIR];
    FastDiv div_g, div_gw, div_n, div_2c, div_c;
    float* out;
};

// swizzle of the 16-byte chunk index by LDS row so that 4 consecutive rows x 64 B fall into 4 different bank quarters
template <int CPR> __device__ __forceinline__ int tw_swz(int row) {
    return CPR == 16 ? (row & 3) << 2 : ((row >> 1) & 1) << 2;
}

// MODE: TC_CONV | TC_GATHER;  KT: 32-channel tiles per block (KCB = 32 KT);  NTB: 32-output tiles per block (NEB = 32 NTB);
// PPW: (tap, output tile) pairs per wave (accumulators: PPW * KT tiles)
template <int MODE, int TAPS, int KT, int NTB, int PPW>
__global__ __launch_bounds__(TW_NT) void tapwgrad_kernel(const TapWgradParams p) {
    typedef bf16_t T;
    constexpr int ESZ = 2, VE = 8;
    constexpr int KCB = 32 * KT, NEB = 32 * NTB;
    constexpr int PA = KCB * ESZ, PD = NEB * ESZ;        // LDS row pitch of the slot tile / gradient tile (128 | 256 B)
    constexpr int CPA = PA / 16, CPD = PD / 16;          // 16-byte chunks per row
    constexpr int SPI_A = 1024 / PA, SPI_D = 1024 / PD;  // rows per DMA instruction
    constexpr int MAXSLOT = TW_BP + TC_MAXHALO;          // 224
    constexpr int NIA = (MAXSLOT / SPI_A + 7) / 8;       // slot-tile DMA instructions per wave (upper bound)
    constexpr int NID = TW_BP / SPI_D / 8;               // gradient-tile DMA instructions per wave
    constexpr int ASTAGE = MAXSLOT * PA, DSTAGE = TW_BP * PD, STAGE = ASTAGE + DSTAGE;
    static_assert(NID >= 1 && 2 * STAGE <= 160 * 1024, "tile config");

    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kb = blockIdx.y % p.nkb, nb = blockIdx.y / p.nkb;
    const int kc0 = kb * KCB, ne0 = nb * NEB;
    const int Pbeg = blockIdx.x * p.pos_per_split;
    const int Pend = min(p.MP, Pbeg + p.pos_per_split);
    if (Pbeg >= Pend) return;
    const int nsteps = (Pend - Pbeg + TW_BP - 1) / TW_BP;
    const int halo = (TAPS - 1) * p.GW + TAPS - 1;
    const int ninstrA = (TW_BP + halo + SPI_A - 1) / SPI_A;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.d, 0, (int)p.d_bytes, 0x00020000);

    // ---------------- DMA roles ----------------
    // slot tile: instruction t = wave + 8 i fills rows SPI_A t ..; lane -> row SPI_A t + lane / CPA, physical chunk lane % CPA
    const int rA = lane / CPA, cA = (lane % CPA) ^ tw_swz<CPA>(rA);        // SPI_A t is a multiple of 4: the swizzle term only sees rA
    const int rD = lane / CPD, cD = (lane % CPD) ^ tw_swz<CPD>(rD);
    // channel part of the source address (fixed for the whole kernel)
    uint32_t a_koff; int a_sub = 0; bool a_kok;
    {
        const int kc = kc0 + cA * VE;
        a_kok = kc < p.KC;
        if constexpr (MODE == TC_CONV) {
            uint32_t phh, r, pww, c;
            p.div_2c.divmod((uint32_t)(a_kok ? kc : 0), phh, r);
            p.div_c.divmod(r, pww, c);
            a_koff = (phh * p.IW * p.C + r) * ESZ; a_sub = (int)(phh * 2 + pww);
        } else a_koff = (uint32_t)kc * ESZ;
    }
    uint32_t d_koff; int d_cls = 0; bool d_kok;
    {
        const int ne = ne0 + cD * VE;
        d_kok = ne < p.NE;
        if constexpr (MODE == TC_CONV) d_koff = (uint32_t)ne * ESZ;
        else {
            uint32_t cls, n;
            p.div_n.divmod((uint32_t)(d_kok ? ne : 0), cls, n);
            d_cls = (int)cls;
            d_koff = ((((cls >> 1) * p.OW + (cls & 1)) * p.N) + n) * ESZ;
        }
    }
    auto issue = [&](int step, int buf) {
        const int Ps = Pbeg + step * TW_BP;
        unsigned char* As = lds + buf * STAGE;
        unsigned char* Ds = As + ASTAGE;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int t = wave + 8 * i;
            if (t >= ninstrA) break;                      // wave-uniform
            const int P = Ps + SPI_A * t + rA;
            const bool ok = P < p.MP && a_kok;
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(ok ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            uint32_t vo;
            if constexpr (MODE == TC_CONV) {
                const int y = 2 * (int)gy + (a_sub >> 1), x = 2 * (int)gx + (a_sub & 1);
                const bool v = ok && y < p.IH && x < p.IW;
                vo = v ? (((b * p.IH + 2 * gy) * p.IW + 2 * gx) * p.C) * ESZ + a_koff : G2_OOB;
            } else {
                const int iy = (int)gy - p.HY, ix = (int)gx - p.HX;
                const bool v = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                vo = v ? (((b * p.IH + iy) * p.IW + ix) * p.C) * ESZ + a_koff : G2_OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(As + t * 1024), 16, (int)vo, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NID; ++i) {
            const int t = wave + 8 * i;
            const int P = Ps + SPI_D * t + rD;
            const bool ok = P < Pend && d_kok;
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(ok ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            uint32_t vo;
            if constexpr (MODE == TC_CONV) {
                const bool v = ok && (int)gy < p.OH && (int)gx < p.OW;
                vo = v ? (((b * p.OH + gy) * p.OW + gx) * p.N) * ESZ + d_koff : G2_OOB;
            } else {
                const int oy = 2 * (int)gy + (d_cls >> 1), ox = 2 * (int)gx + (d_cls & 1);
                const bool v = ok && oy < p.OH && ox < p.OW;
                vo = v ? (((b * p.OH + 2 * gy) * p.OW + 2 * gx) * p.N) * ESZ + d_koff : G2_OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_vptr)(Ds + t * 1024), 16, (int)vo, 0, 0, 0);
        }
    };

    // ---------------- this wave's (tap, output tile) pairs and their per-lane transpose-read offsets ----------------
    // transpose read (see tr_fragment in wgrad_tile.hpp): lane l supplies row r0 + (l>>5)*8 + ((l&15)>>2) (+4 for the high half),
    // element column e0 + ((l>>4)&1)*16 + (l&3)*4, and receives 8 consecutive rows of column e0 + (l & 31).
    const int trow = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int tcol = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    int pr_tap[PPW], pr_nt[PPW];
    bool pr_on[PPW];
    uint32_t aoff[PPW][KT], doff[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int pi = wave + 8 * q;
        pr_on[q] = pi < p.npairs;
        pr_tap[q] = pr_on[q] ? p.pair_tap[pi] : 0;
        pr_nt[q] = pr_on[q] ? p.pair_nt[pi] : 0;
        const int ta = pr_tap[q] / TAPS, tb = pr_tap[q] % TAPS;
        const int row = trow + ta * p.GW + tb;            // + 16 per k-step and + 4 for the high half keep (row & 3)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int col = kt * 32 + tcol;
            aoff[q][kt] = (uint32_t)(row * PA + ((((col >> 3) ^ tw_swz<CPA>(row))) << 4) + (col & 7) * ESZ);
        }
        const int col = pr_nt[q] * 32 + tcol;
        doff[q] = (uint32_t)(ASTAGE + trow * PD + ((((col >> 3) ^ tw_swz<CPD>(trow))) << 4) + (col & 7) * ESZ);
    }

    f32x16 acc[PPW][KT];
#pragma unroll
    for (int q = 0; q < PPW; ++q)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][kt][r] = 0.f;

    typedef __attribute__((address_space(3))) s16x4* lds_v4;
    issue(0, 0);
    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        __syncthreads();                                  // this step's tiles have landed; the other stage is free
        if (step + 1 < nsteps) issue(step + 1, cur ^ 1);
        const uint32_t sbase = (uint32_t)(cur * STAGE);
#pragma unroll
        for (int ks = 0; ks < TW_BP / 16; ++ks) {
#pragma unroll
            for (int q = 0; q < PPW; ++q) {
                if (!pr_on[q]) continue;                  // wave-uniform
                const uint32_t dof = sbase + doff[q] + ks * 16 * PD;
                const s16x4 dlo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + dof));
                const s16x4 dhi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + dof + 4 * PD));
                u16x8 df;
#pragma unroll
                for (int e = 0; e < 4; ++e) { df[e] = (unsigned short)dlo[e]; df[4 + e] = (unsigned short)dhi[e]; }
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const uint32_t aof = sbase + aoff[q][kt] + ks * 16 * PA;
                    const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + aof));
                    const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + aof + 4 * PA));
                    u16x8 af;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { af[e] = (unsigned short)alo[e]; af[4 + e] = (unsigned short)ahi[e]; }
                    // conv form: D[row = channel][col = output] (HWIO: outputs contiguous); gather form: D[row = output][col = channel]
                    if constexpr (MODE == TC_CONV)
                        acc[q][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, df), acc[q][kt], 0, 0, 0);
                    else
                        acc[q][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, df), __builtin_bit_cast(bf16x8, af), acc[q][kt], 0, 0, 0);
                }
            }
        }
    }

    // ---------------- accumulate into dW (fp32 atomics; a wave's 32 lanes of one register hit 128 contiguous bytes) ----------------
    const int lcol = lane & 31, lgrp = lane >> 5;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        if (!pr_on[q]) continue;
        const int ta = pr_tap[q] / TAPS, tb = pr_tap[q] % TAPS;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2) + 4 * lgrp;
                long long idx; bool ok;
                if constexpr (MODE == TC_CONV) {
                    const int kc = kc0 + kt * 32 + rr, ne = ne0 + pr_nt[q] * 32 + lcol;
                    uint32_t phh, rem, pww, c;
                    p.div_2c.divmod((uint32_t)kc, phh, rem);
                    p.div_c.divmod(rem, pww, c);
                    const int kh = 2 * ta + (int)phh, kw = 2 * tb + (int)pww;
                    ok = kc < p.KC && ne < p.NE && kh < p.KH && kw < p.KW;
                    idx = ((long long)(kh * p.KW + kw) * p.C + c) * p.N + ne;
                } else {
                    const int ne = ne0 + pr_nt[q] * 32 + rr, kc = kc0 + kt * 32 + lcol;
                    uint32_t cls, n;
                    p.div_n.divmod((uint32_t)ne, cls, n);
                    const int kh = (int)(cls >> 1) + 2 * (p.HY - ta), kw = (int)(cls & 1) + 2 * (p.HX - tb);
                    ok = kc < p.KC && ne < p.NE && kh < p.KH && kw < p.KW;
                    idx = ((long long)(kh * p.KW + kw) * p.N + n) * p.C + kc;
                }
                if (ok) atomicAdd(&p.out[idx], acc[q][kt][r]);
            }
        }
    }
}

}  // namespace mi
