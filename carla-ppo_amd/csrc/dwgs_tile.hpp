// dwgs_tile.hpp — dense filter gradient WITHOUT LDS (round 5, VERDICT r04 item 2b: the latent layers of the ConvVAE -- heads [6144 x 128], dense1 [64 x 6144] -- and the
// small layers of the MlpVAE; tf.gradients of MatMul, vae/models.py:97-98,259 behind :142):
//   dW[k][n] (+)= sum_m a[m][k] dy[m][n]      a [M, K], dy [M, N] bf16 row-major, M = the minibatch
// What the first-generation kernel (wgrad_tile.hpp) made of these shapes: 64 KB of LDS per block, row splits whose slabs a second launch sums -- 12 us alone, and 31-44 us
// each at the serial end of the backward pass, where its blocks find no CU next to the 147 KB-LDS filter-gradient blocks of the other queue (profiles/r04_a).
// Here a tile of 64 x 64 elements of dW is owned by ONE block of 1, 2 or 4 waves that split the rows between them: no operand staging in LDS, no slab, no second launch
// (16 KB of LDS only for the waves' final in-order sum).
// The reduction index m is the slow index of both operands, so the MFMA fragments (8 consecutive m per lane) are gathered: a lane loads the DWORD (columns 2 i, 2 i + 1)
// of each of its 8 rows -- 32 lanes x 4 bytes = one full 128-byte line per row -- and two v_perm_b32 per row pair split the low / high halves into the fragment of the
// EVEN column and the fragment of the ODD column.  A fragment pair (fi, fj) therefore produces the dW elements (k0 + 2 r + fi, n0 + 2 c + fj): the two column parities
// of a lane are adjacent floats and leave as one 8-byte store (256 contiguous bytes per row of 32 lanes).  16 loads + 16 perms feed 4 MFMAs (+ 2 for the bias row);
// DWGS_DEPTH steps of raw dwords are in flight (64 registers), the compiler's vmcnt counting "all but the newest 16 (DEPTH - 1)" for the step it packs.
// The result is stored or added in place (each element is owned by exactly one block, its waves summed in a fixed order: bitwise reproducible either way); BiasAddGrad = one more MFMA per step and column
// parity against an all-ones operand in the waves of the first k tile.
#pragma once
#include "common.hpp"

namespace mi {

struct DwgsParams {
    const bf16_t* a; const bf16_t* dy;
    float* out; float* dbias;
    int M, K, N;
    int KT, NT;                      // tiles of 64 along k and n
    int overwrite;                   // != 0: out = ..., 0: out += ...
    // Row splits INSIDE the block: blockDim.x / 64 = 1, 2 or 4 waves share a tile, wave z sums the rows [z M / ws, (z + 1) M / ws); waves 0 .. ws - 2 add their tiles into a
    // 16 KB LDS tile one after the other (fixed order: ((w0 + w1) + w2) + w3), the last wave adds that sum to its registers and stores.  Why split at all: one wave per tile
    // is a chain of M / 16 dependent load rounds, and next to the HBM-saturating kernels at the end of the backward pass a round takes microseconds (96 waves x 32 rounds:
    // 64 us in the step, gpurun_out/timeline_r05c.md).  Why not across blocks through slabs (the first row-split form of this round): 25 MB of slabs written and read again by
    // the pass's small-reduce launch for 5 MB of gradients.
};

constexpr int DWGS_DEPTH = 4;

// rows (lo, hi) of one column pair -> the fragment dwords of the even column (low halves) and of the odd column (high halves): bf16 k-values (row lo, row hi)
__device__ __forceinline__ void dwgs_split(uint32_t lo, uint32_t hi, uint32_t& even, uint32_t& odd) {
    even = __builtin_amdgcn_perm(hi, lo, 0x05040100u);     // bytes: lo[0:1], hi[0:1]
    odd = __builtin_amdgcn_perm(hi, lo, 0x07060302u);      // bytes: lo[2:3], hi[2:3]
}

template <bool OVERWRITE>
__device__ __forceinline__ void dwgs_store2(float* q, float x, float y) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 v = {x, y};
    if constexpr (!OVERWRITE) { const f32x2 o = *(const f32x2*)q; v[0] += o[0]; v[1] += o[1]; }
    *(f32x2*)q = v;
}

template <bool OVERWRITE>
__global__ __launch_bounds__(256) void dwgs_kernel(const DwgsParams p) {
    __shared__ float red[64 * 64 + 64];                        // the row splits' running sum: the tile [64][64], the bias row behind it
    const int lane = threadIdx.x & 63, lrow = lane & 31, lgrp = lane >> 5;
    const int z = (int)threadIdx.x >> 6, ws = (int)blockDim.x >> 6;      // this wave's row split / splits per tile
    // tile of this block: XCD x (block b runs on XCD b % 8) owns a contiguous range of tiles, the dimension with fewer tiles fastest (neighbours share operand columns in L2)
    const int T = p.KT * p.NT;
    const int per = (T + 7) >> 3;
    const int t = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (t >= T) return;                                        // (block-uniform)
    int kt, nt;
    if (p.NT <= p.KT) { kt = t / p.NT; nt = t - kt * p.NT; } else { nt = t / p.KT; kt = t - nt * p.KT; }
    const int k0 = kt * 64, n0 = nt * 64;
    const bool with_bias = p.dbias != nullptr && kt == 0;      // (wave-uniform)

    // lane (lrow, lgrp): columns 2 lrow, 2 lrow + 1 of the tile, rows 16 s + 8 lgrp + e (e = 0 .. 7) of step s
    const int ns = (p.M >> 4) / ws;                            // (M % (16 ws) == 0: checked by the launcher)
    const long long m0 = (long long)z * ns * 16 + 8 * lgrp;
    const bf16_t* pa = p.a + m0 * p.K + k0 + 2 * lrow;
    const bf16_t* pb = p.dy + m0 * p.N + n0 + 2 * lrow;
    const long long stepA = 16ll * p.K, stepB = 16ll * p.N;

    uint32_t ra[DWGS_DEPTH][8], rb[DWGS_DEPTH][8];
    auto issue = [&](int d, int s) {
        const bf16_t* qa = pa + (long long)s * stepA;
        const bf16_t* qb = pb + (long long)s * stepB;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            ra[d][e] = *(const uint32_t*)(qa + (long long)e * p.K);
            rb[d][e] = *(const uint32_t*)(qb + (long long)e * p.N);
        }
    };
#pragma unroll
    for (int d = 0; d < DWGS_DEPTH; ++d) if (d < ns) issue(d, d);

    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; accb[i][r] = 0.f; }
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};

    auto step = [&](int d, int s) {
        u32x4 fa[2], fb[2];                                    // fragment of the even / odd column
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            uint32_t ev, od;
            dwgs_split(ra[d][2 * h], ra[d][2 * h + 1], ev, od); fa[0][h] = ev; fa[1][h] = od;
            dwgs_split(rb[d][2 * h], rb[d][2 * h + 1], ev, od); fb[0][h] = ev; fb[1][h] = od;
        }
        if (s + DWGS_DEPTH < ns) issue(d, s + DWGS_DEPTH);      // the slot is free again: DEPTH - 1 steps stay in flight under this step's MFMAs
        // D[row][col]: register r of a lane = row (r & 3) + 8 (r >> 2) + 4 lgrp, column lrow; row R of fragment pair (fi, fj) = dW row k0 + 2 R + fi
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int fj = 0; fj < 2; ++fj)
                acc[fi][fj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[fi]), __builtin_bit_cast(bf16x8, fb[fj]), acc[fi][fj], 0, 0, 0);
        // every row of this product = the column sums of dy.  Issued by every wave (the kernel is bound by its loads, not by 6 instead of 4 MFMAs per step; a
        // wave-uniform branch around them made the compiler move the accumulators between the two register files in every step); stored by the first k tile
        accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, fb[0]), accb[0], 0, 0, 0);
        accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, fb[1]), accb[1], 0, 0, 0);
    };
    int s0 = 0;
    for (; s0 + DWGS_DEPTH <= ns; s0 += DWGS_DEPTH) {           // whole groups: no branch around an MFMA
#pragma unroll
        for (int d = 0; d < DWGS_DEPTH; ++d) step(d, s0 + d);
    }
#pragma unroll
    for (int d = 0; d < DWGS_DEPTH - 1; ++d) if (s0 + d < ns) step(d, s0 + d);      // M % 64 != 0: the last one to three steps (already in flight in slots 0 ..)

    if (ws > 1) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        for (int zz = 0; zz + 1 < ws; ++zz) {                  // waves 0 .. ws - 2, one after the other: a fixed summation order
            if (z == zz) {
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        f32x2* q = (f32x2*)&red[(fi + 8 * lgrp + 2 * ((r & 3) + 8 * (r >> 2))) * 64 + 2 * lrow];
                        f32x2 v = {acc[fi][0][r], acc[fi][1][r]};
                        if (zz > 0) { const f32x2 o = *q; v[0] = o[0] + v[0]; v[1] = o[1] + v[1]; }
                        *q = v;
                    }
                if (lgrp == 0) {
                    f32x2* q = (f32x2*)&red[64 * 64 + 2 * lrow];
                    f32x2 v = {accb[0][0], accb[1][0]};
                    if (zz > 0) { const f32x2 o = *q; v[0] = o[0] + v[0]; v[1] = o[1] + v[1]; }
                    *q = v;
                }
            }
            __syncthreads();
        }
        if (z != ws - 1) return;                               // (no barrier behind this point)
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x2 o = *(const f32x2*)&red[(fi + 8 * lgrp + 2 * ((r & 3) + 8 * (r >> 2))) * 64 + 2 * lrow];
                acc[fi][0][r] = o[0] + acc[fi][0][r]; acc[fi][1][r] = o[1] + acc[fi][1][r];
            }
        const f32x2 ob = *(const f32x2*)&red[64 * 64 + 2 * lrow];      // (every lane reads it; lgrp 0 stores)
        accb[0][0] = ob[0] + accb[0][0]; accb[1][0] = ob[1] + accb[1][0];
    }
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
        float* o = p.out + (long long)(k0 + fi + 8 * lgrp) * p.N + n0 + 2 * lrow;      // row k0 + 2 (4 lgrp + ...) + fi
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dwgs_store2<OVERWRITE>(o + (long long)(2 * ((r & 3) + 8 * (r >> 2))) * p.N, acc[fi][0][r], acc[fi][1][r]);
    }
    if (with_bias && lgrp == 0) dwgs_store2<OVERWRITE>(p.dbias + n0 + 2 * lrow, accb[0][0], accb[1][0]);
}

}  // namespace mi
