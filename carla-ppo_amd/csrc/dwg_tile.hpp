// dwg_tile.hpp — dense filter gradient with a SHORT reduction and a LARGE result (round 4: the two 19.7 M-weight layers of the MlpVAE, vae/models.py:271-299):
//   dW[k][n] = sum_m a[m][k] dy[m][n]      a [M, K] and dy [M, N] bf16 row-major, M = the minibatch (512), K x N = 38400 x 512 / 512 x 38400
// The first-generation kernel (wgrad_tile.hpp) took 100-105 us per layer: 2,400 blocks of eight register-staged steps with ONE step of look-ahead and two
// resident blocks per CU -- a chain of memory latencies for 20 GFLOP and 120 MB.  Here:
//   * one block = one 128 (k) x 128 (n) tile of dW over ALL rows; a stage = 32 rows x 256 bytes of each operand, fetched by LDS-DMA (no staging registers) into a
//     ring of NST stages, NST - 1 in flight (the pipeline of gemm2_tile.hpp);
//   * both MFMA operands are read with the hardware transpose (ds_read_b64_tr_b16: the reduction index m is the slow index of both tiles); the 256-byte rows are
//     unpadded (the DMA writes wave-linear), so the 64-byte group q of row r is stored at q ^ (r & 3): the four rows a half-wave reads then cover all 64 banks;
//   * tiles that share an operand tile are neighbours inside one XCD (block b runs on XCD b % 8: each XCD gets a contiguous range of tiles, the dimension with
//     fewer tiles fastest), so the big operand is read from HBM once;
//   * the result is STORED (every element exactly once, 128 contiguous bytes per 32 lanes), and the bias gradient -- the column sums of dy -- is one extra MFMA per
//     step with an all-ones operand in the blocks of the first k tile.
#pragma once
#include "gemm2_tile.hpp"
#include "wgrad_tile.hpp"

namespace mi {

struct DwgParams {
    const void* a; uint32_t a_bytes;
    const void* dy; uint32_t dy_bytes;
    float* out; float* dbias;
    int M, K, N;
    int KT, NT;                      // tiles of 128 along k and n
};

constexpr int DWG_ROWS = 32;                               // rows (m) per stage
constexpr int DWG_STAGE = 2 * DWG_ROWS * 256;              // bytes: the A rows, then the dy rows

// DWG_NST stages of 16 KB: 4 -> two blocks per CU, 3 -> three (132 VGPRs fit three waves per SIMD)
template <int DWG_NST>
__global__ __launch_bounds__(256, DWG_NST == 3 ? 3 : 2) void dwg_kernel(const DwgParams p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[DWG_NST * DWG_STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;               // the wave's 64 x 64 quarter of the tile
    const int lrow = lane & 31, lgrp = lane >> 5;

    // tile of this block: XCD x owns tiles [x * per, (x + 1) * per)
    const int T = p.KT * p.NT;
    const int per = (T + 7) >> 3;
    const int t = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (t >= T) return;
    int kt, nt;
    if (p.NT <= p.KT) { kt = t / p.NT; nt = t - kt * p.NT; } else { nt = t / p.KT; kt = t - nt * p.KT; }
    const int k0 = kt * 128, n0 = nt * 128;
    const bool with_bias = p.dbias != nullptr && kt == 0;  // (block-uniform)

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)p.dy_bytes, 0x00020000);

    // DMA roles: an instruction moves 4 rows x 256 bytes; wave w issues the row groups 2 w and 2 w + 1 of both operands.  Lane l fills physical 16-byte chunk
    // l & 15 of row l >> 4 and therefore fetches logical chunk (l & 15) ^ ((row & 3) << 2)
    const int drow = lane >> 4, dchunk = (lane & 15) ^ (drow << 2);
    uint32_t voffA[2], voffB[2];
    int mrow[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        mrow[j] = (wave * 2 + j) * 4 + drow;
        voffA[j] = ((uint32_t)mrow[j] * (uint32_t)p.K + (uint32_t)(k0 + dchunk * 8)) * 2u;
        voffB[j] = ((uint32_t)mrow[j] * (uint32_t)p.N + (uint32_t)(n0 + dchunk * 8)) * 2u;
    }
    const uint32_t stepA = (uint32_t)DWG_ROWS * (uint32_t)p.K * 2u, stepB = (uint32_t)DWG_ROWS * (uint32_t)p.N * 2u;
    auto issue = [&](int s, int buf) {
        unsigned char* As = &lds[buf * DWG_STAGE];
        unsigned char* Bs = As + DWG_ROWS * 256;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bool ok = s * DWG_ROWS + mrow[j] < p.M;  // rows past the minibatch: zeros from the descriptor's range check
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(As + (wave * 2 + j) * 1024), 16, (int)(ok ? voffA[j] + (uint32_t)s * stepA : G2_OOB), 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vptr)(Bs + (wave * 2 + j) * 1024), 16, (int)(ok ? voffB[j] + (uint32_t)s * stepB : G2_OOB), 0, 0, 0);
        }
    };

    // transposed fragment reads: a 16-lane group (g = lane >> 4) takes rows 8 (g >> 1) + (c >> 2) (+ 4 for the second read) and the 16 channels 16 (g & 1) .. of
    // a 32-channel tile; lane c supplies the address of 4 channels (8 bytes) of its row.  Byte offset inside a 32-row operand block:
    const int g = lane >> 4, c = lane & 15;
    const int frow = (g >> 1) * 8 + (c >> 2);
    auto frag_off = [&](int tile32) {                      // tile32: 32-channel tile inside the 128-channel row (0 .. 3)
        const int chunk = tile32 * 4 + (g & 1) * 2 + ((c & 3) >> 1);
        return frow * 256 + ((chunk ^ ((frow & 3) << 2)) << 4) + ((c & 1) << 3);
    };
    int offA[2], offB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { offA[i] = frag_off(wk * 2 + i); offB[i] = DWG_ROWS * 256 + frag_off(wn * 2 + i); }
    typedef __attribute__((address_space(3))) s16x4* lds_v4;
    auto frag = [&](const unsigned char* stage, int off, int kk) -> u16x8 {     // rows 16 kk .. 16 kk + 15 of the stage
        const unsigned char* q = stage + off + kk * 16 * 256;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(q + 4 * 256));
        u16x8 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[j] = (unsigned short)lo[j]; r[4 + j] = (unsigned short)hi[j]; }
        return r;
    };

    f32x16 acc[2][2], accb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][0][r] = 0.f; acc[i][1][r] = 0.f; accb[i][r] = 0.f; }
    }
    u16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (unsigned short)0x3F80;

    const int nk = (p.M + DWG_ROWS - 1) / DWG_ROWS;
#pragma unroll
    for (int s = 0; s < DWG_NST - 1; ++s) if (s < nk) issue(s, s);
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks, cur = (cur + 1 == DWG_NST ? 0 : cur + 1)) {
        // every thread issues exactly 4 DMA instructions per stage, in stage order: "all but the newest NST - 2 stages' worth" = stage ks has landed
        if (ks + DWG_NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((DWG_NST - 2) * 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int nb = cur == 0 ? DWG_NST - 1 : cur - 1;
        if (ks + DWG_NST - 1 < nk) issue(ks + DWG_NST - 1, nb);
        const unsigned char* stage = &lds[cur * DWG_STAGE];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const u16x8 a0 = frag(stage, offA[0], kk), a1 = frag(stage, offA[1], kk);
            const u16x8 b0 = frag(stage, offB[0], kk), b1 = frag(stage, offB[1], kk);
            // D[row = k channel][col = n]: register r of a lane = row (r & 3) + 8 (r >> 2) + 4 lgrp, column lrow
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), acc[1][1], 0, 0, 0);
            if (with_bias && wk == 0) {                    // (wave-uniform) every row of this product = the column sums of dy
                accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, b0), accb[0], 0, 0, 0);
                accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, b1), accb[1], 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* o = p.out + (long long)(k0 + wk * 64 + i * 32 + 4 * lgrp) * p.N + n0 + wn * 64 + j * 32 + lrow;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(long long)((r & 3) + 8 * (r >> 2)) * p.N] = acc[i][j][r];
        }
    if (with_bias && wk == 0 && lgrp == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) p.dbias[n0 + wn * 64 + j * 32 + lrow] = accb[j][0];
    }
}

}  // namespace mi
