// narrow_tile.hpp — the two "skinny" ends of the decoder/encoder: layers with 1..8 channels on one side (rgb / segmentation
// frames and logits).  They move ~100 MB through HBM for a few GFLOP, so the kernels are organised around full-line memory
// traffic; the matrix core only does the (tiny) arithmetic.
//
//   gather_narrow_kernel   transposed conv k x k, s2 into a NARROW output (deconv4 fwd: 32 -> 3 channels).  Same slot formulation
//                          as tapconv_tile.hpp (gather form): the 4 output parities x N channels (4N <= 32) are the MFMA
//                          rows, 32 positions the columns.  The slot range of a 128-position tile is staged once by LDS-DMA,
//                          the weights (a few KB) live in registers, and the result leaves through a wave-private LDS
//                          transpose as contiguous 2N-element runs: one lane per (position, output row parity), adjacent lanes
//                          write adjacent bytes.
#pragma once
#include "tapconv_tile.hpp"

namespace mi {

constexpr int GN_BMT = 128;          // positions per block (4 waves x 32)
constexpr int GN_NT = 256;

// chunk swizzle for PA-byte LDS rows read with ds_read_b128 by 32 consecutive rows (see gemm2_tile.hpp / tapconv_tile.hpp)
template <int CPR> __device__ __forceinline__ int gn_swz(int row) { return CPR == 8 ? (row >> 1) & 7 : (row >> 2) & 3; }

// C (input channels) * sizeof(T) must be 64 or 128 bytes: CPR = 4 | 8 sixteen-byte chunks per pixel
// NOUT: output channels when known at compile time (3 rgb, 1 segmentation: packed dword stores), 0 = generic element stores
template <typename T, int TAPS, int CPR, int NOUT>
__global__ __launch_bounds__(GN_NT) void gather_narrow_kernel(const TapParams p) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VE = 16 / ESZ;
    constexpr int PA = CPR * 16;
    constexpr int NT = TAPS * TAPS;
    constexpr int NKK = CPR / 2;                          // MFMA chunk pairs per tap
    constexpr int SPI = 1024 / PA;                        // slots per DMA instruction
    constexpr int MAXSLOT = GN_BMT + TC_MAXHALO;          // 224
    constexpr int NIA = (MAXSLOT / SPI + 3) / 4;          // DMA instructions per wave (upper bound)
    constexpr int OPITCH = 36;                            // floats per position in the output transpose (32 + pad)
    typedef typename Frag<T>::reg freg;

    __shared__ __attribute__((aligned(1024))) unsigned char lds[MAXSLOT * PA + 4 * 32 * OPITCH * 4];
    float* const otile = (float*)(lds + MAXSLOT * PA);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lgrp = lane >> 5;
    const int P0 = xcd_remap(blockIdx.x, gridDim.x) * GN_BMT;
    const int halo = (TAPS - 1) * p.GW + TAPS - 1;
    const int ninstr = (GN_BMT + halo + SPI - 1) / SPI;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);

    // ---- stage the slot range: instruction t = wave + 4 i fills slots SPI t .. ; lane -> slot SPI t + lane / CPR, chunk lane % CPR ----
    {
        const int rs = lane / CPR;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int t = wave + 4 * i;
            if (t >= ninstr) break;
            const int q = SPI * t + rs;
            const int P = P0 + q;
            const int c = (lane % CPR) ^ gn_swz<CPR>(q);
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(P < p.MP ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            const int iy = (int)gy - p.HY, ix = (int)gx - p.HX;
            const bool in = P < p.MP && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW && c * VE < p.C;
            const uint32_t vo = in ? (((b * p.IH + iy) * p.IW + ix) * p.C + c * VE) * ESZ : G2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(lds + t * 1024), 16, (int)vo, 0, 0, 0);
        }
    }

    // ---- weights: fragment (tap, kk) of lane (row ne = lrow, chunk 2 kk + lgrp) straight from global / L2 ----
    freg wf[NT][NKK];
    {
        const int ne = lrow;
        const bool rowok = ne < p.NE;
        const uint32_t cls = rowok ? p.div_n.div((uint32_t)ne) : 0u;
        const int n = ne - (int)cls * p.N;
        const T* __restrict__ W = (const T*)p.b;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int ta = tap / TAPS, tb = tap % TAPS;
            const int kh = (int)(cls >> 1) + 2 * (p.HY - ta), kw = (int)(cls & 1) + 2 * (p.HX - tb);
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int c0 = (2 * kk + lgrp) * VE;
                const bool ok = rowok && kh < p.KH && kw < p.KW && c0 < p.C;
                const T* src = ok ? W + ((long long)(kh * p.KW + kw) * p.N + n) * p.C + c0 : W;
                freg v = *(const freg*)src;
#pragma unroll
                for (int e = 0; e < (int)(sizeof(freg) / sizeof(v[0])); ++e) v[e] = ok ? v[e] : (decltype(v[0] + 0))0;
                wf[tap][kk] = v;
            }
        }
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    __syncthreads();                                      // the slot range has landed (vmcnt(0) + barrier)

    const int q0 = wave * 32 + lrow;
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
        const int q = q0 + (tap / TAPS) * p.GW + (tap % TAPS);
        const int sw = gn_swz<CPR>(q);
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const freg af = *(const freg*)(lds + q * PA + (((2 * kk + lgrp) ^ sw) << 4));
            Frag<T>::mma(wf[tap][kk], af, acc);           // D[row = parity*N + channel][col = position]
        }
    }

    // ---- epilogue: D -> wave-private [position][32] floats -> one lane per (position, output-row parity) ----
    float* ot = otile + wave * 32 * OPITCH;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        f32x4 v = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
        *(f32x4*)(ot + lrow * OPITCH + 8 * qd + 4 * lgrp) = v;
    }
    __builtin_amdgcn_wave_barrier();                      // same wave, in-order LDS queue: no wait needed, only keep the order
    const int P = P0 + wave * 32 + lrow;
    if (P >= p.MP) return;
    uint32_t g, gx, b, gy;
    p.div_gw.divmod((uint32_t)P, g, gx);
    p.div_g.divmod(g, b, gy);
    const int ph = lgrp;
    const int oy = 2 * (int)gy + ph, ox = 2 * (int)gx;
    if (oy >= p.OH || ox >= p.OW) return;
    const int npx = ox + 1 < p.OW ? 2 : 1;                // output pixels of this lane (pw = 0, 1)
    T* __restrict__ out = (T*)p.out + (((long long)b * p.OH + oy) * p.OW + ox) * p.N;
    const float* src = ot + lrow * OPITCH + ph * 2 * p.N;
    if constexpr (NOUT > 0) {
        if (npx == 2) {                                   // 2 N contiguous elements = ND dwords, adjacent lanes adjacent bytes
            constexpr int CNT = 2 * NOUT;
            constexpr int ND = CNT * ESZ / 4;
            float v[CNT];
#pragma unroll
            for (int j = 0; j < CNT; ++j) {
                v[j] = src[j];
                if (p.bias) v[j] += p.bias[j % NOUT];
                if (p.relu) v[j] = fmaxf(v[j], 0.f);
            }
            uint32_t w[ND];
            if constexpr (ESZ == 2) {
#pragma unroll
                for (int d = 0; d < ND; ++d) w[d] = (uint32_t)f32_to_bf16(v[2 * d]) | ((uint32_t)f32_to_bf16(v[2 * d + 1]) << 16);
            } else {
#pragma unroll
                for (int d = 0; d < ND; ++d) w[d] = __builtin_bit_cast(uint32_t, v[d]);
            }
            uint32_t* o32 = (uint32_t*)out;
#pragma unroll
            for (int d = 0; d < ND; ++d) o32[d] = w[d];
            return;
        }
    }
    const int cnt = npx * p.N;
    for (int j = 0; j < cnt; ++j) {
        const int n = j >= p.N ? j - p.N : j;
        float v = src[j];
        if (p.bias) v += p.bias[n];
        if (p.relu) v = fmaxf(v, 0.f);
        out[j] = Elem<T>::from_f32(v);
    }
}

}  // namespace mi
