// wgrad_tile.hpp — weight-gradient MFMA tile kernel (see gemm_core.hpp for the family overview)
#pragma once
#include "gemm_tile.hpp"

namespace mi {

// =====================================================================================================
// wgrad: out[kc, n] += sum_{m in this block's pixel range} A(m, kc) * S(m, n)
//   the pixel splits (blockIdx.z) meet either in per-split slabs that reduce_small_fused_kernel (mi_reduce_slabs) sums in a FIXED order
//   (caller scratch: two runs are bitwise equal -- round 4) or, without scratch, in fp32 atomics on out
//   A(m,kc) = im2col view (A_CONV map) of the BIG tensor, S = the SMALL tensor [M, N] (pixel-aligned rows)
// Tile 64(kc) x 64(n), 2x2 waves of one 32x32 accumulator, BP pixels per step staged in LDS pixel-major;
// MFMA operands are read "transposed" (k = pixel) with scalar LDS reads.
// =====================================================================================================
struct WgradParams {
    const void* big;
    const int* frame_idx;
    long long frame_stride;
    int IH, IW, C, OH, OW, KH, KW, stride;
    int M, Kc, N;
    FastDiv div_ohw, div_ow, div_run, div_kw;
    int run, merged;
    const void* small;
    int s_vec;                       // vector loads of S legal (N % VB == 0, aligned)
    float* out;
    int m_per_split;                 // multiple of BP
    int debug_skip_out;              // debug (mi_set_tuning key 2): drop the atomic accumulation to time the main loop alone
    float* slabs; long long slab_stride;   // optional: split z stores its partial sums at slabs[z * slab_stride + kc * N + n] (plain stores, every in-range element exactly once)
    int overwrite;                   // one split, no slabs: out = result with plain stores instead of out += by atomics (the caller's buffer need not be zeroed)
    int ones_row; float* dbias;      // ones_row != 0: the big tensor carries a virtual column kc == Kc of ones, so row Kc of the result is the column sum of S -- the layer's
                                     // BiasAddGrad rides on the filter gradient (dense layers: one launch less per layer; slab row Kc, or atomics on dbias without slabs)
};

template <typename T> struct WgradCfg;
template <> struct WgradCfg<float>  { static constexpr int BP = 16; };
template <> struct WgradCfg<bf16_t> { static constexpr int BP = 64; };
template <> struct WgradCfg<split_t> { static constexpr int BP = 32; };   // split storage: 8 pixels per MFMA pair, 4 pairs per step and accumulator

typedef short s16x4 __attribute__((ext_vector_type(4)));

// bf16 MFMA operand (8 consecutive pixels of one channel) from a pixel-major LDS tile through the hardware transpose
// read: one ds_read_b64_tr_b16 hands each 16-lane group a [4 pixels][16 channels] block column-wise
// (lane c supplies &tile[p + (c>>2)][ch + 4*(c&3)] and receives tile[p..p+3][ch + c]); two of them fill a fragment.
__device__ __forceinline__ u16x8 tr_fragment(const bf16_t* tile, int pitch, int pix0, int ch0, int lane) {
    const int g = lane >> 4, c = lane & 15;
    const bf16_t* p0 = tile + (pix0 + (g >> 1) * 8 + (c >> 2)) * pitch + ch0 + (g & 1) * 16 + (c & 3) * 4;
    typedef __attribute__((address_space(3))) s16x4* lds_v4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p0 + 4 * pitch));
    u16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = (unsigned short)lo[j]; r[4 + j] = (unsigned short)hi[j]; }
    return r;
}

// BKC: kc rows per block (64 | 128); 2x2 waves, each wave BKC/64 accumulators of 32(kc) x 32(n)
// the block (bx, by, bz) of one launch: kc block, output block, pixel split (wgrad_kernel: the grid's own indices; wgrad_pair_kernel: two launches in one grid)
template <typename T, typename TIn, int VA, int AALIGN, int BKC>
__device__ __forceinline__ void wgrad_block(const WgradParams& p, const int bx, const int by, const int bz) {
    constexpr int NT = GEMM_NT;
    constexpr int BP = WgradCfg<T>::BP;
    constexpr int BN = 64;
    constexpr int TMW = BKC / 64;
    constexpr int VB = 16 / (int)sizeof(T);
    // row pitch (elements).  bf16: pitch in dwords == 16 or 48 (mod 64) makes the 4 pixel rows x 64 B that one half-wave
    // transpose-read touches tile all 64 banks -> conflict free.  fp32 (scalar reads): +4 elements keeps 16-B alignment.
    constexpr int LDA = sizeof(T) == 2 ? (BKC == 64 ? 96 : 160) : BKC + 4;
    constexpr int LDB = sizeof(T) == 2 ? 96 : BN + 4;
    // thread -> (pixel, sub): TPP threads share one pixel row, so the (b,y,x) decomposition is done once per thread per step
    constexpr int TPP = NT / BP;
    constexpr int VPR_A = BKC / VA;
    constexpr int NVA = VPR_A / TPP;
    constexpr int VPR_B = BN / VB;
    constexpr int NVB = VPR_B / TPP;
    static_assert(NVA >= 1 && NVB >= 1 && VPR_A % TPP == 0 && VPR_B % TPP == 0, "wgrad thread mapping");
    constexpr int KSTEP = (sizeof(T) == 2) ? 16 : (is_split<T>::value ? 8 : 2);     // pixels per MFMA (split: per MFMA pair)

    __shared__ __attribute__((aligned(16))) T lds[2][BP * LDA + BP * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lrow = lane & 31, lgrp = lane >> 5;
    const int kc0 = bx * BKC, n0 = by * BN;
    const int mbeg = bz * p.m_per_split;
    const int mend = min(p.M, mbeg + p.m_per_split);
    const int nsteps = (mend - mbeg + BP - 1) / BP;
    const int tpix = tid / TPP, tsub = tid % TPP;

    const TIn* __restrict__ Ag = (const TIn*)p.big;
    const T* __restrict__ Sg = (const T*)p.small;

    // this thread's fixed kc-vector offsets
    long long a_koff[NVA];
    bool a_kok[NVA], a_one[NVA];
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int kc = kc0 + (tsub + i * TPP) * VA;
        a_kok[i] = kc < p.Kc;
        a_one[i] = p.ones_row && kc == p.Kc;              // (Kc is a multiple of VA: the ones column is element 0 of its vector)
        uint32_t seg, j;
        p.div_run.divmod((uint32_t)(a_kok[i] ? kc : 0), seg, j);
        if (p.merged) a_koff[i] = (long long)seg * p.IW * p.C + j;
        else { uint32_t kh, kw; p.div_kw.divmod(seg, kh, kw); a_koff[i] = ((long long)kh * p.IW + kw) * p.C + j; }
    }

    PackN<TIn, VA> a_reg[NVA];
    PackN<T, VB> b_reg[NVB];
    auto load = [&](int step) {
        const int m = mbeg + step * BP + tpix;
        const bool mok = m < mend;
        uint32_t b, rem, y, x;
        p.div_ohw.divmod((uint32_t)(mok ? m : 0), b, rem);
        p.div_ow.divmod(rem, y, x);
        const long long fr = p.frame_idx ? (long long)p.frame_idx[b] : (long long)b;
        const TIn* arow = Ag + fr * p.frame_stride + ((long long)(y * p.stride) * p.IW + x * p.stride) * p.C;
        // branch-free guarded loads (see gemm_kernel::load_a)
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const bool ok = mok && a_kok[i];
            const TIn* src = ok ? arow + a_koff[i] : Ag;
            const PackU<TIn, VA, AALIGN> t = *(const PackU<TIn, VA, AALIGN>*)src;
#pragma unroll
            for (int e = 0; e < VA; ++e) a_reg[i].v[e] = ok ? t.v[e] : zero_of<TIn>();
            if (a_one[i] && mok) a_reg[i].v[0] = one_of<TIn>();
        }
        const T* srow = Sg + (long long)(mok ? m : 0) * p.N;
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int n = n0 + (tsub + i * TPP) * VB;
#pragma unroll
            for (int e = 0; e < VB; ++e) b_reg[i].v[e] = zero_of<T>();
            if (p.s_vec) {                                // wave-uniform: N % VB == 0, whole vectors in range
                const bool ok = mok && n < p.N;
                const PackN<T, VB> t = *(const PackN<T, VB>*)(ok ? srow + n : Sg);
#pragma unroll
                for (int e = 0; e < VB; ++e) b_reg[i].v[e] = ok ? t.v[e] : zero_of<T>();
            } else if (mok) {
#pragma unroll
                for (int e = 0; e < VB; ++e) if (n + e < p.N) b_reg[i].v[e] = srow[n + e];
            }
        }
    };
    auto store = [&](int buf) {
        T* As = &lds[buf][tpix * LDA];
        T* Bs = &lds[buf][BP * LDA + tpix * LDB];
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            PackN<T, VA> t;
#pragma unroll
            for (int e = 0; e < VA; ++e) {
                if constexpr (std::is_same<TIn, T>::value) t.v[e] = a_reg[i].v[e];
                else t.v[e] = Elem<T>::from_f32((float)a_reg[i].v[e]);
            }
            *(PackN<T, VA>*)(&As[(tsub + i * TPP) * VA]) = t;
        }
#pragma unroll
        for (int i = 0; i < NVB; ++i) *(PackN<T, VB>*)(&Bs[(tsub + i * TPP) * VB]) = b_reg[i];
    };

    f32x16 acc[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    if (nsteps > 0) { load(0); store(0); }
    __syncthreads();
    for (int st = 0; st < nsteps; ++st) {
        const int cur = st & 1;
        const bool more = st + 1 < nsteps;
        if (more) load(st + 1);
        const T* As = &lds[cur][0];
        const T* Bs = &lds[cur][BP * LDA];
#pragma unroll
        for (int kk = 0; kk < BP / KSTEP; ++kk) {
            if constexpr (sizeof(T) == 2) {
                const u16x8 b = tr_fragment((const bf16_t*)Bs, LDB, kk * 16, wn * 32, lane);
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    const u16x8 a = tr_fragment((const bf16_t*)As, LDA, kk * 16, (wm * TMW + i) * 32, lane);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
                }
            } else if constexpr (is_split<T>::value) {
                // lane (row, g) gathers the (lo, hi) words of pixels 8 kk + 4 g .. + 3 of its channel: the bf16x8 fragment [l0 h0 .. l3 h3];
                // a . b + a . swap16(b) = all four partial products (gemm_tile.hpp, Frag<split_t>); the swap is shared by the TMW tiles
                u32x4 b;
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Bs[(kk * 8 + lgrp * 4 + j) * LDB + wn * 32 + lrow].u;
                const u32x4 bs = Frag<split_t>::swap16(b);
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    u32x4 a;
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = As[(kk * 8 + lgrp * 4 + j) * LDA + (wm * TMW + i) * 32 + lrow].u;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bs), acc[i], 0, 0, 0);
                }
            } else {
                const float b = Bs[(kk * 2 + lgrp) * LDB + wn * 32 + lrow];
#pragma unroll
                for (int i = 0; i < TMW; ++i) {
                    const float a = As[(kk * 2 + lgrp) * LDA + (wm * TMW + i) * 32 + lrow];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                }
            }
        }
        if (more) store(cur ^ 1);
        __syncthreads();
    }

    if (nsteps <= 0) return;
    if (p.debug_skip_out && acc[0][0] != 123.456f) return;
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kc = kc0 + (wm * TMW + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lgrp;
            const int n = n0 + wn * 32 + lrow;
            if (kc < p.Kc && n < p.N) {
                if (p.slabs) p.slabs[(long long)bz * p.slab_stride + (long long)kc * p.N + n] = acc[i][r];
                else if (p.overwrite) p.out[(long long)kc * p.N + n] = acc[i][r];
                else atomicAdd(&p.out[(long long)kc * p.N + n], acc[i][r]);
            } else if (p.ones_row && kc == p.Kc && n < p.N) {
                if (p.slabs) p.slabs[(long long)bz * p.slab_stride + (long long)kc * p.N + n] = acc[i][r];
                else if (p.overwrite) p.dbias[n] = acc[i][r];
                else atomicAdd(&p.dbias[n], acc[i][r]);
            }
        }
    }
}

template <typename T, typename TIn, int VA, int AALIGN, int BKC>
__global__ __launch_bounds__(GEMM_NT) void wgrad_kernel(const WgradParams p) {
    wgrad_block<T, TIn, VA, AALIGN, BKC>(p, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// Two independent filter gradients of the same kernel configuration as ONE launch (round 6: dense1's and the heads' at the end of the ConvVAE's backward pass -- two ~190-block
// grids of a latency-bound kernel, back to back on the caller's stream, became one grid of both): blocks [0, n0) are launch 0's grid in x-fastest order, the rest launch 1's.
// Every block computes exactly what it would have in its own launch: bit-identical results.
struct WgradPair { WgradParams p[2]; int n0; int gx[2], gy[2]; };
template <typename T, typename TIn, int VA, int AALIGN, int BKC>
__global__ __launch_bounds__(GEMM_NT) void wgrad_pair_kernel(const WgradPair q) {
    const int sel = (int)blockIdx.x >= q.n0 ? 1 : 0;       // block-uniform
    int b = (int)blockIdx.x - (sel ? q.n0 : 0);
    const int gx = q.gx[sel], gy = q.gy[sel];
    const int bx = b % gx; b /= gx;
    wgrad_block<T, TIn, VA, AALIGN, BKC>(q.p[sel], bx, b % gy, b / gy);
}

}  // namespace mi
