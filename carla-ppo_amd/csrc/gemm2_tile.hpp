// gemm2_tile.hpp — second-generation MFMA tile kernel for the wide (C % 16 B == 0) conv / deconv layers.
//
// Same contractions and epilogue as gemm_kernel (gemm_tile.hpp); what changed is how tiles reach the LDS:
//   * every 16-byte chunk of the A (im2col / deconv-gather) and B (weight) tiles is fetched by an LDS-DMA buffer load
//     (`buffer_load_dwordx4 ... lds`): no staging VGPRs, no ds_write pass, no per-element zero-fill selects.  The hardware
//     range check of the buffer descriptor supplies the zeros: a chunk that falls outside the image (deconv borders),
//     past M, N or K gets voffset = G2_OOB and lands in LDS as zeros.
//   * a stage is 128 bytes of K per row (64 bf16 / 32 f32): half the barriers of gemm_kernel, 4x the MFMAs per barrier.
//   * LDS rows are unpadded (the DMA writes wave-linear: lane l -> base + 16 l); bank conflicts of the ds_read_b128
//     fragment reads are removed by an XOR swizzle applied on the SOURCE side: the lane that fills physical chunk p of tile
//     row r fetches logical chunk p ^ ((r >> 1) & 7).
//   * per-row address state is ONE 32-bit byte offset (+ a 12-bit tap-validity mask for the gather form); the per-stage
//     k -> (tap, channel) arithmetic is done once per thread (wave-uniform when a stage stays inside one tap).
#pragma once
#include "gemm_tile.hpp"

namespace mi {

typedef __attribute__((address_space(3))) void* lds_vptr;
constexpr uint32_t G2_OOB = 0x40000000u;      // any voffset >= this is outside every descriptor we build (tensors < 1 GiB)

struct Gemm2Params {
    const void* a; uint32_t a_bytes;
    const void* b; uint32_t b_bytes;
    int IH, IW, C;                   // A tensor dims
    int OH, OW;                      // A_CONV: output grid; A_DECONV: output tensor dims
    int KH, KW, stride;
    int M, N, K, nbatch;
    int run;                         // A_CONV: contiguous k-run = KW*C
    FastDiv div_ohw, div_ow, div_run;
    int OHc[2], OWc[2], Th[2], Tw[2];
    FastDiv dc_ohw[4], dc_ow[4], dc_c, dc_tw[2];
    int ldb;
    void* out; const float* bias; const void* mask; int relu; int out_f32;
    int remap3;                      // A_CONV, dense layers (round 5): the WHOLE (x, y, z) grid is renumbered so that every XCD owns a contiguous range of tiles with x (the row tiles
                                     // that share one weight tile) fastest -- the 2.9-3.9 x L2 -> LDS re-reads of profiles/r04_c section 5 (consecutive block ids = consecutive XCDs)
    int ksplit_len;                  // A_CONV, > 0: split-K -- blockIdx.z owns k in [z * len, (z + 1) * len) (len a multiple of the 128-byte stage) and writes raw fp32 slab z
};

// BM x BN output tile (pixels x channels), 256 threads = 4 waves.  UTAP: every 128-byte stage lies inside one deconv tap
// (C*sizeof(T) % 128 == 0), so the tap arithmetic is wave-uniform (scalar unit).
// NST: LDS stages of the K pipeline.  Round 3: with two stages the loads of step k + 1 are issued after the barrier of step k and must land before the barrier of
// step k + 1 -- one step of MFMAs (0.1-0.2 us) to cover an L2 / HBM round trip (0.6-0.9 us): the small-grid layers (conv4 forward, deconv1's input gradient: 32
// k-steps, 1.5 blocks per CU) ran as a chain of memory latencies, 30 us for 5 us of MFMAs.  With NST stages NST - 1 are in flight: the barrier of step k waits with
// s_waitcnt vmcnt((NST - 2) x loads per stage) -- "all but the newest NST - 2 stages" (LDS-DMA returns in order) -- instead of __syncthreads()'s vmcnt(0).
template <typename T, int AMODE, int BMODE, int BM, int BN, bool UTAP, int NST = 2>
__global__ __launch_bounds__(GEMM_NT) void gemm2_kernel(const Gemm2Params p) {
    constexpr int RB = 128;
    constexpr int ESZ = (int)sizeof(T);
    constexpr int BKE = RB / ESZ;
    constexpr int VE = 16 / ESZ;
    constexpr int WN = (BN >= 64) ? 2 : 1;
    constexpr int WM = 4 / WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    constexpr int NJA = BM / 32, NJB = BN / 32;          // DMA instructions per wave per stage
    constexpr int STAGE = (BM + BN) * RB;
    static_assert(BM >= 64 && TM >= 1 && TN >= 1, "tile config");
    static_assert(BMODE == B_NK || BMODE == B_DECONV, "gemm2 wants K-contiguous weights");
    typedef typename Frag<T>::reg freg;

    static_assert(NST >= 2 && NST <= 4 && (NST - 2) * (NJA + NJB) < 64, "stage ring");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lgrp = lane >> 5;

    int bx = (int)blockIdx.x, by = (int)blockIdx.y, bz = (int)blockIdx.z;
    if (AMODE == A_CONV && p.remap3) {
        const int gx = (int)gridDim.x, gy = (int)gridDim.y;
        const int lin = xcd_remap(bx + gx * (by + gy * bz), gx * gy * (int)gridDim.z);
        bx = lin % gx; by = (lin / gx) % gy; bz = lin / (gx * gy);
    } else {
        bx = xcd_remap(bx, (int)gridDim.x);
    }
    const int m0 = bx * BM;
    const int n0 = by * BN;

    int M = p.M, K = p.K;
    int cls = 0, ph = 0, pw = 0;
    if constexpr (AMODE == A_DECONV) {
        cls = bz; ph = cls >> 1; pw = cls & 1;
        M = p.nbatch * p.OHc[ph] * p.OWc[pw];
        K = p.Th[ph] * p.Tw[pw] * p.C;
        if (m0 >= M) return;
    }
    int kbeg = 0;
    if constexpr (AMODE == A_CONV) {
        if (p.ksplit_len > 0) { kbeg = bz * p.ksplit_len; K = min(K, kbeg + p.ksplit_len); }      // (K = one past this split's last k)
    }
    const int nk = (K - kbeg + BKE - 1) / BKE;

    // ---------------- DMA lane roles ----------------
    // wave w fills 8-row groups of one parity (w & 1), so the swizzle term (row >> 1) & 7 is the same for all of a thread's rows
    const int par = wave & 1, half = wave >> 1, r8 = lane >> 3;
    const int cch = (lane & 7) ^ ((4 * par + (lane >> 4)) & 7);        // logical 16-B chunk this thread fetches in every stage
    auto rowA = [&](int j) { return (BM / 2) * half + 16 * j + 8 * par + r8; };
    auto rowB = [&](int j) { return BN == 32 ? 8 * wave + r8 : (BN / 2) * half + 16 * j + 8 * par + r8; };

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, (int)p.b_bytes, 0x00020000);

    uint32_t offA[NJA], offB[NJB];
    uint32_t tapmask[AMODE == A_DECONV ? NJA : 1];
#pragma unroll
    for (int j = 0; j < NJA; ++j) {
        const int m = m0 + rowA(j);
        const bool ok = m < M;
        const uint32_t mm = ok ? (uint32_t)m : 0u;
        uint32_t b, rem, y, x;
        if constexpr (AMODE == A_CONV) {
            p.div_ohw.divmod(mm, b, rem);
            p.div_ow.divmod(rem, y, x);
            offA[j] = ok ? (((b * p.IH + y * p.stride) * p.IW + x * p.stride) * p.C) * ESZ : G2_OOB;
        } else {
            p.dc_ohw[cls].divmod(mm, b, rem);
            p.dc_ow[cls].divmod(rem, y, x);
            offA[j] = (((b * p.IH + y) * p.IW + x) * p.C) * ESZ;
            uint32_t mk = 0;
#pragma unroll
            for (int th = 0; th < 3; ++th)
#pragma unroll
                for (int tw = 0; tw < 3; ++tw) {
                    const int ih = (int)y - th, iw = (int)x - tw;
                    if (ok && th < p.Th[ph] && tw < p.Tw[pw] && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW) mk |= 1u << (th * 4 + tw);
                }
            tapmask[j] = mk;
        }
    }
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
        const int n = n0 + rowB(j);
        if constexpr (BMODE == B_NK) offB[j] = n < p.N ? (uint32_t)n * (uint32_t)p.ldb * ESZ : G2_OOB;
        else offB[j] = n < p.N ? (uint32_t)n * (uint32_t)p.C * ESZ : G2_OOB;
    }

    auto issue = [&](int ks, int buf) {
        const int k0 = kbeg + ks * BKE;                    // wave-uniform
        const int k = k0 + cch * VE;
        const bool kok = k < K;
        uint32_t koffA, koffB, bit = 0;
        if constexpr (AMODE == A_CONV) {
            uint32_t seg, j;
            p.div_run.divmod((uint32_t)(kok ? k : 0), seg, j);
            koffA = kok ? (seg * p.IW * p.C + j) * ESZ : G2_OOB;
        } else {
            uint32_t tap, cc, th, tw;
            if constexpr (UTAP) { tap = p.dc_c.div((uint32_t)k0); cc = (uint32_t)k - tap * p.C; }
            else p.dc_c.divmod((uint32_t)(kok ? k : 0), tap, cc);
            p.dc_tw[pw].divmod(tap, th, tw);
            koffA = (cc - (th * p.IW + tw) * p.C) * ESZ;   // may be "negative": the sum with a valid row offset is not
            bit = kok ? 1u << (th * 4 + tw) : 0u;
            if constexpr (BMODE == B_DECONV) {
                const uint32_t kh = ph + 2 * th, kw = pw + 2 * tw;
                koffB = kok ? (((kh * p.KW + kw) * p.N) * p.C + cc) * ESZ : G2_OOB;
            }
        }
        if constexpr (BMODE == B_NK) koffB = kok ? (uint32_t)k * ESZ : G2_OOB;
        unsigned char* As = &lds[buf * STAGE];
        unsigned char* Bs = As + BM * RB;
#pragma unroll
        for (int j = 0; j < NJA; ++j) {
            uint32_t vo;
            if constexpr (AMODE == A_CONV) vo = offA[j] + koffA;
            else vo = (tapmask[j] & bit) ? offA[j] + koffA : G2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(As + (rowA(j) - r8) * RB), 16, (int)vo, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_vptr)(Bs + (rowB(j) - r8) * RB), 16, (int)(offB[j] + koffB), 0, 0, 0);
    };

    // ---------------- main loop ----------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int xs = (lrow >> 1) & 7;                        // fragment-read swizzle term (same for every 32-row subtile)
    if constexpr (NST == 2) { if (nk > 0) issue(0, 0); }
    else {
#pragma unroll
        for (int s_ = 0; s_ < NST - 1; ++s_) if (s_ < nk) issue(s_, s_);
    }
    int cur = 0;
    for (int ks = 0; ks < nk; ++ks, cur = (cur + 1 == NST ? 0 : cur + 1)) {
        if constexpr (NST == 2) {
            __syncthreads();                               // stage ks has landed (vmcnt(0) + barrier); buffer cur^1 is free
            if (ks + 1 < nk) issue(ks + 1, cur ^ 1);
        } else {
            // every thread issues exactly NJA + NJB DMA instructions per stage, in stage order: "all but the newest (NST - 2) stages' worth" = stage ks is in LDS.
            // (the last NST - 2 steps have fewer stages behind them: wait for everything there)
            if (ks + NST - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * (NJA + NJB)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                  // everybody's share of stage ks has landed, and everybody is done reading stage ks - 1 ...
            const int nb = cur == 0 ? NST - 1 : cur - 1;   // ... whose buffer takes stage ks + NST - 1
            if (ks + NST - 1 < nk) issue(ks + NST - 1, nb);
        }
        const unsigned char* As = &lds[cur * STAGE];
        const unsigned char* Bs = As + BM * RB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int pc = ((kk * 2 + lgrp) ^ xs) * 16;
            freg af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const freg*)(&As[((wm * TM + i) * 32 + lrow) * RB + pc]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *(const freg*)(&Bs[((wn * TN + j) * 32 + lrow) * RB + pc]);
            // operands swapped on purpose (see gemm_kernel): D[row = channel][col = pixel]
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Frag<T>::mma(bf[j], af[i], acc[i][j]);
        }
    }

    store_tile<T, AMODE, TM, TN>(p, acc, m0, n0, wm, wn, lrow, lgrp, M, cls, ph, pw, (AMODE == A_CONV && p.ksplit_len > 0) ? bz : 0);
}

}  // namespace mi

namespace mi {

// =====================================================================================================================
// dense_smallm_kernel — out[M,N] = act(x[M,K] W[K,N] + b) in exact fp32 for SMALL M (the PPO minibatch: 32 rows).  The tile
// kernels give such a layer 4 blocks that walk K serially (46-80 us, launch-latency class); here one block owns a 32 x 32 output
// tile, its four waves each take a quarter of K (operands straight from global / L2: x rows as 16-byte vectors, W rows coalesced
// across the 32 columns), and the four partial tiles meet in LDS.  ~3 us per layer.
// =====================================================================================================================
struct DenseSmallParams {
    const float* x; const float* w; const float* bias; float* out;
    int M, N, K, relu;
};

static __global__ __launch_bounds__(256) void dense_smallm_kernel(const DenseSmallParams p) {
    __shared__ float red[3][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lgrp = lane >> 5;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int ksteps = (p.K + 7) / 8;                     // 8 k per step (two 4-float chunks: lane group g takes chunk g)
    const int per = (ksteps + 3) / 4;
    const int sb = wave * per, se = min(ksteps, sb + per);
    const int m = m0 + lrow, n = n0 + lrow;
    const bool mok = m < p.M, nok = n < p.N;
    const float* xrow = p.x + (long long)(mok ? m : 0) * p.K;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = sb; s < se; ++s) {
        const int k = s * 8 + lgrp * 4;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b;
        if (mok && k + 4 <= p.K) a = *(const f32x4*)(xrow + k);
        else if (mok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = k + e < p.K ? xrow[k + e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = (nok && k + e < p.K) ? p.w[(long long)(k + e) * p.N + n] : 0.f;
        Frag<float>::mma(a, b, acc);                      // D[row = m][col = n]
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = ((acc[r] + red[0][r][lane]) + (red[1][r][lane] + red[2][r][lane]));
            const int mm = m0 + (r & 3) + 8 * (r >> 2) + 4 * lgrp;
            if (mm < p.M && nok) {
                if (p.bias) v += p.bias[n];
                if (p.relu) v = fmaxf(v, 0.f);
                p.out[(long long)mm * p.N + n] = v;
            }
        }
    }
}

}  // namespace mi
