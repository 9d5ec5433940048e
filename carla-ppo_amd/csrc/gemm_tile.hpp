// gemm_tile.hpp — LDS-staged MFMA tile kernels for gfx950 (wave64, v_mfma_f32_32x32x{16_bf16,2_f32}).
//
// Three kernel families cover every contraction on the ConvVAE / PPO hot path:
//   gemm_kernel<A_CONV,...>    C[m,n] = sum_k im2col(X)[m,k] * W[k,n]      conv fwd, deconv dgrad, dense fwd/dgrad
//   gemm_kernel<A_DECONV,...>  stride-2 transposed conv in gather form, one GEMM per output-parity class
//                              (deconv fwd, conv dgrad)
//   wgrad_kernel               dW[kc,n] += sum_m im2col(big)[m,kc] * small[m,n]   (conv/deconv/dense wgrad)
//
// Tile: 128 x BN x 64-byte-K per step, 256 threads = 4 waves, register-prefetched double-buffered LDS,
// one barrier per K step.  Both operands live in LDS K-contiguous with an 80-byte row pitch so the
// ds_read_b128 fragment reads are bank-conflict free (row*20 dwords mod 64 hits 16 distinct 16-B slots).
// Fragment convention (k-permutation is shared by A and B, so only row/col maps matter):
//   lane l: row/col = l & 31, k-group g = l >> 5; a 16-byte read at k-byte-offset kk*32 + g*16 feeds
//   bf16: one 32x32x16 MFMA;  f32: four 32x32x2 MFMAs (element s of the float4 -> MFMA s).
//   C/D: acc reg r of lane l -> row (r&3) + 8*(r>>2) + 4*g, col l&31   (dtype independent on gfx950).
#pragma once
#include <type_traits>
#include "common.hpp"

namespace mi {

enum { A_CONV = 0, A_DECONV = 1 };
enum { B_KN = 0, B_NK = 1, B_DECONV = 2 };

constexpr int GEMM_BM = 128;
constexpr int GEMM_NT = 256;
constexpr int GEMM_PITCH = 80;      // bytes per LDS row: 64 B of K + 16 B pad

struct GemmParams {
    // ---- A operand: NHWC activation tensor viewed through an im2col (A_CONV) or deconv-gather (A_DECONV) map ----
    const void* a;
    const int* a_frame_idx;          // optional b -> frame indirection (minibatch gather fused into the loader)
    long long a_frame_stride;        // elements per frame = IH*IW*C
    int IH, IW, C;                   // A tensor dims
    int OH, OW;                      // A_CONV: output grid (rows m = (b,oh,ow)); A_DECONV: output tensor dims
    int KH, KW, stride;
    int M, N, K;                     // GEMM dims (A_DECONV: per-class M,K derived in-kernel)
    int nbatch;
    FastDiv div_ohw, div_ow, div_run, div_kw;
    int run, merged;                 // contiguous k-run: C (seg = kh*KW+kw) or KW*C when merged (seg = kh)
    int OHc[2], OWc[2], Th[2], Tw[2];
    FastDiv dc_ohw[4], dc_ow[4], dc_c, dc_tw[2];
    // ---- B operand (weights) ----
    const void* b;
    int ldb;
    int b_vec;                       // 1: vector loads legal (alignment + divisibility checked on host)
    // ---- epilogue ----
    void* out;
    const float* bias;               // fp32 master bias or nullptr
    const void* mask;                // same layout as out; out = mask > 0 ? v : 0  (ReLU-grad) or nullptr
    int relu;
    int out_f32;                     // 1: store fp32 (split-K slabs / fp32 consumers) regardless of T
    int ksplit_len;                  // K range per blockIdx.z (multiple of BK); 0 = single pass
};

template <typename TT, int N, int ALIGN> struct PackU { TT v[N]; } __attribute__((packed, aligned(ALIGN)));

template <typename T> struct Frag;
template <> struct Frag<float> {
    typedef f32x4 reg;
    static __device__ __forceinline__ void mma(const reg& a, const reg& b, f32x16& c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], c, 0, 0, 0);
    }
};
template <> struct Frag<bf16_t> {
    typedef u16x8 reg;
    static __device__ __forceinline__ void mma(const reg& a, const reg& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// split storage: a fragment is four (lo, hi) elements = the bf16x8 vector [l0 h0 l1 h1 l2 h2 l3 h3] per lane.  a . b + a . swap16(b) = the
// full (ha + la)(hb + lb) products of the lane group's four k-values (common.hpp); the swap is one v_alignbit_b32 per dword of the FIRST operand
// (callers pass the operand they reuse across the inner tile loop first, so the compiler hoists the four rotates out of it).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <> struct Frag<split_t> {
    typedef u32x4 reg;
    static __device__ __forceinline__ reg swap16(const reg& a) {
        reg r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = __builtin_amdgcn_alignbit(a[e], a[e], 16);
        return r;
    }
    static __device__ __forceinline__ void mma(const reg& a, const reg& b, f32x16& c) {
        const reg as = swap16(a);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, as), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

__device__ __forceinline__ int xcd_remap(int bid, int n) {
    // give each of the 8 XCDs (block b runs on XCD b%8) a contiguous chunk of tiles: neighbours share halo rows in L2
    int q = n >> 3, r = n & 7, x = bid & 7, w = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + w;
}

// Epilogue shared by gemm_kernel and gemm2_kernel: lane owns pixel m = tile row lrow;
// acc[i][j][4q+t] = C[m][n_base + 8q + t], n_base = subtile + 4*lgrp.  +bias, ReLU, ReLU-grad mask, store T / fp32.
template <typename T, int AMODE, int TM, int TN, typename P>
__device__ __forceinline__ void store_tile(const P& p, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                           int lrow, int lgrp, int M, int cls, int ph, int pw, int zslab) {
    const T* __restrict__ maskp = (const T*)p.mask;
    const bool vec4 = (p.N & 3) == 0;                     // 4-channel groups never straddle N and stay vector-aligned
    auto row_offset = [&](int m, bool rowok) -> long long {
        if constexpr (AMODE == A_CONV) return ((long long)zslab * M + (rowok ? m : 0)) * p.N;
        else {
            uint32_t b, rem, y, x;
            p.dc_ohw[cls].divmod((uint32_t)(rowok ? m : 0), b, rem);
            p.dc_ow[cls].divmod(rem, y, x);
            return (((long long)b * p.OH + (2 * y + ph)) * p.OW + (2 * x + pw)) * p.N;
        }
    };
    if constexpr (sizeof(T) == 2) {
        // bf16, whole 16-channel groups (wave-uniform condition: all lanes swap): pair the 4-channel groups of the two half-waves with
        // v_permlane32_swap and write 16 bytes per lane.  Two phases: every address, bias vector and ReluGrad-mask vector is requested
        // first, then the results are converted and stored back to back -- a load between two stores makes the wave wait for the
        // acknowledgement of the earlier stores (vmcnt counts both), which cost more than the whole main loop of the small layers.
        if ((p.N & 15) == 0 && !p.out_f32) {
            long long off[TM][TN][2]; bool ok[TM][TN];
            PackN<uint32_t, 4> mk[TM][TN][2];
            f32x4 bs[TN][2][2];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nt = n0 + (wn * TN + j) * 32;
#pragma unroll
                for (int mq = 0; mq < 2; ++mq)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        bs[j][mq][h] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (p.bias && nt < p.N) bs[j][mq][h] = *(const f32x4*)(p.bias + nt + 4 * lgrp + 16 * mq + 8 * h);
                    }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + (wm * TM + i) * 32 + lrow;
                const bool rowok = m < M;
                const long long rowoff = row_offset(m, rowok);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nt = n0 + (wn * TN + j) * 32;
                    ok[i][j] = rowok && nt < p.N;
#pragma unroll
                    for (int mq = 0; mq < 2; ++mq) {
                        off[i][j][mq] = ok[i][j] ? rowoff + nt + 16 * mq + 8 * lgrp : 0;
                        if (maskp) mk[i][j][mq] = *(const PackN<uint32_t, 4>*)(maskp + off[i][j][mq]);     // offset 0 is always readable
                    }
                }
            }
            const int lo = p.relu ? 0 : (int)0x80000000;  // ReLU as ONE integer max on the bit pattern (negative floats are negative integers)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int mq = 0; mq < 2; ++mq) {
                        float va[4], vb[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float a = acc[i][j][8 * mq + t] + bs[j][mq][0][t], b = acc[i][j][8 * mq + 4 + t] + bs[j][mq][1][t];
                            const int ia = __builtin_bit_cast(int, a), ib = __builtin_bit_cast(int, b);
                            va[t] = __builtin_bit_cast(float, ia > lo ? ia : lo); vb[t] = __builtin_bit_cast(float, ib > lo ? ib : lo);
                        }
                        const PackN<T, 4> pa = pack4<T>(va), pb = pack4<T>(vb);
                        uint32_t ax = (uint32_t)pa.v[0] | ((uint32_t)pa.v[1] << 16), ay = (uint32_t)pa.v[2] | ((uint32_t)pa.v[3] << 16);
                        uint32_t bx = (uint32_t)pb.v[0] | ((uint32_t)pb.v[1] << 16), by = (uint32_t)pb.v[2] | ((uint32_t)pb.v[3] << 16);
                        auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false); ax = r0[0]; bx = r0[1];
                        auto r1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false); ay = r1[0]; by = r1[1];
                        uint32_t w4[4] = {ax, ay, bx, by};
                        if (maskp) {
#pragma unroll
                            for (int d = 0; d < 4; ++d) {       // bf16 > 0  <=>  signed 16-bit pattern > 0
                                const short lo16 = (short)(mk[i][j][mq].v[d] & 0xffffu), hi16 = (short)(mk[i][j][mq].v[d] >> 16);
                                w4[d] = (lo16 > 0 ? w4[d] & 0xffffu : 0u) | (hi16 > 0 ? w4[d] & 0xffff0000u : 0u);
                            }
                        }
                        if (ok[i][j]) *(PackN<uint32_t, 4>*)((T*)p.out + off[i][j][mq]) = PackN<uint32_t, 4>{{w4[0], w4[1], w4[2], w4[3]}};
                    }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + (wm * TM + i) * 32 + lrow;
        const bool rowok = m < M;
        const long long rowoff = row_offset(m, rowok);
        if (!rowok) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + (wn * TN + j) * 32 + 4 * lgrp + 8 * q;
                if (n >= p.N) continue;
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = acc[i][j][4 * q + t];
                if (vec4) {
                    if (p.bias) {
                        const f32x4 bb = *(const f32x4*)(p.bias + n);
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] += bb[t];
                    }
                    if (p.relu) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
                    }
                    if (maskp) {
                        const PackN<T, 4> mk = *(const PackN<T, 4>*)(maskp + rowoff + n);
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = Elem<T>::to_f32(mk.v[t]) > 0.f ? v[t] : 0.f;
                    }
                    if (p.out_f32) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *(f32x4*)((float*)p.out + rowoff + n) = o;
                    } else {
                        PackN<T, 4> o;
#pragma unroll
                        for (int t = 0; t < 4; ++t) o.v[t] = Elem<T>::from_f32(v[t]);
                        *(PackN<T, 4>*)((T*)p.out + rowoff + n) = o;
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (n + t >= p.N) continue;
                        float x = v[t];
                        if (p.bias) x += p.bias[n + t];
                        if (p.relu) x = fmaxf(x, 0.f);
                        if (maskp) x = Elem<T>::to_f32(maskp[rowoff + n + t]) > 0.f ? x : 0.f;
                        if (p.out_f32) ((float*)p.out)[rowoff + n + t] = x;
                        else ((T*)p.out)[rowoff + n + t] = Elem<T>::from_f32(x);
                    }
                }
            }
        }
    }
}

// T   : storage/MFMA element type (float | bf16_t)       TIn : element type of the A tensor in HBM (float | T)
// VA  : elements per A vector load (must divide the contiguous k-run)   AALIGN: guaranteed byte alignment of an A vector
template <typename T, typename TIn, int AMODE, int BMODE, int VA, int AALIGN, int BN>
__global__ __launch_bounds__(GEMM_NT) void gemm_kernel(const GemmParams p) {
    constexpr int BM = GEMM_BM, NT = GEMM_NT, PITCH = GEMM_PITCH;
    constexpr int BK = 64 / (int)sizeof(T);
    constexpr int VB = 16 / (int)sizeof(T);
    constexpr int VPR_A = BK / VA;
    constexpr int NVA = BM * VPR_A / NT;
    constexpr int RSTEP_A = NT / VPR_A;
    constexpr int WN = (BN >= 64) ? 2 : 1;
    constexpr int WM = 4 / WN;
    constexpr int TM = BM / WM / 32;
    constexpr int TN = BN / WN / 32;
    static_assert(NVA >= 1 && TM >= 1 && TN >= 1, "tile config");
    typedef typename Frag<T>::reg freg;

    __shared__ __attribute__((aligned(16))) unsigned char lds[2][(BM + BN) * PITCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = lane & 31, lgrp = lane >> 5;

    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const int n0 = blockIdx.y * BN;

    int M = p.M, K = p.K;
    int cls = 0, ph = 0, pw = 0, Tw = 1;
    if constexpr (AMODE == A_DECONV) {
        cls = blockIdx.z; ph = cls >> 1; pw = cls & 1;
        M = p.nbatch * p.OHc[ph] * p.OWc[pw];
        Tw = p.Tw[pw];
        K = p.Th[ph] * Tw * p.C;
        if (m0 >= M) return;
    }
    int kbeg = 0, kend = K;
    if constexpr (AMODE == A_CONV) {
        if (p.ksplit_len > 0) { kbeg = blockIdx.z * p.ksplit_len; kend = min(K, kbeg + p.ksplit_len); }
    }
    const int nk = (kend - kbeg + BK - 1) / BK;

    // ---------------- per-thread A row state ----------------
    const int a_kv = tid % VPR_A;
    long long a_base[NVA];
    int a_y[NVA], a_x[NVA];
    bool a_ok[NVA];
    const TIn* __restrict__ Ag = (const TIn*)p.a;
#pragma unroll
    for (int i = 0; i < NVA; ++i) {
        const int m = m0 + tid / VPR_A + i * RSTEP_A;
        a_ok[i] = m < M;
        const uint32_t mm = a_ok[i] ? (uint32_t)m : 0u;
        uint32_t b, rem, y, x;
        if constexpr (AMODE == A_CONV) {
            p.div_ohw.divmod(mm, b, rem);
            p.div_ow.divmod(rem, y, x);
            const long long fr = p.a_frame_idx ? (long long)p.a_frame_idx[b] : (long long)b;
            a_base[i] = fr * p.a_frame_stride + ((long long)(y * p.stride) * p.IW + x * p.stride) * p.C;
            a_y[i] = 0; a_x[i] = 0;
        } else {
            p.dc_ohw[cls].divmod(mm, b, rem);
            p.dc_ow[cls].divmod(rem, y, x);
            a_base[i] = (long long)b * p.a_frame_stride;
            a_y[i] = (int)y; a_x[i] = (int)x;
        }
    }

    PackN<TIn, VA> a_reg[NVA];
    auto load_a = [&](int ks) {
        // branch-free guarded gathers: an out-of-range vector reads the (always legal) tensor base and is zeroed by a
        // select, so all of a thread's loads issue back to back instead of one exec-masked block (+ wait) each
        const int k0 = kbeg + ks * BK + a_kv * VA;
        const bool kok = k0 < kend;
        if constexpr (AMODE == A_CONV) {
            uint32_t seg, j;
            p.div_run.divmod((uint32_t)(kok ? k0 : 0), seg, j);
            long long off;
            if (p.merged) off = (long long)seg * p.IW * p.C + j;
            else { uint32_t kh, kw; p.div_kw.divmod(seg, kh, kw); off = ((long long)kh * p.IW + kw) * p.C + j; }
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const bool ok = a_ok[i] && kok;
                const TIn* src = ok ? Ag + a_base[i] + off : Ag;
                const PackU<TIn, VA, AALIGN> t = *(const PackU<TIn, VA, AALIGN>*)src;
#pragma unroll
                for (int e = 0; e < VA; ++e) a_reg[i].v[e] = ok ? t.v[e] : zero_of<TIn>();
            }
        } else {
            uint32_t tap, c, th, tw;
            p.dc_c.divmod((uint32_t)(kok ? k0 : 0), tap, c);
            p.dc_tw[pw].divmod(tap, th, tw);
#pragma unroll
            for (int i = 0; i < NVA; ++i) {
                const int ih = a_y[i] - (int)th, iw = a_x[i] - (int)tw;
                const bool ok = a_ok[i] && kok && ih >= 0 && ih < p.IH && iw >= 0 && iw < p.IW;
                const TIn* src = ok ? Ag + a_base[i] + ((long long)ih * p.IW + iw) * p.C + c : Ag;
                const PackU<TIn, VA, AALIGN> t = *(const PackU<TIn, VA, AALIGN>*)src;
#pragma unroll
                for (int e = 0; e < VA; ++e) a_reg[i].v[e] = ok ? t.v[e] : zero_of<TIn>();
            }
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NVA; ++i) {
            const int r = tid / VPR_A + i * RSTEP_A;
            PackN<T, VA> t;
#pragma unroll
            for (int e = 0; e < VA; ++e) {
                if constexpr (std::is_same<TIn, T>::value) t.v[e] = a_reg[i].v[e];
                else t.v[e] = Elem<T>::from_f32((float)a_reg[i].v[e]);      // fp32 frames into a bf16 / split engine
            }
            *(PackN<T, VA>*)(&lds[buf][r * PITCH + a_kv * VA * (int)sizeof(T)]) = t;
        }
    };

    // ---------------- B operand ----------------
    const T* __restrict__ Bg = (const T*)p.b;
    constexpr int NVB_TOT = BN * BK / VB;                 // 16-byte vectors in the B tile
    constexpr int NVB = (NVB_TOT + NT - 1) / NT;
    PackN<T, VB> b_reg[NVB];
    auto load_b = [&](int ks) {
        const int kb = kbeg + ks * BK;
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + i * NT;
#pragma unroll
            for (int e = 0; e < VB; ++e) b_reg[i].v[e] = zero_of<T>();
            if (v >= NVB_TOT) continue;
            if constexpr (BMODE == B_KN) {                // W[k*ldb + n], n contiguous
                constexpr int VPR = BN / VB;
                const int k = kb + v / VPR, n = n0 + (v % VPR) * VB;
                if (k < kend) {
                    const T* src = Bg + (long long)k * p.ldb + n;
                    if (p.b_vec && n + VB <= p.N) b_reg[i] = *(const PackN<T, VB>*)src;
                    else {
#pragma unroll
                        for (int e = 0; e < VB; ++e) if (n + e < p.N) b_reg[i].v[e] = src[e];
                    }
                }
            } else if constexpr (BMODE == B_NK) {         // W[n*ldb + k], k contiguous
                constexpr int VPR = BK / VB;
                const int n = n0 + v / VPR, k0 = kb + (v % VPR) * VB;
                const bool ok = n < p.N && k0 < kend;
                if (p.b_vec) {                            // wave-uniform: whole vectors are in range (K % VB == 0)
                    const T* src = ok ? Bg + (long long)n * p.ldb + k0 : Bg;
                    const PackN<T, VB> t = *(const PackN<T, VB>*)src;
#pragma unroll
                    for (int e = 0; e < VB; ++e) b_reg[i].v[e] = ok ? t.v[e] : zero_of<T>();
                } else if (ok) {
                    const T* src = Bg + (long long)n * p.ldb + k0;
#pragma unroll
                    for (int e = 0; e < VB; ++e) if (k0 + e < kend) b_reg[i].v[e] = src[e];
                }
            } else {                                      // deconv weights W[kh][kw][n][c], (tap,c) -> k
                constexpr int VPR = BK / VB;
                const int n = n0 + v / VPR, k0 = kb + (v % VPR) * VB;
                const bool ok = n < p.N && k0 < kend;
                uint32_t tap, c, th, tw;
                p.dc_c.divmod((uint32_t)(ok ? k0 : 0), tap, c);
                p.dc_tw[pw].divmod(tap, th, tw);
                const int kh = ph + 2 * (int)th, kw = pw + 2 * (int)tw;
                const T* src = ok ? Bg + (((long long)kh * p.KW + kw) * p.N + n) * p.C + c : Bg;
                const PackN<T, VB> t = *(const PackN<T, VB>*)src;
#pragma unroll
                for (int e = 0; e < VB; ++e) b_reg[i].v[e] = ok ? t.v[e] : zero_of<T>();
            }
        }
    };
    auto store_b = [&](int buf) {
        unsigned char* Bs = &lds[buf][BM * PITCH];
#pragma unroll
        for (int i = 0; i < NVB; ++i) {
            const int v = tid + i * NT;
            if (v >= NVB_TOT) continue;
            if constexpr (BMODE == B_KN) {
                constexpr int VPR = BN / VB;
                const int kk = v / VPR, nn = (v % VPR) * VB;
#pragma unroll
                for (int e = 0; e < VB; ++e) *(T*)(&Bs[(nn + e) * PITCH + kk * (int)sizeof(T)]) = b_reg[i].v[e];
            } else {
                constexpr int VPR = BK / VB;
                const int nn = v / VPR, kv = v % VPR;
                *(PackN<T, VB>*)(&Bs[nn * PITCH + kv * 16]) = b_reg[i];
            }
        }
    };

    // ---------------- main loop ----------------
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) { load_a(0); load_b(0); store_a(0); store_b(0); }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        const bool more = ks + 1 < nk;
        if (more) { load_a(ks + 1); load_b(ks + 1); }
        const unsigned char* As = &lds[cur][0];
        const unsigned char* Bs = &lds[cur][BM * PITCH];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            freg af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *(const freg*)(&As[((wm * TM + i) * 32 + lrow) * PITCH + kk * 32 + lgrp * 16]);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *(const freg*)(&Bs[((wn * TN + j) * 32 + lrow) * PITCH + kk * 32 + lgrp * 16]);
            // operands swapped on purpose: D[row = channel][col = pixel], so a lane owns ONE pixel (col = lane&31) and
            // 16 channels of it -> the epilogue decodes one pixel address per lane and stores 4-channel vectors
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) Frag<T>::mma(bf[j], af[i], acc[i][j]);
        }
        if (more) { store_a(cur ^ 1); store_b(cur ^ 1); }
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    store_tile<T, AMODE, TM, TN>(p, acc, m0, n0, wm, wn, lrow, lgrp, M, cls, ph, pw,
                                 (AMODE == A_CONV && p.ksplit_len > 0) ? (int)blockIdx.z : 0);
}

}  // namespace mi
