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
#include "tapwgrad_tile.hpp"

namespace mi {

constexpr int GN_BMT = 128;          // positions per block (4 waves x 32)
constexpr int GN_NT = 256;

// chunk swizzle for PA-byte LDS rows read with ds_read_b128 by 32 consecutive rows (see gemm2_tile.hpp / tapconv_tile.hpp)
template <int CPR> __device__ __forceinline__ int gn_swz(int row) { return CPR == 8 ? (row >> 1) & 7 : (row >> 2) & 3; }

// C (input channels) * sizeof(T) must be 64 or 128 bytes: CPR = 4 | 8 sixteen-byte chunks per pixel
// NOUT: output channels when known at compile time (3 rgb, 1 segmentation: packed dword stores), 0 = generic element stores
// FASTBCE (bf16 storage, loss_kind 0 only): the fused sigmoid-cross-entropy on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32,
// 1 ulp each, arguments in (1, 2]) -- 13 VALU instructions per logit against 93 for the correctly rounded library forms.  The layer is VALU-issue
// bound (a wave64 VALU instruction holds the 16-lane SIMD for 4 cycles; 1,100 of them per 32 positions were 55 us of a 72 us launch), so the
// instruction count IS its run time; the fp32 engine (parity mode) keeps the library forms.
template <typename T, int TAPS, int CPR, int NOUT, bool FASTBCE = false>
__global__ __launch_bounds__(GN_NT) __attribute__((amdgpu_waves_per_eu(7, 8))) void gather_narrow_kernel(const TapParams p) {
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

    constexpr int OTILE = 4 * 32 * OPITCH * 4;            // the output transpose reuses the slot tile (after a block barrier)
    constexpr int LBUF = MAXSLOT * PA > OTILE ? MAXSLOT * PA : OTILE;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LBUF + 64];       // + 16 floats for the fused-loss block reduction
    float* const otile = (float*)lds;

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

    // ---- this lane's output pixels (epilogue role: position lrow of the wave's 32, output-row parity lgrp) ----
    const int P = P0 + wave * 32 + lrow;
    uint32_t g, gx, b, gy;
    p.div_gw.divmod((uint32_t)(P < p.MP ? P : 0), g, gx);
    p.div_g.divmod(g, b, gy);
    const int ph = lgrp;
    const int oy = 2 * (int)gy + ph, ox = 2 * (int)gx;
    const bool live = P < p.MP && oy < p.OH && ox < p.OW;
    const int npx = ox + 1 < p.OW ? 2 : 1;                // output pixels of this lane (pw = 0, 1)
    // ---- weights: fragment (tap, kk) of lane (row ne = lrow, chunk 2 kk + lgrp) straight from global / L2 ----
    freg wf[NT][NKK];
    {
        const int ne = lrow;
        const bool rowok = ne < p.NE;
        const uint32_t cls = rowok ? p.div_n.div((uint32_t)ne) : 0u;
        const int n = ne - (int)cls * p.N;
        // through a descriptor that ends with the kernel: rows / taps / channel chunks outside it read 0 from the range check (no per-element selects)
        const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.b, 0, p.KH * p.KW * p.N * p.C * ESZ, 0x00020000);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int ta = tap / TAPS, tb = tap % TAPS;
            const int kh = (int)(cls >> 1) + 2 * (p.HY - ta), kw = (int)(cls & 1) + 2 * (p.HX - tb);
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int c0 = (2 * kk + lgrp) * VE;
                const bool ok = rowok && kh < p.KH && kw < p.KW && c0 < p.C;
                const uint32_t off = ok ? (uint32_t)((((kh * p.KW + kw) * p.N + n) * p.C + c0) * ESZ) : G2_OOB;
                wf[tap][kk] = __builtin_bit_cast(freg, __builtin_amdgcn_raw_buffer_load_b128(rsW, (int)off, 0, 0));
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

    // labels of the fused loss: requested here -- after the MFMAs (holding them through the main phase costs the registers that decide
    // between 5 and 7 resident blocks per CU) but BEFORE the logits store (a load behind a store waits for the store's acknowledgement)
    constexpr int CNTL = NOUT > 0 ? 2 * NOUT : 2;
    float yv[CNTL];
#pragma unroll
    for (int j = 0; j < CNTL; ++j) yv[j] = 0.f;
    if constexpr (NOUT > 0) {
        if (p.labels && live && npx == 2) {
            const long long fr = p.lab_idx ? (long long)p.lab_idx[b] : (long long)b;
            if (p.lab_u8) {                                // raw camera bytes: 2 NOUT consecutive bytes at an even offset; exact k / 255 in registers
                const unsigned char* y = (const unsigned char*)p.labels + fr * p.lab_stride + ((long long)oy * p.OW + ox) * NOUT;
#pragma unroll
                for (int d = 0; d < CNTL / 2; ++d) {
                    const PackU<unsigned char, 2, 2> t = *(const PackU<unsigned char, 2, 2>*)(y + 2 * d);
                    yv[2 * d] = u8_to_unit_exact((float)t.v[0]); yv[2 * d + 1] = u8_to_unit_exact((float)t.v[1]);
                }
            } else {
                const float* y = p.labels + fr * p.lab_stride + ((long long)oy * p.OW + ox) * NOUT;
#pragma unroll
                for (int d = 0; d < CNTL / 2; ++d) {
                    const PackU<float, 2, (NOUT * 8) % 8 == 0 ? 8 : 4> t = *(const PackU<float, 2, (NOUT * 8) % 8 == 0 ? 8 : 4>*)(y + 2 * d);
                    yv[2 * d] = t.v[0]; yv[2 * d + 1] = t.v[1];
                }
            }
        }
    }

    // ---- epilogue: D -> wave-private [position][32] floats -> one lane per (position, output-row parity) ----
    __syncthreads();                                      // every wave is done reading the slot tile
    float* ot = otile + wave * 32 * OPITCH;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        f32x4 v = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
        *(f32x4*)(ot + lrow * OPITCH + 8 * qd + 4 * lgrp) = v;
    }
    __builtin_amdgcn_wave_barrier();                      // same wave, in-order LDS queue: no wait needed, only keep the order
    T* __restrict__ out = (T*)p.out + (((long long)b * p.OH + oy) * p.OW + ox) * p.N;
    const float* src = ot + lrow * OPITCH + ph * 2 * p.N;
    float lsum = 0.f, gs0 = 0.f, gs1 = 0.f, gs2 = 0.f;   // fused loss: this lane's loss terms and per-channel dlogits sums
    bool done = !live;
    if constexpr (NOUT > 0) {
        if (live && npx == 2) {                           // 2 N contiguous elements = ND dwords, adjacent lanes adjacent bytes
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
            } else if constexpr (is_split<T>::value) {
#pragma unroll
                for (int d = 0; d < ND; ++d) w[d] = split_from_f32(v[d]).u;
            } else {
#pragma unroll
                for (int d = 0; d < ND; ++d) w[d] = __builtin_bit_cast(uint32_t, v[d]);
            }
            if (p.out) {                                   // out == nullptr (fused loss only): the logits never go to HBM
                uint32_t* o32 = (uint32_t*)out;
#pragma unroll
                for (int d = 0; d < ND; ++d) o32[d] = w[d];
            }
            if (p.labels) {
                // reconstruction loss on the STORED logits (vae/models.py:11-22,123-128; same math as recon_loss_kernel)
                // (labels yv[]: requested at the top of the kernel)
                T gq[CNT];
#pragma unroll
                for (int j = 0; j < CNT; ++j) {
                    float xv;
                    if constexpr (ESZ == 2) xv = bf16_to_f32((bf16_t)(j & 1 ? w[j >> 1] >> 16 : w[j >> 1] & 0xffffu));
                    else if constexpr (is_split<T>::value) { split_t sv; sv.u = w[j]; xv = split_to_f32(sv); }
                    else xv = v[j];
                    if constexpr (FASTBCE) {                 // loss_kind 0: max(x, 0) - x y + log(1 + exp(-|x|)), gradient sigmoid(x) - y
                        const float e = __builtin_amdgcn_exp2f(-fabsf(xv) * 1.44269504f);
                        const float s1 = 1.0f + e;
                        const float r = __builtin_amdgcn_rcpf(s1);
                        const float sg = xv >= 0.f ? r : e * r;
                        lsum += fmaf(__builtin_amdgcn_logf(s1), 0.69314718f, fmaf(-xv, yv[j], fmaxf(xv, 0.f)));
                        gq[j] = Elem<T>::from_f32((sg - yv[j]) * p.inv_b);
                        const float gst = Elem<T>::to_f32(gq[j]);
                        const int c = j % NOUT;
                        if (c == 0) gs0 += gst; else if (c == 1) gs1 += gst; else gs2 += gst;
                        continue;
                    }
                    const float e = __expf(-fabsf(xv));
                    const float r = __frcp_rn(1.0f + e);
                    const float sg = xv >= 0.f ? r : e * r;
                    float l, gr;
                    if (p.loss_kind == 0) { l = fmaxf(xv, 0.f) - xv * yv[j] + __logf(1.0f + e); gr = sg - yv[j]; }
                    else if (p.loss_kind == 1) {
                        l = -(yv[j] * __logf(1e-10f + sg) + (1.0f - yv[j]) * __logf(1e-10f + 1.0f - sg));
                        gr = (-yv[j] / (1e-10f + sg) + (1.0f - yv[j]) / (1e-10f + 1.0f - sg)) * sg * (1.0f - sg);
                    } else { const float dd = yv[j] - sg; l = dd * dd; gr = -2.0f * dd * sg * (1.0f - sg); }
                    lsum += l;
                    gq[j] = Elem<T>::from_f32(gr * p.inv_b);
                    const float gst = Elem<T>::to_f32(gq[j]);
                    const int c = j % NOUT;
                    if (c == 0) gs0 += gst; else if (c == 1) gs1 += gst; else gs2 += gst;
                }
                if (p.dlogits) {
                    uint32_t dw[ND];
                    if constexpr (ESZ == 2) {
#pragma unroll
                        for (int d = 0; d < ND; ++d) dw[d] = (uint32_t)gq[2 * d] | ((uint32_t)gq[2 * d + 1] << 16);
                    } else {
#pragma unroll
                        for (int d = 0; d < ND; ++d) dw[d] = __builtin_bit_cast(uint32_t, gq[d]);
                    }
                    uint32_t* d32 = (uint32_t*)((T*)p.dlogits + (((long long)b * p.OH + oy) * p.OW + ox) * p.N);
#pragma unroll
                    for (int d = 0; d < ND; ++d) d32[d] = dw[d];
                }
            }
            done = true;
        }
    }
    if (!done && p.out) {                                 // generic path (odd OW edge, other N): element stores, no fused loss
        const int cnt = npx * p.N;
        for (int j = 0; j < cnt; ++j) {
            const int n = j >= p.N ? j - p.N : j;
            float v = src[j];
            if (p.bias) v += p.bias[n];
            if (p.relu) v = fmaxf(v, 0.f);
            out[j] = Elem<T>::from_f32(v);
        }
    }
    if (p.labels) {                                       // block partials (fixed order -> deterministic; mi_vae_finalize_losses_flat adds them up)
        lsum = wave_sum(lsum); gs0 = wave_sum(gs0); gs1 = wave_sum(gs1); gs2 = wave_sum(gs2);
        float* red = (float*)(lds + LBUF);
        if (lane == 0) { red[wave * 4 + 0] = lsum; red[wave * 4 + 1] = gs0; red[wave * 4 + 2] = gs1; red[wave * 4 + 3] = gs2; }
        __syncthreads();
        if (tid < 4) {
            const float t = (red[tid] + red[4 + tid]) + (red[8 + tid] + red[12 + tid]);
            if (tid == 0) p.lpart[blockIdx.x] = t; else p.bpart[(long long)blockIdx.x * 4 + tid - 1] = t;
        }
    }
}

}  // namespace mi

namespace mi {

// =====================================================================================================================
// narrow_wgrad_kernel — filter gradient of the k x k, s2 layers whose NARROW side has Cs = 1..3 channels:
//   conv1   dW[kh,kw,cs,n]  = sum_pix frame[b, 2oy+kh, 2ox+kw, cs] * dy[b,oy,ox,n]     (frames fp32, gathered through frame_idx)
//   deconv4 dW[kh,kw,co,ci] = sum_pix dlogits[b, 2iy+kh, 2ix+kw, co] * x[b,iy,ix,ci]   (dlogits bf16)
// Both are [KH*KW*Cs <= 64] x [32] outer-product sums over ~1.6 M pixels of the 32-channel tensor: HBM-bound (read both
// tensors once).  Per 16-pixel step a WAVE builds the im2col rows [pixel][64] (bf16, one lane per (pixel, kh): KW*Cs
// contiguous source values) and DMA-stages the 16 x 32 wide-tensor rows, then runs 2 MFMAs on transpose-read operands (+ the
// optional all-ones MFMA for the bias gradient).  Waves never synchronise inside the loop; the four partial tiles of a block
// are added in LDS at the end and leave as one set of atomics per block.
// =====================================================================================================================
struct NarrowWgradParams {
    const void* src; const int* frame_idx; long long frame_stride;    // narrow tensor, elements per frame
    const void* s; uint32_t s_bytes;                                  // wide tensor [B,OH,OW,32] bf16
    int B, IH, IW, Cs, OH, OW, KH, KW;
    int M, pix_per_block;                                             // M = B*OH*OW ; pixels per WAVE, multiple of 16
    FastDiv div_ohw, div_ow;
    float* out; float* dbias;
    float* slabs;                                                     // optional [gridDim.x][NW_SLAB] per-block partial sums (then reduce_slabs_kernel)
    int dbg_skip_out;                                                 // debug (mi_set_tuning key 2 == 1): drop the final atomics (wrong results): shows their cost
};

constexpr int NW_SLAB = 64 * 32 + 32;  // dW rows (padded to 64) x 32, then the 32 bias sums
constexpr int NW_BP = 16;            // pixels per wave step

// NGRP: 4-value groups per (pixel, kernel row) known at compile time (3 for the 4 x 4 x 3-channel layers), 0 = runtime;
// BIAS: 1 = with the all-ones MFMA of the bias gradient, 0 = without, -1 = runtime.  With both fixed the 3-step loop body is ONE basic
// block, which is what lets hipcc keep partial `s_waitcnt vmcnt(N)` counts across the back edge: with branches in the body (and with
// the frame-index lookup as a dependent load inside it) every step drained the whole load queue and the "3-deep" pipeline ran 1 deep.
// NWV: waves per block.  All blocks run one equal share of the pixels and finish together, so their final atomics (2080 addresses, one
// add per block and address) queue up at the end: 18 us of a 73 us launch with 747 four-wave blocks; 12-wave blocks (one per CU) cut the
// queue depth by three.
constexpr int NW_WAVES = 12;
// DEPTH: steps in flight per wave (register pipeline).  The kernel is latency-bound -- 75 % of its wave cycles wait, VALU 20 %, MFMA 5 % (profiles/r03_b) -- so
// what a wave has in flight IS its speed: 3 steps = 5.3 KB per wave, 64 KB per CU.
template <typename TS, int NGRP = 0, int BIAS = -1, int NWV = NW_WAVES, int DEPTH = 3>
__global__ __launch_bounds__(NWV * 64) void narrow_wgrad_kernel(const NarrowWgradParams p) {
    constexpr int PA = 128, PS = 64;                      // LDS row pitch: im2col rows (64 bf16), wide rows (32 bf16)
    constexpr int WSTG = NW_BP * PA + NW_BP * PS;         // per wave: one im2col tile + one wide-row tile = 3 KB
    constexpr int RED = 3 * 1024 * 4;                     // cross-wave reduction buffer
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NWV * WSTG + RED];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // every wave owns a contiguous pixel range and runs on its own (no block barrier until the final reduction): memory
    // latency is hidden by the ~20 resident waves per CU, not by a per-wave software pipeline
    const int wid = blockIdx.x * NWV + wave;
    const int mbeg = min(p.M, wid * p.pix_per_block);     // pix_per_block: pixels per WAVE here (multiple of 16)
    const int mend = min(p.M, mbeg + p.pix_per_block);
    const int nsteps = (mend - mbeg + NW_BP - 1) / NW_BP;
    const int run = p.KW * p.Cs;                          // contiguous source values per (pixel, kh): 4 | 8 | 12
    const int ngrp = NGRP > 0 ? NGRP : run >> 2;
    // frame gather: the (at most three: the host keeps a wave's range within 2 x OH x OW pixels) frames this wave's pixel range
    // touches are looked up ONCE, here
    const uint32_t b0 = p.div_ohw.div((uint32_t)(mbeg < p.M ? mbeg : 0));
    long long fr3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const uint32_t bb = min(b0 + (uint32_t)k, (uint32_t)(p.B - 1));
        fr3[k] = p.frame_idx ? (long long)p.frame_idx[bb] : (long long)bb;
    }

    unsigned char* const wl = lds + wave * WSTG;          // this wave's tiles: [0, 2 KB) im2col, then 1 KB of wide rows
    const TS* __restrict__ src = (const TS*)p.src;

    const int tp = lane >> 2, kh = lane & 3;              // im2col role: (pixel of the step, kernel row)
    // Everything is fetched with ordinary loads (in-order return, so hipcc's own vmcnt(N) bookkeeping lets the loads of the
    // next two steps stay in flight while this step is consumed): 12 source values + one 16-byte piece of the wide rows per lane.
    // (the loaded vectors are kept RAW: converting or zero-filling them at load time would be a use of the load result inside the same
    //  iteration, i.e. a wait for it before the back edge -- the validity flags travel with them and are applied in store_step)
    typedef PackU<TS, 4, (int)sizeof(TS) * 2> SrcVec;   // uint8 camera frames: 4 bytes at 2-byte alignment (pixel offsets are multiples of 6 bytes)
    struct StepRegs { SrcVec raw[3]; f32x4 s; bool ok; bool sok; };
    auto load_step = [&](int step, StepRegs& R) {
        const int m = mbeg + step * NW_BP + tp;
        const bool ok = m < mend && kh < p.KH;
        uint32_t b, rem, y, x;
        p.div_ohw.divmod((uint32_t)(ok ? m : 0), b, rem);
        p.div_ow.divmod(rem, y, x);
        const long long fr = b == b0 ? fr3[0] : (b == b0 + 1 ? fr3[1] : fr3[2]);
        const TS* row = src + fr * p.frame_stride + ((long long)(2 * y + kh) * p.IW + 2 * x) * p.Cs;
        R.ok = ok;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const bool gok = ok && g < ngrp;
            R.raw[g] = *(const SrcVec*)(gok ? row + 4 * g : src);
        }
        // wide rows: lane -> row lane / 4 (pixel tp), 16-byte chunk lane % 4 (= kh)
        const bool sok = m < mend;
        R.sok = sok;
        R.s = *(const f32x4*)((const unsigned char*)p.s + (sok ? ((long long)m * 32 + kh * 8) * 2 : 0));
    };
    auto store_step = [&](const StepRegs& R) {            // im2col row tp, columns kh*run .. (columns >= KH*run stay zero) + the wide-row piece
        const int sw = ((tp >> 1) & 1) << 2;
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if (g >= ngrp) break;
            const int col = kh * run + 4 * g;             // multiple of 4 elements = 8 bytes
            float f4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = src_to_f32<TS>(R.raw[g].v[e]);
                f4[e] = R.ok ? f : 0.f;
            }
            *(PackN<bf16_t, 4>*)(wl + tp * PA + ((((col >> 3) ^ sw)) << 4) + (col & 7) * 2) = pack4<bf16_t>(f4);
        }
        *(f32x4*)(wl + NW_BP * PA + tp * PS + kh * 16) = R.sok ? R.s : f32x4{0.f, 0.f, 0.f, 0.f};
    };

    for (int i = lane; i < NW_BP * PA / 16; i += 64) *(f32x4*)(wl + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};   // zero the im2col tile once

    // transpose-read offsets (tr_fragment layout): row = (lane>>5)*8 + ((lane&15)>>2), column = ((lane>>4)&1)*16 + (lane&3)*4
    const int trow = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int tcol = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    uint32_t aoff[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        const int col = kt * 32 + tcol;
        aoff[kt] = (uint32_t)(trow * PA + ((((col >> 3) ^ (((trow >> 1) & 1) << 2))) << 4) + (col & 7) * 2);
    }
    const uint32_t soff = (uint32_t)(NW_BP * PA + trow * PS + tcol * 2);

    f32x16 acc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    const u16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    typedef __attribute__((address_space(3))) s16x4* lds_v4;

    auto compute = [&]() {
        u16x8 sf;
        {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(wl + soff));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(wl + soff + 4 * PS));
#pragma unroll
            for (int e = 0; e < 4; ++e) { sf[e] = (unsigned short)lo[e]; sf[4 + e] = (unsigned short)hi[e]; }
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(wl + aoff[kt]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(wl + aoff[kt] + 4 * PA));
            u16x8 af;
#pragma unroll
            for (int e = 0; e < 4; ++e) { af[e] = (unsigned short)lo[e]; af[4 + e] = (unsigned short)hi[e]; }
            acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, sf), acc[kt], 0, 0, 0);
        }
        if (BIAS > 0 || (BIAS < 0 && p.dbias)) acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, sf), acc[2], 0, 0, 0);
    };

    // DEPTH-deep register pipeline (steps s .. s + DEPTH - 1 in flight); out-of-range steps load nothing and store zeros
    StepRegs R[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) load_step(d, R[d]);
    for (int step = 0; step < nsteps; step += DEPTH) {   // steps past the range load nothing and add zeros: no branch in the body
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            load_step(step + u + DEPTH - 1, R[(u + DEPTH - 1) % DEPTH]);
            store_step(R[u]); compute();
        }
    }

    // cross-wave reduction (waves take turns on one 12 KB buffer), then one set of atomics per block:
    // acc[k][r] -> kc = k*32 + (r&3) + 8(r>>2) + 4(lane>>5), n = lane&31 ; k == 2: all-ones rows (bias gradient)
    float* red = (float*)(lds + NWV * WSTG);
#pragma unroll
    for (int w = 0; w < NWV; ++w) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float* q = &red[(k * 16 + r) * 64 + lane];
                    *q = w == 0 ? acc[k][r] : *q + acc[k][r];
                }
        }
        __syncthreads();
    }
    // same-address atomics from ~1300 blocks serialise (~60 ns each): with scratch the block writes its partial sums plainly
    const int kcn = p.KH * run;
    if (p.dbg_skip_out) return;
    for (int i = tid; i < 3 * 16 * 64; i += NWV * 64) {
        const float sum = red[i];
        const int k = i >> 10, r = (i >> 6) & 15, l = i & 63;
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = l & 31;
        if (k < 2) {
            const int kc = k * 32 + row;
            if (p.slabs) p.slabs[(long long)blockIdx.x * NW_SLAB + kc * 32 + n] = sum;
            else if (kc < kcn) atomicAdd(&p.out[kc * 32 + n], sum);
        } else if (row == 0) {
            if (p.slabs) p.slabs[(long long)blockIdx.x * NW_SLAB + 64 * 32 + n] = sum;
            else if (p.dbias) atomicAdd(&p.dbias[n], sum);
        }
    }
}

}  // namespace mi

namespace mi {

// =====================================================================================================================
// narrow_conv_kernel — conv k4 s2 FROM a 1..3-channel tensor into 32 channels (conv1 fwd: fp32 frames gathered through
// frame_idx; deconv4 dgrad: bf16 dlogits, with the ReLU-grad mask of the layer below).  K = KH*KW*Cs = 48 (16 for one
// channel): three MFMA k-steps per 32 output pixels.  Waves are fully independent (no block barrier): a lane builds its
// half of its pixel's 48-value patch straight from global memory (6 groups of 4 contiguous values; neighbouring pixels
// overlap, L1 absorbs the re-reads), the K-contiguous weight copy (3 KB) sits in registers, and the 32 x 32 result leaves
// through a wave-private LDS transpose as two 16-byte vectors per lane: every store instruction writes 1 KB of full lines.
// Latency is hidden by occupancy (one 32-pixel tile per wave, ~6 waves per SIMD), which is all an HBM-bound layer needs.
// =====================================================================================================================
struct NarrowConvParams {
    const void* src; const int* frame_idx; long long frame_stride;    // narrow input [*,IH,IW,Cs]
    const void* w;                                                    // K-contiguous weights [32][KH*KW*Cs], T
    int B, IH, IW, Cs, OH, OW, KH, KW, M;
    FastDiv div_ohw, div_ow, div_g3;
    const float* bias; int relu; const void* mask; void* out;         // out / mask [B,OH,OW,32], T
    // ReLU bit words (bf16 only; 2 x uint32 per pixel: word j = channels 16 j .. 16 j + 15, bit i = channel 16 j + 2 i > 0, bit 16 + i = channel
    // 16 j + 2 i + 1 > 0): bits_out = written from the stored (post-ReLU) values; mask_bits = read INSTEAD of the 64-byte mask row
    uint32_t* bits_out; const uint32_t* mask_bits;
};

template <typename T, typename TS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void narrow_conv_kernel(const NarrowConvParams p) {
    constexpr int ESZ = (int)sizeof(T);
    constexpr int VE = 16 / ESZ;
    constexpr int OPITCH = 32 * ESZ + 16;                 // output transpose: 32 channels per pixel + 16 B pad
    __shared__ __attribute__((aligned(16))) unsigned char lds[4 * 32 * OPITCH];
    typedef typename Frag<T>::reg freg;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lgrp = lane >> 5;
    const int m0 = (blockIdx.x * 4 + wave) * 32;
    if (m0 >= p.M) return;
    const int K = p.KH * p.KW * p.Cs, run = p.KW * p.Cs, gpr = run >> 2;      // gpr: 4-value groups per kernel row (3 | 1)
    const int NS = sizeof(T) == 2 ? 3 : 6;                // MFMA k-steps covered below: ceil(48 / (32-byte chunk pair))
    constexpr int KPS = 32 / ESZ;                         // k values per step (two 16-byte chunks): 16 bf16 | 8 f32
    constexpr int GPL = KPS / 8;                          // 4-value groups per lane per step: 2 | 1

    // ---- this lane's pixel ----
    const int m = m0 + lrow;
    const bool mok = m < p.M;
    uint32_t b, rem, y, x;
    p.div_ohw.divmod((uint32_t)(mok ? m : 0), b, rem);
    p.div_ow.divmod(rem, y, x);
    // frame gather: the 32 pixels of a wave lie in at most two frames (the host requires OH x OW >= 32); their indices are two wave-uniform
    // (scalar) loads instead of a per-lane load that every patch load then depends on
    const uint32_t b0 = p.div_ohw.div((uint32_t)__builtin_amdgcn_readfirstlane(m0));
    const uint32_t b1 = min(b0 + 1u, (uint32_t)(p.B - 1));
    const long long fr0 = p.frame_idx ? (long long)p.frame_idx[b0] : (long long)b0, fr1 = p.frame_idx ? (long long)p.frame_idx[b1] : (long long)b1;
    const long long fr = b == b0 ? fr0 : fr1;
    const TS* __restrict__ src = (const TS*)p.src;
    const TS* pix = src + fr * p.frame_stride + ((long long)(2 * y) * p.IW + 2 * x) * p.Cs;

    // ---- operands ----
    freg wf[6], xf[6];
    const T* __restrict__ W = (const T*)p.w;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        if (s >= NS) break;
        const int k0 = s * KPS + lgrp * (KPS / 2);        // first k of this lane's chunk (VE consecutive k)
        // weights: row n = lrow
        {
            const bool ok = k0 < K;
            freg v = *(const freg*)(ok ? W + lrow * K + k0 : W);
#pragma unroll
            for (int e = 0; e < VE; ++e) v[e] = ok ? v[e] : (decltype(v[0] + 0))0;
            wf[s] = v;
        }
        // patch: VE consecutive k = GPL groups of 4 values; group q = k / 4 -> kernel row q / gpr, offset (q % gpr) * 4
        float pv[VE];
#pragma unroll
        for (int gi = 0; gi < GPL; ++gi) {
            const int q = (k0 >> 2) + gi;
            uint32_t kh, qr;
            p.div_g3.divmod((uint32_t)q, kh, qr);
            const bool ok = mok && 4 * q < K;
            const PackU<TS, 4, (int)sizeof(TS) * 2> t = *(const PackU<TS, 4, (int)sizeof(TS) * 2>*)(ok ? pix + (long long)kh * p.IW * p.Cs + qr * 4 : src);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f = src_to_f32<TS>(t.v[e]);
                pv[4 * gi + e] = ok ? f : 0.f;
            }
        }
        freg xv;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            if constexpr (sizeof(T) == 2) xv[e] = f32_to_bf16(pv[e]);
            else if constexpr (is_split<T>::value) xv[e] = split_from_f32(pv[e]).u;
            else xv[e] = pv[e];
        }
        xf[s] = xv;
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        if (s >= NS) break;
        Frag<T>::mma(wf[s], xf[s], acc);                  // D[row = channel][col = pixel]
    }

    // ---- epilogue: +bias, ReLU -> wave-private [pixel][32] -> 16-byte vectors (ReLU-grad mask applied there) ----
    unsigned char* ot = lds + wave * 32 * OPITCH;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const int n = 8 * qd + 4 * lgrp;
        float v[4] = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
        if (p.bias) {
            const f32x4 bb = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] += bb[t];
        }
        if (p.relu) {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = fmaxf(v[t], 0.f);
        }
        *(PackN<T, 4>*)(ot + lrow * OPITCH + n * ESZ) = pack4<T>(v);
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int CPP = 32 * ESZ / 16;                    // 16-byte chunks per pixel (4 | 8)
    const T* __restrict__ maskp = (const T*)p.mask;
    constexpr int NK = 32 * CPP / 64;
    PackN<T, VE> mk[NK];                                  // every mask vector is requested before the first store (a load behind a store
    uint32_t mword[NK];                                   //  waits for the store's acknowledgement: vmcnt counts both)
    const bool use_bits = sizeof(T) == 2 && p.mask_bits != nullptr;
    if (use_bits) {                                       // 4 bytes per lane instead of 16: word (c16 >> 1) of the pixel
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int id = lane + 64 * k, px = id / CPP, c16 = id % CPP;
            mword[k] = p.mask_bits[m0 + px < p.M ? (long long)(m0 + px) * 2 + (c16 >> 1) : 0];
        }
    } else if (maskp) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int id = lane + 64 * k, px = id / CPP, c16 = id % CPP;
            mk[k] = *(const PackN<T, VE>*)(maskp + (m0 + px < p.M ? (long long)(m0 + px) * 32 + c16 * VE : 0));
        }
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int id = lane + 64 * k, px = id / CPP, c16 = id % CPP;
        const bool pok = m0 + px < p.M;
        PackN<T, VE> o = *(const PackN<T, VE>*)(ot + px * OPITCH + c16 * 16);
        const long long off = (long long)(m0 + px) * 32 + c16 * VE;
        if constexpr (sizeof(T) == 2) {
            if (use_bits) {                               // pair d of this lane's 8 channels is pair 4 (c16 & 1) + d of the word
                PackN<uint32_t, 4> ow = __builtin_bit_cast(PackN<uint32_t, 4>, o);
                const uint32_t sh = mword[k] >> (4 * (c16 & 1));
#pragma unroll
                for (int d = 0; d < 4; ++d) ow.v[d] &= ((sh >> d) & 0x00010001u) * 0xffffu;
                o = __builtin_bit_cast(PackN<T, VE>, ow);
            }
            if (p.bits_out) {                             // ReLU bits of the values being stored (post-ReLU: never negative)
                const PackN<uint32_t, 4> ow = __builtin_bit_cast(PackN<uint32_t, 4>, o);
                uint32_t part = 0;
#pragma unroll
                for (int d = 0; d < 4; ++d) {             // min(x, 1) per 16-bit half = "non-zero" (one packed op), shifted into place
                    uint32_t nz;
                    asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(ow.v[d]), "v"(0x00010001u));
                    part |= nz << d;
                }
                part <<= 4 * (c16 & 1);
                part |= __shfl_xor(part, 1, 64);          // the neighbouring lane holds the other 8 channels of the same word
                if (pok && (c16 & 1) == 0) p.bits_out[(long long)(m0 + px) * 2 + (c16 >> 1)] = part;
            }
        }
        if (!pok) continue;
        if (!use_bits && maskp) {
#pragma unroll
            for (int t = 0; t < VE; ++t) o.v[t] = Elem<T>::to_f32(mk[k].v[t]) > 0.f ? o.v[t] : zero_of<T>();
        }
        *(PackN<T, VE>*)((T*)p.out + off) = o;
    }
}

// =====================================================================================================================
// narrow_conv48_kernel — the bf16 form of narrow_conv_kernel for the geometry the model has (KH = 4, KW * Cs = 12: K = 48), rebuilt around
// its real bound.  The layer moves 120-150 MB for 5 GFLOP, but at ~570 VALU instructions per 32-pixel tile (a wave64 VALU instruction holds
// the 16-lane SIMD for 4 cycles) 49 k tiles were 45-50 us of VALU issue against a 16-18 us HBM floor.  Here a tile costs ~150:
//   * no per-element selects: rows past M recompute the last pixel and their stores fall outside the output descriptor; weights come
//     through a descriptor; the 6 patch groups of a lane are two precomputed (per half-wave) offset sets,
//   * bias is the INITIAL accumulator (its 16 loads land directly in the MFMA's C operand), ReLU is one v_pk_max_i16 per two outputs,
//   * no LDS: one v_permlane32_swap per dword gives every lane 16 CONSECUTIVE channels of its pixel (32 bytes, two 16-byte stores) --
//     which is also exactly one ReLU bit word, so the words are produced / consumed without crossing lanes.
// MODE 0: out = relu(conv + bias) (conv1 forward), bits_out optional; MODE 1: out = conv masked by mask_bits (deconv4 input gradient).
// =====================================================================================================================
typedef uint32_t nc_u32x4 __attribute__((ext_vector_type(4)));
template <typename TS, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 8))) void narrow_conv48_kernel(const NarrowConvParams p) {
    constexpr int SSZ = (int)sizeof(TS), GSZ = 4 * SSZ;   // bytes of one 4-value patch group
    constexpr int GDW = GSZ / 4;                          // dwords per group: 1 (bytes) | 2 (bf16) | 4 (fp32)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lgrp = lane >> 5;
    const int ntiles = (p.M + 31) >> 5, nw = (int)gridDim.x * 4;
    int tile = (int)blockIdx.x * 4 + wave;
    if (tile >= ntiles) return;
    const uint32_t rowb = (uint32_t)(p.IW * p.Cs * SSZ);

    // ---- per wave, once: weights (MFMA step s, row n = lrow, this half-wave's 8 k), bias as the initial accumulator, descriptors ----
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 32 * 48 * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.M * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsBI = __builtin_amdgcn_make_buffer_rsrc((void*)(MODE == 1 ? (const void*)p.mask_bits : (const void*)p.out), 0, MODE == 1 ? p.M * 8 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsBO = __builtin_amdgcn_make_buffer_rsrc(p.bits_out ? (void*)p.bits_out : p.out, 0, (MODE == 0 && p.bits_out) ? p.M * 8 : 0, 0x00020000);
    u16x8 wf[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) wf[s] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, (lrow * 48 + s * 16 + lgrp * 8) * 2, 0, 0));
    f32x16 acc0;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
    if constexpr (MODE == 0) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const f32x4 bb = *(const f32x4*)(p.bias + 8 * qd + 4 * lgrp);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc0[4 * qd + t] = bb[t];
        }
    }
    // group j = 2 s + gi of this lane: q = 4 s + gi (+ 2 for the upper half-wave) -> kernel row q / 3, value offset (q % 3) * 4
    uint32_t goff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int qa = 4 * (j >> 1) + (j & 1), qb = qa + 2;
        goff[j] = lgrp ? (uint32_t)(qb / 3) * rowb + (uint32_t)((qb % 3) * GSZ) : (uint32_t)(qa / 3) * rowb + (uint32_t)((qa % 3) * GSZ);
    }

    // the raw patch of one tile: 6 groups per lane, requested one tile AHEAD of its use (a wave walks tiles tile, tile + nw, ...; the
    // persistent grid keeps 8 waves per SIMD resident, and within a wave the next tile's loads fly under this tile's arithmetic and stores)
    struct Raw { uint32_t d[6][GDW]; uint32_t mw; };
    // the frame of a lane's pixel (minibatch gather) is looked up TWO tiles ahead: looked up inside request() its load was followed by s_waitcnt vmcnt(0) -- the address of the
    // six patch loads depends on it -- which every tile drained the whole vector-memory queue, the previous tile's store acknowledgements included (round 3; dectail_tile.hpp)
    auto frame_of = [&](int t) -> int {
        const int m = min(t * 32 + lrow, p.M - 1);
        const uint32_t b = p.div_ohw.div((uint32_t)m);
        return p.frame_idx ? p.frame_idx[b] : (int)b;
    };
    auto request = [&](int t, int frame, Raw& r) {
        const int m = min(t * 32 + lrow, p.M - 1);       // pixels past M recompute the last one; their stores fall outside the descriptors
        uint32_t b, rem, y, x;
        p.div_ohw.divmod((uint32_t)m, b, rem);
        p.div_ow.divmod(rem, y, x);
        const long long fr = frame;
        const unsigned char* pix = (const unsigned char*)p.src + fr * p.frame_stride * SSZ + (2u * y * (uint32_t)p.IW + 2u * x) * (uint32_t)(p.Cs * SSZ);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const PackU<uint32_t, GDW, 2> v = *(const PackU<uint32_t, GDW, 2>*)(pix + goff[j]);
#pragma unroll
            for (int e = 0; e < GDW; ++e) r.d[j][e] = v.v[e];
        }
        if constexpr (MODE == 1) r.mw = __builtin_amdgcn_raw_buffer_load_b32(rsBI, (t * 32 + lrow) * 8 + lgrp * 4, 0, 0);
    };
    Raw cur, nxt;
    request(tile, frame_of(tile), cur);
    int fr_nxt = frame_of(min(tile + nw, ntiles - 1));
    for (; tile < ntiles; tile += nw) {
        const int fr_use = fr_nxt;
        fr_nxt = frame_of(min(tile + 2 * nw, ntiles - 1));
        request(min(tile + nw, ntiles - 1), fr_use, nxt);        // (the last iteration re-requests its own tile: no branch in the loop body)
        u16x8 xf[3];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            uint32_t* dst = (uint32_t*)&xf[j >> 1] + 2 * (j & 1);
            if constexpr (SSZ == 2) { dst[0] = cur.d[j][0]; dst[1] = cur.d[j][1]; }      // bf16 source: the fragment's 8 bytes as they are
            else {
                float f[4];
                if constexpr (SSZ == 1) {                 // camera bytes: k * (1 / 255), which rounds to the same bf16 as the exact quotient (common.hpp)
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = (float)((cur.d[j][0] >> (8 * e)) & 255u) * U8_RCP255;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) f[e] = __builtin_bit_cast(float, cur.d[j][e]);
                }
                const PackN<uint32_t, 2> h = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(f));
                dst[0] = h.v[0]; dst[1] = h.v[1];
            }
        }
        f32x16 acc = acc0;
#pragma unroll
        for (int s = 0; s < 3; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[s]), __builtin_bit_cast(bf16x8, xf[s]), acc, 0, 0, 0);

        // ---- epilogue: lane (pixel, lgrp) holds channels 8 qd + 4 lgrp .. + 3 (qd = 0..3) as two dwords R[qd][0..1] ----
        uint32_t R[4][2];
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float v[4] = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
            const PackN<uint32_t, 2> w = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(v));
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                uint32_t u = w.v[d];
                if constexpr (MODE == 0) asm("v_pk_max_i16 %0, %1, 0" : "=v"(u) : "v"(w.v[d]));      // ReLU on two bf16: negative floats are negative int16
                R[qd][d] = u;
            }
        }
        // half-wave exchange: (R[0], R[2]) and (R[1], R[3]) -> lane (pixel, g) owns channels 16 g .. 16 g + 15 in the order R0 R2 R1 R3
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            auto r0 = __builtin_amdgcn_permlane32_swap(R[0][d], R[2][d], false, false); R[0][d] = r0[0]; R[2][d] = r0[1];
            auto r1 = __builtin_amdgcn_permlane32_swap(R[1][d], R[3][d], false, false); R[1][d] = r1[0]; R[3][d] = r1[1];
        }
        uint32_t o[8] = {R[0][0], R[0][1], R[2][0], R[2][1], R[1][0], R[1][1], R[3][0], R[3][1]};       // dword d: channels 16 g + 2 d, + 1
        if constexpr (MODE == 1) {                        // ReluGrad: bit d / bit 16 + d of the word select the two halves of dword d
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] &= ((cur.mw >> d) & 0x00010001u) * 0xffffu;
        }
        const uint32_t mr = (uint32_t)(tile * 32 + lrow);
        const uint32_t obyte = mr * 64u + (uint32_t)lgrp * 32u;
        __builtin_amdgcn_raw_buffer_store_b128(nc_u32x4{o[0], o[1], o[2], o[3]}, rsO, (int)obyte, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(nc_u32x4{o[4], o[5], o[6], o[7]}, rsO, (int)obyte + 16, 0, 0);
        if constexpr (MODE == 0) {                        // ReLU bits of the stored (non-negative) values: min(x, 1) per 16-bit half = "non-zero"
            uint32_t word = 0;
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                uint32_t nz;
                asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(o[d]), "v"(0x00010001u));
                word |= nz << d;
            }
            __builtin_amdgcn_raw_buffer_store_b32(word, rsBO, (int)(mr * 8u + (uint32_t)lgrp * 4u), 0, 0);     // (no bit words wanted: empty descriptor)
        }
        cur = nxt;
    }
}

}  // namespace mi
