// rwconv.hip — "register-weight" MFMA kernel for the THIN stride-2 transposed-conv layers of the decoder / encoder backward
// (gather form: deconv3 fwd 64 -> 32 channels k5, conv2 dgrad 64 -> 32 channels k4; reference vae/models.py:250-253,261-264 and their
// tf.gradients counterparts Conv2DBackpropInput).  gfx950, wave64, bf16 storage / fp32 accumulate.
//
// Why a second kernel next to tapconv_tile.hpp: these layers have ONE 128-byte channel slice (C = 64) and 4-9 taps, i.e. 2-5 barrier steps
// per tapconv block -- the block is all prologue / epilogue, its weight tiles (64-147 KB per 256 positions) stream through LDS behind a
// barrier per step, and a 154 KB block leaves no room for a second one on the CU (round-1 profile: 15-20 % MFMA busy, 12 VALU per MFMA).
// Here
//   * a wave owns ONE output parity class (32 output channels) and keeps that class's weight fragments in REGISTERS for the whole block
//     (taps x 4 k-steps x 4 VGPRs: 64 for k = 4, up to 144 for the 9-tap class of k = 5): no weight traffic through LDS at all;
//   * the block stages its slot range (256 positions + halo, one 128-byte row per slot, zero fill by the buffer range check) ONCE by LDS-DMA;
//     after the single barrier the four waves never synchronise again: each walks the 8 position tiles on its own, one ds_read_b128 +
//     (at most) one v_xor per MFMA, tap shifts are address offsets;
//   * 45 KB of LDS and 4 waves per block: 2-3 blocks are resident per CU, so one block's staging / epilogue overlaps the others' MFMAs;
//   * the epilogue is tapconv's direct form (bias as the accumulator's initial value, integer-max ReLU, hardware bf16 rounding, one
//     v_permlane32_swap per dword, two 16-byte stores per 32 x 32 tile, ReluGrad mask requested before the MFMAs of its tile).
// Slot formulation, swizzle and operand roles are those of tapconv_tile.hpp (gather form).
#include <stdlib.h>
#include <type_traits>
#include "gemm2_tile.hpp"
#include "tapconv_tile.hpp"
#include "mi_internal.hpp"

namespace mi {

// packed 16-bit integer ops on dwords of two bf16 (hipcc lowers the ext-vector formulations to compare / select chains)
__device__ __forceinline__ uint32_t pk_relu_bf16(uint32_t x) {        // max(x, 0) per half as signed 16-bit: negative bf16 (and -0) -> +0
    uint32_t r; asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(x)); return r;
}
__device__ __forceinline__ uint32_t pk_positive_mask(uint32_t m, uint32_t ones, uint32_t ffff) {   // 0xffff per half whose bf16 value is > 0
    uint32_t t, u, r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(t) : "v"(m));                  // negative (sign bit set) -> 0
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(u) : "v"(t), "v"(ones));      // > 0 -> 1
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(u), "v"(ffff));   // 1 -> 0xffff
    return r;
}

constexpr int RW_BMT = 256;          // positions per block
constexpr int RW_MAXHALO = 96;
constexpr int RW_MAXSLOT = RW_BMT + RW_MAXHALO;

// One wave = one output parity class CLS (ph = CLS >> 1, pw = CLS & 1).  TAPS taps per axis, KH kernel size (compile time: which
// (tap, class) pairs exist), C = 64 input channels (one 128-byte row per slot), N = 32 output channels per class.
template <int TAPS, int KH, int CLS, bool RELU, bool MASK>
__device__ __forceinline__ void rw_class(const TapParams& p, const unsigned char* lds, const float* bias_lds, int P0, int lane, long long* tr, int tr_n) {
    constexpr int RB = 128, NT = TAPS * TAPS, NTILE = RW_BMT / 32;
    constexpr int PH = CLS >> 1, PW = CLS & 1, H = TAPS - 1;
    typedef u16x8 freg;
    const int lrow = lane & 31, lgrp = lane >> 5;

    // ---- weights: fragment (tap, kk) = 8 consecutive input channels (kk*16 + lgrp*8) of output channel lrow, straight from L2 ----
    freg wf[NT][4];
    {
        const bf16_t* __restrict__ W = (const bf16_t*)p.b;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            constexpr int dummy = 0; (void)dummy;
            const int ta = tap / TAPS, tb = tap % TAPS;
            const int kh = PH + 2 * (H - ta), kw = PW + 2 * (H - tb);
            if (kh >= KH || kw >= KH) continue;           // literal after unrolling: this class does not reach that kernel row / column
            const bf16_t* wrow = W + ((long long)(kh * KH + kw) * 32 + lrow) * 64 + lgrp * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) wf[tap][kk] = *(const freg*)(wrow + kk * 16);
        }
    }

    // ---- LDS read addresses: row q = lrow + delta_tap (+ 32 per tile, which leaves the swizzle term alone); rows are 128 B, bits 4..6 are
    //      the (swizzled) 16-byte chunk, so the four k-steps of a tap are v_tap ^ (kk << 5) ----
    uint32_t vtap[NT];
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
        const int q = lrow + (tap / TAPS) * p.GW + (tap % TAPS);
        vtap[tap] = (uint32_t)(q * RB + ((lgrp ^ ((q >> 1) & 7)) << 4));
    }

#define RW_STAMP() do { if (tr && tr_n < 32) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    RW_STAMP();                                           // 2: weights requested
    const bf16_t* __restrict__ maskp = (const bf16_t*)p.mask;
    // the slot range (LDS-DMA issued at the top of the kernel) and the bias row have landed; the weight fragments requested above are
    // waited for here as well (vmcnt(0)), so that no wait on them remains inside the tile loop.  Every wave executes exactly one s_barrier.
    __syncthreads();
    RW_STAMP();                                           // 3: slot range + weights landed

#pragma unroll 1
    for (int tile = 0; tile < NTILE; ++tile) {
        // ---- this lane's output pixel of the tile, store offsets, ReluGrad-mask request (all before the MFMAs) ----
        const int P = P0 + tile * 32 + lrow;
        const bool pin = P < p.MP;
        uint32_t g, gx, b, gy;
        p.div_gw.divmod((uint32_t)(pin ? P : 0), g, gx);
        p.div_g.divmod(g, b, gy);
        const int oy = 2 * (int)gy + PH, ox = 2 * (int)gx + PW;
        const bool ok = pin && oy < p.OH && ox < p.OW;
        const uint32_t e0 = ok ? ((b * p.OH + oy) * p.OW + ox) * 32u + 8u * lgrp : 0u;      // element offset of unit 0 (unit 1: + 16)
        PackN<uint32_t, 4> umk[2];
        if constexpr (MASK) {
            umk[0] = *(const PackN<uint32_t, 4>*)(maskp + e0);                               // offset 0 is always readable
            umk[1] = *(const PackN<uint32_t, 4>*)(maskp + e0 + 16);
        }
        // ---- accumulators start at the bias of their channel: register r = channel 8 (r >> 2) + 4 lgrp + (r & 3) ----
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *(const f32x4*)(bias_lds + 8 * q + 4 * lgrp);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[4 * q + t] = b4[t];
        }
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int ta = tap / TAPS, tb = tap % TAPS;
            if (PH + 2 * (H - ta) >= KH || PW + 2 * (H - tb) >= KH) continue;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const freg af = *(const freg*)(lds + (vtap[tap] ^ (uint32_t)(kk << 5)));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[tap][kk]), __builtin_bit_cast(bf16x8, af), acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) vtap[tap] += 32 * RB;                              // next tile: 32 rows further
        RW_STAMP();                                       // 4 + 2 tile: MFMAs issued

        // ---- epilogue: ReLU, bf16, pair the 4-channel groups of the two half-waves, mask, two 16-byte stores ----
        uint32_t w[4][2];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float v[4] = {acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]};
            const PackN<bf16_t, 4> pk = pack4<bf16_t>(v);                                     // hardware round-to-nearest-even
            w[gq][0] = (uint32_t)pk.v[0] | ((uint32_t)pk.v[1] << 16); w[gq][1] = (uint32_t)pk.v[2] | ((uint32_t)pk.v[3] << 16);
            if constexpr (RELU) { w[gq][0] = pk_relu_bf16(w[gq][0]); w[gq][1] = pk_relu_bf16(w[gq][1]); }   // ReLU on the rounded value: same result, 8 instead of 16 ops
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                auto r = __builtin_amdgcn_permlane32_swap(w[2 * x][d], w[2 * x + 1][d], false, false);
                w[2 * x][d] = r[0]; w[2 * x + 1][d] = r[1];
            }
        if constexpr (MASK) {                             // ReluGrad: keep where the mask tensor (the forward activation) is > 0
            const uint32_t ones = 0x00010001u, ffff = 0xffffffffu;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int d = 0; d < 2; ++d) w[gq][d] &= pk_positive_mask(umk[gq >> 1].v[2 * (gq & 1) + d], ones, ffff);
        }
        if (ok) {
            bf16_t* __restrict__ out = (bf16_t*)p.out;
            *(PackN<uint32_t, 4>*)(out + e0) = PackN<uint32_t, 4>{{w[0][0], w[0][1], w[1][0], w[1][1]}};
            *(PackN<uint32_t, 4>*)(out + e0 + 16) = PackN<uint32_t, 4>{{w[2][0], w[2][1], w[3][0], w[3][1]}};
        }
        RW_STAMP();                                       // 5 + 2 tile: epilogue issued
    }
    if (tr && tr_n < 32) tr[tr_n++] = (long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_ID: wave / simd / cu / se / xcc placement
#undef RW_STAMP
}

template <int TAPS, int KH, bool RELU, bool MASK>
__global__ __launch_bounds__(256, TAPS == 2 ? 3 : 2) void rwconv_gather_kernel(const TapParams p) {
    constexpr int RB = 128;
    constexpr int NIA = (RW_MAXSLOT / 8 + 3) / 4;         // A-tile DMA instructions per wave (upper bound)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[RW_MAXSLOT * RB + 128];
    float* const bias_lds = (float*)(lds + RW_MAXSLOT * RB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P0 = xcd_remap(blockIdx.x, gridDim.x) * RW_BMT;
    // debug (mi_debug_set_trace): s_memtime stamps of lane 0 of every wave, 32 per wave, 8 wave slots per block
    long long* tr = nullptr; int tr_n = 0;
    if (p.trace && ((long long)blockIdx.x * 8 + 8) * 32 <= p.trace_cap && lane == 0) tr = p.trace + ((long long)blockIdx.x * 8 + wave) * 32;
    if (tr) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime();       // 0: start
    const int halo = (TAPS - 1) * p.GW + TAPS - 1;
    const int ninstr = (RW_BMT + halo + 7) >> 3;          // 8-slot DMA instructions covering the staged range

    // ---- stage the slot range: instruction t = wave + 4 i fills slots 8 t .. 8 t + 7 (lane: slot 8 t + lane / 8, physical chunk lane % 8) ----
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
    {
        const int r8 = lane >> 3;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int t = wave + 4 * i;
            if (t >= ninstr) break;                       // wave-uniform
            const int q = 8 * t + r8;
            const int P = P0 + q;
            const int c = (lane & 7) ^ ((q >> 1) & 7);    // logical chunk this lane fetches (source-side swizzle)
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(P < p.MP ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            const int iy = (int)gy - p.HY, ix = (int)gx - p.HX;
            const bool in = P < p.MP && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const uint32_t vo = in ? (((b * p.IH + iy) * p.IW + ix) * 64u + c * 8u) * 2u : G2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(lds + t * 1024), 16, (int)vo, 0, 0, 0);
        }
    }
    if (tid < 32) bias_lds[tid] = p.bias ? p.bias[tid] : 0.f;
    if (tr) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime();       // 1: slot-range DMA issued
    // each wave: request its class's weight fragments, ONE block barrier (inside rw_class), then its 8 tiles on its own
    if (wave == 0) rw_class<TAPS, KH, 0, RELU, MASK>(p, lds, bias_lds, P0, lane, tr, tr_n);
    else if (wave == 1) rw_class<TAPS, KH, 1, RELU, MASK>(p, lds, bias_lds, P0, lane, tr, tr_n);
    else if (wave == 2) rw_class<TAPS, KH, 2, RELU, MASK>(p, lds, bias_lds, P0, lane, tr, tr_n);
    else rw_class<TAPS, KH, 3, RELU, MASK>(p, lds, bias_lds, P0, lane, tr, tr_n);
}

}  // namespace mi

using namespace mi;

int g_rwconv_mode = -1;                                  // mi_set_tuning key 13 / MI355_RWCONV: 0 off, 1 auto, 2 whenever the layer is eligible
int mi_rwconv_mode(int set) {                            // set < 0: query
    if (g_rwconv_mode < 0) { const char* e = getenv("MI355_RWCONV"); g_rwconv_mode = e ? atoi(e) : 1; if (g_rwconv_mode < 0 || g_rwconv_mode > 2) g_rwconv_mode = 1; }
    const int prev = g_rwconv_mode;
    if (set >= 0) g_rwconv_mode = set > 2 ? 2 : set;
    return prev;
}

// gather-form transposed conv / conv input gradient on the register-weight kernel.  Same contract as try_tapconv (conv_ops.hip):
// returns 1 launched, 0 not eligible, < 0 error.  x [B,IH,IW,64] bf16, w [KH][KW][32][64] bf16, out / mask [B,OH,OW,32] bf16.
int mi_try_rwconv_gather(hipStream_t st, int dtype, const void* a, const void* w, int B, int IH, int IW, int C, int OH, int OW, int N,
                         int KH, int KW, void* out, const float* bias, const void* mask, int relu) {
    mi_rwconv_mode(-1);
    if (dtype != MI_BF16 || C != 64 || N != 32 || KH != KW || (KH != 4 && KH != 5)) return 0;
    if ((((uintptr_t)a) | ((uintptr_t)w) | ((uintptr_t)out) | ((uintptr_t)mask)) & 15) return 0;
    if ((long long)B * OH * OW * N >= (1ll << 31)) return 0;
    TapParams q = {};
    q.TH = q.TW = (KH + 1) / 2; q.HY = q.HX = q.TH - 1;
    q.GH = (OH + 1) / 2 + q.HY; q.GW = (OW + 1) / 2 + q.HX;
    if ((q.TH - 1) * q.GW + q.TW - 1 > RW_MAXHALO) return 0;
    const long long MP = (long long)B * q.GH * q.GW, a_bytes = (long long)B * IH * IW * C * 2;
    if (MP >= (1ll << 30) || a_bytes <= 0 || a_bytes >= (long long)G2_OOB) return 0;
    q.a = a; q.a_bytes = (uint32_t)a_bytes; q.b = w; q.b_bytes = 0;
    q.B = B; q.IH = IH; q.IW = IW; q.C = C; q.OH = OH; q.OW = OW; q.N = N; q.KH = KH; q.KW = KW; q.MP = (int)MP;
    q.KC = C; q.NE = 4 * N;
    q.div_g = make_fastdiv(q.GH); q.div_gw = make_fastdiv(q.GW);
    q.out = out; q.bias = bias; q.mask = mask; q.relu = relu;
    mi_get_trace(&q.trace, &q.trace_cap);
    const dim3 g((unsigned)((MP + RW_BMT - 1) / RW_BMT));
    if (g_rwconv_mode == 0 || (g_rwconv_mode == 1 && g.x < 300)) return 0;      // auto: only where the grid fills the chip (as tapconv)
#define RW_LAUNCH(TAPS_, KH_) do { \
        if (relu && mask) hipLaunchKernelGGL((rwconv_gather_kernel<TAPS_, KH_, true, true>), g, dim3(256), 0, st, q); \
        else if (relu) hipLaunchKernelGGL((rwconv_gather_kernel<TAPS_, KH_, true, false>), g, dim3(256), 0, st, q); \
        else if (mask) hipLaunchKernelGGL((rwconv_gather_kernel<TAPS_, KH_, false, true>), g, dim3(256), 0, st, q); \
        else hipLaunchKernelGGL((rwconv_gather_kernel<TAPS_, KH_, false, false>), g, dim3(256), 0, st, q); } while (0)
    if (KH == 4) RW_LAUNCH(2, 4); else RW_LAUNCH(3, 5);
#undef RW_LAUNCH
    const int rc = mi_check_launch("rwconv_gather_kernel");
    return rc == MI_OK ? 1 : rc;
}
