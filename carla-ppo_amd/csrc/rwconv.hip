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

#ifndef MI_RW_STORE_AUX                                     // cache policy of the output stores (variant builds: 2 = nt, the streaming hint)
#define MI_RW_STORE_AUX 0
#endif
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

// LDS fragment reads issued and waited for BY HAND: hipcc's scheduler sinks ds_reads next to their MFMA (one read in flight: every MFMA then
// waits a full LDS latency), whatever distance the source puts between them.  asm volatile keeps the issue order; the wait names the fragment
// it releases as an in/out operand, so the MFMAs that consume it cannot be scheduled above the wait.  LDS returns in order: lgkmcnt(N) = all
// but the N most recent reads have landed (an outstanding scalar load can only make the wait stricter).
template <int OFF>
__device__ __forceinline__ void lds_read_b128_pair(u16x8& a, u16x8& b, uint32_t addr) {      // rows q and q + 32 (the next 32-position tile: OFF = 32 rows)
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:%3" : "=&v"(a), "=&v"(b) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lds_wait_pair(u16x8& a, u16x8& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

// A block is PERSISTENT: it keeps its weights in registers and walks chunks of NTILE x 32 positions; the slot range of the next chunk is
// staged (LDS-DMA, second buffer) while the current one is computed.  One block barrier per chunk: "my DMA of this chunk has landed and
// everybody is done with the previous chunk", after which the buffer of the previous chunk is refilled.
// Block shapes.  k = 4 (equal work per class): 4 waves, 128 positions per chunk, 3 blocks per CU.  k = 5: the classes reach 9 / 6 / 6 / 4
// taps, so a 4-wave block leaves the MFMA pipes of three SIMDs waiting for the fourth; there the block has 8 waves = two halves of four, each
// half walks its own 128 positions of a 256-position chunk and the second half takes the classes in REVERSE order: waves w and w + 4 share
// a SIMD (the dispatcher places a workgroup's waves on the SIMDs cyclically), which pairs the classes {0,3} and {1,2}: 13 and 12 tap-units
// per SIMD instead of 9 | 6 | 6 | 4.  (The pairing is a performance assumption only: every wave computes its own class whatever its SIMD.)
// CK = C / 64 = N / 32 (1: the 64 -> 32 channel layers; 2, k = 4 only: 128 -> 64 channels -- deconv2 forward, conv3's input gradient): slot rows of
// 128 CK bytes, 4 CK k-steps per tap, the weights of ONE 32-output tile of a class in registers (64 CK VGPRs for k = 4), 4 CK waves = (class, tile)
template <int TAPS, int CK = 1> struct RwCfg {
    static constexpr int NW = TAPS == 2 ? 4 * CK : 8;                 // waves per block
    // 32-position tiles per wave and chunk.  CK = 2: 64-position chunks -- these layers have 102 k slots, i.e. 800 chunks of 128 on 256 blocks
    // (3.1 each: a quarter of the chip idles through the fourth round); 1,600 chunks of 64 lose a tenth instead, for 14 % more staged halo rows:
    // conv3.dgrad 40.3 -> 35.7 us, deconv2.fwd 38.3 -> 36.4 (same-box A/B of the two builds).  The same change on the 64 -> 32 channel shapes: k = 4
    // (3 blocks per CU, 4.5 chunks each) -2 us in the step's conv2.dgrad but nothing on the step; k = 5 +2 % (its halo is two slot rows).
    static constexpr int NTILE = CK == 2 ? 2 : 4;
    static constexpr int BMT = 32 * NTILE * (TAPS == 2 ? 1 : 2);      // positions per chunk (k = 5: two position halves of four waves)
    static constexpr int MAXHALO = TAPS == 2 ? 48 : 96;               // largest (TAPS-1) * GW + TAPS - 1 the buffers are sized for
    static constexpr int MAXSLOT = BMT + MAXHALO;                     // 176 | 352 rows
    static constexpr int RB = 128 * CK;                               // bytes per slot row
    static constexpr int BUF = MAXSLOT * RB;                          // 22 | 44 | 44 KB per buffer, two buffers: 3 blocks | 1 block | 1 block per CU
    static constexpr int RPI = 1024 / RB;                             // rows per DMA instruction (8 | 4)
    static constexpr int NIA = ((MAXSLOT + RPI - 1) / RPI + NW - 1) / NW;   // DMA instructions per wave and chunk (upper bound)
};

// stage the slot range [P0, P0 + BMT + halo) into `buf`: instruction t = wave + 4 i fills rows 8 t .. 8 t + 7 (lane: row 8 t + lane / 8,
// physical 16-byte chunk lane % 8, fetching the logical chunk (lane % 8) ^ ((row >> 1) & 7): the swizzle of tapconv_tile.hpp)
template <int TAPS, int CK = 1>
__device__ __forceinline__ void rw_stage(const TapParams& p, const __amdgpu_buffer_rsrc_t rsA, unsigned char* buf, int P0, int wave, int lane, int ninstr) {
    typedef RwCfg<TAPS, CK> Cfg;
    constexpr int NW = Cfg::NW;
    if constexpr (CK == 1) {
        const int r8 = lane >> 3;
        // slot -> (frame, gy, gx) once per chunk; the following instructions of this wave are 8 NW slots further each (GW > 32 on this path: at
        // most NW / 4 row wraps per step)
        uint32_t g, gx, b, gy;
        const int Pf = P0 + 8 * wave + r8;
        p.div_gw.divmod((uint32_t)Pf, g, gx);
        p.div_g.divmod(g, b, gy);
#pragma unroll
        for (int i = 0; i < Cfg::NIA; ++i) {
            const int t = wave + NW * i;
            if (t >= ninstr) break;                           // wave-uniform
            const int q = 8 * t + r8;
            const int c = (lane & 7) ^ ((q >> 1) & 7);
            const int iy = (int)gy - p.HY, ix = (int)gx - p.HX;
            const bool in = P0 + q < p.MP && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const uint32_t vo = in ? (((b * p.IH + iy) * p.IW + ix) * 64u + c * 8u) * 2u : G2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(buf + t * 1024), 16, (int)vo, 0, 0, 0);
            gx += 8 * NW;
#pragma unroll
            for (int r = 0; r < NW / 4; ++r)
                if ((int)gx >= p.GW) { gx -= p.GW; ++gy; if ((int)gy >= p.GH) { gy = 0; ++b; } }
        }
    } else {
        // 256-byte rows: instruction t fills rows 4 t .. 4 t + 3 (lane: row 4 t + lane / 16, physical chunk lane % 16, logical chunk ^ (row & 15)).
        // The slot grid of these layers is narrow (GW = 20): a plain slot decode per instruction, 6 per wave and chunk
        const int r4 = lane >> 4, cp = lane & 15;
#pragma unroll
        for (int i = 0; i < Cfg::NIA; ++i) {
            const int t = wave + NW * i;
            if (t >= ninstr) break;                           // wave-uniform
            const int q = 4 * t + r4;
            const int c = cp ^ (q & 15);                      // 4-bit swizzle (see vtap0 in rw_class)
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)min(P0 + q, p.MP - 1), g, gx);
            p.div_g.divmod(g, b, gy);
            const int iy = (int)gy - p.HY, ix = (int)gx - p.HX;
            const bool in = P0 + q < p.MP && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            const uint32_t vo = in ? (((b * p.IH + iy) * p.IW + ix) * (uint32_t)(64 * CK) + c * 8u) * 2u : G2_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(buf + t * 1024), 16, (int)vo, 0, 0, 0);
        }
    }
}

// One wave = one output parity class CLS (ph = CLS >> 1, pw = CLS & 1).  TAPS taps per axis, KH kernel size (compile time: which
// (tap, class) pairs exist), C = 64 input channels (one 128-byte row per slot), N = 32 output channels per class.
template <int TAPS, int KH, int CLS, bool RELU, int MASK, bool BITS, int DBG = 0, int CK = 1>
__device__ __forceinline__ void rw_class(const TapParams& p, unsigned char* lds, const float* bias_lds, const __amdgpu_buffer_rsrc_t rsA,
                                         int chunk, const int cstride, const int cend, int wave, int half, int lane, long long* tr, int tr_n, int nt = 0) {
    typedef RwCfg<TAPS, CK> Cfg;
    constexpr int RB = Cfg::RB, NT = TAPS * TAPS, NTILE = Cfg::NTILE, KS = 4 * CK;      // KS: k-steps of 16 channels per tap
    constexpr uint32_t NCH = 32u * CK;                    // output channels per pixel (nt: this wave's 32-channel tile of them)
    constexpr int PH = CLS >> 1, PW = CLS & 1, H = TAPS - 1;
    typedef u16x8 freg;
    const int lrow = lane & 31, lgrp = lane >> 5;
    const int halo = (TAPS - 1) * p.GW + TAPS - 1;
    const int ninstr = (Cfg::BMT + halo + Cfg::RPI - 1) / Cfg::RPI;

    rw_stage<TAPS, CK>(p, rsA, lds, chunk * Cfg::BMT, wave, lane, ninstr);       // first chunk -> buffer 0
#define RW_STAMP() do { if (tr && tr_n < 32) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    RW_STAMP();                                           // 1: first slot range requested

    // ---- weights: fragment (tap, kk) = 8 consecutive input channels (kk*16 + lgrp*8) of output channel lrow, straight from L2, ONCE ----
    freg wf[NT][KS];
    if (p.bfrag) {                                         // (round 6; k = 5, CK = 1: deconv3 forward) the fragment-ordered copy, 1 KB contiguous per load: [class][tap][kk][lane] (pack form 4)
        const bf16_t* __restrict__ F = (const bf16_t*)p.bfrag;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int ta = tap / TAPS, tb = tap % TAPS;
            if (PH + 2 * (H - ta) >= KH || PW + 2 * (H - tb) >= KH) continue;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) wf[tap][kk] = *(const freg*)(F + ((size_t)((CLS * NT + tap) * KS + kk) * 64 + lane) * 8);
        }
    } else
    {
        const bf16_t* __restrict__ W = (const bf16_t*)p.b;
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int ta = tap / TAPS, tb = tap % TAPS;
            const int kh = PH + 2 * (H - ta), kw = PW + 2 * (H - tb);
            if (kh >= KH || kw >= KH) continue;           // literal after unrolling: this class does not reach that kernel row / column
            const bf16_t* wrow = W + ((long long)(kh * KH + kw) * (int)NCH + nt * 32 + lrow) * (64 * CK) + lgrp * 8;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) wf[tap][kk] = *(const freg*)(wrow + kk * 16);
        }
    }
    RW_STAMP();                                           // 2: weights requested

    // ---- LDS read addresses: row q = lrow + delta_tap (+ 32 per tile, which leaves the swizzle term alone); rows are 128 B, bits 4..6 are
    //      the (swizzled) 16-byte chunk, so the four k-steps of a tap are v_tap ^ (kk << 5) ----
    uint32_t vtap0[NT];
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
        const int q = lrow + (tap / TAPS) * p.GW + (tap % TAPS);
        // 256-byte rows span all 64 banks, so rows q and q + 1 must not keep a chunk in the same place: the lane groups of ds_read_b128 hold 16 rows with
        // 16 different values of q & 15 (MI355X_MICROARCH.md, LDS) -- the 3-bit term of the 128-byte rows left every read 2-way conflicted here
        vtap0[tap] = CK == 1 ? (uint32_t)(q * RB + ((lgrp ^ ((q >> 1) & 7)) << 4)) : (uint32_t)(q * RB + ((lgrp ^ (q & 15)) << 4));
    }
    const bf16_t* __restrict__ maskp = (const bf16_t*)p.mask;

    // output stores go through a buffer descriptor: lanes without a valid pixel get an out-of-range offset (dropped by the hardware) instead
    // of a branch around the store, so every tile issues EXACTLY two store instructions -- which is what lets the chunk barrier wait for the
    // staged slot range with s_waitcnt vmcnt(4 | 6) ("all but the newest VMEM operations", i.e. not for the last stores' acknowledgements:
    // those cost ~2.3k cycles per chunk with the compiler's vmcnt(0) in front of __syncthreads())
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.b_bytes, 0x00020000);    // (b_bytes carries the OUTPUT size in bytes here)
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(BITS ? (void*)p.bits_out : p.out, 0, (int)(p.b_bytes >> 3), 0x00020000);   // bit words: 8 of every 64 bytes

    // the first slot range (requested BEFORE this class's NLIVE weight fragments: "all but the newest NLIVE - 8 vector-memory operations" covers it,
    // returns are in order) and the bias row have landed; after the barrier so has everybody's share.  The first chunk's MFMAs start while the tail
    // of the weights is still on its way (the compiler waits per fragment).
    constexpr int NLIVE = [] { int n = 0; for (int tap = 0; tap < NT; ++tap) if (PH + 2 * (H - tap / TAPS) < KH && PW + 2 * (H - tap % TAPS) < KH) n += KS; return n; }();
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NLIVE < 63 ? NLIVE : 63) - 8) : "memory");
    __builtin_amdgcn_s_barrier();
    int cur = 0;
#pragma unroll 1
    for (; chunk < cend; chunk += cstride, cur ^= 1) {
        RW_STAMP();                                       // 3 + 2 i: chunk i landed, buffers handed over
        if (chunk + cstride < cend) rw_stage<TAPS, CK>(p, rsA, lds + (cur ^ 1) * Cfg::BUF, (chunk + cstride) * Cfg::BMT, wave, lane, ninstr);
        const int P0 = chunk * Cfg::BMT + half * 32 * NTILE;                                  // this wave's half of the chunk
        const uint32_t bufoff = (uint32_t)(cur * Cfg::BUF + half * 32 * NTILE * RB);
        // two 32-position tiles at a time: two independent accumulator chains, and the fragment reads run RW_AHEAD (tap, k-step) pairs ahead
        // of the MFMAs that consume them (hipcc's own schedule kept a single read in flight: every MFMA then waits one LDS latency)
#pragma unroll 1
        for (int tile = 0; tile < NTILE; tile += 2) {
            // ---- this lane's two output pixels, store offsets, ReluGrad-mask requests (all before the MFMAs) ----
            uint32_t e0[2]; bool ok[2];
            PackN<uint32_t, 4> umk[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int P = P0 + (tile + h) * 32 + lrow;
                const bool pin = P < p.MP;
                uint32_t g, gx, b, gy;
                p.div_gw.divmod((uint32_t)(pin ? P : 0), g, gx);
                p.div_g.divmod(g, b, gy);
                const int oy = 2 * (int)gy + PH, ox = 2 * (int)gx + PW;
                ok[h] = pin && oy < p.OH && ox < p.OW;
                e0[h] = ok[h] ? ((b * p.OH + oy) * p.OW + ox) * NCH + (uint32_t)(nt * 32) + 8u * lgrp : 0u;   // element offset of unit 0 (unit 1: + 16)
                if constexpr (MASK == 1) {
                    umk[h][0] = *(const PackN<uint32_t, 4>*)(maskp + e0[h]);                 // offset 0 is always readable
                    umk[h][1] = *(const PackN<uint32_t, 4>*)(maskp + e0[h] + 16);
                } else if constexpr (MASK == 2) {         // ReLU bit words: 8 bytes per pixel instead of 64 (both half-waves read the same two words)
                    const PackN<uint32_t, 2> mw = *(const PackN<uint32_t, 2>*)(p.mask_bits + (e0[h] >> 5) * 2);
                    umk[h][0].v[0] = mw.v[0] >> (4 * lgrp); umk[h][0].v[1] = mw.v[1] >> (4 * lgrp);
                }
            }
            // ---- accumulators start at the bias of their channel: register r = channel 8 (r >> 2) + 4 lgrp + (r & 3) ----
            f32x16 acc[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *(const f32x4*)(bias_lds + nt * 32 + 8 * q + 4 * lgrp);
#pragma unroll
                for (int t = 0; t < 4; ++t) { acc[0][4 * q + t] = b4[t]; acc[1][4 * q + t] = b4[t]; }
            }
            const uint32_t toff = bufoff + (uint32_t)(tile * 32 * RB);                       // wave-uniform
            // live (tap, k-step) steps of this class in issue order (compile time)
            struct Steps { int n; int id[NT * KS]; };
            constexpr Steps ST = [] {
                Steps r = {0, {}};
                for (int tap = 0; tap < NT; ++tap) {
                    const int ta = tap / TAPS, tb = tap % TAPS;
                    if (PH + 2 * (H - ta) >= KH || PW + 2 * (H - tb) >= KH) continue;
                    for (int kk = 0; kk < KS; ++kk) r.id[r.n++] = tap * KS + kk;
                }
                return r;
            }();
            constexpr int AH = 3;                         // steps (pairs of reads) in flight ahead of the MFMAs
            freg ring[AH + 1][2];
            auto fetch = [&](int k) {                     // k-th live step -> ring slot k % (AH + 1)
                const int sid = ST.id[k];
                const int tap_ = sid / KS, kk_ = sid % KS;                                    // k-step kk: 128-byte half kk >> 2, swizzled chunk pair kk & 3
                const uint32_t vt = (vtap0[tap_] + toff) ^ (uint32_t)(kk_ << 5);          // k-step kk: chunk pair kk (CK = 1: kk < 4); toff: whole rows
                if constexpr (DBG == 2) { if (k < AH + 1) lds_read_b128_pair<32 * RB>(ring[k % (AH + 1)][0], ring[k % (AH + 1)][1], vt); }   // debug: no LDS traffic beyond the first reads
                else lds_read_b128_pair<32 * RB>(ring[k % (AH + 1)][0], ring[k % (AH + 1)][1], vt);
            };
#pragma unroll
            for (int k = 0; k < AH && k < ST.n; ++k) fetch(k);
#pragma unroll
            for (int k = 0; k < ST.n; ++k) {
                if (k + AH < ST.n) fetch(k + AH);
                freg (&cur_)[2] = ring[k % (AH + 1)];
                const int after = (ST.n - 1 - k) < AH ? (ST.n - 1 - k) : AH;                 // steps requested after this one
                if (DBG == 2) lds_wait_pair<0>(cur_[0], cur_[1]);
                else if (after == 3) lds_wait_pair<6>(cur_[0], cur_[1]);
                else if (after == 2) lds_wait_pair<4>(cur_[0], cur_[1]);
                else if (after == 1) lds_wait_pair<2>(cur_[0], cur_[1]);
                else lds_wait_pair<0>(cur_[0], cur_[1]);
                const int sid = ST.id[k];
                if constexpr (DBG == 3) {                 // debug: no MFMAs (the fragments are still consumed)
                    acc[0][k & 15] += __builtin_bit_cast(float, (uint32_t)cur_[0][0] | ((uint32_t)wf[sid / KS][sid % KS][0] << 16));
                    acc[1][k & 15] += __builtin_bit_cast(float, (uint32_t)cur_[1][0] | ((uint32_t)wf[sid / KS][sid % KS][1] << 16));
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[sid / KS][sid % KS]), __builtin_bit_cast(bf16x8, cur_[0]), acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[sid / KS][sid % KS]), __builtin_bit_cast(bf16x8, cur_[1]), acc[1], 0, 0, 0);
                }
            }

            // ---- epilogue: bf16 (hardware rounding), ReLU, pair the 4-channel groups of the two half-waves, mask, two 16-byte stores ----
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint32_t w[4][2];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float v[4] = {acc[h][4 * gq], acc[h][4 * gq + 1], acc[h][4 * gq + 2], acc[h][4 * gq + 3]};
                    const PackN<bf16_t, 4> pk = pack4<bf16_t>(v);
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
                if constexpr (MASK == 1) {                // ReluGrad: keep where the mask tensor (the forward activation) is > 0
                    const uint32_t ones = 0x00010001u, ffff = 0xffffffffu;
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                        for (int d = 0; d < 2; ++d) w[gq][d] &= pk_positive_mask(umk[h][gq >> 1].v[2 * (gq & 1) + d], ones, ffff);
                } else if constexpr (MASK == 2) {         // unit x = word x; its dwords are the pairs 4 lgrp + {0,1,2,3}: bits d and 16 + d of the shifted word
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                        for (int d = 0; d < 2; ++d) w[gq][d] &= ((umk[h][0].v[gq >> 1] >> (2 * (gq & 1) + d)) & 0x00010001u) * 0xffffu;
                }
                if constexpr (BITS) {                     // ReLU bit words of the values being stored (post-ReLU: never negative)
                    uint32_t pw_[2] = {0u, 0u};
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                        for (int d = 0; d < 2; ++d) {
                            uint32_t nz;
                            asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(w[gq][d]), "v"(0x00010001u));
                            pw_[gq >> 1] |= nz << (4 * lgrp + 2 * (gq & 1) + d);
                        }
                    auto r = __builtin_amdgcn_permlane32_swap(pw_[0], pw_[1], false, false);  // lower half: both parts of word 0, upper half: of word 1
                    const uint32_t word = r[0] | r[1];
                    const bool st_ok = ok[h];
                    const uint32_t bo = st_ok ? ((e0[h] >> 5) * 2u + (uint32_t)lgrp) * 4u : G2_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(word, rsB, (int)bo, 0, 0);
                }
                {
                    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
                    const bool st_ok = ok[h] && (DBG != 1 || w[0][0] == 0x12345678u);       // debug 1: (practically) nothing stored
                    const uint32_t bo = st_ok ? e0[h] * 2u : G2_OOB;                         // byte offset; out of range = dropped
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{w[0][0], w[0][1], w[1][0], w[1][1]}, rsO, (int)bo, 0, MI_RW_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{w[2][0], w[2][1], w[3][0], w[3][1]}, rsO, (int)bo, 32, MI_RW_STORE_AUX);
                }
            }
        }
        RW_STAMP();                                       // 4 + 2 i: chunk i computed
        // next chunk's slot range (requested at the top of this iteration, before 8 | 16 newer VMEM operations) has landed: everything but
        // the last tile pair's 4 stores is complete; after the barrier every wave is also done reading this chunk's buffer
        if constexpr (BITS) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                  // (+ one bit-word store per tile)
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef RW_STAMP
}

template <int TAPS, int KH, bool RELU, int MASK, bool BITS, int DBG = 0, int CK = 1>
__global__ __launch_bounds__((RwCfg<TAPS, CK>::NW * 64), ((TAPS == 2 && CK == 1) ? 3 : 2)) void rwconv_gather_kernel(const TapParams p, const int nchunks) {
    static_assert(CK == 1 || TAPS == 2, "128-channel rows: k = 4 only");
    typedef RwCfg<TAPS, CK> Cfg;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * Cfg::BUF + 128 * CK];
    float* const bias_lds = (float*)(lds + 2 * Cfg::BUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // k = 5: two position halves of four class waves (second half in reverse class order); k = 4, CK = 2: wave = (class, 32-output tile), one position range
    const int half = CK == 1 ? wave >> 2 : 0, nt = CK == 1 ? 0 : wave >> 2, cls = CK == 1 ? (half ? 3 - (wave & 3) : (wave & 3)) : (wave & 3);
    // chunk schedule: XCD x (= blockIdx % 8: the dispatcher's round-robin) owns a contiguous range of chunks; at any time its blocks work on
    // neighbouring chunks (shared halo rows stay in that XCD's L2)
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;                 // the host launches a multiple of 8 blocks
    const int q8 = nchunks >> 3, r8 = nchunks & 7;
    const int cbeg = xcd * q8 + (xcd < r8 ? xcd : r8), cend = cbeg + q8 + (xcd < r8 ? 1 : 0);
    // debug (mi_debug_set_trace): s_memtime stamps of lane 0 of every wave, 32 per wave, 8 wave slots per block
    long long* tr = nullptr; int tr_n = 0;
    if (p.trace && ((long long)blockIdx.x * 8 + 8) * 32 <= p.trace_cap && lane == 0) tr = p.trace + ((long long)blockIdx.x * 8 + wave) * 32;
    if (tr) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime();       // 0: start
    if (tid < 32 * CK) bias_lds[tid] = p.bias ? p.bias[tid] : 0.f;
    if (cbeg + j >= cend) return;                         // block-uniform: more blocks than chunks on this XCD
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
    // each wave: stage its share of the first chunk, request its class's weight fragments, then the chunk loop (one block barrier per chunk)
    if (cls == 0) rw_class<TAPS, KH, 0, RELU, MASK, BITS, DBG, CK>(p, lds, bias_lds, rsA, cbeg + j, per, cend, wave, half, lane, tr, tr_n, nt);
    else if (cls == 1) rw_class<TAPS, KH, 1, RELU, MASK, BITS, DBG, CK>(p, lds, bias_lds, rsA, cbeg + j, per, cend, wave, half, lane, tr, tr_n, nt);
    else if (cls == 2) rw_class<TAPS, KH, 2, RELU, MASK, BITS, DBG, CK>(p, lds, bias_lds, rsA, cbeg + j, per, cend, wave, half, lane, tr, tr_n, nt);
    else rw_class<TAPS, KH, 3, RELU, MASK, BITS, DBG, CK>(p, lds, bias_lds, rsA, cbeg + j, per, cend, wave, half, lane, tr, tr_n, nt);
    if (tr) tr[31] = (long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID of this wave (wave / SIMD / CU placement)
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// CONV form (tapconv_tile.hpp): out[b,oy,ox,n] = sum_{ta,tb} sum_{ph,pw,c} x[b,2(oy+ta)+ph,2(ox+tb)+pw,c] W[n][(2ta+ph) KH + 2tb+pw][c], 32 -> 64 channels
// (deconv3's input gradient, k = 5; conv2 forward, k = 4; vae/models.py:236-239, 250-253).  A slot is the 2 x 2 pixel block: 4 x 32 channels =
// one 256-byte row, i.e. the addressing of the CK = 2 rows above with 8 k-steps per tap (k-step s = 4 ph + 2 pw + kk), of which k = 5 leaves
// 50 of 72 alive.  There are no parity classes to hand to the waves, so a wave owns one 32-OUTPUT tile (nt = wave & 1) for ALL of K -- 50 | 32
// weight fragments = 200 | 128 registers, the reason for one wave per SIMD and one block per CU -- and one 64-position half of the 128-position
// chunk (wave >> 1).  Every fragment read serves two MFMAs (the two 32-position tiles of the half).  im2col-gather tiles (gemm2) move every
// input pixel through the per-CU load path 6.25 times for this layer (562 MB of L2 -> LDS traffic for a 101 MB tensor) and re-stage the 102 KB of
// weights per 128 x 64 tile; here a pixel is staged 1.6 times (halo) and the weights are read once per CU.
template <int KH, int CK = 1> struct RcCfg {
    static constexpr int TAPS = (KH + 1) / 2;
    // CK = 1: 32 -> 64 channels, waves = 2 output tiles x 2 position halves of a 128-position chunk.  CK = 2 (k = 4 only): 64 -> 128 channels (conv3
    // forward, deconv2's input gradient): 512-byte slot rows, 16 k-steps per tap, 64 weight fragments = 256 registers per wave, waves = the 4 output
    // tiles of ONE 64-position range (every fragment read then serves one of four waves' MFMA pairs: 128 B/clk of the LDS's 256)
    static constexpr int NW = 4, NTILE = 2, BMT = CK == 1 ? 128 : 64;
    static constexpr int MAXHALO = TAPS == 2 ? (CK == 1 ? 48 : 32) : 96;     // staged halo: (TAPS-1) * GW + TAPS - 1 rounded up to 16 rows
    // A block walks CONSECUTIVE chunks in runs of RUN: the slot rows of a run lie in one linear buffer (row r = slot P_run + r; chunk j of the run reads
    // rows BMT j .. BMT j + BMT + halo), filled in RUN instalments: rows [0, BMT + H) before the run's first chunk, rows BMT j + H .. BMT (j + 1) + H
    // during chunk j - 1.  The halo of a chunk is the body of the next one, so a run stages BMT RUN + H rows instead of RUN (BMT + H): 10 instead of
    // 14 DMA pieces per wave and chunk for k = 5 (a piece costs a lone wave ~180 cycles of issue, see the kernel).  While the last chunk of a run reads
    // the rows of its own range, the first instalment of the next run goes to rows [0, BMT + H): no second buffer.
    static constexpr int RUN = 3;
    static constexpr int ROWS = RUN * BMT + MAXHALO;                  // 432 | 480 rows (CK = 2: 224)
    static constexpr int RB = 256 * CK;
    static constexpr int RPP = 1024 / RB;                             // rows per DMA piece: 4 | 2
    static constexpr int LDSB = ROWS * RB;                            // 108 | 120 KB (CK = 2: 112 KB): one block per CU (its waves fill the register file anyway)
    static constexpr int NIA = ((BMT + MAXHALO) / RPP + NW - 1) / NW; // DMA pieces per wave of a run's first instalment (11 | 14 | 12); the others: 8
};

// An instalment of slots: piece t = wave + 4 i fills rows 4 t .. 4 t + 3 of it (lane: row 4 t + lane / 16, physical chunk lane % 16).
// Logical chunk c = 8 ph + 4 pw + (16-byte quarter of the pixel's 32 channels): pixel (2 gy + ph, 2 gx + pw).  A wave's instructions are 16
// slots apart, so row & 15 -- the swizzle term: physical chunk = c ^ (row & 15) -- and with it c are the same for all of them, and the slot walk is
// incremental (GW > 16):
// one decode per chunk, then ~10 VALU instructions per DMA instruction.
template <int CK> struct RcWalk {
    int P, sy, sx;                   // slot number, input row 2 gy + ph, input column 2 gx of this lane's next row
    uint32_t srow;                   // (b IH + sy) IW
    __device__ __forceinline__ void start(const TapParams& p, int Pq, int ph) {
        uint32_t g, gx, b, gy;
        p.div_gw.divmod((uint32_t)Pq, g, gx);
        p.div_g.divmod(g, b, gy);
        P = Pq; sy = 2 * (int)gy + ph; sx = 2 * (int)gx; srow = (b * p.IH + (uint32_t)sy) * p.IW;
    }
    __device__ __forceinline__ uint32_t offset(const TapParams& p, int pw, uint32_t cofs) const {
        const bool in = P < p.MP && sy < p.IH && sx + pw < p.IW;
        return in ? (srow + (uint32_t)sx) * (64u * CK) + cofs : G2_OOB;
    }
    __device__ __forceinline__ void advance(const TapParams& p) {                 // to this wave's next piece: 16 | 8 slots on (GW is larger)
        constexpr int STEP = 4 * (4 / CK);
        P += STEP; sx += 2 * STEP;
        if (sx >= 2 * p.GW) {
            sx -= 2 * p.GW; sy += 2; srow += 2 * p.IW;
            if (sy >= 2 * p.GH) { sy -= 2 * p.GH; srow += (uint32_t)((p.IH - 2 * p.GH) * p.IW); }
        }
    }
};

template <int KH, int CK, bool RELU, bool MASK, bool WFRAG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rwconv_conv_kernel(const TapParams p, const int nchunks) {
    static_assert(CK == 1 || (CK == 2 && KH == 4), "64 -> 128 channels: k = 4 only");
    typedef RcCfg<KH, CK> Cfg;
    constexpr int TAPS = Cfg::TAPS, NT = TAPS * TAPS, RB = Cfg::RB, RPP = Cfg::RPP, SPT = 8 * CK;      // SPT: k-steps per tap
    typedef u16x8 freg;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[Cfg::LDSB + 256 * CK];
    float* const bias_lds = (float*)(lds + Cfg::LDSB);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = CK == 1 ? (wave & 1) : wave, half = CK == 1 ? (wave >> 1) : 0;
    const int lrow = lane & 31, lgrp = lane >> 5;
    // block b owns the chunks [c0, c1): consecutive, so that a chunk's halo rows are the next chunk's body (RcCfg)
    const int nbk = (int)gridDim.x, bk = (int)blockIdx.x;
    const int qn = nchunks / nbk, rn = nchunks % nbk;
    const int c0 = bk * qn + (bk < rn ? bk : rn), c1 = c0 + qn + (bk < rn ? 1 : 0);
    // debug (mi_debug_set_trace): s_memtime stamps of lane 0 of every wave: 0 start, 1 first slot range requested, 2 weights requested, then per
    // chunk: handed over / pixel decode + mask requests done / MFMA loop done / epilogue done (tools/trace_rwconv.py deconv3.dgrad)
    long long* tr = nullptr; int tr_n = 0;
    if (p.trace && ((long long)blockIdx.x * 8 + 8) * 32 <= p.trace_cap && lane == 0) tr = p.trace + ((long long)blockIdx.x * 8 + wave) * 32;
#define RC_STAMP() do { if (tr && tr_n < 31) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    RC_STAMP();
    if (tid < 64 * CK) bias_lds[tid] = p.bias ? p.bias[tid] : 0.f;
    int chunk = c0;
    if (chunk >= c1) return;                                 // block-uniform
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, (int)p.a_bytes, 0x00020000);
    const int halo = (TAPS - 1) * p.GW + TAPS - 1;
    const int hs = (halo + 15) & ~15;                        // staged halo rows: instalments start on multiples of 16 rows (the swizzle term of a DMA lane is fixed)
    const int np0 = (Cfg::BMT + hs) / RPP;                   // pieces of a run's first instalment
    // DMA lane role (see RcWalk): row r of the piece, physical chunk cp; logical chunk = cp ^ (row & 15) on its low 4 bits.  A wave's pieces are 16 rows
    // apart (CK = 1: one role) or 8 (CK = 2: the role of odd pieces has the other row & 15, i.e. the other pixel of the pair)
    constexpr int LPR = RB / 16;                             // lanes per row
    const int sq0 = RPP * wave + lane / LPR;
    const int scp = lane % LPR;
    int spw[2]; uint32_t scofs[2];
    int sph = 0;
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int sc = (scp & ~15) | ((scp & 15) ^ ((sq0 + par * 4 * RPP) & 15));
        sph = sc >> (2 + CK); spw[par] = (sc >> (1 + CK)) & 1; scofs[par] = (uint32_t)(sc & (8 * CK - 1)) * 16u;
    }
    RcWalk<CK> walk;
    walk.start(p, chunk * Cfg::BMT + sq0, sph);
#pragma unroll
    for (int i = 0; i < Cfg::NIA; ++i) {
        if (wave + Cfg::NW * i >= np0) break;                // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(lds + (wave + Cfg::NW * i) * 1024), 16, (int)walk.offset(p, spw[i & 1], scofs[i & 1]), 0, 0, 0);
        walk.advance(p);
    }
    RC_STAMP();

    // live (tap, k-step) pairs in issue order (compile time): k-step s = 4 ph + 2 pw + kk reaches kernel row 2 ta + ph, column 2 tb + pw
    struct Steps { int n; int id[NT * SPT]; };
    constexpr Steps ST = [] {
        Steps r = {0, {}};
        for (int tap = 0; tap < NT; ++tap)
            for (int s = 0; s < SPT; ++s)
                if (2 * (tap / TAPS) + s / (4 * CK) < KH && 2 * (tap % TAPS) + (s / (2 * CK)) % 2 < KH) r.id[r.n++] = tap * SPT + s;
        return r;
    }();
    static_assert(ST.n == KH * KH * 2 * CK, "live steps");

    // ---- weights: step k -> 8 consecutive input channels (16 kk + 8 lgrp) of kernel position (kh, kw), output channel 32 nt + lrow; ONCE per block ----
    freg wf[ST.n];
    {
        const bf16_t* __restrict__ wrow = (const bf16_t*)p.b + (long long)(nt * 32 + lrow) * p.ldb + lgrp * 8;
#pragma unroll
        for (int k = 0; k < ST.n; ++k) {
            const int tap = ST.id[k] / SPT, s = ST.id[k] % SPT;
            const int kh = 2 * (tap / TAPS) + s / (4 * CK), kw = 2 * (tap % TAPS) + (s / (2 * CK)) % 2;
            // WFRAG (round 6): the fragment-ordered copy -- fragment (nt, k) is 1 KB contiguous, lane-linear -- instead of 32 rows x 32 B at a 2 KB pitch per load:
            // conv3 forward 35.5 -> 30.9 us, deconv2's input gradient 35.6 -> 32.7 us, step -1.5 % (timing build with coalesced but wrong addresses, three interleaved rounds)
            if constexpr (WFRAG) wf[k] = *(const freg*)((const bf16_t*)p.bfrag + ((size_t)(nt * ST.n + k) * 64 + lane) * 8);
            else wf[k] = *(const freg*)(wrow + (kh * KH + kw) * (32 * CK) + (s % (2 * CK)) * 16);
        }
    }
    RC_STAMP();
    uint32_t vtap0[NT];
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
        const int q = lrow + (tap / TAPS) * p.GW + (tap % TAPS);
        vtap0[tap] = (uint32_t)(q * RB + ((lgrp ^ (q & 15)) << 4));          // k-step s of the tap: ^ (s << 5)
    }
    const bf16_t* __restrict__ maskp = (const bf16_t*)p.mask;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.b_bytes, 0x00020000);    // (b_bytes carries the OUTPUT size in bytes)

    // the first instalment was requested BEFORE the ST.n weight fragments: "all but the newest ST.n - 8 vector-memory operations" covers it (in-order
    // returns), and the MFMA loop of the first chunk starts while the tail of the weights is still on its way (the compiler waits per fragment)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((ST.n < 63 ? ST.n : 63) - 8) : "memory");
    __builtin_amdgcn_s_barrier();
    int jrun = 0;                                            // position of `chunk` in its run
#pragma unroll 1
    for (; chunk < c1; ++chunk) {
        RC_STAMP();
        const bool more = chunk + 1 < c1;                    // block-uniform
        const int jn = jrun == Cfg::RUN - 1 ? 0 : jrun + 1;
        if (more) {
            // the next chunk's instalment.  A wave's pieces take ~180 cycles EACH to issue (22 B/clk per CU with four waves at it: tools/trace_rwconv.py
            // deconv3.dgrad), here as well as spread between the MFMA steps (where the MFMA loop grew from 3.7k to 6.1k cycles), as LDS-DMA as well as
            // through registers + ds_write: the CU's load path, not the instruction form.  With one wave per SIMD nothing covers it.
            const int fs = (chunk + 1) * Cfg::BMT + (jn ? hs : 0), drow = jn ? jn * Cfg::BMT + hs : 0, np = jn ? Cfg::BMT / RPP : np0;
            walk.start(p, fs + sq0, sph);
#pragma unroll
            for (int i = 0; i < Cfg::NIA; ++i) {
                if (wave + Cfg::NW * i >= np) break;             // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_vptr)(lds + drow * RB + (wave + Cfg::NW * i) * 1024), 16, (int)walk.offset(p, spw[i & 1], scofs[i & 1]), 0, 0, 0);
                walk.advance(p);
            }
        }
        const int P0 = chunk * Cfg::BMT + half * 64;
        const uint32_t toff = (uint32_t)((jrun * Cfg::BMT + half * 64) * RB);
        // ---- this lane's two output pixels (conv form: slot (gy, gx) IS output pixel (gy, gx) when inside the output) ----
        uint32_t e0[2]; bool ok[2];
        PackN<uint32_t, 4> umk[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int P = P0 + h * 32 + lrow;
            const bool pin = P < p.MP;
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(pin ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            ok[h] = pin && (int)gy < p.OH && (int)gx < p.OW;
            e0[h] = ok[h] ? ((b * p.OH + gy) * p.OW + gx) * (64u * CK) + (uint32_t)(nt * 32) + 8u * lgrp : 0u;
            if constexpr (MASK) {
                umk[h][0] = *(const PackN<uint32_t, 4>*)(maskp + e0[h]);
                umk[h][1] = *(const PackN<uint32_t, 4>*)(maskp + e0[h] + 16);
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 b4 = *(const f32x4*)(bias_lds + nt * 32 + 8 * q + 4 * lgrp);
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc[0][4 * q + t] = b4[t]; acc[1][4 * q + t] = b4[t]; }
        }
        RC_STAMP();
        constexpr int AH = 3;                             // steps of fragment reads in flight (6 measured the same: the loop runs at 37 cycles per MFMA)
        freg ring[AH + 1][2];
        auto fetch = [&](int k) {
            const int tap_ = ST.id[k] / SPT, s_ = ST.id[k] % SPT;
            const uint32_t vt = (vtap0[tap_] + toff) ^ (uint32_t)(s_ << 5);          // (toff: whole rows, a multiple of 256)
            lds_read_b128_pair<32 * RB>(ring[k % (AH + 1)][0], ring[k % (AH + 1)][1], vt);
        };
#pragma unroll
        for (int k = 0; k < AH; ++k) fetch(k);
#pragma unroll
        for (int k = 0; k < ST.n; ++k) {
            if (k + AH < ST.n) fetch(k + AH);
            freg (&cur_)[2] = ring[k % (AH + 1)];
            const int after = (ST.n - 1 - k) < AH ? (ST.n - 1 - k) : AH;
            if (after == 3) lds_wait_pair<6>(cur_[0], cur_[1]);
            else if (after == 2) lds_wait_pair<4>(cur_[0], cur_[1]);
            else if (after == 1) lds_wait_pair<2>(cur_[0], cur_[1]);
            else lds_wait_pair<0>(cur_[0], cur_[1]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[k]), __builtin_bit_cast(bf16x8, cur_[0]), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[k]), __builtin_bit_cast(bf16x8, cur_[1]), acc[1], 0, 0, 0);
        }
        RC_STAMP();
        // ---- epilogue (rw_class's): bf16, ReLU, pair the 4-channel groups of the two half-waves, mask, two 16-byte stores per tile.  (Running its tail --
        //      pairing, mask, stores -- one chunk late between the next chunk's MFMA steps was measured: the stamps showed the epilogue gone and the loop
        //      +630 cycles; un-instrumented the kernel was 4 % SLOWER.  Stores do not stall a wave; instructions between MFMAs do.) ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint32_t w[4][2];
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float v[4] = {acc[h][4 * gq], acc[h][4 * gq + 1], acc[h][4 * gq + 2], acc[h][4 * gq + 3]};
                const PackN<bf16_t, 4> pk = pack4<bf16_t>(v);
                w[gq][0] = (uint32_t)pk.v[0] | ((uint32_t)pk.v[1] << 16); w[gq][1] = (uint32_t)pk.v[2] | ((uint32_t)pk.v[3] << 16);
                if constexpr (RELU) { w[gq][0] = pk_relu_bf16(w[gq][0]); w[gq][1] = pk_relu_bf16(w[gq][1]); }
            }
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    auto r = __builtin_amdgcn_permlane32_swap(w[2 * x][d], w[2 * x + 1][d], false, false);
                    w[2 * x][d] = r[0]; w[2 * x + 1][d] = r[1];
                }
            if constexpr (MASK) {
                const uint32_t ones = 0x00010001u, ffff = 0xffffffffu;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                    for (int d = 0; d < 2; ++d) w[gq][d] &= pk_positive_mask(umk[h][gq >> 1].v[2 * (gq & 1) + d], ones, ffff);
            }
            typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
            const uint32_t bo = ok[h] ? e0[h] * 2u : G2_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{w[0][0], w[0][1], w[1][0], w[1][1]}, rsO, (int)bo, 0, MI_RW_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4_t{w[2][0], w[2][1], w[3][0], w[3][1]}, rsO, (int)bo, 32, MI_RW_STORE_AUX);
        }
        RC_STAMP();
        // the next chunk's instalment was requested before this chunk's mask loads (consumed above) and 4 stores: everything but the stores has landed
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        jrun = jn;
    }
#undef RC_STAMP
}

}  // namespace mi

using namespace mi;

int g_rwconv_dbg = 0;
int g_rwconv_blocks = 0;                                 // debug (MI355_RWCONV_BLOCKS): persistent blocks per XCD, 0 = as many as stay resident
int g_rwconv_mode = -1;                                  // mi_set_tuning key 13 / MI355_RWCONV: 0 off, 1 auto, 2 whenever the layer is eligible
int mi_rwconv_mode(int set) {                            // set < 0: query
    if (g_rwconv_mode < 0) { const char* b = getenv("MI355_RWCONV_BLOCKS"); g_rwconv_blocks = b ? atoi(b) : 0; const char* d = getenv("MI355_RWCONV_DBG"); g_rwconv_dbg = d ? atoi(d) : 0; }
    if (g_rwconv_mode < 0) { const char* e = getenv("MI355_RWCONV"); g_rwconv_mode = e ? atoi(e) : 1; if (g_rwconv_mode < 0 || g_rwconv_mode > 2) g_rwconv_mode = 1; }
    const int prev = g_rwconv_mode;
    if (set >= 0) g_rwconv_mode = set > 2 ? 2 : set;
    return prev;
}

// gather-form transposed conv / conv input gradient on the register-weight kernel.  Same contract as try_tapconv (conv_ops.hip):
// returns 1 launched, 0 not eligible, < 0 error.  x [B,IH,IW,64] bf16, w [KH][KW][32][64] bf16, out / mask [B,OH,OW,32] bf16.
int mi_try_rwconv_gather(hipStream_t st, int dtype, const void* a, const void* w, int B, int IH, int IW, int C, int OH, int OW, int N,
                         int KH, int KW, void* out, const float* bias, const void* mask, int relu, const void* mask_bits, void* bits_out) {
    const void* const wfrag = mi_tl_rc_wfrag;              // fragment-ordered weights announced for this launch (mi_rwconv_next_weights_fragment_ordered; pack form 4: k = 5, 64 -> 32 channels)
    mi_tl_rc_wfrag = nullptr;
    mi_rwconv_mode(-1);
    static int wide = -1;                                 // MI355_RWCONV_WIDE=0: the 128 -> 64 channel layers stay on tapconv (A/B runs)
    if (wide < 0) { const char* ev = getenv("MI355_RWCONV_WIDE"); wide = (ev && ev[0] == '0') ? 0 : 1; }
    const int ck = (C == 128 && N == 64 && KH == 4 && wide) ? 2 : 1;                      // 128 -> 64 channels: k = 4 only
    if (dtype != MI_BF16 || C != 64 * ck || N != 32 * ck || KH != KW || (KH != 4 && KH != 5)) return 0;
    if ((((uintptr_t)a) | ((uintptr_t)w) | ((uintptr_t)out) | ((uintptr_t)mask)) & 15) return 0;
    if ((long long)B * OH * OW * N >= (1ll << 31)) return 0;
    TapParams q = {};
    q.TH = q.TW = (KH + 1) / 2; q.HY = q.HX = q.TH - 1;
    q.GH = (OH + 1) / 2 + q.HY; q.GW = (OW + 1) / 2 + q.HX;
    const int halo = (q.TH - 1) * q.GW + q.TW - 1;
    if (halo > (q.TH == 2 ? RwCfg<2>::MAXHALO : RwCfg<3>::MAXHALO) || (ck == 1 && q.GW <= 32)) return 0;     // (GW > 32: the incremental slot decode of the 128-byte-row staging loop)
    const long long MP = (long long)B * q.GH * q.GW, a_bytes = (long long)B * IH * IW * C * 2;
    if (MP >= (1ll << 30) || a_bytes <= 0 || a_bytes >= (long long)G2_OOB) return 0;
    const long long o_bytes = (long long)B * OH * OW * N * 2;
    if (o_bytes >= (long long)G2_OOB) return 0;
    q.a = a; q.a_bytes = (uint32_t)a_bytes; q.b = w; q.b_bytes = (uint32_t)o_bytes;      // b_bytes: size of the OUTPUT tensor (its buffer descriptor)
    q.B = B; q.IH = IH; q.IW = IW; q.C = C; q.OH = OH; q.OW = OW; q.N = N; q.KH = KH; q.KW = KW; q.MP = (int)MP;
    q.KC = C; q.NE = 4 * N;
    q.div_g = make_fastdiv(q.GH); q.div_gw = make_fastdiv(q.GW);
    q.bfrag = (KH == 5 && ck == 1) ? wfrag : nullptr;
    q.out = out; q.bias = bias; q.mask = mask; q.relu = relu;
    q.mask_bits = mask ? (const uint32_t*)mask_bits : nullptr; q.bits_out = relu ? (uint32_t*)bits_out : nullptr;
    if ((bits_out && !q.bits_out) || ((((uintptr_t)mask_bits) | ((uintptr_t)bits_out)) & 7)) return 0;
    mi_get_trace(&q.trace, &q.trace_cap);
    const int bmt = KH == 4 ? (ck == 2 ? RwCfg<2, 2>::BMT : RwCfg<2>::BMT) : RwCfg<3>::BMT;
    const int nchunks = (int)((MP + bmt - 1) / bmt);
    if (g_rwconv_mode == 0 || (g_rwconv_mode == 1 && MP < 75000)) return 0;      // auto: only where the grid fills the chip (as tapconv)
    // persistent grid: as many blocks as stay resident (3 per CU for k = 4, 2 for k = 5), a multiple of 8 (one share per XCD)
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; hipDeviceProp_t pr; n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
    int per_xcd = (n_cu / 8) * ((KH == 4 && ck == 1) ? 3 : 1);
    if (g_rwconv_blocks > 0) per_xcd = g_rwconv_blocks;
    if (per_xcd > (nchunks + 7) / 8) per_xcd = (nchunks + 7) / 8;
    const dim3 g((unsigned)(8 * per_xcd));
#define RW_LAUNCH(TAPS_, KH_) do { \
        const dim3 t_(TAPS_ == 2 ? 256 : 512); \
        if (relu && q.bits_out && !mask) MI_LAUNCH((rwconv_gather_kernel<TAPS_, KH_, true, 0, true>), g, t_, 0, st, q, nchunks); \
        else if (bits_out) return 0; \
        else if (relu && mask) MI_LAUNCH((rwconv_gather_kernel<TAPS_, KH_, true, 1, false>), g, t_, 0, st, q, nchunks); \
        else if (relu) MI_LAUNCH((rwconv_gather_kernel<TAPS_, KH_, true, 0, false>), g, t_, 0, st, q, nchunks); \
        else if (q.mask_bits) MI_LAUNCH((rwconv_gather_kernel<TAPS_, KH_, false, 2, false>), g, t_, 0, st, q, nchunks); \
        else if (mask) MI_LAUNCH((rwconv_gather_kernel<TAPS_, KH_, false, 1, false>), g, t_, 0, st, q, nchunks); \
        else MI_LAUNCH((rwconv_gather_kernel<TAPS_, KH_, false, 0, false>), g, t_, 0, st, q, nchunks); } while (0)
    if (KH == 5 && relu && !mask && !bits_out && g_rwconv_dbg > 0) {  // debug variants of the deconv3-forward instantiation (MI355_RWCONV_DBG: 1 no stores, 2 no LDS reads, 3 no MFMAs)
        if (g_rwconv_dbg == 1) MI_LAUNCH((rwconv_gather_kernel<3, 5, true, 0, false, 1>), g, dim3(512), 0, st, q, nchunks);
        else if (g_rwconv_dbg == 2) MI_LAUNCH((rwconv_gather_kernel<3, 5, true, 0, false, 2>), g, dim3(512), 0, st, q, nchunks);
        else MI_LAUNCH((rwconv_gather_kernel<3, 5, true, 0, false, 3>), g, dim3(512), 0, st, q, nchunks);
    } else if (KH == 4 && ck == 2) {
        const dim3 t_(512);
        if (relu && q.bits_out && !mask) MI_LAUNCH((rwconv_gather_kernel<2, 4, true, 0, true, 0, 2>), g, t_, 0, st, q, nchunks);
        else if (bits_out) return 0;
        else if (relu && !mask) MI_LAUNCH((rwconv_gather_kernel<2, 4, true, 0, false, 0, 2>), g, t_, 0, st, q, nchunks);
        else if (!relu && q.mask_bits) MI_LAUNCH((rwconv_gather_kernel<2, 4, false, 2, false, 0, 2>), g, t_, 0, st, q, nchunks);
        else if (!relu && mask) MI_LAUNCH((rwconv_gather_kernel<2, 4, false, 1, false, 0, 2>), g, t_, 0, st, q, nchunks);
        else if (!relu) MI_LAUNCH((rwconv_gather_kernel<2, 4, false, 0, false, 0, 2>), g, t_, 0, st, q, nchunks);
        else return 0;
    } else if (KH == 4) RW_LAUNCH(2, 4); else RW_LAUNCH(3, 5);
#undef RW_LAUNCH
    const int rc = mi_check_launch("rwconv_gather_kernel");
    return rc == MI_OK ? 1 : rc;
}

int mi_rwconv_blocks(int set) {                          // mi_set_tuning key 16: persistent blocks per XCD of the register-weight kernels, 0 = one (k = 4 gather: three) per CU
    mi_rwconv_mode(-1);
    const int prev = g_rwconv_blocks;
    if (set >= 0) g_rwconv_blocks = set;
    return prev;
}
// The NEXT register-weight launch issued by this thread reads its weights from `wf`, the same kernel in the fragment order its prologue loads registers in (mi_ares_pack_weights forms
// 3 / 4 / 5): conv form 64 -> 128 channels k = 4 (mi_conv2d_nhwc_fwd[_bits] = conv3 forward, mi_deconv2d_nhwc_dgrad[_bits] = deconv2's input gradient; form 3), gather form
// 64 -> 32 channels k = 5 (mi_deconv2d_nhwc_fwd[_bits] = deconv3 forward; form 4), the fused encoder head (mi_conv2d_enc12_fwd: conv2's kernel, form 5).  Consumed by that call
// whether or not such a kernel takes the layer; NULL clears.  The VAE engine sets it in front of those four launches (round 6).
thread_local const void* mi_tl_rc_wfrag = nullptr;
extern "C" int mi_rwconv_next_weights_fragment_ordered(const void* wf) {
    if (((uintptr_t)wf) & 15) return mi_fail(MI_ERR_ARG, "mi_rwconv_next_weights_fragment_ordered: the copy must be 16-byte aligned");
    mi_tl_rc_wfrag = wf;
    return MI_OK;
}
int g_rwconv_conv = -1;                                  // mi_set_tuning key 15 / MI355_RWCONV_CONV: 0 off, 1 k = 5 only, 2 also k = 4, 3 also the 64 -> 128 channel shape (default)
int mi_rwconv_conv_mode(int set) {                       // set < 0: query
    if (g_rwconv_conv < 0) { const char* e = getenv("MI355_RWCONV_CONV"); g_rwconv_conv = e ? atoi(e) : 3; if (g_rwconv_conv < 0 || g_rwconv_conv > 3) g_rwconv_conv = 3; }
    const int prev = g_rwconv_conv;
    if (set >= 0) g_rwconv_conv = set > 3 ? 3 : set;
    return prev;
}

// conv-form layer on the register-weight kernel: x [B,IH,IW,C] bf16, w [N][ldb] bf16 (K-contiguous, k = (kh KW + kw) C + c), out / mask [B,OH,OW,N]
// bf16, (C, N) = (32, 64) with k = 4 | 5 or (64, 128) with k = 4.  Same contract as try_tapconv: 1 launched, 0 not eligible, < 0 error.
// mi_set_tuning key 15 / MI355_RWCONV_CONV: 0 off, 1 the k = 5 layer, 2 also 32 -> 64 channels k = 4, 3 also 64 -> 128 channels (default).
int mi_try_rwconv_conv(hipStream_t st, int dtype, const void* a, const void* w, int B, int IH, int IW, int C, int OH, int OW, int N,
                       int KH, int KW, int ldb, void* out, const float* bias, const void* mask, int relu) {
    const void* const wfrag = mi_tl_rc_wfrag;              // (consumed by THIS call whatever it dispatches to)
    mi_tl_rc_wfrag = nullptr;
    const int on = mi_rwconv_conv_mode(-1);
    mi_rwconv_mode(-1);
    const int ck = (C == 64 && N == 128) ? 2 : 1;
    if (!on || g_rwconv_mode == 0 || dtype != MI_BF16 || C != 32 * ck || N != 64 * ck || KH != KW) return 0;
    if (!((KH == 5 && ck == 1) || (KH == 4 && ck == 1 && on >= 2) || (KH == 4 && ck == 2 && on >= 3))) return 0;
    if (OH != (IH - KH) / 2 + 1 || OW != (IW - KW) / 2 + 1 || (ldb % 8) != 0 || ldb < KH * KW * C) return 0;
    if ((((uintptr_t)a) | ((uintptr_t)w) | ((uintptr_t)out) | ((uintptr_t)mask)) & 15) return 0;
    TapParams q = {};
    q.TH = q.TW = (KH + 1) / 2;
    q.GH = OH + q.TH - 1; q.GW = OW + q.TW - 1;
    const int halo = (q.TH - 1) * q.GW + q.TW - 1;
    if (halo > (ck == 2 ? RcCfg<4, 2>::MAXHALO : KH == 4 ? RcCfg<4>::MAXHALO : RcCfg<5>::MAXHALO) || q.GW <= 16) return 0;     // (GW > 16: the incremental slot walk of the staging)
    const long long MP = (long long)B * q.GH * q.GW, a_bytes = (long long)B * IH * IW * C * 2, o_bytes = (long long)B * OH * OW * N * 2;
    if (MP >= (1ll << 30) || a_bytes <= 0 || a_bytes >= (long long)G2_OOB || o_bytes >= (long long)G2_OOB) return 0;
    if (g_rwconv_mode == 1 && MP < 75000) return 0;
    q.a = a; q.a_bytes = (uint32_t)a_bytes; q.b = w; q.b_bytes = (uint32_t)o_bytes; q.ldb = ldb;
    q.B = B; q.IH = IH; q.IW = IW; q.C = C; q.OH = OH; q.OW = OW; q.N = N; q.KH = KH; q.KW = KW; q.MP = (int)MP;
    q.KC = 4 * C; q.NE = N;
    q.div_g = make_fastdiv(q.GH); q.div_gw = make_fastdiv(q.GW);
    q.out = out; q.bias = bias; q.mask = mask; q.relu = relu;
    mi_get_trace(&q.trace, &q.trace_cap);
    const int bmt = ck == 2 ? RcCfg<4, 2>::BMT : RcCfg<5>::BMT;
    const int nchunks = (int)((MP + bmt - 1) / bmt);
    static int n_cu = 0;
    if (!n_cu) { int dev = 0; hipDeviceProp_t pr; n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
    int nblk = g_rwconv_blocks > 0 ? 8 * g_rwconv_blocks : n_cu;       // one persistent block per CU; each walks a contiguous range of chunks
    if (nblk > nchunks) nblk = nchunks;
    const dim3 g((unsigned)nblk), t(256);
#define RC_LAUNCH(KH_, CK_) do { \
        if (relu && mask) MI_LAUNCH((rwconv_conv_kernel<KH_, CK_, true, true>), g, t, 0, st, q, nchunks); \
        else if (relu) MI_LAUNCH((rwconv_conv_kernel<KH_, CK_, true, false>), g, t, 0, st, q, nchunks); \
        else if (mask) MI_LAUNCH((rwconv_conv_kernel<KH_, CK_, false, true>), g, t, 0, st, q, nchunks); \
        else MI_LAUNCH((rwconv_conv_kernel<KH_, CK_, false, false>), g, t, 0, st, q, nchunks); } while (0)
    if (KH == 5 && wfrag) {                                // k = 5, 32 -> 64 channels (deconv3's input gradient) with the fragment-ordered copy (pack form 6: [tile nt (2)][live k-step (50)][lane])
        q.bfrag = wfrag;
        if (relu && mask) MI_LAUNCH((rwconv_conv_kernel<5, 1, true, true, true>), g, t, 0, st, q, nchunks);
        else if (relu) MI_LAUNCH((rwconv_conv_kernel<5, 1, true, false, true>), g, t, 0, st, q, nchunks);
        else if (mask) MI_LAUNCH((rwconv_conv_kernel<5, 1, false, true, true>), g, t, 0, st, q, nchunks);
        else MI_LAUNCH((rwconv_conv_kernel<5, 1, false, false, true>), g, t, 0, st, q, nchunks);
    } else
    if (KH == 5) RC_LAUNCH(5, 1); else if (ck == 1) RC_LAUNCH(4, 1);
    else if (wfrag) {                                      // 64 -> 128 channels with the caller's fragment-ordered weight copy (mi_rwconv_next_weights_fragment_ordered)
        q.bfrag = wfrag;
        if (relu && mask) MI_LAUNCH((rwconv_conv_kernel<4, 2, true, true, true>), g, t, 0, st, q, nchunks);
        else if (relu) MI_LAUNCH((rwconv_conv_kernel<4, 2, true, false, true>), g, t, 0, st, q, nchunks);
        else if (mask) MI_LAUNCH((rwconv_conv_kernel<4, 2, false, true, true>), g, t, 0, st, q, nchunks);
        else MI_LAUNCH((rwconv_conv_kernel<4, 2, false, false, true>), g, t, 0, st, q, nchunks);
    } else RC_LAUNCH(4, 2);
#undef RC_LAUNCH
    const int rc = mi_check_launch("rwconv_conv_kernel");
    return rc == MI_OK ? 1 : rc;
}
