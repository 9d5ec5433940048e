// ares_tile.hpp — "activation-resident" kernels for the four SMALL-GRID layers of the ConvVAE (round 4).
//
// conv4 forward / deconv1 input gradient (conv form, k4 s2: [B,8,18,128] -> [B,3,8,256]) and deconv1 forward / conv4 input gradient (gather form,
// k4 s2: [B,3,8,256] -> [B,8,18,128]) are GEMMs with a LONG reduction (K = 2048 / 1024 per parity class) and FEW output rows (24 / 144 per frame).
// As im2col tiles (gemm2: 96 x 4 blocks of 128 x 64; tapconv: 100 x 4 blocks) they ran 29.5 + 29.5 + 41.4 + 41.4 us at 15-24 % MFMA: 1.5 blocks per CU
// (two rounds of 32 latency-bound k-steps), every A row re-fetched once per column tile, and ~1,100 cycles per k-step for 256 cycles of MFMA because a
// 128 x 64 x 64 stage costs each wave six LDS-DMA instructions (~150 issue cycles apiece next to MFMAs, MI355X_MICROARCH.md).
//
// Here the roles are swapped.  A whole FRAME of input is tiny (36,864 B conv form, 12,288 B gather form), so a block keeps the complete inputs of a GROUP
// of frames resident in LDS -- loaded ONCE, contiguously, no im2col duplication, no per-step DMA -- and streams the WEIGHTS through registers:
//   * the weights live in HBM/L2 in FRAGMENT ORDER (mi_ares_pack_weights: [n tile][k16 step][lane][8 bf16] = one contiguous 1 KiB wave load per MFMA
//     operand), so a wave walks its 32 output channels' weights as a purely sequential stream with eight loads in flight; 1 MB per layer: L2 resident;
//   * a wave owns 32 output channels and ALL rows of the block (3 / 9 accumulator tiles of 32 x 32): one weight fragment feeds 3 / 9 MFMAs;
//   * the A operand of every (tap, channel step) is a 16-byte LDS read per lane at pixel(tap) + chunk; pixels are stored with their 16-byte chunks
//     ROTATED by a per-pixel amount that is linear in the pixel coordinates, chosen so that the 16 lanes of every ds_read_b128 service group hit 16 different
//     chunk positions for every tap (conflict-free; a per-tap constant shifts all lanes alike);
//   * grid = exactly one block per CU at batch 512: conv form 128 frame groups (4 frames = 96 rows) x 2 column halves; gather form 64 frame groups
//     (8 frames = 288 rows per parity class) x 4 classes.
// MFMA work per block: conv form 3 x 128 = 384 v_mfma_f32_32x32x16_bf16 per wave, gather form 5 x 64 = 320 per wave x two waves per SIMD.  At the rate the chip
// sustains with all 1,024 SIMDs on that instruction (tools/probes/mfma_probe.hip: 16.6 ns per MFMA per SIMD = 32 cycles at the ~1.93 GHz the power limit leaves
// under full matrix load -- 2.0 PFLOP/s, not the 2.5 of the 2.4 GHz boost clock) that is 6.4 / 10.6 us; the k loops measure ~10.5 / ~16 us (60-65 %), the kernels
// 19 / 26 us in the step (staging 1-5 us warm / cold, epilogue, launch) against 29.5 / 41.4 us for the im2col tile kernels they replace.
// Measured and dropped (round 4): two frames per block / two blocks per CU for the conv form (1.5 row tiles computed as two: 24.1 vs 20.4 us); the conv form's
// reduction split over two waves per SIMD (20.0 vs 20.6 us: the loop is not issue-bound at one wave per SIMD -- the "missing" cycles were the clock);
// eight frames per block for the gather form (29.4 vs 27.1 us).
#pragma once
#include "gemm_tile.hpp"
#include "gemm2_tile.hpp"

// the timing switches of these kernels (MI355_ARES_DBG: bit 0 no MFMA loop, bit 1 no staging) exist only in a -DMI355_ARES_DBG build: as run-time values they made the trip counts of
// the staging loops run-time values too (round 6: the same class of cost as the filter-gradient kernels' run-time debug branch)
#ifdef MI355_ARES_DBG
#define AR_DBG(p) ((p).dbg)
#else
#define AR_DBG(p) 0
#endif
namespace mi {

struct AresParams {
    const void* x; uint32_t x_bytes;                      // input tensor [B, IH, IW, C] (bf16)
    const void* wf;                                       // fragment-ordered weights (mi_ares_pack_weights)
    int B, M;                                             // frames; output rows (conv form: B * 24; gather form: B * 36 per parity class)
    // epilogue (store_tile)
    void* out; const float* bias; const void* mask; int relu, out_f32;
    int N, OH, OW;
    FastDiv dc_ohw[4], dc_ow[4];
    int dbg;                                              // MI355_ARES_DBG (timing experiments, wrong results): 1 no k-loop, 2 no staging, 3 neither
};

// LDS fragment reads issued and waited for BY HAND (the idiom of rwconv.hip): hipcc sinks every ds_read next to the MFMA that consumes it -- the first build of these
// kernels ran "ds_read; s_waitcnt lgkmcnt(0); v_mfma" three times per k-step, a full LDS latency in front of every MFMA (208 cycles per step for 96 of MFMA).
// asm volatile keeps the issue order; the wait names the fragments it releases as in/out operands, so their MFMAs cannot move above it.  LDS returns in order:
// lgkmcnt(N) = all but the N most recent reads have landed.
__device__ __forceinline__ void ar_lds_read(u16x8& a, uint32_t addr) { asm volatile("ds_read_b128 %0, %1" : "=&v"(a) : "v"(addr)); }
template <int N> __device__ __forceinline__ void ar_lds_wait(u16x8 (&a)[2]) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(N)); }
template <int N> __device__ __forceinline__ void ar_lds_wait(u16x8 (&a)[3]) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(N)); }
template <int N> __device__ __forceinline__ void ar_lds_wait(u16x8 (&a)[5]) {
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]) : "n"(N));
}
template <int N> __device__ __forceinline__ void ar_lds_wait(u16x8 (&a)[9]) {
    asm volatile("s_waitcnt lgkmcnt(%9)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(a[8]) : "n"(N));
}

// ---- conv form: [B,8,18,128] -> [B,3,8,256], k = 4, s = 2 ----
constexpr int AC_IH = 8, AC_IW = 18, AC_C = 128, AC_OH = 3, AC_OW = 8, AC_N = 256;
constexpr int AC_PIX = AC_IH * AC_IW;                     // 144 pixels per frame, 256 B each
constexpr int AC_KS = 16 * AC_C / 16;                     // 128 k16-steps (16 taps x 8)
constexpr int AR_D = 8;                                   // weight fragments in flight per wave

// F frames per block (4: 96 rows = three full 32-row tiles, 147 KB of LDS, one block per CU; 2: 48 rows = 1.5 tiles computed as two, 74 KB, TWO blocks per CU --
// one block's staging runs under the other's MFMAs and a second wave per SIMD covers the weight stream's latency: round 4 A/B, MI355_ARES_CFG)
template <int F, int WPE>
__global__ __launch_bounds__(256, WPE) void ares_conv_kernel(const AresParams p) {
    constexpr int ROWS = F * 24, TM = (ROWS + 31) / 32;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[F * AC_PIX * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, g = lane >> 5;
    // blocks b and b + 8 (same XCD: b % 8) are the two column halves of one frame group: its frames come out of that XCD's L2 the second time
    const int b = (int)blockIdx.x;
    const int nh = (b >> 3) & 1, fg = (b & 7) + 8 * (b >> 4);
    const int f0 = fg * F;
    if (f0 >= p.B) return;

    // ---- stage the group's frames: F x 36 LDS-DMA instructions of 1 KiB (4 pixels x 16 chunks), source-side chunk rotation ----
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    {
        const int pq = lane >> 4, pc = lane & 15;         // pixel of the quad, PHYSICAL chunk this lane fills
        // block-local pixel q = 4 t + pq is pixel f0 * 144 + q of the tensor (frames are contiguous; pixels past the last frame fall outside the descriptor:
        // zeros); its (frame, y, x) -- needed for the chunk rotation only -- advance by 16 pixels per iteration without a division
        int f = 0, y = 0, x = 4 * wave + pq;              // q < 16 < IW on the first iteration
        uint32_t vq = (uint32_t)((f0 * AC_PIX + 4 * wave + pq) * 256);
#pragma unroll 4
        for (int t = wave; t < ((AR_DBG(p) & 2) ? 0 : F * AC_PIX / 4); t += 4) {
            const int s = (8 * f + 8 * (y >> 1) + (x >> 1)) & 15;
            const int jc = (pc - s) & 15;                 // logical chunk that lives at physical position pc
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vptr)(lds + t * 1024), 16, (int)(vq + (uint32_t)jc * 16u), 0, 0, 0);
            vq += 16u * 256u;
            x += 16;
            if (x >= AC_IW) { x -= AC_IW; y += 1; if (y >= AC_IH) { y = 0; f += 1; } }
        }
    }

    // ---- this wave's weight stream: 32 output channels, 128 fragments of 1 KiB, AR_D in flight ----
    // (scalar base + 16 lane: the loads take the saddr form, no per-load 64-bit vector address arithmetic -- at one wave per SIMD every instruction of the
    //  k-step loop is kernel time: the first version spent 23 VALU + 19 SALU instructions per step on addresses and register copies for 3 MFMAs, 27 us)
    const int nt = nh * 4 + wave;                         // 32-channel tile of the 256 outputs
    const char* const wbase = (const char*)p.wf + (size_t)nt * AC_KS * 1024;
    const uint32_t lo16 = (uint32_t)lane * 16u;
    auto wload = [&](int idx) -> u16x8 { return *(const u16x8*)(wbase + (size_t)idx * 1024 + lo16); };      // idx: wave-uniform
    // Every block walks the SAME weight stream: started together, the 32 CUs of an XCD would ask one L2 channel for the same 1 KiB at the same moment, step after
    // step.  Each block therefore starts its reduction at its own tap and wraps around (fp32 accumulation in a fixed, per-block order).
    const int tap0 = (b >> 4) & 15;                       // 16 starting taps x 2 column halves = the 32 blocks of an XCD
    u16x8 bq[2 * AR_D];                                   // two taps' worth of fragments in flight (16 KiB per wave: the L2 round trip under load is ~1.5k cycles, a tap is 768)
#pragma unroll
    for (int d = 0; d < 2 * AR_D; ++d) bq[d] = wload(((tap0 + (d >> 3)) & 15) * 8 + (d & 7));

    // ---- per-lane A addressing: row r = 32 i + lrow of the block = (frame f, output pixel (oy, ox)) ----
    uint32_t pixbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = i * 32 + lrow;
        int f = r / 24;
        const int rem = r - f * 24;
        if (f >= F) f = F - 1;                            // (rows past the block's last frame: any resident pixel; store_tile's row limit drops them)
        const int oy = rem >> 3, ox = rem & 7;
        pixbase[i] = (uint32_t)((f * AC_PIX + 2 * oy * AC_IW + 2 * ox) * 256);
    }
    const int s0g = (lrow & 15) + g;                      // (8 f + 8 oy + ox) & 15 = r & 15: the row's share of the chunk rotation (+ the lane group's chunk)

    f32x16 acc[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    __syncthreads();                                      // (s_waitcnt vmcnt(0) + barrier: the frames are in LDS -- and so are the first weight fragments)

    // tap (ky, kx): LDS byte offset of its pixel relative to the row's base pixel, and the byte position of chunk 0 of lane group g inside the (rotated) pixel;
    // channel step c of the tap reads chunk position (q0 + 32 c) & 255
    auto tap_of = [&](int tap, uint32_t (&tb)[TM], uint32_t& q0) {
        const int ky = tap >> 2, kx = tap & 3;            // (wave-uniform: scalar unit)
        const uint32_t toff = (uint32_t)((ky * AC_IW + kx) * 256);
        q0 = (uint32_t)((s0g + 8 * (ky >> 1) + (kx >> 1)) & 15) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) tb[i] = pixbase[i] + toff;
    };
    uint32_t tb[TM], q0;
    tap_of(tap0, tb, q0);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    u16x8 A[2][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) ar_lds_read(A[0][i], lds0 + tb[i] + q0);
#pragma unroll 1
    for (int t2 = 0; t2 < ((AR_DBG(p) & 1) ? 0 : 8); ++t2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {                     // tap number tl of this block's order; its fragments sit in ring half h
            const int tl = 2 * t2 + h;
            uint32_t tbn[TM], q0n;
            tap_of((tap0 + tl + 1) & 15, tbn, q0n);       // the next tap's addressing: its first fragments are requested during this tap's last step
            const int wnext = ((tap0 + tl + 2) & 15) * 8; // this ring half is refilled with the fragments of the tap after next
#pragma unroll
            for (int d = 0; d < AR_D; ++d) {
                if (d + 1 < AR_D) {
                    const uint32_t r = lds0 + ((q0 + 32u * (d + 1)) & 255u);
#pragma unroll
                    for (int i = 0; i < TM; ++i) ar_lds_read(A[(d + 1) & 1][i], tb[i] + r);
                } else {                                  // (unconditional: behind the last tap this re-reads the first one's fragments -- a branch per step costs more)
#pragma unroll
                    for (int i = 0; i < TM; ++i) ar_lds_read(A[0][i], lds0 + tbn[i] + q0n);
                }
                ar_lds_wait<TM>(A[d & 1]);                // this step's fragments have landed (the TM reads just issued may still be in flight)
                const u16x8 bw = bq[8 * h + d];
                bq[8 * h + d] = wload(wnext + d);         // (unconditional as well: the last two taps re-request fragments nobody consumes)
#pragma unroll
                for (int i = 0; i < TM; ++i) Frag<bf16_t>::mma(bw, A[d & 1][i], acc[i][0]);      // D[row = channel][col = pixel]
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) tb[i] = tbn[i];
            q0 = q0n;
        }
    }
    ar_lds_wait<0>(A[0]);                                 // (the over-read of the last step: its destination registers are free only once it has landed)
    // (rows of the block beyond ROWS belong to the next frame group: the row limit of the call stops at this block's last row)
    const int mlim = min(p.M, (fg + 1) * ROWS);
    store_tile<bf16_t, A_CONV, TM, 1>(p, acc, fg * ROWS, nt * 32, 0, 0, lrow, g, mlim, 0, 0, 0, 0);
}

// ---- gather form: [B,3,8,256] -> [B,8,18,128], k = 4, s = 2 (stride-2 transposed conv; one GEMM per output parity class) ----
constexpr int AG_IH = 3, AG_IW = 8, AG_C = 256, AG_OH = 8, AG_OW = 18, AG_N = 128;
constexpr int AG_PIX = AG_IH * AG_IW;                     // 24 pixels per frame, 512 B each
constexpr int AG_KS = 4 * AG_C / 16;                      // 64 k16-steps (2 x 2 taps x 16)
constexpr int AG_RPF = 36;                                // output pixels of one parity class per frame (4 x 9)

// F frames per block (8: 288 rows per class = nine 32-row tiles, 98 KB, one block per CU; 4: 144 rows = 4.5 tiles computed as five, 49 KB, two blocks per CU)
template <int F, int WPE>
__global__ __launch_bounds__(256, WPE) void ares_gather_kernel(const AresParams p) {
    constexpr int ROWS = F * AG_RPF, TM = (ROWS + 31) / 32;
    constexpr int AG_ZERO = F * AG_PIX * 512;             // byte offset of the all-zero pixel (taps that fall outside the input)
    __shared__ __attribute__((aligned(1024))) unsigned char lds[AG_ZERO + 512];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, g = lane >> 5;
    const int b = (int)blockIdx.x;
    // the 32 blocks of an XCD (b % 8) cover the four parity classes x eight starting points of the reduction: no two of them walk the same weight stream in step
    const int cls = (b >> 3) & 3, fg = (b & 7) + 8 * (b >> 5);
    const int ph = cls >> 1, pw = cls & 1;
    const int f0 = fg * F;
    if (f0 >= p.B) return;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    {
        const int pq = lane >> 5, pc = lane & 31;         // pixel of the pair, PHYSICAL chunk (two 256-byte halves of 16 chunks, each rotated by itself)
        // block-local pixel q = 2 t + pq = tensor pixel f0 * 24 + q; 8 pixels (one input row) further per iteration: x stays, y and the frame advance
        const int x = (2 * wave + pq) & 7;
        int f = 0, y = 0;
        uint32_t vq = (uint32_t)((f0 * AG_PIX + 2 * wave + pq) * 512);
#pragma unroll 4
        for (int t = wave; t < ((AR_DBG(p) & 2) ? 0 : F * AG_PIX / 2); t += 4) {
            const int s = (4 * f + 9 * y + x) & 15;
            const int jc = (pc & 16) | ((pc - s) & 15);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vptr)(lds + t * 1024), 16, (int)(vq + (uint32_t)jc * 16u), 0, 0, 0);
            vq += 8u * 512u;
            y += 1;
            if (y >= AG_IH) { y = 0; f += 1; }
        }
        if (tid < 32) *(f32x4*)(lds + AG_ZERO + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const char* const wbase = (const char*)p.wf + (size_t)(cls * 4 + wave) * AG_KS * 1024;
    const uint32_t lo16 = (uint32_t)lane * 16u;
    auto wload = [&](int idx) -> u16x8 { return *(const u16x8*)(wbase + (size_t)idx * 1024 + lo16); };      // idx: wave-uniform
    const int o0 = (b >> 5) & 7;                          // starting point of this block's reduction, in units of 8 k-steps (see ares_conv_kernel)
    u16x8 bq[AR_D];
#pragma unroll
    for (int d = 0; d < AR_D; ++d) bq[d] = wload(o0 * 8 + d);

    // row r = 32 i + lrow of the block = (frame f, class pixel (j, ii)); tap (th, tw) reads input pixel (j - th, ii - tw) or the zero pixel
    int rf[TM], rj[TM], ri[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = i * 32 + lrow;
        rf[i] = r / AG_RPF;
        const int rem = r - rf[i] * AG_RPF;
        if (rf[i] >= F) rf[i] = F - 1;                    // (rows past the block's last frame: any resident pixel; never stored)
        rj[i] = rem / 9; ri[i] = rem - rj[i] * 9;
    }
    const int s0g = (lrow & 15) + g;                      // (4 f + 9 j + ii) & 15 = r & 15 (+ the lane group's chunk)
    uint32_t abase[TM], q0;
    auto set_tap = [&](int tap) {
        const int th = tap >> 1, tw = tap & 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int y = rj[i] - th, x = ri[i] - tw;
            const bool ok = y >= 0 && y < AG_IH && x >= 0 && x < AG_IW;
            abase[i] = ok ? (uint32_t)((rf[i] * AG_PIX + y * AG_IW + x) * 512) : (uint32_t)AG_ZERO;
        }
        q0 = (uint32_t)((s0g - 9 * th - tw) & 15) << 4;  // byte position of chunk 0 of lane group g inside a rotated 256-byte half (the zero pixel is zero everywhere)
    };

    f32x16 acc[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    __syncthreads();

    // 8 octets of 8 k-steps: octet o = (tap o >> 1, channel half o & 1); channel step c of the octet reads chunk position half * 256 + ((q0 + 32 c) & 255)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    u16x8 A[2][TM];
    set_tap(o0 >> 1);
    {
        const uint32_t r = lds0 + (uint32_t)(o0 & 1) * 256u + q0;
#pragma unroll
        for (int i = 0; i < TM; ++i) ar_lds_read(A[0][i], abase[i] + r);
    }
#pragma unroll 1
    for (int t8 = 0; t8 < ((AR_DBG(p) & 1) ? 0 : 8); ++t8) {
        const int o = (o0 + t8) & 7, on = (o0 + t8 + 1) & 7;
        const uint32_t hoff = (uint32_t)(o & 1) * 256u;
#pragma unroll
        for (int d = 0; d < AR_D; ++d) {
            if (d + 1 < AR_D) {
                const uint32_t r = lds0 + hoff + ((q0 + 32u * (d + 1)) & 255u);
#pragma unroll
                for (int i = 0; i < TM; ++i) ar_lds_read(A[(d + 1) & 1][i], abase[i] + r);
            } else {                                      // (unconditional, see ares_conv_kernel)
                if ((on & 1) == 0) set_tap(on >> 1);      // (wave-uniform; this step's fragments were requested with the old addresses)
                const uint32_t r = lds0 + (uint32_t)(on & 1) * 256u + q0;
#pragma unroll
                for (int i = 0; i < TM; ++i) ar_lds_read(A[0][i], abase[i] + r);
            }
            ar_lds_wait<TM>(A[d & 1]);
            const u16x8 bw = bq[d];
            bq[d] = wload(on * 8 + d);
#pragma unroll
            for (int i = 0; i < TM; ++i) Frag<bf16_t>::mma(bw, A[d & 1][i], acc[i][0]);
        }
    }
    ar_lds_wait<0>(A[0]);
    const int mlim = min(p.M, (fg + 1) * ROWS);
    store_tile<bf16_t, A_DECONV, TM, 1>(p, acc, fg * ROWS, wave * 32, 0, 0, lrow, g, mlim, cls, ph, pw, 0);
}

// ---- gather form, mid layer: [B,8,18,128] -> [B,18,38,64], k = 4, s = 2 (deconv2 forward / conv3's input gradient) ----
// One FRAME per block (36,864 B of LDS; two blocks per CU by registers -- one block's staging and epilogues run under the other's MFMAs; three would spill), all four parity classes in
// turn (171 = 9 x 19 output pixels per class and frame).  The four waves are 2 output tiles (N = 64) x 2 row halves of 96 rows (three 32-row tiles; rows
// 171 .. 191 of the second half are computed on whatever pixel their address hits and never stored).  Per class 4 taps x 8 channel steps; the weight
// stream of a wave is [class][tile][32 fragments]; the class order starts at block-dependent class (no two neighbours walk the same stream in step).
constexpr int G2_IH = 8, G2_IW = 18, G2_OH = 18, G2_OW = 38, G2_N = 64;
constexpr int G2_PIX = G2_IH * G2_IW;                     // 144 pixels of 256 B
constexpr int G2_RPF = 9 * 19;                            // 171 output pixels of one parity class per frame
constexpr int G2_ZERO = G2_PIX * 256;

__global__ __launch_bounds__(256, 2) void ares_gather2_kernel(const AresParams p) {
    constexpr int TM = 3;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[G2_ZERO + 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, g = lane >> 5;
    const int frame = (int)blockIdx.x;
    if (frame >= p.B) return;
    const int nt = wave & 1, rh = wave >> 1;

    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)p.x_bytes, 0x00020000);
    {
        const int pq = lane >> 4, pc = lane & 15;
        int y = 0, x = 4 * wave + pq;                     // pixel q = 4 t + pq, 16 pixels further per iteration; rotation s(y, x) = (19 y + x) & 15
        uint32_t vq = (uint32_t)((frame * G2_PIX + 4 * wave + pq) * 256);
#pragma unroll 3
        for (int t = wave; t < ((AR_DBG(p) & 2) ? 0 : G2_PIX / 4); t += 4) {
            const int s = (3 * y + x) & 15;
            const int jc = (pc - s) & 15;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_vptr)(lds + t * 1024), 16, (int)(vq + (uint32_t)jc * 16u), 0, 0, 0);
            vq += 16u * 256u;
            x += 16;
            if (x >= G2_IW) { x -= G2_IW; y += 1; }
        }
        if (tid < 16) *(f32x4*)(lds + G2_ZERO + tid * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const uint32_t lo16 = (uint32_t)lane * 16u;
    const char* const wf = (const char*)p.wf + (size_t)nt * 32 * 1024;
    auto wload = [&](int cls_, int idx) -> u16x8 { return *(const u16x8*)(wf + ((size_t)cls_ * 64 + (size_t)idx) * 1024 + lo16); };      // fragment idx (0 .. 31) of (class, tile nt)
    const int cls0 = frame & 3;
    u16x8 bq[AR_D];
#pragma unroll
    for (int d = 0; d < AR_D; ++d) bq[d] = wload(cls0, d);

    int rj[TM], ri[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = rh * 96 + i * 32 + lrow;            // class-local output pixel of the frame (>= 171: dropped by the store's row limit)
        rj[i] = r / 19; ri[i] = r - rj[i] * 19;
    }
    const int s0g = (lrow & 15) + g;                      // (19 j + ii) & 15 = r & 15 (96 and 32 are multiples of 16) + the lane group's chunk
    auto tap_of = [&](int tap, uint32_t (&tb)[TM], uint32_t& q0) {
        const int th = tap >> 1, tw = tap & 1;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int y = rj[i] - th, x = ri[i] - tw;
            const bool ok = y >= 0 && y < G2_IH && x >= 0 && x < G2_IW;
            tb[i] = ok ? (uint32_t)((y * G2_IW + x) * 256) : (uint32_t)G2_ZERO;
        }
        q0 = (uint32_t)((s0g - 3 * th - tw) & 15) << 4;
    };

    f32x16 acc[TM][1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    __syncthreads();

    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    uint32_t tb[TM], q0;
    tap_of(0, tb, q0);
    u16x8 A[2][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) ar_lds_read(A[0][i], lds0 + tb[i] + q0);
    const int mlim = min(p.M, (frame + 1) * G2_RPF);
#pragma unroll 1
    for (int G = 0; G < ((AR_DBG(p) & 1) ? 0 : 16); ++G) {    // group = (class in this block's order, tap): 8 channel steps
        const int cls = (cls0 + (G >> 2)) & 3, tap = G & 3;
        const int Gn = (G + 1) & 15, clsn = (cls0 + (Gn >> 2)) & 3;
        uint32_t tbn[TM], q0n;
        tap_of(Gn & 3, tbn, q0n);
#pragma unroll
        for (int d = 0; d < AR_D; ++d) {
            if (d + 1 < AR_D) {
                const uint32_t r = lds0 + ((q0 + 32u * (d + 1)) & 255u);
#pragma unroll
                for (int i = 0; i < TM; ++i) ar_lds_read(A[(d + 1) & 1][i], tb[i] + r);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i) ar_lds_read(A[0][i], lds0 + tbn[i] + q0n);
            }
            ar_lds_wait<TM>(A[d & 1]);
            const u16x8 bw = bq[d];
            bq[d] = wload(clsn, (Gn & 3) * 8 + d);
#pragma unroll
            for (int i = 0; i < TM; ++i) Frag<bf16_t>::mma(bw, A[d & 1][i], acc[i][0]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) tb[i] = tbn[i];
        q0 = q0n;
        if (tap == 3) {                                   // the class is complete: store its three tiles, start the next class from zero
            store_tile<bf16_t, A_DECONV, TM, 1>(p, acc, frame * G2_RPF + rh * 96, nt * 32, 0, 0, lrow, g, mlim, cls, cls >> 1, cls & 1, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        }
    }
    ar_lds_wait<0>(A[0]);
}

// ---- weights -> fragment order (bf16), from the fp32 master tensor ----
// form 0 (conv form): src is [K = 16 x 128][N = 256] (HWIO conv kernel, or a [kh,kw,out,in] transposed-conv kernel read as HWIO for its input gradient):
//     dst[((nt * 128 + ks) * 64 + l) * 8 + e] = src[(ks * 16 + (l >> 5) * 8 + e) * 256 + nt * 32 + (l & 31)]
// form 1 (gather form): src is [kh][kw][n = 128][c = 256] ([kh,kw,out,in] transposed-conv kernel, or an HWIO conv kernel read that way for its input gradient):
//     dst[(((cls * 4 + nt) * 64 + ks) * 64 + l) * 8 + e] = src[((kh * 4 + kw) * 128 + nt * 32 + (l & 31)) * 256 + (ks & 15) * 16 + (l >> 5) * 8 + e],
//     tap = ks >> 4 = (th, tw), (kh, kw) = (ph + 2 th, pw + 2 tw), cls = (ph, pw)
struct AresPackJobs { const float* src[8]; bf16_t* dst[8]; int form[8]; int n; };      // up to eight copies in one launch: job = blockIdx.x >> 8
__global__ __launch_bounds__(256) void ares_pack_kernel(const AresPackJobs jobs) {
    const int job = (int)blockIdx.x >> 8;
    if (job >= jobs.n) return;
    const float* __restrict__ src = jobs.src[job];
    bf16_t* __restrict__ dst = jobs.dst[job];
    const int form = jobs.form[job];
    const int gid = ((int)blockIdx.x & 255) * 256 + (int)threadIdx.x;           // one thread per (fragment, lane): 8 values; 65,536 threads x 8 = 524,288 weights = 1 MB of bf16
    const int l = gid & 63, frag = gid >> 6;
    float v[8];
    if (form == 0) {
        const int ks = frag & 127, nt = frag >> 7;
        const int k = ks * 16 + (l >> 5) * 8, n = nt * 32 + (l & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(long long)(k + e) * 256 + n];
    } else if (form == 2) {                               // gather form, mid layer: [kh][kw][n = 64][c = 128] -> [class][tile (2)][32 fragments]; 256 fragments = 256 KB
        if (frag >= 256) return;
        const int ks = frag & 31, nt = (frag >> 5) & 1, cls = frag >> 6;
        const int tap = ks >> 3, th = tap >> 1, tw = tap & 1;
        const int kh = (cls >> 1) + 2 * th, kw = (cls & 1) + 2 * tw;
        const float* s = src + ((long long)((kh * 4 + kw) * 64 + nt * 32 + (l & 31))) * 128 + (ks & 7) * 16 + (l >> 5) * 8;
        const f32x4 a = *(const f32x4*)s, c = *(const f32x4*)(s + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = c[0]; v[5] = c[1]; v[6] = c[2]; v[7] = c[3];
    } else if (form == 4) {                               // gather form of the register-weight kernel, k = 5 (deconv3 forward; round 6): kernel [5][5][n = 32][c = 64] ->
        if (frag >= 144) return;                          // [class (ph, pw)][tap (ta, tb) of 3 x 3][k-step kk (4)][lane (lgrp, n)]: kernel position (ph + 2 (2 - ta), pw + 2 (2 - tb)); taps past the 5 x 5 kernel: zeros
        const int kk = frag & 3, tap = (frag >> 2) % 9, cls = frag / 36;
        const int kh = (cls >> 1) + 2 * (2 - tap / 3), kw = (cls & 1) + 2 * (2 - tap % 3);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (kh < 5 && kw < 5) ? src[((long long)((kh * 5 + kw) * 32 + (l & 31))) * 64 + kk * 16 + (l >> 5) * 8 + e] : 0.f;
    } else if (form == 5) {                               // conv2's kernel for the fused encoder head (enc12_tile.hpp; round 6): HWIO [k = 512][n = 64] -> [tile nt (2)][k-step f (32)][lane]
        if (frag >= 64) return;
        const int f = frag & 31, nt = frag >> 5;
        const int k = f * 16 + (l >> 5) * 8, n = nt * 32 + (l & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(long long)(k + e) * 64 + n];
    } else if (form == 6) {                               // conv form of the register-weight kernel, k = 5, 32 -> 64 channels (rwconv_conv_kernel<5, 1>: deconv3's input gradient; round 6): deconv3's
        if (frag >= 100) return;                          // [k = (kh, kw, c = 32)][n = 64] kernel -> [tile nt (2)][live k-step (50)][lane]; the 50 live (tap, s) pairs in the prologue's issue order:
        const int nt = frag / 50;                         // tap (ta, tb) of 3 x 3, s = 4 ph + 2 pw + kk with kh = 2 ta + ph < 5, kw = 2 tb + pw < 5
        int r = frag - nt * 50;
        const int ta = r < 20 ? 0 : r < 40 ? 1 : 2; r -= ta < 2 ? 20 * ta : 40;
        const int nph = ta < 2 ? 2 : 1;
        const int tb = r < nph * 4 ? 0 : r < nph * 8 ? 1 : 2; r -= tb * nph * 4;
        const int npw = tb < 2 ? 2 : 1;
        const int ph = r / (npw * 2); r -= ph * npw * 2;
        const int pw = r >> 1, kk = r & 1;
        const int k = ((2 * ta + ph) * 5 + 2 * tb + pw) * 32 + kk * 16 + (l >> 5) * 8, n = nt * 32 + (l & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(long long)(k + e) * 64 + n];
    } else if (form == 3) {                               // conv form of the register-weight kernel (rwconv_conv_kernel<4, 2>: conv3 forward, deconv2's input gradient; round 6):
        if (frag >= 256) return;                          // [k = (kh, kw, c = 64)][n = 128] -> [tile nt (4)][k-step (tap 4 x s 16)][lane]: lane (lgrp, n & 31) holds channels 16 (s & 3) + 8 lgrp .. + 7
        const int kidx = frag & 63, nt = frag >> 6;       // of kernel position kh = 2 (tap >> 1) + (s >> 3), kw = 2 (tap & 1) + ((s >> 2) & 1): the order the kernel's prologue loads its 64 fragments in
        const int tap = kidx >> 4, st = kidx & 15;
        const int kh = 2 * (tap >> 1) + (st >> 3), kw = 2 * (tap & 1) + ((st >> 2) & 1);
        const int k = (kh * 4 + kw) * 64 + (st & 3) * 16 + (l >> 5) * 8, n = nt * 32 + (l & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[(long long)(k + e) * 128 + n];
    } else {
        const int ks = frag & 63, nt = (frag >> 6) & 3, cls = frag >> 8;
        const int tap = ks >> 4, th = tap >> 1, tw = tap & 1;
        const int kh = (cls >> 1) + 2 * th, kw = (cls & 1) + 2 * tw;
        const float* s = src + ((long long)((kh * 4 + kw) * 128 + nt * 32 + (l & 31))) * 256 + (ks & 15) * 16 + (l >> 5) * 8;
        const f32x4 a = *(const f32x4*)s, c = *(const f32x4*)(s + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = c[0]; v[5] = c[1]; v[6] = c[2]; v[7] = c[3];
    }
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(v[e]);
    *(u16x8*)(dst + (long long)gid * 8) = o;
}

}  // namespace mi
