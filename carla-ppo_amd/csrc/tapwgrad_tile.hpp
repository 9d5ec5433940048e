// tapwgrad_tile.hpp — weight gradients of the stride-2 conv / transposed-conv layers on raw-staged slot tiles (bf16, gfx950).
//
// Same slot formulation as tapconv_tile.hpp.  With A[P][kc] the slot stream (conv form: the 2x2 pixel block, 4C channels;
// gather form: the pixel, C channels) and D[P][ne] the output-side gradient of position P (conv form: dy[b,gy,gx,:], N
// columns; gather form: dy of the four output parities, 4N columns):
//     dW'[tap][kc][ne] = sum_P A[P + ta*GW + tb][kc] * D[P][ne]
// i.e. a GEMM whose reduction index is the POSITION.  A block owns a [KT*32 channels] x [NTB*32 outputs] x [all taps] slab of
// dW' (4..8 accumulator tiles per wave) and walks a contiguous range of positions, 128 per step:
//   * the slot range [P, P+128+halo) x KCB channels and the 128 x NEB gradient rows are staged once per step by LDS-DMA
//     (zero fill outside the images / past the range through the buffer range check); every tap reuses the staged slots,
//   * MFMA operands need 8 consecutive POSITIONS per lane: they are read from the position-major LDS tiles with the hardware
//     transpose read (ds_read_b64_tr_b16); the tap shift is a row offset; all per-lane offsets are precomputed once,
//   * rows are unpadded (DMA writes lane-linear); the 4 rows x 64 B a half-wave transpose read touches are spread over the
//     four bank quarters by an XOR on the 16-byte chunk index applied on the DMA source side,
//   * one block per CU: the reduction over positions is split ~256 ways in total, 4x fewer fp32 atomics than wgrad_kernel.
// fp32 (parity mode) stays on wgrad_kernel: the transpose read is a 16-bit instruction.
#pragma once
#include <type_traits>
#include "tapconv_tile.hpp"
#include "wgrad_tile.hpp"

namespace mi {

constexpr int TW_BP = 128;           // positions per step
constexpr int TW_NT = 512;           // threads (8 waves)
constexpr int TW_MAXPAIR = 32;

struct TapWgradParams {
    const void* a; uint32_t a_bytes;
    const void* d; uint32_t d_bytes;
    int B, IH, IW, C;                // slot-side tensor
    int OH, OW, N;                   // gradient tensor dy[B,OH,OW,N]
    int KH, KW;
    int GH, GW, HY, HX;
    int KC, NE, MP;
    int nkb;                         // channel blocks (block column by = nb * nkb + kb)
    int gx, gy;                      // position splits x block columns.  The launch is 1-D over ceil(gx / 8) * 8 * gy blocks: block id -> XCD id % 8, slot
                                     // id / 8; the gy column blocks of one position split take consecutive slots of ONE XCD, so the slot / gradient tiles
                                     // that several column blocks read (each covers a channel x output slice of dW) are fetched from HBM once and
                                     // served to the others by that XCD's L2 (conv3 / deconv2 filter gradients read 126 / 119 MB for 64 MB of tensors before)
    int pos_per_split;               // multiple of TW_BP
    int npairs;                      // (tap, output tile) pairs of a block, dealt round-robin to the 8 waves
    unsigned char pair_tap[TW_MAXPAIR], pair_nt[TW_MAXPAIR], pair_first[TW_MAXPAIR];   // first: first pair of its output tile
    float* dbias;                    // optional fused bias gradient: dbias[n] += sum over positions (and parities) of dy
    FastDiv div_g, div_gw, div_n, div_2c, div_c;
    float* out;
    float* slabs; long long slab_stride;   // optional: per-split partial sums [gridDim.x][slab_stride] (plain stores) reduced by reduce_slabs_kernel
    float* bias_part; int bias_nh;   // optional (with slabs): the bias-gradient partial sums of position split bx, half h go to bias_part[(bx * bias_nh + h) * NE + ne] (plain
                                     // stores, every element exactly once) and the ordered reduce adds them to dbias; NULL: fp32 atomics on dbias
    int slab_bf16;                   // the partial sums are stored ROUNDED TO BF16 (round 3; the bf16 engine's default, mi_set_tuning key 18): ~256 slabs per
                                     // element, each 2^-9 relative with independent signs, add ~1e-4 of the element's own scale to a gradient whose operands
                                     // were bf16 to begin with -- and halve the 211 MB written + 214 MB read per step that the slabs cost
    long long* trace; int trace_cap;   // debug stamps (mi_debug_set_trace)
    int dbg_cheap_addr;                // TIMING INSTANTIATIONS (mi_set_tuning key 2; wrong results, honest durations -- tools/wgrad_ablate.py): 2 = trivial DMA addresses inside ONE
                                       // megabyte (every load an L2 hit: NOT "the cost of the address arithmetic" as round 1 read it, but the kernel without its HBM traffic), 3 = no
                                       // loads at all (stale LDS), 4 = no fragment reads / MFMAs (loads, barriers and stores only), 5 = no slab stores
};

// LDS-DMA issued through inline asm: hipcc drains every builtin LDS-DMA (s_waitcnt vmcnt(0)) in front of the next
// ds_read_b64_tr_b16 it cannot prove disjoint, which would serialise prefetch and compute; asm loads are invisible to that
// pass, so the step barrier carries an explicit vmcnt(0).  m0 = LDS byte address of lane 0's 16 bytes (wave-uniform).
#ifdef MI355_TW_SLAB_PLAIN                                 // (variant build: the bf16 slab stores without the streaming hint)
#define TW_SLAB_STORE(v, ptr) (*(ptr) = (v))
#else
#define TW_SLAB_STORE(v, ptr) __builtin_nontemporal_store((v), (ptr))
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t tw_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 make_srd(const void* base, uint32_t bytes) {
    const unsigned long long b = (unsigned long long)base;
    u32x4 r;
    r[0] = (uint32_t)b; r[1] = (uint32_t)(b >> 32) & 0xffffu; r[2] = bytes; r[3] = 0x00020000u;
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    return r;
}
__device__ __forceinline__ void dma16_asm(const u32x4 srd, uint32_t lds_addr, uint32_t voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(srd) : "memory", "m0");
}

// swizzle of the 16-byte chunk index by LDS row so that 4 consecutive rows x 64 B fall into 4 different bank quarters
template <int CPR> __device__ __forceinline__ int tw_swz(int row) {
    return CPR == 16 ? (row & 3) << 2 : ((row >> 1) & 1) << 2;
}

// dW element of accumulator row group (rows row .. row + 3 at column lcol) of the tile (tap, kt, nt) of block column (kc0, ne0):
// index of the first of the 4 rows, the stride between them, and whether the group exists (C and N are multiples of 4)
template <int MODE, int TAPS>
__device__ __forceinline__ bool tw_dw_index(const TapWgradParams& p, int kc0, int ne0, int tap, int kt, int nt, int row, int lcol, long long& base, long long& stride) {
    const int ta = tap / TAPS, tb = tap % TAPS;
    if constexpr (MODE == TC_CONV) {                      // rows = channels kc = (ph, pw, c); column = output
        const int kc = kc0 + kt * 32 + row, ne = ne0 + nt * 32 + lcol;
        uint32_t phh, rem, pww, c;
        p.div_2c.divmod((uint32_t)(kc < p.KC ? kc : 0), phh, rem);
        p.div_c.divmod(rem, pww, c);
        const int kh = 2 * ta + (int)phh, kw = 2 * tb + (int)pww;
        base = ((long long)(kh * p.KW + kw) * p.C + c) * p.N + ne; stride = p.N;
        return kc < p.KC && ne < p.NE && kh < p.KH && kw < p.KW;
    } else {                                              // rows = outputs ne = (class, n); column = channel
        const int ne = ne0 + nt * 32 + row, kc = kc0 + kt * 32 + lcol;
        uint32_t cls, n;
        p.div_n.divmod((uint32_t)(ne < p.NE ? ne : 0), cls, n);
        const int kh = (int)(cls >> 1) + 2 * (p.HY - ta), kw = (int)(cls & 1) + 2 * (p.HX - tb);
        base = ((long long)(kh * p.KW + kw) * p.N + n) * p.C + kc; stride = p.C;
        return kc < p.KC && ne < p.NE && kh < p.KH && kw < p.KW;
    }
}

// MODE: TC_CONV | TC_GATHER;  KT: 32-channel tiles per block (KCB = 32 KT);  NTB: 32-output tiles per block (NEB = 32 NTB);
// PPW: (tap, output tile) pairs per wave (accumulators: PPW * KT tiles)
// SPLIT (needs 4 taps x NTB == 2, PPW == NTB): wave = (tap, position half); a wave keeps the KT x NTB tiles of its tap over its 64
// positions of every step, so a slot fragment is read from LDS once for both output tiles (48 instead of 80 transpose reads per
// 32 MFMAs); the two halves meet in LDS at the end.
// LDEC (round 6, VERDICT r05 item 1 -- the VALU diet): the source offset of a DMA row is a function of its POSITION only, and the 64 lanes of a DMA instruction cover 4 or 8
// rows: decoding it per lane and per instruction (two magic-number divisions = 6 quarter-rate multiplies, the image bounds, the address: ~25 VALU instructions, ~200 issue
// cycles, nine times per wave and 128-position step -- as many VALU cycles per SIMD as the step's MFMAs take) computed every row's offset 8 or 16 times over.  Here each wave decodes
// ALL the rows it is going to request for a step at once, one row per LANE (28 + 16 or 32 + 16 rows: 44 / 48 lanes), two steps ahead (at the top of a step, while its first
// fragment reads are in flight), and a DMA instruction fetches its rows'
// offsets with ONE ds_bpermute_b32 (crossbar only, no LDS banks); what depends on the lane's 16-byte chunk rides along: the channel offset is added,
// and the two "this pixel's odd row / column is outside the image" bits of the 2 x 2 forms travel in the low bits of the row offset (rows are >= 64 bytes apart) and are tested
// against the chunk's own (ph, pw).  Same addresses, same zero fill, bit-identical sums (tests/test_ops_gpu.py runs both forms).  VALU instructions per MFMA 7.3 -> 4.5 -- and
// the step time is EQUAL to the per-instruction decode (0.7929 / 0.7932 ms, DESIGN_HISTORY 3.16c): these kernels are not VALU-issue bound.  Default off (MI355_TW_LDEC, key 24).
template <int MODE, int TAPS, int KT, int NTB, int PPW, bool SPLIT = false, bool LDEC = false, bool DBG = false>
__global__ __launch_bounds__(TW_NT) void tapwgrad_kernel(const TapWgradParams p) {
    // the timing instantiations (DBG) are their own kernels: a wave-uniform `if (p.dbg...) continue;` inside the unrolled MFMA loop of the PRODUCT kernel cuts its basic
    // blocks at every k-step and cost the k = 5 kernel 6 us of 57 (round 6, found in the counter pass)
    const int dbgm = DBG ? p.dbg_cheap_addr : 0;
    static_assert(!SPLIT || (TAPS == 2 && NTB == 2 && PPW == NTB), "split layout: 4 taps x 2 position halves = 8 waves");
    typedef bf16_t T;
    constexpr int ESZ = 2, VE = 8;
    constexpr int KCB = 32 * KT, NEB = 32 * NTB;
    constexpr int PA = KCB * ESZ, PD = NEB * ESZ;        // LDS row pitch of the slot tile / gradient tile (128 | 256 B)
    constexpr int CPA = PA / 16, CPD = PD / 16;          // 16-byte chunks per row
    constexpr int SPI_A = 1024 / PA, SPI_D = 1024 / PD;  // rows per DMA instruction
    constexpr int MAXSLOT = TW_BP + TC_MAXHALO;          // 224
    constexpr int NIA = (MAXSLOT / SPI_A + 7) / 8;       // slot-tile DMA instructions per wave (upper bound)
    constexpr int NID = TW_BP / SPI_D / 8;               // gradient-tile DMA instructions per wave
    constexpr int ASTAGE = MAXSLOT * PA, DSTAGE = TW_BP * PD, STAGE = ASTAGE + DSTAGE;
    static_assert(NID >= 1 && 2 * STAGE <= 160 * 1024, "tile config");

    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tr_n = 0;
    const int bslot = (int)blockIdx.x >> 3;
    const int by = bslot % p.gy, bx = (bslot / p.gy) * 8 + ((int)blockIdx.x & 7);
    if (bx >= p.gx) return;
    long long* const tr = p.trace ? p.trace + ((long long)(by * p.gx + bx) * 8 + (tid >> 6)) * 32 : nullptr;
    const bool tr_on = tr && ((long long)(by * p.gx + bx) * 8 + 8) * 32 <= p.trace_cap && lane == 0;
// (round 6: compiling the stamps of this file, rwconv.hip and tapconv_tile.hpp out of the product build -- per-lane `if (tr_on)` branches at the phase boundaries -- measured
//  SLOWER, 0.8142 against 0.8080 ms per step over four interleaved rounds: the hand-scheduled loops were tuned with those block boundaries in place.  They stay.)
#define TW_STAMP() do { if (tr_on && tr_n < 32) tr[tr_n++] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
    TW_STAMP();
#ifdef MI355_TW_EPI_STAMPS                                // (variant build of tools/trace_tapwgrad.py: the phases of the epilogue; the product kernel keeps its stamp sites as they are, see above)
#define TW_EPI_STAMP() TW_STAMP()
#else
#define TW_EPI_STAMP() do {} while (0)
#endif
    const int kb = by % p.nkb, nb = by / p.nkb;
    const int kc0 = kb * KCB, ne0 = nb * NEB;
    const int Pbeg = bx * p.pos_per_split;
    const int Pend = min(p.MP, Pbeg + p.pos_per_split);
    if (Pbeg >= Pend) return;
    const int nsteps = (Pend - Pbeg + TW_BP - 1) / TW_BP;
    const int halo = (TAPS - 1) * p.GW + TAPS - 1;
    const int ninstrA = (TW_BP + halo + SPI_A - 1) / SPI_A;

    const u32x4 rsA = make_srd(p.a, p.a_bytes), rsD = make_srd(p.d, p.d_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;   // LDS byte address of the array

    // ---------------- DMA roles ----------------
    // slot tile: instruction t = wave + 8 i fills rows SPI_A t ..; lane -> row SPI_A t + lane / CPA, physical chunk lane % CPA
    const int rA = lane / CPA, cA = (lane % CPA) ^ tw_swz<CPA>(rA);        // SPI_A t is a multiple of 4: the swizzle term only sees rA
    const int rD = lane / CPD, cD = (lane % CPD) ^ tw_swz<CPD>(rD);
    // channel part of the source address (fixed for the whole kernel)
    uint32_t a_koff; int a_sub = 0; bool a_kok;
    {
        const int kc = kc0 + cA * VE;
        a_kok = kc < p.KC;
        if constexpr (MODE == TC_CONV) {
            uint32_t phh, r, pww, c;
            p.div_2c.divmod((uint32_t)(a_kok ? kc : 0), phh, r);
            p.div_c.divmod(r, pww, c);
            a_koff = (phh * p.IW * p.C + r) * ESZ; a_sub = (int)(phh * 2 + pww);
        } else a_koff = (uint32_t)kc * ESZ;
    }
    uint32_t d_koff; int d_cls = 0; bool d_kok;
    {
        const int ne = ne0 + cD * VE;
        d_kok = ne < p.NE;
        if constexpr (MODE == TC_CONV) d_koff = (uint32_t)ne * ESZ;
        else {
            uint32_t cls, n;
            p.div_n.divmod((uint32_t)(d_kok ? ne : 0), cls, n);
            d_cls = (int)cls;
            d_koff = ((((cls >> 1) * p.OW + (cls & 1)) * p.N) + n) * ESZ;
        }
    }
    // one DMA instruction of step `step`'s tiles: idx < NIA -> slot tile, else gradient tile.  Called between the MFMA groups
    // of the previous step so that the ~1.4k cycles the per-CU load path needs for a step's 70 KB hide behind the MFMAs.
    auto issue_one = [&](int step, int buf, int idx) {
        const int Ps = Pbeg + step * TW_BP;
        const uint32_t As = lds0 + buf * STAGE, Ds = As + ASTAGE;
        if (dbgm == 3) return;
        if (dbgm == 2) {
            const int t = wave + 8 * (idx < NIA ? idx : idx - NIA);
            if (idx < NIA && t >= ninstrA) return;
            const uint32_t vo = (uint32_t)((Ps * 64 + t * 1024 + lane * 16) & 0xFFFFF);
            if (idx < NIA) dma16_asm(rsA, As + t * 1024, vo); else dma16_asm(rsD, Ds + t * 1024, vo);
            return;
        }
        if (idx < NIA) {
            const int t = wave + 8 * idx;
            if (t >= ninstrA) return;                     // wave-uniform
            const int P = Ps + SPI_A * t + rA;
            const bool ok = P < p.MP && a_kok;
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(ok ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            uint32_t vo;
            if constexpr (MODE == TC_CONV) {
                const int y = 2 * (int)gy + (a_sub >> 1), x = 2 * (int)gx + (a_sub & 1);
                const bool v = ok && y < p.IH && x < p.IW;
                vo = v ? (((b * p.IH + 2 * gy) * p.IW + 2 * gx) * p.C) * ESZ + a_koff : G2_OOB;
            } else {
                const int iy = (int)gy - p.HY, ix = (int)gx - p.HX;
                const bool v = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
                vo = v ? (((b * p.IH + iy) * p.IW + ix) * p.C) * ESZ + a_koff : G2_OOB;
            }
            dma16_asm(rsA, As + t * 1024, vo);
        } else {
            const int t = wave + 8 * (idx - NIA);
            const int P = Ps + SPI_D * t + rD;
            const bool ok = P < Pend && d_kok;
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(ok ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            uint32_t vo;
            if constexpr (MODE == TC_CONV) {
                const bool v = ok && (int)gy < p.OH && (int)gx < p.OW;
                vo = v ? (((b * p.OH + gy) * p.OW + gx) * p.N) * ESZ + d_koff : G2_OOB;
            } else {
                const int oy = 2 * (int)gy + (d_cls >> 1), ox = 2 * (int)gx + (d_cls & 1);
                const bool v = ok && oy < p.OH && ox < p.OW;
                vo = v ? (((b * p.OH + 2 * gy) * p.OW + 2 * gx) * p.N) * ESZ + d_koff : G2_OOB;
            }
            dma16_asm(rsD, Ds + t * 1024, vo);
        }
    };
    constexpr int NDMA = NIA + NID, NKS = SPLIT ? TW_BP / 32 : TW_BP / 16;   // k16-steps a wave runs per position step
    // ---------------- LDEC: one row per lane (see the kernel's header) ----------------
    constexpr int RA = NIA * SPI_A, RD = NID * SPI_D;     // rows this wave requests per step: slot tile, gradient tile
    static_assert(!LDEC || RA + RD <= 64, "lane-parallel row decode: one lane per requested row");
    const bool lpA = lane < RA;
    const int lpj = lpA ? lane : lane - RA;
    const int lpt = wave + 8 * (lpA ? lpj / SPI_A : lpj / SPI_D);
    const int lprow = lpA ? SPI_A * lpt + lpj % SPI_A : SPI_D * lpt + lpj % SPI_D;
    const bool lplive = lane < RA + RD && (!lpA || lpt < ninstrA);
    auto decode_rows = [&](int step) -> uint32_t {        // this lane's row of step `step`: byte offset of its pixel (| edge bits) or G2_OOB
        const int P = Pbeg + step * TW_BP + lprow;
        const bool ok = lplive && P < (lpA ? p.MP : Pend);
        uint32_t g, gx, b, gy;
        p.div_gw.divmod((uint32_t)(ok ? P : 0), g, gx);
        p.div_g.divmod(g, b, gy);
        if constexpr (MODE == TC_CONV) {                  // slot rows: the 2 x 2 pixel block at (2 gy, 2 gx); gradient rows: pixel (gy, gx)
            const int yy = lpA ? 2 * (int)gy : (int)gy, xx = lpA ? 2 * (int)gx : (int)gx;
            const int H = lpA ? p.IH : p.OH, W = lpA ? p.IW : p.OW, CB = (lpA ? p.C : p.N) * ESZ;
            const bool v = ok && yy < H && xx < W;
            const uint32_t edge = lpA ? (uint32_t)(yy + 1 >= H) | ((uint32_t)(xx + 1 >= W) << 1) : 0u;
            return v ? (uint32_t)(((int)b * H + yy) * W + xx) * (uint32_t)CB | edge : G2_OOB;
        } else {                                          // slot rows: pixel (gy - HY, gx - HX); gradient rows: the 2 x 2 output block at (2 gy, 2 gx)
            const int yy = lpA ? (int)gy - p.HY : 2 * (int)gy, xx = lpA ? (int)gx - p.HX : 2 * (int)gx;
            const int H = lpA ? p.IH : p.OH, W = lpA ? p.IW : p.OW, CB = (lpA ? p.C : p.N) * ESZ;
            const bool v = ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const uint32_t edge = lpA ? 0u : (uint32_t)(yy + 1 >= H) | ((uint32_t)(xx + 1 >= W) << 1);
            return v ? (uint32_t)(((int)b * H + yy) * W + xx) * (uint32_t)CB | edge : G2_OOB;
        }
    };
    // which odd row / column this lane's chunk needs (tested against the row's edge bits)
    const uint32_t a_edge = MODE == TC_CONV ? (uint32_t)((a_sub >> 1) | ((a_sub & 1) << 1)) : 0u;
    const uint32_t d_edge = MODE == TC_CONV ? 0u : (uint32_t)((d_cls >> 1) | ((d_cls & 1) << 1));
    uint32_t lp_offs = 0;                                 // the decoded rows of the step whose loads are issued next
    auto fetch_row = [&](int idx) -> uint32_t {           // the row offset DMA instruction idx of that step needs in this lane
        const int src = idx < NIA ? idx * SPI_A + rA : RA + (idx - NIA) * SPI_D + rD;
        return (uint32_t)__builtin_amdgcn_ds_bpermute(src * 4, (int)lp_offs);
    };
    auto issue_row = [&](int buf, int idx, uint32_t row) {
        const uint32_t As = lds0 + buf * STAGE, Ds = As + ASTAGE;
        if (idx < NIA) {
            const int t = wave + 8 * idx;
            if (t >= ninstrA) return;                     // wave-uniform
            const bool bad = (row & a_edge) != 0u || !a_kok;
            dma16_asm(rsA, As + t * 1024, bad ? G2_OOB : (row & ~3u) + a_koff);
        } else {
            const int t = wave + 8 * (idx - NIA);
            const bool bad = (row & d_edge) != 0u || !d_kok;
            dma16_asm(rsD, Ds + t * 1024, bad ? G2_OOB : (row & ~3u) + d_koff);
        }
    };
    uint32_t lp_nn = 0;                                   // ... and of the step after that: decoded at the TOP of a step, while the step's first fragment reads are in flight

    // ---------------- this wave's (tap, output tile) pairs and their per-lane transpose-read offsets ----------------
    // transpose read (see tr_fragment in wgrad_tile.hpp): lane l supplies row r0 + (l>>5)*8 + ((l&15)>>2) (+4 for the high half),
    // element column e0 + ((l>>4)&1)*16 + (l&3)*4, and receives 8 consecutive rows of column e0 + (l & 31).
    const int trow0 = (lane >> 5) * 8 + ((lane & 15) >> 2);
    const int tcol = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    int pr_tap[PPW], pr_nt[PPW];
    bool pr_on[PPW], pr_bias[PPW];
    uint32_t aoff[PPW][KT], doff[PPW];
    const int half = SPLIT ? (wave & 1) : 0;             // SPLIT: positions 64 half .. 64 half + 63 of every step
    const int trow = trow0 + 64 * half;
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        if constexpr (SPLIT) {
            pr_on[q] = true; pr_tap[q] = wave >> 1; pr_nt[q] = q;
            pr_bias[q] = q == 0 && p.dbias && kb == 0 && (wave >> 1) < 2;      // split layout: the waves of taps 0 and 1 sum output tile 0 and 1 (one extra MFMA each)
        } else {
            const int pi = wave + 8 * q;
            pr_on[q] = pi < p.npairs;
            pr_tap[q] = pr_on[q] ? p.pair_tap[pi] : 0;
            pr_nt[q] = pr_on[q] ? p.pair_nt[pi] : 0;
            pr_bias[q] = pr_on[q] && p.dbias && kb == 0 && p.pair_first[pi];
        }
        const int ta = pr_tap[q] / TAPS, tb = pr_tap[q] % TAPS;
        const int row = trow + ta * p.GW + tb;            // + 16 per k-step and + 4 for the high half keep (row & 3)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int col = kt * 32 + tcol;
            aoff[q][kt] = (uint32_t)(row * PA + ((((col >> 3) ^ tw_swz<CPA>(row))) << 4) + (col & 7) * ESZ);
        }
        const int col = pr_nt[q] * 32 + tcol;
        doff[q] = (uint32_t)(ASTAGE + trow * PD + ((((col >> 3) ^ tw_swz<CPD>(trow))) << 4) + (col & 7) * ESZ);
    }

    f32x16 acc[PPW][KT];
#pragma unroll
    for (int q = 0; q < PPW; ++q)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][kt][r] = 0.f;

    // fused bias gradient: column sums of the gradient tile = (all-ones rows) x D on the matrix core, one extra MFMA per
    // k-step in the first pair of every output tile of the kb == 0 blocks; row 0 of the result carries the sums
    // (pair layout: the first pair of an output tile is one of pairs 0 .. NTB-1, i.e. slot q = 0 of a wave; split layout: the waves of tap t < 2
    //  take output tile t.  Accumulating ones x D for EVERY slot cost the 9-tap config 64 registers -- no room to pipeline the fragment
    //  reads -- and made its bias waves issue 12 MFMAs per k-step against 8 in the other four waves.)
    constexpr int NBQ = 1;
    const int bias_nt = SPLIT ? (wave >> 1) & 1 : 0;     // split layout: which of the two gradient fragments of a k-step this wave sums
    f32x16 accb[NBQ];
#pragma unroll
    for (int q = 0; q < NBQ; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[q][r] = 0.f;
    bool any_bias = false;
#pragma unroll
    for (int q = 0; q < NBQ; ++q) any_bias = any_bias || pr_bias[q];
    const u16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

    typedef __attribute__((address_space(3))) s16x4* lds_v4;
    if constexpr (LDEC) {
        lp_offs = decode_rows(0);
#pragma unroll
        for (int i = 0; i < NDMA; ++i) issue_row(0, i, fetch_row(i));
        lp_offs = decode_rows(1);                         // (past the last step: nothing is issued from it)
    } else {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) issue_one(0, 0, i);
    }
    TW_STAMP();
    // One "unit" = the fragments one group of MFMAs needs: (k16-step, pair) in the pair layout (1 gradient + KT slot fragments,
    // KT MFMAs), a whole k16-step in the split layout (NTB gradient + KT slot fragments, NTB KT MFMAs).  The fragments of unit
    // u + 1 are read from LDS BEFORE the MFMAs of unit u are issued (explicit register double buffering): with two waves per
    // SIMD nothing else hides the ~100+ cycles of a transpose read.
    constexpr int ND = SPLIT ? PPW : 1;
    constexpr int NU = SPLIT ? NKS : NKS * PPW;
    constexpr int NKI = NKS;                              // k16-steps that carry DMA issue (bunching them into the first half: measured 8 % slower)
    constexpr bool PIPE = true;
    struct Frag { u16x8 d[ND]; u16x8 a[KT]; };
    auto tr_read = [&](uint32_t off, int pitch) -> u16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + off));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + off + 4 * pitch));
        u16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[e] = (unsigned short)lo[e]; f[4 + e] = (unsigned short)hi[e]; }
        return f;
    };
    auto load_unit = [&](Frag& f, uint32_t sbase, int u) {
        const int ks = SPLIT ? u : u / PPW, q = SPLIT ? 0 : u % PPW;
#pragma unroll
        for (int j = 0; j < ND; ++j) f.d[j] = tr_read(sbase + doff[SPLIT ? j : q] + ks * 16 * PD, PD);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) f.a[kt] = tr_read(sbase + aoff[q][kt] + ks * 16 * PA, PA);
    };
    auto run_steps = [&](auto bias_c) {
        constexpr bool BIAS = decltype(bias_c)::value;    // wave-uniform and constant over the kernel: two copies of the loop, no branch in it
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the step's tiles has landed ...
            __syncthreads();                                  // ... and so has everybody else's; the other stage is free
            if (step < 4) TW_STAMP();
            const bool more = step + 1 < nsteps;
            const uint32_t sbase = (uint32_t)(cur * STAGE);
            Frag fr[2];
            if constexpr (PIPE) load_unit(fr[0], sbase, 0);
            if constexpr (LDEC) {                             // in the latency shadow of the reads just issued (behind the last MFMAs of a step the same ~300 cycles delayed the
                lp_nn = decode_rows(step + 2);                // barrier: +2.7 % on the ConvVAE step, measured)
                asm volatile("" : "+v"(lp_nn));
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int ks = SPLIT ? u : u / PPW, q0 = SPLIT ? 0 : u % PPW;
                if constexpr (LDEC) {
                    if (more && (SPLIT || q0 == 0) && ks < NKI) {   // next step's loads: the row offsets come out of lp_offs (one ds_bpermute_b32 each, under the previous group's MFMAs)
                        uint32_t rows[(NDMA + NKI - 1) / NKI];
#pragma unroll
                        for (int i = ks, g = 0; i < NDMA; i += NKI, ++g) rows[g] = fetch_row(i);
#pragma unroll
                        for (int i = ks, g = 0; i < NDMA; i += NKI, ++g) issue_row(cur ^ 1, i, rows[g]);
                    }
                } else
                if (more && (SPLIT || q0 == 0) && ks < NKI) { // next step's loads, a few per k16-step
#pragma unroll
                    for (int i = ks; i < NDMA; i += NKI) issue_one(step + 1, cur ^ 1, i);
                }
                if (dbgm == 4) continue;                      // (timing instantiation: loads and barriers only)
                if constexpr (!PIPE) load_unit(fr[u & 1], sbase, u);
                else if (u + 1 < NU) load_unit(fr[(u + 1) & 1], sbase, u + 1);
                const Frag& f = fr[u & 1];
#pragma unroll
                for (int j = 0; j < ND; ++j) {
                    const int q = SPLIT ? j : q0;
                    // no branch on pr_on: an unused pair slot recomputes pair 0 and is dropped
                    if constexpr (BIAS) {
                        if (q == 0) {
                            u16x8 dsel = f.d[j];
                            if constexpr (SPLIT) dsel = bias_nt ? f.d[1] : f.d[0];
                            accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, dsel), accb[0], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        // orientation: the 32 lanes of a register are 32 CONTIGUOUS elements of dW (coalesced 128-byte stores into the slabs):
                        // conv form (HWIO): rows = channels, col = output; gather form ([kh,kw,N,C]): rows = outputs, col = channel.
                        // (the transposed choice gives 16-byte stores per lane but at a 256..1024-byte stride: measured 40 % slower on conv2)
                        if constexpr (MODE == TC_CONV)
                            acc[q][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[kt]), __builtin_bit_cast(bf16x8, f.d[j]), acc[q][kt], 0, 0, 0);
                        else
                            acc[q][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.d[j]), __builtin_bit_cast(bf16x8, f.a[kt]), acc[q][kt], 0, 0, 0);
                    }
                }
                if constexpr (LDEC) {
                    if (u == NU - 1) lp_offs = lp_nn;          // every load of step + 1 is out: the next step issues from the rows of step + 2
                }
            }
        }
    };
    if (any_bias) run_steps(std::true_type{}); else run_steps(std::false_type{});
    TW_STAMP();
    const int lcol = lane & 31, lgrp = lane >> 5;
#pragma unroll
    for (int q = 0; q < NBQ; ++q) {                       // bias gradient: row 0 of (ones x D) = register 0 of lanes 0..31
        if (!pr_bias[q] || lane >= 32) continue;
        const int ne = ne0 + (SPLIT ? bias_nt : pr_nt[q]) * 32 + lane;
        if (ne >= p.NE) continue;
        int n = ne;
        if constexpr (MODE == TC_GATHER) n = ne - (int)p.div_n.div((uint32_t)ne) * p.N;
        if (p.bias_part) p.bias_part[((long long)bx * p.bias_nh + (SPLIT ? half : 0)) * p.NE + ne] = accb[q][0];
        else atomicAdd(&p.dbias[n], accb[q][0]);
    }
    // ---------------- dW: register r of a lane is row (r&3) + 8(r>>2) + 4 lgrp, column lcol; the row -> kernel-tap decode is done once
    // per group of 4 registers (4 consecutive rows share it: C and N are multiples of 4) ----------------
    // with caller scratch: partial sums of this position split go to its slab in ACCUMULATOR order, [block column][pair][kt][row group]
    // [lane][4 rows] (one contiguous 1 KiB store per wave-instruction); reduce_tiled_kernel sums the slabs and does the decode.
    // Without scratch: fp32 atomics straight into dW.
    auto emit = [&](const f32x16 (&tiles)[KT], int tap, int nt, int pi) {
        if (dbgm == 5) return;
        if (p.slabs) {
            const long long eoff = (long long)bx * p.slab_stride + ((long long)(by * p.npairs + pi) * KT) * 1024 + lane * 4;
            if (p.slab_bf16) {
                bf16_t* const dst = (bf16_t*)p.slabs + eoff;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float v[4] = {tiles[kt][4 * g4], tiles[kt][4 * g4 + 1], tiles[kt][4 * g4 + 2], tiles[kt][4 * g4 + 3]};
                        TW_SLAB_STORE(__builtin_bit_cast(tw_u32x2, pack4<bf16_t>(v)), (tw_u32x2*)(dst + (kt * 4 + g4) * 256));
                    }
                return;
            }
            float* const dst = p.slabs + eoff;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 v = {tiles[kt][4 * g4], tiles[kt][4 * g4 + 1], tiles[kt][4 * g4 + 2], tiles[kt][4 * g4 + 3]};
                    __builtin_nontemporal_store(v, (f32x4*)(dst + (kt * 4 + g4) * 256));      // written once, read once by the reduce: streaming
                }
            return;
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                long long base, stride;
                const bool ok = tw_dw_index<MODE, TAPS>(p, kc0, ne0, tap, kt, nt, 8 * g4 + 4 * lgrp, lcol, base, stride);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (ok) atomicAdd(p.out + base + t * stride, tiles[kt][4 * g4 + t]);
            }
        }
    };
    if constexpr (SPLIT) {
        // the two position halves of a tap meet in LDS: half h keeps output tile h and hands the other one to its partner wave
        constexpr int WSZ = KT * 16 * 64;                 // floats per wave (16 KB at KT = 4)
        static_assert(8 * WSZ * 4 <= 2 * STAGE, "reduction scratch");
        __syncthreads();                                  // the stage tiles are dead
        TW_EPI_STAMP();
        float* const red = (float*)lds;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave * WSZ + (kt * 16 + r) * 64 + lane] = half ? acc[0][kt][r] : acc[1][kt][r];
        TW_EPI_STAMP();
        __syncthreads();
        TW_EPI_STAMP();
        f32x16 fin[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[kt][r] = (half ? acc[1][kt][r] : acc[0][kt][r]) + red[(wave ^ 1) * WSZ + (kt * 16 + r) * 64 + lane];
        TW_EPI_STAMP();
        emit(fin, wave >> 1, half, (wave >> 1) * 2 + half);
    } else {
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            if (!pr_on[q]) continue;
            emit(acc[q], pr_tap[q], pr_nt[q], wave + 8 * q);
        }
    }
    TW_STAMP();
#undef TW_STAMP
#undef TW_EPI_STAMP
}

// =====================================================================================================================
// tapwgrad_cw_kernel — the k = 5 gather-form filter gradient (3 x 3 slot taps, C = 64 channels, 4 parity classes x 32 outputs) with a wave per
// (parity class, tap ROW) instead of per (tap, output tile) pair.
//
// tapwgrad_kernel's step for this layer is bound by LDS reads: every (tap, class) pair re-reads its gradient fragment and both slot fragments
// (six ds_read_b64_tr_b16 per two MFMAs; 1,536 per 128-position step = 6.1k LDS-pipe cycles against 4.1k MFMA cycles per SIMD).  Here a wave keeps
// the live taps of ONE tap row of its class for both 32-channel tiles (3 x 2 or 2 x 2 accumulator tiles), reads the class's gradient fragment
// once per k-step, and reads only the ALIGNED (tb = 0) slot fragments of its row: the tb = 1, 2 operands are the same 8 positions shifted by one /
// two, formed in registers from this k-step's fragment and the next one's first dword (one v_perm_b32 per dword / a dword move).  For that the
// positions are mapped so that a lane's k-step j + 1 continues its k-step j: lane half g walks positions 64 g + 8 j .. + 7 (a sum over
// positions does not care about its order).  Per k-step a wave issues 2 + 4 transpose reads for 4 or 6 MFMAs: 480 per block step -> MFMA-bound.
// The live (class, tap row) units of k = 5 are 3 + 3 + 2 + 2 = TEN: ten waves per block, dealt to the SIMDs (wave w runs on SIMD w % 4) so that
// with the four BiasAddGrad MFMAs (ones x D, one wave per class) every SIMD issues 14 / 14 / 13 / 13 MFMAs per k-step (the pair layout: 16).
// Staging (LDS-DMA of the slot range and the gradient rows, two stages), the slab layout and the reduce are tapwgrad_kernel's.
// =====================================================================================================================
constexpr int TWC_NW = 10, TWC_NT = TWC_NW * 64;
template <bool LDEC, bool DBG = false>                    // LDEC: the rows of a step decoded once per wave, one row per lane (tapwgrad_kernel's header); DBG: timing instantiation
__global__ __launch_bounds__(TWC_NT) void tapwgrad_cw_kernel(const TapWgradParams p) {
    const int dbgm = DBG ? p.dbg_cheap_addr : 0;
    constexpr int TAPS = 3, KT = 2, NTB = 4;
    constexpr int ESZ = 2, VE = 8;
    constexpr int KCB = 32 * KT, NEB = 32 * NTB;
    constexpr int PA = KCB * ESZ, PD = NEB * ESZ;        // 128 | 256 B
    constexpr int CPA = PA / 16, CPD = PD / 16;
    constexpr int SPI_A = 1024 / PA, SPI_D = 1024 / PD;
    constexpr int MAXSLOT = TW_BP + TC_MAXHALO;
    constexpr int NINSD = TW_BP / SPI_D;                  // gradient-tile DMA instructions per step (32)
    constexpr int NIA = (MAXSLOT / SPI_A + TWC_NW - 1) / TWC_NW, NID = (NINSD + TWC_NW - 1) / TWC_NW;
    constexpr int ASTAGE = MAXSLOT * PA, DSTAGE = TW_BP * PD, STAGE = ASTAGE + DSTAGE;
    static_assert(2 * STAGE + 64 <= 160 * 1024, "tile config");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE + 64];          // (+ 64: the last k-step's look-ahead read may run past the gradient tile)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bslot = (int)blockIdx.x >> 3;
    const int by = bslot % p.gy, bx = (bslot / p.gy) * 8 + ((int)blockIdx.x & 7);
    if (bx >= p.gx) return;
    const int kb = by % p.nkb, nb = by / p.nkb;
    const int kc0 = kb * KCB, ne0 = nb * NEB;
    const int Pbeg = bx * p.pos_per_split;
    const int Pend = min(p.MP, Pbeg + p.pos_per_split);
    if (Pbeg >= Pend) return;
    const int nsteps = (Pend - Pbeg + TW_BP - 1) / TW_BP;
    const int halo = (TAPS - 1) * p.GW + TAPS - 1;
    const int ninstrA = (TW_BP + halo + SPI_A - 1) / SPI_A;
    const u32x4 rsA = make_srd(p.a, p.a_bytes), rsD = make_srd(p.d, p.d_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

    // ---------------- DMA roles (as tapwgrad_kernel, gather form; instruction t = wave + 10 i) ----------------
    const int rA = lane / CPA, cA = (lane % CPA) ^ tw_swz<CPA>(rA);
    const int rD = lane / CPD, cD = (lane % CPD) ^ tw_swz<CPD>(rD);
    const bool a_kok = kc0 + cA * VE < p.KC;
    const uint32_t a_koff = (uint32_t)(kc0 + cA * VE) * ESZ;
    uint32_t d_koff; int d_cls; bool d_kok;
    {
        const int ne = ne0 + cD * VE;
        d_kok = ne < p.NE;
        uint32_t c4, n;
        p.div_n.divmod((uint32_t)(d_kok ? ne : 0), c4, n);
        d_cls = (int)c4;
        d_koff = ((((c4 >> 1) * p.OW + (c4 & 1)) * p.N) + n) * ESZ;
    }
    auto issue_one = [&](int step, int buf, int idx) {
        const int Ps = Pbeg + step * TW_BP;
        const uint32_t As = lds0 + buf * STAGE, Ds = As + ASTAGE;
        if (dbgm == 3) return;
        if (dbgm == 2) {                                  // (timing instantiation: every load inside one megabyte)
            const int t = wave + TWC_NW * (idx < NIA ? idx : idx - NIA);
            if (idx < NIA ? t >= ninstrA : t >= NINSD) return;
            const uint32_t vo = (uint32_t)((Ps * 64 + t * 1024 + lane * 16) & 0xFFFFF);
            if (idx < NIA) dma16_asm(rsA, As + t * 1024, vo); else dma16_asm(rsD, Ds + t * 1024, vo);
            return;
        }
        if (idx < NIA) {
            const int t = wave + TWC_NW * idx;
            if (t >= ninstrA) return;                     // wave-uniform
            const int P = Ps + SPI_A * t + rA;
            const bool ok = P < p.MP && a_kok;
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(ok ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            const int iy = (int)gy - p.HY, ix = (int)gx - p.HX;
            const bool v = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
            dma16_asm(rsA, As + t * 1024, v ? (((b * p.IH + iy) * p.IW + ix) * p.C) * ESZ + a_koff : G2_OOB);
        } else {
            const int t = wave + TWC_NW * (idx - NIA);
            if (t >= NINSD) return;                       // wave-uniform
            const int P = Ps + SPI_D * t + rD;
            const bool ok = P < Pend && d_kok;
            uint32_t g, gx, b, gy;
            p.div_gw.divmod((uint32_t)(ok ? P : 0), g, gx);
            p.div_g.divmod(g, b, gy);
            const int oy = 2 * (int)gy + (d_cls >> 1), ox = 2 * (int)gx + (d_cls & 1);
            const bool v = ok && oy < p.OH && ox < p.OW;
            dma16_asm(rsD, Ds + t * 1024, v ? (((b * p.OH + 2 * gy) * p.OW + 2 * gx) * p.N) * ESZ + d_koff : G2_OOB);
        }
    };
    constexpr int NDMA = NIA + NID, NKS = TW_BP / 16;    // 8 k16-steps per position step; at most one DMA instruction per wave and k-step
    static_assert(NDMA <= NKS, "DMA schedule");
    // ---------------- LDEC: one requested row per lane, decoded a step ahead; a DMA instruction fetches its rows with one ds_bpermute_b32, one k-step ahead ----------------
    constexpr int RA = NIA * SPI_A, RD = NID * SPI_D;     // 24 + 16 rows per wave and step
    static_assert(RA + RD <= 64, "lane-parallel row decode: one lane per requested row");
    const bool lpA = lane < RA;
    const int lpj = lpA ? lane : lane - RA;
    const int lpt = wave + TWC_NW * (lpA ? lpj / SPI_A : lpj / SPI_D);
    const int lprow = lpA ? SPI_A * lpt + lpj % SPI_A : SPI_D * lpt + lpj % SPI_D;
    const bool lplive = lane < RA + RD && (lpA ? lpt < ninstrA : lpt < NINSD);
    auto decode_rows = [&](int step) -> uint32_t {
        const int P = Pbeg + step * TW_BP + lprow;
        const bool ok = lplive && P < (lpA ? p.MP : Pend);
        uint32_t g, gx, b, gy;
        p.div_gw.divmod((uint32_t)(ok ? P : 0), g, gx);
        p.div_g.divmod(g, b, gy);
        const int yy = lpA ? (int)gy - p.HY : 2 * (int)gy, xx = lpA ? (int)gx - p.HX : 2 * (int)gx;
        const int H = lpA ? p.IH : p.OH, W = lpA ? p.IW : p.OW, CB = (lpA ? p.C : p.N) * ESZ;
        const bool v = ok && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const uint32_t edge = lpA ? 0u : (uint32_t)(yy + 1 >= H) | ((uint32_t)(xx + 1 >= W) << 1);      // the 2 x 2 output block's odd row / column outside the image
        return v ? (uint32_t)(((int)b * H + yy) * W + xx) * (uint32_t)CB | edge : G2_OOB;
    };
    const uint32_t d_edge = (uint32_t)((d_cls >> 1) | ((d_cls & 1) << 1));
    uint32_t lp_offs = 0, lp_nn = 0;                     // rows of the step whose loads are issued next / of the one after it
    auto fetch_row = [&](int idx) -> uint32_t {
        const int src = idx < NIA ? idx * SPI_A + rA : RA + (idx - NIA) * SPI_D + rD;
        return (uint32_t)__builtin_amdgcn_ds_bpermute(src * 4, (int)lp_offs);
    };
    auto issue_row = [&](int buf, int idx, uint32_t row) {
        const uint32_t As = lds0 + buf * STAGE, Ds = As + ASTAGE;
        if (idx < NIA) {
            const int t = wave + TWC_NW * idx;
            if (t >= ninstrA) return;                     // wave-uniform
            dma16_asm(rsA, As + t * 1024, a_kok ? row + a_koff : G2_OOB);
        } else {
            const int t = wave + TWC_NW * (idx - NIA);
            if (t >= NINSD) return;                       // wave-uniform
            const bool bad = (row & d_edge) != 0u || !d_kok;
            dma16_asm(rsD, Ds + t * 1024, bad ? G2_OOB : (row & ~3u) + d_koff);
        }
    };

    // ---------------- role: (class, tap row) -- k = 5: class (ph, pw) keeps tap rows ta >= ph and taps tb >= pw ----------------
    //   SIMD 0: w0 (c0, row 2)  w4 (c1, row 0)         w8 (c1, row 1)       6 + 4 + 4       = 14
    //   SIMD 1: w1 (c1, row 2)+ w5 (c3, row 1)+        w9 (c3, row 2)       4+1 + 4+1 + 4   = 14      (+ = carries its class's bias MFMA)
    //   SIMD 2: w2 (c0, row 0)+ w6 (c2, row 1)                              6+1 + 6         = 13
    //   SIMD 3: w3 (c0, row 1)  w7 (c2, row 2)+                             6 + 6+1         = 13
    const int cls = (int)((0x3122310010ull >> (4 * wave)) & 15ull);    // nibble w: class of wave w   {0,1,0,0,1,3,2,2,1,3}
    const int ta = (int)((0x2121101022ull >> (4 * wave)) & 15ull);    //                 tap row      {2,2,0,1,0,1,1,2,1,2}
    const bool bias_wave = (0x0A6 >> wave) & 1;           // waves 1, 2, 5, 7
    const int pw = cls & 1;
    const bool cls_live = ne0 + cls * 32 < p.NE;
    const bool bias_on = p.dbias && kb == 0 && bias_wave && cls_live;

    // transpose-read offsets: lane half g walks positions 64 g + 8 j (+ 0..7): row = 64 (l >> 5) + ((l & 15) >> 2) (+ 4 for the high half), + 8 rows per k-step
    const int trow = 64 * (lane >> 5) + ((lane & 15) >> 2);
    const int tcol = ((lane >> 4) & 1) * 16 + (lane & 3) * 4;
    uint32_t aoff[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int row = trow + ta * p.GW, col = kt * 32 + tcol;
        aoff[kt] = (uint32_t)(row * PA + ((((col >> 3) ^ tw_swz<CPA>(row))) << 4) + (col & 7) * ESZ);
    }
    uint32_t doff;
    {
        const int col = cls * 32 + tcol;
        doff = (uint32_t)(ASTAGE + trow * PD + ((((col >> 3) ^ tw_swz<CPD>(trow))) << 4) + (col & 7) * ESZ);
    }
    const u16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    typedef __attribute__((address_space(3))) s16x4* lds_v4;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4_f __attribute__((ext_vector_type(4)));
    auto tr_half = [&](uint32_t off) -> u32x2_t {         // 4 consecutive positions of this lane's column: two dwords
        return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(lds + off)));
    };
    auto tr_full = [&](uint32_t off, int pitch) -> u32x4_f {
        const u32x2_t lo = tr_half(off), hi = tr_half(off + 4 * pitch);
        return u32x4_f{lo[0], lo[1], hi[0], hi[1]};
    };

    if constexpr (LDEC) {
        lp_offs = decode_rows(0);
#pragma unroll
        for (int i = 0; i < NDMA; ++i) issue_row(0, i, fetch_row(i));
        lp_offs = decode_rows(1);
    } else {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) issue_one(0, 0, i);
    }

    // The position loop and the epilogue, specialised on the first live tap of the row (tb >= TB0): every tap loop is a literal loop
    auto run = [&](auto tb0_c) {
        constexpr int TB0 = decltype(tb0_c)::value;
        f32x16 acc[TAPS][KT], accb;
#pragma unroll
        for (int tb = TB0; tb < TAPS; ++tb)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tb][kt][r] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) accb[r] = 0.f;
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const bool more = step + 1 < nsteps;
            const uint32_t sb = (uint32_t)(cur * STAGE);
            // Fa[kt][j & 1] = the aligned fragment of k-step j, the other slot the next one's; Dq likewise.  The slot a k-step has consumed is
            // refilled (k-step j + 2) right behind the MFMAs that read it: two k-steps of look-ahead with no registers beyond the two slots
            u32x4_f Fa[KT][2], Dq[2];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) { Fa[kt][0] = tr_full(sb + aoff[kt], PA); Fa[kt][1] = tr_full(sb + aoff[kt] + 8 * PA, PA); }
            Dq[0] = tr_full(sb + doff, PD); Dq[1] = tr_full(sb + doff + 8 * PD, PD);
            if constexpr (LDEC) {                             // the rows of step + 2, decoded in the latency shadow of the twelve reads just issued
                lp_nn = decode_rows(step + 2);
                asm volatile("" : "+v"(lp_nn));
            }
#pragma unroll
            for (int j = 0; j < NKS; ++j) {
                if constexpr (LDEC) {
                    if (more && j < NDMA) issue_row(cur ^ 1, j, fetch_row(j));
                } else
                if (more && j < NDMA) issue_one(step + 1, cur ^ 1, j);
                if (dbgm == 4) continue;                      // (timing instantiation: loads and barriers only)
                const bf16x8 dfrag = __builtin_bit_cast(bf16x8, Dq[j & 1]);
                if (bias_on) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), dfrag, accb, 0, 0, 0);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const u32x4_f f0 = Fa[kt][j & 1], f1 = Fa[kt][(j + 1) & 1];
#pragma unroll
                    for (int tb = TB0; tb < TAPS; ++tb) {
                        u32x4_f fr;
                        if (tb == 0) fr = f0;
                        else if (tb == 1)                   // positions + 1: every dword takes its upper half and the next dword's lower half
                            fr = u32x4_f{__builtin_amdgcn_alignbit(f0[1], f0[0], 16), __builtin_amdgcn_alignbit(f0[2], f0[1], 16),
                                         __builtin_amdgcn_alignbit(f0[3], f0[2], 16), __builtin_amdgcn_alignbit(f1[0], f0[3], 16)};
                        else fr = u32x4_f{f0[1], f0[2], f0[3], f1[0]};      // positions + 2: one dword on
                        acc[tb][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dfrag, __builtin_bit_cast(bf16x8, fr), acc[tb][kt], 0, 0, 0);
                    }
                    if (j + 2 < NKS) Fa[kt][j & 1] = tr_full(sb + aoff[kt] + (j + 2) * 8 * PA, PA);
                    else if (j + 2 == NKS) { const u32x2_t lo = tr_half(sb + aoff[kt] + (j + 2) * 8 * PA); Fa[kt][j & 1] = u32x4_f{lo[0], lo[1], 0u, 0u}; }
                }
                if (j + 2 < NKS) Dq[j & 1] = tr_full(sb + doff + (j + 2) * 8 * PD, PD);
                if constexpr (LDEC) {
                    if (j == NKS - 1) lp_offs = lp_nn;       // every load of step + 1 is out
                }
            }
        }
        if (bias_on && lane < 32) {                       // bias gradient: row 0 of (ones x D) = register 0 of lanes 0..31
            const int ne = ne0 + cls * 32 + lane;
            if (ne < p.NE) {
                if (p.bias_part) p.bias_part[(long long)bx * p.bias_nh * p.NE + ne] = accb[0];
                else atomicAdd(&p.dbias[ne - (int)p.div_n.div((uint32_t)ne) * p.N], accb[0]);
            }
        }
        if (!cls_live || dbgm == 5) return;
        // dW tiles: same accumulator order / slab layout as tapwgrad_kernel (pair index looked up in the host's (tap, output tile) list)
#pragma unroll
        for (int tb = TB0; tb < TAPS; ++tb) {
            const int tap = ta * TAPS + tb;
            int pi = -1;
            for (int i = 0; i < p.npairs; ++i) if (p.pair_tap[i] == tap && p.pair_nt[i] == cls) pi = i;
            if (pi < 0) continue;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const f32x16& t = acc[tb][kt];
                const long long eoff = (long long)bx * p.slab_stride + (((long long)(by * p.npairs + pi) * KT) + kt) * 1024 + lane * 4;
                if (p.slab_bf16) {
                    bf16_t* const dst = (bf16_t*)p.slabs + eoff;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float v[4] = {t[4 * g4], t[4 * g4 + 1], t[4 * g4 + 2], t[4 * g4 + 3]};
                        TW_SLAB_STORE(__builtin_bit_cast(tw_u32x2, pack4<bf16_t>(v)), (tw_u32x2*)(dst + g4 * 256));
                    }
                    continue;
                }
                float* const dst = p.slabs + eoff;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) __builtin_nontemporal_store(f32x4{t[4 * g4], t[4 * g4 + 1], t[4 * g4 + 2], t[4 * g4 + 3]}, (f32x4*)(dst + g4 * 256));
            }
        }
    };
    if (pw == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{});
}

// dW += sum over the position-split slabs written in accumulator order by tapwgrad_kernel, in a FIXED order (round 4: the ry slab chains of an element met in
// atomics before).  The unit of work is a PAIR of adjacent 16-byte groups = 8 consecutive slab values (block column, pair, kt, row group, two lanes): one 16-byte
// load per bf16 slab (8-byte accesses run at 0.54-0.70 x the 16-byte rate, MI355X_MICROARCH.md).  A block owns 256 / ry units; the ry threads of a unit take
// every ry-th slab each and meet in LDS behind ONE barrier; thread (value, part) then adds its share of the ry partial sums in slab-lane order (parts combined by
// a fixed shuffle tree), does the filter-index decode and the read-modify-write of dW.  ry: a power of two <= 64, chosen from the layer's shape only
// (conv_ops.hip, reduce_ry) so that small filters still fill the chip.
template <int MODE, int TAPS, int KT, int NTB>
__device__ __forceinline__ void reduce_tiled_body(const TapWgradParams& p, int nslab, int ngroups, int bx, int ry, f32x4* sm) {
    const int upb = 256 / ry;                             // units per block
    const int nunits = ngroups >> 1;                      // (ngroups is a multiple of 256)
    const int ul = (int)threadIdx.x & (upb - 1), by = (int)threadIdx.x / upb;
    const int uid = bx * upb + ul;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
    if (uid < nunits) {
        if (p.slab_bf16) {
            const bf16_t* src = (const bf16_t*)p.slabs + (long long)uid * 8;
#pragma unroll 8
            for (int k = by; k < nslab; k += ry) {
                const u32x4 w = __builtin_nontemporal_load((const u32x4*)(src + k * p.slab_stride));
                s0[0] += __builtin_bit_cast(float, w[0] << 16); s0[1] += __builtin_bit_cast(float, w[0] & 0xffff0000u);
                s0[2] += __builtin_bit_cast(float, w[1] << 16); s0[3] += __builtin_bit_cast(float, w[1] & 0xffff0000u);
                s1[0] += __builtin_bit_cast(float, w[2] << 16); s1[1] += __builtin_bit_cast(float, w[2] & 0xffff0000u);
                s1[2] += __builtin_bit_cast(float, w[3] << 16); s1[3] += __builtin_bit_cast(float, w[3] & 0xffff0000u);
            }
        } else {
            const float* src = p.slabs + (long long)uid * 8;
#pragma unroll 8
            for (int k = by; k < nslab; k += ry) {
                s0 += __builtin_nontemporal_load((const f32x4*)(src + k * p.slab_stride));
                s1 += __builtin_nontemporal_load((const f32x4*)(src + k * p.slab_stride + 4));
            }
        }
    }
    float* const smf = (float*)sm;                        // [by][unit][8]
    sm[2 * threadIdx.x] = s0; sm[2 * threadIdx.x + 1] = s1;
    __syncthreads();
    const int nval = upb * 8;                             // values of this block: 2048 / ry
    auto emit = [&](int vi, float v) {
        const int u2 = vi >> 3, e = vi & 7;
        const int gid = 2 * (bx * upb + u2) + (e >> 2);
        const int lane = gid & 63, g4 = (gid >> 6) & 3;
        const int slot = gid >> 8;                        // (block column * npairs + pair) * KT + kt
        const int kt = slot % KT, bp = slot / KT;
        const int pi = bp % p.npairs, bcol = bp / p.npairs;
        const int kb = bcol % p.nkb, nb = bcol / p.nkb;
        long long base, stride;
        const bool ok = tw_dw_index<MODE, TAPS>(p, kb * 32 * KT, nb * 32 * NTB, p.pair_tap[pi], kt, p.pair_nt[pi], 8 * g4 + 4 * (lane >> 5), lane & 31, base, stride);
        if (ok) p.out[base + (e & 3) * stride] += v;
    };
    if (ry >= 8) {                                        // 256 = nval x P threads: part `part` adds slab lanes 8 part .. 8 part + 7, the P parts meet in a shuffle tree
        const int P = ry >> 3;
        const int part = (int)threadIdx.x & (P - 1), vi = (int)threadIdx.x / P;
        const int u2 = vi >> 3, e = vi & 7;
        float v = smf[((part * 8) * upb + u2) * 8 + e];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += smf[((part * 8 + k) * upb + u2) * 8 + e];
        for (int o = 1; o < P; o <<= 1) v += __shfl_xor(v, o, 64);
        if (part == 0 && bx * upb + u2 < nunits) emit(vi, v);
    } else {
        for (int vi = (int)threadIdx.x; vi < nval; vi += 256) {
            const int u2 = vi >> 3, e = vi & 7;
            if (bx * upb + u2 >= nunits) continue;
            float v = smf[u2 * 8 + e];
            for (int k = 1; k < ry; ++k) v += smf[(k * upb + u2) * 8 + e];
            emit(vi, v);
        }
    }
}
template <int MODE, int TAPS, int KT, int NTB>
__global__ __launch_bounds__(256) void reduce_tiled_kernel(const TapWgradParams p, int nslab, int ngroups, int ry) {
    __shared__ f32x4 sm[512];
    reduce_tiled_body<MODE, TAPS, KT, NTB>(p, nslab, ngroups, (int)blockIdx.x, ry, sm);
}

// One job of an ordered slab sum: out[i] += sum_k slabs[k * stride + i], i < n.  The block is 256 / KL groups of four elements x KL slab lanes (KL: a power of
// two chosen from the job's shape only); the lanes' partial sums meet in LDS (one barrier) and are added in lane order.
__device__ __forceinline__ void sr_job_body(const float* __restrict__ slabs, float* __restrict__ out, long long stride, long long n, int nslab, int KL, int vec_ok,
                                            int local_block, f32x4* sm, int overwrite = 0) {
    const int QPB = 256 / KL;
    const int q = (int)threadIdx.x & (QPB - 1), kl = (int)threadIdx.x / QPB;
    const long long i4 = ((long long)local_block * QPB + q) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i4 < n) {
        if (vec_ok && i4 + 4 <= n) {
#pragma unroll 8
            for (int k = kl; k < nslab; k += KL) s += *(const f32x4*)(slabs + k * stride + i4);
        } else {
            for (int k = kl; k < nslab; k += KL)
                for (int e = 0; e < 4; ++e) if (i4 + e < n) s[e] += slabs[k * stride + i4 + e];
        }
    }
    if (KL > 1) {                                        // (block-uniform)
        sm[threadIdx.x] = s;
        __syncthreads();
        if (KL > 16) {                                   // two levels: lanes kl < 16 add every 16th lane's sum, then lane 0 adds those sixteen
            if (kl < 16) { s = sm[kl * QPB + q]; for (int k = kl + 16; k < KL; k += 16) s += sm[k * QPB + q]; }
            __syncthreads();
            if (kl < 16) sm[kl * QPB + q] = s;
            __syncthreads();
            if (kl == 0) { s = sm[q]; for (int k = 1; k < 16; ++k) s += sm[k * QPB + q]; }
        } else if (kl == 0) for (int k = 1; k < KL; ++k) s += sm[k * QPB + q];
    }
    if (kl == 0 && i4 < n) {
        if (overwrite) {                                  // out = sum (the caller's buffer need not be zeroed)
            if (vec_ok && i4 + 4 <= n) *(f32x4*)(out + i4) = s;
            else for (int e = 0; e < 4; ++e) if (i4 + e < n) out[i4 + e] = s[e];
        } else if (vec_ok && i4 + 4 <= n) { f32x4 o = *(f32x4*)(out + i4); o += s; *(f32x4*)(out + i4) = o; }
        else for (int e = 0; e < 4; ++e) if (i4 + e < n) out[i4 + e] += s[e];
    }
}
constexpr int SR_MAX = 12;
struct SmallReduceParams {
    const float* slabs[SR_MAX]; float* out[SR_MAX]; long long stride[SR_MAX], n[SR_MAX];
    int nslab[SR_MAX], kl[SR_MAX], vec[SR_MAX], first[SR_MAX + 1];
    int njobs;
    unsigned ovw;                                        // bit j: job j stores its sums (out = ...) instead of adding them to out
};
__device__ __forceinline__ void sr_dispatch(const SmallReduceParams& f, int b, f32x4* sm) {
    int j = 0;
#pragma unroll
    for (int i = 1; i < SR_MAX; ++i) j += (i < f.njobs && b >= f.first[i]) ? 1 : 0;
    sr_job_body(f.slabs[j], f.out[j], f.stride[j], f.n[j], f.nslab[j], f.kl[j], f.vec[j], b - f.first[j], sm, (int)((f.ovw >> j) & 1u));
}

// All deferred reductions of one backward pass in ONE launch (six layers: ~190 MB of slabs, 35-40 us at HBM speed against 130 us as six
// latency-bound launches): block b belongs to the layer l with first[l] <= b < first[l + 1]; inside the layer, b - first[l] = the block of 256 / ry groups.
// Blocks from first[n] on: the layers' bias-gradient partial sums as generic ordered jobs (several blocks each: a single block per layer summed up to 2,048
// rows by itself and was the long pole of the launch).
constexpr int TW_MAX_FUSED = 8;
struct FusedReduceParams {
    TapWgradParams q[TW_MAX_FUSED];
    int splits[TW_MAX_FUSED], ngroups[TW_MAX_FUSED], kind[TW_MAX_FUSED], ry[TW_MAX_FUSED], first[TW_MAX_FUSED + 1];
    int n;
    SmallReduceParams bias;
};
__global__ __launch_bounds__(256) void reduce_fused_kernel(const FusedReduceParams f) {
    __shared__ f32x4 sm[512];
    const int b = (int)blockIdx.x;
    if (b >= f.first[f.n]) {                              // (block-uniform)
        sr_dispatch(f.bias, b - f.first[f.n], sm);
        return;
    }
    int l = 0;
#pragma unroll
    for (int i = 1; i < TW_MAX_FUSED; ++i) l += (i < f.n && b >= f.first[i]) ? 1 : 0;
    const int local = b - f.first[l], ry = f.ry[l];
    const TapWgradParams& p = f.q[l];
    if (f.kind[l] == 0) reduce_tiled_body<TC_CONV, 2, 4, 2>(p, f.splits[l], f.ngroups[l], local, ry, sm);
    else if (f.kind[l] == 1) reduce_tiled_body<TC_GATHER, 2, 4, 2>(p, f.splits[l], f.ngroups[l], local, ry, sm);
    else reduce_tiled_body<TC_GATHER, 3, 2, 4>(p, f.splits[l], f.ngroups[l], local, ry, sm);
}

// Several ordered slab sums in ONE launch (round 4: the small reductions at the end of a backward pass -- encoder-head / decoder-tail slabs, the latent layers'
// filter- and bias-gradient splits -- were seven latency-bound launches in a row on the caller's stream).  Job j owns blocks first[j] .. first[j + 1) - 1.
__global__ __launch_bounds__(256) void reduce_small_fused_kernel(const SmallReduceParams f) {
    __shared__ f32x4 sm[256];
    sr_dispatch(f, (int)blockIdx.x, sm);
}

}  // namespace mi
