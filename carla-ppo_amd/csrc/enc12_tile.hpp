// enc12_tile.hpp — the encoder head of a FORWARD pass as ONE kernel (round 5): conv1 (3 -> 32 channels, k4 s2, + bias + ReLU,
// vae/models.py:250) computed straight into LDS and consumed there by conv2 (32 -> 64 channels, k4 s2, + bias + ReLU, vae/models.py:251).
//
// Unfused: conv1 (narrow_conv48_kernel) reads the frames (20 MB as camera bytes at batch 512) and writes its activation (111 MB bf16 + 12.6 MB ReLU bit words); conv2
// (rwconv_conv_kernel) reads those 111 MB back.  Here a block owns a BAND of a frame -- 6 of the 18 output rows of conv2 = 14 rows of conv1's activation -- computes the
// band's conv1 pixels with the loader / MFMA / epilogue of narrow_conv48_kernel (bit for bit the same activation and bit words), keeps them in LDS, and runs conv2 on
// them with its 64 KB of weights in REGISTERS (a wave = one 32-output tile over all of K = 16 taps x 32 channels: 32 fragments = 128 VGPRs, the form of rwconv.hip).
// conv1's activation is still WRITTEN (conv2's filter gradient reads it in the backward pass) -- every pixel by the band that owns it -- but never read back in the
// forward pass: one launch and 100 MB of reads less on the serial forward chain.
//
// LDS: the band's activation as two PLANES (even / odd columns), [plane][row 0 .. 14][column / 2][32 channels] bf16 = 64 bytes per pixel, the 16-byte chunk c of pixel
// index i stored at c ^ ((i >> 2) & 3): conv2's fragment reads (lane = output pixel, consecutive lanes = consecutive output columns = consecutive entries of ONE plane)
// then cover all 64 banks per 16-lane group.  2 x 15 x 40 x 64 = 76,800 bytes: two blocks per CU.
// Band r of a frame: conv2 rows 6 r .. 6 r + 5, conv1 rows 12 r .. 12 r + 13 (band 2 also row 38, which conv2 never reads but the activation tensor holds); rows
// 12 r + 12, 12 r + 13 are computed twice (they are the next band's first two): 10 % more conv1 work, stored by their owner only.
#pragma once
#include "wgrad_tile.hpp"

namespace mi {

typedef uint32_t e12_u32x4 __attribute__((ext_vector_type(4)));

constexpr int E12_PW = 40;                                // pixels per plane row
constexpr int E12_ROWS = 15;
constexpr int E12_PLANE = E12_ROWS * E12_PW * 64;         // bytes
constexpr int E12_LDS = 2 * E12_PLANE;                    // 76,800

struct Enc12Params {
    const void* frames; const int* frame_idx; long long frame_stride;   // [*, 80, 160, 3] camera bytes / fp32, elements per frame
    const bf16_t* w1; const float* b1;                    // conv1: K-contiguous [32][48]
    const bf16_t* w2; const float* b2;                    // conv2: K-contiguous [64][512], k = (kh * 4 + kw) * 32 + c
    const bf16_t* w2f;                                    // optional: conv2's kernel in fragment order (mi_ares_pack_weights form 5): the prologue's 32 loads per wave are then 1 KB contiguous each
    bf16_t* act1; uint32_t* bits1;                        // [B, 39, 79, 32]; ReLU bit words [B * 39 * 79][2] (may be NULL)
    bf16_t* act2;                                         // [B, 18, 38, 64]
    int B, ntiles;                                        // ntiles = 3 B
    int dbg;                                              // ablation mask of the DBG instantiation (tools/enc12_ablate.py; results are WRONG with any bit set): 1 no act1 / bit-word stores, 2 no bit words,
                                                          // 4 no conv2 stage, 8 no act2 stores, 16 no frame loads, 32 no LDS writes of conv1, 64 no LDS reads in conv2, 128 conflict-free (wrong) read addresses in the pipelined conv2
};

// conv2 stage, pipelined form (C2 = 1): the 32 (tap, channel half) fragments of a position tile are read from LDS by hand, NF of them in flight, in the order the MFMAs
// consume them (the compiler sinks every ds_read next to its MFMA: "two reads, s_waitcnt lgkmcnt(1), v_mfma" -- a full LDS latency in front of every other MFMA; measured
// with the reads taken out: 17 of the kernel's 73 us).  LDS returns in order: fragment F has landed when at most (reads issued after it) are outstanding; the fragment
// passes THROUGH the wait statement, so its MFMA cannot move above it.  Buffer (F - 1) % NF is refilled behind MFMA F: its reader issued a whole MFMA earlier.
template <int IMM> __device__ __forceinline__ void e12_lds_read(u16x8& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(d) : "v"(addr), "n"(IMM)); }
template <int N> __device__ __forceinline__ void e12_lds_wait(u16x8& d) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(d) : "n"(N)); }
template <int F> __device__ __forceinline__ void e12_frag_read(u16x8& d, const uint32_t (&base)[2][2]) {      // F = (kh * 4 + kw) * 2 + ks
    constexpr int kh = F >> 3, kw = (F >> 1) & 3, ks = F & 1;
    e12_lds_read<(kw & 1) * E12_PLANE + kh * E12_PW * 64>(d, base[kw >> 1][ks]);
}
template <int F, int NF> struct E12Conv2Pipe {
    static __device__ __forceinline__ void run(u16x8 (&fb)[NF], const u16x8 (&wf2)[32], f32x16& acc, const uint32_t (&base)[2][2]) {
        constexpr int issued = F == 0 ? NF - 1 : (F - 2 + NF < 31 ? F - 2 + NF : 31);      // highest fragment requested so far
        e12_lds_wait<issued - F>(fb[F % NF]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf2[F]), __builtin_bit_cast(bf16x8, fb[F % NF]), acc, 0, 0, 0);
        if constexpr (F >= 1 && F - 1 + NF < 32) e12_frag_read<F - 1 + NF>(fb[(F - 1) % NF], base);
        if constexpr (F + 1 < 32) E12Conv2Pipe<F + 1, NF>::run(fb, wf2, acc, base);
    }
};
template <int F, int NF> struct E12Conv2Prologue {
    static __device__ __forceinline__ void run(u16x8 (&fb)[NF], const uint32_t (&base)[2][2]) {
        e12_frag_read<F>(fb[F], base);
        if constexpr (F + 1 < NF) E12Conv2Prologue<F + 1, NF>::run(fb, base);
    }
};

template <typename TS, int DBG = 0, int RING = 0, int C2 = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void enc12_fwd_kernel(const Enc12Params p) {
    const int dbg = DBG ? p.dbg : 0;                      // (DBG = 0, the product: every `dbg &` test below folds away)
    constexpr int SSZ = (int)sizeof(TS), GSZ = 4 * SSZ, GDW = GSZ / 4;
    constexpr int FW = 160, A1H = 39, A1W = 79, A2H = 18, A2W = 38;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[E12_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 31, lgrp = lane >> 5;

    // ---- per wave, once: conv1's weights (3 fragments), conv2's weights of this wave's 32-output tile (32 fragments), the biases as initial accumulators ----
    const int nt = wave & 1, mh = wave >> 1;              // conv2: output tile (32 of 64 channels), half of the band's 8 position tiles
    u16x8 wf1[3], wf2[32];
    {
        const __amdgpu_buffer_rsrc_t rsW1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, 32 * 48 * 2, 0x00020000);
        const bool w2frag = p.w2f != nullptr;               // (wave-uniform: an address select per load, no branch)
        const __amdgpu_buffer_rsrc_t rsW2 = __builtin_amdgcn_make_buffer_rsrc((void*)(w2frag ? p.w2f : p.w2), 0, 64 * 512 * 2, 0x00020000);
#pragma unroll
        for (int s = 0; s < 3; ++s) wf1[s] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW1, (lrow * 48 + s * 16 + lgrp * 8) * 2, 0, 0));
#pragma unroll
        for (int f = 0; f < 32; ++f)                       // (round 6) p.w2f: conv2's kernel in fragment order (pack form 5), 1 KB contiguous per load instead of 32 rows x 32 B at a 1 KB pitch
            wf2[f] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW2, w2frag ? ((nt * 32 + f) * 64 + lane) * 16 : ((nt * 32 + lrow) * 512 + f * 16 + lgrp * 8) * 2, 0, 0));
    }
    f32x16 acc1_0, acc2_0;                                // register r of a lane = channel (r & 3) + 8 (r >> 2) + 4 lgrp of the 32-channel tile
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        const f32x4 ba = *(const f32x4*)(p.b1 + 8 * qd + 4 * lgrp), bb = *(const f32x4*)(p.b2 + nt * 32 + 8 * qd + 4 * lgrp);
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc1_0[4 * qd + t] = ba[t]; acc2_0[4 * qd + t] = bb[t]; }
    }
    // frame patch of a conv1 pixel (the loader of narrow_conv48_kernel): group j = 2 s + gi of this lane: q = 4 s + gi (+ 2 for the upper half-wave) -> kernel row q / 3,
    // value offset (q % 3) * 4
    const uint32_t rowb = (uint32_t)(FW * 3 * SSZ);
    uint32_t goff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int qa = 4 * (j >> 1) + (j & 1), qb = qa + 2;
        goff[j] = lgrp ? (uint32_t)(qb / 3) * rowb + (uint32_t)((qb % 3) * GSZ) : (uint32_t)(qa / 3) * rowb + (uint32_t)((qa % 3) * GSZ);
    }
    struct Raw { uint32_t d[6][GDW]; };

    const int G = (int)gridDim.x;
    // a band: frame b, band index, first conv1 row, rows / pixels / 32-pixel groups, and where its frame starts
    struct Band { int b, band, y0, npix, own_rows, ngrp; const unsigned char* fbase; };
    auto band_of = [&](int tile) -> Band {
        Band q;
        q.b = tile / 3; q.band = tile - 3 * q.b; q.y0 = 12 * q.band;
        const int nrows = q.band == 2 ? 15 : 14;
        q.npix = nrows * A1W; q.own_rows = q.band == 2 ? 15 : 12;      // rows this band stores to HBM (the rest belong to the next band)
        q.ngrp = (q.npix + 31) >> 5;
        long long fr = q.b;
        if constexpr (RING && SSZ == 1) {                  // (ring form: a SCALAR load by hand -- a vector load here would be the one load the compiler sees next to the stores, and it
            if (p.frame_idx) {                             // answers its use with s_waitcnt vmcnt(0): every request in flight drained at the head of every band)
                int v;
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p.frame_idx + q.b));
                fr = v;
            }
        } else if (p.frame_idx) fr = p.frame_idx[q.b];
        q.fbase = (const unsigned char*)p.frames + fr * p.frame_stride * SSZ;
        return q;
    };
    auto request3 = [&](const unsigned char* fbase, int y0r, int npixr, int grp, Raw& r) {
        const int pq = min(grp * 32 + lrow, npixr - 1);    // pixels past the band recompute its last one; their stores are skipped
        const int row = pq / A1W, col = pq - row * A1W;
        const unsigned char* pix = fbase + (2 * (y0r + row) * FW + 2 * col) * 3 * SSZ;
        if constexpr (RING && SSZ == 1) {
            // ring form on camera bytes: the six loads are INLINE ASSEMBLY, i.e. invisible to the compiler's s_waitcnt placement.  On gfx9-family targets loads and stores share
            // vmcnt and may retire out of order with respect to each other, so LLVM answers every use of a loaded register with s_waitcnt vmcnt(0) while ANY store is
            // pending (SIInsertWaitcnts: mixed pending events) -- in a loop that stores what it computes that is every step: no load is ever in flight across a step.
            // The wait is written by hand instead (ring_wait: vmcnt(6 x the younger requests); loads retire in order among themselves and younger STORES are not counted,
            // which errs on the waiting side whatever order they retire in).  Nothing may read r between this request and its ring_wait (no copies, no spills: the build's
            // resource line must say 0 spills for this instantiation; tools/check_enc12_isa.py reads the listing).
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (dbg & 16) { r.d[j][0] = 0x40404040u + (uint32_t)pq; continue; }
                asm volatile("global_load_dword %0, %1, off" : "=&v"(r.d[j][0]) : "v"(pix + goff[j]));
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (dbg & 16) {
#pragma unroll
                for (int e = 0; e < GDW; ++e) r.d[j][e] = 0x40404040u + (uint32_t)pq;
                continue;
            }
            const PackU<uint32_t, GDW, 2> v = *(const PackU<uint32_t, GDW, 2>*)(pix + goff[j]);
#pragma unroll
            for (int e = 0; e < GDW; ++e) r.d[j][e] = v.v[e];
        }
    };
    auto request = [&](const Band& q, int grp, Raw& r) { request3(q.fbase, q.y0, q.npix, grp, r); };
    // the hand-written wait of the ring form: the request of `r` is complete when at most 6 x (NB - 1) younger loads are outstanding; r's registers pass THROUGH the
    // statement, so no use of them can be scheduled above it
    auto ring_wait = [&](Raw& r) {
        if constexpr (RING && SSZ == 1)
            asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r.d[0][0]), "+v"(r.d[1][0]), "+v"(r.d[2][0]), "+v"(r.d[3][0]), "+v"(r.d[4][0]), "+v"(r.d[5][0]) : "n"(6 * ((SSZ == 1 ? 3 : 2) - 1)));
    };
    // (camera bytes: two steps ahead, 6 registers per step; fp32 frames hold 24 per step: one step ahead keeps the kernel at two waves per SIMD without spills)
    constexpr bool DEEP = SSZ == 1;
    constexpr int NB = DEEP ? 3 : 2;                      // ring form: request buffers = steps a load is ahead of its use
    Raw r0, r1, ra, rb, rc;
    int tile = (int)blockIdx.x;
    if (tile >= p.ntiles) return;                          // (block-uniform)
    Band cb = band_of(tile);
    if constexpr (RING) {
        request(cb, wave, ra); request(cb, wave + 4, rb);
        if constexpr (NB == 3) request(cb, wave + 8, rc);
    } else {
        request(cb, min(wave, cb.ngrp - 1), r0);
        if constexpr (DEEP) request(cb, min(wave + 4, cb.ngrp - 1), r1);
    }
    for (; tile < p.ntiles; tile += G) {
        const int b = cb.b, band = cb.band, y0 = cb.y0, npix = cb.npix, own_rows = cb.own_rows, ngrp = cb.ngrp;

        // ================= conv1: the band's pixels, 32 per wave step (groups wave, wave + 4, ...), loads two steps ahead (the first ones were requested before the
        // previous band's conv2 stage) =================
        if constexpr (RING) {
            // ---- ring form (round 5, late): NB statically named request buffers, step k of this wave (group wave + 4 k) consumes buffer k % NB and re-requests it for step
            // k + NB right after its conversion -- no register rotation (the `cur = r0; r0 = r1` copies made every step wait for the loads issued ONE step earlier, and the
            // conditional request left an s_waitcnt vmcnt(0) behind it: no prefetch at all in the generated code), no conditional requests (every wave runs a multiple of NB
            // steps; steps past the band recompute its last pixel and store nothing; requests past the band are the NEXT band's first steps, so the stream of loads never stops)
            const int nsteps = (ngrp - wave + 3) >> 2, nsp = (nsteps + NB - 1) / NB * NB;      // (wave-uniform)
            const bool more = tile + G < p.ntiles;
            const Band nb = more ? band_of(tile + G) : cb;
            auto step = [&](Raw& cur, int k) {
                const int grp = wave + 4 * k;
                ring_wait(cur);
                u16x8 xf[3];
    #pragma unroll
                for (int j = 0; j < 6; ++j) {
                    uint32_t* dst = (uint32_t*)&xf[j >> 1] + 2 * (j & 1);
                    float f[4];
                    if constexpr (SSZ == 1) {                  // camera bytes: k * (1 / 255), which rounds to the same bf16 as the exact quotient (common.hpp)
    #pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] = (float)((cur.d[j][0] >> (8 * e)) & 255u) * U8_RCP255;
                    } else {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] = __builtin_bit_cast(float, cur.d[j][e < GDW ? e : 0]);
                    }
                    const PackN<uint32_t, 2> h = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(f));
                    dst[0] = h.v[0]; dst[1] = h.v[1];
                }
                {   // (selects, not a branch: both sides issue the same six loads)
                    const int j = k + NB; const bool nx = j >= nsp;
                    request3(nx ? nb.fbase : cb.fbase, nx ? nb.y0 : y0, nx ? nb.npix : npix, wave + 4 * (nx ? j - nsp : j), cur);
                }
                f32x16 acc = acc1_0;
    #pragma unroll
                for (int s = 0; s < 3; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf1[s]), __builtin_bit_cast(bf16x8, xf[s]), acc, 0, 0, 0);
                // epilogue of narrow_conv48_kernel<., 0>: bf16, ReLU, half-wave exchange -> lane (pixel, g) owns channels 16 g .. 16 g + 15 as 8 dwords
                uint32_t R[4][2];
    #pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float v[4] = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
                    const PackN<uint32_t, 2> w = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(v));
    #pragma unroll
                    for (int d = 0; d < 2; ++d) { uint32_t u; asm("v_pk_max_i16 %0, %1, 0" : "=v"(u) : "v"(w.v[d])); R[qd][d] = u; }
                }
    #pragma unroll
                for (int d = 0; d < 2; ++d) {
                    auto s0 = __builtin_amdgcn_permlane32_swap(R[0][d], R[2][d], false, false); R[0][d] = s0[0]; R[2][d] = s0[1];
                    auto s1 = __builtin_amdgcn_permlane32_swap(R[1][d], R[3][d], false, false); R[1][d] = s1[0]; R[3][d] = s1[1];
                }
                const uint32_t o[8] = {R[0][0], R[0][1], R[2][0], R[2][1], R[1][0], R[1][1], R[3][0], R[3][1]};       // dword d: channels 16 g + 2 d, + 1
                const int pq = grp * 32 + lrow;
                if (pq < npix) {
                    const int row = pq / A1W, col = pq - row * A1W;
                    // LDS: plane col & 1, entry row * 40 + col / 2, chunks 2 g and 2 g + 1 swizzled by the entry index
                    const int idx = col >> 1;
                    unsigned char* q = lds + (col & 1) * E12_PLANE + (row * E12_PW + idx) * 64;
                    const int sw = (idx >> 2) & 3;
                    if (!(dbg & 32)) {
                        *(e12_u32x4*)(q + (((2 * lgrp) ^ sw) << 4)) = e12_u32x4{o[0], o[1], o[2], o[3]};
                        *(e12_u32x4*)(q + (((2 * lgrp + 1) ^ sw) << 4)) = e12_u32x4{o[4], o[5], o[6], o[7]};
                    }
                    if (row < own_rows && !(dbg & 1)) {                      // this band owns the pixel: the activation tensor and its ReLU bit words
                        const long long m = ((long long)b * A1H + y0 + row) * A1W + col;
                        unsigned char* g = (unsigned char*)p.act1 + m * 64 + lgrp * 32;
                        *(e12_u32x4*)g = e12_u32x4{o[0], o[1], o[2], o[3]};
                        *(e12_u32x4*)(g + 16) = e12_u32x4{o[4], o[5], o[6], o[7]};
                        if (p.bits1 && !(dbg & 2)) {            // bit d / bit 16 + d of the word: the two halves of dword d are non-zero (post-ReLU: positive)
                            uint32_t word = 0;
    #pragma unroll
                            for (int d = 0; d < 8; ++d) { uint32_t nz; asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(o[d]), "v"(0x00010001u)); word |= nz << d; }
                            p.bits1[m * 2 + lgrp] = word;
                        }
                    }
                }
            };
            for (int k = 0; k < nsp; k += NB) {
                step(ra, k); step(rb, k + 1);
                if constexpr (NB == 3) step(rc, k + 2);
            }
            __syncthreads();                               // the band's activation is in LDS
            if (more) cb = nb;
        } else {
            for (int grp = wave; grp < ngrp; grp += 4) {
                Raw cur = r0;
                if constexpr (DEEP) { r0 = r1; if (grp + 8 < ngrp) request(cb, grp + 8, r1); }      // (wave-uniform branches: nothing is requested past the band)
                else { if (grp + 4 < ngrp) request(cb, grp + 4, r0); }
                u16x8 xf[3];
    #pragma unroll
                for (int j = 0; j < 6; ++j) {
                    uint32_t* dst = (uint32_t*)&xf[j >> 1] + 2 * (j & 1);
                    float f[4];
                    if constexpr (SSZ == 1) {                  // camera bytes: k * (1 / 255), which rounds to the same bf16 as the exact quotient (common.hpp)
    #pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] = (float)((cur.d[j][0] >> (8 * e)) & 255u) * U8_RCP255;
                    } else {
    #pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] = __builtin_bit_cast(float, cur.d[j][e < GDW ? e : 0]);
                    }
                    const PackN<uint32_t, 2> h = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(f));
                    dst[0] = h.v[0]; dst[1] = h.v[1];
                }
                f32x16 acc = acc1_0;
    #pragma unroll
                for (int s = 0; s < 3; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf1[s]), __builtin_bit_cast(bf16x8, xf[s]), acc, 0, 0, 0);
                // epilogue of narrow_conv48_kernel<., 0>: bf16, ReLU, half-wave exchange -> lane (pixel, g) owns channels 16 g .. 16 g + 15 as 8 dwords
                uint32_t R[4][2];
    #pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const float v[4] = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
                    const PackN<uint32_t, 2> w = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(v));
    #pragma unroll
                    for (int d = 0; d < 2; ++d) { uint32_t u; asm("v_pk_max_i16 %0, %1, 0" : "=v"(u) : "v"(w.v[d])); R[qd][d] = u; }
                }
    #pragma unroll
                for (int d = 0; d < 2; ++d) {
                    auto s0 = __builtin_amdgcn_permlane32_swap(R[0][d], R[2][d], false, false); R[0][d] = s0[0]; R[2][d] = s0[1];
                    auto s1 = __builtin_amdgcn_permlane32_swap(R[1][d], R[3][d], false, false); R[1][d] = s1[0]; R[3][d] = s1[1];
                }
                const uint32_t o[8] = {R[0][0], R[0][1], R[2][0], R[2][1], R[1][0], R[1][1], R[3][0], R[3][1]};       // dword d: channels 16 g + 2 d, + 1
                const int pq = grp * 32 + lrow;
                if (pq < npix) {
                    const int row = pq / A1W, col = pq - row * A1W;
                    // LDS: plane col & 1, entry row * 40 + col / 2, chunks 2 g and 2 g + 1 swizzled by the entry index
                    const int idx = col >> 1;
                    unsigned char* q = lds + (col & 1) * E12_PLANE + (row * E12_PW + idx) * 64;
                    const int sw = (idx >> 2) & 3;
                    if (!(dbg & 32)) {
                        *(e12_u32x4*)(q + (((2 * lgrp) ^ sw) << 4)) = e12_u32x4{o[0], o[1], o[2], o[3]};
                        *(e12_u32x4*)(q + (((2 * lgrp + 1) ^ sw) << 4)) = e12_u32x4{o[4], o[5], o[6], o[7]};
                    }
                    if (row < own_rows && !(dbg & 1)) {                      // this band owns the pixel: the activation tensor and its ReLU bit words
                        const long long m = ((long long)b * A1H + y0 + row) * A1W + col;
                        unsigned char* g = (unsigned char*)p.act1 + m * 64 + lgrp * 32;
                        *(e12_u32x4*)g = e12_u32x4{o[0], o[1], o[2], o[3]};
                        *(e12_u32x4*)(g + 16) = e12_u32x4{o[4], o[5], o[6], o[7]};
                        if (p.bits1 && !(dbg & 2)) {            // bit d / bit 16 + d of the word: the two halves of dword d are non-zero (post-ReLU: positive)
                            uint32_t word = 0;
    #pragma unroll
                            for (int d = 0; d < 8; ++d) { uint32_t nz; asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(o[d]), "v"(0x00010001u)); word |= nz << d; }
                            p.bits1[m * 2 + lgrp] = word;
                        }
                    }
                }
            }
            __syncthreads();                                   // the band's activation is in LDS
            if (tile + G < p.ntiles) {                         // the NEXT band's first patch loads fly under this band's conv2 stage (block-uniform)
                cb = band_of(tile + G);
                request(cb, min(wave, cb.ngrp - 1), r0);
                if constexpr (DEEP) request(cb, min(wave + 4, cb.ngrp - 1), r1);
            }
        }

        // ================= conv2: 6 x 38 outputs = 8 position tiles of 32 (the last one ragged); this wave: output tile nt, position tiles 4 mh .. 4 mh + 3 =================
#pragma unroll 1
        for (int mt = (dbg & 4) ? 4 : 0; mt < 4; ++mt) {
            const int oq = (mh * 4 + mt) * 32 + lrow;
            const int oc = min(oq, 6 * A2W - 1);
            const int oy = oc / A2W, ox = oc - oy * A2W;
            f32x16 acc = acc2_0;
            if constexpr (C2) {
                constexpr int NF = 4;
                uint32_t base[2][2];                       // LDS address of (row 2 oy, entry ox + kwh, chunk 2 ks + lane group) of plane 0; taps = immediates on top
#pragma unroll
                for (int kwh = 0; kwh < 2; ++kwh) {
                    const int idx = ox + kwh, sw = (idx >> 2) & 3;
                    const uint32_t pb = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (uint32_t)((2 * oy * E12_PW + idx) * 64);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) base[kwh][ks] = pb + (uint32_t)((((2 * ks + lgrp) ^ sw)) << 4);
                }
                if (dbg & 128) {                           // (ablation: every lane its own 16 bytes of one KiB -- the reads without any bank conflict)
#pragma unroll
                    for (int i = 0; i < 4; ++i) base[i >> 1][i & 1] = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (uint32_t)(lane * 16 + i * 1024);
                }
                u16x8 fb[NF];
                E12Conv2Prologue<0, NF>::run(fb, base);
                E12Conv2Pipe<0, NF>::run(fb, wf2, acc, base);
            } else
#pragma unroll
            for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                for (int kw = 0; kw < 4; ++kw) {
                    const int idx = ox + (kw >> 1);
                    const unsigned char* q = lds + (kw & 1) * E12_PLANE + ((2 * oy + kh) * E12_PW + idx) * 64;
                    const int sw = (idx >> 2) & 3;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const u16x8 bfr = (dbg & 64) ? wf1[ks] : *(const u16x8*)(q + (((2 * ks + lgrp) ^ sw) << 4));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf2[(kh * 4 + kw) * 2 + ks]), __builtin_bit_cast(bf16x8, bfr), acc, 0, 0, 0);
                    }
                }
            uint32_t R[4][2];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const float v[4] = {acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]};
                const PackN<uint32_t, 2> w = __builtin_bit_cast(PackN<uint32_t, 2>, pack4<bf16_t>(v));
#pragma unroll
                for (int d = 0; d < 2; ++d) { uint32_t u; asm("v_pk_max_i16 %0, %1, 0" : "=v"(u) : "v"(w.v[d])); R[qd][d] = u; }
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                auto s0 = __builtin_amdgcn_permlane32_swap(R[0][d], R[2][d], false, false); R[0][d] = s0[0]; R[2][d] = s0[1];
                auto s1 = __builtin_amdgcn_permlane32_swap(R[1][d], R[3][d], false, false); R[1][d] = s1[0]; R[3][d] = s1[1];
            }
            if (oq < 6 * A2W && !(dbg & 8)) {
                const long long m2 = ((long long)b * A2H + 6 * band + oy) * A2W + ox;
                unsigned char* g = (unsigned char*)p.act2 + m2 * 128 + nt * 64 + lgrp * 32;
                *(e12_u32x4*)g = e12_u32x4{R[0][0], R[0][1], R[2][0], R[2][1]};
                *(e12_u32x4*)(g + 16) = e12_u32x4{R[1][0], R[1][1], R[3][0], R[3][1]};
            }
        }
        __syncthreads();                                   // every wave is done with the band before the next one's conv1 overwrites it
    }
    if constexpr (RING && SSZ == 1) asm volatile("s_waitcnt vmcnt(0)");      // (the last band's look-ahead requests: nothing reads them, nothing is left in flight)
}

}  // namespace mi
