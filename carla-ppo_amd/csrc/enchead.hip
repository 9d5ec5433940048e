// enchead.hip — host side of the fused encoder-head backward kernel (enchead_tile.hpp): conv2's input gradient + conv1's filter / bias gradient in one launch.
#include <stdlib.h>
#include "enchead_tile.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"

using namespace mi;

namespace mi {
// dw1[1536] += and db1[32] += the per-block partial sums: a block of 32 x 32 threads owns 32 consecutive outputs (block 48: the bias sums at slab offset 2048), its
// 32 thread rows take every 32nd slab, eight loads in flight per thread; fixed summation order (dectail_reduce_kernel's form: the generic reduce_slabs_kernel took
// 20 + 10 us for these two at the end of the caller's stream)
__global__ __launch_bounds__(1024) void enchead_reduce_kernel(const float* __restrict__ slabs, int n, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ float red[32][33];
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    const bool bias = blockIdx.x == 48;
    const int col = bias ? 64 * 32 + c : (int)blockIdx.x * 32 + c;
    float s = 0.f;
    for (int k0 = r0; k0 < n; k0 += 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + 32 * u; v[u] = k < n ? __builtin_nontemporal_load(&slabs[(long long)k * EH_SLAB + col]) : 0.f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    red[r0][c] = s;
    __syncthreads();
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
        if (bias) db[threadIdx.x] += t; else dw[col] += t;
    }
}
}  // namespace mi

static int enchead_grid(int u8) {
    static int resident[2];
    if (!resident[u8]) {
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t pr;
        const void* fn = u8 ? (const void*)enchead_bwd_kernel<unsigned char> : (const void*)enchead_bwd_kernel<float>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        resident[u8] = per_cu * cus;
    }
    return resident[u8];
}

extern "C" int mi_conv2d_head_bwd_blocks(void) { const int a = enchead_grid(0), b = enchead_grid(1); return a > b ? a : b; }

// Backward of the encoder head in ONE launch: frames [*, FH, FW, 3] -conv1 (k4 s2, 32 ch, ReLU)-> act1 [B, IH, IW, 32] -conv2 (k4 s2, 64 ch)-> [B, OH, OW, 64].
// Given dy2 = gradient wrt conv2's pre-activation, conv2's kernel and the ReLU bit words of act1: dw1 += conv1's filter gradient [4][4][3][32], db1 += its bias gradient.
// The gradient of act1 itself is never written (conv1 has no input gradient: nothing else reads it).  *n_blocks = blocks launched, or 0 when the shapes / dtypes are
// not the ones the kernel takes (nothing was launched: use mi_conv2d_nhwc_dgrad_bits + mi_conv2d_nhwc_wgrad_ws).  scratch >= mi_conv2d_head_bwd_blocks() * 8320 bytes.
extern "C" int mi_conv2d_head_bwd_fused(void* stream, int dtype, const void* frames, int frames_fmt, const int* frame_idx, int B, int FH, int FW,
                                        const void* dy2, const void* w2, const void* bits_act1, float* dw1, float* db1, void* scratch, long long scratch_bytes, int* n_blocks) {
    if (!n_blocks || !frames || !dy2 || !w2 || !dw1 || !db1) return mi_fail(MI_ERR_ARG, "mi_conv2d_head_bwd_fused: missing buffers");
    *n_blocks = 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("MI355_ENCHEAD"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || !mi_narrow_enabled() || dtype != MI_BF16 || !bits_act1 || !scratch || B < 1) return MI_OK;
    if (frames_fmt != 1 && frames_fmt != 2) return MI_OK;                     // fp32 or uint8 camera frames
    if (FH < 10 || FW < 10) return MI_OK;
    const int IH = (FH - 4) / 2 + 1, IW = (FW - 4) / 2 + 1;
    if (IH < 4 || IW < 4) return MI_OK;
    const int OH = (IH - 4) / 2 + 1, OW = (IW - 4) / 2 + 1;
    const int ssz = frames_fmt == 2 ? 1 : 4;
    // 4-value patch groups are read as 2-byte-aligned (uint8) / 8-byte-aligned (fp32) vectors: pixel offsets are multiples of 6 values
    if ((((uintptr_t)frames) & (2 * ssz - 1)) || ((long long)FW * 3 * ssz) % (2 * ssz) != 0 || ((long long)FH * FW * 3 * ssz) % (2 * ssz) != 0) return MI_OK;
    if ((((uintptr_t)dy2) | ((uintptr_t)w2) | ((uintptr_t)scratch) | ((uintptr_t)dw1) | ((uintptr_t)db1)) & 15) return MI_OK;
    if ((long long)B * OH * OW * 128 >= 0x7fffff00ll || (long long)B * IH * IW * 8 >= 0x7fffff00ll) return MI_OK;      // 32-bit buffer offsets
    EncHeadParams q = {};
    q.dy = (const bf16_t*)dy2; q.dy_bytes = (unsigned)((long long)B * OH * OW * 128); q.B = B; q.OH = OH; q.OW = OW;
    q.w = (const bf16_t*)w2; q.bits = (const uint32_t*)bits_act1; q.bits_bytes = (unsigned)((long long)B * IH * IW * 8);
    q.frames = frames; q.frame_idx = frame_idx; q.frame_stride = (long long)FH * FW * 3;
    q.IH = IH; q.IW = IW; q.FW = FW;
    const int tiles_y = (IH + EH_TY - 1) / EH_TY;
    q.tiles_x = (IW + EH_TX - 1) / EH_TX; q.tiles_per_frame = tiles_y * q.tiles_x;
    const long long ntiles = (long long)B * q.tiles_per_frame;
    if (ntiles >= (1ll << 30)) return MI_OK;
    q.ntiles = (int)ntiles; q.div_tpf = make_fastdiv(q.tiles_per_frame); q.div_tx = make_fastdiv(q.tiles_x);
    int nblocks = enchead_grid(frames_fmt == 2 ? 1 : 0);
    if (nblocks > q.ntiles) nblocks = q.ntiles;
    if (nblocks >= 16) nblocks &= ~7;                       // whole rounds of the eight XCDs (the kernel's tile order)
    if (scratch_bytes < (long long)nblocks * EH_SLAB * 4) return MI_OK;
    q.slabs = (float*)scratch;
    hipStream_t st = (hipStream_t)stream;
    if (frames_fmt == 2) MI_LAUNCH(enchead_bwd_kernel<unsigned char>, dim3(nblocks), dim3(256), 0, st, q);
    else MI_LAUNCH(enchead_bwd_kernel<float>, dim3(nblocks), dim3(256), 0, st, q);
    int rc = mi_check_launch("enchead_bwd_kernel");
    if (rc != MI_OK) return rc;
    if (mi_small_reduce_deferring()) {                       // two jobs of the pass's fused small reduce: the 48 x 32 filter-gradient sums and the 32 bias sums at slab offset 2048
        rc = mi_reduce_slabs(st, q.slabs, (long long)EH_SLAB, nblocks, 48ll * 32, dw1);
        if (rc == MI_OK) rc = mi_reduce_slabs(st, q.slabs + 64 * 32, (long long)EH_SLAB, nblocks, 32ll, db1);
    } else {
        MI_LAUNCH(enchead_reduce_kernel, dim3(49), dim3(1024), 0, st, (const float*)q.slabs, nblocks, dw1, db1);
        rc = mi_check_launch("enchead_reduce_kernel");
    }
    if (rc == MI_OK) *n_blocks = nblocks;
    return rc;
}
