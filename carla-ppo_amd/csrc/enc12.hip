// enc12.hip — host side of the fused encoder-head FORWARD kernel (enc12_tile.hpp): conv1 + conv2 of vae/models.py:250-251 in one launch.
#include <stdlib.h>
#include "enc12_tile.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"

using namespace mi;

static int g_enc12_dbg = 0;
int mi_enc12_debug(int mask) { const int prev = g_enc12_dbg; g_enc12_dbg = mask < 0 ? 0 : mask; return prev; }

static int enc12_grid(int u8) {
    static int resident[2];
    if (!resident[u8]) {
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t pr;
        const void* fn = u8 ? (const void*)enc12_fwd_kernel<unsigned char> : (const void*)enc12_fwd_kernel<float>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        resident[u8] = per_cu * cus;
    }
    return resident[u8];
}

// The encoder head of a forward pass in ONE launch (round 5): frames [*, 80, 160, 3] as raw uint8 camera bytes (frames_fmt 2) or float32 in [0, 1] (1), optionally gathered through frame_idx, -conv1 k4 s2 + bias +
// ReLU-> act1 [B, 39, 79, 32] -conv2 k4 s2 + bias + ReLU-> act2 [B, 18, 38, 64], bf16 storage.  w1_t / w2_t: the K-contiguous kernel copies ([32][48] and [64][512]);
// act1 and (optionally) its ReLU bit words are written exactly as mi_conv2d_nhwc_fwd_bits writes them (bit for bit), act2 as conv2 of that activation with fp32 accumulation
// (another summation order than the unfused kernel).  *launched = 0: not eligible -- bf16 storage and this geometry only; nothing was launched: call the two layer ops.
extern "C" int mi_conv2d_enc12_fwd(void* stream, int dtype, const void* frames, int frames_fmt, const int* frame_idx, int B, int FH, int FW, const void* w1_t, const float* b1,
                                   const void* w2_t, const float* b2, void* act1, void* relu_bits1, void* act2, int* launched) {
    if (!launched || !frames || !w1_t || !b1 || !w2_t || !b2 || !act1 || !act2) return mi_fail(MI_ERR_ARG, "mi_conv2d_enc12_fwd: missing buffers");
    *launched = 0;
    static int on = -1;
    if (on < 0) { const char* e = getenv("MI355_ENC12"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || !mi_narrow_enabled() || dtype != MI_BF16 || (frames_fmt != 1 && frames_fmt != 2) || B < 1 || FH != 80 || FW != 160) return MI_OK;
    if ((((uintptr_t)frames) & (frames_fmt == 2 ? 1 : 7)) || (((uintptr_t)w1_t) | ((uintptr_t)w2_t) | ((uintptr_t)b1) | ((uintptr_t)b2) | ((uintptr_t)act1) | ((uintptr_t)act2)) & 15) return MI_OK;
    if (relu_bits1 && (((uintptr_t)relu_bits1) & 3)) return MI_OK;
    if ((long long)B * 3 >= (1ll << 30)) return MI_OK;
    Enc12Params q = {};
    q.frames = frames; q.frame_idx = frame_idx; q.frame_stride = (long long)FH * FW * 3;
    q.w1 = (const bf16_t*)w1_t; q.b1 = b1; q.w2 = (const bf16_t*)w2_t; q.b2 = b2;
    q.act1 = (bf16_t*)act1; q.bits1 = (uint32_t*)relu_bits1; q.act2 = (bf16_t*)act2;
    q.w2f = (const bf16_t*)mi_tl_rc_wfrag; mi_tl_rc_wfrag = nullptr;      // (announced by mi_rwconv_next_weights_fragment_ordered; consumed by this launch)
    q.B = B; q.ntiles = 3 * B;
    int nblocks = enc12_grid(frames_fmt == 2 ? 1 : 0);
    if (nblocks > q.ntiles) nblocks = q.ntiles;
    // camera bytes (the production format): the ring form of the conv1 stage's frame loads and conv2's LDS fragment reads pipelined by hand (enc12_tile.hpp; late round 5:
    // 55.5 -> 53.0 -> 51.1 us for the op alone at batch 512, interleaved medians; step -0.5 ... -0.8 %).  MI355_ENC12_RING=0 / MI355_ENC12_C2=0: the compiler-scheduled forms (A/B).
    // Both forms issue loads by inline assembly and wait by hand: tools/check_enc12_isa.py (run by tests/test_host_logic.py) proves on the generated code that no
    // register is read while its load can be outstanding.
    static int ring = -1;
    if (ring < 0) { const char* e = getenv("MI355_ENC12_RING"); ring = (e && e[0] == '0') ? 0 : 1; }
    static int c2 = -1;
    if (c2 < 0) { const char* e = getenv("MI355_ENC12_C2"); c2 = (e && e[0] == '0') ? 0 : 1; }
    int use_ring = ring, use_c2 = c2;
    int dbg = g_enc12_dbg;
    if (dbg & 4096) { use_ring = (dbg >> 13) & 1; use_c2 = (dbg >> 14) & 1; dbg = 0; }      // (bit 4096: pick a PRODUCT form by mask -- bit 8192 ring, 16384 pipelined conv2 -- for interleaved timing in one process)
    if (dbg && frames_fmt == 2) {                           // ablation timing (mi_set_tuning key 23; tools/enc12_ablate.py): WRONG results by construction
        q.dbg = dbg;
        if (dbg & 2048) MI_LAUNCH((enc12_fwd_kernel<unsigned char, 1, 1, 1>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);      // (bit 2048: the ring + pipelined form)
        else MI_LAUNCH((enc12_fwd_kernel<unsigned char, 1>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);
    } else if (use_ring && use_c2 && frames_fmt == 2) MI_LAUNCH((enc12_fwd_kernel<unsigned char, 0, 1, 1>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);
    else if (use_ring && frames_fmt == 2) MI_LAUNCH((enc12_fwd_kernel<unsigned char, 0, 1>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);
    else if (frames_fmt == 2) MI_LAUNCH(enc12_fwd_kernel<unsigned char>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);
    else MI_LAUNCH(enc12_fwd_kernel<float>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);
    const int rc = mi_check_launch("enc12_fwd_kernel");
    if (rc == MI_OK) *launched = 1;
    return rc;
}
