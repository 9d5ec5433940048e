// conv_ops.hip — host launchers (C ABI) for the MFMA tile kernels in gemm_core.hpp.
// Every conv on this path is NHWC, stride 2, VALID (reference vae/models.py:250-253,261-264).
#include <stdlib.h>
#include "gemm_core.hpp"
#include "tallk_tile.hpp"
#include "mi_internal.hpp"
#include "mi355_carla.h"

using namespace mi;

namespace {

inline int esz_of(int dtype) { return dtype == MI_BF16 ? 2 : 4; }      // MI_F32 and MI_BF16X3 (split storage) are 4-byte elements
inline bool dtype_ok(int dtype) { return dtype == MI_F32 || dtype == MI_BF16 || dtype == MI_BF16X3; }

template <typename T, typename TIn, int AMODE, int BMODE, int VA, int AALIGN>
int launch_gemm_bn(hipStream_t st, const GemmParams& p, int M_for_grid, int gz) {
    const int gx = (M_for_grid + GEMM_BM - 1) / GEMM_BM;
    if (gx <= 0) return MI_OK;
    if (p.N <= 32) {
        dim3 g(gx, 1, gz);
        MI_LAUNCH((gemm_kernel<T, TIn, AMODE, BMODE, VA, AALIGN, 32>), g, dim3(GEMM_NT), 0, st, p);
    } else if (p.N <= 64) {
        dim3 g(gx, 1, gz);
        MI_LAUNCH((gemm_kernel<T, TIn, AMODE, BMODE, VA, AALIGN, 64>), g, dim3(GEMM_NT), 0, st, p);
    } else {
        dim3 g(gx, (p.N + 127) / 128, gz);
        MI_LAUNCH((gemm_kernel<T, TIn, AMODE, BMODE, VA, AALIGN, 128>), g, dim3(GEMM_NT), 0, st, p);
    }
    return mi_check_launch("gemm_kernel");
}

void fill_conv_geom(GemmParams& p, int B, int IH, int IW, int C, int OH, int OW, int KH, int KW, int stride, bool merged) {
    p.IH = IH; p.IW = IW; p.C = C; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.stride = stride;
    p.nbatch = B;
    p.M = B * OH * OW;
    p.K = KH * KW * C;
    p.a_frame_stride = (long long)IH * IW * C;
    p.div_ohw = make_fastdiv(OH * OW);
    p.div_ow = make_fastdiv(OW);
    p.merged = merged ? 1 : 0;
    p.run = merged ? KW * C : C;
    p.div_run = make_fastdiv(p.run);
    p.div_kw = make_fastdiv(KW);
}

// ---------------------------------------------------------------------------------------------------------------
// gemm2 (LDS-DMA tiles, gemm2_tile.hpp): taken whenever the layer is "wide" (whole 16-byte chunks per pixel), the
// weights are K-contiguous and every tensor fits a 1 GiB buffer descriptor.  MI355_GEMM2=0 forces the first-generation
// register-staged kernel (A/B comparisons, bisecting).
// ---------------------------------------------------------------------------------------------------------------
long long* g_trace = nullptr; int g_trace_cap = 0;
int g_wgrad_skip = 0;
int g_tap_mask_prefetch = 1;                                // tapconv: touch the ReluGrad-mask lines in the last main-loop step; mi_set_tuning key 12
int g_gemm2_on = -1;
static int gemm2_stages_env() { const char* e = getenv("MI355_GEMM2_STAGES"); return e ? atoi(e) : 2; }
int g_gemm2_stages = gemm2_stages_env();                    // gemm2 128 x 64 tiles: LDS stages of the K pipeline (2 | 3 | 4); mi_set_tuning key 20
static int gemm2_tile_env() { const char* e = getenv("MI355_GEMM2_TILE"); return e ? atoi(e) : 2; }
static int gemm2_splitk_env() { const char* e = getenv("MI355_GEMM2_SPLITK"); return (e && e[0] == '0') ? 0 : 1; }
int g_gemm2_splitk = gemm2_splitk_env();                                   // split-K dense layers (raw fp32 slabs) on the LDS-DMA tiles instead of the first-generation kernel (round 4: the 38400-long reductions of the MlpVAE)
int g_gemm2_tile = gemm2_tile_env();                                       // wide-output gemm2 layers: 0 auto (64 x 64 tiles on small grids), 1 always 64 x 64, 2 never, 3 always 128 x 128 (64 x 64 wave tiles: 1 KB of LDS reads per MFMA instead of 1.5); mi_set_tuning key 17
int g_tap_min = -2;
bool gemm2_enabled() {
    if (g_gemm2_on < 0) { const char* e = getenv("MI355_GEMM2"); g_gemm2_on = (e && e[0] == '0') ? 0 : 1; }
    return g_gemm2_on != 0;
}

template <typename T, int AMODE, int BMODE, bool UTAP>
void launch_gemm2_128x64(hipStream_t st, dim3 g, const Gemm2Params& p) {
    const int nk = (p.K * (int)sizeof(T) + 127) / 128;    // (upper bound for the gather form: its K depends on the parity class)
    const int ns = nk >= 6 ? g_gemm2_stages : 2;           // a deeper ring needs steps to fill
    if (ns >= 4) MI_LAUNCH((gemm2_kernel<T, AMODE, BMODE, 128, 64, UTAP, 4>), g, dim3(GEMM_NT), 0, st, p);
    else if (ns == 3) MI_LAUNCH((gemm2_kernel<T, AMODE, BMODE, 128, 64, UTAP, 3>), g, dim3(GEMM_NT), 0, st, p);
    else MI_LAUNCH((gemm2_kernel<T, AMODE, BMODE, 128, 64, UTAP, 2>), g, dim3(GEMM_NT), 0, st, p);
}

template <typename T, int AMODE, int BMODE, bool UTAP>
int launch_gemm2_tiles(hipStream_t st, const Gemm2Params& p, int M_for_grid, int gz) {
    if (M_for_grid <= 0) return MI_OK;
    if (p.N <= 32) {
        dim3 g((M_for_grid + 255) / 256, 1, gz);
        MI_LAUNCH((gemm2_kernel<T, AMODE, BMODE, 256, 32, UTAP>), g, dim3(GEMM_NT), 0, st, p);
    } else if (p.N <= 64) {
        dim3 g((M_for_grid + 127) / 128, 1, gz);
        launch_gemm2_128x64<T, AMODE, BMODE, UTAP>(st, g, p);
    } else {
        if (g_gemm2_tile == 1 || (g_gemm2_tile == 0 && (long long)((M_for_grid + 127) / 128) * ((p.N + 63) / 64) * gz < 512)) {
            // small grids (conv4 forward / deconv1 input gradient at batch 512: 96 x 4 tiles of 128 x 64 = 1.5 blocks per CU, each a serial chain of
            // 32 latency-bound k-steps): 64 x 64 tiles double the blocks in flight (32 KB of LDS each: four resident per CU cover each other's waits)
            dim3 g((M_for_grid + 63) / 64, (p.N + 63) / 64, gz);
            MI_LAUNCH((gemm2_kernel<T, AMODE, BMODE, 64, 64, UTAP>), g, dim3(GEMM_NT), 0, st, p);
            return mi_check_launch("gemm2_kernel");
        }
        const int gx = (M_for_grid + 127) / 128;
        if (g_gemm2_tile == 3 || (long long)gx * ((p.N + 127) / 128) * gz >= 384) {
            dim3 g(gx, (p.N + 127) / 128, gz);
            const int nk = (p.K * (int)sizeof(T) + 127) / 128;
            if (g_gemm2_tile == 3 && g_gemm2_stages >= 3 && nk >= 6) MI_LAUNCH((gemm2_kernel<T, AMODE, BMODE, 128, 128, UTAP, 3>), g, dim3(GEMM_NT), 0, st, p);
            else MI_LAUNCH((gemm2_kernel<T, AMODE, BMODE, 128, 128, UTAP>), g, dim3(GEMM_NT), 0, st, p);
        } else {                                          // few tiles: narrower blocks fill the 256 CUs
            dim3 g(gx, (p.N + 63) / 64, gz);
            launch_gemm2_128x64<T, AMODE, BMODE, UTAP>(st, g, p);
        }
    }
    return mi_check_launch("gemm2_kernel");
}

inline bool fits_desc(long long bytes) { return bytes > 0 && bytes < (long long)G2_OOB; }

void copy_epilogue(Gemm2Params& q, const GemmParams& p) {
    q.out = p.out; q.bias = p.bias; q.mask = p.mask; q.relu = p.relu; q.out_f32 = p.out_f32;
}

// ---------------------------------------------------------------------------------------------------------------
// tapconv (raw-staged slot tiles, tapconv_tile.hpp): stride-2 k=4/5 layers with whole 16-byte chunks per pixel and enough
// positions to fill the chip.  MI355_TAPCONV=0 disables it; MI355_TAPCONV_MINBLOCKS overrides the occupancy threshold.
// ---------------------------------------------------------------------------------------------------------------
int tapconv_minblocks() {
    if (g_tap_min == -2) {
        const char* e = getenv("MI355_TAPCONV");
        if (e && e[0] == '0') g_tap_min = -1;
        else { const char* m = getenv("MI355_TAPCONV_MINBLOCKS"); g_tap_min = m ? atoi(m) : 300; }
    }
    return g_tap_min;
}

int g_tap_direct = 1;                                      // tapconv epilogue: 1 registers -> 16-byte stores, 0 LDS-staged; mi_set_tuning key 6
int g_tap_variant = 0;                                    // 0 auto, 1 big tile (256 x 96), 2 small tile (128 x 48); mi_set_tuning key 5

template <typename T, int MODE, int TAPS, int BMT, int MAXHALO>
int launch_tapconv_v(hipStream_t st, const TapParams& q) {
    const int gx = (q.MP + BMT - 1) / BMT;
    if (q.NE >= 128) {
        dim3 g(gx, (q.NE + 127) / 128, 1);
        MI_LAUNCH((tapconv_kernel<T, MODE, 128, TAPS, BMT, MAXHALO>), g, dim3(BMT * 2), 0, st, q);
    } else {
        dim3 g(gx, (q.NE + 63) / 64, 1);
        MI_LAUNCH((tapconv_kernel<T, MODE, 64, TAPS, BMT, MAXHALO>), g, dim3(BMT * 2), 0, st, q);
    }
    return mi_check_launch("tapconv_kernel");
}
template <typename T, int MODE, int TAPS>
int launch_tapconv_t(hipStream_t st, const TapParams& q, bool small_tile) {
    return small_tile ? launch_tapconv_v<T, MODE, TAPS, 128, 48>(st, q) : launch_tapconv_v<T, MODE, TAPS, TC_BMT, TC_MAXHALO>(st, q);
}
template <typename T, int MODE>
int launch_tapconv(hipStream_t st, const TapParams& q, bool small_tile) {
    return q.TH == 2 ? launch_tapconv_t<T, MODE, 2>(st, q, small_tile) : launch_tapconv_t<T, MODE, 3>(st, q, small_tile);
}

// mode TC_CONV: x[B,IH,IW,C] -> out[B,OH,OW,N], weights K-contiguous [N][KH*KW*C];  mode TC_GATHER: x[B,IH,IW,C] -> out[B,OH,OW,N],
// weights [KH][KW][N][C].  Returns 1 launched, 0 not eligible, <0 error.
int try_tapconv(hipStream_t st, int dtype, int mode, const void* a, const void* w, int B, int IH, int IW, int C, int OH, int OW, int N,
                int KH, int KW, int ldb, void* out, const float* bias, const void* mask, int relu) {
    const int minblocks = tapconv_minblocks();
    if (minblocks < 0) return 0;
    const int esz = esz_of(dtype);
    if (KH != KW || KH < 3 || KH > 6) return 0;
    if ((C * esz) % 16 != 0 || (((uintptr_t)a) & 15) || (((uintptr_t)w) & 15)) return 0;
    // coalesced epilogue: whole 16-byte chunks of one pixel's channels, 32-bit element offsets
    if ((N * esz) % 16 != 0 || (((uintptr_t)out) & 15) || (mask && (((uintptr_t)mask) & 15)) || (long long)B * OH * OW * N >= (1ll << 31)) return 0;
    TapParams q = {};
    q.TH = (KH + 1) / 2; q.TW = (KW + 1) / 2;
    if (mode == TC_CONV) {
        if ((ldb * esz) % 16 != 0 || ldb < KH * KW * C) return 0;
        if (q.TH == 3 && minblocks > 1) return 0;        // measured: k=5 conv form (deconv3 dgrad) is faster on gemm2 (76 vs 93 us)
        q.GH = OH + q.TH - 1; q.GW = OW + q.TW - 1; q.HY = q.HX = 0;
        q.KC = 4 * C; q.NE = N;
    } else {
        if (N % 32 != 0) return 0;                        // a 32-wide output tile must stay inside one parity class
        q.HY = q.TH - 1; q.HX = q.TW - 1;
        q.GH = (OH + 1) / 2 + q.HY; q.GW = (OW + 1) / 2 + q.HX;
        q.KC = C; q.NE = 4 * N;
    }
    if ((q.TH - 1) * q.GW + q.TW - 1 > TC_MAXHALO) return 0;
    const long long MP = (long long)B * q.GH * q.GW;
    const long long a_bytes = (long long)B * IH * IW * C * esz;
    const long long b_bytes = (mode == TC_CONV ? (long long)N * ldb : (long long)KH * KW * N * C) * esz;
    if (MP >= (1ll << 30) || !fits_desc(a_bytes) || !fits_desc(b_bytes)) return 0;
    const long long blocks = ((MP + TC_BMT - 1) / TC_BMT) * ((q.NE + (q.NE >= 128 ? 127 : 63)) / (q.NE >= 128 ? 128 : 64));
    if (blocks < minblocks) return 0;
    q.a = a; q.a_bytes = (uint32_t)a_bytes; q.b = w; q.b_bytes = (uint32_t)b_bytes;
    q.B = B; q.IH = IH; q.IW = IW; q.C = C; q.OH = OH; q.OW = OW; q.N = N; q.KH = KH; q.KW = KW;
    q.MP = (int)MP; q.ldb = ldb;
    q.div_g = make_fastdiv(q.GH); q.div_gw = make_fastdiv(q.GW); q.div_n = make_fastdiv(N);
    q.div_2c = make_fastdiv(2 * C); q.div_c = make_fastdiv(C);
    q.out = out; q.bias = bias; q.mask = mask; q.relu = relu;
    q.direct_epilogue = (g_tap_direct && N % 32 == 0) ? 1 : 0;       // a 32-output tile is all valid or all out of range
    q.trace = g_trace; q.trace_cap = g_trace_cap; q.dbg = g_wgrad_skip; q.mask_prefetch = g_tap_mask_prefetch;
    const int halo = (q.TH - 1) * q.GW + q.TW - 1;
    // measured (tools/trace_tapconv.py variants): the 128-position tile wins 5-12 % where the 256-position grid is only 1.3-3 rounds
    // of blocks (tile quantisation), loses a little on the 4-column grids and ties on the big grids
    const int gy_t = (q.NE + (q.NE >= 128 ? 127 : 63)) / (q.NE >= 128 ? 128 : 64);
    const bool auto_small = blocks >= 300 && blocks <= 1000 && gy_t <= 2;
    const bool small_tile = halo <= 48 && dtype == MI_BF16 && (g_tap_variant == 2 || (g_tap_variant == 0 && auto_small));
    int rc;
    if (dtype == MI_F32) rc = mode == TC_CONV ? launch_tapconv<float, TC_CONV>(st, q, false) : launch_tapconv<float, TC_GATHER>(st, q, false);
    else if (dtype == MI_BF16X3) rc = mode == TC_CONV ? launch_tapconv<split_t, TC_CONV>(st, q, false) : launch_tapconv<split_t, TC_GATHER>(st, q, false);
    else rc = mode == TC_CONV ? launch_tapconv<bf16_t, TC_CONV>(st, q, small_tile) : launch_tapconv<bf16_t, TC_GATHER>(st, q, small_tile);
    return rc == MI_OK ? 1 : rc;
}

// ---------------------------------------------------------------------------------------------------------------
// Deferred split reductions.  The reduce that follows a tapwgrad launch is a small, bandwidth-trivial kernel, but on the filter-gradient
// stream of the engine it shares the chip with a big input-gradient kernel and takes 15-30 us instead of 7.  In deferred mode the launch
// is recorded instead (parameters by value; every layer then needs its OWN scratch region until the flush) and mi_tapwgrad_flush issues
// all of them back to back.  Per host THREAD (thread_local): a backward pass is issued by one thread from defer(1) to flush, so two engines driven by two
// threads keep separate lists (round 3; the tuning knobs stay process-global configuration).  Used by the VAE engine only.
// one more job of a fused ordered slab sum (kernel: tapwgrad_tile.hpp, sr_job_body); the lane count depends on the job's shape only
static int sr_kl(long long n, int nslab) {                // slab lanes per element group: a power of two, >= ~96 blocks per job where the slabs allow
    const long long quads = (n + 3) / 4;
    int kl = 1;
    while (kl < 256 && kl * 2 <= nslab && (quads * kl + 255) / 256 < 96) kl *= 2;
    return kl;
}
static void sr_add(SmallReduceParams& f, const float* slabs, long long stride, int nslab, long long n, float* out, int overwrite = 0) {
    const int j = f.njobs++;
    if (j == 0) { f.first[0] = 0; f.ovw = 0; }
    if (overwrite) f.ovw |= 1u << j;
    f.slabs[j] = slabs; f.out[j] = out; f.stride[j] = stride; f.n[j] = n; f.nslab[j] = nslab;
    f.kl[j] = sr_kl(n, nslab);
    f.vec[j] = ((((uintptr_t)slabs) | ((uintptr_t)out)) & 15) == 0 && stride % 4 == 0;
    const long long quads = (n + 3) / 4;
    f.first[j + 1] = f.first[j] + (int)((quads * f.kl[j] + 255) / 256);
}
struct PendingReduce { TapWgradParams q; int splits, ngroups, kind; const float* bpart; float* bout; int bnslab, bN; };
// Split storage (MI_BF16X3) on the bf16 filter-gradient kernels: the tensors are handed over as bf16 tensors with TWICE the channels (channel 2c = lo half, 2c + 1 = hi half of
// element c), the kernel produces dW'[t][2R][2Q] in a temporary, and the fold  dW[t][r][q] += dW'[t][2r][2q] + dW'[t][2r][2q+1] + dW'[t][2r+1][2q] + dW'[t][2r+1][2q+1]
// (all four partial products of the two-term expansion) follows the slab reduce -- deferred with it when the reduces are deferred.
struct PendingFold { const float* tmp; float* out; const float* tmp_bias; float* dbias; int T, R, Q, NB; };
static thread_local PendingFold g_folds[16];
static thread_local int g_nfolds = 0;
__global__ __launch_bounds__(256) void fold_split_kernel(const PendingFold f) {
    const long long n = (long long)f.T * f.R * f.Q;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const int q = (int)(i % f.Q); const long long tr = i / f.Q; const int r = (int)(tr % f.R); const long long t = tr / f.R;
        const float* s = f.tmp + ((t * 2 * f.R + 2 * r) * 2ll * f.Q + 2 * q);
        f.out[i] += (s[0] + s[1]) + (s[2ll * f.Q] + s[2ll * f.Q + 1]);
    } else if (f.dbias && i - n < f.NB) {
        const int b = (int)(i - n);
        f.dbias[b] += f.tmp_bias[2 * b] + f.tmp_bias[2 * b + 1];
    }
}
static int launch_fold(hipStream_t st, const PendingFold& f) {
    const long long n = (long long)f.T * f.R * f.Q + (f.dbias ? f.NB : 0);
    MI_LAUNCH(fold_split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f);
    return mi_check_launch("fold_split_kernel");
}
static thread_local PendingReduce g_pending[16];
static thread_local int g_npending = 0, g_defer_reduces = 0, g_defer_pause = 0;
static unsigned reduce_ry_cap() { static int c = -1; if (c < 0) { const char* e = getenv("MI355_REDUCE_RY_CAP"); c = e ? atoi(e) : 16; if (c < 1 || c > 64) c = 16; } return (unsigned)c; }      // (16: 0.7900 / 0.7923 against 0.7929 / 0.7949 ms per step at 64, two interleaved A/B runs of four rounds)
static unsigned reduce_ry(const PendingReduce& r) {       // slab chains per element (a power of two <= 64, from the shape only): about 512 blocks in flight, at most ~16 slabs per thread
    unsigned ry = 1;                                        // (MI355_REDUCE_RY_CAP: A/B knob -- fewer chains = longer contiguous pieces per slab and block, fewer blocks)
    while (((unsigned)(r.ngroups / 512 + 1) * ry < 512 || r.splits / (int)ry > 16) && (int)(ry * 2) <= r.splits / 4 && ry < reduce_ry_cap()) ry *= 2;
    return ry;
}
static unsigned reduce_blocks(const PendingReduce& r, unsigned ry) { const unsigned upb = 256 / ry, nunits = (unsigned)r.ngroups / 2; return (nunits + upb - 1) / upb; }      // units = pairs of 16-byte groups
static void launch_tiled_reduce(hipStream_t st, const PendingReduce& r) {
    const unsigned ry = reduce_ry(r);
    const dim3 rg(reduce_blocks(r, ry), 1, 1);
    if (r.kind == 0) MI_LAUNCH((reduce_tiled_kernel<TC_CONV, 2, 4, 2>), rg, dim3(256), 0, st, r.q, r.splits, r.ngroups, (int)ry);
    else if (r.kind == 1) MI_LAUNCH((reduce_tiled_kernel<TC_GATHER, 2, 4, 2>), rg, dim3(256), 0, st, r.q, r.splits, r.ngroups, (int)ry);
    else MI_LAUNCH((reduce_tiled_kernel<TC_GATHER, 3, 2, 4>), rg, dim3(256), 0, st, r.q, r.splits, r.ngroups, (int)ry);
    if (r.bpart) mi_reduce_slabs(st, r.bpart, (long long)r.bN, r.bnslab, (long long)r.bN, r.bout);
}
extern "C" int mi_tapwgrad_defer(int on) {               // switching the mode drops whatever an aborted pass may have left in the list
    const int prev = g_defer_reduces;
    g_defer_reduces = on ? 1 : 0;
    g_npending = 0; g_nfolds = 0; g_defer_pause = 0;
    return prev;
}
// pause != 0: the next filter gradients reduce their slabs right behind their own launch although the pass defers (a layer issued on ANOTHER stream than the
// one the deferred list will be flushed on); returns the previous setting
extern "C" int mi_tapwgrad_defer_pause(int pause) { const int prev = g_defer_pause; g_defer_pause = pause ? 1 : 0; return prev; }
int g_slab_bf16 = 0;                                       // tapwgrad partial-sum slabs rounded to bf16 (half the slab traffic): off for the layer-op entry points (exact fp32
                                                           // partial sums), switched on by the VAE engine around its backward pass (mi_tapwgrad_slab_bf16); mi_set_tuning key 18
static thread_local int t_slab_bf16 = -1;                  // this thread's override for the pass it is issuing (-1: the process default above)
static inline int slab_bf16_now() { return t_slab_bf16 >= 0 ? t_slab_bf16 : g_slab_bf16; }
extern "C" int mi_tapwgrad_slab_bf16(int on) { const int prev = t_slab_bf16; t_slab_bf16 = on < 0 ? -1 : (on ? 1 : 0); return prev; }
static int tapwgrad_flush_reduces(void* stream);
extern "C" int mi_tapwgrad_flush(void* stream) {
    int rc = tapwgrad_flush_reduces(stream);
    const int nf = g_nfolds;
    g_nfolds = 0;
    for (int i = 0; i < nf && rc == MI_OK; ++i) rc = launch_fold((hipStream_t)stream, g_folds[i]);
    return rc;
}
static int tapwgrad_flush_reduces(void* stream) {
    const int n = g_npending;
    g_npending = 0;
    if (n == 0) return MI_OK;
    if (n == 1 || n > TW_MAX_FUSED) {
        for (int i = 0; i < n; ++i) launch_tiled_reduce((hipStream_t)stream, g_pending[i]);
        return mi_check_launch("reduce_tiled_kernel");
    }
    static thread_local FusedReduceParams f;              // (4 KB: kept off the stack; the launch copies it into the kernel-argument buffer)
    f.n = n; f.first[0] = 0; f.bias.njobs = 0; f.bias.first[0] = 0; f.bias.ovw = 0;
    for (int i = 0; i < n; ++i) {
        const PendingReduce& r = g_pending[i];
        f.q[i] = r.q; f.splits[i] = r.splits; f.ngroups[i] = r.ngroups; f.kind[i] = r.kind; f.ry[i] = (int)reduce_ry(r);
        f.first[i + 1] = f.first[i] + (int)reduce_blocks(r, (unsigned)f.ry[i]);
        if (r.bpart && f.bias.njobs < SR_MAX) sr_add(f.bias, r.bpart, (long long)r.bN, r.bnslab, (long long)r.bN, r.bout);
    }
    for (int i = n; i < TW_MAX_FUSED; ++i) f.first[i + 1] = f.first[n];
    for (int i = f.bias.njobs; i < SR_MAX; ++i) f.bias.first[i + 1] = f.bias.first[f.bias.njobs];
    MI_LAUNCH(reduce_fused_kernel, dim3((unsigned)(f.first[n] + f.bias.first[f.bias.njobs])), dim3(256), 0, (hipStream_t)stream, f);
    return mi_check_launch("reduce_fused_kernel");
}

// tapwgrad (tapwgrad_tile.hpp): bf16 weight gradients of the wide stride-2 layers on raw-staged slot tiles.
// mi_set_tuning key 3 / MI355_TAPWGRAD=0 disables it.
// ---------------------------------------------------------------------------------------------------------------
static int x3_tapwgrad_env() { const char* e = getenv("MI355_X3_TAPWGRAD"); return e ? atoi(e) : 1; }
int g_x3_tapwgrad = x3_tapwgrad_env();                   // split-storage filter gradients on the doubled-channel bf16 kernel: 0 off, 1 conv2 / conv3 (default since round 5: 2.637 -> 2.595 ms per bf16x3 step, two interleaved pairs on one box, gpurun_out/ab_x3_r05e.txt; round 4 measured no difference), 2 every eligible layer (2.870: the wide layers lose); mi_set_tuning key 21
int g_tapwgrad_on = -1;
int g_tapwgrad_split = 1;
static int dwgs_env() { const char* e = getenv("MI355_DWGS"); return (e && e[0] == '0') ? 0 : 1; }
int g_dwgs_on = dwgs_env();                              // LDS-free one-wave-per-tile dense filter gradient (dwgs_tile.hpp, round 5) for bf16 layers of up to 2048 tiles of 64 x 64; mi_set_tuning key 22
static int dense_wgrad_blocks_env() { const char* e = getenv("MI355_DENSE_WGRAD_BLOCKS"); return e ? atoi(e) : 256; }
int g_dense_wgrad_blocks = dense_wgrad_blocks_env();       // dense filter gradients: target block count (row splits); mi_set_tuning key 11
static int nw_depth_env() { const char* e = getenv("MI355_NW_DEPTH"); return e ? atoi(e) : 3; }
int g_nw_depth = nw_depth_env();                          // narrow_wgrad (uint8 conv1 shape): steps in flight per wave (3 | 5 | 6); mi_set_tuning key 19
int g_nw_waves = 12;                                       // narrow_wgrad: waves per block (4 | 8 | 12); mi_set_tuning key 10
int g_tapwgrad_cw = 1;                                     // k = 5 filter gradient: class-wave layout (tapwgrad_cw_kernel); mi_set_tuning key 14
static int dectail_split5_env() { const char* e = getenv("MI355_DECTAIL_SPLIT5"); return (e && e[0] == '1') ? 1 : 0; }      // measured neutral (74.4-77.4 vs 76.2-77.6 us alone, 0.8440 = 0.8440 ms per step): off
int g_dectail_split5 = dectail_split5_env();               // decoder tail: the fifth slot group's loss shared by three waves (dectail_tile.hpp, round 6); mi_set_tuning key 26
int g_dectail_dbg = 0;                                     // ablation mask of the decoder tail's timing instantiation (wrong results); mi_set_tuning key 25
static int tw_ldec_env() { const char* e = getenv("MI355_TW_LDEC"); const int v = e ? atoi(e) : 0; return v < 0 || v > 3 ? 0 : v; }      // bit 0: the 2 x 2-tap kernels, bit 1: the k = 5 class-wave kernel.  Default 0: once the product kernels lost their run-time debug branch (below) the two forms are equal (0.7929 / 0.7932 / 0.7942 ms for 0 / 1 / 3)
int g_tw_ldec = tw_ldec_env();                             // raw-staged filter gradients: a step's DMA rows decoded once per wave, one row per lane (tapwgrad_tile.hpp, round 6); mi_set_tuning key 24
int g_tapwgrad_blocks = 256;                               // tapwgrad: target number of blocks (position splits x block columns); mi_set_tuning key 9
bool tapwgrad_enabled() {
    if (g_tapwgrad_on < 0) { const char* e = getenv("MI355_TAPWGRAD"); g_tapwgrad_on = (e && e[0] == '0') ? 0 : 1; }
    return g_tapwgrad_on != 0;
}

// a: slot-side tensor [B,IH,IW,C]; d: gradient tensor [B,OH,OW,N]; out: dW (conv form HWIO [kh,kw,C,N]; gather form [kh,kw,N,C])
int try_tapwgrad(hipStream_t st, int dtype, int mode, const void* a, const void* d, int B, int IH, int IW, int C, int OH, int OW, int N,
                 int KH, int KW, float* out, void* scratch, long long scratch_bytes, float* dbias) {
    if (!tapwgrad_enabled() || dtype != MI_BF16) return 0;
    if (KH != KW || KH < 3 || KH > 6) return 0;
    if ((((uintptr_t)a) & 15) || (((uintptr_t)d) & 15) || C % 8 != 0 || N % 8 != 0) return 0;
    TapWgradParams q = {};
    const int taps = (KH + 1) / 2;
    int KCB, NEB;
    if (mode == TC_CONV) {
        if (taps != 2 || (4 * C) % 128 != 0 || N % 64 != 0) return 0;
        q.GH = OH + 1; q.GW = OW + 1; q.HY = q.HX = 0; q.KC = 4 * C; q.NE = N; KCB = 128; NEB = 64;
    } else {
        q.HY = q.HX = taps - 1;
        q.GH = (OH + 1) / 2 + q.HY; q.GW = (OW + 1) / 2 + q.HX; q.KC = C; q.NE = 4 * N;
        if (taps == 2) { if (C % 128 != 0 || N % 64 != 0) return 0; KCB = 128; NEB = 64; }
        else { if (C % 64 != 0 || N != 32) return 0; KCB = 64; NEB = 128; }
    }
    if ((taps - 1) * q.GW + taps - 1 > TC_MAXHALO) return 0;
    const long long MP = (long long)B * q.GH * q.GW, a_bytes = (long long)B * IH * IW * C * 2, d_bytes = (long long)B * OH * OW * N * 2;
    if (MP >= (1ll << 30) || !fits_desc(a_bytes) || !fits_desc(d_bytes)) return 0;
    q.a = a; q.a_bytes = (uint32_t)a_bytes; q.d = d; q.d_bytes = (uint32_t)d_bytes;
    q.B = B; q.IH = IH; q.IW = IW; q.C = C; q.OH = OH; q.OW = OW; q.N = N; q.KH = KH; q.KW = KW; q.MP = (int)MP;
    q.nkb = q.KC / KCB;
    const int gy = q.nkb * (q.NE / NEB);
    // (tap, 32-output tile) pairs of one block; gather form k = 5: a parity class only owns the taps its kh / kw reach
    const int ntb = NEB / 32;
    q.npairs = 0;
    for (int tap = 0; tap < taps * taps; ++tap)
        for (int nt = 0; nt < ntb; ++nt) {
            bool valid = true;
            if (mode == TC_GATHER && NEB == 4 * N) {      // tile nt is parity class nt (N == 32)
                const int ta = tap / taps, tb = tap % taps;
                valid = (nt >> 1) + 2 * (q.HY - ta) < KH && (nt & 1) + 2 * (q.HX - tb) < KW;
            }
            if (valid) {
                if (q.npairs >= 32) return 0;             // (k = 6 with every (tap, class) pair live: 36 pairs -- not a layer of this model)
                bool first = true;
                for (int k = 0; k < q.npairs; ++k) if (q.pair_nt[k] == nt) first = false;
                q.pair_tap[q.npairs] = (unsigned char)tap; q.pair_nt[q.npairs] = (unsigned char)nt; q.pair_first[q.npairs] = first ? 1 : 0; ++q.npairs;
            }
        }
    // the first pair of every output tile (the one that also accumulates the tile's bias gradient) goes to the front of the list: pairs
    // 0 .. ntb-1 are slot 0 of waves 0 .. ntb-1, the only slot the kernel keeps a bias accumulator for
    {
        unsigned char t2[32], n2[32], f2[32];
        int k = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int i = 0; i < q.npairs; ++i)
                if ((q.pair_first[i] != 0) == (pass == 0)) { t2[k] = q.pair_tap[i]; n2[k] = q.pair_nt[i]; f2[k] = q.pair_first[i]; ++k; }
        for (int i = 0; i < q.npairs; ++i) { q.pair_tap[i] = t2[i]; q.pair_nt[i] = n2[i]; q.pair_first[i] = f2[i]; }
    }
    q.dbias = dbias;
    int splits = g_tapwgrad_blocks / gy; if (splits < 1) splits = 1;
    long long pps = (MP + splits - 1) / splits; pps = (pps + TW_BP - 1) / TW_BP * TW_BP;
    splits = (int)((MP + pps - 1) / pps);
    q.pos_per_split = (int)pps;
    q.div_g = make_fastdiv(q.GH); q.div_gw = make_fastdiv(q.GW); q.div_n = make_fastdiv(N);
    q.div_2c = make_fastdiv(2 * C); q.div_c = make_fastdiv(C);
    q.out = out;
    q.trace = g_trace; q.trace_cap = g_trace_cap; q.dbg_cheap_addr = (g_wgrad_skip >= 2 && g_wgrad_skip <= 5) ? g_wgrad_skip : 0;
    // partial sums per split in the caller's scratch (accumulator-order 16-byte stores + one reduce that does the dW index decode)
    // when it is large enough; otherwise fp32 atomics straight into dW (~1 element per clock per CU: 35-45 % of the kernel at 256 splits)
    const int kt_tiles = taps == 2 ? 4 : 2;
    const long long slab_floats = (long long)gy * q.npairs * kt_tiles * 1024;
    q.slabs = nullptr; q.slab_stride = slab_floats; q.slab_bf16 = slab_bf16_now() ? 1 : 0;
    const bool split = g_tapwgrad_split && taps == 2 && q.npairs == 8;   // wave = (tap, position half): fewer LDS reads per MFMA
    // behind the slabs: the bias-gradient partial sums of every position split (x 2 position halves in the split layout), summed in a fixed order with the slabs
    const long long slab_bytes = ((long long)splits * slab_floats * (q.slab_bf16 ? 2 : 4) + 255) / 256 * 256;
    q.bias_part = nullptr; q.bias_nh = split ? 2 : 1;
    const long long bias_bytes = dbias ? (long long)splits * q.bias_nh * q.NE * 4 : 0;
    if (scratch && splits > 1 && (((uintptr_t)scratch) & 15) == 0 && scratch_bytes >= slab_bytes + bias_bytes && slab_floats < (1ll << 29) && (!dbias || N <= 256)) {
        q.slabs = (float*)scratch;
        if (dbias) q.bias_part = (float*)((char*)scratch + slab_bytes);
    }
    q.gx = splits; q.gy = gy;
    dim3 g((unsigned)((splits + 7) / 8 * 8 * gy), 1, 1);   // 1-D: the column blocks of a position split share an XCD (tapwgrad_tile.hpp)
    const bool ldec = (g_tw_ldec & 1) && !q.dbg_cheap_addr && !q.trace, ldec_cw = (g_tw_ldec & 2) && !q.dbg_cheap_addr && !q.trace;
    const bool dbg = q.dbg_cheap_addr != 0;                // timing instantiations (tools/wgrad_ablate.py): their own kernels, the split layouts and the k = 5 kernel only
    if (mode == TC_CONV) {
        if (split && dbg) MI_LAUNCH((tapwgrad_kernel<TC_CONV, 2, 4, 2, 2, true, false, true>), g, dim3(TW_NT), 0, st, q);
        else if (split && ldec) MI_LAUNCH((tapwgrad_kernel<TC_CONV, 2, 4, 2, 2, true, true>), g, dim3(TW_NT), 0, st, q);
        else if (split) MI_LAUNCH((tapwgrad_kernel<TC_CONV, 2, 4, 2, 2, true>), g, dim3(TW_NT), 0, st, q);
        else MI_LAUNCH((tapwgrad_kernel<TC_CONV, 2, 4, 2, 1>), g, dim3(TW_NT), 0, st, q);
    } else if (taps == 2) {
        if (split && dbg) MI_LAUNCH((tapwgrad_kernel<TC_GATHER, 2, 4, 2, 2, true, false, true>), g, dim3(TW_NT), 0, st, q);
        else if (split && ldec) MI_LAUNCH((tapwgrad_kernel<TC_GATHER, 2, 4, 2, 2, true, true>), g, dim3(TW_NT), 0, st, q);
        else if (split) MI_LAUNCH((tapwgrad_kernel<TC_GATHER, 2, 4, 2, 2, true>), g, dim3(TW_NT), 0, st, q);
        else MI_LAUNCH((tapwgrad_kernel<TC_GATHER, 2, 4, 2, 1>), g, dim3(TW_NT), 0, st, q);
    }
    else {
        if (q.npairs > 32) return 0;
        // k = 5 with caller scratch: a wave per (parity class, tap row), the shifted slot fragments formed in registers (mi_set_tuning key 14 = 0: the pair layout)
        if (g_tapwgrad_cw && KH == 5 && C == 64 && q.NE == 128 && q.KC == 64 && q.slabs) {
            if (dbg) MI_LAUNCH((tapwgrad_cw_kernel<false, true>), g, dim3(TWC_NT), 0, st, q);
            else if (ldec_cw) MI_LAUNCH(tapwgrad_cw_kernel<true>, g, dim3(TWC_NT), 0, st, q); else MI_LAUNCH(tapwgrad_cw_kernel<false>, g, dim3(TWC_NT), 0, st, q);
        }
        else MI_LAUNCH((tapwgrad_kernel<TC_GATHER, 3, 2, 4, 4>), g, dim3(TW_NT), 0, st, q);
    }
    int rc = mi_check_launch("tapwgrad_kernel");
    if (rc == MI_OK && q.slabs) {
        PendingReduce r;
        r.q = q; r.splits = splits; r.ngroups = (int)(slab_floats / 4); r.kind = mode == TC_CONV ? 0 : (taps == 2 ? 1 : 2);
        r.bpart = q.bias_part; r.bout = dbias; r.bnslab = splits * q.bias_nh * (q.NE / N); r.bN = N;
        if (g_defer_reduces && !g_defer_pause && g_npending < 16) g_pending[g_npending++] = r;
        else { launch_tiled_reduce(st, r); rc = mi_check_launch("reduce_tiled_kernel"); }
    }
    return rc == MI_OK ? 1 : rc;
}

// MI_BF16X3 filter gradients on the bf16 kernel (see PendingFold): scratch = [dW' (4 x the filter, fp32) | bias' (2 N) | slabs ...]
int try_tapwgrad_split(hipStream_t st, int mode, const void* a, const void* d, int B, int IH, int IW, int C, int OH, int OW, int N,
                       int KH, int KW, float* out, void* scratch, long long scratch_bytes, float* dbias) {
    const int on = g_x3_tapwgrad;
    if (!on || !scratch || (((uintptr_t)scratch) & 255)) return 0;
    const long long nw = (long long)KH * KW * C * N;
    const long long tmp_bytes = ((4 * nw + 2 * N) * 4 + 255) / 256 * 256;
    if (scratch_bytes < tmp_bytes + (1 << 20)) return 0;
    // cheap shape test first (the same conditions try_tapwgrad applies to the doubled channel counts): nothing is touched for a layer it will not take
    const int taps = (KH + 1) / 2;
    if (KH != KW || taps != 2) return 0;                    // (the k = 5 gather kernels are built for 32 output channels per parity class: 2 N = 64 does not fit)
    if (mode == TC_CONV ? ((8 * C) % 128 != 0 || (2 * N) % 64 != 0) : ((2 * C) % 128 != 0 || (2 * N) % 64 != 0)) return 0;
    // measured per layer at batch 512 (us, doubled-channel bf16 kernel vs the first-generation split kernel): conv2 125 / 164, conv3 122 / 136, deconv2 150 / 150, conv4 187 / 86,
    // deconv1 227 / 91 -- the wide layers end up with 64 column blocks and four position splits.  MI355_X3_TAPWGRAD=2 takes every eligible layer (A/B).
    if (on != 2 && (mode != TC_CONV || (long long)C * N > 64 * 128)) return 0;
    float* tmp = (float*)scratch; float* tb = tmp + 4 * nw;
    if (hipMemsetAsync(tmp, 0, (size_t)((4 * nw + 2 * N) * 4), st) != hipSuccess) return mi_fail(MI_ERR_LAUNCH, "try_tapwgrad_split: memset failed");
    const int r = try_tapwgrad(st, MI_BF16, mode, a, d, B, IH, IW, 2 * C, OH, OW, 2 * N, KH, KW, tmp, (char*)scratch + tmp_bytes, scratch_bytes - tmp_bytes, dbias ? tb : nullptr);
    if (r <= 0) return r;
    PendingFold f = {tmp, out, tb, dbias, KH * KW, mode == TC_CONV ? C : N, mode == TC_CONV ? N : C, N};
    if (g_defer_reduces && !g_defer_pause && g_nfolds < 16) { g_folds[g_nfolds++] = f; return 1; }
    const int rc = launch_fold(st, f);
    return rc == MI_OK ? 1 : rc;
}

static bool narrow_lean_enabled() {                       // MI355_NARROW_LEAN=0: the first-generation narrow-layer kernels (A/B runs)
    static int on = -1;
    if (on < 0) { const char* e = getenv("MI355_NARROW_LEAN"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}

// gather-form transposed conv into a narrow output (narrow_tile.hpp): 4N <= 32 output columns, 64- or 128-byte input pixels
template <typename T, int TAPS, int CPR>
int launch_gather_narrow(hipStream_t st, const TapParams& q) {
    dim3 g((q.MP + GN_BMT - 1) / GN_BMT);
    if (q.N == 3 && sizeof(T) == 2 && q.labels && q.loss_kind == 0 && narrow_lean_enabled()) MI_LAUNCH((gather_narrow_kernel<T, TAPS, CPR, 3, true>), g, dim3(GN_NT), 0, st, q);
    else if (q.N == 3) MI_LAUNCH((gather_narrow_kernel<T, TAPS, CPR, 3>), g, dim3(GN_NT), 0, st, q);
    else if (q.N == 1) MI_LAUNCH((gather_narrow_kernel<T, TAPS, CPR, 1>), g, dim3(GN_NT), 0, st, q);
    else MI_LAUNCH((gather_narrow_kernel<T, TAPS, CPR, 0>), g, dim3(GN_NT), 0, st, q);
    return mi_check_launch("gather_narrow_kernel");
}

int g_narrow_on = -1;
bool narrow_enabled() {
    if (g_narrow_on < 0) { const char* e = getenv("MI355_NARROW"); g_narrow_on = (e && e[0] == '0') ? 0 : 1; }
    return g_narrow_on != 0;
}

struct NarrowLoss {                                       // optional fused reconstruction loss of gather_narrow_kernel
    const float* labels; const int* idx; long long stride; int kind; float inv_b; void* dlogits; float* lpart; float* bpart; int cap; int nblocks;
    int labels_u8;                                        // labels are raw camera bytes (value k / 255), stride in bytes
};

int try_gather_narrow(hipStream_t st, int dtype, const void* a, const void* w, int B, int IH, int IW, int C, int OH, int OW, int N,
                      int KH, int KW, void* out, const float* bias, const void* mask, int relu, NarrowLoss* loss = nullptr) {
    if (!narrow_enabled() || mask || 4 * N > 32 || KH != KW || KH < 3 || KH > 6) return 0;
    const int esz = esz_of(dtype);
    const int pa = C * esz;
    if ((pa != 64 && pa != 128) || (((uintptr_t)a) & 15) || (((uintptr_t)w) & 15) || (((uintptr_t)out) & 3) || (2 * N * esz) % 4 != 0) return 0;
    if (!out && !loss) return 0;                          // out == nullptr is the fused-loss form (loss partials / dlogits only)
    TapParams q = {};
    q.TH = q.TW = (KH + 1) / 2; q.HY = q.HX = q.TH - 1;
    q.GH = (OH + 1) / 2 + q.HY; q.GW = (OW + 1) / 2 + q.HX; q.KC = C; q.NE = 4 * N;
    if ((q.TH - 1) * q.GW + q.TW - 1 > TC_MAXHALO) return 0;
    const long long MP = (long long)B * q.GH * q.GW, a_bytes = (long long)B * IH * IW * C * esz;
    if (MP >= (1ll << 30) || !fits_desc(a_bytes)) return 0;
    q.a = a; q.a_bytes = (uint32_t)a_bytes; q.b = w; q.b_bytes = 0;
    q.B = B; q.IH = IH; q.IW = IW; q.C = C; q.OH = OH; q.OW = OW; q.N = N; q.KH = KH; q.KW = KW; q.MP = (int)MP;
    q.div_g = make_fastdiv(q.GH); q.div_gw = make_fastdiv(q.GW); q.div_n = make_fastdiv(N);
    q.out = out; q.bias = bias; q.mask = nullptr; q.relu = relu;
    if (loss) {
        const int nblk = (int)((MP + GN_BMT - 1) / GN_BMT);
        const bool lab_ok = loss->labels_u8 ? ((((uintptr_t)loss->labels) & 1) == 0 && loss->stride % 2 == 0)
                                            : ((((uintptr_t)loss->labels) & 7) == 0 && (loss->stride * 4) % 8 == 0);
        if ((N != 1 && N != 3) || (OW & 1) || nblk > loss->cap || !lab_ok ||
            (loss->dlogits && (((uintptr_t)loss->dlogits) & 3))) return 0;
        q.labels = loss->labels; q.lab_idx = loss->idx; q.lab_stride = loss->stride; q.loss_kind = loss->kind; q.inv_b = loss->inv_b; q.lab_u8 = loss->labels_u8;
        q.dlogits = loss->dlogits; q.lpart = loss->lpart; q.bpart = loss->bpart;
        loss->nblocks = nblk;
    }
    int rc;
    if (dtype == MI_F32) {
        if (pa != 128) return 0;
        rc = q.TH == 2 ? launch_gather_narrow<float, 2, 8>(st, q) : launch_gather_narrow<float, 3, 8>(st, q);
    } else if (dtype == MI_BF16X3) {
        if (pa != 128) return 0;
        rc = q.TH == 2 ? launch_gather_narrow<split_t, 2, 8>(st, q) : launch_gather_narrow<split_t, 3, 8>(st, q);
    } else if (pa == 64) rc = q.TH == 2 ? launch_gather_narrow<bf16_t, 2, 4>(st, q) : launch_gather_narrow<bf16_t, 3, 4>(st, q);
    else rc = q.TH == 2 ? launch_gather_narrow<bf16_t, 2, 8>(st, q) : launch_gather_narrow<bf16_t, 3, 8>(st, q);
    return rc == MI_OK ? 1 : rc;
}

// filter gradient of a k x k, s2 layer with a 1..3-channel narrow side and a 32-channel wide side (narrow_tile.hpp), bf16 wide tensor.
// narrow: [*,IH,IW,Cs] (fp32 frames, optionally gathered, or bf16); wide: [B,OH,OW,32]; out [KH*KW*Cs][32] fp32 (+=)
int try_narrow_wgrad(hipStream_t st, int dtype, const void* narrow, int narrow_f32, const int* frame_idx, const void* wide,
                     int B, int IH, int IW, int Cs, int OH, int OW, int Nwide, int KH, int KW, float* out, float* dbias,
                     void* scratch, long long scratch_bytes) {
    if (!narrow_enabled() || dtype != MI_BF16) return 0;
    const int run = KW * Cs;
    if (Nwide != 32 || KH > 4 || run > 12 || run % 4 != 0 || KH * run > 64 || (((uintptr_t)wide) & 15)) return 0;
    const int nsz = narrow_f32 == 2 ? 1 : (narrow_f32 ? 4 : 2);       // narrow_f32: 0 = bf16, 1 = fp32, 2 = uint8 camera bytes (value k / 255)
    if ((((uintptr_t)narrow) & (2 * nsz - 1)) || ((long long)IW * Cs * nsz) % (2 * nsz) != 0) return 0;
    if ((2 * Cs * nsz) % (2 * nsz) != 0 || ((long long)IH * IW * Cs * nsz) % (2 * nsz) != 0) return 0;
    const long long M = (long long)B * OH * OW, s_bytes = M * 32 * 2;
    if (M >= (1ll << 26) || !fits_desc(s_bytes)) return 0;
    NarrowWgradParams q = {};
    q.src = narrow; q.frame_idx = frame_idx; q.frame_stride = (long long)IH * IW * Cs;
    q.s = wide; q.s_bytes = (uint32_t)s_bytes;
    q.B = B; q.IH = IH; q.IW = IW; q.Cs = Cs; q.OH = OH; q.OW = OW; q.KH = KH; q.KW = KW; q.M = (int)M;
    static int wpc = -1;                                   // resident waves per CU the grid is sized for (131 registers -> 3 per SIMD)
    if (wpc < 0) { const char* e = getenv("MI355_NW_WAVES"); wpc = e ? atoi(e) : 12; }
    long long nwave = 256ll * wpc;                         // one wave-range per resident wave: a single round, no tail
    long long ppw = (M + nwave - 1) / nwave; ppw = (ppw + NW_BP - 1) / NW_BP * NW_BP;
    if ((long long)OH * OW < NW_BP) return 0;
    if (ppw > 2ll * OH * OW) ppw = 2ll * OH * OW / NW_BP * NW_BP;   // a wave's range touches at most 3 frames (their indices are looked up once)
    nwave = (M + ppw - 1) / ppw;
    const int nwv = (narrow_f32 == 2 || g_nw_waves >= 12) ? 12 : g_nw_waves >= 8 ? 8 : 4;
    const int blocks = (int)((nwave + nwv - 1) / nwv);
    q.pix_per_block = (int)ppw;
    q.div_ohw = make_fastdiv(OH * OW); q.div_ow = make_fastdiv(OW);
    q.out = out; q.dbias = dbias; q.dbg_skip_out = g_wgrad_skip == 1;
    q.slabs = (scratch && (((uintptr_t)scratch) & 15) == 0 && (((uintptr_t)out) & 15) == 0 && (!dbias || (((uintptr_t)dbias) & 15) == 0) &&
               scratch_bytes >= (long long)blocks * NW_SLAB * 4) ? (float*)scratch : nullptr;
    const bool g3 = KW * q.Cs == 12;                     // the 4 x 4 x 3-channel layers: fixed-shape loop body (one basic block)
#define NW_LAUNCH(TS_, NWV_) do { \
        const dim3 g((unsigned)((nwave + NWV_ - 1) / NWV_)), t(NWV_ * 64); \
        if (g3 && q.dbias) MI_LAUNCH((narrow_wgrad_kernel<TS_, 3, 1, NWV_>), g, t, 0, st, q); \
        else if (g3) MI_LAUNCH((narrow_wgrad_kernel<TS_, 3, 0, NWV_>), g, t, 0, st, q); \
        else MI_LAUNCH((narrow_wgrad_kernel<TS_, 0, -1, NWV_>), g, t, 0, st, q); } while (0)
    if (narrow_f32 == 2 && g3 && q.dbias && g_nw_depth == 6) {
        MI_LAUNCH((narrow_wgrad_kernel<unsigned char, 3, 1, 12, 6>), dim3((unsigned)((nwave + 11) / 12)), dim3(768), 0, st, q);
    } else if (narrow_f32 == 2 && g3 && q.dbias && g_nw_depth == 5) {
        MI_LAUNCH((narrow_wgrad_kernel<unsigned char, 3, 1, 12, 5>), dim3((unsigned)((nwave + 11) / 12)), dim3(768), 0, st, q);
    } else if (narrow_f32 == 2) NW_LAUNCH(unsigned char, 12);
    else if (narrow_f32) { if (g_nw_waves >= 12) NW_LAUNCH(float, 12); else if (g_nw_waves >= 8) NW_LAUNCH(float, 8); else NW_LAUNCH(float, 4); }
    else { if (g_nw_waves >= 12) NW_LAUNCH(bf16_t, 12); else if (g_nw_waves >= 8) NW_LAUNCH(bf16_t, 8); else NW_LAUNCH(bf16_t, 4); }
#undef NW_LAUNCH
    int rc = mi_check_launch("narrow_wgrad_kernel");
    if (rc == MI_OK && q.slabs) {
        const long long n_out = (long long)KH * run * 32;
        rc = mi_reduce_slabs(st, q.slabs, (long long)NW_SLAB, blocks, n_out, out);
        if (rc == MI_OK && dbias) rc = mi_reduce_slabs(st, q.slabs + 64 * 32, (long long)NW_SLAB, blocks, 32ll, dbias);
    }
    return rc == MI_OK ? 1 : rc;
}

// conv k x k, s2 from a 1..3-channel tensor into exactly 32 channels (narrow_tile.hpp): conv1 fwd, deconv4 dgrad
int try_narrow_conv(hipStream_t st, int dtype, const void* src, int src_f32, const int* frame_idx, const void* wt, int B, int IH, int IW, int Cs,
                    int KH, int KW, int Cout, const float* bias, int relu, const void* mask, void* out, void* bits_out = nullptr, const void* mask_bits = nullptr) {
    if (!narrow_enabled() || Cout != 32 || KH != KW || KH > 4) return 0;
    const int run = KW * Cs, K = KH * run;
    if (run % 4 != 0 || K > 48) return 0;
    const int ssz = src_f32 == 2 ? 1 : (src_f32 ? 4 : 2), esz = esz_of(dtype);   // src_f32: 0 = bf16, 1 = fp32, 2 = uint8 camera bytes, 3 = split storage
    if (dtype == MI_F32 && src_f32 != 1) return 0;
    if (dtype == MI_BF16X3 && src_f32 != 1 && src_f32 != 3) return 0;
    if (dtype == MI_BF16 && src_f32 == 3) return 0;
    if ((((uintptr_t)src) & (2 * ssz - 1)) || ((long long)IW * Cs * ssz) % (2 * ssz) != 0 || (2 * Cs * ssz) % (2 * ssz) != 0 || ((long long)IH * IW * Cs * ssz) % (2 * ssz) != 0) return 0;
    if ((((uintptr_t)wt) & 15) || (K * esz) % 16 != 0 || (((uintptr_t)out) & 15) || (mask && (((uintptr_t)mask) & 15)) || (bias && (((uintptr_t)bias) & 15))) return 0;
    const int OH = (IH - KH) / 2 + 1, OW = (IW - KW) / 2 + 1;
    const long long M = (long long)B * OH * OW;
    if (M >= (1ll << 30)) return 0;
    if ((long long)OH * OW < 32) return 0;               // a wave's 32 pixels must not span more than two frames
    NarrowConvParams q = {};
    q.src = src; q.frame_idx = frame_idx; q.frame_stride = (long long)IH * IW * Cs; q.w = wt;
    q.B = B; q.IH = IH; q.IW = IW; q.Cs = Cs; q.OH = OH; q.OW = OW; q.KH = KH; q.KW = KW; q.M = (int)M;
    q.div_ohw = make_fastdiv(OH * OW); q.div_ow = make_fastdiv(OW); q.div_g3 = make_fastdiv(run / 4);
    q.bias = bias; q.relu = relu; q.mask = mask; q.out = out;
    q.bits_out = (dtype == MI_BF16 && relu) ? (uint32_t*)bits_out : nullptr; q.mask_bits = dtype == MI_BF16 ? (const uint32_t*)mask_bits : nullptr;
    if (bits_out && !q.bits_out) return 0;                // the caller asked for ReLU bits this kernel cannot write
    dim3 g((unsigned)((M + 127) / 128));
    // the model's geometry (K = 48 as 4 rows of 12) in bf16: the instruction-lean form; mode 0 = bias + ReLU (+ bit words), mode 1 = masked by bit words
    const int lean = (dtype == MI_BF16 && KH == 4 && run == 12 && M * 64 < (1ll << 31) && narrow_lean_enabled())
                         ? ((bias && relu && !mask && !mask_bits) ? 1 : ((!bias && !relu && mask_bits && !bits_out) ? 2 : 0)) : 0;
    if (lean) {
        // persistent: as many four-wave blocks as stay resident (5-8 per compute unit by register count), each wave walks tiles with a one-tile prefetch
        const long long tiles4 = (M + 127) / 128;
        static int resident[3][2];                        // [source type][mode]: blocks the device holds at once (queried once)
        auto grid_of = [&](const void* fn, int si, int mi) {
            if (!resident[si][mi]) {
                int per_cu = 0, dev = 0, cus = 256;
                hipDeviceProp_t pr;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 4;
                if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
                resident[si][mi] = per_cu * cus;
            }
            return dim3((unsigned)(tiles4 < resident[si][mi] ? tiles4 : resident[si][mi]));
        };
#define NC48(TS, SI) do { if (lean == 1) MI_LAUNCH((narrow_conv48_kernel<TS, 0>), grid_of((const void*)narrow_conv48_kernel<TS, 0>, SI, 0), dim3(256), 0, st, q); \
                          else MI_LAUNCH((narrow_conv48_kernel<TS, 1>), grid_of((const void*)narrow_conv48_kernel<TS, 1>, SI, 1), dim3(256), 0, st, q); } while (0)
        if (src_f32 == 2) NC48(unsigned char, 0); else if (src_f32) NC48(float, 1); else NC48(bf16_t, 2);
#undef NC48
        const int rc = mi_check_launch("narrow_conv48_kernel");
        return rc == MI_OK ? 1 : rc;
    }
    if (dtype == MI_F32) MI_LAUNCH((narrow_conv_kernel<float, float>), g, dim3(256), 0, st, q);
    else if (dtype == MI_BF16X3 && src_f32 == 1) MI_LAUNCH((narrow_conv_kernel<split_t, float>), g, dim3(256), 0, st, q);
    else if (dtype == MI_BF16X3) MI_LAUNCH((narrow_conv_kernel<split_t, split_t>), g, dim3(256), 0, st, q);
    else if (src_f32 == 2) MI_LAUNCH((narrow_conv_kernel<bf16_t, unsigned char>), g, dim3(256), 0, st, q);
    else if (src_f32) MI_LAUNCH((narrow_conv_kernel<bf16_t, float>), g, dim3(256), 0, st, q);
    else MI_LAUNCH((narrow_conv_kernel<bf16_t, bf16_t>), g, dim3(256), 0, st, q);
    const int rc = mi_check_launch("narrow_conv_kernel");
    return rc == MI_OK ? 1 : rc;
}

// conv-form (A_CONV x B_NK).  Returns 1 if launched, 0 if not eligible, <0 on error.
int try_conv_form_gemm2(hipStream_t st, int dtype, const GemmParams& p, int gz = 1) {
    if (p.a_frame_idx) return 0;
    if (p.ksplit_len > 0 && (!g_gemm2_splitk || p.stride != 1 || p.KH != 1 || p.KW != 1 || (p.ksplit_len * esz_of(dtype)) % 128 != 0)) return 0;   // split-K: dense layers, whole 128-byte stages per split
    if (p.stride == 2 && !p.out_f32) {
        const int r5 = mi_try_rwconv_conv(st, dtype, p.a, p.b, p.nbatch, p.IH, p.IW, p.C, p.OH, p.OW, p.N, p.KH, p.KW, p.ldb, p.out, p.bias, p.mask, p.relu);
        if (r5 != 0) return r5;
        const int r3 = try_tapconv(st, dtype, TC_CONV, p.a, p.b, p.nbatch, p.IH, p.IW, p.C, p.OH, p.OW, p.N, p.KH, p.KW, p.ldb, p.out, p.bias, p.mask, p.relu);
        if (r3 != 0) return r3;
    }
    if (!gemm2_enabled()) return 0;
    const int esz = esz_of(dtype);
    const long long a_bytes = (long long)p.nbatch * p.a_frame_stride * esz, b_bytes = (long long)p.N * p.ldb * esz;
    if ((p.C * esz) % 16 != 0 || (((uintptr_t)p.a) & 15) || (((uintptr_t)p.b) & 15) || (p.ldb * esz) % 16 != 0 || p.ldb < p.K) return 0;
    if (!fits_desc(a_bytes) || !fits_desc(b_bytes)) return 0;
    Gemm2Params q = {};
    q.a = p.a; q.a_bytes = (uint32_t)a_bytes; q.b = p.b; q.b_bytes = (uint32_t)b_bytes;
    q.IH = p.IH; q.IW = p.IW; q.C = p.C; q.OH = p.OH; q.OW = p.OW; q.KH = p.KH; q.KW = p.KW; q.stride = p.stride;
    q.M = p.M; q.N = p.N; q.K = p.K; q.nbatch = p.nbatch;
    q.run = p.KW * p.C; q.div_run = make_fastdiv(q.run); q.div_ohw = p.div_ohw; q.div_ow = p.div_ow;
    q.ldb = p.ldb;
    q.ksplit_len = p.ksplit_len > 0 ? p.ksplit_len : 0;
    {   // dense layers (1 x 1 "convolutions" of one position per row): XCD-contiguous numbering of the whole grid (MI355_GEMM2_REMAP3=0: x only, as for the convolutions)
        static int remap3_on = -1;
        if (remap3_on < 0) { const char* e = getenv("MI355_GEMM2_REMAP3"); remap3_on = (e && e[0] == '0') ? 0 : 1; }
        q.remap3 = (remap3_on && p.stride == 1 && p.KH == 1 && p.KW == 1) ? 1 : 0;
    }
    copy_epilogue(q, p);
    int rc = dtype == MI_F32 ? launch_gemm2_tiles<float, A_CONV, B_NK, false>(st, q, q.M, gz)
           : dtype == MI_BF16X3 ? launch_gemm2_tiles<split_t, A_CONV, B_NK, false>(st, q, q.M, gz)
                             : launch_gemm2_tiles<bf16_t, A_CONV, B_NK, false>(st, q, q.M, gz);
    return rc == MI_OK ? 1 : rc;
}

// gather-form transposed conv (A_DECONV x B_DECONV); p already carries the class geometry from deconv_form_gemm
int try_deconv_form_gemm2(hipStream_t st, int dtype, const GemmParams& p, int maxM) {
    if (!gemm2_enabled()) return 0;
    if (p.N <= 32) return 0;      // measured: narrow gather-form layers (deconv3/4 fwd, conv2 dgrad) are not faster on the DMA tiles
    const int esz = esz_of(dtype);
    const long long a_bytes = (long long)p.nbatch * p.a_frame_stride * esz, b_bytes = (long long)p.KH * p.KW * p.N * p.C * esz;
    if ((p.C * esz) % 16 != 0 || p.KH > 6 || p.KW > 6 || !fits_desc(a_bytes) || !fits_desc(b_bytes)) return 0;
    Gemm2Params q = {};
    q.a = p.a; q.a_bytes = (uint32_t)a_bytes; q.b = p.b; q.b_bytes = (uint32_t)b_bytes;
    q.IH = p.IH; q.IW = p.IW; q.C = p.C; q.OH = p.OH; q.OW = p.OW; q.KH = p.KH; q.KW = p.KW; q.stride = 2;
    q.M = 0; q.N = p.N; q.K = 0; q.nbatch = p.nbatch;
    for (int i = 0; i < 2; ++i) { q.OHc[i] = p.OHc[i]; q.OWc[i] = p.OWc[i]; q.Th[i] = p.Th[i]; q.Tw[i] = p.Tw[i]; q.dc_tw[i] = p.dc_tw[i]; }
    for (int i = 0; i < 4; ++i) { q.dc_ohw[i] = p.dc_ohw[i]; q.dc_ow[i] = p.dc_ow[i]; }
    q.dc_c = p.dc_c;
    copy_epilogue(q, p);
    const bool utap = (p.C * esz) % 128 == 0;
    int rc;
    if (dtype == MI_F32) rc = utap ? launch_gemm2_tiles<float, A_DECONV, B_DECONV, true>(st, q, maxM, 4) : launch_gemm2_tiles<float, A_DECONV, B_DECONV, false>(st, q, maxM, 4);
    else if (dtype == MI_BF16X3) rc = utap ? launch_gemm2_tiles<split_t, A_DECONV, B_DECONV, true>(st, q, maxM, 4) : launch_gemm2_tiles<split_t, A_DECONV, B_DECONV, false>(st, q, maxM, 4);
    else rc = utap ? launch_gemm2_tiles<bf16_t, A_DECONV, B_DECONV, true>(st, q, maxM, 4) : launch_gemm2_tiles<bf16_t, A_DECONV, B_DECONV, false>(st, q, maxM, 4);
    return rc == MI_OK ? 1 : rc;
}

// A_CONV GEMM with dtype/vector dispatch. in_f32: the A tensor is fp32 in HBM even when T = bf16 (input frames).
template <int BMODE>
int conv_form_gemm(hipStream_t st, int dtype, int in_f32, GemmParams& p, int gz) {
    const int C = p.C;
    if (BMODE == B_NK && !(in_f32 && dtype != MI_F32)) {
        const int r2 = try_conv_form_gemm2(st, dtype, p, gz);
        if (r2 != 0) return r2 > 0 ? MI_OK : r2;
    }
    const bool a16 = (((uintptr_t)p.a) & 15) == 0;
    if (dtype == MI_F32) {
        if (!p.merged && C % 4 == 0 && a16) return launch_gemm_bn<float, float, A_CONV, BMODE, 4, 16>(st, p, p.M, gz);
        if (p.merged && (p.KW * C) % 4 == 0 && (p.IW * C) % 2 == 0 && (p.stride * C) % 2 == 0 && (p.a_frame_stride % 2) == 0)
            return launch_gemm_bn<float, float, A_CONV, BMODE, 4, 8>(st, p, p.M, gz);
        return mi_fail(MI_ERR_SHAPE, "conv-form gemm (f32): channel count / alignment not supported");
    }
    if (dtype == MI_BF16X3) {                            // split storage: 4-byte elements, same vector rules as fp32; in_f32 = fp32 frames as the A tensor
        if (in_f32) {
            if (!p.merged && C % 4 == 0 && a16) return launch_gemm_bn<split_t, float, A_CONV, BMODE, 4, 16>(st, p, p.M, gz);
            if (p.merged && (p.KW * C) % 4 == 0 && (p.IW * C) % 2 == 0 && (p.stride * C) % 2 == 0 && (p.a_frame_stride % 2) == 0)
                return launch_gemm_bn<split_t, float, A_CONV, BMODE, 4, 8>(st, p, p.M, gz);
        } else {
            if (!p.merged && C % 4 == 0 && a16) return launch_gemm_bn<split_t, split_t, A_CONV, BMODE, 4, 16>(st, p, p.M, gz);
            if (p.merged && (p.KW * C) % 4 == 0 && (p.IW * C) % 2 == 0 && (p.stride * C) % 2 == 0 && (p.a_frame_stride % 2) == 0)
                return launch_gemm_bn<split_t, split_t, A_CONV, BMODE, 4, 8>(st, p, p.M, gz);
        }
        return mi_fail(MI_ERR_SHAPE, "conv-form gemm (split): channel count / alignment not supported");
    }
    if (in_f32) {
        if (!p.merged && C % 4 == 0 && a16) return launch_gemm_bn<bf16_t, float, A_CONV, BMODE, 4, 16>(st, p, p.M, gz);
        if (p.merged && (p.KW * C) % 4 == 0 && (p.IW * C) % 2 == 0 && (p.stride * C) % 2 == 0 && (p.a_frame_stride % 2) == 0)
            return launch_gemm_bn<bf16_t, float, A_CONV, BMODE, 4, 8>(st, p, p.M, gz);
        return mi_fail(MI_ERR_SHAPE, "conv-form gemm (bf16, fp32 input): channel count / alignment not supported");
    }
    if (!p.merged && C % 8 == 0 && a16) return launch_gemm_bn<bf16_t, bf16_t, A_CONV, BMODE, 8, 16>(st, p, p.M, gz);
    if (p.merged && (p.KW * C) % 4 == 0 && (p.IW * C) % 2 == 0 && (p.stride * C) % 2 == 0 && (p.a_frame_stride % 2) == 0)
        return launch_gemm_bn<bf16_t, bf16_t, A_CONV, BMODE, 4, 4>(st, p, p.M, gz);
    return mi_fail(MI_ERR_SHAPE, "conv-form gemm (bf16): channel count / alignment not supported");
}

bool vec_ok(const void* ptr, long long ld, int dtype) {
    const int vb = dtype == MI_BF16 ? 8 : 4;
    return ((((uintptr_t)ptr) & 15) == 0) && (ld % vb == 0);
}

int deconv_form_gemm(hipStream_t st, int dtype, GemmParams& p, int B, int IH, int IW, int C, int OH, int OW, int N, int KH, int KW) {
    // gather-form stride-2 transposed conv; output may be larger than the natural (IH-1)*2+KH (extra rows/cols get bias only)
    p.IH = IH; p.IW = IW; p.C = C; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.stride = 2;
    p.nbatch = B; p.N = N; p.M = 0; p.K = 0;
    p.a_frame_stride = (long long)IH * IW * C;
    p.a_frame_idx = nullptr;
    p.ksplit_len = 0;
    int maxM = 0;
    for (int ph = 0; ph < 2; ++ph) {
        p.OHc[ph] = (OH - ph + 1) / 2; p.Th[ph] = (KH - ph + 1) / 2;
        p.OWc[ph] = (OW - ph + 1) / 2; p.Tw[ph] = (KW - ph + 1) / 2;
        p.dc_tw[ph] = make_fastdiv(p.Tw[ph] > 0 ? p.Tw[ph] : 1);
    }
    for (int c = 0; c < 4; ++c) {
        const int oh = p.OHc[c >> 1], ow = p.OWc[c & 1];
        p.dc_ohw[c] = make_fastdiv(oh * ow > 0 ? oh * ow : 1);
        p.dc_ow[c] = make_fastdiv(ow > 0 ? ow : 1);
        if (B * oh * ow > maxM) maxM = B * oh * ow;
    }
    p.dc_c = make_fastdiv(C);
    const int vb = dtype == MI_BF16 ? 8 : 4;
    if (C % vb != 0 || (((uintptr_t)p.a) & 15) || (((uintptr_t)p.b) & 15))
        return mi_fail(MI_ERR_SHAPE, "deconv-form gemm: input channels must be a multiple of the 16-byte vector and pointers 16-B aligned");
    if (KH < 2 || KW < 2) return mi_fail(MI_ERR_SHAPE, "deconv-form gemm: kernel must be >= 2");
    if (!p.out_f32) {
        const int r4 = try_gather_narrow(st, dtype, p.a, p.b, B, IH, IW, C, OH, OW, N, KH, KW, p.out, p.bias, p.mask, p.relu);
        if (r4 != 0) return r4 > 0 ? MI_OK : r4;
    }
    if (!p.out_f32) {
        const int r5 = mi_try_rwconv_gather(st, dtype, p.a, p.b, B, IH, IW, C, OH, OW, N, KH, KW, p.out, p.bias, p.mask, p.relu, nullptr, nullptr);
        if (r5 != 0) return r5 > 0 ? MI_OK : r5;
    }
    if (!p.out_f32) {
        const int r3 = try_tapconv(st, dtype, TC_GATHER, p.a, p.b, B, IH, IW, C, OH, OW, N, KH, KW, 0, p.out, p.bias, p.mask, p.relu);
        if (r3 != 0) return r3 > 0 ? MI_OK : r3;
    }
    {
        const int r2 = try_deconv_form_gemm2(st, dtype, p, maxM);
        if (r2 != 0) return r2 > 0 ? MI_OK : r2;
    }
    if (dtype == MI_F32) return launch_gemm_bn<float, float, A_DECONV, B_DECONV, 4, 16>(st, p, maxM, 4);
    if (dtype == MI_BF16X3) return launch_gemm_bn<split_t, split_t, A_DECONV, B_DECONV, 4, 16>(st, p, maxM, 4);
    return launch_gemm_bn<bf16_t, bf16_t, A_DECONV, B_DECONV, 8, 16>(st, p, maxM, 4);
}

// pixel splits of a gen-1 filter-gradient launch (and the rows each takes): a function of the shape only
static int wgrad_splits(int dtype, int M, int Kc, int N, int target_blocks, int* m_per_split) {
    const int BP = dtype == MI_F32 ? WgradCfg<float>::BP : (dtype == MI_BF16X3 ? WgradCfg<split_t>::BP : WgradCfg<bf16_t>::BP);
    const int gx = Kc > 64 ? (Kc + 127) / 128 : 1, gy = (N + 63) / 64;
    int splits = target_blocks / (gx * gy);
    if (splits < 1) splits = 1;
    int mps = (M + splits - 1) / splits;
    mps = ((mps + BP - 1) / BP) * BP;
    if (mps < BP) mps = BP;
    if (m_per_split) *m_per_split = mps;
    return (M + mps - 1) / mps;
}

// scratch (optional): the pixel splits store per-split slabs there and one ordered pass adds them to out -- two runs are bitwise equal; without
// scratch (or with too little of it) the splits meet in fp32 atomics on out (run-to-run differences in the last bit)
// grid, row splits and slab placement of a gen-1 filter-gradient launch (shared by the single and the pair launch)
static int prepare_wgrad(int dtype, WgradParams& p, int target_blocks, void* scratch, long long scratch_bytes, int overwrite, dim3* grid, bool* wide_out) {
    const int Kce = p.Kc + (p.ones_row ? 1 : 0);          // rows of the result incl. the bias row
    const bool wide = Kce > 64;                           // 128 kc rows per block: halves the re-reads of the small tensor
    const int gx = wide ? (Kce + 127) / 128 : 1, gy = (p.N + 63) / 64;
    int mps = 0;
    const int splits = wgrad_splits(dtype, p.M, Kce, p.N, target_blocks, &mps);
    p.m_per_split = mps;
    p.debug_skip_out = g_wgrad_skip;
    p.slabs = nullptr; p.slab_stride = ((long long)Kce * p.N + 3) / 4 * 4;
    if (splits > 1 && scratch && (((uintptr_t)scratch) & 15) == 0 && scratch_bytes >= (long long)splits * p.slab_stride * 4) p.slabs = (float*)scratch;
    // overwrite: out = result (no zeroed buffer, no atomics): plain stores from the single split, or the ordered slab sum storing instead of adding
    if (overwrite && splits > 1 && !p.slabs) return mi_fail(MI_ERR_ARG, "wgrad: the overwriting form needs scratch for its row splits (mi_gemm_wgrad_scratch_bytes)");
    p.overwrite = overwrite && splits == 1 ? 1 : 0;
    *grid = dim3(gx, gy, splits); *wide_out = wide;
    return MI_OK;
}

int launch_wgrad(hipStream_t st, int dtype, int in_f32, WgradParams& p, int target_blocks, void* scratch = nullptr, long long scratch_bytes = 0, int overwrite = 0) {
    dim3 g; bool wide = false;
    { const int rc0 = prepare_wgrad(dtype, p, target_blocks, scratch, scratch_bytes, overwrite, &g, &wide); if (rc0 != MI_OK) return rc0; }
    const int splits = (int)g.z;
    const bool a16 = (((uintptr_t)p.big) & 15) == 0;
    const bool mergedok = p.merged && (p.KW * p.C) % 4 == 0 && (p.IW * p.C) % 2 == 0 && (p.stride * p.C) % 2 == 0 && (p.frame_stride % 2) == 0;
#define WG_LAUNCH(T_, TIn_, VA_, AL_) do { \
        if (wide) MI_LAUNCH((wgrad_kernel<T_, TIn_, VA_, AL_, 128>), g, dim3(GEMM_NT), 0, st, p); \
        else MI_LAUNCH((wgrad_kernel<T_, TIn_, VA_, AL_, 64>), g, dim3(GEMM_NT), 0, st, p); } while (0)
    if (dtype == MI_F32) {
        if (!p.merged && p.C % 4 == 0 && a16) WG_LAUNCH(float, float, 4, 16);
        else if (mergedok) WG_LAUNCH(float, float, 4, 8);
        else return mi_fail(MI_ERR_SHAPE, "wgrad (f32): channel count / alignment not supported");
    } else if (dtype == MI_BF16X3) {
        if (in_f32) {
            if (!p.merged && p.C % 4 == 0 && a16) WG_LAUNCH(split_t, float, 4, 16);
            else if (mergedok) WG_LAUNCH(split_t, float, 4, 8);
            else return mi_fail(MI_ERR_SHAPE, "wgrad (split, fp32 input): channel count / alignment not supported");
        } else {
            if (!p.merged && p.C % 4 == 0 && a16) WG_LAUNCH(split_t, split_t, 4, 16);
            else if (mergedok) WG_LAUNCH(split_t, split_t, 4, 8);
            else return mi_fail(MI_ERR_SHAPE, "wgrad (split): channel count / alignment not supported");
        }
    } else if (in_f32) {
        if (!p.merged && p.C % 4 == 0 && a16) WG_LAUNCH(bf16_t, float, 4, 16);
        else if (mergedok) WG_LAUNCH(bf16_t, float, 4, 8);
        else return mi_fail(MI_ERR_SHAPE, "wgrad (bf16, fp32 input): channel count / alignment not supported");
    } else {
        if (!p.merged && p.C % 8 == 0 && a16) WG_LAUNCH(bf16_t, bf16_t, 8, 16);
        else if (mergedok) WG_LAUNCH(bf16_t, bf16_t, 4, 4);
        else return mi_fail(MI_ERR_SHAPE, "wgrad (bf16): channel count / alignment not supported");
    }
#undef WG_LAUNCH
    int rc = mi_check_launch("wgrad_kernel");
    if (rc == MI_OK && p.slabs) {
        rc = mi_reduce_slabs(st, p.slabs, p.slab_stride, splits, (long long)p.Kc * p.N, p.out, overwrite);
        if (rc == MI_OK && p.ones_row) rc = mi_reduce_slabs(st, p.slabs + (long long)p.Kc * p.N, p.slab_stride, splits, (long long)p.N, p.dbias, overwrite);
    }
    return rc;
}

void fill_wgrad_geom(WgradParams& p, int B, int IH, int IW, int C, int OH, int OW, int KH, int KW, int stride, bool merged) {
    p.IH = IH; p.IW = IW; p.C = C; p.OH = OH; p.OW = OW; p.KH = KH; p.KW = KW; p.stride = stride;
    p.M = B * OH * OW; p.Kc = KH * KW * C;
    p.frame_stride = (long long)IH * IW * C;
    p.div_ohw = make_fastdiv(OH * OW); p.div_ow = make_fastdiv(OW);
    p.merged = merged ? 1 : 0; p.run = merged ? KW * C : C;
    p.div_run = make_fastdiv(p.run); p.div_kw = make_fastdiv(KW);
}

inline bool needs_merge(int C, int dtype, int in_f32) {
    const int v = (dtype != MI_BF16 || in_f32) ? 4 : 8;
    return C % v != 0;
}

}  // namespace

// internal entry points for the other translation units (mi_internal.hpp)
// Ordered slab sums: out[0 .. n) += sum_k slabs[k * stride + i], fixed order.  Deferred mode (per host thread; the VAE engine around the end of a full backward pass):
// the jobs are recorded and mi_small_reduce_flush issues all of them as ONE launch -- every job's slabs must stay untouched until then and be complete on the
// flush stream (the engine gives each its own piece of scratch and flushes on the stream that produced them).
static thread_local SmallReduceParams t_sr;
static thread_local int t_sr_defer = 0;
// The recorded list belongs to ONE stream (round 5): the stream of its first job, or the one the engine named with mi_small_reduce_bind.  A job that
// arrives from another stream while the pass defers (a layer the engine issues on its filter-gradient stream, e.g. conv1's narrow filter gradient with
// MI355_WGRAD_MAIN_MASK bit 0 clear) is launched at once on ITS stream instead of being flushed later on a stream that never waited for its slabs.
static thread_local hipStream_t t_sr_stream = nullptr;
static thread_local bool t_sr_bound = false;
static int sr_launch_params(hipStream_t st, SmallReduceParams& f) {
    if (f.njobs == 0) return MI_OK;
    for (int i = f.njobs; i < SR_MAX; ++i) f.first[i + 1] = f.first[f.njobs];
    MI_LAUNCH(reduce_small_fused_kernel, dim3((unsigned)f.first[f.njobs]), dim3(256), 0, st, f);
    f.njobs = 0;
    return mi_check_launch("reduce_small_fused_kernel");
}
int mi_reduce_slabs(hipStream_t st, const float* slabs, long long stride, int nslab, long long n, float* out, int overwrite) {
    if (nslab < 1 || n < 1) return MI_OK;
    SmallReduceParams& f = t_sr;
    if (!t_sr_defer || ((f.njobs > 0 || t_sr_bound) && st != t_sr_stream)) {      // not deferring, or not the list's stream: one launch of its own, right here
        SmallReduceParams one = {};
        sr_add(one, slabs, stride, nslab, n, out, overwrite);
        return sr_launch_params(st, one);
    }
    if (f.njobs == SR_MAX) { const int rc = sr_launch_params(t_sr_stream, f); if (rc != MI_OK) return rc; }      // (a full list is issued as it is, on its own stream: still one fixed order per job)
    if (f.njobs == 0 && !t_sr_bound) t_sr_stream = st;
    sr_add(f, slabs, stride, nslab, n, out, overwrite);
    return MI_OK;
}
extern "C" int mi_small_reduce_defer(int on) {              // returns the previous mode; switching drops what an aborted pass may have left in the list
    const int prev = t_sr_defer;
    t_sr_defer = on ? 1 : 0;
    t_sr.njobs = 0; t_sr.first[0] = 0;
    t_sr_bound = false; t_sr_stream = nullptr;
    return prev;
}
extern "C" int mi_small_reduce_bind(void* stream) {         // the deferring pass names the stream its list will be flushed on (jobs from any other stream launch at once)
    if (t_sr.njobs > 0 && (hipStream_t)stream != t_sr_stream) return mi_fail(MI_ERR_STATE, "mi_small_reduce_bind: jobs of another stream are pending");
    t_sr_stream = (hipStream_t)stream; t_sr_bound = true;
    return MI_OK;
}
extern "C" int mi_small_reduce_flush(void* stream) {
    if (t_sr.njobs == 0) return MI_OK;
    if ((hipStream_t)stream != t_sr_stream) {               // a programming error in the calling engine: nothing is lost (the list runs where its slabs are complete), but say so
        const int rc = sr_launch_params(t_sr_stream, t_sr);
        return rc != MI_OK ? rc : mi_fail(MI_ERR_STATE, "mi_small_reduce_flush: the recorded jobs belong to another stream (issued there)");
    }
    return sr_launch_params(t_sr_stream, t_sr);
}
extern "C" int mi_small_reduce_deferring(void) { return t_sr_defer; }

bool mi_narrow_enabled() { return narrow_enabled(); }

void mi_get_trace(long long** buf, int* cap) { *buf = g_trace; *cap = g_trace_cap; }

static bool tallk_enabled() {                              // MI355_TALLK=0: the general split-K kernel for the latent-side layers (A/B runs)
    static int on = -1;
    if (on < 0) { const char* e = getenv("MI355_TALLK"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}

extern "C" {

// debug: device buffer of int64 stamps, 32 per wave of every tapconv block (see tools/trace_tapconv.py); nullptr switches it off
int mi_debug_set_trace(void* dev_ptr, int capacity_entries) { g_trace = (long long*)dev_ptr; g_trace_cap = dev_ptr ? capacity_entries : 0; return MI_OK; }

int mi_set_tuning(int key, int value) {
    int prev;
    if (key == 0) { prev = gemm2_enabled() ? 1 : 0; g_gemm2_on = value ? 1 : 0; }
    else if (key == 1) { prev = tapconv_minblocks(); g_tap_min = value < 0 ? -1 : value; }
    else if (key == 2) { prev = g_wgrad_skip; g_wgrad_skip = value; }
    else if (key == 5) { prev = g_tap_variant; g_tap_variant = value; }
    else if (key == 6) { prev = g_tap_direct; g_tap_direct = value ? 1 : 0; }
    else if (key == 8) { prev = 0; }                     // (was: persistent tapconv blocks, removed -- rwconv.hip is the persistent form that won)
    else if (key == 4) { prev = narrow_enabled() ? 1 : 0; g_narrow_on = value ? 1 : 0; }
    else if (key == 3) { prev = tapwgrad_enabled() ? 1 : 0; g_tapwgrad_on = value ? 1 : 0; }
    else if (key == 7) { prev = g_tapwgrad_split; g_tapwgrad_split = value ? 1 : 0; }
    else if (key == 9) { prev = g_tapwgrad_blocks; g_tapwgrad_blocks = value < 16 ? 16 : value; }
    else if (key == 10) { prev = g_nw_waves; g_nw_waves = value; }
    else if (key == 12) { prev = g_tap_mask_prefetch; g_tap_mask_prefetch = value ? 1 : 0; }
    else if (key == 11) { prev = g_dense_wgrad_blocks; g_dense_wgrad_blocks = value < 1 ? 1 : value; }
    else if (key == 13) { prev = mi_rwconv_mode(value < 0 ? 0 : value); }
    else if (key == 14) { prev = g_tapwgrad_cw; g_tapwgrad_cw = value ? 1 : 0; }
    else if (key == 15) { prev = mi_rwconv_conv_mode(value < 0 ? 0 : value); }
    else if (key == 16) { prev = mi_rwconv_blocks(value < 0 ? 0 : value); }
    else if (key == 17) { prev = g_gemm2_tile; g_gemm2_tile = value; }
    else if (key == 18) { prev = g_slab_bf16; g_slab_bf16 = value ? 1 : 0; }
    else if (key == 19) { prev = g_nw_depth; g_nw_depth = value; }
    else if (key == 20) { prev = g_gemm2_stages; g_gemm2_stages = value; }
    else if (key == 21) { prev = g_x3_tapwgrad; g_x3_tapwgrad = value; }
    else if (key == 22) { prev = g_dwgs_on; g_dwgs_on = value ? 1 : 0; }
    else if (key == 24) { prev = g_tw_ldec; g_tw_ldec = value < 0 || value > 3 ? 0 : value; }
    else if (key == 25) { prev = g_dectail_dbg; g_dectail_dbg = value; }
    else if (key == 26) { prev = g_dectail_split5; g_dectail_split5 = value ? 1 : 0; }
    else if (key == 23) { prev = mi_enc12_debug(value); }  // (debug: ablation mask of the fused encoder head's timing instantiation -- results are wrong with any bit set)
    else return mi_fail(MI_ERR_ARG, "mi_set_tuning: unknown key");
    return prev;
}

// conv2d NHWC stride-2 VALID forward: out[B,OH,OW,Cout] = relu?(im2col(x) * W[kh,kw,ci,co] + bias)
// replaces tf.layers.conv2d in ConvVAE.build_encoder (reference vae/models.py:250-253)
// w_transposed = 1: w holds the kernel as [Cout][KH*KW*Cin] (K-contiguous: vector loads + conflict-free LDS staging)
int mi_conv2d_nhwc_fwd(void* stream, int dtype, const void* x, const int* frame_idx, int x_is_f32,
                       int B, int IH, int IW, int Cin, const void* w, int w_transposed, const float* bias, int KH, int KW, int Cout,
                       int relu, void* out) {
    return mi_conv2d_nhwc_fwd_bits(stream, dtype, x, frame_idx, x_is_f32, B, IH, IW, Cin, w, w_transposed, bias, KH, KW, Cout, relu, out, nullptr, nullptr);
}

// same; relu_bits != NULL: the kernel also writes the ReLU bit words of its output (2 x uint32 per pixel and 32 channels, see the header)
// where it can -- *wrote_bits tells -- so that the input gradient of the NEXT layer reads 8 bytes per pixel instead of the 64-byte row
int mi_conv2d_nhwc_fwd_bits(void* stream, int dtype, const void* x, const int* frame_idx, int x_is_f32,
                            int B, int IH, int IW, int Cin, const void* w, int w_transposed, const float* bias, int KH, int KW, int Cout,
                            int relu, void* out, void* relu_bits, int* wrote_bits) {
    const int OH = (IH - KH) / 2 + 1, OW = (IW - KW) / 2 + 1;
    if (wrote_bits) *wrote_bits = 0;
    if (w_transposed) {
        const int r4 = try_narrow_conv((hipStream_t)stream, dtype, x, x_is_f32 == 2 ? 2 : ((x_is_f32 || dtype == MI_F32) ? 1 : (dtype == MI_BF16X3 ? 3 : 0)), frame_idx, w, B, IH, IW, Cin, KH, KW, Cout, bias, relu, nullptr, out,
                                       (dtype == MI_BF16 && relu && Cout == 32) ? relu_bits : nullptr);
        if (r4 > 0 && wrote_bits && relu_bits && dtype == MI_BF16 && relu && Cout == 32) *wrote_bits = 1;
        if (r4 != 0) return r4 > 0 ? MI_OK : r4;
    }
    if (x_is_f32 == 2) return mi_fail(MI_ERR_ARG, "mi_conv2d_nhwc_fwd: uint8 frames are only read by the narrow-layer kernel (bf16 mode, 1..3 channels -> 32, K-contiguous weights)");
    GemmParams p = {};
    p.a = x; p.a_frame_idx = frame_idx;
    fill_conv_geom(p, B, IH, IW, Cin, OH, OW, KH, KW, 2, needs_merge(Cin, dtype, x_is_f32));
    p.N = Cout; p.b = w; p.ldb = w_transposed ? p.K : Cout; p.b_vec = vec_ok(w, p.ldb, dtype);
    p.out = out; p.bias = bias; p.mask = nullptr; p.relu = relu; p.out_f32 = 0; p.ksplit_len = 0;
    if (w_transposed) return conv_form_gemm<B_NK>((hipStream_t)stream, dtype, x_is_f32, p, 1);
    return conv_form_gemm<B_KN>((hipStream_t)stream, dtype, x_is_f32, p, 1);
}

// conv2d input gradient: dx[B,IH,IW,Cin] = mask>0 ? deconv_gather(dy[B,OH,OW,Cout], W[kh,kw,ci,co]) : 0
// (TF Conv2DBackpropInput; the HWIO conv kernel read as a [kh,kw,out=ci,in=co] transposed-conv kernel)
int mi_conv2d_nhwc_dgrad(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout,
                         const void* w, int KH, int KW, int Cin, int IH, int IW, const void* mask, void* dx) {
    return mi_conv2d_nhwc_dgrad_bits(stream, dtype, dy, B, OH, OW, Cout, w, KH, KW, Cin, IH, IW, mask, nullptr, dx);
}

// same; mask_bits != NULL: the ReLU bit words of the mask tensor (written by mi_conv2d_nhwc_fwd_bits) are read instead of the tensor itself
// by the kernels that can (the register-weight kernel); the others still read `mask`
int mi_conv2d_nhwc_dgrad_bits(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout,
                              const void* w, int KH, int KW, int Cin, int IH, int IW, const void* mask, const void* mask_bits, void* dx) {
    if (mask_bits) {
        const int r5 = mi_try_rwconv_gather((hipStream_t)stream, dtype, dy, w, B, OH, OW, Cout, IH, IW, Cin, KH, KW, dx, nullptr, mask, 0, mask_bits, nullptr);
        if (r5 != 0) return r5 > 0 ? MI_OK : r5;
    }
    GemmParams p = {};
    p.a = dy; p.b = w; p.ldb = 0; p.b_vec = 1;
    p.out = dx; p.bias = nullptr; p.mask = mask; p.relu = 0; p.out_f32 = 0;
    return deconv_form_gemm((hipStream_t)stream, dtype, p, B, OH, OW, Cout, IH, IW, Cin, KH, KW);
}

// conv2d filter gradient: dw[kh,kw,ci,co] += im2col(x)^T * dy      (TF Conv2DBackpropFilter), fp32 accumulate
int mi_conv2d_nhwc_wgrad(void* stream, int dtype, const void* x, const int* frame_idx, int x_is_f32,
                         int B, int IH, int IW, int Cin, const void* dy, int KH, int KW, int Cout, float* dw) {
    return mi_conv2d_nhwc_wgrad_ws(stream, dtype, x, frame_idx, x_is_f32, B, IH, IW, Cin, dy, KH, KW, Cout, dw, nullptr, 0, nullptr);
}

// same with caller-provided scratch (bytes): lets the bf16 kernel reduce its position splits without atomics (deterministic)
int mi_conv2d_nhwc_wgrad_ws(void* stream, int dtype, const void* x, const int* frame_idx, int x_is_f32,
                            int B, int IH, int IW, int Cin, const void* dy, int KH, int KW, int Cout, float* dw, void* scratch, long long scratch_bytes,
                            float* dbias) {
    const int OH = (IH - KH) / 2 + 1, OW = (IW - KW) / 2 + 1;
    {
        const int r4 = try_narrow_wgrad((hipStream_t)stream, dtype, x, x_is_f32 == 2 ? 2 : ((x_is_f32 || dtype == MI_F32) ? 1 : 0), frame_idx, dy, B, IH, IW, Cin, OH, OW, Cout, KH, KW, dw, dbias, scratch, scratch_bytes);   // per-block slabs + ordered reduce when the caller brings scratch (deterministic; atomics measured 92 vs 119 us -- the engines' default path is the fused encoder-head kernel anyway)
        if (r4 != 0) return r4 > 0 ? MI_OK : r4;
    }
    if (x_is_f32 == 2) return mi_fail(MI_ERR_ARG, "mi_conv2d_nhwc_wgrad: uint8 frames are only read by the narrow-layer kernel (bf16 mode)");
    if (!frame_idx && !x_is_f32) {
        const int r3 = dtype == MI_BF16X3 ? try_tapwgrad_split((hipStream_t)stream, TC_CONV, x, dy, B, IH, IW, Cin, OH, OW, Cout, KH, KW, dw, scratch, scratch_bytes, dbias)
                                          : try_tapwgrad((hipStream_t)stream, dtype, TC_CONV, x, dy, B, IH, IW, Cin, OH, OW, Cout, KH, KW, dw, scratch, scratch_bytes, dbias);
        if (r3 != 0) return r3 > 0 ? MI_OK : r3;
    }
    if (dbias) {                                          // not fused on this path: BiasAddGrad as its own pass, with the first piece of the scratch (its reduce may be deferred: the filter gradient below must not reuse it)
        long long cb = (mi_colsum_scratch_bytes(dtype, (long long)B * OH * OW, Cout) + 255) / 256 * 256;
        if (!scratch || cb > scratch_bytes) cb = 0;
        const int rcb = mi_colsum_ws(stream, dtype, dy, (long long)B * OH * OW, Cout, dbias, cb ? scratch : nullptr, cb);
        if (rcb != MI_OK) return rcb;
        if (cb) { scratch = (char*)scratch + cb; scratch_bytes -= cb; }
    }
    WgradParams p = {};
    p.big = x; p.frame_idx = frame_idx;
    fill_wgrad_geom(p, B, IH, IW, Cin, OH, OW, KH, KW, 2, needs_merge(Cin, dtype, x_is_f32));
    p.N = Cout; p.small = dy; p.s_vec = vec_ok(dy, Cout, dtype); p.out = dw;
    return launch_wgrad((hipStream_t)stream, dtype, x_is_f32, p, 1024, scratch, scratch_bytes);
}

// conv2d_transpose NHWC stride-2 VALID forward, kernel [kh,kw,co,ci] (reference vae/models.py:261-264)
int mi_deconv2d_nhwc_fwd(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin,
                         const void* w, const float* bias, int KH, int KW, int Cout, int relu, void* out) {
    return mi_deconv2d_nhwc_fwd_bits(stream, dtype, x, B, IH, IW, Cin, w, bias, KH, KW, Cout, relu, out, nullptr, nullptr);
}

// same; relu_bits != NULL: also writes the ReLU bit words of the output where the kernel can (*wrote_bits tells)
int mi_deconv2d_nhwc_fwd_bits(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin,
                              const void* w, const float* bias, int KH, int KW, int Cout, int relu, void* out, void* relu_bits, int* wrote_bits) {
    if (wrote_bits) *wrote_bits = 0;
    if (relu_bits && relu) {
        const int r5 = mi_try_rwconv_gather((hipStream_t)stream, dtype, x, w, B, IH, IW, Cin, (IH - 1) * 2 + KH, (IW - 1) * 2 + KW, Cout, KH, KW, out, bias, nullptr, relu, nullptr, relu_bits);
        if (r5 > 0 && wrote_bits) *wrote_bits = 1;
        if (r5 != 0) return r5 > 0 ? MI_OK : r5;
    }
    const int OH = (IH - 1) * 2 + KH, OW = (IW - 1) * 2 + KW;
    GemmParams p = {};
    p.a = x; p.b = w; p.ldb = 0; p.b_vec = 1;
    p.out = out; p.bias = bias; p.mask = nullptr; p.relu = relu; p.out_f32 = 0;
    return deconv_form_gemm((hipStream_t)stream, dtype, p, B, IH, IW, Cin, OH, OW, Cout, KH, KW);
}

// conv2d_transpose into narrow logits + fused reconstruction loss (see the header); *n_partial = 0: not eligible, nothing launched
int mi_deconv2d_nhwc_fwd_bce(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin, const void* w, const float* bias, int KH, int KW, int Cout,
                             void* logits, const float* labels, const int* frame_idx, long long label_stride, int loss_kind, float inv_batch, void* dlogits,
                             float* loss_partial, float* bias_partial, int partial_capacity, int* n_partial) {
    return mi_deconv2d_nhwc_fwd_bce_u8(stream, dtype, x, B, IH, IW, Cin, w, bias, KH, KW, Cout, logits, labels, 0, frame_idx, label_stride, loss_kind, inv_batch, dlogits,
                                       loss_partial, bias_partial, partial_capacity, n_partial);
}

int mi_deconv2d_nhwc_fwd_bce_u8(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin, const void* w, const float* bias, int KH, int KW, int Cout,
                                void* logits, const void* labels_any, int labels_u8, const int* frame_idx, long long label_stride, int loss_kind, float inv_batch, void* dlogits,
                                float* loss_partial, float* bias_partial, int partial_capacity, int* n_partial) {
    const float* labels = (const float*)labels_any;
    if (!n_partial || !labels || !loss_partial || !bias_partial) return mi_fail(MI_ERR_ARG, "mi_deconv2d_nhwc_fwd_bce: missing buffers");
    *n_partial = 0;
    if (loss_kind < 0 || loss_kind > 2) return mi_fail(MI_ERR_ARG, "mi_deconv2d_nhwc_fwd_bce: loss_kind must be 0 (bce), 1 (bce_v2) or 2 (mse)");
    const int OH = (IH - 1) * 2 + KH, OW = (IW - 1) * 2 + KW;
    NarrowLoss L = {labels, frame_idx, label_stride, loss_kind, inv_batch, dlogits, loss_partial, bias_partial, partial_capacity, 0, labels_u8 ? 1 : 0};
    const int r = try_gather_narrow((hipStream_t)stream, dtype, x, w, B, IH, IW, Cin, OH, OW, Cout, KH, KW, logits, bias, nullptr, 0, &L);
    if (r < 0) return r;
    if (r > 0) *n_partial = L.nblocks;
    return MI_OK;
}

// The decoder tail of a training step in ONE launch (dectail_tile.hpp): deconv4 forward, the reconstruction loss on its logits, and the two gradients of
// that layer -- dx = gradient wrt deconv3's pre-activation (ReluGrad mask = x itself), dw += filter gradient -- with logits and dlogits kept on chip.
// w: kernel [kh,kw,out,in]; w_t: its K-contiguous copy [in][kh*kw*out] (mi_transpose_weights); loss / bias partial sums per block as
// mi_deconv2d_nhwc_fwd_bce.  scratch: >= mi_deconv2d_tail_blocks() * 6144 bytes.  *n_partial = blocks written, or 0 when the layer is not eligible
// (bf16, 32 -> 3 channels, k = 4; nothing was launched: use the separate ops).
static int dectail_grid(int fast) {
    static int resident[2];
    if (!resident[fast]) {
        int per_cu = 0, dev = 0, cus = 256;
        hipDeviceProp_t pr;
        const void* fn = fast ? (const void*)dectail_kernel<true> : (const void*)dectail_kernel<false>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        resident[fast] = per_cu * cus;
    }
    return resident[fast];
}
int mi_deconv2d_tail_blocks(void) { const int a = dectail_grid(0), b = dectail_grid(1); return a > b ? a : b; }

int mi_deconv2d_tail_fused(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin, const void* w, const void* w_t, const float* bias, int KH, int KW, int Cout,
                           const void* labels, int labels_u8, const int* frame_idx, long long label_stride, int loss_kind, float inv_batch,
                           void* dx, float* dw, float* loss_partial, float* bias_partial, int partial_capacity, int* n_partial, void* scratch, long long scratch_bytes, int reduce_now) {
    if (!n_partial || !labels || !loss_partial || !bias_partial || !x || !w || !w_t || !dx || !dw) return mi_fail(MI_ERR_ARG, "mi_deconv2d_tail_fused: missing buffers");
    *n_partial = 0;
    if (loss_kind < 0 || loss_kind > 2) return mi_fail(MI_ERR_ARG, "mi_deconv2d_tail_fused: loss_kind must be 0 (bce), 1 (bce_v2) or 2 (mse)");
    static int on = -1;
    if (on < 0) { const char* e = getenv("MI355_DECTAIL"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || !narrow_enabled() || dtype != MI_BF16 || Cin != 32 || Cout != 3 || KH != 4 || KW != 4 || B < 1 || IH < 1 || IW < 1) return MI_OK;
    if (((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)w_t) | ((uintptr_t)dx) | ((uintptr_t)scratch) | ((uintptr_t)dw)) & 15) || !scratch) return MI_OK;
    // the label tile is staged in 4-value items: rows of 3 OW values must be a whole number of items, frames 4-byte (uint8) / 16-byte (fp32) aligned
    if ((3 * (2 * IW + 2)) % 4 != 0 || label_stride % 4 != 0 || (((uintptr_t)labels) & (labels_u8 ? 3 : 15))) return MI_OK;
    DecTailParams q = {};
    q.x = (const bf16_t*)x; q.B = B; q.IH = IH; q.IW = IW; q.w = (const bf16_t*)w; q.wt = (const bf16_t*)w_t; q.bias = bias;
    q.labels = labels; q.lab_u8 = labels_u8 ? 1 : 0; q.lab_idx = frame_idx; q.lab_stride = label_stride; q.loss_kind = loss_kind; q.inv_b = inv_batch;
    q.dx = (bf16_t*)dx; q.lpart = loss_partial; q.bpart = bias_partial;
    q.OH = 2 * IH + 2; q.OW = 2 * IW + 2; q.GH = IH + 1; q.GW = IW + 1;
    static int edge = -1;                                 // A/B knob: tiles over the pixel grid, last slot row / column owned by the last tiles
    if (edge < 0) { const char* e = getenv("MI355_DECTAIL_EDGE"); edge = (e && e[0] == '0') ? 0 : 1; }
    q.edge_own = edge;
    if ((long long)B * IH * IW * 64 >= 0x7fffff00ll || (long long)q.OH * q.OW * 12 >= 0x7fffff00ll) return MI_OK;      // 32-bit buffer offsets
    q.x_bytes = (unsigned)((long long)B * IH * IW * 64);
    const int tiles_y = ((edge ? IH : q.GH) + DT_TY - 1) / DT_TY;
    q.tiles_x = ((edge ? IW : q.GW) + DT_TX - 1) / DT_TX; q.tiles_per_frame = tiles_y * q.tiles_x;
    const long long ntiles = (long long)B * q.tiles_per_frame;
    if (ntiles >= (1ll << 30) || (long long)B * q.OH * q.OW * 3 >= (1ll << 40)) return MI_OK;
    q.ntiles = (int)ntiles; q.div_tpf = make_fastdiv(q.tiles_per_frame); q.div_tx = make_fastdiv(q.tiles_x);
    const int fast = loss_kind == 0 ? 1 : 0;
    int nblocks = dectail_grid(fast);
    if (nblocks > q.ntiles) nblocks = q.ntiles;
    if (nblocks >= 16) nblocks &= ~7;                       // whole rounds of the eight XCDs (the kernel's tile order)
    if (nblocks > partial_capacity || scratch_bytes < (long long)nblocks * DT_SLAB * 4) return MI_OK;
    q.slabs = (float*)scratch;
    q.dbg = g_dectail_dbg;
    if (fast && q.dbg) MI_LAUNCH((dectail_kernel<true, true>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);      // (timing instantiation: tools/dectail_ablate.py)
    else if (fast && g_dectail_split5) MI_LAUNCH((dectail_kernel<true, false, true>), dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);
    else if (fast) MI_LAUNCH(dectail_kernel<true>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);
    else MI_LAUNCH(dectail_kernel<false>, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, q);
    int rc = mi_check_launch("dectail_kernel");
    if (rc != MI_OK) return rc;
    if (reduce_now) rc = mi_deconv2d_tail_reduce(stream, scratch, nblocks, dw);
    if (rc == MI_OK) *n_partial = nblocks;
    return rc;
}

// dw[1536] += the per-block partial filter gradients mi_deconv2d_tail_fused(reduce_now = 0) left in scratch (n_partial of them): any stream position
// behind that launch and in front of the optimiser step (the VAE engine runs it where its stream would otherwise wait for the other one)
int mi_deconv2d_tail_reduce(void* stream, const void* scratch, int n_partial, float* dw) {
    if (!scratch || !dw || n_partial < 1) return mi_fail(MI_ERR_ARG, "mi_deconv2d_tail_reduce: missing buffers");
    if (t_sr_defer) return mi_reduce_slabs((hipStream_t)stream, (const float*)scratch, (long long)DT_SLAB, n_partial, (long long)DT_SLAB, dw);      // one of the jobs of the pass's fused small reduce
    MI_LAUNCH(dectail_reduce_kernel, dim3(DT_SLAB / 32), dim3(1024), 0, (hipStream_t)stream, (const float*)scratch, n_partial, dw);
    return mi_check_launch("dectail_reduce_kernel");
}

// conv2d_transpose input gradient = plain stride-2 conv of dy with the same kernel read as HWIO [kh,kw,I=co,O=ci]
// w_transposed = 1: w holds the kernel as [Cin][KH*KW*Cout]
int mi_deconv2d_nhwc_dgrad(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout,
                           const void* w, int w_transposed, int KH, int KW, int Cin, const void* mask, void* dx) {
    return mi_deconv2d_nhwc_dgrad_bits(stream, dtype, dy, B, OH, OW, Cout, w, w_transposed, KH, KW, Cin, mask, nullptr, dx);
}

int mi_deconv2d_nhwc_dgrad_bits(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout,
                                const void* w, int w_transposed, int KH, int KW, int Cin, const void* mask, const void* mask_bits, void* dx) {
    const int IH = (OH - KH) / 2 + 1, IW = (OW - KW) / 2 + 1;
    if (w_transposed) {
        const int r4 = try_narrow_conv((hipStream_t)stream, dtype, dy, dtype == MI_F32 ? 1 : (dtype == MI_BF16X3 ? 3 : 0), nullptr, w, B, OH, OW, Cout, KH, KW, Cin, nullptr, 0, mask, dx, nullptr, mask_bits);
        if (r4 != 0) return r4 > 0 ? MI_OK : r4;
    }
    GemmParams p = {};
    p.a = dy; p.a_frame_idx = nullptr;
    fill_conv_geom(p, B, OH, OW, Cout, IH, IW, KH, KW, 2, needs_merge(Cout, dtype, 0));
    p.N = Cin; p.b = w; p.ldb = w_transposed ? p.K : Cin; p.b_vec = vec_ok(w, p.ldb, dtype);
    p.out = dx; p.bias = nullptr; p.mask = mask; p.relu = 0; p.out_f32 = 0; p.ksplit_len = 0;
    if (w_transposed) return conv_form_gemm<B_NK>((hipStream_t)stream, dtype, 0, p, 1);
    return conv_form_gemm<B_KN>((hipStream_t)stream, dtype, 0, p, 1);
}

// conv2d_transpose filter gradient: dw[kh,kw,co,ci] += im2col(dy)^T * x
int mi_deconv2d_nhwc_wgrad(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout,
                           const void* x, int KH, int KW, int Cin, float* dw) {
    return mi_deconv2d_nhwc_wgrad_ws(stream, dtype, dy, B, OH, OW, Cout, x, KH, KW, Cin, dw, nullptr, 0, nullptr);
}

int mi_deconv2d_nhwc_wgrad_ws(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout,
                              const void* x, int KH, int KW, int Cin, float* dw, void* scratch, long long scratch_bytes, float* dbias) {
    const int IH = (OH - KH) / 2 + 1, IW = (OW - KW) / 2 + 1;
    {   // deconv with a narrow OUTPUT: dW[kh,kw,co,ci] = sum patches(dy)[.,(kh,kw,co)] x[.,ci]; its bias gradient is not a by-product here
        const int r4 = dbias ? 0 : try_narrow_wgrad((hipStream_t)stream, dtype, dy, 0, nullptr, x, B, OH, OW, Cout, IH, IW, Cin, KH, KW, dw, nullptr, scratch, scratch_bytes);
        if (r4 != 0) return r4 > 0 ? MI_OK : r4;
    }
    {
        const int r3 = dtype == MI_BF16X3 ? try_tapwgrad_split((hipStream_t)stream, TC_GATHER, x, dy, B, IH, IW, Cin, OH, OW, Cout, KH, KW, dw, scratch, scratch_bytes, dbias)
                                          : try_tapwgrad((hipStream_t)stream, dtype, TC_GATHER, x, dy, B, IH, IW, Cin, OH, OW, Cout, KH, KW, dw, scratch, scratch_bytes, dbias);
        if (r3 != 0) return r3 > 0 ? MI_OK : r3;
    }
    if (dbias) {
        long long cb = (mi_colsum_scratch_bytes(dtype, (long long)B * OH * OW, Cout) + 255) / 256 * 256;
        if (!scratch || cb > scratch_bytes) cb = 0;
        const int rcb = mi_colsum_ws(stream, dtype, dy, (long long)B * OH * OW, Cout, dbias, cb ? scratch : nullptr, cb);
        if (rcb != MI_OK) return rcb;
        if (cb) { scratch = (char*)scratch + cb; scratch_bytes -= cb; }
    }
    WgradParams p = {};
    p.big = dy; p.frame_idx = nullptr;
    fill_wgrad_geom(p, B, OH, OW, Cout, IH, IW, KH, KW, 2, needs_merge(Cout, dtype, 0));
    p.N = Cin; p.small = x; p.s_vec = vec_ok(x, Cin, dtype); p.out = dw;
    return launch_wgrad((hipStream_t)stream, dtype, 0, p, 1024, scratch, scratch_bytes);
}

// Dense: out[M,N] = act(a[M,K] * W + bias).  w_layout 0: W[K,N] (tf dense kernel), 1: W[N,K] (used for x * W^T).
// nsplit > 1: split-K, raw fp32 partial slabs out[nsplit][M][N] (bias/act/mask must be unset; consumer reduces).
int mi_gemm_bias_act(void* stream, int dtype, const void* a, int M, int K, const void* w, int w_layout, int N,
                     const float* bias, int relu, const void* mask, void* out, int out_f32, int nsplit) {
    if (dtype == MI_F32 && w_layout == 0 && nsplit <= 1 && !mask && M <= 256 && K % 4 == 0 && gemm2_enabled() &&
        ((((uintptr_t)a) | ((uintptr_t)out)) & 15) == 0) {                 // small-M fp32 dense layer (PPO): K split across the waves of a block
        DenseSmallParams q = {(const float*)a, (const float*)w, bias, (float*)out, M, N, K, relu};
        MI_LAUNCH(dense_smallm_kernel, dim3((N + 31) / 32, (M + 31) / 32), dim3(256), 0, (hipStream_t)stream, q);
        return mi_check_launch("dense_smallm_kernel");
    }
    // long reduction into a narrow output, K-contiguous weights, raw split-K slabs wanted (the latent-side layers): tallk_tile.hpp
    if (dtype == MI_BF16 && w_layout == 1 && nsplit > 1 && out_f32 && !bias && !relu && !mask && tallk_enabled() && (N == 32 || N == 64 || N == 128) &&
        K % (nsplit * 16) == 0 && ((((uintptr_t)a) | ((uintptr_t)w)) & 15) == 0 && (long long)M * K * 2 < (1ll << 30) && (long long)N * K * 2 < (1ll << 30)) {
        const int nt = N / 32, kp = 4 / nt;
        if (nsplit % kp == 0) {
            TallKParams q = {a, w, (float*)out, M, N, K, K / nsplit, (unsigned)((long long)M * K * 2), (unsigned)((long long)N * K * 2)};
            const dim3 g((M + 31) / 32, nsplit / kp);
            if (nt == 4) MI_LAUNCH(tallk_kernel<4>, g, dim3(256), 0, (hipStream_t)stream, q);
            else if (nt == 2) MI_LAUNCH(tallk_kernel<2>, g, dim3(256), 0, (hipStream_t)stream, q);
            else MI_LAUNCH(tallk_kernel<1>, g, dim3(256), 0, (hipStream_t)stream, q);
            return mi_check_launch("tallk_kernel");
        }
    }
    GemmParams p = {};
    p.a = a; p.a_frame_idx = nullptr;
    fill_conv_geom(p, M, 1, 1, K, 1, 1, 1, 1, 1, false);
    const int vb = dtype == MI_BF16 ? 8 : 4;
    if (K % vb != 0) return mi_fail(MI_ERR_SHAPE, "mi_gemm_bias_act: K must be a multiple of the 16-byte vector (pad K)");
    p.N = N; p.b = w; p.ldb = w_layout == 0 ? N : K; p.b_vec = vec_ok(w, p.ldb, dtype);
    p.out = out; p.bias = bias; p.mask = mask; p.relu = relu; p.out_f32 = out_f32;
    int gz = 1;
    if (nsplit > 1) {
        if (bias || relu || mask || !out_f32) return mi_fail(MI_ERR_ARG, "mi_gemm_bias_act: split-K writes raw fp32 slabs only");
        const int bk = dtype == MI_BF16 ? 32 : 16;
        int len = (K + nsplit - 1) / nsplit;
        len = ((len + bk - 1) / bk) * bk;
        if ((long long)len * (nsplit - 1) >= K) return mi_fail(MI_ERR_ARG, "mi_gemm_bias_act: nsplit too large for K (empty slab)");
        p.ksplit_len = len; gz = nsplit;
    }
    if (w_layout == 0) return conv_form_gemm<B_KN>((hipStream_t)stream, dtype, 0, p, gz);
    return conv_form_gemm<B_NK>((hipStream_t)stream, dtype, 0, p, gz);
}

// Dense filter gradient: dw[K,N] += a[M,K]^T * dy[M,N]   (row splits meet in fp32 atomics)
int mi_gemm_wgrad(void* stream, int dtype, const void* a, const void* dy, int M, int K, int N, float* dw) {
    return mi_gemm_wgrad_ws(stream, dtype, a, dy, M, K, N, dw, nullptr, 0);
}

// same with caller scratch (>= mi_gemm_wgrad_scratch_bytes): the row splits store per-split slabs that one ordered pass adds to dw -- deterministic
// the LDS-free one-wave-per-tile kernel (dwgs_tile.hpp) takes this layer
static bool dwgs_eligible(int dtype, int M, int K, int N) {
    // up to 64 tiles (K N <= 256 K): the small layers of the MlpVAE (-1.7 ... -2.3 % on its step).  The latent layers of the ConvVAE (96 / 192 tiles) were measured on all
    // three forms of this kernel -- one wave per tile, row splits through slabs, row splits inside the block -- at 0.8573 / 0.8168 / 0.8163 ms per step against 0.8449 / 0.8153 /
    // 0.8032 on the first-generation kernel (same box within a pair): at the end of the backward pass their time is set by the 147 KB-LDS filter gradient that holds the
    // CUs and by the slab reduce that saturates HBM, and the 64 KB-LDS row-split kernel rides that out better
    return g_dwgs_on && dtype == MI_BF16 && M >= 16 && M % 16 == 0 && K % 64 == 0 && N % 64 == 0 && (long long)K * N <= 262144 &&
           (long long)M * K < (1ll << 31) && (long long)M * N < (1ll << 31);
}
// ... with this many waves per tile (row splits inside the block; a function of the row count only): at least two load rounds per wave
static int dwgs_wsplit(int M) {
    const int steps = M / 16;
    if (steps % 4 == 0 && steps / 4 >= 2) return 4;
    if (steps % 2 == 0 && steps / 2 >= 2) return 2;
    return 1;
}

long long mi_gemm_wgrad_scratch_bytes(int dtype, int M, int K, int N) {
    if (M < 1 || K < 1 || N < 1) return 0;
    const int splits = wgrad_splits(dtype, M, K + 1, N, g_dense_wgrad_blocks, nullptr);      // (+ 1: room for the bias row of mi_gemm_wgrad_bias_ws)
    return splits > 1 ? (long long)splits * (((long long)(K + 1) * N + 3) / 4 * 4) * 4 : 0;  // (one row split: a single block per element adds straight into dw)
}

int mi_gemm_wgrad_ws(void* stream, int dtype, const void* a, const void* dy, int M, int K, int N, float* dw, void* scratch, long long scratch_bytes) {
    return mi_gemm_wgrad_bias_ws(stream, dtype, a, dy, M, K, N, dw, nullptr, scratch, scratch_bytes);
}

// same; dbias != NULL: dbias[n] += sum_m dy[m, n] as well -- the layer's BiasAddGrad as one more row of the same product (a column of ones appended to `a` inside
// the kernel's loader): no separate column-sum launch
int mi_gemm_wgrad_bias_ws(void* stream, int dtype, const void* a, const void* dy, int M, int K, int N, float* dw, float* dbias, void* scratch, long long scratch_bytes) {
    return mi_gemm_wgrad_bias_set(stream, dtype, a, dy, M, K, N, dw, dbias, scratch, scratch_bytes, 0);
}

// same; overwrite != 0: dw (and dbias) = the gradient instead of += : plain stores (one row split) or the storing form of the ordered slab sum -- the gradient buffer
// need not be zeroed between steps and no element is touched by an atomic (the MlpVAE engine: 39.5 M weights, the zeroing alone was 158 MB per step)
int mi_gemm_wgrad_bias_set(void* stream, int dtype, const void* a, const void* dy, int M, int K, int N, float* dw, float* dbias, void* scratch, long long scratch_bytes, int overwrite) {
    // large result, short reduction, storing form: whole 128 x 128 tiles of dW per block over all rows (dwg_tile.hpp)
    static int dwg_on = -1;
    if (dwg_on < 0) { const char* e = getenv("MI355_DWG"); dwg_on = (e && e[0] == '0') ? 0 : 1; }
    if (overwrite && dwg_on && dtype == MI_BF16 && M >= 1 && K % 128 == 0 && N % 128 == 0 && (long long)(K / 128) * (N / 128) >= 256 &&
        ((((uintptr_t)a) | ((uintptr_t)dy) | ((uintptr_t)dw)) & 15) == 0 && fits_desc((long long)M * K * 2) && fits_desc((long long)M * N * 2)) {
        DwgParams q = {a, (uint32_t)((long long)M * K * 2), dy, (uint32_t)((long long)M * N * 2), dw, dbias, M, K, N, K / 128, N / 128};
        const int per = (q.KT * q.NT + 7) / 8;
        static int nst = -1;
        if (nst < 0) { const char* e = getenv("MI355_DWG_NST"); nst = (e && atoi(e) == 4) ? 4 : 3; }
        if (nst == 3) MI_LAUNCH(dwg_kernel<3>, dim3((unsigned)(per * 8)), dim3(256), 0, (hipStream_t)stream, q);
        else MI_LAUNCH(dwg_kernel<4>, dim3((unsigned)(per * 8)), dim3(256), 0, (hipStream_t)stream, q);
        return mi_check_launch("dwg_kernel");
    }
    // medium results, short reduction (the latent layers of the ConvVAE at any minibatch that is a multiple of 16, the small layers of the MlpVAE): one WAVE per 64 x 64
    // tile over all rows, no LDS, no row split, no scratch, adds or stores in place (dwgs_tile.hpp; MI355_DWGS=0 / key 22: the first-generation kernel below)
    if (dwgs_eligible(dtype, M, K, N)) {
        DwgsParams q = {(const bf16_t*)a, (const bf16_t*)dy, dw, dbias, M, K, N, K / 64, N / 64, overwrite ? 1 : 0};
        const int per = (q.KT * q.NT + 7) / 8, ws = dwgs_wsplit(M);
        if (overwrite) MI_LAUNCH(dwgs_kernel<true>, dim3((unsigned)(per * 8)), dim3(64 * ws), 0, (hipStream_t)stream, q);
        else MI_LAUNCH(dwgs_kernel<false>, dim3((unsigned)(per * 8)), dim3(64 * ws), 0, (hipStream_t)stream, q);
        return mi_check_launch("dwgs_kernel");
    }
    WgradParams p = {};
    p.ones_row = dbias ? 1 : 0; p.dbias = dbias;
    p.big = a; p.frame_idx = nullptr;
    fill_wgrad_geom(p, M, 1, 1, K, 1, 1, 1, 1, 1, false);
    const int vb = dtype == MI_BF16 ? 8 : 4;
    if (K % vb != 0) return mi_fail(MI_ERR_SHAPE, "mi_gemm_wgrad: K must be a multiple of the 16-byte vector (pad K)");
    p.N = N; p.small = dy; p.s_vec = vec_ok(dy, N, dtype); p.out = dw;
    return launch_wgrad((hipStream_t)stream, dtype, 0, p, g_dense_wgrad_blocks, scratch, scratch_bytes, overwrite);
}

// TWO dense filter gradients (+ bias rows) as ONE launch: what mi_gemm_wgrad_bias_ws(problem 0) followed by mi_gemm_wgrad_bias_ws(problem 1) computes, bit for bit, when both take the
// first-generation kernel in its bf16 / 128-row configuration (the latent layers of the ConvVAE: dense1 [B, 64] x [B, 6144] and the heads [B, 6144] x [B, 128]); any other pair of
// shapes: the two single calls.  Round 6: the two ~190-block grids sat back to back at the very end of the backward pass.
int mi_gemm_wgrad_bias_pair_ws(void* stream, int dtype, const void* a0, const void* dy0, int M0, int K0, int N0, float* dw0, float* db0, void* scratch0, long long scratch_bytes0,
                               const void* a1, const void* dy1, int M1, int K1, int N1, float* dw1, float* db1, void* scratch1, long long scratch_bytes1) {
    static int pair_on = -1;
    if (pair_on < 0) { const char* e = getenv("MI355_DENSE_PAIR"); pair_on = (e && e[0] == '0') ? 0 : 1; }
    const void* A[2] = {a0, a1}; const void* DY[2] = {dy0, dy1}; const int M[2] = {M0, M1}, K[2] = {K0, K1}, N[2] = {N0, N1};
    float* DW[2] = {dw0, dw1}; float* DB[2] = {db0, db1}; void* WS[2] = {scratch0, scratch1}; const long long NB[2] = {scratch_bytes0, scratch_bytes1};
    bool ok = pair_on && dtype == MI_BF16;
    WgradPair q = {};
    dim3 g[2];
    for (int i = 0; i < 2 && ok; ++i) {
        if (dwgs_eligible(dtype, M[i], K[i], N[i]) || K[i] % 8 != 0 || (((uintptr_t)A[i]) & 15)) { ok = false; break; }
        WgradParams& p = q.p[i];
        p.ones_row = DB[i] ? 1 : 0; p.dbias = DB[i];
        p.big = A[i]; p.frame_idx = nullptr;
        fill_wgrad_geom(p, M[i], 1, 1, K[i], 1, 1, 1, 1, 1, false);
        p.N = N[i]; p.small = DY[i]; p.s_vec = vec_ok(DY[i], N[i], dtype); p.out = DW[i];
        bool wide = false;
        if (prepare_wgrad(dtype, p, g_dense_wgrad_blocks, WS[i], NB[i], 0, &g[i], &wide) != MI_OK || !wide || p.C % 8 != 0) ok = false;
    }
    if (!ok) {
        const int rc = mi_gemm_wgrad_bias_ws(stream, dtype, a0, dy0, M0, K0, N0, dw0, db0, scratch0, scratch_bytes0);
        return rc != MI_OK ? rc : mi_gemm_wgrad_bias_ws(stream, dtype, a1, dy1, M1, K1, N1, dw1, db1, scratch1, scratch_bytes1);
    }
    q.n0 = (int)(g[0].x * g[0].y * g[0].z);
    for (int i = 0; i < 2; ++i) { q.gx[i] = (int)g[i].x; q.gy[i] = (int)g[i].y; }
    const unsigned nblk = (unsigned)q.n0 + g[1].x * g[1].y * g[1].z;
    MI_LAUNCH((wgrad_pair_kernel<bf16_t, bf16_t, 8, 16, 128>), dim3(nblk), dim3(GEMM_NT), 0, (hipStream_t)stream, q);
    int rc = mi_check_launch("wgrad_pair_kernel");
    for (int i = 0; i < 2 && rc == MI_OK; ++i) {
        const WgradParams& p = q.p[i];
        if (!p.slabs) continue;
        rc = mi_reduce_slabs((hipStream_t)stream, p.slabs, p.slab_stride, (int)g[i].z, (long long)p.Kc * p.N, p.out, 0);
        if (rc == MI_OK && p.ones_row) rc = mi_reduce_slabs((hipStream_t)stream, p.slabs + (long long)p.Kc * p.N, p.slab_stride, (int)g[i].z, (long long)p.N, p.dbias, 0);
    }
    return rc;
}

}  // extern "C"
