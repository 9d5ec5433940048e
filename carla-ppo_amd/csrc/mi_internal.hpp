// mi_internal.hpp — error plumbing shared by the launchers (not part of the public C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

enum { MI_F32 = 0, MI_BF16 = 1, MI_BF16X3 = 2 };   // MI_BF16X3: split storage (hi | lo bf16 halves per 4-byte element), common.hpp
enum { MI_OK = 0, MI_ERR_ARG = -1, MI_ERR_SHAPE = -2, MI_ERR_LAUNCH = -3, MI_ERR_STATE = -4 };

// A kernel launch that carries the caller's completion event ON ITS OWN DISPATCH PACKET (hipExtLaunchKernelGGL's stop event) instead of a separate
// hipEventRecord marker behind it: the VAE engine's backward pass hands every gradient tensor from the caller's stream to the filter-gradient stream, and each
// marker packet cost the producing queue ~6-8 us of bubble (profiles/r03_d, dispatch timeline; round 4).  The engine sets mi_tl_stop_event right before the
// layer call whose (single) kernel produces the tensor; the first MI_LAUNCH of that call consumes it.  Unset (the default): a plain launch.
extern thread_local hipEvent_t mi_tl_stop_event;
#define MI_LAUNCH(kernel, grid, block, shmem, stream, ...) do { \
        if (mi_tl_stop_event) { const hipEvent_t ev__ = mi_tl_stop_event; mi_tl_stop_event = nullptr; \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, ev__, 0, __VA_ARGS__); } \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); } while (0)

extern thread_local const void* mi_tl_rc_wfrag;  // rwconv.hip: fragment-ordered weights for the next conv-form register-weight launch of this thread (mi_rwconv_next_weights_fragment_ordered)
int mi_enc12_debug(int mask);                    // enc12.hip: ablation mask of the fused encoder-head forward kernel (tools/enc12_ablate.py); returns the previous one
int mi_fail(int code, const char* msg);          // records msg (thread-local) and returns code
int mi_check_launch(const char* what);           // hipGetLastError() -> MI_OK / MI_ERR_LAUNCH

// deferred split reductions of the raw-staged filter-gradient kernel (conv_ops.hip; used by the VAE engine)
extern "C" int mi_tapwgrad_defer(int on);        // returns the previous mode
extern "C" int mi_tapwgrad_flush(void* stream);  // launches the recorded reduces on `stream`
extern "C" int mi_tapwgrad_defer_pause(int pause);   // != 0: reduce right behind the launch although the pass defers (a layer issued on another stream); returns the previous setting
extern "C" int mi_tapwgrad_slab_bf16(int on);    // partial-sum slabs rounded to bf16 (the engine's bf16 backward); returns the previous setting

// register-weight kernel of the thin gather-form layers (rwconv.hip): 1 launched, 0 not eligible, < 0 error
int mi_try_rwconv_gather(hipStream_t st, int dtype, const void* a, const void* w, int B, int IH, int IW, int C, int OH, int OW, int N,
                         int KH, int KW, void* out, const float* bias, const void* mask, int relu, const void* mask_bits, void* bits_out);
int mi_try_rwconv_conv(hipStream_t st, int dtype, const void* a, const void* w, int B, int IH, int IW, int C, int OH, int OW, int N,
                       int KH, int KW, int ldb, void* out, const float* bias, const void* mask, int relu);   // rwconv.hip, conv form (32 -> 64 channels)
int mi_rwconv_conv_mode(int set);                // mi_set_tuning key 15: 0 off, 1 k = 5 layers, 2 also k = 4; set < 0 queries
int mi_rwconv_blocks(int set);                   // mi_set_tuning key 16: persistent blocks per XCD (0 = resident maximum); set < 0 queries
int mi_rwconv_mode(int set);                     // mi_set_tuning key 13: 0 off, 1 auto, 2 whenever eligible; set < 0 queries
void mi_get_trace(long long** buf, int* cap);     // the debug stamp buffer of mi_debug_set_trace

// out[0 .. n) += sum over nslab slabs of slabs[k * stride + i] (reduce_small_fused_kernel, tapwgrad_tile.hpp; fixed summation order, no atomics) -- conv_ops.hip
int mi_reduce_slabs(hipStream_t st, const float* slabs, long long stride, int nslab, long long n, float* out, int overwrite = 0);   // overwrite: out = sum
// deferred mode of mi_reduce_slabs (per host thread): jobs are recorded and mi_small_reduce_flush(stream) issues all of them as ONE launch -- conv_ops.hip
extern "C" int mi_small_reduce_defer(int on);
extern "C" int mi_small_reduce_flush(void* stream);
extern "C" int mi_small_reduce_bind(void* stream);      // the list of the deferring pass belongs to `stream`; mi_reduce_slabs calls from other streams launch immediately (round 5)
extern "C" int mi_small_reduce_deferring(void);
bool mi_narrow_enabled();                        // the narrow-layer kernels are switched on (mi_set_tuning key 4 / MI355_NARROW) -- conv_ops.hip
