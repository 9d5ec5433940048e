/* mi355_carla.h — C ABI of libmi355_carla.so: the MI355X (gfx950 / CDNA4) implementation of the
 * ConvVAE + PPO hot path of bitsauce/Carla-ppo.
 *
 * The reference has no FFI layer (pure Python on TensorFlow 1.13); the boundary it exposes is the Python class
 * surface of vae/models.py and ppo.py.  Those classes are re-implemented in carla-ppo_amd/{vae/models.py,ppo.py,utils.py}
 * and call ONLY the functions below through ctypes.  Each entry cites the reference op(s) it replaces
 * (file:line relative to the reference checkout).
 *
 * Conventions
 *   - plain C; raw DEVICE pointers (from torch tensors' data_ptr()) and sizes; no torch types.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream); every call is asynchronous on it.
 *   - return value: 0 = ok, negative = error (MI_ERR_*); message via mi_last_error() (thread local).
 *   - no allocation inside the library: all buffers/workspaces are passed in (sizes via *_floats / *_bytes queries).
 *   - dtype: 0 = fp32 storage + v_mfma_f32_32x32x2_f32 (parity mode), 1 = bf16 storage + v_mfma_f32_32x32x16_bf16
 *     (throughput mode; fp32 accumulate, fp32 master weights / grads / optimiser state), 2 = SPLIT storage ("bf16x3" precision): every
 *     activation / weight element is 4 bytes = two bf16 halves hi + lo of one value (hi = bf16(x), lo = bf16(x - hi); word = hi << 16 | lo),
 *     products formed on the bf16 MFMA pipe as a.b + a.swap16(b) (all four partial products, fp32 accumulate): 16-17 significand bits per
 *     operand at a quarter of the exact-fp32 MFMA time -- the fast mode that meets the 1e-4 tolerance.  Same sizes / alignments as dtype 0.
 *   - layouts are TensorFlow's: activations NHWC, conv kernels HWIO [kh,kw,in,out], transposed-conv kernels
 *     [kh,kw,out,in], dense kernels [in,out].  All convolutions are stride 2, VALID.
 *
 * The binding is generated from THIS file: carla-ppo_amd/mi355/lib.py parses the prototypes below.
 */
#ifndef MI355_CARLA_H
#define MI355_CARLA_H

#ifdef __cplusplus
extern "C" {
#endif

#define MI_F32 0
#define MI_BF16 1
#define MI_BF16X3 2

#define MI_OK 0
#define MI_ERR_ARG (-1)
#define MI_ERR_SHAPE (-2)
#define MI_ERR_LAUNCH (-3)
#define MI_ERR_STATE (-4)


/* ---- engine descriptors (all 4-byte fields; mirrored as ctypes.Structure in carla-ppo_amd/mi355/lib.py) ---- */
typedef struct MiVaeDesc {
    int dtype;          /* MI_F32 | MI_BF16 | MI_BF16X3 */
    int max_batch;      /* largest per-GPU minibatch the workspace is sized for */
    int ih, iw, cin;    /* source frame shape (80,160,3) — NHWC */
    int ct;             /* target depth: 3 (rgb) or 1 (segmentation) */
    int z_dim;
    int loss_kind;      /* 0 bce_loss, 1 bce_loss_v2, 2 mse_loss (vae/models.py:11-22) */
    float beta;
    float kl_tolerance;
    int inference_only; /* != 0: no backward pass will ever run on this engine (VAE(training=False): rollout / evaluation / encode): the workspace carries no
                         * filter-gradient scratch (~0.5 GB less); mi_vae_create then wants grads == NULL.  0 (the zero-initialised default): a training engine */
} MiVaeDesc;

/* MlpVAE (vae/models.py:271-299): dense encoder / decoder around the same latent block; up to MI_MLP_MAX_HIDDEN hidden layers per side */
#define MI_MLP_MAX_HIDDEN 4
typedef struct MiMlpVaeDesc {
    int dtype;          /* MI_F32 | MI_BF16 */
    int max_batch;
    int source_size;    /* prod(source_shape), e.g. 80*160*3 = 38400 */
    int target_size;    /* prod(target_shape) */
    int z_dim;
    int n_enc, n_dec;   /* hidden layers: encoder_sizes (512, 256) -> 2, decoder_sizes (256, 512) -> 2; the output layer to target_size is implied */
    int enc[MI_MLP_MAX_HIDDEN];
    int dec[MI_MLP_MAX_HIDDEN];
    int loss_kind;      /* 0 bce_loss, 1 bce_loss_v2, 2 mse_loss */
    int with_optimizer; /* 0: inference-only engine (no gradient buffers in the workspace) */
    float beta;
    float kl_tolerance;
} MiMlpVaeDesc;

typedef struct MiPpoDesc {
    int max_batch;
    int input_dim;      /* z_dim + measurements = 67 (train.py:85) */
    int num_actions;    /* 2 */
    int h1, h2;         /* (500, 300) for both pi and V trunks (ppo.py:18) */
    float clip_eps;     /* epsilon */
    float value_scale;
    float entropy_scale;
} MiPpoDesc;

/* ---- library ---- */
const char* mi_last_error(void);
int mi_abi_version(void);
int mi_device_info(int device, int* cu_count, int* wave_size, char* arch, int arch_len);
/* CRC-32C of a HOST buffer (running value in, 0 to start): the checksum of the reference's TensorFlow bundle checkpoints
 * (tf.train.Saver files written / read by vae/models.py:154,172-186 and ppo.py:184,202-216); used by mi355/tf_bundle.py.  Returns the crc. */
unsigned int mi_crc32c(unsigned int crc, const void* data, long long n);
/* Box calibration (round 5; SURVEY 8d "confirm on the box"; no reference counterpart): what THIS GPU sustains right now on the two resources the rooflines are priced
 * against -- bench.py prints it as "box" next to every datasheet fraction.  scratch: device buffer, >= 64 MiB, ideally mi_device_probe_scratch_bytes() (6 x the
 * Infinity Cache); millis: length of each of the three MFMA launches (1..50).  SYNCHRONOUS (HIP events).  out8: 0 sustained v_mfma_f32_32x32x16_bf16 rate in TFLOP/s
 * (mean of launches 2 and 3), 1 shader clock during them in MHz (s_memtime / s_memrealtime), 2 / 3 the same for the FIRST launch (clocks as the caller left them),
 * 4 HBM streaming read TB/s, 5 HBM copy TB/s (read + written bytes), 6 compute units, 7 bytes the HBM probes walked. */
long long mi_device_probe_scratch_bytes(void);
int mi_device_probe(void* stream, void* scratch, long long scratch_bytes, int millis, float* out8);
/* Kernel-selection knobs (process-global, not part of the reference surface; defaults come from the environment variables of
 * the same meaning).  key 0: gemm2 LDS-DMA tiles on/off (MI355_GEMM2); key 1: minimum block count for the raw-staged
 * tapconv kernel, -1 = never (MI355_TAPCONV / MI355_TAPCONV_MINBLOCKS); key 2: debug, drop the wgrad_kernel atomics;
 * key 3: raw-staged bf16 weight-gradient kernel on/off (MI355_TAPWGRAD); key 4: narrow-layer kernels on/off (MI355_NARROW);
 * key 5: tapconv tile (0 auto, 1 = 256 positions / 1 block per CU, 2 = 128 positions / 2 blocks per CU);
 * key 6: tapconv epilogue (1 = registers -> 16-byte stores through a half-wave swap, 0 = LDS-staged coalesced stores);
 * key 7: tapwgrad wave layout for 2x2-tap layers (1 = wave per (tap, position half), 0 = wave per (tap, output tile));
 * key 8: tapconv grid, EXPERIMENTAL (0 = one block per tile, 1 = persistent blocks walking contiguous tile ranges, N > 1 = at most N
 * of them per output column; MI355_TAP_PERSIST);
 * key 9: tapwgrad target block count (position splits x block columns; default 256 = one block per CU); key 10: waves per block of
 * the narrow filter-gradient kernel (4 | 8 | 12); key 11: target block count of the dense filter gradients;
 * key 12: tapconv ReluGrad-mask prefetch in the last main-loop step on/off;
 * key 13: register-weight kernel of the thin gather-form layers (rwconv.hip: deconv3 fwd, conv2 dgrad): 0 off, 1 auto (grids that fill the
 * chip), 2 whenever the layer is eligible (MI355_RWCONV); key 14: the column-walk filter-gradient kernel of deconv3 on/off;
 * key 15: register-weight kernel of the conv-form layers (rwconv.hip): 0 off, 1 deconv3 dgrad (32 -> 64 channels, k = 5), 2 also conv2 fwd (k = 4),
 * 3 also the 64 -> 128 channel k = 4 shape, conv3 fwd / deconv2 dgrad (default; MI355_RWCONV_CONV); subject to key 13's off / auto / always; key 16: persistent blocks per XCD of the
 * register-weight kernels, 0 = as many as stay resident (tests use 1: every block then walks several chunks); key 22: the LDS-free one-wave-per-tile dense filter
 * gradient of round 5 (csrc/dwgs_tile.hpp; MI355_DWGS) on / off; key 23 (debug): ablation mask of the fused encoder-head forward kernel's timing instantiation
 * (tools/enc12_ablate.py; any bit set = wrong results; 0 = the product kernel).  Returns the previous value. */
int mi_set_tuning(int key, int value);
/* debug only: s_memtime stamps of the tapconv kernel (32 int64 per wave per block) into a caller-provided device buffer; NULL = off */
int mi_debug_set_trace(void* dev_ptr, int capacity_entries);

/* ---- convolution family (implicit-GEMM on MFMA, LDS-staged tiles) ---- */
/* tf.layers.conv2d k x k, s2, VALID + BiasAdd + Relu — vae/models.py:250-253.  x may be frames gathered through frame_idx: x_is_f32 = 1 fp32
 * frames, 2 = raw uint8 camera frames (value k / 255, the host preprocessing of vae/train_vae.py:15-18 done in registers; bf16 narrow-layer kernel only). */
int mi_conv2d_nhwc_fwd(void* stream, int dtype, const void* x, const int* frame_idx, int x_is_f32, int B, int IH, int IW, int Cin, const void* w, int w_transposed, const float* bias, int KH, int KW, int Cout, int relu, void* out);
/* ReLU bit words (bf16 engine): a forward kernel can also write, per output pixel and group of 16 channels, one uint32 whose bit i (i < 8) says
 * "channel 16 j + 2 i > 0" and bit 16 + i "channel 16 j + 2 i + 1 > 0"; the ReluGrad of the next layer's input gradient then reads those words
 * (1/16 of the bytes) instead of the activation tensor.  relu_bits: [B*OH*OW][Cout/16] uint32, *wrote_bits = 1 when the kernel produced them. */
int mi_conv2d_nhwc_fwd_bits(void* stream, int dtype, const void* x, const int* frame_idx, int x_is_f32, int B, int IH, int IW, int Cin, const void* w, int w_transposed, const float* bias, int KH, int KW, int Cout, int relu, void* out, void* relu_bits, int* wrote_bits);
int mi_conv2d_nhwc_dgrad_bits(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout, const void* w, int KH, int KW, int Cin, int IH, int IW, const void* mask, const void* mask_bits, void* dx);
int mi_deconv2d_nhwc_fwd_bits(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin, const void* w, const float* bias, int KH, int KW, int Cout, int relu, void* out, void* relu_bits, int* wrote_bits);
int mi_deconv2d_nhwc_dgrad_bits(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout, const void* w, int w_transposed, int KH, int KW, int Cin, const void* mask, const void* mask_bits, void* dx);
/* Conv2DBackpropInput (+ fused ReluGrad of the layer below through `mask`) — backward of vae/models.py:250-253 */
int mi_conv2d_nhwc_dgrad(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout, const void* w, int KH, int KW, int Cin, int IH, int IW, const void* mask, void* dx);
/* Conv2DBackpropFilter: dw += im2col(x)^T dy.  NON-DETERMINISTIC summation order: the position splits meet in fp32 atomics (two runs may differ in the last bits);
 * the _ws form below (caller scratch) is the deterministic one and the only one the engines use */
int mi_conv2d_nhwc_wgrad(void* stream, int dtype, const void* x, const int* frame_idx, int x_is_f32, int B, int IH, int IW, int Cin, const void* dy, int KH, int KW, int Cout, float* dw);
/* same, with caller scratch for the split reduction (every kernel generation and storage type: per-split partial sums added in a fixed order, no atomics,
 * bitwise reproducible; 64 MiB covers every layer of the model at any batch size; may be NULL) and, when dbias != NULL,
 * the BiasAddGrad of the same layer (dbias[n] += sum dy) fused into the kernel where it is eligible, else run as mi_colsum */
int mi_conv2d_nhwc_wgrad_ws(void* stream, int dtype, const void* x, const int* frame_idx, int x_is_f32, int B, int IH, int IW, int Cin, const void* dy, int KH, int KW, int Cout, float* dw, void* scratch, long long scratch_bytes, float* dbias);
/* tf.layers.conv2d_transpose k x k, s2, VALID + BiasAdd (+ Relu) — vae/models.py:261-264 */
int mi_deconv2d_nhwc_fwd(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin, const void* w, const float* bias, int KH, int KW, int Cout, int relu, void* out);
/* conv2d_transpose into the 1- or 3-channel logits WITH the reconstruction loss of vae/models.py:11-22,123-128 fused into its epilogue
 * (labels: fp32 target frames [*, OH*OW*Cout], optional gather through frame_idx): writes logits, dlogits (NULL = loss only), one loss
 * partial per block into loss_partial[] and 4 floats of per-channel dlogits sums per block into bias_partial[].  *n_partial = number of
 * blocks written, or 0 when the layer is not eligible for the fused kernel (nothing was launched: call the two ops separately). */
int mi_deconv2d_nhwc_fwd_bce(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin, const void* w, const float* bias, int KH, int KW, int Cout, void* logits, const float* labels, const int* frame_idx, long long label_stride, int loss_kind, float inv_batch, void* dlogits, float* loss_partial, float* bias_partial, int partial_capacity, int* n_partial);
/* same; labels_u8 != 0: labels are raw uint8 frames (label_stride in bytes), normalised to exactly float32(k) / float32(255) in registers; logits == NULL: the logits are not stored */
int mi_deconv2d_nhwc_fwd_bce_u8(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin, const void* w, const float* bias, int KH, int KW, int Cout, void* logits, const void* labels_any, int labels_u8, const int* frame_idx, long long label_stride, int loss_kind, float inv_batch, void* dlogits, float* loss_partial, float* bias_partial, int partial_capacity, int* n_partial);
/* The decoder tail of a TRAINING step in one launch (round 3): conv2d_transpose into the 3-channel logits (vae/models.py:264), the reconstruction loss of
 * vae/models.py:11-22,123-128, and both gradients of that layer behind tf.gradients (vae/models.py:142: Conv2D of dlogits + ReluGrad -> dx, the gradient
 * wrt the previous layer's pre-activation; Conv2DBackpropFilter -> dw +=) -- logits and dlogits never leave the chip.  x: [B,IH,IW,32] (post-ReLU: it is
 * its own ReluGrad mask); w: [kh,kw,out,in]; w_t: the K-contiguous copy [in][kh*kw*out]; labels as mi_deconv2d_nhwc_fwd_bce_u8; loss_partial[] /
 * bias_partial[][4] per block as there; scratch >= mi_deconv2d_tail_blocks() * 6144 bytes.  *n_partial = blocks written, or 0 when the layer is not
 * eligible (bf16 storage, 32 -> 3 channels, k = 4 only; nothing was launched: use the separate ops). */
int mi_deconv2d_tail_blocks(void);
int mi_deconv2d_tail_fused(void* stream, int dtype, const void* x, int B, int IH, int IW, int Cin, const void* w, const void* w_t, const float* bias, int KH, int KW, int Cout, const void* labels, int labels_u8, const int* frame_idx, long long label_stride, int loss_kind, float inv_batch, void* dx, float* dw, float* loss_partial, float* bias_partial, int partial_capacity, int* n_partial, void* scratch, long long scratch_bytes, int reduce_now);
/* reduce_now = 0 above leaves the per-block filter-gradient partial sums in scratch; this adds them to dw (anywhere behind that launch on the same
 * stream order and in front of the optimiser step) */
int mi_deconv2d_tail_reduce(void* stream, const void* scratch, int n_partial, float* dw);
/* The encoder head of a backward pass in ONE launch (csrc/enchead_tile.hpp; replaces Conv2DBackpropInput of conv2 + Conv2DBackpropFilter / BiasAddGrad of conv1,
 * vae/models.py:250-251 behind :142): frames [*, FH, FW, 3] (frames_fmt 1 = fp32, 2 = uint8 camera bytes; optionally gathered through frame_idx) -conv1 k4 s2-> 32 ch
 * -conv2 k4 s2-> 64 ch; dy2 = gradient wrt conv2's pre-activation [B, OH, OW, 64], w2 = conv2's kernel [4][4][32][64], bits_act1 = ReLU bit words of conv1's output.
 * dw1 [4][4][3][32] += , db1 [32] += .  The gradient of conv1's output is never written.  *n_blocks = 0: not eligible (nothing launched; call
 * mi_conv2d_nhwc_dgrad_bits and mi_conv2d_nhwc_wgrad_ws).  scratch: >= mi_conv2d_head_bwd_blocks() * 8320 bytes. */
int mi_conv2d_head_bwd_blocks(void);
int mi_conv2d_head_bwd_fused(void* stream, int dtype, const void* frames, int frames_fmt, const int* frame_idx, int B, int FH, int FW, const void* dy2, const void* w2, const void* bits_act1, float* dw1, float* db1, void* scratch, long long scratch_bytes, int* n_blocks);
/* The encoder head of a FORWARD pass in ONE launch (round 5, csrc/enc12_tile.hpp; replaces the two tf.layers.conv2d of vae/models.py:250-251): frames
 * [*, 80, 160, 3] as uint8 camera bytes (frames_fmt 2) or float32 (1), optionally gathered through frame_idx, -conv1 k4 s2 + bias + ReLU-> act1 [B, 39, 79, 32] kept in LDS -conv2 k4 s2 + bias + ReLU-> act2
 * [B, 18, 38, 64].  w1_t / w2_t: the K-contiguous kernel copies [32][48] / [64][512] (mi_transpose_weights).  act1 and (relu_bits1 != NULL) its ReLU bit words are still
 * written, bit for bit as mi_conv2d_nhwc_fwd_bits writes them (the backward pass reads them); act2 = conv2 of that activation.  *launched = 0: not eligible (bf16 storage,
 * this geometry only; nothing was launched: call mi_conv2d_nhwc_fwd_bits and mi_conv2d_nhwc_fwd). */
int mi_conv2d_enc12_fwd(void* stream, int dtype, const void* frames, int frames_fmt, const int* frame_idx, int B, int FH, int FW, const void* w1_t, const float* b1, const void* w2_t, const float* b2, void* act1, void* relu_bits1, void* act2, int* launched);
/* The four SMALL-GRID layers of the ConvVAE on the activation-resident kernels (round 4, csrc/ares_tile.hpp): a block keeps the whole inputs of a group of frames
 * in LDS and streams fragment-ordered weights through registers.  form 0 (conv form, k4 s2, [B,8,18,128] -> [B,3,8,256]): conv4 forward (vae/models.py:253) and
 * deconv1's input gradient (Conv2D of dy behind :142); form 1 (gather form, [B,3,8,256] -> [B,8,18,128]): deconv1 forward (:261) and conv4's input gradient
 * (Conv2DBackpropInput).  mi_ares_pack_weights rewrites a [4][4][128][256] fp32 master kernel into the form's fragment order (mi_ares_weight_bytes() of bf16;
 * once per optimiser step); mi_ares_conv: out = mask(relu?(op(x) + bias)), *launched = 0 when the call is not eligible (bf16 storage and this geometry only;
 * nothing was launched: use the general ops above / below). */
long long mi_ares_weight_bytes(void);
int mi_ares_pack_weights(void* stream, int form, const float* w_fp32, void* wf_out);
/* the four copies the VAE engine keeps, in one launch: wf0 conv4 forward (form 0 of conv4's kernel), wf1 conv4 input gradient (form 1 of it), wf2 deconv1 forward (form 1 of
 * deconv1's kernel), wf3 deconv1 input gradient (form 0 of it) */
int mi_ares_pack_weights4(void* stream, const float* conv4_w, const float* deconv1_w, void* wf0, void* wf1, void* wf2, void* wf3);
/* form 2 of mi_ares_conv / mi_ares_pack_weights: the MID-layer gather form, [B,8,18,128] -> [B,18,38,64] (deconv2 forward, vae/models.py:262, and conv3's input
 * gradient): w is [4][4][64][128] read as [kh][kw][n][c] (deconv2's [kh,kw,out,in] kernel, or conv3's HWIO kernel); its fragment copy is 256 KB.
 * mi_ares_pack_weights6 = mi_ares_pack_weights4 + those two copies (conv3_w -> wf4, deconv2_w -> wf5; either pair may be NULL) in one launch. */
int mi_ares_pack_weights6(void* stream, const float* conv4_w, const float* deconv1_w, const float* conv3_w, const float* deconv2_w, void* wf0, void* wf1, void* wf2, void* wf3, void* wf4, void* wf5);
/* round 6: ... plus the conv-form copies of the same two mid-layer kernels for the register-weight kernel (form 3: the order rwconv_conv_kernel<4, 2> loads its 64 weight fragments in, 256 KB
 * each): conv3_w -> wf6 (conv3 forward, vae/models.py:252), deconv2_w -> wf7 (deconv2's input gradient, Conv2D of dy behind :262).  NULL pairs are skipped. */
int mi_ares_pack_weights8(void* stream, const float* conv4_w, const float* deconv1_w, const float* conv3_w, const float* deconv2_w, void* wf0, void* wf1, void* wf2, void* wf3, void* wf4, void* wf5, void* wf6, void* wf7);
/* The NEXT register-weight launch of the calling thread reads its weights from `wf` instead of the K-contiguous copy: 1 KB contiguous per wave load in the kernel's prologue
 * (the K-contiguous copy gives 16 B per lane at a 1-2 KB lane stride).  Which form `wf` must hold follows from the call:
 *   form 3 (above)  mi_conv2d_nhwc_fwd[_bits] of a 64 -> 128 channel k = 4 layer (conv3), mi_deconv2d_nhwc_dgrad[_bits] of a 128 -> 64 channel one (deconv2)
 *   form 4          mi_deconv2d_nhwc_fwd of the k = 5, 64 -> 32 channel layer (deconv3, vae/models.py:263): mi_ares_pack_weights(form 4) of its [5][5][32][64] kernel,
 *                   144 fragments of 1 KB ordered (output-parity class, tap of the 3 x 3 class window, 16-channel K step); fragments of taps outside the 5 x 5 kernel are never read
 *   form 6          mi_deconv2d_nhwc_dgrad[_bits] of the k = 5, 32 -> 64 channel layer (deconv3's input gradient): mi_ares_pack_weights(form 6) of deconv3's kernel read as [800][64],
 *                   2 x 50 fragments (32-wide output tile, the 50 live (tap, k-step) pairs of the 3 x 3 slot taps in the prologue's order)
 *   form 5          mi_conv2d_enc12_fwd: conv2's [4][4][32][64] HWIO kernel (vae/models.py:251), mi_ares_pack_weights(form 5), 64 fragments (32-wide output half, K step)
 * Consumed by that call whether or not the kernel it picks uses it; NULL clears.  Same values in the same registers: results are bit-identical either way. */
int mi_rwconv_next_weights_fragment_ordered(const void* wf);
int mi_ares_conv(void* stream, int dtype, int form, const void* x, int B, const void* wf, const float* bias, int relu, const void* mask, void* out, int* launched);
/* backward of conv2d_transpose wrt its input (= a plain s2 conv of dy) with fused ReluGrad mask */
int mi_deconv2d_nhwc_dgrad(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout, const void* w, int w_transposed, int KH, int KW, int Cin, const void* mask, void* dx);
/* backward of conv2d_transpose wrt its kernel: dw[kh,kw,co,ci] += im2col(dy)^T x.  The form WITHOUT scratch is NON-DETERMINISTIC (fp32 atomics across the position
 * splits); the _ws form is the deterministic one (per-split partial sums added in a fixed order) */
int mi_deconv2d_nhwc_wgrad(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout, const void* x, int KH, int KW, int Cin, float* dw);
int mi_deconv2d_nhwc_wgrad_ws(void* stream, int dtype, const void* dy, int B, int OH, int OW, int Cout, const void* x, int KH, int KW, int Cin, float* dw, void* scratch, long long scratch_bytes, float* dbias);
/* tf.layers.dense (MatMul + BiasAdd + Relu) and its input gradient — vae/models.py:97-98,259; utils.py:25-28; ppo.py:43-55.
 * w_layout 0: W[K,N]; 1: W[N,K] (x * W^T).  nsplit > 1: split-K raw fp32 slabs out[nsplit][M][N]. */
int mi_gemm_bias_act(void* stream, int dtype, const void* a, int M, int K, const void* w, int w_layout, int N, const float* bias, int relu, const void* mask, void* out, int out_f32, int nsplit);
/* the finishing pass of a split-K dense layer: out[m,n] = mask(act(sum_s slabs[s][m][n] + bias[n])) -- the bias / ReLU / ReluGrad epilogue mi_gemm_bias_act
 * cannot apply to raw slabs (MlpVAE, vae/models.py:271-299: the 38400-long reductions of its first layer and of its last layer's input gradient) */
int mi_splitk_finish(void* stream, int dtype, const float* slabs, int nsplit, int M, int N, const float* bias, int relu, const void* mask, void* out, int out_f32);
/* dense kernel gradient dw[K,N] += a^T dy (MatMul's weight gradient behind tf.gradients -- vae/models.py:142, ppo.py:143-144).  NON-DETERMINISTIC summation order: the row splits meet in fp32 atomics; use the _ws forms below */
int mi_gemm_wgrad(void* stream, int dtype, const void* a, const void* dy, int M, int K, int N, float* dw);
/* same with caller scratch (>= mi_gemm_wgrad_scratch_bytes; may be NULL = the form above): every row split stores its partial sums and one pass adds them to dw
 * in a fixed order -- two runs are bitwise equal (round 4; the parity engines use only this form) */
long long mi_gemm_wgrad_scratch_bytes(int dtype, int M, int K, int N);
int mi_gemm_wgrad_ws(void* stream, int dtype, const void* a, const void* dy, int M, int K, int N, float* dw, void* scratch, long long scratch_bytes);
/* same + the layer's BiasAddGrad in the same launch: dbias[n] += sum_m dy[m,n] (a column of ones appended to `a` inside the kernel's loader; dbias may be NULL) */
int mi_gemm_wgrad_bias_ws(void* stream, int dtype, const void* a, const void* dy, int M, int K, int N, float* dw, float* dbias, void* scratch, long long scratch_bytes);
/* round 6: TWO such gradients as ONE launch -- exactly what the two single calls compute (problem 0, then problem 1), bit for bit: the ConvVAE's backward pass ends with dense1's
 * (vae/models.py:259) and the heads' (:97-98) filter + bias gradients, two ~190-block grids of a latency-bound kernel back to back on the caller's stream.  Pairs the one-launch form
 * does not cover (other dtypes / shapes) run as the two single calls. */
int mi_gemm_wgrad_bias_pair_ws(void* stream, int dtype, const void* a0, const void* dy0, int M0, int K0, int N0, float* dw0, float* dbias0, void* scratch0, long long scratch_bytes0,
                               const void* a1, const void* dy1, int M1, int K1, int N1, float* dw1, float* dbias1, void* scratch1, long long scratch_bytes1);
/* same; overwrite != 0: dw (and dbias) = the gradient instead of += (plain stores / the storing form of the ordered sum): the gradient buffer need not be zeroed and no
 * element is touched by an atomic.  Row splits then NEED the scratch (MI_ERR_ARG otherwise). */
int mi_gemm_wgrad_bias_set(void* stream, int dtype, const void* a, const void* dy, int M, int K, int N, float* dw, float* dbias, void* scratch, long long scratch_bytes, int overwrite);

/* ---- VAE elementwise / reduction kernels ---- */
/* Normal(mean, exp(.5 lv)).sample + kl_divergence — vae/models.py:7-9,101-105 (eps injected; TF RNG is unseeded) */
int mi_vae_reparam_kl_fwd(void* stream, int dtype, const float* heads, int nsplit, const float* bias_mean, const float* bias_lv, const float* eps, int sample, int B, int Z, float* mean, float* logvar, void* z, float* kl_row);
/* same with the engine-side noise source: eps == NULL and sample != 0 draws N(0,1) from the Philox4x32-10 stream in rng_state (device, 4 x uint64:
 * seed, next element offset, 2 words of kernel bookkeeping) and stores the draw in eps_out [B,Z] for the backward pass — tfp Normal.sample, vae/models.py:101-105 */
int mi_vae_reparam_kl_fwd_rng(void* stream, int dtype, const float* heads, int nsplit, const float* bias_mean, const float* bias_lv, const float* eps, int sample, int B, int Z, float* mean, float* logvar, void* z, float* kl_row, unsigned long long* rng_state, float* eps_out);
/* out[i] = N(0,1) element (offset + i) of Philox stream `seed` (exploration noise of ppo.py:58-60, tests) */
int mi_normal_philox(void* stream, unsigned long long seed, unsigned long long offset, float* out, long long n);
/* the two halves behind ONE entry point, as SURVEY 8b names it: heads != NULL runs the forward (as mi_vae_reparam_kl_fwd), dz_slabs != NULL runs the backward (as
 * mi_vae_reparam_kl_bwd, on the mean / logvar / kl_row the forward half of this or an earlier call left in the same buffers); both: forward, then backward */
int mi_vae_reparam_kl_fwd_bwd(void* stream, int dtype, const float* heads, int nsplit, const float* bias_mean, const float* bias_lv, const float* eps, int sample, int B, int Z, float* mean, float* logvar, void* z, float* kl_row, const float* dz_slabs, int dz_nsplit, float beta, float kl_floor, float inv_batch, void* dheads);
int mi_vae_reparam_kl_bwd(void* stream, int dtype, const float* dz_slabs, int nsplit, const float* mean, const float* logvar, const float* eps, const float* kl_row, float beta, float kl_floor, float inv_batch, int B, int Z, void* dheads);
/* bce_loss / bce_loss_v2 / mse_loss + reduce_sum(axis=1) + gradient — vae/models.py:11-22,123-128 */
int mi_recon_loss_chunks(int P);
int mi_bce_logits_fwd_bwd(void* stream, int dtype, const void* logits, const float* labels, const int* frame_idx, long long label_stride, int B, int P, int loss_kind, float inv_batch, void* dlogits, float* partial);
/* same + the BiasAddGrad of the layer that produced the logits: dbias[c] += sum of the stored dlogits of channel c (channels = 1..3) */
/* same with the labels as raw uint8 camera bytes (label_stride in values = bytes): float32(k) / float32(255) formed exactly in registers */
int mi_bce_logits_fwd_bwd_u8(void* stream, int dtype, const void* logits, const unsigned char* labels, const int* frame_idx, long long label_stride, int B, int P, int loss_kind, float inv_batch, void* dlogits, float* partial);
int mi_bce_logits_fwd_bwd_bias(void* stream, int dtype, const void* logits, const float* labels, const int* frame_idx, long long label_stride, int B, int P, int loss_kind, float inv_batch, void* dlogits, float* partial, int channels, float* dbias);
/* reduce_mean over the batch, kl_tolerance clamp, tf.metrics.mean accumulators — vae/models.py:124-137,145-146 */
int mi_vae_finalize_losses(void* stream, const float* partial, int nchunks, const float* kl_row, float kl_floor, int B, float inv_batch, float* out2, float* metrics3, float metric_weight);
/* same over a flat list of loss partial sums; optionally folds per-block channel sums [n][4] of a fused loss pass into dbias */
int mi_vae_finalize_losses_flat(void* stream, const float* partial, int n_partial, const float* kl_row, float kl_floor, int B, float inv_batch, float* out2, float* metrics3, float metric_weight, const float* bias_partial, int n_bias_partial, int channels, float* dbias);
/* tf.train.AdamOptimizer ApplyAdam x N fused over one flat buffer — vae/models.py:141-142, ppo.py:143-144 */
int mi_adam_tf_flat(void* stream, float* param, float* m, float* v, float* grad, long long n, float alpha, float beta1, float beta2, float epsilon, void* bf16_shadow, int clear_grad);
/* same; alpha_dev != NULL: the step size is read from device memory (a captured step is replayed with a new value) */
int mi_adam_tf_flat_dev(void* stream, float* param, float* m, float* v, float* grad, long long n, float alpha, const float* alpha_dev, float beta1, float beta2, float epsilon, void* bf16_shadow, int clear_grad);
/* same; the shadow weight copy the MFMA kernels read is of storage type shadow_dtype: MI_BF16 (2 bytes per weight) or MI_BF16X3 (split, 4 bytes) */
int mi_adam_tf_flat_shadow(void* stream, float* param, float* m, float* v, float* grad, long long n, float alpha, const float* alpha_dev, float beta1, float beta2, float epsilon, void* shadow, int shadow_dtype, int clear_grad);
/* TF ApplyAdam (tf.train.AdamOptimizer.minimize, vae/models.py:140-142) over the flat buffer [0, n) that ALSO refreshes both weight copies of the `count` (<= 16) listed [K, N] kernels (offsets in floats, ascending, N % 4 == 0,
 * offsets % 4 == 0) in the same launch: shadow = storage-type copy in the master layout (NULL for fp32 engines), wt = K-contiguous copy wt[off + n * K + k] (what
 * mi_transpose_weights writes; may be NULL).  dtype MI_F32 | MI_BF16 = element type of both copies.  skip (may be NULL): per kernel, bit 0 = do not write its shadow
 * copy, bit 1 = do not write its K-contiguous copy (copies that nobody reads).  Bit-identical p / m / v to mi_adam_tf_flat. */
int mi_adam_tf_layouts(void* stream, int dtype, float* param, float* m, float* v, float* grad, long long n, const long long* offsets, const int* K, const int* N, const int* skip, int count, float alpha, const float* alpha_dev, float beta1, float beta2, float epsilon, void* shadow, void* wt, int clear_grad);
/* same + FRAGMENT-ORDERED bf16 copies of some kernels for the activation-resident convolutions (round 5): frag_ptrs[2 i + q] / frag_forms[2 i + q], q = 0, 1, for kernel i -- what
 * mi_ares_pack_weights(form, master kernel) writes (form 0 | 1: a [2048, 256] kernel, 2 | 3: a [1024, 128] kernel, 4 | 6: deconv3's [800, 64], 5: conv2's [512, 64]), emitted by the optimiser launch from the tile it holds anyway;
 * NULL pointer = none.  Both arrays may be NULL (= mi_adam_tf_layouts). */
int mi_adam_tf_layouts_frag(void* stream, int dtype, float* param, float* m, float* v, float* grad, long long n, const long long* offsets, const int* K, const int* N, const int* skip, int count, float alpha, const float* alpha_dev, float beta1, float beta2, float epsilon, void* shadow, void* wt, int clear_grad, void* const* frag_ptrs, const int* frag_forms);
/* out[b, :] = storage_type(src[idx[b], :]) for b < B (idx NULL: rows 0 .. B-1): the frame rows of a minibatch (the feed_dict slice of vae/models.py:211-216) gathered and
 * converted in one launch; dtype MI_F32 | MI_BF16.  The table's row count is not an argument: idx[b] must lie inside the table (the host mirror builds every index vector from
 * arange(N) permutations, vae/models.py:207-212) */
int mi_gather_rows_cast(void* stream, int dtype, const float* src, const int* idx, int B, long long row_len, void* out);
/* the same from a uint8 table of raw camera bytes: out[b, :] = storage_type(float32(src[idx[b], :]) / float32(255)), correctly rounded (vae/train_vae.py:15-18 on the device) */
int mi_gather_rows_cast_u8(void* stream, int dtype, const unsigned char* src, const int* idx, int B, long long row_len, void* out);
int mi_cast_f32_to_bf16(void* stream, const float* src, void* dst, long long n);
/* fp32 <-> split storage (dtype MI_BF16X3): word = bf16(x) << 16 | bf16(x - bf16(x)); back: hi + lo */
int mi_cast_f32_to_split(void* stream, const float* src, void* dst, long long n);
int mi_cast_split_to_f32(void* stream, const void* src, float* dst, long long n);
/* raw uint8 frames -> float32(k) / float32(255), correctly rounded: the reference's host preprocessing (vae/train_vae.py:15-18) done on the device */
int mi_u8_to_unit_f32(void* stream, const unsigned char* src, float* dst, long long n);
/* K-contiguous copies of the [K,N] kernels for the MFMA B operand: dst[off + n*K + k] = (T) src[off + k*N + n], count <= 16 tensors */
int mi_transpose_weights(void* stream, int dtype, const float* src, void* dst, const long long* offsets, const int* K, const int* N, int count);
/* BiasAddGrad: out[n] += sum_m x[m,n].  NON-DETERMINISTIC summation order (row blocks meet in fp32 atomics); mi_colsum_ws is the deterministic form */
int mi_colsum(void* stream, int dtype, const void* x, long long M, int N, float* out);
/* same with caller scratch (>= mi_colsum_scratch_bytes; NULL = the form above): per-block column sums added up in a fixed order -- bitwise reproducible */
long long mi_colsum_scratch_bytes(int dtype, long long M, int N);
int mi_colsum_ws(void* stream, int dtype, const void* x, long long M, int N, float* out, void* scratch, long long scratch_bytes);
/* tf.nn.sigmoid(reconstructed_logits) — vae/models.py:113 */
int mi_sigmoid(void* stream, int dtype, const void* x, float* out, long long n);
/* verify_range — vae/models.py:24-30 */
int mi_range_check(void* stream, const float* x, long long n, float lo, float hi, int* flag);

/* ---- PPO kernels ---- */
/* prob ratio, clipped surrogate, value loss, entropy, and their gradients wrt the heads — ppo.py:58-66,112-132 */
int mi_ppo_loss_blocks(int M);
int mi_ppo_loss_partial_floats(int M);
int mi_ppo_loss_fwd_bwd(void* stream, const float* u, const float* u_old, const float* logstd, const float* logstd_old, const float* vraw, const float* actions, const float* returns, const float* advantage, const float* low, const float* high, int M, int A, float clip_eps, float value_scale, float entropy_scale, float inv_m, float grad_scale, float* du, float* dv, float* partial, float* losses5, float* dlogstd);
/* build_mlp trunk (utils.py:25-28 with ppo.py:42-44,51-53: two dense layers, ReLU on both) as op-level calls; exact fp32.  din, H1, H2 multiples of 4 (pad the state).
 * fwd: h1 = relu(x W1 + b1), h2 = relu(h1 W2 + b2).  bwd: from g2 = dL/dh2 [M, H2]: dW1, db1, dW2, db2 are ACCUMULATED into; scratch M * (H1 + H2) floats.
 * (The engines run the fused forms: mi_ppo_train_step / mi_ppo_predict / mi_rollout_step.) */
int mi_mlp_policy_fwd(void* stream, const float* x, int M, int din, const float* W1, const float* b1, int H1, const float* W2, const float* b2, int H2, float* h1, float* h2);
int mi_mlp_policy_bwd(void* stream, const float* x, int M, int din, const float* W2, int H1, int H2, const float* h1, const float* h2, const float* g2, float* dW1, float* db1, float* dW2, float* db2, float* scratch);
/* action_mean rescale, Normal.sample, clip_by_value — ppo.py:47,58-62 */
int mi_policy_head(void* stream, const float* u, const float* logstd, const float* noise, const float* low, const float* high, int M, int A, int greedy, float* action, float* mean_out);
/* compute_gae — utils.py:45-50 (fp64, rounding sequence of numpy + scipy.signal.lfilter) */
int mi_gae_scan(void* stream, const double* rewards, const double* values, const double* terminals, int R, int T, double gamma, double lam, double* adv);
/* returns = adv + values; advantage normalisation — train.py:176-177 (fp64, population std, per row) */
int mi_adv_normalize(void* stream, double* adv, const double* values, int R, int T, double* returns);


/* ---- ConvVAE engine: the graph of VAE.__init__ + ConvVAE (vae/models.py:85-142,249-266) on caller-owned buffers ---- */
int mi_vae_desc_size(void);
int mi_vae_tensor_count(void);
long long mi_vae_param_floats(const MiVaeDesc* d);
int mi_vae_param_layout(const MiVaeDesc* d, long long* offsets, long long* sizes, int n);
long long mi_vae_workspace_bytes(const MiVaeDesc* d);
void* mi_vae_create(const MiVaeDesc* d, float* params, float* grads, float* adam_m, float* adam_v, void* bf16_shadow, void* weights_t, void* workspace, long long workspace_bytes);
void mi_vae_destroy(void* h);
int mi_vae_sync_shadow(void* h, void* stream);
void* mi_vae_buffer(void* h, int which);
/* Debug mode of the engine workspace (sanitizer row): with MI355_DEBUG_GUARDS=1 in the environment of BOTH mi_vae_workspace_bytes and mi_vae_create every
 * workspace region is followed by 256 guard bytes of a known pattern (armed by mi_vae_create).  *n_regions = guarded regions (0: mode off), *n_bad = guards that no longer
 * hold the pattern (a kernel wrote past the region in front of it; mi_last_error() names the first); guard_index >= 0 additionally returns that guard's byte offset.
 * Synchronises the device; never used on the production path. */
int mi_vae_debug_check_guards(void* h, int* n_regions, int* n_bad, int guard_index, long long* guard_offset);
/* forward + ELBO terms: the per-minibatch sess.run of VAE.evaluate (vae/models.py:226-229) / forward half of train_step (:213-216) */
/* frames_u8 != 0: src / tgt are raw uint8 frame tables (bf16 engine, rgb target == source format); eps == NULL with sample != 0: the engine draws the noise (mi_vae_set_seed) */
int mi_vae_forward(void* h, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps, int sample, int want_grad, float* metrics3, float metric_weight);
/* gradients of loss = recon + beta*kl wrt all 22 variables (optimizer.minimize, vae/models.py:142); part 0 all, 1 decoder half,
 * 2 encoder half = 3 (heads + conv4) then 4 (conv3..conv1): the data-parallel host all-reduces a finished part's bucket under the next part */
int mi_vae_backward(void* h, void* stream, const void* src, const int* idx, const float* eps, float inv_batch, int part);
int mi_vae_apply_adam(void* h, void* stream, float alpha, float beta1, float beta2, float epsilon);
/* seed of the engine's own N(0,1) source (TF's graph-level seed, train.py:50-51) */
int mi_vae_set_seed(void* h, unsigned long long seed);
/* one whole SGD step = the reference's sess.run([train_step, ...]) (vae/models.py:213-216) in ONE call (eager launches on the caller's stream + the engine's filter-gradient stream; nothing synchronises the host) */
int mi_vae_train_step(void* h, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps, float alpha, float beta1, float beta2, float epsilon, float* metrics3, float metric_weight);
/* the data-parallel form of the same step in ONE call (no reference counterpart: the reference is single-process): this rank's rows with inv_batch = 1 / B_global,
 * backward in the three parts of mi_vae_dp_buckets, each finished bucket's all-reduce on the communicator's own stream under the next part, mi_comm_wait, TF-Adam.
 * mi_vae_dp_buckets (pure, no GPU): out9 = 3 x {part, first float, one past the last float of the flat gradient buffer} in completion order. */
int mi_vae_dp_buckets(const MiVaeDesc* d, long long* out9);
int mi_vae_train_step_dp(void* h, void* comm, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps, float alpha, float beta1, float beta2, float epsilon, float* metrics3, float metric_weight);
/* VAE.encode / generate_from_latent (= north_star "decode") / reconstruct — vae/models.py:188-202 */
int mi_vae_encode(void* h, void* stream, const void* src, int frames_u8, const int* idx, int B, float* mean_out);
int mi_vae_decode(void* h, void* stream, const float* z, int B, float* recon_out);
int mi_vae_reconstruct(void* h, void* stream, const void* src, int frames_u8, const int* idx, int B, const float* eps, int sample, float* recon_out);

/* ---- MlpVAE engine (round 4; vae/models.py:271-299 on the base graph :85-142): the same surface as the ConvVAE engine for the dense variant.  Frame tables are float32 (or, round 5, uint8 camera bytes: frames_u8)
 * [n_frames, source_size] / [n_frames, target_size] on the device; the noise eps [B, z_dim] comes from the caller (mi_normal_philox); gradients are STORED into the
 * gradient buffer by every backward pass (not accumulated).  Tensor order of mi_mlpvae_param_layout: encoder layers {kernel [K, N], bias}, heads {kernel [K, 2 z] =
 * [mean | logstd_sqare], bias}, decoder layers incl. the output layer {kernel, bias}; no padding between tensors. ---- */
int mi_mlpvae_desc_size(void);
long long mi_mlpvae_param_floats(const MiMlpVaeDesc* d);
int mi_mlpvae_tensor_count(const MiMlpVaeDesc* d);
int mi_mlpvae_param_layout(const MiMlpVaeDesc* d, long long* offsets, long long* sizes, int n);
long long mi_mlpvae_workspace_bytes(const MiMlpVaeDesc* d);
void* mi_mlpvae_create(const MiMlpVaeDesc* d, float* params, float* grads, float* adam_m, float* adam_v, void* shadow, void* weights_t, void* workspace, long long workspace_bytes);
void mi_mlpvae_destroy(void* h);
int mi_mlpvae_sync_shadow(void* h, void* stream);
void* mi_mlpvae_buffer(void* h, int which);
long long mi_mlpvae_decoder_offset(void* h);
/* round 5: src / tgt are `const void*` + frames_u8 (bit 0: src is a uint8 table of raw camera bytes, bit 1: tgt is; 0: float32 tables as before): the bytes are normalised to
 * float32(k) / float32(255) -- the reference's host preprocessing, vae/train_vae.py:15-18 -- where the minibatch rows are staged / inside the loss kernel */
int mi_mlpvae_forward(void* h, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps, int sample, int want_grad, float* metrics3, float metric_weight);
int mi_mlpvae_backward(void* h, void* stream, const float* eps, float inv_batch, int part);
int mi_mlpvae_apply_adam(void* h, void* stream, float alpha, float beta1, float beta2, float epsilon);
int mi_mlpvae_train_step(void* h, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps, float alpha, float beta1, float beta2, float epsilon, float* metrics3, float metric_weight);
/* round 6 (ABI 7): one DATA-PARALLEL SGD step of the MlpVAE (vae/models.py:271-299 behind :140-142,213-216) in ONE call -- forward + ELBO of this rank's rows (inv_batch =
 * 1 / B_global), the decoder half of the backward pass, its gradients' all-reduce on the communicator's own stream under the encoder half, that half's all-reduce, the join,
 * TF-Adam.  mi_mlpvae_dp_buckets: out6 = 2 x {part, first float, one past the last float of the flat gradient buffer} in completion order. */
int mi_mlpvae_dp_buckets(void* h, long long* out6);
int mi_mlpvae_train_step_dp(void* h, void* comm, void* stream, const void* src, const void* tgt, int frames_u8, const int* idx, int B, float inv_batch, const float* eps, float alpha, float beta1, float beta2, float epsilon, float* metrics3, float metric_weight);
int mi_mlpvae_encode(void* h, void* stream, const void* src, int frames_u8, const int* idx, int B, float* mean_out);
int mi_mlpvae_decode(void* h, void* stream, const float* z, int B, float* recon_out);
int mi_mlpvae_reconstruct(void* h, void* stream, const void* src, int frames_u8, const int* idx, int B, const float* eps, int sample, float* recon_out);

/* one environment step of the rollout loop in one call — vae_common.py:45-61 (encode_state) + ppo.py:231-251 (predict): raw uint8 frame [IH,IW,3] and
 * measurements -> out [num_actions + 1 + z_dim] = action | value | z; noise [num_actions] for sampling or NULL with greedy.  Exact fp32.
 * Every buffer may be HBM or pinned (device-mapped) host memory: with pinned buffers the step needs no copy in either direction. */
int mi_rollout_step(void* vae_h, void* ppo_h, void* stream, const unsigned char* frame_u8, const float* measurements, int n_meas, const float* noise, int greedy, float* out);

/* ---- collectives of the data-parallel path (no reference counterpart: the reference is single-process) ----
 * RCCL over xGMI, one communicator per process = per GPU; librccl.so.1 is bound at mi_comm_init (a single-GPU process never loads it).
 * Rendezvous: rank 0 calls mi_comm_unique_id, the mi_comm_id_bytes() = 128 bytes travel to the other ranks by any out-of-band channel, every
 * rank calls mi_comm_init (collective).  All calls below enqueue on HIP streams and never synchronise the host.
 *   mi_allreduce_sum_f32        buf <- sum over ranks, in stream order on `stream` (gradient buffer / metric accumulators)
 *   mi_allreduce_sum_f32_async  the same on the communicator's own stream, after what `stream` holds so far: a gradient bucket's all-reduce
 *                               runs under the next part of the backward pass; mi_comm_wait(comm, stream) joins before the optimiser step
 *   mi_broadcast                rank `root`'s bytes to every rank (the initial parameter replica, vae/models.py / ppo.py init_session)
 *   mi_comm_probe               binds librccl (dlopen + entry points) and nothing else: called on EVERY rank before the collective mi_comm_init so that
 *                               a rank that cannot load RCCL is agreed on while everybody can still fall back together
 *   mi_comm_set_algo            gradient-bucket schedule of this communicator: 0 = ncclAllReduce (default), 1 = reduce-scatter + all-gather (one hop per phase on the
 *                               fully connected xGMI mesh).  The SAME value on every rank: the host agrees on it first (mi355/dist.py reads
 *                               MI355_COMM_ALGO=rsag per process and takes the minimum over the ranks)
 *   mi_comm_allreduce_plan      what one all-reduce of n floats issues on a given rank under a schedule (pure function, no GPU: checked on the CPU for every rank):
 *                               out5 = {rsag?, floats per rank slice, this rank's slice offset, tail offset, tail floats} */
int mi_comm_probe(void);
/* 1: the bound RCCL offers the reduce-scatter + all-gather schedule (both entry points); the host takes the minimum over the ranks before anybody calls mi_comm_set_algo(1) */
int mi_comm_has_rsag(void);
/* A communicator that RECORDS what it would issue instead of communicating (no RCCL, no GPU; test infrastructure of the data-parallel step): log = HOST memory, 4 long long per
 * entry {op: 1 all-reduce 2 reduce-scatter 3 all-gather 4 broadcast 5 wait, floats (bytes for 4; buckets joined for 5), issued by the _async form?, buffer address}; the data is
 * left alone (the sum over one rank).  mi_comm_recorded: entries issued so far, -1 for a real communicator.  mi_comm_set_algo / mi_comm_destroy work on it as usual. */
int mi_comm_init_recording(void** comm_out, int rank, int world, long long* log, int log_capacity);
int mi_comm_recorded(void* comm);
/* round 6: the number of ranks the communicator spans as RCCL reports it (ncclCommCount); recording communicator: the world it was created for */
int mi_comm_ranks(void* comm);
int mi_comm_set_algo(void* comm, int algo);
int mi_comm_allreduce_plan(int algo, int world, int rank, long long n, long long* out5);
int mi_comm_id_bytes(void);
int mi_comm_unique_id(unsigned char* id_out);
int mi_comm_init(void** comm_out, int rank, int world, const unsigned char* id);
int mi_comm_destroy(void* comm);
int mi_allreduce_sum_f32(void* comm, void* stream, float* buf, long long n);
int mi_allreduce_sum_f32_async(void* comm, void* stream, float* buf, long long n);
int mi_comm_wait(void* comm, void* stream);
int mi_broadcast(void* comm, void* stream, void* buf, long long bytes, int root);

/* per-op timing with HIP events recorded on the launch stream (bench.py's live roofline numbers) */
int mi_vae_op_count(void);
const char* mi_vae_op_name(int op);
int mi_vae_timing_begin(void* h, int mode, int op_filter, int max_records);
int mi_vae_timing_collect(void* h, float* ms_sum, int* count, int n_ops);

/* ---- PPO engine: PolicyGraph x2 + losses + Adam (ppo.py:16-66,112-147) ---- */
int mi_ppo_desc_size(void);
int mi_ppo_tensor_count(void);
long long mi_ppo_param_floats(const MiPpoDesc* d);
int mi_ppo_param_layout(const MiPpoDesc* d, long long* offsets, long long* sizes, int n);
long long mi_ppo_workspace_bytes(const MiPpoDesc* d);
void* mi_ppo_create(const MiPpoDesc* d, float* params, float* params_old, float* grads, float* adam_m, float* adam_v, void* workspace, long long workspace_bytes, const float* action_low, const float* action_high);
void mi_ppo_destroy(void* h);
void* mi_ppo_buffer(void* h, int which);
/* PPO.update_old_policy — ppo.py:275-276 */
int mi_ppo_update_old(void* h, void* stream);
/* PPO.predict — ppo.py:231-251.  mi_ppo_buffer(h, 0): losses of the last step: [policy, value, entropy, total, mean ratio, mean action_mean[A], std[A]] (the
 * scalars of ppo.py:150-163); mi_ppo_buffer(h, 1): action_mean [M,A] of the last predict / train step */
int mi_ppo_predict(void* h, void* stream, const float* states, int M, const float* noise, int greedy, float* action, float* value);
/* gradient half + optimiser half of PPO.train (= north_star "learn") — ppo.py:218-229 */
int mi_ppo_forward_backward(void* h, void* stream, const float* states, const float* actions, const float* returns, const float* advantage, int M, float inv_m, float grad_scale);
int mi_ppo_apply_adam(void* h, void* stream, float alpha, float beta1, float beta2, float epsilon);
/* PPO.train's device work in ONE call (single rank): fused forward / losses / backward / Adam, five launches — ppo.py:218-229; logp_old (optional):
 * log pi_old(a|s) of the samples from mi_ppo_logp_old, computed once per horizon batch (theta_old is constant between update_old_policy() calls) */
int mi_ppo_train_step(void* h, void* stream, const float* states, const float* actions, const float* returns, const float* advantage, const float* logp_old, int M, float inv_m, float grad_scale, float alpha, float beta1, float beta2, float epsilon);
/* the same step with the minibatch gather fused in — train.py:199-204 (`states[mb_idx]`, ...): the five operands are the horizon-batch tables (n_rows rows,
 * device resident for the whole update) and row_idx [M] (int32, device) names this minibatch's rows */
int mi_ppo_train_step_idx(void* h, void* stream, const float* states, const float* actions, const float* returns, const float* advantage, const float* logp_old, const int* row_idx, int n_rows, int M, float inv_m, float grad_scale, float alpha, float beta1, float beta2, float epsilon);
/* round 6 (ABI 7): one DATA-PARALLEL SGD step of PPO.train (ppo.py:218-229 under the minibatch loop of train.py:193-207) in ONE call: the fused chain of
 * mi_ppo_train_step[_idx] with the gradients of this rank's M rows (inv_m = 1 / M_global, grad_scale = M / M_global) left in the flat buffer, ONE all-reduce of that
 * buffer through `comm` in stream order, tf.train.AdamOptimizer.  row_idx != NULL: the operands are this rank's horizon-batch tables (n_rows rows) and the gather of
 * train.py:199-204 stays inside the kernels; row_idx == NULL: contiguous minibatch tensors. */
int mi_ppo_train_step_dp(void* h, void* comm, void* stream, const float* states, const float* actions, const float* returns, const float* advantage, const float* logp_old, const int* row_idx, int n_rows, int M, float inv_m, float grad_scale, float alpha, float beta1, float beta2, float epsilon);
/* 1: this engine's shape (1 <= num_actions <= 8, h2 <= 320 and a multiple of 4, padded input width <= 96) is inside the range of the fused kernels and they are
 * switched on, i.e. mi_ppo_train_step_idx / mi_ppo_logp_old will run; 0: only the per-layer path exists for it (mi_ppo_train_step and mi_ppo_forward_backward fall
 * back by themselves; gather the minibatch on the host side instead of calling the _idx form).  Row indices are clamped into [0, n_rows) by the kernels. */
int mi_ppo_fused_shape_ok(void* h);
int mi_ppo_logp_old(void* h, void* stream, const float* states, const float* actions, int M, float* out);

#ifdef __cplusplus
}
#endif
#endif /* MI355_CARLA_H */
