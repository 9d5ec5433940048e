// gemm_core.hpp — the MFMA tile-kernel family of the ConvVAE / PPO hot path (gfx950, wave64).
//
//   gemm_tile.hpp   gemm_kernel<A_CONV,...>    C[m,n] = sum_k im2col(X)[m,k] * W[k,n]    conv fwd, deconv dgrad, dense fwd/dgrad
//                   gemm_kernel<A_DECONV,...>  stride-2 transposed conv in gather form, one GEMM per output-parity class
//                                              (deconv fwd, conv dgrad)
//   gemm2_tile.hpp  gemm2_kernel               same contractions for the wide layers: LDS-DMA tiles, XOR-swizzled LDS, 128-B K stages
//   tapconv_tile.hpp tapconv_kernel            stride-2 conv / transposed conv as a stride-1 tap conv on raw-staged slot tiles
//   wgrad_tile.hpp  wgrad_kernel               dW[kc,n] += sum_m im2col(big)[m,kc] * small[m,n]   (conv/deconv/dense wgrad)
//
//   tapwgrad_tile.hpp tapwgrad_kernel          bf16 weight gradients of the wide stride-2 layers on raw-staged slot tiles
//
//   narrow_tile.hpp  gather_narrow_kernel      transposed conv into a 1..8-channel output (logits)
//                    narrow_wgrad_kernel       filter gradient of the layers with a 1..3-channel side (conv1, deconv4)
//                    narrow_conv_kernel        conv from a 1..3-channel tensor into 32 channels (conv1 fwd, deconv4 dgrad)
//
//   dectail_tile.hpp dectail_kernel            deconv4 forward + reconstruction loss + its input and filter gradients in one launch (round 3)
//
// Launchers (C ABI) live in conv_ops.hip.
#pragma once
#include "gemm_tile.hpp"
#include "gemm2_tile.hpp"
#include "tapconv_tile.hpp"
#include "wgrad_tile.hpp"
#include "dwg_tile.hpp"
#include "dwgs_tile.hpp"
#include "tapwgrad_tile.hpp"
#include "narrow_tile.hpp"
#include "dectail_tile.hpp"
