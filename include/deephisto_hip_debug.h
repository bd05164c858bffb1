/* deephisto_hip_debug.h -- TEST HOOKS of libdeephisto_hip.so.  NOT part of the drop-in boundary and NOT versioned.
 *
 * include/deephisto_hip.h is the boundary SURVEY.md section 8(b) describes (ABI version dh_abi_version()); the entry points below
 * exist so that tests/ can run ONE kernel of the engines on caller-provided data and localise an error to it (per-kernel parity
 * against float64 autograd, tests/test_gpu_train_kernels.py; single conv layers, tests/test_gpu_resnet.py; the cycle-stamped
 * diagnostic variants, tools/conv_stamps.py).  They are exported by the same shared object and bound by deephisto_amd/_lib.py,
 * but their signatures follow the kernels' internals and may change in any build without a change of dh_abi_version();
 * the product path (the deephisto_amd package, bench.py's timed region) never calls them.
 */
#ifndef DEEPHISTO_HIP_DEBUG_H
#define DEEPHISTO_HIP_DEBUG_H

#include "deephisto_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* test hook: float32 copy of conv `conv_name`'s raw output (what = 0) or BN/ReLU output (what = 1) of the last forward */
int dh_train2_debug_act(dh_train2* net, const char* conv_name, int32_t what, float* out_dev, int64_t n_elem, void* stream);

/* test hooks of the engine's GEMM-shaped kernels on caller data (bf16 bits as uint16; synchronise; `repeat` launches for timing):
 * gemm1x1: out[M][N] = A[rows][K] . W[N][K]^T (+ res), stride 2 = row gather (b, 2 oy, 2 ox) from [B][Hi][Wi][K];
 * wgrad:   float32 dW[cout][cin][ks][ks] from x [B][Hi][Wi][cin] and dz [B][Ho][Wo][cout] (ks 1 or 3). */
/* both bf16 operators of float32 w[cout][cin][3][3]: the engine's re-pack kernel (..._tile) and the element-wise packer (..._elem) */
int dh_debug_pack_bf16(const float* w_dev, int32_t cout, int32_t cin, uint16_t* wf_tile, uint16_t* wd_tile, uint16_t* wf_elem,
                       uint16_t* wd_elem, void* stream);
int dh_debug_gemm1x1_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const uint16_t* res_dev, uint16_t* out_dev, int64_t M,
                          int32_t N, int32_t K, int32_t stride, int32_t Ho, int32_t Wo, int32_t Hi, int32_t Wi, int32_t repeat,
                          void* stream);
/* dh_debug_gemm1x1_fused_bf16: the 1x1 GEMM with its fused epilogues -- + res, ReLU mask of another tensor (output zeroed where
 * mask_dev <= 0), batch statistics of the bf16-rounded output (mean / invstd as the BN finalize writes them; both or neither). */
int dh_debug_gemm1x1_fused_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const uint16_t* res_dev, const uint16_t* mask_dev,
                                uint16_t* out_dev, float* mean_dev, float* invstd_dev, int64_t M, int32_t N, int32_t K, int32_t stride,
                                int32_t Ho, int32_t Wo, int32_t Hi, int32_t Wi, int32_t repeat, void* stream);
/* dh_debug_gemm1x1_bwdsums_bf16: the dgrad GEMM whose output is the dY of a batch norm, with that BN's backward sums fused into the
 * epilogue: sums_dev[0..N) = sum g, sums_dev[N..2N) = sum g xhat (relu_mode 0: g = out as stored; 2: the BN's ReLU recomputed from z). */
int dh_debug_gemm1x1_bwdsums_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const uint16_t* res_dev, const uint16_t* mask_dev,
                                  uint16_t* out_dev, const uint16_t* z_dev, const float* mean_dev, const float* invstd_dev,
                                  const float* scale_dev, const float* shift_dev, int32_t relu_mode, float* sums_dev, int64_t M, int32_t N,
                                  int32_t K, void* stream);
/* dh_debug_bn2_bf16: the bf16 engine's batch-norm kernels on caller data ([rows][C] channels-last, bf16 bits): forward with batch
 * statistics (y, saved mean / invstd), and, when dy_dev is given, backward (dz, dgamma, dbeta; relu_mode 0 none, 1 mask from y,
 * 2 mask recomputed from z; g_out_dev: the masked gradient, may be null). */
int dh_debug_bn2_bf16(const uint16_t* z_dev, const uint16_t* res_dev, const float* gamma_dev, const float* beta_dev, int32_t relu,
                      uint16_t* y_dev, float* mean_dev, float* invstd_dev, const uint16_t* dy_dev, int32_t relu_mode, uint16_t* dz_dev,
                      uint16_t* g_out_dev, float* dgamma_dev, float* dbeta_dev, int64_t rows, int32_t C, void* stream);
/* the stem's fused tail (bf16 engine): pooled = maxpool3x3/2(relu(bn(z))) straight from z with batch statistics (+ positions, mean, invstd);
 * with dpool_dev: dz, dgamma, dbeta with the maxpool's gradient gathered inside the BN backward passes. */
int dh_debug_bn2_pool_bf16(const uint16_t* z_dev, const float* gamma_dev, const float* beta_dev, uint16_t* pooled_dev, uint8_t* idx_dev,
                           float* mean_dev, float* invstd_dev, const uint16_t* dpool_dev, uint16_t* dz_dev, float* dgamma_dev,
                           float* dbeta_dev, int32_t B, int32_t Hi, int32_t Wi, int32_t C, void* stream);
/* The remaining HBM-bound kernels of the bf16 engine on caller data: max-pool 3x3/2 forward (+ backward when dy_dev is given),
 * the strided add of the downsample branch's gradient, and the average-pool + fc backward. */
int dh_debug_maxpool2_bf16(const uint16_t* x_dev, uint16_t* y_dev, const uint16_t* dy_dev, uint16_t* dx_dev, int32_t B, int32_t Hi,
                           int32_t Wi, int32_t C, void* stream);
int dh_debug_upsample2_add_bf16(const uint16_t* t_dev, uint16_t* dx_dev, int32_t B, int32_t Ho, int32_t Wo, int32_t Hi, int32_t Wi,
                                int32_t C, void* stream);
int dh_debug_avgpool_fc_dgrad2(const float* dlogits_dev, const float* w_dev, uint16_t* dx_dev, int32_t B, int32_t HW, int32_t C,
                               int32_t n_cls, void* stream);
int dh_debug_stem_wgrad_bf16(const uint16_t* dz_dev, const float* x_nchw_dev, float* dw_dev, int32_t B, int32_t P, void* stream);
int dh_debug_wgrad_bf16(const uint16_t* dz_dev, const uint16_t* x_dev, float* dw_dev, int32_t B, int32_t Hi, int32_t Wi,
                        int32_t cin, int32_t cout, int32_t ks, int32_t stride, int32_t repeat, void* stream);

/* ---- debug / test hooks (not part of the drop-in boundary) ---------------------
 * dh_debug_conv_bn_act: one conv (ks in {1,3}, pad ks/2) + per-channel scale/shift
 * (+ residual) (+ ReLU) on NHWC data of `dtype`; weights are float32
 * [cout][cin][ks][ks] on the host.  Synchronises the stream.
 * dh_debug_stem_out: float32 NHWC copy of the stem activation (conv1+bn1+relu) left
 * in the workspace by the last forward of n tiles of size P. */
int dh_debug_conv_bn_act(const void* in_dev, const float* w_host, const float* scale_host,
                         const float* shift_host, const void* res_dev, void* out_dev, int32_t B,
                         int32_t H, int32_t W, int32_t cin, int32_t cout, int32_t ks, int32_t stride,
                         int32_t relu, int32_t dtype, void* stream);
int dh_debug_stem_out(dh_resnet18* net, int64_t n, int32_t patch, float* out_dev, void* stream);
/* dh_debug_stem_pool_bf16: the fused bf16 stem (conv1 + bn1 + relu + maxpool, models/patch_cls_simple/model.py:6 = the first four
 * modules of torchvision's resnet18) of n tiles read from the uint8 slide, as float32 NHWC [n][H2][W2][64]; synchronises. */
int dh_debug_stem_pool_bf16(dh_resnet18* net, const uint8_t* slide_dev, int64_t slide_h, int64_t slide_w, const int32_t* yx_dev,
                            int64_t n, int32_t patch, float* out_dev, void* stream);
/* Backward building blocks of the float32 training engine, one at a time, exactly as dh_resnet18_backward launches them
 * (all tensors float32 on the device, activations NHWC; each call allocates its own scratch and synchronises):
 *   wgrad       dW[cout][cin][ks][ks] from x [B][Hi][Wi][cin] and dz [B][Ho][Wo][cout]; mode 1 forces the per-tap kernel
 *   stem_wgrad  dW[64][3][7][7] from the NCHW image and dz [B][P/2][P/2][64]
 *   dgrad       dX from dz and the weights [cout][cin][ks][ks] (+ res): stride 2 = zero-upsampled gradient (3x3) /
 *               low-resolution product scattered back (1x1)
 *   bn          training-mode BN forward (+ res, ReLU) and, when dy is given, backward (dz, masked gradient g, dgamma, dbeta);
 *               stats_out = [mean | invstd | running_mean | running_var] after one update from (0, 1)
 *   maxpool     3x3/2 forward and (dy given) backward through the recorded first-maximum positions */
int dh_debug_wgrad_f32(const float* dz_dev, const float* x_dev, float* dw_dev, int32_t B, int32_t Hi, int32_t Wi,
                       int32_t cin, int32_t cout, int32_t ks, int32_t stride, int32_t mode, void* stream);
int dh_debug_stem_wgrad_f32(const float* dz_dev, const float* x_nchw_dev, float* dw_dev, int32_t B, int32_t P, void* stream);
int dh_debug_dgrad_f32(const float* dz_dev, const float* w_dev, const float* res_dev, float* dx_dev, int32_t B, int32_t Hi,
                       int32_t Wi, int32_t cin, int32_t cout, int32_t ks, int32_t stride, void* stream);
/* both packed operators of w[cout][cin][3][3]: the training step's sub-tile kernel (..._tile) and the element-wise kernel (..._elem) */
int dh_debug_pack_f32(const float* w_dev, int32_t cout, int32_t cin, float* wf_tile, float* wd_tile, float* wf_elem, float* wd_elem,
                      void* stream);
int dh_debug_bn_f32(const float* z_dev, const float* gamma_dev, const float* beta_dev, const float* res_dev, int32_t relu,
                    float* y_dev, const float* dy_dev, float* dz_dev, float* g_dev, float* dgamma_dev, float* dbeta_dev,
                    float* stats_out_dev, int64_t rows, int32_t C, void* stream);
int dh_debug_maxpool_f32(const float* x_dev, float* y_dev, const float* dy_dev, float* dx_dev, int32_t B, int32_t Hi, int32_t Wi,
                         int32_t C, void* stream);
/* the stem's fused tail (float32 engine): pooled = maxpool3x3/2(relu(bn(z))) straight from z with batch statistics (+ positions); with
 * dpool_dev: dz, dgamma, dbeta with the maxpool's gradient gathered inside the BN backward passes */
int dh_debug_bn_pool_f32(const float* z_dev, const float* gamma_dev, const float* beta_dev, float* pooled_dev, uint8_t* idx_dev,
                         const float* dpool_dev, float* dz_dev, float* dgamma_dev, float* dbeta_dev, int32_t B, int32_t Hi, int32_t Wi,
                         int32_t C, void* stream);
/* dh_debug_stamps: switch the 3x3-conv kernel to its cycle-stamped diagnostic variant and/or read
 * (and clear) its 8x8 table of summed phase cycles; out64_host may be NULL. */
int dh_debug_stamps(int32_t enable, unsigned long long* out64_host);
/* dh_debug_env_knobs: the table of every environment variable the library reads (csrc/env_knobs.h), one "NAME default lo hi read"
 * line per knob, NUL-terminated, into buf_host[cap]; INTEGRATION.md lists the same table (tests/test_abi.py compares them). */
int dh_debug_env_knobs(char* buf_host, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* DEEPHISTO_HIP_DEBUG_H */
