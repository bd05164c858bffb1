/* deephisto_hip.h -- C ABI of libdeephisto_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the deephisto whole-slide patch hot path
 * (patch_samplers.full_samplers.FullImageDenseSampler ->
 *  examples.predict_full_patched.{batch_predictor, ImagePredictorPatched} ->
 *  models.patch_cls_simple.model.get_model).  The reference is pure Python and
 * has no FFI of its own; each entry point below names
 * the reference lines whose work it replaces, and INTEGRATION.md shows the
 * ctypes stub a reference maintainer would add at that line.
 *
 * Conventions
 *   - every function returns 0 on success or a negative DH_E* code; the text of
 *     the last failure on the calling thread is dh_last_error();
 *   - "dev" pointers are device (HBM) addresses owned by the caller, "host"
 *     pointers are ordinary host memory; nothing is retained past the call
 *     unless stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); device
 *     work is enqueued, never synchronised, unless stated;
 *   - no torch types, no C++ types: plain pointers and sizes only.
 */
#ifndef DEEPHISTO_HIP_H
#define DEEPHISTO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DH_OK 0
#define DH_EINVAL (-22)  /* bad argument (shape, alignment, null pointer) */
#define DH_ENOMEM (-12)  /* device or host allocation failed */
#define DH_EHIP (-5)     /* a HIP runtime call or kernel launch failed */
#define DH_ENOSYS (-38)  /* entry point not available in this build */

/* layouts / dtypes of gathered tiles */
#define DH_LAYOUT_NHWC 0 /* [n, P, P, 3]  (FullImageDenseSampler.generator_torch) */
#define DH_LAYOUT_NCHW 1 /* [n, 3, P, P]  (batch_predictor's model input)          */
#define DH_DTYPE_F32 0
#define DH_DTYPE_BF16 1

int dh_abi_version(void);
const char* dh_last_error(void);

/* ---- a1: tile-origin grid (host, integer) -------------------------------
 * Replaces FullImageDenseSampler._create_batched_coords,
 * patch_samplers/full_samplers.py:374-404.  Order: interior grid (y-major) over
 * range(0,h-P,S) x range(0,w-P,S), last column, last row, corner; then the list
 * is padded with copies of the corner up to a multiple of `batch`.
 * dh_tile_grid_count: *n_unique = origins before padding, *n_padded = after.
 * dh_tile_grid: writes n_padded (y,x) int32 pairs to host memory out_yx. */
int dh_tile_grid_count(int64_t h, int64_t w, int32_t patch, int32_t stride, int32_t batch,
                       int64_t* n_unique, int64_t* n_padded);
int dh_tile_grid(int64_t h, int64_t w, int32_t patch, int32_t stride, int32_t batch,
                 int32_t* out_yx_host, int64_t capacity_pairs);

/* ---- synthetic slide (device) -------------------------------------------
 * Fills slide_dev[h][w][3] (uint8, HWC, row pitch w*3) with the closed-form
 * benchmark slide (formula frozen in oracle/synth.py / DESIGN.md).  The reference
 * reads real slides via psimage (full_samplers.py:328-330); this is bench input. */
int dh_synth_slide(uint8_t* slide_dev, int64_t h, int64_t w, uint32_t seed, void* stream);

/* ---- a2/a4/a5: tile gather + /255 + layout --------------------------------
 * Replaces _generate_batch_memory (full_samplers.py:353-369) + the feature
 * build of generator_torch (full_samplers.py:441-443; NHWC f32) or of
 * batch_predictor (examples/predict_full_patched.py:67-71; NCHW f32).
 * slide_dev: uint8[h][w][3]; yx_dev: int32[n][2] (y,x) origins on device;
 * out_dev: n*P*P*3 elements of `dtype` in `layout`.  Values are exactly
 * float32(k)/255 (bit-exact with NumPy), rounded to nearest-even for bf16.
 * Origins must satisfy 0 <= y <= h-P, 0 <= x <= w-P (checked on host only when
 * yx_host_check != NULL, which must then hold the same n pairs). */
int dh_tile_gather(const uint8_t* slide_dev, int64_t h, int64_t w, const int32_t* yx_dev,
                   const int32_t* yx_host_check, int64_t n, int32_t patch, int32_t layout,
                   int32_t dtype, void* out_dev, void* stream);

/* Same gather with the batch-level augmentation of the training pipeline fused in
 * (models/patch_cls_simple/train.py:71-81: permute to NCHW, then RandomHorizontalFlip /
 * RandomVerticalFlip applied to the whole batch, i.e. one coin per batch): flip_h mirrors
 * columns, flip_v mirrors rows of every tile.  Feeds the a9 output contract
 * (patch_samplers/region_samplers.py:616-621, 729-738).  Origins may lie partly or wholly outside the
 * slide (the region samplers' origin bounds, region_samplers.py:112-118, allow a patch to hang over the
 * border): pixels outside [0,h) x [0,w) are written as 0, nothing outside the slide is read. */
int dh_tile_gather_aug(const uint8_t* slide_dev, int64_t h, int64_t w, const int32_t* yx_dev,
                       int64_t n, int32_t patch, int32_t layout, int32_t dtype, int32_t flip_h,
                       int32_t flip_v, void* out_dev, void* stream);

/* FullImageRndSampler.generator_torch (full_samplers.py:277-290) stacks the uint8 patches into a
 * float tensor WITHOUT dividing by 255: float32[n][P][P][3] with values 0..255 (0 outside the slide). */
int dh_tile_gather_raw(const uint8_t* slide_dev, int64_t h, int64_t w, const int32_t* yx_dev,
                       int64_t n, int32_t patch, float* out_dev, void* stream);

/* coords tensor of generator_torch (full_samplers.py:444-451): float32[n][2]. */
int dh_tile_coords_f32(const int32_t* yx_dev, int64_t n, float* out_dev, void* stream);

/* ---- a8: logit accumulation + argmax ------------------------------------
 * Replaces the per-patch `prediction[y//d:(y+P)//d, x//d:(x+P)//d, :] += logits[i]`
 * loop and the final argmax of ImagePredictorPatched.process
 * (examples/predict_full_patched.py:41-62).  Tiles are applied in list order
 * (duplicates included) with one float32 add each, so the canvas is bit-exact
 * with NumPy for identical logits.  yx_host: int32[n][2] host copy of origins;
 * logits_dev: float32[n][n_cls]; canvas_dev: float32[dh][dw][n_cls] (accumulated
 * into -- zero it first for a fresh prediction), dh = h/d, dw = w/d (floor).
 * map_dev (optional, may be NULL): int64[dh][dw] = first index of the maximum
 * over classes of the updated canvas (NumPy argmax tie/NaN rule).
 * The per-bin tile lists built from yx_host are cached per thread and reused while the same
 * origins are passed again (whole-slide prediction repeats one grid): such calls are fully
 * asynchronous; a call with new origins synchronises the stream once. */
int dh_accumulate_logits(const float* logits_dev, const int32_t* yx_host, int64_t n,
                         int32_t patch, int32_t downscale, int32_t n_cls, int64_t h, int64_t w,
                         float* canvas_dev, int64_t* map_dev, void* stream);
int dh_argmax_map(const float* canvas_dev, int64_t n_cells, int32_t n_cls, int64_t* map_dev,
                  void* stream);

/* ---- e1: the exchange step of the tile-sharded prediction -------------------------------
 * The reference predicts every tile on one device and sums the logits into the canvas in list order
 * (examples/predict_full_patched.py:40-63).  Sharded over ranks each rank predicts a
 * contiguous range of that list; this call is the ONE exchange: every rank contributes
 * float32[n_per_rank][n_cls] (ranges padded to a common length by the caller) and receives
 * float32[world][n_per_rank][n_cls] in rank order, after which each rank runs dh_accumulate_logits on the
 * whole list.  comm: the caller's RCCL communicator (ncclComm_t), created by the caller with the RCCL of
 * its process.  The ncclAllGather that runs must belong to the SAME library instance that made `comm`: this
 * library never links against or loads an RCCL of its own.  dh_set_rccl(handle) hands over the dlopen handle of
 * the host's RCCL (exact; NULL = back to automatic); without it the first exchange takes DH_RCCL_LIB=<file>
 * (must already be loaded), else a global ncclAllGather symbol, else an already-loaded librccl.so.1 / librccl.so;
 * nothing loaded => DH_EINVAL.  Asynchronous on `stream`.  The Python shims exchange through torch.distributed
 * instead (examples/predict_full_patched.py: exchange_logits). */
int dh_set_rccl(void* dl_handle);
int dh_allgather_logits(void* comm, const float* send_dev, float* recv_dev, int64_t n_per_rank,
                        int32_t n_cls, void* stream);

/* ---- visualisation of the class map (examples/predict_full_patched.py:81-113) ----
 * dh_colorize_map: `colored[pred == anno.id] = anno.color` for every class (:93-95):
 *   rgb[i] = lut[map[i]] when 0 <= map[i] < n_cls, else (0,0,0); lut = uint8[n_cls][3] on the device.
 * dh_overlay_blend: `(img * alpha + colored * (1 - alpha)).astype(np.uint8)` (:109-110) in float64,
 *   truncating like NumPy's cast; n_bytes = h*w*3. */
int dh_colorize_map(const int64_t* map_dev, int64_t n_cells, const uint8_t* lut_dev, int32_t n_cls,
                    uint8_t* rgb_dev, void* stream);
int dh_overlay_blend(const uint8_t* img_dev, const uint8_t* colored_dev, int64_t n_bytes, double alpha,
                     uint8_t* out_dev, void* stream);

/* ---- a6: ResNet-18 patch classifier forward ---------------------------------
 * Replaces `model(features)` for the network built by get_model
 * (models/patch_cls_simple/model.py:5-11: torchvision resnet18 + fc[n_cls,512])
 * in eval mode, as called from batch_predictor (predict_full_patched.py:77).
 * A handle owns device copies of the parameters (packed for MFMA) and the
 * activation workspace; one handle per thread/stream.
 * compute dtype: DH_DTYPE_F32 (f32 MFMA, logits within 1e-4 of the CPU path) or
 * DH_DTYPE_BF16 (bf16 MFMA, f32 accumulate). */
typedef struct dh_resnet18 dh_resnet18;
int dh_resnet18_create(dh_resnet18** out, int32_t n_classes, int32_t compute_dtype);
void dh_resnet18_destroy(dh_resnet18* net);
/* Set one parameter/buffer by its torchvision state_dict name ("conv1.weight",
 * "layer2.0.downsample.1.running_var", "fc.bias", ...).  data_host: float32,
 * n_elem must match.  "num_batches_tracked" entries are accepted and ignored. */
int dh_resnet18_set_param(dh_resnet18* net, const char* name, const float* data_host,
                          int64_t n_elem);
/* Call after all parameters are set (or changed): folds BN running stats into
 * per-channel scale/shift and packs conv weights into MFMA fragment order. */
int dh_resnet18_finalize(dh_resnet18* net, void* stream);
/* x_dev: float32[n][3][P][P] NCHW in [0,1] (what batch_predictor feeds the model);
 * logits_dev: float32[n][n_classes].  P must be a multiple of 32. */
int dh_resnet18_forward(dh_resnet18* net, const float* x_dev, int64_t n, int32_t patch,
                        float* logits_dev, void* stream);
/* Fused a2+a5+a6: gathers the tiles straight from the uint8 slide inside the
 * stem kernel (no float copy of the pixels is ever written to HBM). */
int dh_resnet18_forward_tiles(dh_resnet18* net, const uint8_t* slide_dev, int64_t h, int64_t w,
                              const int32_t* yx_dev, int64_t n, int32_t patch,
                              float* logits_dev, void* stream);

/* ---- a7: training step (float32) -------------------------------------------------------
 * Replaces, for the same network, models/patch_cls_simple/train.py:168-172:
 *   outputs = model(inputs); loss = criterion(outputs, labels); loss.backward(); optimizer.step()
 * with CrossEntropyLoss (mean) at train.py:117 and Adam(lr) at train.py:118 (PyTorch
 * defaults: betas (0.9, 0.999), eps 1e-8, no weight decay; BatchNorm momentum 0.1, eps 1e-5,
 * batch statistics, running statistics updated with the unbiased variance).
 * Training state (master parameters, gradients, Adam moments, running statistics, saved
 * activations; all in HBM) is created by the first dh_resnet18_forward_train / _train_begin
 * for a batch shape from the parameters set with dh_resnet18_set_param, and folded back into
 * the handle (for eval-mode forwards) by dh_resnet18_train_end.
 *   forward_train : x_dev float32[n][3][P][P] (must stay alive until backward) -> logits
 *   backward      : dlogits_dev float32[n][n_classes] -> every parameter gradient
 *   dh_ce_loss    : mean cross entropy of logits vs int64 labels -> *loss_dev, and (optional)
 *                   dlogits_dev = (softmax - onehot)/n
 *   adam_step     : one Adam update of all parameters from the current gradients (step >= 1),
 *                   then re-packs the MFMA weight copies
 *   train_tensor  : copy one tensor by state_dict name between the library and caller memory
 *                   (host or device): kind 0 parameter, 1 gradient, 2 running statistic;
 *                   to_lib != 0 writes into the library (call train_repack after parameters). */
int dh_resnet18_train_begin(dh_resnet18* net, int64_t n, int32_t patch, void* stream);
int dh_resnet18_train_end(dh_resnet18* net);
int dh_resnet18_forward_train(dh_resnet18* net, const float* x_dev, int64_t n, int32_t patch,
                              float* logits_dev, void* stream);
int dh_resnet18_backward(dh_resnet18* net, const float* dlogits_dev, void* stream);
int dh_ce_loss(const float* logits_dev, const int64_t* labels_dev, int64_t n, int32_t n_cls,
               float* loss_dev, float* dlogits_dev, void* stream);
int dh_resnet18_adam_step(dh_resnet18* net, float lr, float beta1, float beta2, float eps,
                          int64_t step, void* stream);
/* backward + Adam of a single-rank step in one call: each residual block's update and repack follow its weight
 * gradients on the engine's side stream.  Bit-identical to dh_resnet18_backward + dh_resnet18_adam_step; refused
 * while gradient buckets are armed. */
int dh_resnet18_backward_adam(dh_resnet18* net, const float* dlogits_dev, float lr, float beta1, float beta2,
                              float eps, int64_t step, void* stream);
int dh_resnet18_train_tensor(dh_resnet18* net, const char* name, int32_t kind, void* ptr,
                             int64_t n_elem, int32_t to_lib, void* stream);
int dh_resnet18_train_repack(dh_resnet18* net, void* stream);
/* Device pointer + element count of a whole arena (kind as above): data-parallel training
 * all-reduces the gradient arena in place (RCCL) between backward and adam_step. */
int dh_resnet18_train_flat(dh_resnet18* net, int32_t kind, void** ptr_out, int64_t* n_out);
/* Gradient buckets for the overlapped all-reduce of data-parallel training: the gradient arena is cut
 * into buckets of about bucket_bytes (0 = one) in the order the backward pass completes them (from the END of the arena:
 * fc, layer4 ... layer1, stem); `cb(bucket, offset, count, user)` is called on the calling thread from inside
 * dh_resnet18_backward as soon as the kernels that complete a bucket are enqueued. */
int dh_resnet18_set_buckets(dh_resnet18* net, int64_t bucket_bytes,
                            void (*cb)(int32_t bucket, int64_t offset, int64_t count, void* user), void* user,
                            int32_t* n_buckets_out);
int dh_resnet18_bucket(dh_resnet18* net, int32_t i, int64_t* offset, int64_t* count);

/* ---- configs[4]: ResNet-50 (and ResNet-18) training / evaluation in bf16 ---------------------------
 * The same step (models/patch_cls_simple/train.py:166-172; CrossEntropyLoss(mean) :117, Adam :118) for the network the
 * factory of models/patch_cls_simple/model.py:5-11 returns when it is given a ResNet-50 backbone (BASELINE.json
 * configs[4]: torchvision resnet50 v1.5 + fc[n_cls, 2048]; "resnet18" selects the reference's own backbone).
 * Activations and their gradients are bf16 (bf16 MFMA, f32 accumulation: forward, dgrad AND wgrad); master weights,
 * gradients, Adam moments and BN statistics are float32.  Parameters are set / read by torchvision state_dict name
 * with dh_train2_tensor (kind 0 parameter, 1 gradient, 2 running statistic; host or device pointers).
 *   forward(training != 0): batch-statistic BN, running statistics updated; x_dev float32[n][3][P][P] must stay alive
 *                           until backward.  forward(training == 0): BN from the running statistics (evaluation).
 *   backward              : dlogits float32[n][n_classes] -> every gradient (no float atomics: reproducible)
 *   adam_step             : step <= 0 uses the library's own step count
 *   set_buckets           : cut the gradient arena (laid out in backward-completion order, fc first) into buckets of about
 *                           bucket_bytes; `cb` is called on the calling thread inside backward as soon as the kernels that
 *                           complete a bucket are enqueued -- the hook for the bucketed, overlapped RCCL all-reduce of
 *                           data-parallel training.  dh_train2_bucket returns a bucket's element range. */
typedef struct dh_train2 dh_train2;
typedef void (*dh_bucket_cb)(int32_t bucket, int64_t offset, int64_t count, void* user);
int dh_train2_create(dh_train2** out, const char* arch, int32_t n_classes);
void dh_train2_destroy(dh_train2* net);
int dh_train2_tensor(dh_train2* net, const char* name, int32_t kind, void* ptr, int64_t n_elem, int32_t to_lib,
                     void* stream);
int dh_train2_flat(dh_train2* net, int32_t kind, void** ptr_out, int64_t* n_out);
int dh_train2_set_buckets(dh_train2* net, int64_t bucket_bytes, dh_bucket_cb cb, void* user, int32_t* n_buckets_out);
int dh_train2_bucket(dh_train2* net, int32_t i, int64_t* offset, int64_t* count);
/* optional bf16 wire format of the gradient exchange (either engine's arena; n % 4 == 0): float32 -> bf16 (round to nearest even)
 * before a bucket's all-reduce, bf16 sums -> float32 * scale (1 / world) after it.  Halves the bytes on xGMI. */
int dh_grad_pack_bf16(const float* src_dev, uint16_t* dst_dev, int64_t n, void* stream);
int dh_grad_unpack_bf16(const uint16_t* src_dev, float* dst_dev, int64_t n, float scale, void* stream);
int dh_train2_forward(dh_train2* net, const float* x_dev, int64_t n, int32_t patch, float* logits_dev,
                      int32_t training, void* stream);
int dh_train2_backward(dh_train2* net, const float* dlogits_dev, void* stream);
int dh_train2_adam_step(dh_train2* net, float lr, float beta1, float beta2, float eps, int64_t step, void* stream);
/* backward + Adam of a single-rank step in one call: each residual block's update and bf16 repack follow its weight gradients on the
 * engine's side stream.  Bit-identical to dh_train2_backward + dh_train2_adam_step; refused while gradient buckets are armed. */
int dh_train2_backward_adam(dh_train2* net, const float* dlogits_dev, float lr, float beta1, float beta2, float eps, int64_t step,
                            void* stream);
/* ---- measurement -----------------------------------------------------------------
 * Times the dominant kernel (3x3 stride-1 conv, ~85 % of the model FLOPs) with HIP
 * events recorded on the launch stream around every `sample_every`-th launch (at most
 * max_samples).  dh_profile_stop waits for the sampled launches and returns the summed
 * kernel time, the summed algorithmic FLOPs (2*pixels*cout*9*cin) and the sample count. */
int dh_profile_start(int32_t sample_every, int32_t max_samples);
int dh_profile_stop(double* total_ms, double* total_flops, int64_t* n_samples);

#ifdef __cplusplus
}
#endif
#endif /* DEEPHISTO_HIP_H */
