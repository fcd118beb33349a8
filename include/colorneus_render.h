/* colorneus_render.h -- C ABI of the MI355X-native Color-NeuS volume renderer (libcolorneus_hip.so).
 *
 * The reference (Colmar-zlicheng/Color-NeuS) is pure Python and has no FFI boundary of its own; its plug-in
 * interface for this path is the RENDERER registry class whose forward() is
 *     NeuS.forward(rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0)
 *                                                              (lib/models/renderers/NeuS.py:294-408)
 * with Color_NeuS.render_core (lib/models/renderers/Color_NeuS.py:24-138) underneath, plus
 *     NeuS.extract_geometry / extract_fields   (NeuS.py:14-40, 410-417)   -> cnr_sdf_grid / cnr_sdf_eval
 *     NeuS.extract_color                       (NeuS.py:44-64, 419-420)   -> cnr_vertex_color
 * The entry points below are what a ctypes / cffi binding of that class calls.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 unless stated otherwise; the caller owns all memory
 *     (outputs, context, scratch); the library allocates nothing.  No RESULT depends on state kept between calls; what the
 *     process does keep: the first-error latch behind cnr_last_error(), the optional timing records (cnr_timing_enable), debug
 *     switches read once from the environment (CNR_*), the per-(kernel, device) "dynamic LDS opt-in applied" flags, and an
 *     atomic launch counter whose parity only picks the order in which a layer launch walks its tiles (speed, never a value)
 *   - all work is enqueued on the given hipStream_t (passed as void*); no internal synchronisation
 *   - return value: 0 on success, negative on error; cnr_last_error() returns a message for the calling thread
 *   - parameters are passed as an array of device pointers in the canonical order reported by cnr_param_info()
 *     (names are the reference's state_dict names, e.g. "sdf_network.lin0.weight_v")
 */
#ifndef COLORNEUS_RENDER_H_
#define COLORNEUS_RENDER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CNR_ABI_VERSION 8

typedef struct cnr_config {
  int32_t type;              /* 0 = NeuS (NeuS.py:68), 1 = Color_NeuS (Color_NeuS.py:10) */
  int32_t n_samples;         /* N_SAMPLES      (NeuS.py:80) */
  int32_t n_importance;      /* N_IMPORTANCE   (NeuS.py:81) */
  int32_t up_sample_steps;   /* UP_SAMPLE_STEPS(NeuS.py:83) */
  /* SDFNetwork (fields.py:19-29) */
  int32_t sdf_d_hidden, sdf_n_layers, sdf_d_out, sdf_multires, sdf_skip_mask /* bit l set <=> l in SKIP_IN */, sdf_weight_norm;
  float sdf_scale;
  /* RenderingNetwork (fields.py:126-134): mode 0 = idr, 1 = no_view_dir, 2 = no_normal */
  int32_t col_mode, col_d_feature, col_d_hidden, col_n_layers, col_multires_view, col_weight_norm, col_squeeze_out;
  /* RelightNetwork (fields.py:296-303), ignored when type == 0 */
  int32_t rel_d_hidden, rel_n_layers, rel_y_in_layer, rel_multires_view, rel_include_grad, rel_inv_sigmoid;
} cnr_config;

typedef struct cnr_render_inputs {
  const float* rays_o;          /* [R][3] */
  const float* rays_d;          /* [R][3] */
  const float* near_;           /* [R]    */
  const float* far_;            /* [R]    */
  const float* t_rand;          /* [R] uniform [0,1) jitter draw (the reference's torch.rand([R,1]), NeuS.py:325) or NULL = no perturb */
  const float* z_vals_override; /* [R][M] or NULL: skip the sampler and render at these z (parity gate G2) */
  const float* background_rgb;  /* [3] or NULL */
  int64_t n_rays;               /* > 0 (an empty batch is the host layer's business: renderer.py renders it like the reference, as empty
                                   outputs with zero gradients; the entry points reject n_rays <= 0 with an error) */
  float cos_anneal_ratio;
  float prune_eps;              /* > 0: INFERENCE ONLY early-termination compaction -- the colour / relight networks run only on samples
                                   whose compositing weight is >= prune_eps (pixel error < prune_eps per skipped sample); per-sample
                                   colour outputs of skipped samples are zero and cnr_render_backward must not be called */
} cnr_render_inputs;

/* the reference's return dict (NeuS.py:387-408) plus the final z_vals; M = n_samples + n_importance */
typedef struct cnr_render_outputs {
  float* color_fine;     /* [R][3]    */
  float* s_val;          /* [R]       */
  float* cdf_fine;       /* [R][M]    */
  float* weight_sum;     /* [R]       */
  float* weight_max;     /* [R]       */
  float* gradients;      /* [R][M][3] or NULL (the library then keeps them in its context buffer; see delta_relight_ray_sum) */
  float* weights;        /* [R][M]    */
  float* gradient_error; /* [1]       */
  float* inside_sphere;  /* [R][M]    */
  float* depth;          /* [R]       */
  float* global_color;   /* [R][3]    (Color_NeuS only, else NULL) */
  float* delta_relight;  /* [R][M][3] (Color_NeuS only, else NULL); may also be NULL for Color_NeuS when only delta_relight_ray_sum is wanted */
  float* z_vals;         /* [R][M]    */
  float* eik_sums;       /* [2] or NULL: {sum relax*(|g|-1)^2, sum relax} of this call's rays -- lets a ray-sharded run rebuild the global eikonal ratio */
  /* optional per-sample network outputs, for callers that do their own alpha / compositing (the N_OUTSIDE > 0 background mixing of
     NeuS.py:262-268 / Color_NeuS.py:97-102 runs above the ABI): */
  float* sdf_samples;           /* [R][M]    or NULL: sdf at the section midpoints */
  float* color_samples;         /* [R][M][3] or NULL: the colour that is composited (relit colour for Color_NeuS) */
  float* global_color_samples;  /* [R][M][3] or NULL: colour-network output before relighting (Color_NeuS only) */
  /* "loss only" training outputs: compute_loss (NeuS_Trainer.py:129-171) consumes `gradients` only through
     gradient_error and `delta_relight` only through mean(delta_relight * mask) -- with this per-ray sum the two [R][M][3] dict tensors
     need not be materialised for the caller (pass gradients = delta_relight = NULL): */
  float* delta_relight_ray_sum; /* [R] or NULL: sum over the samples and rgb of delta_relight of each ray (Color_NeuS only) */
} cnr_render_outputs;

/* upstream gradients of the outputs; any member may be NULL (= zero) */
typedef struct cnr_render_out_grads {
  const float* color_fine; const float* s_val; const float* cdf_fine; const float* weight_sum; const float* weight_max;
  const float* gradients; const float* weights; const float* gradient_error; const float* depth;
  const float* global_color; const float* delta_relight;
  const float* sdf_samples; const float* color_samples; const float* global_color_samples;   /* gradients of the per-sample outputs, or NULL */
  const float* delta_relight_per_ray;   /* [R] or NULL: the gradient of delta_relight when it is constant along a ray and over rgb -- the
                                           relight term of the training loss, mean(delta_relight * mask)^2 (NeuS_Trainer.py:153), yields
                                           exactly that; lets the loss seed the backward pass without an [R][M][3] buffer.  Added to
                                           delta_relight when both are given. */
} cnr_render_out_grads;

typedef struct cnr_render_in_grads {
  float* const* d_params;   /* host array of device pointers, same order/shapes as the parameters; overwritten (not accumulated) */
  float* d_rays_o;          /* [R][3] or NULL */
  float* d_rays_d;          /* [R][3] or NULL */
  float* d_near;            /* [R] or NULL; with d_far.  Non-zero only when n_importance == 0 and no z override: z is then an affine   */
  float* d_far;             /* function of near / far (NeuS.py:311-313); with importance sampling z is built under no_grad (:343)     */
} cnr_render_in_grads;

/* optional per-launch timing (HIP events on the launch stream); used by bench.py for the roofline figures */
typedef struct cnr_kernel_timing {
  char name[32];
  int32_t kind;     /* 0 = layer GEMM, 1 = weight-gradient GEMM, 2 = other */
  int32_t nt;       /* tile variant */
  int64_t P;        /* points (rows) or rays */
  int32_t N, K, pairs;
  float ms;
  double bytes;     /* algorithmic HBM bytes of the launch (operands read once + outputs written once); 0 = not modelled */
} cnr_kernel_timing;
void cnr_timing_enable(int on);
int cnr_timing_collect(cnr_kernel_timing* out, int max_records);

/* ---- loss of the training step (replaces NeuS_Trainer.compute_loss, NeuS_Trainer.py:129-171, the consumer right after the path) ----
 * loss = lambda_fine * mean((color_fine - rgb_gt)^2 or |.|) + lambda_eikonal * gradient_error
 *      + lambda_mask * BCE(clip(weight_sum, 1e-3, 1 - 1e-3), mask) + lambda_relight * mean(delta_relight * mask)^2
 * Two phases so that a ray-sharded run can all-reduce the three sums in between:
 *   cnr_loss_sums : sums[0] = sum of squared (or absolute) colour errors, sums[1] = sum of the BCE terms, sums[2] = sum of
 *                   delta_relight (* mask); fixed-order reductions.
 *   cnr_loss_grads: d_color_fine = coef[0] * (c - gt) (or sign), d_weight_sum = coef[1] * dBCE/dws, d_delta_relight = coef[2] (* mask);
 *                   the caller folds lambda, 1/N and the upstream gradient into the three device-side coefficients. */
typedef struct cnr_loss_config {
  float lambda_fine, lambda_eikonal, lambda_mask, lambda_relight;
  int32_t rgb_l1;        /* 0: MSE (RGB_LOSS_TYPE "mse"), 1: L1 */
  int32_t include_mask;  /* relight term uses delta_relight * mask */
} cnr_loss_config;
size_t cnr_loss_scratch_bytes(int64_t n_rays);
int cnr_loss_sums(const cnr_loss_config* cfg, const float* color_fine, const float* weight_sum, const float* delta_relight /* or NULL */,
                  const float* rgb_gt, const float* mask /* [R] or NULL */, int64_t n_rays, int32_t n_samples, float* sums /* device [4] */,
                  void* scratch, size_t scratch_bytes, void* stream);
/* cnr_loss_sums with the relight term given as the per-ray sums cnr_render_outputs.delta_relight_ray_sum [R] instead of [R][M][3] */
int cnr_loss_sums_ray(const cnr_loss_config* cfg, const float* color_fine, const float* weight_sum, const float* delta_relight_ray_sum,
                      const float* rgb_gt, const float* mask, int64_t n_rays, int32_t n_samples, float* sums, void* scratch, size_t scratch_bytes,
                      void* stream);
int cnr_loss_grads(const cnr_loss_config* cfg, const float* color_fine, const float* weight_sum, const float* rgb_gt, const float* mask,
                   int64_t n_rays, int32_t n_samples, const float* coef /* device [4] */, float* d_color_fine, float* d_weight_sum,
                   float* d_delta_relight /* or NULL */, void* stream);
/* The scalar arithmetic around the two phases on the device, one launch each (single-process runs; a ray-sharded run all-reduces the sums
 * in between and keeps these few operations in the host language).  fp32, the operation order of NeuS_Trainer.compute_loss:
 *   cnr_loss_combine: out[1] = rgb = sums[0] / (3 Rg); out[2] = eikonal = gradient_error[0]; out[3] = mask = sums[1] / Rg (0 unless use_mask);
 *                     out[5] = mean_rel = sums[2] / (3 Rg M); out[4] = relight = mean_rel^2 (0 unless use_relight);
 *                     out[0] = loss = lambda_fine * rgb + lambda_eikonal * eikonal (+ lambda_mask * mask) (+ lambda_relight * relight)
 *   cnr_loss_coef:    the coefficients of cnr_loss_grads from the upstream gradient g = g_loss[0]:
 *                     coef[0] = g * lambda_fine * (1 for L1, 2 for MSE) / (3 Rg); coef[1] = g * lambda_mask / Rg (0 unless use_mask);
 *                     coef[2] = g * lambda_relight * 2 / (3 Rg M) * mean_rel[0] (0 unless use_relight); coef[3] = g * lambda_eikonal = d loss / d gradient_error */
int cnr_loss_combine(const cnr_loss_config* cfg, const float* sums /* device [4] */, const float* gradient_error /* device [1] */,
                     float n_rays_global, int32_t n_samples, int32_t use_mask, int32_t use_relight, float* out /* device [6] */, void* stream);
int cnr_loss_coef(const cnr_loss_config* cfg, const float* g_loss /* device [1] */, const float* mean_rel /* device [1] */,
                  float n_rays_global, int32_t n_samples, int32_t use_mask, int32_t use_relight, float* coef /* device [4] */, void* stream);

/* The same two sides as ONE launch each (single process; replaces NeuS_Trainer.compute_loss, NeuS_Trainer.py:129-171, and its autograd graph):
 *   cnr_loss_forward : cnr_loss_sums (delta_per_ray == 0: delta_relight is [R][M][3]; != 0: the per-ray sums [R]) + cnr_loss_combine; the block that
 *                      finishes last folds the partial sums in the fixed order of cnr_loss_sums -- bitwise the same sums and scalars.  Its completion
 *                      counter is the 4 bytes at offset cnr_loss_scratch_bytes() - 16 of the CALLER's scratch: zero before the first call that uses
 *                      the buffer, left zero by every call.  A buffer may be reused call after call on one stream; calls that may overlap (other
 *                      streams or threads) need buffers of their own.  cnr_loss_sums / cnr_loss_sums_ray never touch the counter.
 *   cnr_loss_backward: cnr_loss_coef (also written to coef[4] for the caller: coef[2] * mask is d / d delta_relight, coef[3] is d / d gradient_error)
 *                      + cnr_loss_grads for d_color_fine / d_weight_sum.  eik_factor (device [1] or NULL = 1): coef[3] = g * lambda_eikonal * eik_factor
 *                      (ray-sharded runs, below). */
int cnr_loss_forward(const cnr_loss_config* cfg, const float* color_fine, const float* weight_sum, const float* delta_relight /* or NULL */, int32_t delta_per_ray,
                     const float* rgb_gt, const float* mask /* or NULL */, const float* gradient_error /* device [1] */, int64_t n_rays, int32_t n_samples,
                     float n_rays_global, int32_t use_mask, int32_t use_relight, float* sums /* device [4] */, float* out /* device [6] */,
                     void* scratch, size_t scratch_bytes, void* stream);
int cnr_loss_backward(const cnr_loss_config* cfg, const float* color_fine, const float* weight_sum, const float* rgb_gt, const float* mask /* or NULL */,
                      int64_t n_rays, int32_t n_samples, const float* g_loss /* device [1] */, const float* mean_rel /* device [1] */,
                      const float* eik_factor /* device [1] or NULL */, float n_rays_global, int32_t use_mask, int32_t use_relight,
                      float* coef /* device [4] */, float* d_color_fine, float* d_weight_sum /* or NULL */,
                      float* d_delta_relight_per_ray /* [R] or NULL: coef[2] (* mask[r]), what cnr_render_out_grads.delta_relight_per_ray takes */, void* stream);

/* Ray-sharded runs (one process per GPU, the rays of a batch split over the ranks; the reference is single-GPU, train.py:111): the eikonal term is a
 * ratio of sums over ALL rays (Color_NeuS.py:122-123) and the relight term the square of a mean over ALL samples (NeuS_Trainer.py:153), so the
 * objective needs ONE exchange between the two sides.  Three launches around one 5-float all-reduce (the caller's: RCCL):
 *   cnr_loss_shard_stats  : one launch (the fold by the last block, as cnr_loss_forward; same scratch contract): stats[0..2] = the three sums of
 *                           cnr_loss_sums over this rank's rays, stats[3..4] = eik_sums of this rank (cnr_render_outputs.eik_sums),
 *                           stats[5] = stats[4] again (the rank's own value survives the all-reduce there), stats[6..7] = 0
 *   caller                : all-reduce(sum) of stats[0..5) over the ranks
 *   cnr_loss_shard_combine: out[0..5] as cnr_loss_combine with eikonal = stats[3] / (stats[4] + 1e-5), the GLOBAL loss on every rank;
 *                           out[6] = eik_factor = (stats[5] + 1e-5) / (stats[4] + 1e-5) = d eikonal / d (this rank's gradient_error output); out[7] = 0
 *   cnr_loss_backward     : with n_rays_global, mean_rel = out + 5 and eik_factor = out + 6: the rank-local gradients whose sum over the ranks is the
 *                           single-GPU gradient. */
int cnr_loss_shard_stats(const cnr_loss_config* cfg, const float* color_fine, const float* weight_sum, const float* delta_relight /* or NULL */,
                         int32_t delta_per_ray, const float* rgb_gt, const float* mask /* or NULL */, const float* eik_sums /* device [2] */, int64_t n_rays,
                         int32_t n_samples, float* stats /* device [8] */, void* scratch, size_t scratch_bytes, void* stream);
int cnr_loss_shard_combine(const cnr_loss_config* cfg, const float* stats /* device [8] */, float n_rays_global, int32_t n_samples, int32_t use_mask,
                           int32_t use_relight, float* out /* device [8] */, void* stream);

/* ---- ray generation for the selected pixels, the producer right in front of the path (NeuS_Trainer.render, NeuS_Trainer.py:104-120):
 * get_rays_multicam / get_rays_at (lib/models/tools/ray_utils.py:16-119) evaluated ONLY for the chosen pixels (the reference builds the
 * rays of all N*H*W pixels and gathers), with the trainer's (rays_o - origin) / radius normalisation and near_far_from_sphere
 * (ray_utils.py:7-13) folded in.  pix_idx: device int64 [n], flat index cam*H*W + row*W + col (NULL: pixel i of camera 0, i.e.
 * get_rays_at over a whole H x W image when n = H*W); c2w [n_cams][4][4]; focal device [2]; image [n_cams][H][W][3] / mask
 * [n_cams][H][W] may be NULL together with rgb / mask_sel; origin (device [3]) may be NULL; near_ / far_ may be NULL.
 * cnr_gen_rays_backward: d c2w [n_cams][4][4] and d focal [2] from d rays_o, d rays_d (and d near / d far when given) for learnable
 * poses / focal (config/Color_NeuS_iho.yml:18-20); scratch: n_cams * 2 floats. */
int cnr_gen_rays(const int64_t* pix_idx, int64_t n, const float* c2w, int32_t n_cams, const float* focal, int32_t H, int32_t W,
                 int32_t normalize, int32_t opengl, const float* image, const float* mask, const float* origin, float radius,
                 float* rays_o, float* rays_d, float* rgb, float* mask_sel, float* near_, float* far_,
                 int32_t* bad_index_count /* ABI 8; device or NULL: incremented once per index outside [0, n_cams*H*W) -- such a ray's outputs are NaN (the
                                             reference's torch indexing raises; the caller reads the counter at its next synchronisation point and raises) */,
                 void* stream);
int cnr_gen_rays_backward(const int64_t* pix_idx, int64_t n, const float* c2w, int32_t n_cams, const float* focal, int32_t H, int32_t W,
                          int32_t normalize, int32_t opengl, const float* origin, float radius, const float* d_rays_o, const float* d_rays_d,
                          const float* d_near, const float* d_far, float* d_c2w, float* d_focal, void* scratch, size_t scratch_bytes, void* stream);

/* ---- optimiser step of the training loop (the consumer after loss.backward(), train.py:72-77): per-parameter gradient clipping
 * (clip_gradient -> torch.nn.utils.clip_grad_norm_ on EACH parameter tensor, lib/utils/net_utils.py:174-184) followed by
 * torch.optim.Adam (net_utils.py:88: betas (0.9, 0.99), eps 1e-8, no weight decay) in ONE launch over all tensors.
 * params / grads: host arrays of n_tensors device pointers, sizes: host array of element counts; exp_avg / exp_avg_sq: device, flat,
 * sum(sizes) floats each, tensor i at offset sum(sizes[0..i)); step is 1-based; max_norm <= 0 disables the clip.
 * scratch: device, cnr_clip_adam_scratch_bytes(n_tensors, sizes) bytes (per-chunk partial sums of the gradient norms). */
typedef struct cnr_adam_config {
  float lr, beta1, beta2, eps, max_norm;
  int32_t step;
  const float* hyper_dev;   /* ABI 8; NULL, or DEVICE [3] = {lr, 1 - beta1^step, sqrt(1 - beta2^step)}: the step-dependent scalars read from device memory
                               instead of lr / step above, so that a launch sequence captured ONCE in a HIP graph (hipStreamBeginCapture; every entry point
                               only enqueues on the caller's stream and allocates nothing) can be replayed for every step -- the caller refreshes the three
                               floats before each replay (optim.ClipAdam(capturable=True).prepare_step) */
} cnr_adam_config;
size_t cnr_clip_adam_scratch_bytes(int32_t n_tensors, const int64_t* sizes);
int cnr_clip_adam_step(const cnr_adam_config* cfg, int32_t n_tensors, const int64_t* sizes, float* const* params, const float* const* grads,
                       float* exp_avg, float* exp_avg_sq, void* scratch, size_t scratch_bytes, void* stream);

int cnr_abi_version(void);
const char* cnr_backend_name(void);
const char* cnr_last_error(void);

/* parameter inventory in canonical order */
int cnr_param_count(const cnr_config* cfg);
int cnr_param_info(const cnr_config* cfg, int index, char* name, int name_len, int* rows, int* cols);

/* bytes of the context buffer (saved activations, written by forward, read by backward) and of the backward scratch */
size_t cnr_ctx_bytes(const cnr_config* cfg, int64_t n_rays);
size_t cnr_bwd_scratch_bytes(const cnr_config* cfg, int64_t n_rays);

/* renderer(rays_o, rays_d, near, far) -- NeuS.forward / Color_NeuS.render_core */
int cnr_render_forward(const cnr_config* cfg, const float* const* params, const cnr_render_inputs* in,
                       const cnr_render_outputs* out, void* ctx, size_t ctx_bytes, void* stream);

/* Forward-only render -- the inference use of the path: NeuS_Trainer.validate_image (NeuS_Trainer.py:236-245 consumes color_fine and depth of every
 * chunk) and evaluation.py.  Same inputs, same outputs, BIT-IDENTICAL values to cnr_render_forward (every value-producing launch is the same
 * kernel on the same operands), but nothing is written for cnr_render_backward: no row scales, no hidden activations of the colour / relight
 * stacks, two reused buffers instead of the saved V_l of the gradient chain.  `scratch` (cnr_infer_scratch_bytes, about half of
 * cnr_ctx_bytes: 14.1 GB against 29.7 GB for 8192 rays) holds no state after the call.  prune_eps > 0 (cnr_render_inputs): the colour / relight stacks run only on the samples
 * whose compositing weight is >= prune_eps -- per ray a wavefront ballot + popcount builds the index list, the chain-fused launch reads its
 * rows through it; the pixel error per skipped sample is < prune_eps, the per-sample colour outputs of skipped samples are zero. */
size_t cnr_infer_scratch_bytes(const cnr_config* cfg, int64_t n_rays);
int cnr_render_forward_only(const cnr_config* cfg, const float* const* params, const cnr_render_inputs* in, const cnr_render_outputs* out,
                            void* scratch, size_t scratch_bytes, void* stream);

/* the hierarchical sampler on its own (NeuS.forward up to the render_core call, NeuS.py:309-356): writes the final z_vals [R][M];
 * ctx is a buffer of cnr_ctx_bytes(cfg, n_rays) bytes used as scratch */
int cnr_sample_z(const cnr_config* cfg, const float* const* params, const cnr_render_inputs* in, float* z_vals, void* ctx, size_t ctx_bytes,
                 void* stream);

/* autograd backward of cnr_render_forward (loss.backward(), train.py:70): parameter gradients incl. the
 * second-order terms through grad_x SDF, and optionally d rays_o / d rays_d */
int cnr_render_backward(const cnr_config* cfg, const float* const* params, const cnr_render_inputs* in,
                        const cnr_render_outputs* out, const void* ctx, size_t ctx_bytes,
                        const cnr_render_out_grads* gout, const cnr_render_in_grads* gin,
                        void* scratch, size_t scratch_bytes, void* stream);

/* the two sampler functions on their own (the render call runs them inside the hierarchical sampler; these entry points expose the
 * same kernel so that its tie / flat-cdf / denom < 1e-5 semantics can be exercised directly):
 *   cnr_sample_pdf = ray_utils.sample_pdf(bins, weights, n_samples, det=True)            (lib/models/tools/ray_utils.py:123-154)
 *   cnr_up_sample  = NeuS.up_sample(rays_o, rays_d, z_vals, sdf, n_importance, inv_s)     (lib/models/renderers/NeuS.py:136-181)
 * n <= 256 bins / samples per ray, n_samples (n_importance) <= 64 */
int cnr_sample_pdf(const float* bins /* [R][n] */, const float* weights /* [R][n-1] */, int64_t n_rays, int32_t n, int32_t n_samples,
                   float* out /* [R][n_samples] */, void* stream);
/* ray_utils.sample_pdf(..., det=False) (ray_utils.py:135-136): the same inversion of the cdf at the caller's uniform draws u [R][n_samples] -- the
 * reference's torch.rand(list(cdf.shape[:-1]) + [n_samples]), taken by the host layer from the CPU generator exactly like the reference takes it.
 * Not on the render path (NeuS.up_sample calls det=True, NeuS.py:180); exported so that the function is complete. */
int cnr_sample_pdf_u(const float* bins, const float* weights, const float* u, int64_t n_rays, int32_t n, int32_t n_samples, float* out, void* stream);
int cnr_up_sample(const float* rays_o, const float* rays_d, const float* z_vals /* [R][n] */, const float* sdf /* [R][n] */, int64_t n_rays,
                  int32_t n, int32_t n_importance, float inv_s, float* out /* [R][n_importance] */, void* stream);

/* sdf_network.sdf(pts): out[i] = sign * sdf(pts[i]); extract_fields uses sign = -1 (NeuS.py:416) */
size_t cnr_sdf_eval_scratch_bytes(const cnr_config* cfg, int64_t n_points);
int cnr_sdf_eval(const cnr_config* cfg, const float* const* params, const float* pts, int64_t n_points, float sign,
                 float* out, void* scratch, size_t scratch_bytes, void* stream);

/* extract_fields: u[x][y][z] = -sdf on linspace(bmin,bmax,res)^3 (NeuS.py:14-28); bmin/bmax are HOST floats */
size_t cnr_sdf_grid_scratch_bytes(const cnr_config* cfg, int32_t resolution);
int cnr_sdf_grid(const cnr_config* cfg, const float* const* params, const float* bound_min, const float* bound_max,
                 int32_t resolution, float* u, void* scratch, size_t scratch_bytes, void* stream);
/* the slab x in [x_begin, x_end) of the same lattice (u_slab[x - x_begin][y][z], identical values): extract_fields sharded over the GPUs
 * of a node, one slab per rank, gathered by the caller (the reference walks the lattice in 64^3 blocks, NeuS.py:19-27) */
size_t cnr_sdf_grid_slab_scratch_bytes(const cnr_config* cfg, int32_t resolution, int32_t x_begin, int32_t x_end);
int cnr_sdf_grid_slab(const cnr_config* cfg, const float* const* params, const float* bound_min, const float* bound_max, int32_t resolution,
                      int32_t x_begin, int32_t x_end, float* u_slab, void* scratch, size_t scratch_bytes, void* stream);

/* extract_geometry's iso-surface step on the device-resident lattice (NeuS.py:31-40 calls the third-party CPU mcubes.marching_cubes on
 * a host copy of u): marching cubes over u[x][y][z] at `threshold`, "inside" = u > threshold, vertices on the lattice edges by linear
 * interpolation, shared between neighbouring cells, already mapped to world coordinates (v / (res - 1) * (bmax - bmin) + bmin,
 * NeuS.py:36-39), triangle normals pointing out of the u > threshold region.  Triangle table: built by tools/gen_mc_table.py
 * (no mcubes fixture exists: vertex / triangle ORDER and the resolution of ambiguous faces are this library's own; the surface is
 * the same piecewise-linear level set).
 * Two phases because the caller owns the output buffers: cnr_mc_count classifies, scans and leaves {n_vertices, n_triangles} in the
 * device int32 pair `totals`; after reading them the caller allocates and calls cnr_mc_emit with the same scratch. */
size_t cnr_mc_scratch_bytes(int32_t resolution);
int cnr_mc_count(const float* u, int32_t resolution, float threshold, void* scratch, size_t scratch_bytes, int32_t* totals, void* stream);
int cnr_mc_emit(const float* u, int32_t resolution, float threshold, const float* bound_min /* host [3] */, const float* bound_max /* host [3] */,
                void* scratch, size_t scratch_bytes, float* vertices /* [V][3] */, int32_t* triangles /* [F][3] */, void* stream);

/* extract_color: rgb = color_network(pts, g, -g, feat) per vertex (NeuS.py:44-64) */
size_t cnr_vertex_color_scratch_bytes(const cnr_config* cfg, int64_t n_points);
int cnr_vertex_color(const cnr_config* cfg, const float* const* params, const float* verts, int64_t n_points,
                     float* rgb, void* scratch, size_t scratch_bytes, void* stream);

/* ---- one plain fully-connected layer on the layer / weight-gradient kernels of the render path: y = act(x W^T + b), act = ReLU or none.
 * The NeRF++ background network of NeuS (NeRF, fields.py:192-274; N_OUTSIDE > 0, NeuS.py:95-134) is a chain of nn.Linear (+ ReLU) layers;
 * color-neus_amd/background.py evaluates each of them through these two entry points.  x [n][k], y / dy [n][n_out], W [n_out][k] and b [n_out]
 * are compact row-major device fp32 (nn.Linear's layout); backward: dx (or NULL), dW [n_out][k], db (or NULL) are overwritten. */
size_t cnr_linear_scratch_bytes(int64_t n, int32_t k, int32_t n_out, int32_t backward);
int cnr_linear_forward(const float* x, int64_t n, int32_t k, const float* W, const float* b /* or NULL */, int32_t n_out, int32_t relu, float* y,
                       void* scratch, size_t scratch_bytes, void* stream);
int cnr_linear_backward(const float* x, const float* y /* the forward output (ReLU gate); may be NULL without ReLU */, const float* dy, int64_t n,
                        int32_t k, const float* W, int32_t n_out, int32_t relu, float* dx, float* dW, float* db, void* scratch,
                        size_t scratch_bytes, void* stream);

/* ---- N_OUTSIDE > 0: the NeRF++ background of NeuS.forward (lib/models/renderers/NeuS.py:313-369) -----------------------------------------
 * The reference evaluates a second network (NeRF, fields.py:192-274, built with its defaults: NeuS.py:87-91) on the foreground samples
 * merged with n_outside inverse-depth samples beyond `far`, and mixes its alpha / colour into render_core by the inside-sphere mask
 * (NeuS.py:262-268, Color_NeuS.py:97-102).  No shipped configuration sets N_OUTSIDE.  Four steps, each a forward / backward pair; the
 * foreground fields come from cnr_render_forward's per-sample outputs (sdf_samples, gradients, color_samples, global_color_samples) and
 * their gradients go back through cnr_render_out_grads:
 *   cnr_outside_z                  z_vals_outside (NeuS.py:315-338) merged into the sorted z_vals_feed (NeuS.py:353-355)
 *   cnr_background_forward         render_core_outside (NeuS.py:95-134) up to its per-sample outputs: alpha and sigmoid(rgb)
 *   cnr_composite_background_*     the tail of render_core with background_alpha / background_sampled_color given
 * Parameters of the background network: cnr_nerf_param_info order = nn.Module order of NeRF (pts_linears.i.weight / .bias, views_linears.0.*,
 * feature_linear.*, alpha_linear.*, rgb_linear.*), plain nn.Linear tensors. */
typedef struct cnr_nerf_config {
  int32_t D, W;              /* depth / width of the pts stack (NeRF defaults 8 / 256) */
  int32_t multires;          /* PE of the 4-vector (x / r, 1 / r): default 10 */
  int32_t multires_view;     /* PE of the view direction: default 4 */
  int32_t skip_mask;         /* bit i: [PE | h] is concatenated after pts layer i (default 1 << 4) */
} cnr_nerf_config;
int cnr_nerf_param_count(const cnr_nerf_config* cfg);
int cnr_nerf_param_info(const cnr_nerf_config* cfg, int index, char* name, int name_len, int* rows, int* cols);

/* z_feed [R][n_z + n_outside] = sort(cat(z_vals [R][n_z] (ascending), far / flip(zz) + 1 / n_samples)); t_rand [R][n_outside] = the
 * torch.rand draw of NeuS.py:335 or NULL (no perturbation); src [R][n_z + n_outside] (int32, for the backward call): where each entry came from.
 * Backward: d_far [R] through the background samples, d_z [R][n_z] (or NULL) for the copies of z_vals. */
int cnr_outside_z(const float* far_, const float* t_rand, const float* z_vals, int64_t n_rays, int32_t n_z, int32_t n_outside, int32_t n_samples,
                  float* z_feed, int32_t* src, void* stream);
int cnr_outside_z_backward(const float* t_rand, const int32_t* src, const float* d_z_feed, int64_t n_rays, int32_t n_z, int32_t n_outside,
                           int32_t n_samples, float* d_far, float* d_z, void* stream);

/* alpha [R][n_feed], color [R][n_feed][3] of the background network at the n_feed samples of every ray; ctx keeps the activations for
 * cnr_background_backward, which overwrites d_params (host array of device pointers, cnr_nerf_param_info order), d_rays_o / d_rays_d [R][3]
 * and d_z_feed [R][n_feed]. */
size_t cnr_background_ctx_bytes(const cnr_nerf_config* cfg, int64_t n_rays, int32_t n_feed);
size_t cnr_background_bwd_scratch_bytes(const cnr_nerf_config* cfg, int64_t n_rays, int32_t n_feed);
int cnr_background_forward(const cnr_nerf_config* cfg, const float* const* params, const float* rays_o, const float* rays_d, const float* z_feed,
                           int64_t n_rays, int32_t n_feed, float sample_dist, float* alpha, float* color, void* ctx, size_t ctx_bytes, void* stream);
int cnr_background_backward(const cnr_nerf_config* cfg, const float* const* params, const float* rays_o, const float* rays_d, const float* z_feed,
                            int64_t n_rays, int32_t n_feed, float sample_dist, const void* ctx, size_t ctx_bytes, const float* color,
                            const float* d_alpha, const float* d_color, float* const* d_params, float* d_rays_o, float* d_rays_d, float* d_z_feed,
                            void* scratch, size_t scratch_bytes, void* stream);

/* render_core's alpha / mixing / compositing over n_feed = n_z + n_outside samples.  Outputs: the members of cnr_render_outputs that render_core
 * returns (weights is [R][n_feed] here; gradients / delta_relight / z_vals / the per-sample members are not written); eik_sums must be given.
 * Backward: upstream gradients in cnr_render_out_grads (color_fine, s_val, cdf_fine, weight_sum, weight_max, weights [R][n_feed],
 * gradient_error, depth, global_color, gradients), results in cnr_bg_composite_grads (all overwritten; d_variance is [1]). */
typedef struct cnr_bg_composite_in {
  const float* rays_o; const float* rays_d; const float* z_vals /* [R][n_z] */; const float* z_feed /* [R][n_feed] */;
  int64_t n_rays; int32_t n_z, n_feed; float sample_dist;
  const float* sdf_samples; const float* gradients /* [R][n_z][3] */; const float* color_samples; const float* global_color_samples /* or NULL (NeuS) */;
  const float* bg_alpha /* [R][n_feed] */; const float* bg_color /* [R][n_feed][3] */;
  const float* variance /* deviation_network.variance, device [1] */; float cos_anneal_ratio; const float* background_rgb /* [3] or NULL */;
} cnr_bg_composite_in;
typedef struct cnr_bg_composite_grads {
  float* d_sdf_samples; float* d_gradients; float* d_color_samples; float* d_global_color_samples; float* d_bg_alpha; float* d_bg_color;
  float* d_variance; float* d_rays_d /* [R][3] */; float* d_z_vals /* [R][n_z] */; float* d_z_feed /* [R][n_feed] */;
} cnr_bg_composite_grads;
size_t cnr_composite_background_scratch_bytes(int64_t n_rays);
int cnr_composite_background_forward(const cnr_bg_composite_in* in, const cnr_render_outputs* out, void* scratch, size_t scratch_bytes, void* stream);
int cnr_composite_background_backward(const cnr_bg_composite_in* in, const cnr_render_outputs* out, const cnr_render_out_grads* gout,
                                      const cnr_bg_composite_grads* gin, void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* COLORNEUS_RENDER_H_ */
