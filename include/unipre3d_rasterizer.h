/*
 * unipre3d_rasterizer.h -- C-ABI of the MI355X (gfx950) differentiable Gaussian-splat rasterizer.
 *
 * Drop-in boundary for the operator UniPre3D imports at gaussian_renderer/__init__.py:8
 * (`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`)
 * and calls at gaussian_renderer/__init__.py:45-61 (settings) and :89-97 (keyword call).
 * The third-party extension behind that import exposes three native operations
 * ([UPSTREAM-RECALL]):
 *     rasterize_gaussians            -> u3d_rasterize_forward
 *     rasterize_gaussians_backward   -> u3d_rasterize_backward
 *     mark_visible                   -> u3d_mark_visible
 * Here they are plain `extern "C"` functions over raw DEVICE pointers, sizes and a HIP stream;
 * no torch types.  All tensors are contiguous fp32 (row-major), matrices are 4x4 row-major and
 * stored for ROW-vector use (p_view = [p,1] * viewmatrix), exactly as the reference's datasets
 * produce them (dataset/shapenet.py:305-320).
 *
 * One call renders `n_items * views_per_item` views: view v uses Gaussian set v / views_per_item.
 * The reference's per-view operator is the special case n_items = views_per_item = 1; the batched
 * form replaces the B x V Python loop of train_network.py:418-446 with one launch
 * sequence.  Nothing here allocates, frees or synchronises: scratch is caller-provided (sizes from
 * u3d_scratch_query), kernels are enqueued on `stream`, and -- unlike the original operator, which
 * copies `num_rendered` back to the host on every forward -- there is no device->host copy.
 *
 * Ownership: every pointer is borrowed for the duration of the enqueued work.  The three forward
 * scratch buffers must be kept unmodified until the matching backward has run.  Not re-entrant per
 * scratch set.  Return value: U3D_OK or an error code (u3d_error_string); with U3D_FLAG_DEBUG the
 * call also synchronises the stream and reports asynchronous kernel faults.
 */
#ifndef UNIPRE3D_RASTERIZER_H
#define UNIPRE3D_RASTERIZER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define U3D_ABI_VERSION 5

/* flags (fields 11-13 of GaussianRasterizationSettings, gaussian_renderer/__init__.py:56-58) */
#define U3D_FLAG_PREFILTERED 1   /* accepted, no effect: culled points are dropped either way */
#define U3D_FLAG_ANTIALIASING 2  /* opacity *= sqrt(max(2.5e-5, det(cov)/det(cov+0.3 I))) */
#define U3D_FLAG_DEBUG 4         /* synchronise + check after the launch sequence (and validate item_offsets on the device first):
                                    the call blocks on the stream, so it cannot be captured into a HIP graph */
#define U3D_FLAG_EXACT_AA_GRAD 8 /* exact derivative of the anti-aliasing factor (DEV(vi)) */
#define U3D_FLAG_STATS 16        /* also accumulate num_rendered[view] = sum of tiles touched (same-address atomics:
                                    ~12 ns each, 0.25 ms at 1.6 M Gaussian-views -- statistics only, off by default) */
#define U3D_FLAG_ACC_CLEAN 32    /* u3d_render_loss_step and u3d_rasterize_backward: the caller guarantees that the gradient accumulators
                                    at the start of `backward_scratch` (the first acc_bytes) are all zero on entry -- because it zeroed
                                    them once, or because the previous call that used this scratch was one of these two entry points
                                    WITH THE SAME DESCRIPTOR SHAPE that returned U3D_OK (every such call leaves them zero again; another
                                    shape lays the scratch out differently).  The call then skips clearing 80 bytes per (view,
                                    Gaussian) pair: most of the step's projection-kernel traffic at scene level, one launch per call of
                                    the per-view operator route */

#define U3D_SPARSE_BWD_MIN_P 4096  /* U3D_FLAG_SPARSE_BWD is honoured for sets of MORE than this many Gaussians (the library's LDS-sort limit) */
#define U3D_FLAG_SPARSE_BWD 64   /* u3d_render_loss_step_forward / _backward (scene-level head, P > 4096 only; ignored otherwise): the
                                    caller hands the forward half the gradient buffer `d_head_out` it will pass to the backward half.
                                    The forward half zero-fills it (extra workgroups of the gradient reduction, off the critical path)
                                    and records which Gaussians received a gradient; the backward half then runs the chain rule over
                                    those few thousand instead of visiting every one of the scene's 10^5 Gaussians.  Meant to be set in
                                    both halves' descriptors or in neither; a mismatch costs speed, not correctness: a forward half
                                    without the flag marks the list as not built and a backward half with it then visits every Gaussian,
                                    a backward half without it always does.  `d_head_out` is validated (non-null, 16-byte aligned) with
                                    the other arguments, before anything is enqueued */

#define U3D_OK 0
#define U3D_ERR_INVALID_ARGUMENT 1
#define U3D_ERR_UNSUPPORTED 2
#define U3D_ERR_LAUNCH 3
#define U3D_ERR_NO_DEVICE 4

typedef struct u3d_raster_desc {
  int32_t n_items;        /* independent Gaussian sets (objects / scenes) in this call, <= 65535 */
  int32_t views_per_item; /* cameras per set; n_views = n_items * views_per_item               */
  int32_t P;              /* Gaussians per set (uniform batch); the LARGEST set of a ragged batch */
  int32_t image_height;   /* settings field 1                                                  */
  int32_t image_width;    /* settings field 2                                                  */
  float tanfovx;          /* settings field 3                                                  */
  float tanfovy;          /* settings field 4                                                  */
  float scale_modifier;   /* settings field 6                                                  */
  int32_t sh_degree;      /* settings field 9: active SH degree D in 0..3                      */
  int32_t sh_coeffs;      /* M = shs.shape[1] >= (D+1)^2; 0 when colors_precomp is used        */
  int32_t flags;          /* U3D_FLAG_*                                                        */
  /* Ragged batches -- the reference's scene-level branch returns python LISTS of per-item (M_i, .) tensors
     (model/gaussian_predictor.py:331-364).  total_P == 0: every set has exactly P Gaussians (the layouts documented below).
     total_P > 0: set i has item_offsets[i+1] - item_offsets[i] Gaussians (at most P); every per-Gaussian input and gradient is
     PACKED, [total_P][...] in set order; per-(view, Gaussian) outputs (radii, dL_dmeans2D) are packed
     [views_per_item * total_P] with the pairs of set i, view v at  views_per_item * item_offsets[i] + v * P_i . */
  int32_t total_P;
  const int32_t* item_offsets; /* DEVICE pointer to n_items + 1 prefix sums (first 0, last total_P); NULL iff total_P == 0.
                                  Trusted: checked on the device only under U3D_FLAG_DEBUG (U3D_ERR_INVALID_ARGUMENT); a set that
                                  claims more than P Gaussians is truncated to P */
} u3d_raster_desc;

typedef struct u3d_scratch_sizes {
  size_t geom_bytes;     /* per (view, Gaussian) projected state  ("geomBuffer")              */
  size_t binning_bytes;  /* depth-sorted ids / tile rects / sort temporaries ("binningBuffer") */
  size_t image_bytes;    /* per-pixel final transmittance + position limit, per-tile last contributor (bit 31: loop variant the
                            forward took, read back by the backward) ("imgBuffer") */
  size_t backward_bytes; /* per (view, Gaussian) screen-space gradient accumulators           */
  size_t num_rendered_offset; /* byte offset inside geom of uint32 num_rendered[n_views] (U3D_FLAG_STATS) */
  size_t fused_bytes;    /* u3d_render_loss_*: quaternion norms/dots + per-tile loss partials  */
} u3d_scratch_sizes;

/* Gaussian head layout for the fused entry points (model/gaussian_predictor.py:174-181, 249-254). */
typedef struct u3d_head_desc {
  int32_t mode;        /* 1 = object level (quaternions normalised ACROSS THE SET'S POINTS, the reference's
                          F.normalize on (B,4,N), :254,:318), 2 = scene level (per quaternion, :347-349) */
  int32_t channels;    /* C = 11 + 3*(D+1)^2: xyz 3 | opacity 1 | scaling 3 | rotation 4 | SH 3*(D+1)^2 */
  float offset_scale;  /* cfg.model.offset_scale                                                         */
  int32_t isotropic;   /* cfg.model.isotropic (model/gaussian_predictor.py:308-310): the first scaling channel is used for all
                          three axes; the other two channels receive zero gradient                        */
} u3d_head_desc;

/* Render loss fused into the rasterizer (utils/loss_utils.py:17-45 via train_network.py:260-302). */
typedef struct u3d_loss_desc {
  int32_t kind;                  /* 1 = l2, 2 = focal_l2, 3 = l1                                   */
  float non_bg_color_loss_rate;  /* focal_l2 only (configs/transformer_pretraining.yaml:31-32: 4, 1) */
  float bg_color_loss_rate;
} u3d_loss_desc;

int u3d_abi_version(void);
const char* u3d_error_string(int code);

/* Sizes of the caller-allocated scratch buffers for `desc` (pure host arithmetic). */
int u3d_scratch_query(const u3d_raster_desc* desc, u3d_scratch_sizes* out);

/*
 * Forward: replaces `_C.rasterize_gaussians` behind gaussian_renderer/__init__.py:89-97.
 *   bg              [3]
 *   means3D         [n_items][P][3]
 *   shs             [n_items][P][M][3]   or NULL  (exactly one of shs / colors_precomp)
 *   colors_precomp  [n_items][P][3]      or NULL
 *   opacities       [n_items][P]
 *   scales          [n_items][P][3], rotations [n_items][P][4] (r,x,y,z; NOT normalised here)
 *   cov3D_precomp   [n_items][P][6]      or NULL  (exactly one of scales+rotations / cov3D_precomp)
 *   viewmatrix, projmatrix [n_views][16]; campos [n_views][3]
 * outputs
 *   out_color       [n_views][3][H][W]
 *   out_invdepth    [n_views][1][H][W]   or NULL
 *   radii           [n_views][P] int32   (0 = culled / off-screen)
 */
int u3d_rasterize_forward(const u3d_raster_desc* desc, const float* bg, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, const float* campos, float* out_color, float* out_invdepth,
                          int32_t* radii, void* geom, void* binning, void* image, void* stream);

/*
 * Backward: replaces `_C.rasterize_gaussians_backward` (triggered by loss.backward(),
 * train_network.py:333).  Inputs as in forward plus the forward's radii/scratch and
 *   dL_dcolor     [n_views][3][H][W]
 *   dL_dinvdepth  [n_views][1][H][W] or NULL (treated as zeros; the reference drops that output)
 * outputs (written, not accumulated; per set, summed over that set's views)
 *   dL_dmeans3D [n_items][P][3], dL_dopacity [n_items][P],
 *   dL_dshs [n_items][P][M][3] (NULL iff shs NULL), dL_dcolors [n_items][P][3] (may be NULL),
 *   dL_dscales [n_items][P][3], dL_drotations [n_items][P][4] (NULL iff scales NULL),
 *   dL_dcov3D [n_items][P][6] (may be NULL),
 *   dL_dmeans2D [n_views][P][3] (may be NULL): screen-space gradient, the `viewspace_points` sink of
 *                                               gaussian_renderer/__init__.py:29.
 *   backward_scratch: backward_bytes; the call clears its accumulators first unless U3D_FLAG_ACC_CLEAN vouches for them, and
 *                     always hands them back zero.
 */
int u3d_rasterize_backward(const u3d_raster_desc* desc, const float* bg, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                           const float* projmatrix, const float* campos, const int32_t* radii,
                           const float* dL_dcolor, const float* dL_dinvdepth, const void* geom,
                           const void* binning, const void* image, void* backward_scratch, float* dL_dmeans3D,
                           float* dL_dmeans2D, float* dL_dshs, float* dL_dcolors, float* dL_dopacity,
                           float* dL_dscales, float* dL_drotations, float* dL_dcov3D, void* stream);

/*
 * Fused render-loss step (SURVEY.md N2 + N3): everything between the Gaussian head's Linear output and the
 * scalar loss in ONE launch sequence -- replaces, for all B objects x V views of a step,
 *   model/gaussian_predictor.py:279-328 (activations), gaussian_renderer/__init__.py:78-97 (SH concat + operator),
 *   train_network.py:418-446 (per-object / per-view loop, torch.stack) and utils/loss_utils.py:17-45 (loss).
 *   head_out [n_items][P][C]  raw head output, point-major (the contiguous result of `final`, before the permute
 *                             of model/point_predictor.py:100)
 *   center   [n_items][P][3]  point centres added to tanh(xyz)*offset_scale
 *   gt       [n_views][3][H][W]
 * outputs: out_color [n_views][3][H][W] (kept: the backward reads it), radii, loss_out[1] = mean loss.
 * The loss value is reduced in a fixed order (deterministic).  `bg` is both the render background and the
 * colour focal_l2 compares gt against (train_network.py:270-283 uses the same tensor for both).
 */
int u3d_render_loss_forward(const u3d_raster_desc* desc, const u3d_head_desc* head, const u3d_loss_desc* loss,
                            const float* bg, const float* head_out, const float* center, const float* viewmatrix,
                            const float* projmatrix, const float* campos, const float* gt, float* out_color,
                            int32_t* radii, float* loss_out, void* geom, void* binning, void* image, void* fused,
                            void* stream);

/*
 * Backward of the fused step:
 *   d_head_out [n_items][P][C] = d(head_out) of   dloss[0] * loss  +  <dL_dcolor_extra, out_color>
 * dloss: device scalar dL/dloss.  dL_dcolor_extra [n_views][3][H][W] or NULL: the gradient of any further image-space term of
 * the caller's objective with respect to the rendered images -- the reference's loss becomes
 * `l12 + lambda_lpips * LPIPS(rendered, gt)` after `start_lpips_after` iterations (train_network.py:284-300); its dL/dcolor is
 * added to the in-kernel loss seed, so that objective stays one launch sequence.
 */
int u3d_render_loss_backward(const u3d_raster_desc* desc, const u3d_head_desc* head, const u3d_loss_desc* loss,
                             const float* bg, const float* head_out, const float* center, const float* viewmatrix,
                             const float* projmatrix, const float* campos, const float* gt, const int32_t* radii,
                             const float* out_color, const float* dloss, const float* dL_dcolor_extra, const void* geom,
                             const void* binning, const void* image, void* fused, void* backward_scratch, float* d_head_out,
                             void* stream);

/*
 * Training form of the fused step: forward AND backward in one launch sequence whose tile kernel blends, evaluates the
 * loss term, seeds dL/dcolor and walks the same LDS-resident Gaussian batch back to front, so final transmittance, last
 * contributor and the colour image never round-trip through HBM.
 *   out_color   [n_views][3][H][W] or NULL (not needed for training)
 *   loss_out[1] mean loss;  d_head_out [n_items][P][C] = d loss / d head_out  (i.e. for dL/dloss = 1; scale on the host)
 * Scratch: geom, binning, fused and backward_scratch as above (no image buffer).  On return the gradient accumulators in
 * backward_scratch are zero again (see U3D_FLAG_ACC_CLEAN).
 */
int u3d_render_loss_step(const u3d_raster_desc* desc, const u3d_head_desc* head, const u3d_loss_desc* loss, const float* bg,
                         const float* head_out, const float* center, const float* viewmatrix, const float* projmatrix,
                         const float* campos, const float* gt, float* out_color, int32_t* radii, float* loss_out,
                         float* d_head_out, void* geom, void* binning, void* fused, void* backward_scratch, void* stream);

/*
 * The same step in the two halves autograd calls it in (ABI 4) -- so that a plain `loss.backward()` (train_network.py:333) costs
 * what the one-call form does: the forward half runs everything up to the loss value and the REDUCED screen-space gradient
 * accumulators (projection [+ depth sort], the single-pass tile kernel, the fixed-order reduce); the backward half runs the chain
 * rule from those accumulators to d_head_out (projection backward, across-point quaternion term) and multiplies by the device
 * scalar `dloss` (autograd's grad_output; NULL = 1) as it READS the accumulators -- no separate d_head * g launch.
 *   geom, binning, fused, backward_scratch, radii: the forward half's, unmodified in between.
 * On return of the backward half the gradient accumulators are zero again (U3D_FLAG_ACC_CLEAN); a forward half whose backward
 * never runs leaves them dirty (the caller withdraws its promise).
 * u3d_render_loss_step == forward half + backward half with dloss = NULL.
 */
int u3d_render_loss_step_forward(const u3d_raster_desc* desc, const u3d_head_desc* head, const u3d_loss_desc* loss,
                                 const float* bg, const float* head_out, const float* center, const float* viewmatrix,
                                 const float* projmatrix, const float* campos, const float* gt, float* out_color,
                                 int32_t* radii, float* loss_out, void* geom, void* binning, void* fused,
                                 void* backward_scratch, float* d_head_out /* U3D_FLAG_SPARSE_BWD: required; else may be NULL */,
                                 void* stream);
int u3d_render_loss_step_backward(const u3d_raster_desc* desc, const u3d_head_desc* head, const float* head_out,
                                  const float* center, const float* viewmatrix, const float* projmatrix, const float* campos,
                                  const int32_t* radii, const float* dloss, const void* geom, const void* binning, void* fused,
                                  void* backward_scratch, float* d_head_out, void* stream);

/*
 * The body of the reference's per-view wrapper as ONE launch sequence (ABI 5): replaces, inside `render_predicted`,
 *   gaussian_renderer/__init__.py:78-79   shs = torch.cat([features_dc, features_rest], dim=1)   (a new (P, M, 3) tensor per view),
 *   gaussian_renderer/__init__.py:89-97   the operator call, whose third output (inverse depth) the wrapper drops, and
 *   gaussian_renderer/__init__.py:100-104 "visibility_filter": radii > 0.
 * The SH coefficients are read through TWO pointers -- features_dc [n_items][P][1][3] (coefficient 0) and features_rest
 * [n_items][P][M-1][3] (coefficients 1..M-1; NULL iff M == 1) with desc->sh_coeffs = M -- and their gradients are written through two
 * pointers as well, so neither the concatenation nor its backward (two strided copies per view) exists.  No inverse-depth plane
 * is written (the tile kernel's variant without it).  visibility [n_views][P] (uint8, may be NULL) = radii > 0, written by the
 * projection kernel.  Everything else -- arguments, scratch, ownership, ragged batches, error codes -- as u3d_rasterize_forward /
 * u3d_rasterize_backward (dL_dinvdepth is taken as zero: that output does not exist here).  The reference calls the wrapper once
 * per object and view (train_network.py:418-446), where launches, not bytes, are the cost: 2 launches forward, 3 backward
 * (the same fixed-order gradient reduction as the operator: run-to-run bit-identical).
 */
int u3d_render_view_forward(const u3d_raster_desc* desc, const float* bg, const float* means3D, const float* features_dc,
                            const float* features_rest, const float* opacities, const float* scales, const float* rotations,
                            const float* viewmatrix, const float* projmatrix, const float* campos, float* out_color,
                            int32_t* radii, uint8_t* visibility, void* geom, void* binning, void* image, void* stream);
int u3d_render_view_backward(const u3d_raster_desc* desc, const float* bg, const float* means3D, const float* features_dc,
                             const float* features_rest, const float* opacities, const float* scales, const float* rotations,
                             const float* viewmatrix, const float* projmatrix, const float* campos, const int32_t* radii,
                             const float* dL_dcolor, const void* geom, const void* binning, const void* image,
                             void* backward_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dfeatures_dc,
                             float* dL_dfeatures_rest, float* dL_dopacity, float* dL_dscales, float* dL_drotations, void* stream);

/* Frustum test only: replaces `_C.mark_visible` (no caller in the reference tree). present[P] = z_view > 0.2 */
int u3d_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream);

/*
 * Measurement hooks (bench.py / profiling only; no reference counterpart).  While enabled, every kernel
 * of the launch sequences is bracketed by a pair of HIP events recorded on the caller's stream, so that
 * per-kernel durations can be read without an external profiler.
 *   kind: 0 preprocess_fwd, 1 depth_sort, 2 render_fwd, 3 render_bwd (+ partial reduce), 4 preprocess_bwd,
 *         5 render_fb (fused forward+backward tile kernel + partial reduce)
 * u3d_profile_begin(max_records) allocates the event ring and enables recording (U3D_ERR_INVALID_ARGUMENT
 * if already enabled); a NEGATIVE argument -((stride << 26) | (kind_mask << 20) | max_records) records only the kinds in
 * kind_mask and only every stride-th launch of a kind (stride 0 = 1; every recorded scope puts two event records on the
 * stream, ~4-5 us of GPU idle each); u3d_profile_end waits for the recorded events, writes total milliseconds and launch
 * counts per kind into ms[U3D_PROFILE_KINDS] / count[U3D_PROFILE_KINDS], frees the events and disables.
 */
#define U3D_PROFILE_KINDS 6
int u3d_profile_begin(int32_t max_records);
int u3d_profile_end(float* ms, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* UNIPRE3D_RASTERIZER_H */
