// extern "C" entry points of libunipre3d_rasterizer.so (see include/unipre3d_rasterizer.h).
// Host side only: argument checks, scratch carving, launch sequence on the caller's stream.
#include <cstdio>

#include "u3d_common.h"

void u3d_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s);

#include <vector>

namespace {

// ---- optional per-kernel HIP-event timing (u3d_profile_begin / _end) ---------------------------
struct ProfRec { hipEvent_t a, b; int kind; };
bool g_prof_on = false;
unsigned g_prof_mask = 0xffffffffu;   // kinds to record (bit k = kind k)
unsigned g_prof_stride = 1;           // record every g_prof_stride-th scope of a kind
unsigned g_prof_seen[U3D_PROFILE_KINDS] = {0};
std::vector<ProfRec> g_prof;   // pre-created events
size_t g_prof_used = 0;

struct ProfScope {
  ProfRec* r = nullptr;
  hipStream_t s;
  ProfScope(int kind, hipStream_t st) : s(st) {
    if (g_prof_on && ((g_prof_mask >> kind) & 1u) && (g_prof_seen[kind]++ % g_prof_stride) == 0 && g_prof_used < g_prof.size()) {
      r = &g_prof[g_prof_used++];
      r->kind = kind;
      (void)hipEventRecord(r->a, s);
    }
  }
  ~ProfScope() {
    if (r) (void)hipEventRecord(r->b, s);
  }
};

int check_desc(const u3d_raster_desc* d) {
  if (!d) return U3D_ERR_INVALID_ARGUMENT;
  if (d->n_items < 0 || d->views_per_item < 0 || d->P < 0 || d->total_P < 0) return U3D_ERR_INVALID_ARGUMENT;
  if ((d->total_P > 0) != (d->item_offsets != nullptr)) return U3D_ERR_INVALID_ARGUMENT;   // ragged batches bring their prefix sums
  if (d->total_P > 0 && ((long long)d->total_P > (long long)d->n_items * d->P || d->total_P < d->P)) return U3D_ERR_INVALID_ARGUMENT;
  if (d->n_items > 65535) return U3D_ERR_UNSUPPORTED;   // items are a grid y dimension of the per-Gaussian kernels
  if (d->P > U3D_LDS_SORT_MAX && (long long)d->n_items * d->views_per_item > 65535) return U3D_ERR_UNSUPPORTED;   // (so are the views of the radix passes)
  if (d->image_height <= 0 || d->image_width <= 0) return U3D_ERR_INVALID_ARGUMENT;
  if (d->image_height > 65535 * U3D_TILE || d->image_width > 65535 * U3D_TILE) return U3D_ERR_UNSUPPORTED;
  if (d->sh_degree < 0 || d->sh_degree > 3) return U3D_ERR_UNSUPPORTED;
  if (!(d->tanfovx > 0.f) || !(d->tanfovy > 0.f)) return U3D_ERR_INVALID_ARGUMENT;
  // the launchers index views, tiles and (view, Gaussian) pairs with 32-bit integers: refuse shapes whose products leave them
  const long long NV = (long long)d->n_items * d->views_per_item;
  const long long T = (long long)((d->image_width + U3D_TILE - 1) / U3D_TILE) * ((d->image_height + U3D_TILE - 1) / U3D_TILE);
  if (NV >= (1ll << 31) || T >= (1ll << 31) || NV * T >= (1ll << 31)) return U3D_ERR_UNSUPPORTED;
  const long long SP = d->total_P > 0 ? (long long)d->total_P : (long long)d->n_items * d->P;   // Gaussians in the call
  if (SP >= (1ll << 31)) return U3D_ERR_UNSUPPORTED;
  if (SP * d->views_per_item >= (1ll << 32)) return U3D_ERR_UNSUPPORTED;   // sorted positions / pair ids are uint32
  return U3D_OK;
}

U3DSource plain_source(const u3d_raster_desc& d, const float* means3D, const float* shs, const float* colors_precomp,
                       const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp) {
  U3DSource src{};
  src.means = means3D; src.s_means = 3;
  src.shs = shs; src.s_shs = d.sh_coeffs * 3;
  src.colors = colors_precomp;
  src.opac = opacities; src.s_opac = 1;
  src.scales = scales; src.s_scales = 3;
  src.rots = rotations; src.s_rots = 4;
  src.cov = cov3D_precomp;
  src.act = 0;
  return src;
}

U3DSource head_source(const u3d_raster_desc& d, const u3d_head_desc& h, const float* head_out, const float* center,
                      const float* qnorm) {
  U3DSource src{};
  const int C = h.channels;
  src.means = head_out; src.s_means = C;
  src.opac = head_out + 3; src.s_opac = C;
  src.scales = head_out + 4; src.s_scales = C;
  src.rots = head_out + 7; src.s_rots = C;
  src.shs = head_out + 11; src.s_shs = C;
  src.colors = nullptr; src.cov = nullptr;
  src.act = h.mode;
  src.iso = h.isotropic != 0;
  src.center = center;
  src.offset_scale = h.offset_scale;
  src.qnorm = qnorm;
  src.qnorm_out = nullptr; src.qdot_zero = nullptr;
  (void)d;
  return src;
}

int check_fused(const u3d_raster_desc* d, const u3d_head_desc* h, const u3d_loss_desc* l) {
  int rc = check_desc(d);
  if (rc != U3D_OK) return rc;
  if (!h || !l) return U3D_ERR_INVALID_ARGUMENT;
  if (h->mode != 1 && h->mode != 2) return U3D_ERR_INVALID_ARGUMENT;
  const int K = (d->sh_degree + 1) * (d->sh_degree + 1);
  if (h->channels != 11 + 3 * K || d->sh_coeffs != K) return U3D_ERR_INVALID_ARGUMENT;
  if (l->kind < 1 || l->kind > 3) return U3D_ERR_INVALID_ARGUMENT;
  if (l->kind == 2 && !(l->non_bg_color_loss_rate + l->bg_color_loss_rate > 0.f)) return U3D_ERR_INVALID_ARGUMENT;
  return U3D_OK;
}

U3DLoss make_loss(const u3d_raster_desc& d, const u3d_loss_desc& l, const float* gt, float* partial, const float* dloss) {
  U3DLoss L{};
  L.kind = l.kind;
  L.gt = gt;
  const float sum = l.bg_color_loss_rate + l.non_bg_color_loss_rate;
  L.w_bg = l.kind == 2 ? 2.f * l.bg_color_loss_rate / sum : 1.f;
  L.w_non = l.kind == 2 ? 2.f * l.non_bg_color_loss_rate / sum : 1.f;
  L.inv_count = (float)(1.0 / ((double)d.n_items * d.views_per_item * 3.0 * d.image_height * d.image_width));
  L.partial = partial;
  L.dloss = dloss;
  return L;
}

// Ragged batches: the kernels trust `item_offsets` (n_items + 1 prefix sums on the DEVICE).  With U3D_FLAG_DEBUG the layout is
// checked before anything is launched: first 0, non-decreasing, no set larger than desc.P, last == total_P.  (Without the flag a
// set that claims more than desc.P Gaussians is truncated to desc.P by u3d_set_span -- its tail is never projected -- instead
// of indexing the sort's LDS keys out of bounds; the host bindings derive P and the prefix sums from the sets' sizes themselves.)
// The verdict travels through a word of the CALLER's scratch (the first 4 bytes of `geom`, which nothing has written yet at this
// point of a forward call), so two streams / threads / devices validating at once never share state.  The check synchronises the
// stream: a U3D_FLAG_DEBUG call cannot be captured into a HIP graph.
__global__ void validate_offsets_kernel(const int32_t* __restrict__ off, int n_items, int P, int total_P, int* __restrict__ flag) {
  int bad = 0;
  for (int i = threadIdx.x; i < n_items; i += blockDim.x) {
    const int a = off[i], b = off[i + 1];
    if (b < a || b - a > P || a < 0 || b > total_P) bad = 1;
  }
  if (threadIdx.x == 0 && (off[0] != 0 || off[n_items] != total_P)) bad = 1;
  if (bad) *flag = 1;
}
int validate_offsets(const u3d_raster_desc& d, void* scratch_word, hipStream_t s) {
  if (d.total_P <= 0 || !(d.flags & U3D_FLAG_DEBUG)) return U3D_OK;
  if (!scratch_word) return U3D_ERR_INVALID_ARGUMENT;
  int bad = 0;
  if (hipMemsetAsync(scratch_word, 0, sizeof(int), s) != hipSuccess) return U3D_ERR_LAUNCH;
  hipLaunchKernelGGL(validate_offsets_kernel, dim3(1), dim3(256), 0, s, d.item_offsets, d.n_items, d.P, d.total_P, (int*)scratch_word);
  if (hipMemcpyAsync(&bad, scratch_word, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return U3D_ERR_LAUNCH;
  if (hipStreamSynchronize(s) != hipSuccess) return U3D_ERR_LAUNCH;
  return bad ? U3D_ERR_INVALID_ARGUMENT : U3D_OK;
}

int finish(const u3d_raster_desc* d, hipStream_t s) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    std::fprintf(stderr, "[unipre3d_rasterizer] launch failed: %s\n", hipGetErrorString(e));
    return U3D_ERR_LAUNCH;
  }
  if (d->flags & U3D_FLAG_DEBUG) {
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      std::fprintf(stderr, "[unipre3d_rasterizer] kernel fault: %s\n", hipGetErrorString(e));
      return U3D_ERR_LAUNCH;
    }
  }
  return U3D_OK;
}

}  // namespace

extern "C" {

int u3d_abi_version(void) { return U3D_ABI_VERSION; }

const char* u3d_error_string(int code) {
  switch (code) {
    case U3D_OK: return "ok";
    case U3D_ERR_INVALID_ARGUMENT: return "invalid argument";
    case U3D_ERR_UNSUPPORTED: return "unsupported configuration";
    case U3D_ERR_LAUNCH: return "kernel launch or execution failed";
    case U3D_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown error";
  }
}

int u3d_scratch_query(const u3d_raster_desc* desc, u3d_scratch_sizes* out) {
  const int rc = check_desc(desc);
  if (rc != U3D_OK || !out) return rc != U3D_OK ? rc : U3D_ERR_INVALID_ARGUMENT;
  const U3DLayout L = u3d_carve(*desc, nullptr, nullptr, nullptr, nullptr);
  out->geom_bytes = L.geom_bytes;
  out->binning_bytes = L.binning_bytes;
  out->image_bytes = L.image_bytes;
  out->backward_bytes = L.backward_bytes;
  out->num_rendered_offset = L.num_rendered_offset;
  out->fused_bytes = L.fused_bytes;
  return U3D_OK;
}

int u3d_rasterize_forward(const u3d_raster_desc* desc, const float* bg, const float* means3D, const float* shs,
                          const float* colors_precomp, const float* opacities, const float* scales,
                          const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                          const float* projmatrix, const float* campos, float* out_color, float* out_invdepth,
                          int32_t* radii, void* geom, void* binning, void* image, void* stream) {
  int rc = check_desc(desc);
  if (rc != U3D_OK) return rc;
  const u3d_raster_desc& d = *desc;
  const int NV = d.n_items * d.views_per_item;
  if (NV == 0) return U3D_OK;
  if (!bg || !viewmatrix || !projmatrix || !campos || !out_color || !geom || !binning || !image)
    return U3D_ERR_INVALID_ARGUMENT;
  if (d.P > 0) {
    if (!means3D || !opacities || !radii) return U3D_ERR_INVALID_ARGUMENT;
    if ((shs == nullptr) == (colors_precomp == nullptr)) return U3D_ERR_INVALID_ARGUMENT;
    const bool sr = scales != nullptr && rotations != nullptr;
    if (sr == (cov3D_precomp != nullptr) || (scales != nullptr) != (rotations != nullptr)) return U3D_ERR_INVALID_ARGUMENT;
    if (shs && d.sh_coeffs < (d.sh_degree + 1) * (d.sh_degree + 1)) return U3D_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = (hipStream_t)stream;
  if ((rc = validate_offsets(d, geom, s)) != U3D_OK) return rc;
  U3DBuffers b{};
  u3d_carve(d, geom, binning, image, &b);
  if (d.flags & U3D_FLAG_STATS) (void)hipMemsetAsync(b.num_rendered, 0, sizeof(uint32_t) * NV, s);
  if (d.P == 0) (void)hipMemsetAsync(b.n_vis, 0, sizeof(uint32_t) * NV, s);   // the sort writes n_vis whenever P > 0
  if (d.P > 0) {
    {
      ProfScope ps(0, s);
      u3d_launch_preprocess_fwd(d, b, plain_source(d, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp),
                                viewmatrix, projmatrix, campos, radii, nullptr, s);
    }
    if (!u3d_preprocess_sorts(d)) {
      ProfScope ps(1, s);
      u3d_launch_depth_sort(d, b, radii, s);
    }
  }
  {
    ProfScope ps(2, s);
    u3d_launch_render_fwd(d, b, bg, out_color, out_invdepth, U3DLoss{}, s);
  }
  return finish(desc, s);
}

int u3d_rasterize_backward(const u3d_raster_desc* desc, const float* bg, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, const float* viewmatrix,
                           const float* projmatrix, const float* campos, const int32_t* radii, const float* dL_dcolor,
                           const float* dL_dinvdepth, const void* geom, const void* binning, const void* image,
                           void* backward_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dshs,
                           float* dL_dcolors, float* dL_dopacity, float* dL_dscales, float* dL_drotations,
                           float* dL_dcov3D, void* stream) {
  int rc = check_desc(desc);
  if (rc != U3D_OK) return rc;
  const u3d_raster_desc& d = *desc;
  const int NV = d.n_items * d.views_per_item;
  if (NV == 0 || d.P == 0) return U3D_OK;
  if (!bg || !means3D || !opacities || !viewmatrix || !projmatrix || !campos || !radii || !dL_dcolor || !geom ||
      !binning || !image || !backward_scratch || !dL_dmeans3D || !dL_dopacity)
    return U3D_ERR_INVALID_ARGUMENT;
  if ((shs == nullptr) == (colors_precomp == nullptr)) return U3D_ERR_INVALID_ARGUMENT;
  if ((scales != nullptr) != (rotations != nullptr) || (scales != nullptr) == (cov3D_precomp != nullptr))
    return U3D_ERR_INVALID_ARGUMENT;
  if (shs && !dL_dshs) return U3D_ERR_INVALID_ARGUMENT;
  if (scales && (!dL_dscales || !dL_drotations)) return U3D_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  U3DBuffers b{};
  const U3DLayout L = u3d_carve(d, (void*)geom, (void*)binning, (void*)image, &b);
  double* acc = (double*)backward_scratch;
  float* part = (float*)((char*)backward_scratch + L.acc_bytes);
  // U3D_FLAG_ACC_CLEAN: the caller keeps this scratch between calls of this shape and vouches that the accumulators are zero
  // (the previous call handed them back zeroed, see below): no memset node -- one launch less per call of the per-view route
  const bool clean = (d.flags & U3D_FLAG_ACC_CLEAN) != 0;
  if (!clean) (void)hipMemsetAsync(acc, 0, L.acc_bytes, s);
  {
    ProfScope ps(3, s);
    u3d_launch_render_bwd(d, b, bg, dL_dcolor, dL_dinvdepth, nullptr, U3DLoss{}, acc, part, s);
  }
  {
    ProfScope ps(4, s);
    U3DGradSink sink{};
    sink.means = dL_dmeans3D; sink.shs = shs ? dL_dshs : nullptr; sink.colors = dL_dcolors; sink.opac = dL_dopacity;
    sink.scales = scales ? dL_dscales : nullptr; sink.rots = scales ? dL_drotations : nullptr; sink.cov = dL_dcov3D;
    sink.means2D = dL_dmeans2D; sink.qdot = nullptr;
    // (always hands the accumulators it read back zeroed: only the touched (view, Gaussian) pairs were ever written)
    u3d_launch_preprocess_bwd(d, b, plain_source(d, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp),
                              viewmatrix, projmatrix, campos, radii, acc, sink, s, acc);
  }
  return finish(desc, s);
}

// The per-view wrapper's body (header: u3d_render_view_*): split SH pointers, no inverse-depth plane, visibility from the projection.
int u3d_render_view_forward(const u3d_raster_desc* desc, const float* bg, const float* means3D, const float* features_dc,
                            const float* features_rest, const float* opacities, const float* scales, const float* rotations,
                            const float* viewmatrix, const float* projmatrix, const float* campos, float* out_color,
                            int32_t* radii, uint8_t* visibility, void* geom, void* binning, void* image, void* stream) {
  int rc = check_desc(desc);
  if (rc != U3D_OK) return rc;
  const u3d_raster_desc& d = *desc;
  const int NV = d.n_items * d.views_per_item;
  if (NV == 0) return U3D_OK;
  if (!bg || !viewmatrix || !projmatrix || !campos || !out_color || !geom || !binning || !image) return U3D_ERR_INVALID_ARGUMENT;
  if (d.P > 0) {
    if (!means3D || !opacities || !radii || !features_dc || !scales || !rotations) return U3D_ERR_INVALID_ARGUMENT;
    if (d.sh_coeffs < (d.sh_degree + 1) * (d.sh_degree + 1) || (d.sh_coeffs > 1) != (features_rest != nullptr)) return U3D_ERR_INVALID_ARGUMENT;
  }
  hipStream_t s = (hipStream_t)stream;
  if ((rc = validate_offsets(d, geom, s)) != U3D_OK) return rc;
  U3DBuffers b{};
  u3d_carve(d, geom, binning, image, &b);
  if (d.flags & U3D_FLAG_STATS) (void)hipMemsetAsync(b.num_rendered, 0, sizeof(uint32_t) * NV, s);
  if (d.P == 0) (void)hipMemsetAsync(b.n_vis, 0, sizeof(uint32_t) * NV, s);
  if (d.P > 0) {
    U3DSource src = plain_source(d, means3D, features_dc, nullptr, opacities, scales, rotations, nullptr);
    src.s_shs = 3;
    src.shs_rest = features_rest; src.s_shs_rest = (d.sh_coeffs - 1) * 3;
    {
      ProfScope ps(0, s);
      u3d_launch_preprocess_fwd(d, b, src, viewmatrix, projmatrix, campos, radii, nullptr, s, visibility);
    }
    if (!u3d_preprocess_sorts(d)) {
      ProfScope ps(1, s);
      u3d_launch_depth_sort(d, b, radii, s);
    }
  }
  {
    ProfScope ps(2, s);
    u3d_launch_render_fwd(d, b, bg, out_color, nullptr, U3DLoss{}, s);
  }
  return finish(desc, s);
}

int u3d_render_view_backward(const u3d_raster_desc* desc, const float* bg, const float* means3D, const float* features_dc,
                             const float* features_rest, const float* opacities, const float* scales, const float* rotations,
                             const float* viewmatrix, const float* projmatrix, const float* campos, const int32_t* radii,
                             const float* dL_dcolor, const void* geom, const void* binning, const void* image,
                             void* backward_scratch, float* dL_dmeans3D, float* dL_dmeans2D, float* dL_dfeatures_dc,
                             float* dL_dfeatures_rest, float* dL_dopacity, float* dL_dscales, float* dL_drotations, void* stream) {
  int rc = check_desc(desc);
  if (rc != U3D_OK) return rc;
  const u3d_raster_desc& d = *desc;
  const int NV = d.n_items * d.views_per_item;
  if (NV == 0 || d.P == 0) return U3D_OK;
  if (!bg || !means3D || !features_dc || !opacities || !scales || !rotations || !viewmatrix || !projmatrix || !campos || !radii ||
      !dL_dcolor || !geom || !binning || !image || !backward_scratch || !dL_dmeans3D || !dL_dopacity || !dL_dfeatures_dc ||
      !dL_dscales || !dL_drotations)
    return U3D_ERR_INVALID_ARGUMENT;
  if ((d.sh_coeffs > 1) != (features_rest != nullptr) || (features_rest != nullptr) != (dL_dfeatures_rest != nullptr))
    return U3D_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  U3DBuffers b{};
  const U3DLayout L = u3d_carve(d, (void*)geom, (void*)binning, (void*)image, &b);
  double* acc = (double*)backward_scratch;
  float* part = (float*)((char*)backward_scratch + L.acc_bytes);
  if (!(d.flags & U3D_FLAG_ACC_CLEAN)) (void)hipMemsetAsync(acc, 0, L.acc_bytes, s);
  {
    ProfScope ps(3, s);
    u3d_launch_render_bwd(d, b, bg, dL_dcolor, nullptr, nullptr, U3DLoss{}, acc, part, s);
  }
  {
    ProfScope ps(4, s);
    U3DSource src = plain_source(d, means3D, features_dc, nullptr, opacities, scales, rotations, nullptr);
    src.s_shs = 3;
    src.shs_rest = features_rest; src.s_shs_rest = (d.sh_coeffs - 1) * 3;
    U3DGradSink sink{};
    sink.means = dL_dmeans3D; sink.shs = dL_dfeatures_dc; sink.shs_rest = dL_dfeatures_rest; sink.opac = dL_dopacity;
    sink.scales = dL_dscales; sink.rots = dL_drotations; sink.means2D = dL_dmeans2D;
    u3d_launch_preprocess_bwd(d, b, src, viewmatrix, projmatrix, campos, radii, acc, sink, s, acc);
  }
  return finish(desc, s);
}

int u3d_render_loss_forward(const u3d_raster_desc* desc, const u3d_head_desc* head, const u3d_loss_desc* loss,
                            const float* bg, const float* head_out, const float* center, const float* viewmatrix,
                            const float* projmatrix, const float* campos, const float* gt, float* out_color,
                            int32_t* radii, float* loss_out, void* geom, void* binning, void* image, void* fused,
                            void* stream) {
  int rc = check_fused(desc, head, loss);
  if (rc != U3D_OK) return rc;
  const u3d_raster_desc& d = *desc;
  const int NV = d.n_items * d.views_per_item;
  if (NV == 0 || d.P == 0) return U3D_ERR_INVALID_ARGUMENT;
  if (!bg || !head_out || !center || !viewmatrix || !projmatrix || !campos || !gt || !out_color || !radii || !loss_out ||
      !geom || !binning || !image || !fused)
    return U3D_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = validate_offsets(d, geom, s)) != U3D_OK) return rc;
  U3DBuffers b{};
  u3d_carve(d, geom, binning, image, &b);
  U3DFused f{};
  u3d_carve_fused(d, fused, &f);
  if (d.flags & U3D_FLAG_STATS) (void)hipMemsetAsync(b.num_rendered, 0, sizeof(uint32_t) * NV, s);
  U3DSource src = head_source(d, *head, head_out, center, f.qnorm);
  if (head->mode == 1) {
    if (u3d_preprocess_sorts(d)) src.qnorm_out = f.qnorm;   // P <= 256: norms computed inside preprocess_fwd
    else u3d_launch_quat_norms(d, head_out + 7, head->channels, f.qnorm, nullptr, s);
  }
  {
    ProfScope ps(0, s);
    u3d_launch_preprocess_fwd(d, b, src, viewmatrix, projmatrix, campos, radii, nullptr, s);
  }
  if (!u3d_preprocess_sorts(d)) {
    ProfScope ps(1, s);
    u3d_launch_depth_sort(d, b, radii, s);
  }
  const int T = ((d.image_width + U3D_TILE - 1) / U3D_TILE) * ((d.image_height + U3D_TILE - 1) / U3D_TILE);
  const U3DLoss L = make_loss(d, *loss, gt, f.partial, nullptr);
  {
    ProfScope ps(2, s);
    u3d_launch_render_fwd(d, b, bg, out_color, nullptr, L, s);
  }
  u3d_launch_loss_reduce(NV * T, f.partial, L.inv_count, loss_out, s);
  return finish(desc, s);
}

int u3d_render_loss_backward(const u3d_raster_desc* desc, const u3d_head_desc* head, const u3d_loss_desc* loss,
                             const float* bg, const float* head_out, const float* center, const float* viewmatrix,
                             const float* projmatrix, const float* campos, const float* gt, const int32_t* radii,
                             const float* out_color, const float* dloss, const float* dL_dcolor_extra, const void* geom,
                             const void* binning, const void* image, void* fused, void* backward_scratch, float* d_head_out,
                             void* stream) {
  int rc = check_fused(desc, head, loss);
  if (rc != U3D_OK) return rc;
  const u3d_raster_desc& d = *desc;
  const int NV = d.n_items * d.views_per_item;
  if (NV == 0 || d.P == 0) return U3D_ERR_INVALID_ARGUMENT;
  if (!bg || !head_out || !center || !viewmatrix || !projmatrix || !campos || !gt || !radii || !out_color || !dloss ||
      !geom || !binning || !image || !fused || !backward_scratch || !d_head_out)
    return U3D_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  U3DBuffers b{};
  const U3DLayout Lay = u3d_carve(d, (void*)geom, (void*)binning, (void*)image, &b);
  U3DFused f{};
  u3d_carve_fused(d, fused, &f);
  double* acc = (double*)backward_scratch;
  float* part = (float*)((char*)backward_scratch + Lay.acc_bytes);
  (void)hipMemsetAsync(acc, 0, Lay.acc_bytes, s);
  (void)hipMemsetAsync(f.qdot, 0, sizeof(float) * 4 * d.n_items, s);
  const U3DLoss L = make_loss(d, *loss, gt, f.partial, dloss);
  {
    ProfScope ps(3, s);
    u3d_launch_render_bwd(d, b, bg, dL_dcolor_extra, nullptr, out_color, L, acc, part, s);
  }
  const int C = head->channels;
  U3DGradSink sink{};
  sink.means = d_head_out; sink.opac = d_head_out + 3; sink.scales = d_head_out + 4; sink.rots = d_head_out + 7;
  sink.shs = d_head_out + 11; sink.colors = nullptr; sink.cov = nullptr; sink.means2D = nullptr; sink.qdot = f.qdot;
  {
    ProfScope ps(4, s);
    u3d_launch_preprocess_bwd(d, b, head_source(d, *head, head_out, center, f.qnorm), viewmatrix, projmatrix, campos, radii,
                              acc, sink, s);
  }
  if (head->mode == 1) u3d_launch_quat_fixup(d, head_out + 7, C, f.qnorm, f.qdot, d_head_out + 7, s);
  return finish(desc, s);
}

// Training form of the fused step, in the two halves autograd calls it in (see the header): everything up to the loss and the
// reduced screen-space gradient accumulators, then the chain rule back to d(head_out), scaled by the device scalar dL/dloss.
int u3d_render_loss_step_forward(const u3d_raster_desc* desc, const u3d_head_desc* head, const u3d_loss_desc* loss,
                                 const float* bg, const float* head_out, const float* center, const float* viewmatrix,
                                 const float* projmatrix, const float* campos, const float* gt, float* out_color,
                                 int32_t* radii, float* loss_out, void* geom, void* binning, void* fused,
                                 void* backward_scratch, float* d_head_out, void* stream) {
  int rc = check_fused(desc, head, loss);
  if (rc != U3D_OK) return rc;
  const u3d_raster_desc& d = *desc;
  const int NV = d.n_items * d.views_per_item;
  if (NV == 0 || d.P == 0) return U3D_ERR_INVALID_ARGUMENT;
  if (!bg || !head_out || !center || !viewmatrix || !projmatrix || !campos || !gt || !radii || !loss_out || !geom || !binning ||
      !fused || !backward_scratch)
    return U3D_ERR_INVALID_ARGUMENT;
  // U3D_FLAG_SPARSE_BWD: the gradient buffer is zero-filled beside the gradient reduction and the touched Gaussians are listed
  // (checked with the other arguments: nothing is in flight yet when a bad pointer is refused)
  const bool sparse = u3d_sparse_bwd(d, head->mode);
  if (sparse && (!d_head_out || (reinterpret_cast<uintptr_t>(d_head_out) & 15u) != 0u)) return U3D_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  if ((rc = validate_offsets(d, geom, s)) != U3D_OK) return rc;
  U3DBuffers b{};
  const U3DLayout Lay = u3d_carve(d, geom, binning, nullptr, &b);
  U3DFused f{};
  u3d_carve_fused(d, fused, &f);
  double* acc = (double*)backward_scratch;
  float* part = (float*)((char*)backward_scratch + Lay.acc_bytes);
  if (d.flags & U3D_FLAG_STATS) (void)hipMemsetAsync(b.num_rendered, 0, sizeof(uint32_t) * NV, s);
  // no memset nodes: quat_norms clears qdot, preprocess_fwd clears the accumulators of the (view, Gaussian) it projects
  U3DSource src = head_source(d, *head, head_out, center, f.qnorm);
  if (head->mode == 1) {
    if (u3d_preprocess_sorts(d)) { src.qnorm_out = f.qnorm; src.qdot_zero = f.qdot; }   // P <= 256: inside preprocess_fwd
    else u3d_launch_quat_norms(d, head_out + 7, head->channels, f.qnorm, f.qdot, s);
  }
  {
    ProfScope ps(0, s);
    // (the accumulators are cleared per projected pair here unless the caller vouches for them, U3D_FLAG_ACC_CLEAN)
    u3d_launch_preprocess_fwd(d, b, src, viewmatrix, projmatrix, campos, radii, (d.flags & U3D_FLAG_ACC_CLEAN) ? nullptr : acc, s);
  }
  if (!u3d_preprocess_sorts(d)) {
    ProfScope ps(1, s);
    u3d_launch_depth_sort(d, b, radii, s);
  }
  const U3DLoss L = make_loss(d, *loss, gt, f.partial, nullptr);
  {
    ProfScope ps(5, s);
    u3d_launch_render_fb(d, b, bg, out_color, L, acc, part, loss_out, s, sparse ? d_head_out : nullptr,
                         sparse ? u3d_total_P(d) * (size_t)head->channels : 0, sparse);   // + partial reduce + loss reduce
  }
  return finish(desc, s);
}

int u3d_render_loss_step_backward(const u3d_raster_desc* desc, const u3d_head_desc* head, const float* head_out,
                                  const float* center, const float* viewmatrix, const float* projmatrix, const float* campos,
                                  const int32_t* radii, const float* dloss, const void* geom, const void* binning, void* fused,
                                  void* backward_scratch, float* d_head_out, void* stream) {
  const u3d_loss_desc any_loss{1, 0.f, 0.f};   // (the loss is not evaluated here; only the head / shape checks apply)
  int rc = check_fused(desc, head, &any_loss);
  if (rc != U3D_OK) return rc;
  const u3d_raster_desc& d = *desc;
  const int NV = d.n_items * d.views_per_item;
  if (NV == 0 || d.P == 0) return U3D_ERR_INVALID_ARGUMENT;
  if (!head_out || !center || !viewmatrix || !projmatrix || !campos || !radii || !geom || !binning || !fused || !backward_scratch ||
      !d_head_out)
    return U3D_ERR_INVALID_ARGUMENT;
  hipStream_t s = (hipStream_t)stream;
  U3DBuffers b{};
  u3d_carve(d, (void*)geom, (void*)binning, nullptr, &b);
  U3DFused f{};
  u3d_carve_fused(d, fused, &f);
  double* acc = (double*)backward_scratch;
  const U3DSource src = head_source(d, *head, head_out, center, f.qnorm);
  U3DGradSink sink{};
  sink.means = d_head_out; sink.opac = d_head_out + 3; sink.scales = d_head_out + 4; sink.rots = d_head_out + 7;
  sink.shs = d_head_out + 11; sink.colors = nullptr; sink.cov = nullptr; sink.means2D = nullptr; sink.qdot = f.qdot;
  {
    ProfScope ps(4, s);
    u3d_launch_preprocess_bwd(d, b, src, viewmatrix, projmatrix, campos, radii, acc, sink, s, acc, dloss,
                              u3d_sparse_bwd(d, head->mode));   // reads, then re-zeroes, the touched accumulators
  }
  if (head->mode == 1) u3d_launch_quat_fixup(d, head_out + 7, head->channels, f.qnorm, f.qdot, d_head_out + 7, s);
  return finish(desc, s);
}

int u3d_render_loss_step(const u3d_raster_desc* desc, const u3d_head_desc* head, const u3d_loss_desc* loss, const float* bg,
                         const float* head_out, const float* center, const float* viewmatrix, const float* projmatrix,
                         const float* campos, const float* gt, float* out_color, int32_t* radii, float* loss_out,
                         float* d_head_out, void* geom, void* binning, void* fused, void* backward_scratch, void* stream) {
  if (!d_head_out) return U3D_ERR_INVALID_ARGUMENT;
  const int rc = u3d_render_loss_step_forward(desc, head, loss, bg, head_out, center, viewmatrix, projmatrix, campos, gt, out_color, radii,
                                              loss_out, geom, binning, fused, backward_scratch, d_head_out, stream);
  if (rc != U3D_OK) return rc;
  return u3d_render_loss_step_backward(desc, head, head_out, center, viewmatrix, projmatrix, campos, radii, nullptr, geom, binning, fused,
                                       backward_scratch, d_head_out, stream);
}

int u3d_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, void* stream) {
  (void)projmatrix;
  if (P < 0) return U3D_ERR_INVALID_ARGUMENT;
  if (P == 0) return U3D_OK;
  if (!means3D || !viewmatrix || !present) return U3D_ERR_INVALID_ARGUMENT;
  u3d_launch_mark_visible(P, means3D, viewmatrix, present, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? U3D_OK : U3D_ERR_LAUNCH;
}

int u3d_profile_begin(int32_t max_records) {
  if (g_prof_on || max_records == 0) return U3D_ERR_INVALID_ARGUMENT;
  // negative max_records: -(stride << 26 | mask << 20 | records) selects a subset of kinds and samples every stride-th launch (each recorded scope costs two event
  // records on the stream, ~4-5 us of GPU idle; the timed region of bench.py records the dominant kernel only)
  g_prof_mask = 0xffffffffu;
  g_prof_stride = 1;
  for (auto& c : g_prof_seen) c = 0;
  if (max_records < 0) {
    const unsigned v = (unsigned)(-max_records);
    g_prof_mask = (v >> 20) & 0x3fu;
    g_prof_stride = ((v >> 26) & 0xfu) ? ((v >> 26) & 0xfu) : 1u;   // sample every stride-th launch of a kind
    max_records = (int32_t)(v & 0xfffffu);
    if (max_records == 0) return U3D_ERR_INVALID_ARGUMENT;
  }
  g_prof.resize((size_t)max_records);
  for (auto& r : g_prof) {
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return U3D_ERR_NO_DEVICE;
  }
  g_prof_used = 0;
  g_prof_on = true;
  return U3D_OK;
}

int u3d_profile_end(float* ms, int32_t* count) {
  if (!g_prof_on || !ms || !count) return U3D_ERR_INVALID_ARGUMENT;
  g_prof_on = false;
  for (int k = 0; k < U3D_PROFILE_KINDS; ++k) { ms[k] = 0.f; count[k] = 0; }
  int rc = U3D_OK;
  for (size_t i = 0; i < g_prof_used; ++i) {
    float t = 0.f;
    if (hipEventSynchronize(g_prof[i].b) != hipSuccess || hipEventElapsedTime(&t, g_prof[i].a, g_prof[i].b) != hipSuccess) {
      rc = U3D_ERR_LAUNCH;
      continue;
    }
    ms[g_prof[i].kind] += t;
    count[g_prof[i].kind] += 1;
  }
  for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof_used = 0;
  return rc;
}

}  // extern "C"
