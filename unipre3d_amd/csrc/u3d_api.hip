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
std::vector<ProfRec> g_prof;   // pre-created events
size_t g_prof_used = 0;

struct ProfScope {
  ProfRec* r = nullptr;
  hipStream_t s;
  ProfScope(int kind, hipStream_t st) : s(st) {
    if (g_prof_on && g_prof_used < g_prof.size()) {
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
  if (d->n_items < 0 || d->views_per_item < 0 || d->P < 0) return U3D_ERR_INVALID_ARGUMENT;
  if (d->image_height <= 0 || d->image_width <= 0) return U3D_ERR_INVALID_ARGUMENT;
  if (d->image_height > 65535 * U3D_TILE || d->image_width > 65535 * U3D_TILE) return U3D_ERR_UNSUPPORTED;
  if (d->sh_degree < 0 || d->sh_degree > 3) return U3D_ERR_UNSUPPORTED;
  if (!(d->tanfovx > 0.f) || !(d->tanfovy > 0.f)) return U3D_ERR_INVALID_ARGUMENT;
  return U3D_OK;
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
  U3DBuffers b{};
  u3d_carve(d, geom, binning, image, &b);
  (void)hipMemsetAsync(b.num_rendered, 0, sizeof(uint32_t) * NV, s);
  (void)hipMemsetAsync(b.n_vis, 0, sizeof(uint32_t) * NV, s);
  if (d.P > 0) {
    {
      ProfScope ps(0, s);
      u3d_launch_preprocess_fwd(d, b, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                viewmatrix, projmatrix, campos, radii, s);
    }
    {
      ProfScope ps(1, s);
      u3d_launch_depth_sort(d, b, radii, s);
    }
  }
  {
    ProfScope ps(2, s);
    u3d_launch_render_fwd(d, b, bg, out_color, out_invdepth, s);
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
  float* acc = (float*)backward_scratch;
  (void)hipMemsetAsync(acc, 0, L.backward_bytes, s);
  {
    ProfScope ps(3, s);
    u3d_launch_render_bwd(d, b, bg, dL_dcolor, dL_dinvdepth, acc, s);
  }
  {
    ProfScope ps(4, s);
    u3d_launch_preprocess_bwd(d, b, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                              projmatrix, campos, radii, acc, dL_dmeans3D, dL_dmeans2D, shs ? dL_dshs : nullptr, dL_dcolors,
                              dL_dopacity, scales ? dL_dscales : nullptr, scales ? dL_drotations : nullptr, dL_dcov3D, s);
  }
  return finish(desc, s);
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
  if (g_prof_on || max_records <= 0) return U3D_ERR_INVALID_ARGUMENT;
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
