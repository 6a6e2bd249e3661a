// torch binding of the drop-in operator: `_C.rasterize_gaussians` as a C++ autograd function over the C-ABI of
// libunipre3d_rasterizer.so (include/unipre3d_rasterizer.h).  It is what the third-party package's own `_C` extension is to
// its Python wrapper: the reference calls the operator once per object and view
// (train_network.py:418-446 -> gaussian_renderer/__init__.py:89-97), 128 forward + 128 backward calls per C2 step, so the
// per-call host cost IS the cost of that route.  This file holds no arithmetic: it validates, allocates outputs / scratch
// with torch's caching allocator, takes torch's current HIP stream and calls the extern "C" entry points.
//
// Built by unipre3d_amd/csrc/Makefile with g++ against the installed torch headers (no HIP device code here).
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include <array>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <unordered_map>

#include "unipre3d_rasterizer.h"

#ifndef U3D_BINDING_NO_SPARSE
#define U3D_BINDING_NO_SPARSE 0   /* experiments: 1 = never ask for U3D_FLAG_SPARSE_BWD */
#endif
constexpr int64_t kSparseMinP = U3D_SPARSE_BWD_MIN_P;   // the library honours U3D_FLAG_SPARSE_BWD above its LDS-sort limit (static_assert in u3d_common.h)

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

struct Plan {
  u3d_raster_desc d;
  u3d_scratch_sizes s;
  size_t o_binning, o_image, fwd_scratch;   // offsets inside the forward arena (256-byte aligned)
};

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

// descriptors are small PODs: cache the scratch sizes per distinct descriptor (the per-view route reuses ONE shape all step long).
// Returned BY VALUE and stored by value in the autograd node (`plan_save` / `plan_load`), so the cache can be bounded: ragged
// scene-level batches bring a new total_P almost every step.
constexpr size_t kPlanCacheMax = 64;
std::string desc_key(const u3d_raster_desc& d) {
  u3d_raster_desc k = d;
  k.item_offsets = d.item_offsets ? (const int32_t*)8 : nullptr;   // (a plan depends on whether it is set, not on where it points)
  return std::string(reinterpret_cast<const char*>(&k), sizeof(k));
}
Plan plan_for(const u3d_raster_desc& d) {
  static std::mutex mu;
  static auto* cache = new std::unordered_map<std::string, Plan>();   // (leaked on purpose: no destructor at interpreter exit)
  std::string key = desc_key(d);
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache->find(key);
  if (it != cache->end()) return it->second;
  Plan p{};
  p.d = d;
  const int rc = u3d_scratch_query(&p.d, &p.s);
  TORCH_CHECK(rc == U3D_OK, "u3d_scratch_query failed: ", u3d_error_string(rc));
  p.o_binning = align256(p.s.geom_bytes);
  p.o_image = p.o_binning + align256(p.s.binning_bytes);
  p.fwd_scratch = p.o_image + align256(p.s.image_bytes);
  if (cache->size() >= kPlanCacheMax) cache->clear();
  cache->emplace(std::move(key), p);
  return p;
}
inline std::string plan_save(const Plan& p) { return std::string(reinterpret_cast<const char*>(&p), sizeof(Plan)); }
inline Plan plan_load(const std::string& bytes) {
  Plan p{};
  TORCH_CHECK(bytes.size() == sizeof(Plan), "corrupt plan record");
  std::memcpy(&p, bytes.data(), sizeof(Plan));
  return p;
}

// Backward scratch: ONE grow-only buffer per (device, stream).  A call that completes leaves its gradient accumulators zero, so
// the next call on the same stream WITH THE SAME DESCRIPTOR SHAPE (another shape carves the buffer differently) is told not to
// clear them again (U3D_FLAG_ACC_CLEAN) -- no allocation and one launch less per call.  Ragged batches, whose total changes
// every step, share the buffer without the promise; nothing is keyed on a tensor's address and nothing accumulates.
// The fused step holds its lease from the autograd forward to the autograd backward: a second forward in between (or a node
// dropped without backward) finds the lease outstanding and takes a fresh buffer, so a pending backward never loses its data.
struct Workspace { Tensor buf; std::string clean_key; bool outstanding = false; uint64_t ticket = 0; };
struct Lease { Tensor buf; bool clean; uint64_t ticket; };
using WsKey = std::pair<int, void*>;   // device, stream
std::mutex g_ws_mu;
auto* g_ws = new std::map<WsKey, Workspace>();   // (heap, never destroyed: HIP tensors must not be freed during static destruction)
uint64_t g_ws_ticket = 0;
Lease workspace_acquire(const WsKey& key, const Plan& plan, const std::string& shape_key, const at::TensorOptions& byte_opts) {
  std::lock_guard<std::mutex> lock(g_ws_mu);
  Workspace& ws = (*g_ws)[key];
  const int64_t need = (int64_t)plan.s.backward_bytes;
  if (ws.outstanding || !ws.buf.defined() || ws.buf.numel() < need) {
    ws.buf = at::empty({std::max<int64_t>(need, ws.buf.defined() && !ws.outstanding ? ws.buf.numel() : 0)}, byte_opts);
    ws.clean_key.clear();
  }
  const bool clean = !ws.clean_key.empty() && ws.clean_key == shape_key;
  ws.clean_key.clear();          // the promise is withdrawn while a call is in flight on the host: a failed call leaves none behind
  ws.outstanding = true;
  ws.ticket = ++g_ws_ticket;
  return {ws.buf, clean, ws.ticket};
}
// the call (or, for the fused step, its backward half) succeeded: its accumulators are zero again.  zeroed = false: the lease
// ends without that promise (the backward half was skipped).
void workspace_release(const WsKey& key, uint64_t ticket, const std::string& shape_key, bool zeroed = true) {
  std::lock_guard<std::mutex> lock(g_ws_mu);
  auto it = g_ws->find(key);
  if (it == g_ws->end() || it->second.ticket != ticket || !it->second.outstanding) return;
  it->second.outstanding = false;
  if (zeroed) it->second.clean_key = shape_key;
}

inline const float* fptr(const Tensor& t) { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; }
inline float* fptr_mut(Tensor& t) { return (t.defined() && t.numel() > 0) ? t.data_ptr<float>() : nullptr; }

inline Tensor f32c(const Tensor& t, const c10::Device& dev) {
  if (!t.defined()) return t;
  if (t.device() == dev && t.scalar_type() == at::kFloat && t.is_contiguous()) return t;
  return t.to(dev, at::kFloat).contiguous();
}

inline void* current_stream(const c10::Device& dev) {
  // the kernels are enqueued on the calling thread's current HIP device: refuse tensors that live elsewhere
  const auto cur = c10::hip::current_device();
  TORCH_CHECK(dev.index() < 0 || dev.index() == cur, "tensors live on cuda:", (int)dev.index(), " but the current device is cuda:", (int)cur,
              "; call under torch.cuda.device(...) (one process per GPU sets it once)");
  return (void*)c10::hip::getCurrentHIPStream().stream();
}

struct RasterizeFn : public torch::autograd::Function<RasterizeFn> {
  // (optional inputs travel as c10::optional: the C++ autograd function machinery records device / layout of every plain Tensor
  // argument and refuses undefined ones)
  using OptTensor = c10::optional<Tensor>;
  static variable_list forward(AutogradContext* ctx, Tensor means3D, OptTensor means2D_, OptTensor shs_, OptTensor colors_, Tensor opac,
                               OptTensor scales_, OptTensor rots_, OptTensor cov_, Tensor view, Tensor proj, Tensor campos, Tensor bg,
                               int64_t n_items, int64_t vpi, int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                               int64_t sh_degree, int64_t flags, bool single, OptTensor item_offsets_, int64_t max_P) {
    // ragged batch (item_offsets given): per-Gaussian tensors are PACKED (total_P, ...), max_P is the largest set; the per-(view,
    // Gaussian) outputs radii / dL_dmeans2D are packed (views_per_item * total_P, ...) in the layout of u3d_raster_desc
    // single: ONE view of ONE set with the reference operator's own tensor shapes (means3D (P,3) ... -> color (3,H,W)); no
    // leading set / view dimension, so no unsqueeze / squeeze nodes surround the call in the autograd graph
    const c10::Device dev = means3D.device();
    TORCH_CHECK(dev.is_cuda(), "the MI355X rasterizer needs tensors on a HIP device; there is no CPU fallback "
                               "(the CPU restatement lives in oracle/ and is test infrastructure only)");
    auto val = [](const OptTensor& t) { return t.has_value() ? *t : Tensor(); };
    const Tensor shs = val(shs_), colors = val(colors_), scales = val(scales_), rots = val(rots_), cov = val(cov_);
    // (means2D carries no data forward: it is the gradient sink `viewspace_points` of gaussian_renderer/__init__.py:29)
    const bool ragged = item_offsets_.has_value() && item_offsets_->defined();
    const int64_t total_P = ragged ? means3D.size(0) : 0;
    const int64_t P = ragged ? max_P : (means3D.numel() > 0 ? means3D.size(-2) : 0);
    const int64_t M = shs.defined() ? shs.size(-2) : 0;
    Tensor offsets;
    if (ragged) {
      offsets = *item_offsets_;
      TORCH_CHECK(offsets.device() == dev && offsets.scalar_type() == at::kInt && offsets.is_contiguous() && offsets.numel() == n_items + 1,
                  "item_offsets must be a contiguous int32 tensor of n_items + 1 prefix sums on the Gaussians' device");
      TORCH_CHECK(!single && total_P > 0 && max_P > 0 && max_P <= total_P, "ragged batch: bad total / largest set size");
    }
    {
      // the C-ABI trusts its sizes: refuse tensors whose shapes do not add up (upstream's extension raises on means3D not (P, 3))
      const int64_t G = ragged ? total_P : (single ? P : n_items * P);     // Gaussians in the call
      const int64_t NVc = n_items * vpi;
      TORCH_CHECK(means3D.size(-1) == 3 && means3D.numel() == G * 3, "means3D must be (", single ? "" : "sets, ", "P, 3)");
      TORCH_CHECK(opac.numel() == G, "opacities must hold one value per Gaussian");
      TORCH_CHECK(!shs.defined() || (shs.size(-1) == 3 && shs.numel() == G * M * 3), "shs must be (..., P, M, 3)");
      TORCH_CHECK(!colors.defined() || colors.numel() == G * 3, "colors_precomp must be (..., P, 3)");
      TORCH_CHECK(!scales.defined() || scales.numel() == G * 3, "scales must be (..., P, 3)");
      TORCH_CHECK(!rots.defined() || rots.numel() == G * 4, "rotations must be (..., P, 4)");
      TORCH_CHECK(!cov.defined() || cov.numel() == G * 6, "cov3D_precomp must be (..., P, 6)");
      TORCH_CHECK(view.numel() == NVc * 16 && proj.numel() == NVc * 16 && campos.numel() == NVc * 3 && bg.numel() == 3,
                  "cameras must be (views, 4, 4), (views, 4, 4), (views, 3) and bg (3,)");
      if (means2D_.has_value() && means2D_->defined())
        TORCH_CHECK(means2D_->numel() == (ragged ? vpi * total_P : (single ? P : NVc * P)) * 3, "means2D must be (views, P, 3)");
    }
    u3d_raster_desc d{};
    d.n_items = (int32_t)n_items; d.views_per_item = (int32_t)vpi; d.P = (int32_t)P;
    d.image_height = (int32_t)H; d.image_width = (int32_t)W;
    d.tanfovx = (float)tanfovx; d.tanfovy = (float)tanfovy; d.scale_modifier = (float)scale_modifier;
    d.sh_degree = (int32_t)sh_degree; d.sh_coeffs = (int32_t)M; d.flags = (int32_t)flags;
    d.total_P = (int32_t)total_P;
    d.item_offsets = ragged ? offsets.data_ptr<int32_t>() : nullptr;
    const Plan plan = plan_for(d);
    u3d_raster_desc dd = d;                                    // this call's descriptor: the plan's shape + the device pointer
    const int64_t NV = n_items * vpi;
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    Tensor color = single ? at::empty({3, H, W}, fopt) : at::empty({NV, 3, H, W}, fopt);
    Tensor invdepth = single ? at::empty({1, H, W}, fopt) : at::empty({NV, 1, H, W}, fopt);
    // (every (view, Gaussian) radius is written by the projection kernel; a ragged batch starts from zeros so that pairs a
    // malformed prefix-sum table leaves unprojected read as culled instead of as uninitialised memory)
    Tensor radii = single ? at::empty({P}, fopt.dtype(at::kInt))
                          : (ragged ? at::zeros({vpi * total_P}, fopt.dtype(at::kInt)) : at::empty({NV, P}, fopt.dtype(at::kInt)));
    Tensor arena = at::empty({(int64_t)plan.fwd_scratch}, fopt.dtype(at::kByte));   // geom | binning | image
    char* base = (char*)arena.data_ptr();
    const int rc = u3d_rasterize_forward(&dd, fptr(bg), fptr(means3D), fptr(shs), fptr(colors), fptr(opac), fptr(scales), fptr(rots),
                                         fptr(cov), fptr(view), fptr(proj), fptr(campos), color.data_ptr<float>(), invdepth.data_ptr<float>(),
                                         P > 0 ? radii.data_ptr<int32_t>() : nullptr, base, base + plan.o_binning, base + plan.o_image,
                                         current_stream(dev));
    TORCH_CHECK(rc == U3D_OK, "u3d_rasterize_forward failed: ", u3d_error_string(rc), " (code ", rc, ")");
    ctx->saved_data["plan"] = plan_save(plan);
    ctx->saved_data["has_colors"] = colors.defined();
    ctx->saved_data["single"] = single;
    ctx->saved_data["has_means2D"] = means2D_.has_value() && means2D_->defined();
    ctx->save_for_backward({means3D, shs, colors, opac, scales, rots, cov, view, proj, campos, bg, radii, arena, offsets});
    ctx->mark_non_differentiable({radii});
    ctx->set_materialize_grads(false);    // unused outputs (invdepth) arrive undefined, not as a zero tensor
    return {color, radii, invdepth};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const Plan plan = plan_load(ctx->saved_data["plan"].toStringRef());
    const bool has_colors = ctx->saved_data["has_colors"].toBool();
    const bool single = ctx->saved_data["single"].toBool();
    const bool has_m2d = ctx->saved_data["has_means2D"].toBool();
    auto sv = ctx->get_saved_variables();
    const Tensor &means3D = sv[0], &shs = sv[1], &colors = sv[2], &opac = sv[3], &scales = sv[4], &rots = sv[5], &cov = sv[6], &view = sv[7],
                 &proj = sv[8], &campos = sv[9], &bg = sv[10], &radii = sv[11], &arena = sv[12], &offsets = sv[13];
    const bool ragged = offsets.defined();
    const u3d_raster_desc& d = plan.d;
    const c10::Device dev = means3D.device();
    const int64_t NV = (int64_t)d.n_items * d.views_per_item, P = d.P, M = d.sh_coeffs, n = d.n_items;
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    Tensor gcol = grads[0], ginv = grads[2];
    if (gcol.defined()) gcol = f32c(gcol, dev);
    else gcol = at::zeros({NV * 3, d.image_height, d.image_width}, fopt);     // only the inverse-depth output was used downstream
    if (ginv.defined()) ginv = f32c(ginv, dev);
    const bool live = P > 0 && NV > 0;
    // the backward kernels write every element of every gradient they are handed (zeros for culled / untouched Gaussians)
    // gradient of a per-Gaussian input (per_view false) or of the per-(view, Gaussian) sink means2D (true), trailing dims `tail`:
    //   uniform (sets | views, P, tail...)   single (P, tail...)   ragged, packed (total_P | views_per_item * total_P, tail...)
    auto out = [&](bool per_view, std::initializer_list<int64_t> tail) {
      std::vector<int64_t> sh;
      if (ragged) sh.push_back(per_view ? (int64_t)d.views_per_item * d.total_P : (int64_t)d.total_P);
      else {
        if (!single) sh.push_back(per_view ? NV : n);
        sh.push_back(P);
      }
      sh.insert(sh.end(), tail.begin(), tail.end());
      return live ? at::empty(sh, fopt) : at::zeros(sh, fopt);
    };
    Tensor g_means3D = out(false, {3}), g_means2D = has_m2d ? out(true, {3}) : Tensor(), g_op = out(false, {1});
    Tensor g_shs = shs.defined() ? out(false, {M, 3}) : Tensor();
    Tensor g_col = has_colors ? out(false, {3}) : Tensor();
    Tensor g_scales = scales.defined() ? out(false, {3}) : Tensor();
    Tensor g_rots = scales.defined() ? out(false, {4}) : Tensor();
    Tensor g_cov = cov.defined() ? out(false, {6}) : Tensor();
    if (live) {
      void* stream = current_stream(dev);
      const WsKey key{(int)dev.index(), stream};
      const std::string shape_key = desc_key(plan.d);
      const Lease lease = workspace_acquire(key, plan, shape_key, fopt.dtype(at::kByte));
      const Tensor& scratch = lease.buf;
      u3d_raster_desc dd = plan.d;
      dd.item_offsets = ragged ? offsets.data_ptr<int32_t>() : nullptr;
      if (lease.clean) dd.flags |= U3D_FLAG_ACC_CLEAN;
      const char* base = (const char*)arena.data_ptr();
      const int rc = u3d_rasterize_backward(&dd, fptr(bg), fptr(means3D), fptr(shs), fptr(colors), fptr(opac), fptr(scales), fptr(rots),
                                            fptr(cov), fptr(view), fptr(proj), fptr(campos), radii.data_ptr<int32_t>(), fptr(gcol), fptr(ginv),
                                            base, base + plan.o_binning, base + plan.o_image, scratch.data_ptr(), fptr_mut(g_means3D),
                                            fptr_mut(g_means2D), fptr_mut(g_shs), fptr_mut(g_col), fptr_mut(g_op), fptr_mut(g_scales),
                                            fptr_mut(g_rots), fptr_mut(g_cov), stream);
      if (rc != U3D_OK) workspace_release(key, lease.ticket, shape_key, false);
      TORCH_CHECK(rc == U3D_OK, "u3d_rasterize_backward failed: ", u3d_error_string(rc), " (code ", rc, ")");
      workspace_release(key, lease.ticket, shape_key);
    }
    return {g_means3D, g_means2D, g_shs, g_col, g_op, g_scales, g_rots, g_cov, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(),
            Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// ---- the per-view wrapper's body as one call (u3d_render_view_*): `render_predicted` of gaussian_renderer/__init__.py:13-104 ------
// The reference calls the wrapper once per object and view; each PyTorch-level op it contains costs the host ~8-10 us and the GPU a
// dispatch slot, which -- not bytes -- is what that route is bound by (profiles/r04: 8 % GPU busy).  This function is the whole body:
//   * `screenspace_points = zeros_like(xyz, requires_grad=True) + 0` (a fill, an add and an AddBackward node per view): a fresh LEAF
//     that shares one cached all-zero storage per (device, P) -- same value, same role (its .grad receives dL/dmean2D), no launch;
//     the zeros are shared, so they must not be modified in place (the reference never does);
//   * `torch.cat([features_dc, features_rest], dim=1)` and its backward: the kernels read / write SH through two pointers;
//   * `radii > 0`: written by the projection kernel;
//   * the inverse-depth plane the wrapper drops is not produced.
std::mutex g_sink_mu;
auto* g_sink = new std::map<std::pair<int, int64_t>, Tensor>();   // (heap, never destroyed)
Tensor viewspace_sink(const Tensor& xyz) {
  Tensor z;
  {
    std::lock_guard<std::mutex> lock(g_sink_mu);
    const std::pair<int, int64_t> key{(int)xyz.device().index(), xyz.numel()};
    auto it = g_sink->find(key);
    if (it == g_sink->end() || it->second.is_inference()) {   // (an inference tensor can never become a requires_grad leaf: rebuilt)
      if (g_sink->size() >= 64) g_sink->clear();
      c10::InferenceMode normal(false);   // the first call may come from an eval pass under torch.inference_mode()
      it = g_sink->insert_or_assign(key, at::zeros({xyz.numel()}, xyz.options().dtype(at::kFloat))).first;
    }
    z = it->second;
  }
  c10::InferenceMode normal(false);
  Tensor leaf = z.view(xyz.sizes()).detach();
  leaf.set_requires_grad(true);
  return leaf;
}

struct RenderViewFn : public torch::autograd::Function<RenderViewFn> {
  using OptTensor = c10::optional<Tensor>;
  static variable_list forward(AutogradContext* ctx, Tensor xyz, Tensor sink, Tensor dc, OptTensor rest_, Tensor opac, Tensor scales,
                               Tensor rots, Tensor view, Tensor proj, Tensor campos, Tensor bg, int64_t H, int64_t W, double tanfovx,
                               double tanfovy, double scale_modifier, int64_t sh_degree, int64_t flags) {
    const c10::Device dev = xyz.device();
    TORCH_CHECK(dev.is_cuda(), "the MI355X rasterizer needs tensors on a HIP device; there is no CPU fallback");
    const Tensor rest = rest_.has_value() ? *rest_ : Tensor();
    const int64_t P = xyz.numel() / 3;
    const int64_t M = 1 + (rest.defined() && P > 0 ? rest.numel() / (3 * P) : 0);
    TORCH_CHECK(xyz.dim() == 2 && xyz.size(1) == 3, "xyz must be (P, 3)");
    TORCH_CHECK(dc.numel() == P * 3 && opac.numel() == P && scales.numel() == P * 3 && rots.numel() == P * 4 && sink.numel() == P * 3 &&
                    (!rest.defined() || rest.numel() == P * (M - 1) * 3),
                "features_dc (P,1,3), features_rest (P,M-1,3), opacity (P,1), scaling (P,3), rotation (P,4) must match xyz (P,3)");
    TORCH_CHECK(view.numel() == 16 && proj.numel() == 16 && campos.numel() == 3 && bg.numel() == 3, "cameras must be (4,4), (4,4), (3,) and bg (3,)");
    u3d_raster_desc d{};
    d.n_items = 1; d.views_per_item = 1; d.P = (int32_t)P; d.image_height = (int32_t)H; d.image_width = (int32_t)W;
    d.tanfovx = (float)tanfovx; d.tanfovy = (float)tanfovy; d.scale_modifier = (float)scale_modifier;
    d.sh_degree = (int32_t)sh_degree; d.sh_coeffs = (int32_t)M; d.flags = (int32_t)flags;
    const Plan plan = plan_for(d);
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    Tensor color = at::empty({3, H, W}, fopt);
    Tensor radii = at::empty({P}, fopt.dtype(at::kInt));
    Tensor visible = at::empty({P}, fopt.dtype(at::kBool));
    Tensor arena = at::empty({(int64_t)plan.fwd_scratch}, fopt.dtype(at::kByte));
    char* base = (char*)arena.data_ptr();
    u3d_raster_desc dd = d;
    const int rc = u3d_render_view_forward(&dd, fptr(bg), fptr(xyz), fptr(dc), M > 1 ? fptr(rest) : nullptr, fptr(opac), fptr(scales), fptr(rots),
                                           fptr(view), fptr(proj), fptr(campos), color.data_ptr<float>(), P > 0 ? radii.data_ptr<int32_t>() : nullptr,
                                           P > 0 ? (uint8_t*)visible.data_ptr() : nullptr, base, base + plan.o_binning, base + plan.o_image,
                                           current_stream(dev));
    TORCH_CHECK(rc == U3D_OK, "u3d_render_view_forward failed: ", u3d_error_string(rc), " (code ", rc, ")");
    ctx->saved_data["plan"] = plan_save(plan);
    ctx->save_for_backward({xyz, dc, rest, opac, scales, rots, view, proj, campos, bg, radii, arena});
    ctx->mark_non_differentiable({radii, visible});
    ctx->set_materialize_grads(false);
    return {color, radii, visible};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const Plan plan = plan_load(ctx->saved_data["plan"].toStringRef());
    auto sv = ctx->get_saved_variables();
    const Tensor &xyz = sv[0], &dc = sv[1], &rest = sv[2], &opac = sv[3], &scales = sv[4], &rots = sv[5], &view = sv[6], &proj = sv[7],
                 &campos = sv[8], &bg = sv[9], &radii = sv[10], &arena = sv[11];
    const u3d_raster_desc& d = plan.d;
    const c10::Device dev = xyz.device();
    const int64_t P = d.P, M = d.sh_coeffs;
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    variable_list out(18);
    if (!grads[0].defined() || P == 0) return out;      // the image was not used downstream: no gradient flows
    const Tensor gcol = f32c(grads[0], dev);
    Tensor g_xyz = at::empty({P, 3}, fopt), g_sink = at::empty({P, 3}, fopt), g_dc = at::empty({P, 1, 3}, fopt);
    Tensor g_rest = M > 1 ? at::empty({P, M - 1, 3}, fopt) : Tensor();
    Tensor g_op = at::empty({P, 1}, fopt), g_scales = at::empty({P, 3}, fopt), g_rots = at::empty({P, 4}, fopt);
    void* stream = current_stream(dev);
    const WsKey key{(int)dev.index(), stream};
    const std::string shape_key = desc_key(plan.d);
    const Lease lease = workspace_acquire(key, plan, shape_key, fopt.dtype(at::kByte));
    u3d_raster_desc dd = plan.d;
    if (lease.clean) dd.flags |= U3D_FLAG_ACC_CLEAN;
    const char* base = (const char*)arena.data_ptr();
    const int rc = u3d_render_view_backward(&dd, fptr(bg), fptr(xyz), fptr(dc), M > 1 ? fptr(rest) : nullptr, fptr(opac), fptr(scales), fptr(rots),
                                            fptr(view), fptr(proj), fptr(campos), radii.data_ptr<int32_t>(), fptr(gcol), base,
                                            base + plan.o_binning, base + plan.o_image, lease.buf.data_ptr(), fptr_mut(g_xyz), fptr_mut(g_sink),
                                            fptr_mut(g_dc), M > 1 ? fptr_mut(g_rest) : nullptr, fptr_mut(g_op), fptr_mut(g_scales),
                                            fptr_mut(g_rots), stream);
    workspace_release(key, lease.ticket, shape_key, rc == U3D_OK);
    TORCH_CHECK(rc == U3D_OK, "u3d_render_view_backward failed: ", u3d_error_string(rc), " (code ", rc, ")");
    out[0] = g_xyz; out[1] = g_sink; out[2] = g_dc; out[3] = g_rest; out[4] = g_op.view(opac.sizes()); out[5] = g_scales; out[6] = g_rots;
    return out;
  }
};

// (color (3,H,W), viewspace_points (P,3) leaf, radii (P,) int32, visibility_filter (P,) bool)
std::tuple<Tensor, Tensor, Tensor, Tensor> render_view(const Tensor& xyz, const Tensor& opacity, const Tensor& scaling, const Tensor& rotation,
                                                       const Tensor& features_dc, const c10::optional<Tensor>& features_rest, const Tensor& view,
                                                       const Tensor& proj, const Tensor& campos, const Tensor& bg, int64_t H, int64_t W,
                                                       double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree, int64_t flags) {
  const c10::Device dev = xyz.device();
  const Tensor x = f32c(xyz, dev);
  Tensor sink = viewspace_sink(x);
  c10::optional<Tensor> rest;
  if (features_rest.has_value() && features_rest->defined() && features_rest->numel() > 0) rest = f32c(*features_rest, dev);
  auto r = RenderViewFn::apply(x, sink, f32c(features_dc, dev), rest, f32c(opacity, dev), f32c(scaling, dev), f32c(rotation, dev), f32c(view, dev),
                               f32c(proj, dev), f32c(campos, dev), f32c(bg, dev), H, W, tanfovx, tanfovy, scale_modifier, sh_degree, flags);
  return {r[0], sink, r[1], r[2]};
}

// ---- fused render-loss training step (u3d_render_loss_step): activations + render + loss + backward in one launch sequence ------
// One call per training step, but its host cost is exposed whenever a step's kernels are short (C1, C3: 7 launches in ~0.1 ms).
std::mutex g_unit_mu;
auto* g_unit = new std::map<int, Tensor>();   // (heap, never destroyed)
Tensor unit_tensor(int64_t device_index) {   // the cached dL/dloss = 1 of fused.backward_unit(): recognised by its storage in backward
  std::lock_guard<std::mutex> lock(g_unit_mu);
  auto it = g_unit->find((int)device_index);
  if (it == g_unit->end())
    it = g_unit->emplace((int)device_index, at::ones({}, at::TensorOptions().dtype(at::kFloat).device(c10::Device(c10::kCUDA, (c10::DeviceIndex)device_index)))).first;
  return it->second;
}

struct RenderLossStepFn : public torch::autograd::Function<RenderLossStepFn> {
  // forward  = u3d_render_loss_step_forward : projection [+ sort] -> single-pass tile kernel -> fixed-order reduce  => loss + accumulators
  // backward = u3d_render_loss_step_backward: chain rule accumulators -> d(head_out), scaled IN the kernel by autograd's grad_output
  // (a device scalar): a plain `loss.backward()` (train_network.py:333) launches the same kernels as the one-call C entry point,
  // with no d_head * g multiply; fused.backward_unit() additionally spares autograd's ones_like fill.
  using OptTensor = c10::optional<Tensor>;
  static variable_list forward(AutogradContext* ctx, Tensor head_out, Tensor center, Tensor view, Tensor proj, Tensor campos, Tensor gt,
                               Tensor bg, int64_t H, int64_t W, double tanfov, int64_t mode, double offset_scale, int64_t sh_degree,
                               int64_t loss_kind, double non_bg_rate, double bg_rate, double scale_modifier, int64_t flags, bool want_color,
                               bool isotropic, OptTensor item_offsets_, int64_t max_P) {
    const c10::Device dev = head_out.device();
    TORCH_CHECK(dev.is_cuda(), "the MI355X rasterizer needs tensors on a HIP device; there is no CPU fallback");
    const bool ragged = item_offsets_.has_value() && item_offsets_->defined();
    Tensor offsets;
    int64_t B, P, C, total_P = 0;
    if (ragged) {
      offsets = *item_offsets_;
      TORCH_CHECK(head_out.dim() == 2, "ragged batch: head_out must be packed (sum P_i, C)");
      TORCH_CHECK(offsets.device() == dev && offsets.scalar_type() == at::kInt && offsets.is_contiguous(),
                  "item_offsets must be a contiguous int32 tensor on the Gaussians' device");
      B = offsets.numel() - 1; total_P = head_out.size(0); C = head_out.size(1); P = max_P;
      TORCH_CHECK(P > 0 && P <= total_P, "ragged batch: max_P (the largest set) is required");
    } else {
      TORCH_CHECK(head_out.dim() == 3, "head_out must be (B, P, C); pass item_offsets for a packed ragged batch (sum P_i, C)");
      B = head_out.size(0); P = head_out.size(1); C = head_out.size(2);
    }
    const int64_t NV = view.size(0), K = (sh_degree + 1) * (sh_degree + 1);
    TORCH_CHECK(B > 0 && NV % B == 0, NV, " cameras for ", B, " Gaussian sets: every set needs the same number of views");
    TORCH_CHECK(C == 11 + 3 * K, "head output has ", C, " channels, expected ", 11 + 3 * K, " for SH degree ", sh_degree);
    const int64_t V = NV / B;
    {
      const int64_t G = ragged ? total_P : B * P;
      TORCH_CHECK(center.numel() == G * 3, "center must be (..., P, 3)");
      TORCH_CHECK(view.numel() == NV * 16 && proj.numel() == NV * 16 && campos.numel() == NV * 3 && bg.numel() == 3 && gt.numel() == NV * 3 * H * W,
                  "cameras must be (views, 16), (views, 16), (views, 3), bg (3,) and gt (views, 3, H, W)");
    }
    u3d_raster_desc d{};
    d.n_items = (int32_t)B; d.views_per_item = (int32_t)V; d.P = (int32_t)P; d.image_height = (int32_t)H; d.image_width = (int32_t)W;
    d.tanfovx = d.tanfovy = (float)tanfov; d.scale_modifier = (float)scale_modifier; d.sh_degree = (int32_t)sh_degree;
    d.sh_coeffs = (int32_t)K; d.flags = (int32_t)flags; d.total_P = (int32_t)total_P;   // (U3D_FLAG_SPARSE_BWD: decided by render_loss_step below)
    d.item_offsets = ragged ? offsets.data_ptr<int32_t>() : nullptr;
    const Plan plan = plan_for(d);
    u3d_head_desc hd{(int32_t)mode, (int32_t)C, (float)offset_scale, isotropic ? 1 : 0};
    u3d_loss_desc ld{(int32_t)loss_kind, (float)non_bg_rate, (float)bg_rate};
    const auto fopt = at::TensorOptions().dtype(at::kFloat).device(dev);
    Tensor color = want_color ? at::empty({NV, 3, H, W}, fopt) : at::empty({0}, fopt);
    Tensor radii = ragged ? at::zeros({V * total_P}, fopt.dtype(at::kInt)) : at::empty({NV, P}, fopt.dtype(at::kInt));
    Tensor loss = at::empty({}, fopt);
    const size_t o_fused = plan.o_image;                                 // arena: geom | binning | fused  (no image buffer in this path)
    Tensor arena = at::empty({(int64_t)(o_fused + align256(plan.s.fused_bytes))}, fopt.dtype(at::kByte));
    char* base = (char*)arena.data_ptr();
    void* stream = current_stream(dev);
    const WsKey key{(int)dev.index(), stream};
    const std::string shape_key = desc_key(d);
    const Lease lease = workspace_acquire(key, plan, shape_key, fopt.dtype(at::kByte));
    u3d_raster_desc dd = d;
    if (lease.clean) dd.flags |= U3D_FLAG_ACC_CLEAN;
    // scene-level head: the gradient buffer exists before the backward half (U3D_FLAG_SPARSE_BWD, set in `d` by the caller below):
    // the forward half zero-fills it beside its gradient reduction and the backward half visits the touched Gaussians only.  The
    // library honours the flag at scene-level sizes; below them the backward half writes every row itself, as before.
    // (ragged: from zeros, so that rows a malformed prefix-sum table leaves out read as zero gradient whichever route runs)
    Tensor d_head = (d.flags & U3D_FLAG_SPARSE_BWD) ? (ragged ? at::zeros_like(head_out) : at::empty_like(head_out)) : Tensor();
    const int rc = u3d_render_loss_step_forward(&dd, &hd, &ld, fptr(bg), fptr(head_out), fptr(center), fptr(view), fptr(proj), fptr(campos),
                                                fptr(gt), want_color ? color.data_ptr<float>() : nullptr, radii.data_ptr<int32_t>(),
                                                loss.data_ptr<float>(), base, base + plan.o_binning, base + o_fused, lease.buf.data_ptr(),
                                                d_head.defined() ? d_head.data_ptr<float>() : nullptr, stream);
    if (rc != U3D_OK) workspace_release(key, lease.ticket, shape_key, false);
    TORCH_CHECK(rc == U3D_OK, "u3d_render_loss_step_forward failed: ", u3d_error_string(rc), " (code ", rc, ")");
    // (the lease stays outstanding until the backward half has consumed -- and re-zeroed -- the accumulators)
    ctx->saved_data["plan"] = plan_save(plan);
    ctx->saved_data["head"] = std::vector<int64_t>{mode, C, isotropic ? 1 : 0};
    ctx->saved_data["offset_scale"] = offset_scale;
    ctx->saved_data["ticket"] = (int64_t)lease.ticket;
    ctx->saved_data["stream"] = (int64_t)(intptr_t)stream;
    ctx->saved_data["consumed"] = false;
    ctx->saved_data["loss"] = std::vector<double>{(double)loss_kind, non_bg_rate, bg_rate};
    ctx->save_for_backward({head_out, center, view, proj, campos, radii, arena, lease.buf, offsets, gt, bg, d_head});
    ctx->mark_non_differentiable({color, radii});
    ctx->set_materialize_grads(false);
    return {loss, color, radii};
  }

  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const Plan plan = plan_load(ctx->saved_data["plan"].toStringRef());
    const auto hv = ctx->saved_data["head"].toIntVector();
    uint64_t ticket = (uint64_t)ctx->saved_data["ticket"].toInt();
    auto sv = ctx->get_saved_variables();
    const Tensor &head_out = sv[0], &center = sv[1], &view = sv[2], &proj = sv[3], &campos = sv[4], &radii = sv[5], &arena = sv[6],
                 &offsets = sv[8], &gt = sv[9], &bg = sv[10];
    Tensor scratch = sv[7];
    Tensor d_head_pre = sv[11];            // (U3D_FLAG_SPARSE_BWD: zero-filled by the forward half; used by ONE backward)
    const c10::Device dev = head_out.device();
    const WsKey key{(int)dev.index(), (void*)(intptr_t)ctx->saved_data["stream"].toInt()};
    const std::string shape_key = desc_key(plan.d);
    variable_list out(22);
    if (!grads[0].defined()) {                 // the loss was not used downstream: nothing to chain, and the accumulators stay dirty
      workspace_release(key, ticket, shape_key, false);
      // the lease is gone (another forward may reuse the buffer): a later backward through this retained node must not read it,
      // it recomputes the forward half into a fresh lease like any second backward does
      ctx->saved_data["consumed"] = true;
      out[0] = at::zeros_like(head_out);
      return out;
    }
    void* stream = current_stream(dev);
    TORCH_CHECK(stream == key.second, "the fused render-loss step must be backpropagated on the stream its forward ran on");
    u3d_raster_desc dd = plan.d;
    dd.item_offsets = offsets.defined() ? offsets.data_ptr<int32_t>() : nullptr;
    u3d_head_desc hd{(int32_t)hv[0], (int32_t)hv[1], (float)ctx->saved_data["offset_scale"].toDouble(), (int32_t)hv[2]};
    const char* base = (const char*)arena.data_ptr();
    if (ctx->saved_data["consumed"].toBool()) {
      // backward(retain_graph=True) followed by another backward: the first one consumed (and re-zeroed) the accumulators, so the
      // forward half is run again into a fresh lease -- same inputs, same arena, identical results; the rare path pays a recompute.
      // (What the node DOES pin from forward to backward: the forward arena, the leased backward scratch, gt and bg -- see
      // INTEGRATION.md "memory held between forward and backward".)
      const auto lv = ctx->saved_data["loss"].toDoubleVector();
      u3d_loss_desc ld{(int32_t)lv[0], (float)lv[1], (float)lv[2]};
      const Lease lease = workspace_acquire(key, plan, shape_key, at::TensorOptions().dtype(at::kByte).device(dev));
      u3d_raster_desc d2 = dd;
      if (lease.clean) d2.flags |= U3D_FLAG_ACC_CLEAN;
      Tensor loss_again = at::empty({}, at::TensorOptions().dtype(at::kFloat).device(dev));
      if (d_head_pre.defined()) d_head_pre = offsets.defined() ? at::zeros_like(head_out) : at::empty_like(head_out);   // (the first backward handed its buffer to autograd)
      const int rc2 = u3d_render_loss_step_forward(&d2, &hd, &ld, fptr(bg), fptr(head_out), fptr(center), fptr(view), fptr(proj), fptr(campos),
                                                   fptr(gt), nullptr, radii.data_ptr<int32_t>(), loss_again.data_ptr<float>(), (void*)base,
                                                   (void*)(base + plan.o_binning), (void*)(base + plan.o_image), lease.buf.data_ptr(),
                                                   d_head_pre.defined() ? d_head_pre.data_ptr<float>() : nullptr, stream);
      if (rc2 != U3D_OK) workspace_release(key, lease.ticket, shape_key, false);
      TORCH_CHECK(rc2 == U3D_OK, "u3d_render_loss_step_forward (recompute) failed: ", u3d_error_string(rc2), " (code ", rc2, ")");
      scratch = lease.buf;
      ticket = lease.ticket;
    }
    ctx->saved_data["consumed"] = true;
    Tensor unit;
    {
      std::lock_guard<std::mutex> lock(g_unit_mu);
      auto it = g_unit->find((int)dev.index());
      if (it != g_unit->end()) unit = it->second;
    }
    // dL/dloss: a device scalar the projection-backward kernel multiplies in as it reads the accumulators; THE unit tensor of
    // fused.backward_unit() (recognised by its storage) needs no load at all
    Tensor g = grads[0];
    const float* gptr = nullptr;
    if (!(unit.defined() && g.data_ptr() == unit.data_ptr())) {
      g = f32c(g, dev);
      TORCH_CHECK(g.numel() == 1, "gradient of the scalar loss must be a scalar");
      gptr = g.data_ptr<float>();
    }
    // (every row is written by the projection-backward kernel; a ragged batch starts from zeros so that rows a malformed prefix-sum
    // table leaves out read as zero gradient instead of as uninitialised memory)
    Tensor d_head = d_head_pre.defined() ? d_head_pre : (offsets.defined() ? at::zeros_like(head_out) : at::empty_like(head_out));
    const int rc = u3d_render_loss_step_backward(&dd, &hd, fptr(head_out), fptr(center), fptr(view), fptr(proj), fptr(campos),
                                                 radii.data_ptr<int32_t>(), gptr, base, base + plan.o_binning, (void*)(base + plan.o_image),
                                                 scratch.data_ptr(), d_head.data_ptr<float>(), stream);
    workspace_release(key, ticket, shape_key, rc == U3D_OK);
    TORCH_CHECK(rc == U3D_OK, "u3d_render_loss_step_backward failed: ", u3d_error_string(rc), " (code ", rc, ")");
    out[0] = d_head;
    return out;
  }
};

std::tuple<Tensor, Tensor, Tensor> render_loss_step(const Tensor& head_out, const Tensor& center, const Tensor& view, const Tensor& proj,
                                                    const Tensor& campos, const Tensor& gt, const Tensor& bg, int64_t H, int64_t W, double tanfov,
                                                    int64_t mode, double offset_scale, int64_t sh_degree, int64_t loss_kind, double non_bg_rate,
                                                    double bg_rate, double scale_modifier, int64_t flags, bool want_color, bool isotropic,
                                                    const c10::optional<Tensor>& item_offsets, int64_t max_P) {
  const c10::Device dev = head_out.device();
  const int64_t NV = view.numel() / 16;
  // U3D_FLAG_SPARSE_BWD (gradient buffer allocated in forward, touched list) is asked for only where the library honours it AND a
  // backward can follow: scene-level head, more Gaussians per set than the LDS sort takes, head_out on the tape with grad mode on.
  // Decided here: inside Function::forward grad mode is always off.  Otherwise (eval / no_grad / small sets) the backward half
  // allocates its buffer lazily as before and nothing is pinned from forward to backward.
  const int64_t set_P = item_offsets.has_value() && item_offsets->defined() ? max_P : (head_out.dim() >= 2 ? head_out.size(-2) : 0);
  const bool sparse = mode == 2 && !U3D_BINDING_NO_SPARSE && set_P > kSparseMinP && at::GradMode::is_enabled() && head_out.requires_grad();
  auto r = RenderLossStepFn::apply(f32c(head_out, dev), f32c(center, dev), f32c(view, dev).reshape({NV, 16}), f32c(proj, dev).reshape({NV, 16}),
                                   f32c(campos, dev).reshape({NV, 3}), f32c(gt, dev), f32c(bg, dev).reshape({3}), H, W, tanfov, mode, offset_scale,
                                   sh_degree, loss_kind, non_bg_rate, bg_rate, scale_modifier,
                                   (flags & ~(int64_t)U3D_FLAG_SPARSE_BWD) | (sparse ? (int64_t)U3D_FLAG_SPARSE_BWD : 0), want_color, isotropic, item_offsets, max_P);
  return {r[0], r[1], r[2]};
}

// empty tensors count as absent (upstream's convention for colors_precomp / cov3D_precomp); present ones become contiguous fp32 on `dev`
inline c10::optional<Tensor> opt(const c10::optional<Tensor>& t, const c10::Device& dev) {
  if (t.has_value() && t->defined() && t->numel() > 0) return f32c(*t, dev);
  return c10::nullopt;
}

// Batched form: leading dimension = sets for the Gaussian parameters, = views for cameras and outputs.
std::tuple<Tensor, Tensor, Tensor> rasterize_batched(const Tensor& means3D, const c10::optional<Tensor>& means2D, const c10::optional<Tensor>& shs,
                                                     const c10::optional<Tensor>& colors, const Tensor& opac, const c10::optional<Tensor>& scales,
                                                     const c10::optional<Tensor>& rots, const c10::optional<Tensor>& cov, const Tensor& view,
                                                     const Tensor& proj, const Tensor& campos, const Tensor& bg, int64_t n_items, int64_t vpi,
                                                     int64_t H, int64_t W, double tanfovx, double tanfovy, double scale_modifier,
                                                     int64_t sh_degree, int64_t flags, const c10::optional<Tensor>& item_offsets,
                                                     int64_t max_P) {
  const c10::Device dev = means3D.device();
  auto f = [&](const c10::optional<Tensor>& t) { return opt(t, dev); };
  auto r = RasterizeFn::apply(f32c(means3D, dev), f(means2D), f(shs), f(colors), f32c(opac, dev), f(scales), f(rots), f(cov), f32c(view, dev),
                              f32c(proj, dev), f32c(campos, dev), f32c(bg, dev), n_items, vpi, H, W, tanfovx, tanfovy, scale_modifier, sh_degree,
                              flags, false, item_offsets, max_P);
  return {r[0], r[1], r[2]};
}

// One view: the reference's operator call (gaussian_renderer/__init__.py:89-97).  means3D (P,3), means2D (P,3) gradient sink,
// shs (P,M,3) | colors (P,3), opacities (P,1), scales (P,3) + rotations (P,4) | cov3D (P,6); cameras (4,4), (4,4), (3,), bg (3,).
std::tuple<Tensor, Tensor, Tensor> rasterize_view(const Tensor& means3D, const c10::optional<Tensor>& means2D, const c10::optional<Tensor>& shs,
                                                  const c10::optional<Tensor>& colors, const Tensor& opac, const c10::optional<Tensor>& scales,
                                                  const c10::optional<Tensor>& rots, const c10::optional<Tensor>& cov, const Tensor& view,
                                                  const Tensor& proj, const Tensor& campos, const Tensor& bg, int64_t H, int64_t W,
                                                  double tanfovx, double tanfovy, double scale_modifier, int64_t sh_degree, int64_t flags) {
  const c10::Device dev = means3D.device();
  auto f = [&](const c10::optional<Tensor>& t) { return opt(t, dev); };
  auto r = RasterizeFn::apply(f32c(means3D, dev), f(means2D), f(shs), f(colors), f32c(opac, dev), f(scales), f(rots), f(cov), f32c(view, dev),
                              f32c(proj, dev), f32c(campos, dev), f32c(bg, dev), 1, 1, H, W, tanfovx, tanfovy, scale_modifier, sh_degree, flags,
                              true, c10::nullopt, 0);
  return {r[0], r[1], r[2]};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "torch binding of libunipre3d_rasterizer.so (include/unipre3d_rasterizer.h)";
  m.def("rasterize_view", &rasterize_view, "one view: the reference's per-view operator call");
  m.def("rasterize_batched", &rasterize_batched, "n_items sets x views_per_item cameras in one launch sequence",
        py::arg("means3D"), py::arg("means2D"), py::arg("shs"), py::arg("colors_precomp"), py::arg("opacities"), py::arg("scales"),
        py::arg("rotations"), py::arg("cov3D_precomp"), py::arg("viewmatrix"), py::arg("projmatrix"), py::arg("campos"), py::arg("bg"),
        py::arg("n_items"), py::arg("views_per_item"), py::arg("H"), py::arg("W"), py::arg("tanfovx"), py::arg("tanfovy"),
        py::arg("scale_modifier"), py::arg("sh_degree"), py::arg("flags"), py::arg("item_offsets") = py::none(), py::arg("max_P") = 0);
  m.def("render_view", &render_view,
        "the reference wrapper's body for one view (u3d_render_view_*): image, viewspace_points leaf, radii, visibility_filter");
  m.def("viewspace_sink", &viewspace_sink, "a fresh zero-valued leaf (requires_grad) shaped like xyz, sharing one cached zero storage");
  m.def("render_loss_step", &render_loss_step,
        "fused training step (u3d_render_loss_step): loss, [images], radii; autograd's backward returns the stored d loss / d head_out");
  m.def("unit_tensor", &unit_tensor, "the cached dL/dloss = 1 tensor of a device (fused.backward_unit)");
  m.def("abi_version", []() { return u3d_abi_version(); });
  m.def("clear_workspaces", []() { std::lock_guard<std::mutex> lock(g_ws_mu); g_ws->clear(); },
        "drop the cached backward scratch buffers (the next backward of every shape clears its accumulators itself)");
  m.def("workspaces", []() {
          std::lock_guard<std::mutex> lock(g_ws_mu);
          int n = 0, c = 0, o = 0; int64_t bytes = 0;
          for (auto& kv : *g_ws) { ++n; c += !kv.second.clean_key.empty(); o += kv.second.outstanding; bytes += kv.second.buf.defined() ? kv.second.buf.numel() : 0; }
          return std::make_tuple(n, c, o, bytes);
        },
        "(cached backward scratch buffers -- one per (device, stream) --, how many hold the accumulators-are-zero promise, how many are "
        "leased to a pending backward, total bytes)");
}
