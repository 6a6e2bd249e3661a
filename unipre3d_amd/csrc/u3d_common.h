// Internal definitions shared by the gfx950 kernels and the C-ABI (not part of the public boundary).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "unipre3d_rasterizer.h"

#define U3D_TILE 16
#define U3D_BLOCK 256       // threads per workgroup = 4 wave64 = one 16x16 tile
#define U3D_WAVE 64
// set in `clamped` by the gradient reduction for every (view, Gaussian) that received a non-zero row: preprocess_bwd reads the
// 80 B of accumulators only for those (a pixel saturates after a few dozen entries, so most visible Gaussians get none)
#define U3D_FLAG_INTERNAL_TRIAGE 0x40000000   /* (launcher -> preprocess_bwd only, never part of the ABI flags) */
#define U3D_TOUCHED_BIT 0x80000000u
#define U3D_TILE_PLAIN_BIT 0x80000000u   /* tile_last: the forward ran the loop variant without clamp / pw test */
#define U3D_NACC 10         // mean2D.xy, conic(a, b/2, c), opacity, rgb, invdepth
#define U3D_LDS_SORT_MAX 4096  // largest per-view P sorted by one workgroup in LDS
static_assert(U3D_LDS_SORT_MAX == U3D_SPARSE_BWD_MIN_P, "the public header's sparse-backward threshold is this limit (the torch binding sizes its buffers by it)");
// keys per workgroup and radix pass (P > U3D_LDS_SORT_MAX): small tiles keep more workgroups in flight (the passes are
// latency-bound), large tiles keep the per-block offset scan short
// radix workgroup shapes (threads x keys per thread and pass), measured sweep on C4 (40 k keys/view) and C5 (200 k):
// 1024 x 2 up to 64 k keys, 1024 x 4 beyond (256 x 4 / 256 x 8: 70 / 122 us; 1024 x 2 / 1024 x 4: 66 / 107 us)
#ifndef U3D_RADIX_NT_SMALL
#define U3D_RADIX_NT_SMALL 1024
#define U3D_RADIX_IT_SMALL 2
#define U3D_RADIX_NT_LARGE 1024
#define U3D_RADIX_IT_LARGE 4
#endif
static inline int u3d_radix_tile(int P) { return P <= 65536 ? U3D_RADIX_NT_SMALL * U3D_RADIX_IT_SMALL : U3D_RADIX_NT_LARGE * U3D_RADIX_IT_LARGE; }
#define U3D_MSD_BINS_MAX 2048          /* depth buckets of the large-P sort's first partition (u3d_sort.hip): 512, 1024 beyond 64 k per view, 2048 beyond 256 k */
static inline int u3d_msd_bins(int P) { return P <= 65536 ? 512 : (P <= 262144 ? 1024 : 2048); }
#define U3D_MSD_KEY_BASE 0x3E4CCCCDu   /* bits of 0.2f */

// Per-call view of the carved scratch buffers (device pointers; built on the host).
struct U3DBuffers {
  // geom: index g = view * P + i
  float* depth;       // [NV*P]   view-space z (sort key); 0 for culled
  float2* xy;         // [NV*P]   pixel-space mean
  float4* conic_op;   // [NV*P]   conic a,b,c ; opacity * aa
  float4* rgbd;       // [NV*P]   clamped colour, w = depth
  uint2* rect;        // [NV*P]   x: xmin | ymin<<16 ; y: xmax | ymax<<16   (tile units)
  uint32_t* clamped;  // [NV*P]   bit c set: colour channel c clamped at 0; U3D_TOUCHED_BIT: the backward handed it a gradient
  uint32_t* num_rendered;  // [NV] sum of tiles touched (statistics only)
  // one bit per Gaussian of the call (index gbase + i): some view handed it a gradient.  Cleared by preprocess_fwd, set by the
  // gradient reduction, read by preprocess_bwd's wave triage -- one word per 32 Gaussians instead of 8 bytes per (view, Gaussian)
  uint32_t* touched_words;
  // scene level (P > U3D_LDS_SORT_MAX): the Gaussians some view handed a gradient, as a LIST -- (set, index within the set) in
  // arrival order, appended by whoever sets a Gaussian's touched bit first, count cleared by preprocess_fwd.  The chain-rule kernel
  // of the fused step walks this list (a few thousand entries) instead of all 10^5 Gaussians of a scene (U3D_FLAG_SPARSE_BWD).
  uint2* touched_list;       // [total_P]
  uint32_t* touched_count;   // [1]
  // binning
  uint32_t* sorted_id;   // [NV*P] Gaussian index (within the set) in front-to-back order
  uint2* sorted_rect;    // [NV*P] rect of sorted_id[k]
  uint32_t* n_vis;       // [NV]   number of entries of the sorted list that are on screen
  uint32_t* sort_keys[2];  // [NV*P] x2  radix ping-pong (large P only)
  uint32_t* sort_vals[2];  // [NV*P] x2
  uint2* sort_pairs;       // [NV*P] (key, index) pairs as the partition writes them: one 8-byte store per pair (large P only)
  uint32_t* sort_hist;     // bucket totals [NV][512] (zeroed by preprocess_fwd), then per-workgroup slice offsets [NV][blocks][512]
  uint32_t* sort_over;     // bucket start table [NV][513]
  // image
  float* final_T;        // [NV*H*W]
  uint32_t* n_contrib;   // [NV*H*W]  exclusive sorted-position limit of the pixel (position at which it saturated, else UINT_MAX)
  uint32_t* tile_last;   // [NV*T]    last sorted position that contributed to any pixel of the tile
};

// How the sets' Gaussians are laid out (uniform: every set has P; ragged: prefix sums on the device, see u3d_raster_desc).
//   Gaussians of set `item`:            [gbase, gbase + Pi) of the packed parameter arrays
//   pairs of (set item, its view vk):   [vpi * gbase + vk * Pi, ... + Pi) of the per-(view, Gaussian) arrays
// For a uniform batch this is the old arithmetic, g = view * P + i.
struct U3DSpan {
  const int32_t* off;   // DEVICE [n_items + 1] or null (uniform)
  int P, vpi;
};
static inline U3DSpan u3d_span(const u3d_raster_desc& d) { return U3DSpan{d.total_P > 0 ? d.item_offsets : nullptr, d.P, d.views_per_item}; }
static inline size_t u3d_total_P(const u3d_raster_desc& d) { return d.total_P > 0 ? (size_t)d.total_P : (size_t)d.n_items * (size_t)d.P; }
#ifdef __HIPCC__
__device__ __forceinline__ void u3d_set_span(const U3DSpan& s, int item, int& Pi, size_t& gbase) {
  // (a set never has more than desc.P Gaussians: a malformed prefix-sum table is truncated here, in every kernel alike, rather
  // than indexed past the grids and LDS arrays that were sized from desc.P)
  if (s.off) { const int o0 = s.off[item]; Pi = min(max(s.off[item + 1] - o0, 0), s.P); gbase = (size_t)o0; }
  else { Pi = s.P; gbase = (size_t)item * s.P; }
}
__device__ __forceinline__ size_t u3d_view_gbase(const U3DSpan& s, int view) {   // first Gaussian of the view's set
  const int item = view / s.vpi;
  return s.off ? (size_t)s.off[item] : (size_t)item * s.P;
}
__device__ __forceinline__ void u3d_mark_touched(uint32_t* __restrict__ words, size_t gi) {
  atomicOr(&words[gi >> 5], 1u << (uint32_t)(gi & 31));
}
// ... and, the first time a Gaussian's bit goes up, append it to the touched list (one returning atomic per TOUCHED Gaussian and
// view -- thousands, not the 10^5..10^6 pairs of a launch)
__device__ __forceinline__ void u3d_mark_touched_list(uint32_t* __restrict__ words, size_t gi, uint2* __restrict__ list,
                                                      uint32_t* __restrict__ count, uint32_t item, uint32_t local) {
  const uint32_t bit = 1u << (uint32_t)(gi & 31);
  const uint32_t old = atomicOr(&words[gi >> 5], bit);
  if (list && !(old & bit)) list[atomicAdd(count, 1u)] = make_uint2(item, local);
}
__device__ __forceinline__ size_t u3d_pair_base(const U3DSpan& s, int vk, int Pi, size_t gbase) {
  return (size_t)s.vpi * gbase + (size_t)vk * Pi;
}
__device__ __forceinline__ void u3d_view_span(const U3DSpan& s, int view, int& Pv, size_t& pbase) {
  if (s.off) {
    const int item = view / s.vpi;
    size_t gbase;
    u3d_set_span(s, item, Pv, gbase);
    pbase = u3d_pair_base(s, view - item * s.vpi, Pv, gbase);
  } else { Pv = s.P; pbase = (size_t)view * s.P; }
}
#endif

// Where the per-Gaussian parameters of set `item`, Gaussian `i` come from.
//  act == 0: the operator's own tensors (means3D [P][3], shs [P][M][3], ...), strides are the natural ones.
//  act == 1/2: the Gaussian head's raw output record (model/gaussian_predictor.py:174-181 split
//  [3,1,3,4,3,9] = xyz, opacity, scaling, rotation, features_dc, features_rest), activations of
//  model/gaussian_predictor.py:249-254 applied in-kernel:
//     xyz = tanh(raw)*offset_scale + center, opacity = sigmoid, scale = exp(clamp(raw,-1,20)),
//     rotation = raw / max(norm, 1e-6) with norm taken ACROSS THE SET'S POINTS per component (act 1, the
//     reference's object-level quirk, R1) or per quaternion (act 2, scene level); SH = raw[11:11+3K].
struct U3DSource {
  const float* means;   int s_means;    // element stride between consecutive Gaussians
  const float* shs;     int s_shs;
  // split SH (u3d_render_view_*): `shs` holds coefficient 0 only (features_dc [P][1][3], stride 3) and coefficients 1.. come from
  // `shs_rest` (features_rest [P][M-1][3], stride 3 (M-1)) -- what gaussian_renderer/__init__.py:79 concatenates per view; null otherwise
  const float* shs_rest; int s_shs_rest;
  const float* colors;  // [P][3] precomputed colours or null
  const float* opac;    int s_opac;
  const float* scales;  int s_scales;
  const float* rots;    int s_rots;
  const float* cov;     // [P][6] or null
  int act;              // 0 none, 1 object-level head, 2 scene-level head
  int iso;              // act != 0: the first scaling channel serves all three axes (cfg.model.isotropic)
  const float* center;  // [B][P][3] (act != 0)
  float offset_scale;
  const float* qnorm;   // [B][4] across-point quaternion column norms (act == 1)
  // act == 1 and P <= 256 (one workgroup holds the whole set): preprocess_fwd computes the norms itself, publishes them
  // here for the backward, and clears qdot_zero -- no separate quat_norms launch
  float* qnorm_out;
  float* qdot_zero;
};

// Where per-Gaussian gradients go (same strides as the source; act != 0 chains through the activations).
struct U3DGradSink {
  float* means; float* shs; float* colors; float* opac; float* scales; float* rots; float* cov;
  float* shs_rest;      // split SH: gradient of coefficients 1.. (same layout as U3DSource::shs_rest), else null
  float* means2D;       // [NV][P][3] or null
  float* qdot;          // [B][4] sum_i raw_rot[i][c] * g[i][c]  (act == 1; finished by u3d_quat_fixup)
};

// Optional fused render loss: utils/loss_utils.py:17-45 evaluated in the render epilogue /
// backward prologue.  kind 0 = none, 1 = l2, 2 = focal_l2, 3 = l1.
struct U3DLoss {
  int kind;
  const float* gt;        // [NV][3][H][W]
  float w_bg, w_non;      // normalised focal weights 2*bg/(bg+non), 2*non/(bg+non)
  float inv_count;        // 1 / (NV*3*H*W)
  float* partial;         // [NV*T] per-tile partial sums (forward)
  const float* dloss;     // device scalar dL/dloss (backward)
};

struct U3DLayout {
  size_t geom_bytes, binning_bytes, image_bytes, backward_bytes, acc_bytes, num_rendered_offset, fused_bytes;
};

// fused scratch: [qnorm n_items*4][qdot n_items*4][loss partial NV*T]
struct U3DFused {
  float* qnorm; float* qdot; float* partial;
};
static inline size_t u3d_carve_fused(const u3d_raster_desc& d, void* base, U3DFused* f) {
  const size_t NV = (size_t)d.n_items * d.views_per_item;
  const size_t T = (size_t)((d.image_width + U3D_TILE - 1) / U3D_TILE) * ((d.image_height + U3D_TILE - 1) / U3D_TILE);
  const size_t a = (((size_t)d.n_items * 4 * sizeof(float)) + 255) & ~(size_t)255;
  if (f) {
    f->qnorm = (float*)base;
    f->qdot = (float*)((char*)base + a);
    f->partial = (float*)((char*)base + 2 * a);
  }
  return 2 * a + (((NV * T * sizeof(float)) + 255) & ~(size_t)255) + 256;
}

// partial-row buffer: U3D_PART_BLOCKS blocks of 64 sorted positions x 10 floats per tile (positions beyond go through f64
// atomics); two blocks cover the ~70-95 entries a scene-level pixel needs to saturate
#define U3D_PART_BLOCKS 2          // capacity (scratch is sized for it)
// blocks actually used: one when the whole sorted list fits in it anyway or is short (object level), two otherwise
static inline int u3d_part_blocks(const u3d_raster_desc& d) { return d.P <= 256 ? 1 : U3D_PART_BLOCKS; }
#define U3D_PART_STRIDE (U3D_PART_BLOCKS * U3D_WAVE * 10)   // floats per tile
static inline size_t u3d_align(size_t x) { return (x + 255) & ~(size_t)255; }

// single source of truth for carving; base pointers may be null when only sizes are wanted
static inline U3DLayout u3d_carve(const u3d_raster_desc& d, void* geom, void* binning, void* image, U3DBuffers* b) {
  const size_t NV = (size_t)d.n_items * d.views_per_item, NG = (size_t)d.views_per_item * u3d_total_P(d);   // (view, Gaussian) pairs
  const size_t NP = NV * (size_t)d.image_height * d.image_width;
  U3DLayout L{};
  size_t o = 0;
  char* g = (char*)geom;
#define CARVE(base, field, type, count)        \
  do {                                         \
    if (b) b->field = (type*)(base + o);       \
    o += u3d_align(sizeof(type) * (count));    \
  } while (0)
  CARVE(g, depth, float, NG);
  CARVE(g, xy, float2, NG);
  CARVE(g, conic_op, float4, NG);
  CARVE(g, rgbd, float4, NG);
  CARVE(g, rect, uint2, NG);
  CARVE(g, clamped, uint32_t, NG);
  L.num_rendered_offset = o;
  CARVE(g, num_rendered, uint32_t, NV);
  CARVE(g, touched_words, uint32_t, (u3d_total_P(d) + 31) / 32 + 1);
  CARVE(g, touched_list, uint2, d.P > U3D_LDS_SORT_MAX ? u3d_total_P(d) : 0);
  CARVE(g, touched_count, uint32_t, 1);
  L.geom_bytes = o > 0 ? o : 256;
  o = 0;
  char* bn = (char*)binning;
  CARVE(bn, sorted_id, uint32_t, NG);
  CARVE(bn, sorted_rect, uint2, NG);
  CARVE(bn, n_vis, uint32_t, NV);
  if (d.P > U3D_LDS_SORT_MAX) {
    CARVE(bn, sort_keys[0], uint32_t, NG);
    CARVE(bn, sort_keys[1], uint32_t, NG);
    CARVE(bn, sort_vals[0], uint32_t, NG);
    CARVE(bn, sort_vals[1], uint32_t, NG);
    CARVE(bn, sort_pairs, uint2, NG);
    const size_t nblk = ((size_t)d.P + u3d_radix_tile(d.P) - 1) / u3d_radix_tile(d.P);
    CARVE(bn, sort_hist, uint32_t, NV * (size_t)u3d_msd_bins(d.P) * (nblk + 1));
    CARVE(bn, sort_over, uint32_t, NV * ((size_t)u3d_msd_bins(d.P) + 1));
  } else if (b) {
    b->sort_keys[0] = b->sort_keys[1] = b->sort_vals[0] = b->sort_vals[1] = b->sort_hist = b->sort_over = nullptr;
    b->sort_pairs = nullptr;
  }
  L.binning_bytes = o > 0 ? o : 256;
  o = 0;
  char* im = (char*)image;
  CARVE(im, final_T, float, NP);
  CARVE(im, n_contrib, uint32_t, NP);
  CARVE(im, tile_last, uint32_t, NV * (size_t)((d.image_width + U3D_TILE - 1) / U3D_TILE) * ((d.image_height + U3D_TILE - 1) / U3D_TILE));
  L.image_bytes = o > 0 ? o : 256;
#undef CARVE
  // f64 accumulators: global_atomic_add_f64 makes the cross-tile sum order-insensitive at fp32 output precision
  // + per-tile partials of the first 64 sorted positions: [NV*T][64][10] floats + [NV*T] row counts (see tile_backward)
  L.acc_bytes = u3d_align(sizeof(double) * U3D_NACC * (NG > 0 ? NG : 1));
  {
    const size_t Tn = (size_t)((d.image_width + U3D_TILE - 1) / U3D_TILE) * ((d.image_height + U3D_TILE - 1) / U3D_TILE);
    // per tile: 64 sorted positions x 10 floats (the LDS rows as they are) + a compact count array
    L.backward_bytes = L.acc_bytes + u3d_align(sizeof(float) * U3D_PART_STRIDE * (NV * Tn > 0 ? NV * Tn : 1)) +
                       u3d_align(sizeof(uint32_t) * (NV * Tn > 0 ? NV * Tn : 1));
  }
  L.fused_bytes = u3d_carve_fused(d, nullptr, nullptr);
  return L;
}

// launchers (one per translation unit)
void u3d_launch_preprocess_fwd(const u3d_raster_desc& d, const U3DBuffers& b, const U3DSource& src, const float* viewmatrix,
                               const float* projmatrix, const float* campos, int32_t* radii, double* acc_zero, hipStream_t s,
                               uint8_t* visible = nullptr);   // visible[pair] = radius > 0 (`visibility_filter`, gaussian_renderer/__init__.py:103)
// (preprocess_bwd triages by b.touched_words when d.P > U3D_LDS_SORT_MAX; the reduction kernels set the bits in that case)
static inline bool u3d_uses_touched_words(const u3d_raster_desc& d) { return d.P > U3D_LDS_SORT_MAX; }
#define U3D_TOUCHED_NOT_LISTED 0xFFFFFFFFu   // value of the touched count after a forward half that built no list
// large-P sort: the tile kernels read a sorted entry's rectangle through sorted_id (b.rect) instead of a sorted copy (b.sorted_rect)
static inline int u3d_rect_indirect(const u3d_raster_desc& d) { return d.P > U3D_LDS_SORT_MAX ? 1 : 0; }
void u3d_launch_preprocess_bwd(const u3d_raster_desc& d, const U3DBuffers& b, const U3DSource& src, const float* viewmatrix,
                               const float* projmatrix, const float* campos, const int32_t* radii, const double* acc,
                               const U3DGradSink& sink, hipStream_t s, double* acc_reset = nullptr, const float* gscale = nullptr,
                               bool sparse = false);
// U3D_FLAG_SPARSE_BWD is honoured for the scene-level head at scene-level sizes only (what the touched list exists for)
static inline bool u3d_sparse_bwd(const u3d_raster_desc& d, int head_mode) {
  return (d.flags & U3D_FLAG_SPARSE_BWD) != 0 && head_mode == 2 && d.P > U3D_LDS_SORT_MAX;
}
// true when preprocess_fwd also produces the per-view depth order (P <= 256): skip u3d_launch_depth_sort then
bool u3d_preprocess_sorts(const u3d_raster_desc& d);
void u3d_launch_quat_norms(const u3d_raster_desc& d, const float* rots, int s_rots, float* qnorm, float* qdot_zero, hipStream_t s);
void u3d_launch_quat_fixup(const u3d_raster_desc& d, const float* rots, int s_rots, const float* qnorm, const float* qdot,
                           float* d_rots, hipStream_t s);
void u3d_launch_depth_sort(const u3d_raster_desc& d, const U3DBuffers& b, const int32_t* radii, hipStream_t s);
void u3d_launch_render_fwd(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, float* out_color,
                           float* out_invdepth, const U3DLoss& loss, hipStream_t s);
void u3d_launch_render_bwd(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, const float* dL_dcolor,
                           const float* dL_dinvdepth, const float* out_color, const U3DLoss& loss, double* acc,
                           float* part, hipStream_t s);
void u3d_launch_render_fb(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, float* out_color,
                          const U3DLoss& loss, double* acc, float* part, float* loss_out, hipStream_t s,
                          float* zero_fill = nullptr, size_t zero_floats = 0, bool list_touched = false);
void u3d_launch_loss_reduce(int n, const float* partial, float inv_count, float* loss_out, hipStream_t s);

#ifdef __HIPCC__
// ---- device helpers -------------------------------------------------------------------------
// Workgroup -> logical id remap so that consecutive logical ids (tiles of one view, which share
// that view's sorted Gaussian state) stay on ONE XCD's L2.  Hardware places workgroup b on XCD
// b % 8 (observed, speed only); this bijection sends logical ids [x*q .. ) to XCD x.
__device__ __forceinline__ uint32_t u3d_xcd_remap(uint32_t bid, uint32_t nblocks) {
  const uint32_t q = nblocks >> 3, r = nblocks & 7u;
  const uint32_t xcd = bid & 7u, k = bid >> 3;
  const uint32_t start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + k;
}

// Per-view variant for the tile kernels: XCD x gets a contiguous ~1/8 of EVERY view's tiles (in order), so the XCDs stay
// balanced when the views differ in cost (measured: per-view mean depth 10..28 entries at C2; 1-2 views per XCD at scene
// level) while neighbouring tiles -- which share image cache lines and the view's Gaussian state -- still meet in one L2.
// Bijective on [0, nviews*T) for any T: block b sits on XCD b % 8; within view v (blocks [vT, vT+T)) the blocks of residue
// class x are numbered in order and mapped to the x-th chunk of the view's tile range.
// Round 6 (tools/tile_timeline.sh): with the same chunk of EVERY view on the same XCD, an XCD owns one image region -- at object level the top and
// bottom rows walk 5 % more entries than the centre, and the two XCDs that own them finished 7.5 us after the first (152 .. 159 us).  When the
// chunks are equal (T % 8 == 0) the chunk an XCD takes rotates with `rot` (the tile kernels pass the view at object level, P <= 256: C2 render_fb
// scope 174.1 -> 171.2 us in five alternating pairs; 0 at scene level, where the rotation measured +0.2 ... +1.7 us on C3 / C4 / C5), so every XCD
// sees every region: U3D_XCD_ROTATE (0 = round 5's map).
#ifndef U3D_XCD_ROTATE
#define U3D_XCD_ROTATE 1
#endif
__device__ __forceinline__ uint32_t u3d_xcd_chunk_in_view(uint32_t j, uint32_t view, uint32_t T, uint32_t rot = 0u) {   // tile (within the view) of block j
#if U3D_XCD_ROTATE
  if ((T & 7u) == 0u) return (((j + rot) & 7u) * (T >> 3)) + (j >> 3);   // block j sits on XCD j % 8 (view * T is a multiple of 8)
#endif
  const uint32_t r = (view * T) & 7u, m = r + j, x = m & 7u;
  auto below = [](uint32_t n, uint32_t c) { return (n >> 3) * c + min(n & 7u, c); };   // #{i < n : i % 8 < c}
  const uint32_t k = ((m + 7u - x) >> 3) - ((r + 7u - x) >> 3);                         // rank of this block in its class
  return (below(r + T, x) - below(r, x)) + k;
}
__device__ __forceinline__ uint32_t u3d_xcd_remap_view(uint32_t bid, uint32_t T) {   // linear form: bid = view * T + j
  const uint32_t view = bid / T;
  return view * T + u3d_xcd_chunk_in_view(bid - view * T, view, T);
}

// Sum over the 64 lanes by DPP (quad, half-row, row, then the two row broadcasts of GFX9); the total comes back wave-uniform.
// (A shuffle butterfly costs six LDS round trips and ~40 address instructions.)
__device__ __forceinline__ float u3d_wave_sum(float v) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// Sum over the 4 lanes of a DPP quad (all four lanes receive it): two v_add_f32_dpp instead of two LDS shuffles.
__device__ __forceinline__ float u3d_quad_sum(float v) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(v));
  return v;
}

// Minimum of a 32-bit value over the wave, wave-uniform (SGPR): four DPP steps leave every lane with its 16-lane row's minimum, the four row
// results are read into SGPRs and combined on the scalar unit (no LDS round trips).
template <int CTRL>
__device__ __forceinline__ uint32_t u3d_dpp_min_step(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, false);
  return o < v ? o : v;
}
__device__ __forceinline__ uint32_t u3d_wave_min_u32(uint32_t v) {
  v = u3d_dpp_min_step<0xB1>(v);    // quad_perm [1,0,3,2]
  v = u3d_dpp_min_step<0x4E>(v);    // quad_perm [2,3,0,1]
  v = u3d_dpp_min_step<0x141>(v);   // row_half_mirror
  v = u3d_dpp_min_step<0x140>(v);   // row_mirror
  uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
#pragma unroll
  for (int q = 1; q < 4; ++q) {
    const uint32_t u = (uint32_t)__builtin_amdgcn_readlane((int)v, q * 16);
    r = u < r ? u : r;
  }
  return r;
}

__device__ __forceinline__ uint32_t u3d_lane_id() {
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
#endif
