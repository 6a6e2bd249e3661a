// Per-Gaussian projection kernels for gfx950: forward (3D -> 2D mean, EWA covariance, conic, radius,
// tile rectangle, SH colour) and the matching backward (conic/mean2D/colour gradients -> mean3D,
// scale, rotation, opacity, SH).  Semantics: SURVEY.md R4 steps 1-7 and R5/R6; restated on the CPU in
// oracle/raster_oracle.c.  One thread per (view, Gaussian) forward; one thread per (set, Gaussian)
// backward, looping over that set's views so the per-set gradient needs no atomics.
#include "u3d_common.h"

// The forward projection is evaluated in the CPU restatement's operation order with FMA contraction OFF (the oracle is built
// -ffp-contract=off, oracle/Makefile) and IEEE divide / sqrt (hipcc's default for fp32): depth, pixel mean, conic, colour and
// -- the path's integer outputs -- radius, tile rectangle and num_rendered are then bit-identical to oracle/raster_oracle.c:317-363
// on the operator route (tests/arbiter.py::assert_radii is np.array_equal).  The backward kernel re-enables contraction below.
#pragma clang fp contract(off)

namespace {

constexpr float SH_C0 = 0.28209479177387814f;
constexpr float SH_C1 = 0.4886025119029199f;
__device__ constexpr float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                       -1.0925484305920792f, 0.5462742152960396f};
__device__ constexpr float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                       0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                                       -0.5900435899266435f};

struct Cam {
  float V[16], Pm[16], pos[3];
};

__device__ __forceinline__ void load_cam(Cam& c, const float* __restrict__ view, const float* __restrict__ proj,
                                         const float* __restrict__ campos, int v) {
#pragma unroll
  for (int k = 0; k < 16; ++k) { c.V[k] = view[v * 16 + k]; c.Pm[k] = proj[v * 16 + k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) c.pos[k] = campos[v * 3 + k];
}

struct Ewa {
  float t[3];  // view-space point, x/y clamped
  float xmask, ymask;
  float M2[6];  // J * W (2x3)
  float fx, fy;
};

namespace exact {
#define U3D_FP_CONTRACT _Pragma("clang fp contract(off)")
#include "u3d_proj_helpers.inc"
#undef U3D_FP_CONTRACT
}  // namespace exact
namespace fast {
#define U3D_FP_CONTRACT _Pragma("clang fp contract(fast)")
#include "u3d_proj_helpers.inc"
#undef U3D_FP_CONTRACT
}  // namespace fast

template <int D>
__device__ __forceinline__ void sh_to_rgb(const float* __restrict__ sh /* [M][3] */, const float dir[3], float rgb[3],
                                          uint32_t& clamp_bits) {
  const float x = dir[0], y = dir[1], z = dir[2];
  clamp_bits = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float res = SH_C0 * sh[c];
    if (D > 0) {
      res = res - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
      if (D > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] + SH_C2[2] * (2.f * zz - xx - yy) * sh[18 + c] +
              SH_C2[3] * xz * sh[21 + c] + SH_C2[4] * (xx - yy) * sh[24 + c];
        if (D > 2) {
          res = res + SH_C3[0] * y * (3.f * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
                SH_C3[2] * y * (4.f * zz - xx - yy) * sh[33 + c] + SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * sh[36 + c] +
                SH_C3[4] * x * (4.f * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
                SH_C3[6] * x * (xx - 3.f * yy) * sh[45 + c];
        }
      }
    }
    res += 0.5f;
    if (res < 0.f) clamp_bits |= 1u << c;
    rgb[c] = fmaxf(res, 0.f);
  }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------------
// Loads Gaussian (item, i) and applies the head activations when src.act != 0 (see U3DSource).
struct GaussIn {
  float p[3], op, s[3], q[4];
  float th[3];      // tanh(raw xyz)            (act != 0)
  float raw_s[3];   // raw scaling               (act != 0)
  float raw_q[4];   // raw rotation              (act != 0)
  float qn[4];      // normalisation denominators (act != 0)
};

__device__ __forceinline__ void load_gaussian(const U3DSource& src, int item, size_t gi, GaussIn& g) {
#pragma unroll
  for (int k = 0; k < 3; ++k) g.p[k] = src.means[gi * src.s_means + k];
  g.op = src.opac[gi * src.s_opac];
  if (src.scales) {
#pragma unroll
    for (int k = 0; k < 3; ++k) g.s[k] = src.scales[gi * src.s_scales + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) g.q[k] = src.rots[gi * src.s_rots + k];
  } else {
    g.s[0] = g.s[1] = g.s[2] = 0.f; g.q[0] = g.q[1] = g.q[2] = g.q[3] = 0.f;
  }
  if (src.act != 0) {
    if (src.iso) g.s[1] = g.s[2] = g.s[0];   // cfg.model.isotropic: scaling[:, :1].expand(-1, 3, -1) (model/gaussian_predictor.py:308-310)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      g.th[k] = tanhf(g.p[k]);
      g.p[k] = g.th[k] * src.offset_scale + src.center[gi * 3 + k];
      g.raw_s[k] = g.s[k];
      g.s[k] = expf(fminf(fmaxf(g.s[k], -1.f), 20.f));
    }
    g.op = 1.f / (1.f + expf(-g.op));
    float n4 = sqrtf(g.q[0] * g.q[0] + g.q[1] * g.q[1] + g.q[2] * g.q[2] + g.q[3] * g.q[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      g.raw_q[k] = g.q[k];
      g.qn[k] = fmaxf(src.act == 1 ? src.qnorm[item * 4 + k] : n4, 1e-6f);
      g.q[k] = g.q[k] / g.qn[k];
    }
  }
}

template <int D>
__global__ __launch_bounds__(U3D_BLOCK) void preprocess_fwd_kernel(
    U3DSpan span, int vpi, int vpt, int H, int W, float tanx, float tany, float mod, int flags, U3DSource src,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos,
    int32_t* __restrict__ radii, float* __restrict__ depth, float2* __restrict__ xy, float4* __restrict__ conic_op,
    float4* __restrict__ rgbd, uint2* __restrict__ rect, uint32_t* __restrict__ clamped,
    uint32_t* __restrict__ num_rendered, double* __restrict__ acc_zero, size_t NG, uint32_t* __restrict__ sorted_id,
    uint2* __restrict__ sorted_rect, uint32_t* __restrict__ n_vis, uint32_t* __restrict__ msd_total, int msd_bins,
    uint32_t* __restrict__ touched_words, uint8_t* __restrict__ visible, uint32_t* __restrict__ touched_count, uint32_t touched_count_init) {
  // (the touched list restarts; U3D_TOUCHED_NOT_LISTED when this step does not build one, so that a sparse backward half can tell)
  if (touched_count && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *touched_count = touched_count_init;
  // (msd_total != null: P > 4096; the depth sort that follows partitions by depth bucket with one global atomic per (workgroup,
  // bucket) on these per-view totals -- the first workgroup of every (set, view slice) clears them here instead of a memset node)
  if (msd_total && blockIdx.x == 0) {
    const int nv0 = blockIdx.z * vpt, nv1 = min(vpi, nv0 + vpt);
    for (int e = threadIdx.x; e < (nv1 - nv0) * msd_bins; e += U3D_BLOCK) msd_total[(size_t)(blockIdx.y * vpi + nv0) * msd_bins + e] = 0u;
  }
  // (sorted_id != null: P <= 256, the block owns the whole set -> the per-view depth sort is fused in, see below)
  __shared__ __attribute__((aligned(16))) unsigned long long s_keys[U3D_BLOCK];
  __shared__ uint2 s_rects[U3D_BLOCK];
  // One thread = one Gaussian of set blockIdx.y; it walks the views [v0, v1) of that set, so the view-independent work
  // (head activations, Sigma, SH coefficient fetch) is done once.  blockIdx.z splits the views when P is small.
  extern __shared__ __attribute__((aligned(16))) float s_rec[];   // fused mode: this block's head records, staged coalesced
  constexpr int K = (D + 1) * (D + 1);
  const int item = blockIdx.y;
  int P;           // Gaussians of THIS set (ragged batches: blocks past the end of a short set find nothing to do)
  size_t gbase;    // its first Gaussian in the packed parameter arrays
  u3d_set_span(span, item, P, gbase);
  const int i0 = blockIdx.x * U3D_BLOCK;
  const int i = i0 + threadIdx.x;
  const int v0 = blockIdx.z * vpt, v1 = min(vpi, v0 + vpt);
  U3DSource lsrc = src;
  size_t gi = gbase + (i < P ? i : 0);
  // (scene level: clear the "some view handed this Gaussian a gradient" bits the backward's wave triage reads; a word shared by
  // two sets is cleared by both, all before any reduction kernel sets a bit)
  if (touched_words && blockIdx.z == 0 && i < P && ((gi & 31) == 0 || i == 0)) touched_words[gi >> 5] = 0u;
  if (src.act != 0) {
    // rows i0 .. i0+255 of head_out are contiguous: coalesced copy into LDS, then row-strided reads (C odd: no conflicts)
    const int C = src.s_means;
    const int nrow = min(U3D_BLOCK, P - i0);
    const float* rows = src.means + (gbase + i0) * C;
    for (int e = threadIdx.x; e < nrow * C; e += U3D_BLOCK) s_rec[e] = rows[e];
    __syncthreads();
    lsrc.means = s_rec; lsrc.opac = s_rec + 3; lsrc.scales = s_rec + 4; lsrc.rots = s_rec + 7; lsrc.shs = s_rec + 11;
    lsrc.center = src.center + (gbase + i0) * 3;
    if (src.act == 1 && src.qnorm_out) {
      // the block holds the whole set: across-point quaternion column norms from the staged records (same summation
      // order as quat_norms_kernel); the first view slice publishes them for the backward and clears qdot
      __shared__ float s_qsm[4][4];
      __shared__ float s_qn[4];
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      if (i < P) {
        const float* r = s_rec + (size_t)threadIdx.x * C + 7;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += r[k] * r[k];
      }
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = u3d_wave_sum(a[k]);
        if (lane == 0) s_qsm[wave][k] = v;
      }
      __syncthreads();
      if (threadIdx.x < 4) {
        const float n = sqrtf(s_qsm[0][threadIdx.x] + s_qsm[1][threadIdx.x] + s_qsm[2][threadIdx.x] + s_qsm[3][threadIdx.x]);
        s_qn[threadIdx.x] = n;
        if (blockIdx.z == 0) {
          src.qnorm_out[item * 4 + threadIdx.x] = n;
          if (src.qdot_zero) src.qdot_zero[item * 4 + threadIdx.x] = 0.f;
        }
      }
      __syncthreads();
      lsrc.qnorm = s_qn - item * 4;
    }
  }
  uint32_t touched_local[1] = {0};
  (void)touched_local;
  GaussIn gin;
  float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float shc[K * 3];
  if (i < P) {
    const size_t li = src.act != 0 ? (size_t)threadIdx.x : gi;     // index into the (possibly LDS-resident) source
    load_gaussian(lsrc, item, li, gin);
    if (src.cov) {
#pragma unroll
      for (int k = 0; k < 6; ++k) c6[k] = src.cov[gi * 6 + k];
    } else {
      exact::cov3d_from_scale_rot(gin.s, mod, gin.q, c6);
    }
    if (!src.colors) {
      const float* sh = lsrc.shs + li * (size_t)lsrc.s_shs;
      if (lsrc.shs_rest) {   // split SH: coefficient 0 from features_dc, the rest from features_rest
        const float* shr = lsrc.shs_rest + li * (size_t)lsrc.s_shs_rest;
#pragma unroll
        for (int k = 0; k < K * 3; ++k) shc[k] = k < 3 ? sh[k] : shr[k - 3];
      } else {
#pragma unroll
        for (int k = 0; k < K * 3; ++k) shc[k] = sh[k];
      }
    }
  }
  for (int vk = v0; vk < v1; ++vk) {
  const int view = item * vpi + vk;
  const size_t pbase = u3d_pair_base(span, vk, P, gbase);   // first (view, Gaussian) pair of this view
  uint32_t touched = 0;
  unsigned long long sort_key = ~0ull;
  uint2 sort_rect = make_uint2(0u, 0u);
  if (i < P) {
    const size_t g = pbase + i;
    Cam cam;
    load_cam(cam, viewmatrix, projmatrix, campos, view);
    const float* p = gin.p;
    int radius = 0;
    float zv = cam.V[2] * p[0] + cam.V[6] * p[1] + cam.V[10] * p[2] + cam.V[14];
    float2 pix = make_float2(0.f, 0.f);
    float4 co = make_float4(0.f, 0.f, 0.f, 0.f), col = make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 rc = make_uint2(0u, 0u);
    uint32_t cb = 0;
    if (zv > 0.2f) {
      float hom[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) hom[j] = cam.Pm[j] * p[0] + cam.Pm[4 + j] * p[1] + cam.Pm[8 + j] * p[2] + cam.Pm[12 + j];
      const float p_w = 1.0f / (hom[3] + 0.0000001f);
      const float fx = (float)W / (2.f * tanx), fy = (float)H / (2.f * tany);
      Ewa e;
      exact::ewa_setup(e, p, cam, fx, fy, tanx, tany);
      float abc[3];
      exact::cov2d(e, c6, abc);
      const float det0 = abc[0] * abc[2] - abc[1] * abc[1];
      abc[0] += 0.3f; abc[2] += 0.3f;
      const float det = abc[0] * abc[2] - abc[1] * abc[1];
      float aa = 1.f;
      if (flags & U3D_FLAG_ANTIALIASING) aa = sqrtf(fmaxf(0.000025f, det0 / det));
      if (det != 0.f) {
        const float det_inv = 1.f / det;
        const float mid = 0.5f * (abc[0] + abc[2]);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const float my_radius = ceilf(3.f * sqrtf(fmaxf(mid + sq, mid - sq)));
        const float px = ((hom[0] * p_w + 1.f) * (float)W - 1.f) * 0.5f;
        const float py = ((hom[1] * p_w + 1.f) * (float)H - 1.f) * 0.5f;
        const int gx = (W + U3D_TILE - 1) / U3D_TILE, gy = (H + U3D_TILE - 1) / U3D_TILE;
        const int r = (int)my_radius;
        const int x0 = clampi((int)((px - (float)r) / (float)U3D_TILE), 0, gx);
        const int y0 = clampi((int)((py - (float)r) / (float)U3D_TILE), 0, gy);
        const int x1 = clampi((int)((px + (float)r + (float)(U3D_TILE - 1)) / (float)U3D_TILE), 0, gx);
        const int y1 = clampi((int)((py + (float)r + (float)(U3D_TILE - 1)) / (float)U3D_TILE), 0, gy);
        const int nt = (x1 - x0) * (y1 - y0);
        if (nt > 0) {
          float rgb[3];
          if (src.colors) {
            rgb[0] = src.colors[gi * 3]; rgb[1] = src.colors[gi * 3 + 1]; rgb[2] = src.colors[gi * 3 + 2];
          } else {
            float dir[3] = {p[0] - cam.pos[0], p[1] - cam.pos[1], p[2] - cam.pos[2]};
            const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            dir[0] /= len; dir[1] /= len; dir[2] /= len;
            sh_to_rgb<D>(shc, dir, rgb, cb);
          }
          radius = r;
          touched = (uint32_t)nt;
          pix = make_float2(px, py);
          co = make_float4(abc[2] * det_inv, -abc[1] * det_inv, abc[0] * det_inv, gin.op * aa);
          col = make_float4(rgb[0], rgb[1], rgb[2], zv);
          rc = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16));
        }
      }
    }
    radii[g] = radius;
    if (visible) visible[g] = radius > 0 ? 1 : 0;
    depth[g] = radius > 0 ? zv : 0.f;
    xy[g] = pix;
    conic_op[g] = co;
    rgbd[g] = col;
    rect[g] = rc;
    clamped[g] = cb;   // (every pair is written, culled ones too: skipping them turns full-line stores into scattered partial lines -- measured 86 against 46 us at C5)
    if (radius > 0) sort_key = ((unsigned long long)__float_as_uint(zv) << 32) | (uint32_t)i;
    sort_rect = rc;
    if (acc_zero) {   // single-pass training step: clear this (view, Gaussian)'s gradient accumulators here (saves a memset node)
#pragma unroll
      for (int k = 0; k < U3D_NACC; ++k) acc_zero[(size_t)k * NG + g] = 0.0;
    }
  }
  // statistics (U3D_FLAG_STATS only): num_rendered[view] += sum(tiles touched), wave reduce + one atomic per wave
  if (flags & U3D_FLAG_STATS) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) touched += __shfl_xor(touched, o);
    if ((threadIdx.x & 63) == 0 && touched) atomicAdd(&num_rendered[view], touched);
  }
  if (sorted_id) {
    // fused per-view depth sort (same result as depth_sort_lds_kernel) of the block's 256 (depth bits, index) keys.  Round 5: by RANK -- every
    // thread counts the keys smaller than its own from LDS broadcast reads and writes its key to that position: 3 barriers instead of the 36
    // of the bitonic network this replaced (the kernel is latency-bound: ~2 us of its 10 at C2).  A culled slot's key is ~0 with the thread
    // index below it, so all 256 keys are distinct and the ranks are a permutation.
    const int tid = threadIdx.x;
    __shared__ unsigned long long s_sorted[U3D_BLOCK];
    __syncthreads();               // previous view's readers are done with s_keys / s_rects / s_sorted
    const unsigned long long mine = sort_key != ~0ull ? sort_key : (0xFFFFFFFF00000000ull | (uint32_t)tid);
    s_keys[tid] = mine;
    s_rects[tid] = sort_rect;
    __syncthreads();
    {
      int rank = 0;
      const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(s_keys);
#pragma unroll 8
      for (int q = 0; q < U3D_BLOCK / 2; ++q) {
        const ulonglong2 two = k2[q];          // the same address in every lane: a broadcast
        rank += (two.x < mine ? 1 : 0) + (two.y < mine ? 1 : 0);
      }
      s_sorted[rank] = mine;
    }
    __syncthreads();
    const unsigned long long kk = s_sorted[tid];
    const bool vis = (uint32_t)(kk >> 32) != 0xFFFFFFFFu;
    if (tid == 0 && !vis) n_vis[view] = 0;
    if (vis && (tid == U3D_BLOCK - 1 || (uint32_t)(s_sorted[tid + 1] >> 32) == 0xFFFFFFFFu)) n_vis[view] = (uint32_t)(tid + 1);
    if (tid < P) {
      const uint32_t id = vis ? (uint32_t)kk : 0u;
      sorted_id[pbase + tid] = id;
      sorted_rect[pbase + tid] = vis ? s_rects[id] : make_uint2(0u, 0u);
    }
  }
  }
}

// ---------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(U3D_BLOCK) void preprocess_bwd_kernel(
    U3DSpan span, int vpi, int M, int H, int W, float tanx, float tany, float mod, int flags, size_t NG, U3DSource src,
    const float* __restrict__ viewmatrix, const float* __restrict__ projmatrix, const float* __restrict__ campos,
    const int32_t* __restrict__ radii, const uint32_t* __restrict__ clamped, const double* acc, double* acc_reset,
    U3DGradSink sink, const float* __restrict__ gscale, const uint32_t* __restrict__ touched_words,
    const uint2* __restrict__ sparse_list_in, const uint32_t* __restrict__ sparse_count) {
#pragma clang fp contract(fast)
  // gscale: device scalar dL/dloss of the fused step (autograd's grad_output) or null (= 1).  Every output of this kernel -- and the
  // column dot products quat_fixup finishes -- is linear in the accumulators, so scaling them as they are read IS the d_head * g
  // multiply; a wave-uniform scalar load, no extra launch.
  const float gs = gscale ? gscale[0] : 1.f;
  // 4 consecutive lanes (a DPP quad) share one Gaussian and split its views: lane&3 = view slot
  __shared__ float s_qdot[4][4];
  // sparse mode (U3D_FLAG_SPARSE_BWD, scene level): the quad's Gaussian comes from the touched list the gradient reduction
  // appended to -- (set, index) in arrival order; the rows of every other Gaussian were zero-filled by the forward half and a
  // workgroup beyond the end of the list leaves at once
  // A forward half that ran WITHOUT U3D_FLAG_SPARSE_BWD leaves the count at U3D_TOUCHED_NOT_LISTED (and did not zero-fill): this launch then
  // walks its grid -- sized n_items x ceil(P / 64) in sparse mode for exactly this -- as the dense form does (one flag in one half only
  // used to return U3D_OK with an uninitialised gradient).
  const uint2* sparse_list = sparse_list_in;
  int item = blockIdx.y, bx = blockIdx.x;
  if (sparse_list) {
    const uint32_t n = *sparse_count;
    if (n == U3D_TOUCHED_NOT_LISTED) {
      const int bpi = (span.P + U3D_BLOCK / 4 - 1) / (U3D_BLOCK / 4);
      item = bx / bpi; bx -= item * bpi;
      sparse_list = nullptr;
    } else if ((uint32_t)(bx * (U3D_BLOCK / 4)) >= n) return;
  }
  int i = bx * (U3D_BLOCK / 4) + (threadIdx.x >> 2);
  bool listed = true;
  if (sparse_list) {
    const uint32_t n = *sparse_count;
    listed = (uint32_t)i < n;
    const uint2 e = sparse_list[listed ? i : 0];
    item = (int)e.x; i = (int)e.y;
  }
  int P;
  size_t gbase;
  u3d_set_span(span, item, P, gbase);
  const size_t pbase0 = u3d_pair_base(span, 0, P, gbase);   // pairs of (this set, view slot vk): pbase0 + vk * P + i
  const int vslot = threadIdx.x & 3;
  const bool alive = listed && i < P;
  const size_t gi = gbase + (alive ? i : 0);
  constexpr int K = (D + 1) * (D + 1);
  const bool writer = alive && vslot == 0;
  float qd[4] = {0.f, 0.f, 0.f, 0.f};
  // Does any view hand a Gaussian of this wave a gradient?  At scene level almost none does (a pixel saturates after a few dozen
  // of the view's 10^5 sorted entries): such a wave writes its zeros and skips the parameter loads, the activations and Sigma.
  // (Only asked for at scene level, U3D_FLAG_INTERNAL_TRIAGE: at object level every Gaussian is live and the extra dependent
  // load phase costs this latency-bound kernel ~1 us.)
  bool lane_live = (flags & U3D_FLAG_INTERNAL_TRIAGE) == 0 || (sparse_list != nullptr && alive);
  // scene level, fused mode: the block first clears ITS 64 rows of d(head output) with full-width stores (64 x C consecutive floats),
  // then only the waves that own a touched Gaussian compute anything; a row-by-row zero fill from each idle wave ran at ~1 TB/s
  const bool block_zero = !sparse_list && !lane_live && src.act != 0;
  if (block_zero) {
    const int C = src.s_means, ib = bx * (U3D_BLOCK / 4), rows = min(U3D_BLOCK / 4, P - ib);
    if (rows > 0) {
      float* o = sink.means + (gbase + ib) * C;
      const int n = rows * C;
      if ((reinterpret_cast<uintptr_t>(o) & 15u) == 0u) {
        const int n4 = n >> 2;
        for (int e = threadIdx.x; e < n4; e += U3D_BLOCK) reinterpret_cast<float4*>(o)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = (n4 << 2) + threadIdx.x; e < n; e += U3D_BLOCK) o[e] = 0.f;
      } else {
        for (int e = threadIdx.x; e < n; e += U3D_BLOCK) o[e] = 0.f;
      }
    }
    __syncthreads();   // (the rows a live wave writes below belong to this block)
  }
  if (!lane_live && !sparse_list) lane_live = alive && ((touched_words[gi >> 5] >> (uint32_t)(gi & 31)) & 1u) != 0u;   // one word per 32 Gaussians
  if (sparse_list && __ballot(lane_live) == 0ull) return;      // (a wave past the end of the list: its rows are already zero)
  if (__ballot(lane_live) == 0ull) {
    if (sink.means2D) {
      for (int vk = vslot; vk < (alive ? vpi : 0); vk += 4) {
        const size_t g = pbase0 + (size_t)vk * P + i;
        sink.means2D[g * 3] = 0.f; sink.means2D[g * 3 + 1] = 0.f; sink.means2D[g * 3 + 2] = 0.f;
      }
    }
    if (src.act != 0) {
      if (!block_zero) {
        // fused mode: the gradient rows of the wave's 16 Gaussians are 16 * C consecutive floats of d(head output)
        const int C = src.s_means, iw = bx * (U3D_BLOCK / 4) + (threadIdx.x >> 6) * 16, n = min(16, P - iw) * C;
        float* o = sink.means + (gbase + iw) * C;
        for (int e = threadIdx.x & 63; e < n; e += 64) o[e] = 0.f;
      }
    } else if (writer) {
#pragma unroll
      for (int k = 0; k < 3; ++k) sink.means[gi * src.s_means + k] = 0.f;
      sink.opac[gi * src.s_opac] = 0.f;
      if (sink.colors) { sink.colors[gi * 3] = 0.f; sink.colors[gi * 3 + 1] = 0.f; sink.colors[gi * 3 + 2] = 0.f; }
      if (sink.cov) {
#pragma unroll
        for (int k = 0; k < 6; ++k) sink.cov[gi * 6 + k] = 0.f;
      }
      if (sink.shs) {
        float* o = sink.shs + gi * (size_t)src.s_shs;
        if (sink.shs_rest) {
          float* orr = sink.shs_rest + gi * (size_t)src.s_shs_rest;
          o[0] = o[1] = o[2] = 0.f;
          for (int k = 3; k < M * 3; ++k) orr[k - 3] = 0.f;
        } else {
          for (int k = 0; k < M * 3; ++k) o[k] = 0.f;
        }
      }
      if (sink.scales) {
#pragma unroll
        for (int k = 0; k < 3; ++k) sink.scales[gi * src.s_scales + k] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) sink.rots[gi * src.s_rots + k] = 0.f;
      }
    }
  } else {
  GaussIn gin;
  load_gaussian(src, item, gi, gin);
  const float* p = gin.p;
  const float* s = gin.s;
  const float* q = gin.q;
  const float* shs = src.shs;
  float* dL_dmeans2D = sink.means2D;
  float c6[6];
  if (src.cov) {
#pragma unroll
    for (int k = 0; k < 6; ++k) c6[k] = src.cov[gi * 6 + k];
  } else {
    fast::cov3d_from_scale_rot(s, mod, q, c6);
  }
  const float op_in = gin.op;
  float dmean[3] = {0.f, 0.f, 0.f}, dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, dop = 0.f, dcol[3] = {0.f, 0.f, 0.f};
  float dsh[K * 3];
#pragma unroll
  for (int k = 0; k < K * 3; ++k) dsh[k] = 0.f;
  const float fx = (float)W / (2.f * tanx), fy = (float)H / (2.f * tany);

  for (int vk = vslot; vk < (alive ? vpi : 0); vk += 4) {
    const int view = item * vpi + vk;
    const size_t g = pbase0 + (size_t)vk * P + i;
    // every load of this (view, Gaussian) is issued before anything is decided (this kernel is a chain of dependent memory round
    // trips -- triage word, flags, accumulators, camera -- not arithmetic: the accumulators of an untouched pair are zero anyway)
    const uint32_t cbits = clamped[g];
    const int32_t rad = radii[g];
    // scene level: this kernel is bound by the 80 B of f64 accumulators per (view, Gaussian) of every live Gaussian, most of whose
    // views never touched it -- decide first, then fetch the accumulators of the touched pairs only (one more dependent round trip)
    if ((flags & U3D_FLAG_INTERNAL_TRIAGE) && !(rad > 0 && (cbits & U3D_TOUCHED_BIT) != 0u)) {
      if (dL_dmeans2D) { dL_dmeans2D[g * 3] = 0.f; dL_dmeans2D[g * 3 + 1] = 0.f; dL_dmeans2D[g * 3 + 2] = 0.f; }
      continue;
    }
    double av[U3D_NACC];
#pragma unroll
    for (int k = 0; k < U3D_NACC; ++k) av[k] = acc[(size_t)k * NG + g];
    Cam cam;
    load_cam(cam, viewmatrix, projmatrix, campos, view);
    const bool live = rad > 0 && (cbits & U3D_TOUCHED_BIT) != 0u;   // visible AND handed a gradient by the reduction
    float a[U3D_NACC];
#pragma unroll
    for (int k = 0; k < U3D_NACC; ++k) a[k] = live ? (float)av[k] * gs : 0.f;
    if (acc_reset && live) {   // single-pass step: hand the accumulators back zeroed (only touched pairs were ever written)
#pragma unroll
      for (int k = 0; k < U3D_NACC; ++k) acc_reset[(size_t)k * NG + g] = 0.0;
    }
    if (dL_dmeans2D) {
      dL_dmeans2D[g * 3] = a[0]; dL_dmeans2D[g * 3 + 1] = a[1]; dL_dmeans2D[g * 3 + 2] = 0.f;
    }
    if (!live) continue;
    // no tile handed this (view, Gaussian) any gradient (it lies behind the saturation depth of every tile it touches --
    // the common case: a pixel saturates after a few dozen entries): every term below would be an exact zero
    bool nz = false;
#pragma unroll
    for (int k = 0; k < U3D_NACC; ++k) nz = nz || a[k] != 0.f;
    if (!nz) continue;
    Ewa e;
    fast::ewa_setup(e, p, cam, fx, fy, tanx, tany);
    float abc[3];
    fast::cov2d(e, c6, abc);
    float c_xx = abc[0], c_xy = abc[1], c_yy = abc[2];
    const float x0 = c_xx, y0 = c_yy, h_var = 0.3f;
    float d_inside_root = 0.f;
    float dop_v = a[5];
    if (flags & U3D_FLAG_ANTIALIASING) {
      const float det_cov = c_xx * c_yy - c_xy * c_xy;
      c_xx += h_var; c_yy += h_var;
      const float det_plus = c_xx * c_yy - c_xy * c_xy;
      const float ratio = det_cov / det_plus;
      const float hs = sqrtf(fmaxf(0.000025f, ratio));
      const float d_hs = dop_v * op_in;
      dop_v = dop_v * hs;
      d_inside_root = ratio <= 0.000025f ? 0.f : d_hs / (2.f * hs);
    } else {
      c_xx += h_var; c_yy += h_var;
    }
    dop += dop_v;
    float dcxx = 0.f, dcxy = 0.f, dcyy = 0.f;
    if (flags & U3D_FLAG_ANTIALIASING) {
      const bool exact = (flags & U3D_FLAG_EXACT_AA_GRAD) != 0;
      const float x = exact ? x0 : c_xx, y = exact ? y0 : c_yy, z = c_xy, w = h_var;
      const float dn = w * w + w * (x + y) + x * y - z * z;
      const float denom_f = d_inside_root / (dn * dn);
      dcxx = w * (w * y + y * y + z * z) * denom_f;
      dcyy = w * (w * x + x * x + z * z) * denom_f;
      dcxy = -2.f * w * z * (w + x + y) * denom_f;
    }
    const float dA = a[2], dBh = a[3], dC = a[4];
    const float denom = c_xx * c_yy - c_xy * c_xy;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dt[3] = {0.f, 0.f, 0.f};
    if (denom2inv != 0.f) {
      dcxx += denom2inv * (-c_yy * c_yy * dA + 2.f * c_xy * c_yy * dBh + (denom - c_xx * c_yy) * dC);
      dcyy += denom2inv * (-c_xx * c_xx * dC + 2.f * c_xx * c_xy * dBh + (denom - c_xx * c_yy) * dA);
      dcxy += denom2inv * 2.f * (c_xy * c_yy * dA - (denom + 2.f * c_xy * c_xy) * dBh + c_xx * c_xy * dC);
      const float Gc[4] = {dcxx, 0.5f * dcxy, 0.5f * dcxy, dcyy};
      float GM[6];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) GM[k * 3 + j] = Gc[k * 2] * e.M2[j] + Gc[k * 2 + 1] * e.M2[3 + j];
      // dSigma = M2^T Gc M2  (stored upper triangle: off-diagonals carry both symmetric entries)
      dcov[0] += e.M2[0] * GM[0] + e.M2[3] * GM[3];
      dcov[3] += e.M2[1] * GM[1] + e.M2[4] * GM[4];
      dcov[5] += e.M2[2] * GM[2] + e.M2[5] * GM[5];
      dcov[1] += 2.f * (e.M2[0] * GM[1] + e.M2[3] * GM[4]);
      dcov[2] += 2.f * (e.M2[0] * GM[2] + e.M2[3] * GM[5]);
      dcov[4] += 2.f * (e.M2[1] * GM[2] + e.M2[4] * GM[5]);
      // dM2 = 2 Gc M2 Sigma ; dJ = dM2 W^T
      const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
      float dM2[6];
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 3; ++j) dM2[k * 3 + j] = 2.f * (GM[k * 3] * S[j] + GM[k * 3 + 1] * S[3 + j] + GM[k * 3 + 2] * S[6 + j]);
      // dJ[k][l] = sum_m dM2[k][m] * W[l][m],  W[l][m] = V[m*4+l]
      const float dJ00 = dM2[0] * cam.V[0] + dM2[1] * cam.V[4] + dM2[2] * cam.V[8];
      const float dJ02 = dM2[0] * cam.V[2] + dM2[1] * cam.V[6] + dM2[2] * cam.V[10];
      const float dJ11 = dM2[3] * cam.V[1] + dM2[4] * cam.V[5] + dM2[5] * cam.V[9];
      const float dJ12 = dM2[3] * cam.V[2] + dM2[4] * cam.V[6] + dM2[5] * cam.V[10];
      const float tz = 1.f / e.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
      dt[0] = e.xmask * -fx * tz2 * dJ02;
      dt[1] = e.ymask * -fy * tz2 * dJ12;
      dt[2] = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * e.t[0]) * tz3 * dJ02 + (2.f * fy * e.t[1]) * tz3 * dJ12;
    }
    dt[2] -= a[9] / (e.t[2] * e.t[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) dmean[k] += cam.V[k * 4] * dt[0] + cam.V[k * 4 + 1] * dt[1] + cam.V[k * 4 + 2] * dt[2];
    // screen-space mean
    float hom[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hom[j] = cam.Pm[j] * p[0] + cam.Pm[4 + j] * p[1] + cam.Pm[8 + j] * p[2] + cam.Pm[12 + j];
    const float m_w = 1.0f / (hom[3] + 0.0000001f);
    const float mul1 = hom[0] * m_w * m_w, mul2 = hom[1] * m_w * m_w;
#pragma unroll
    for (int k = 0; k < 3; ++k)
      dmean[k] += (cam.Pm[k * 4] * m_w - cam.Pm[k * 4 + 3] * mul1) * a[0] + (cam.Pm[k * 4 + 1] * m_w - cam.Pm[k * 4 + 3] * mul2) * a[1];
    dcol[0] += a[6]; dcol[1] += a[7]; dcol[2] += a[8];
    if (shs) {
      // (coefficients 1.. : `sh` itself, or -- split SH -- features_rest re-based so that sh[3 + ...] addresses it)
      const float* sh = src.shs_rest ? src.shs_rest + gi * (size_t)src.s_shs_rest - 3 : shs + gi * (size_t)src.s_shs;
      const float dorig[3] = {p[0] - cam.pos[0], p[1] - cam.pos[1], p[2] - cam.pos[2]};
      const float sum2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
      const float inv = 1.f / sqrtf(sum2);
      const float x = dorig[0] * inv, y = dorig[1] * inv, z = dorig[2] * inv;
      const uint32_t cb = cbits;
      float ddir[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float gr = (cb >> c) & 1u ? 0.f : a[6 + c];
        dsh[c] += SH_C0 * gr;
        float dx_ = 0.f, dy_ = 0.f, dz_ = 0.f;
        if (D > 0) {
          dsh[3 + c] += -SH_C1 * y * gr;
          dsh[6 + c] += SH_C1 * z * gr;
          dsh[9 + c] += -SH_C1 * x * gr;
          dx_ = -SH_C1 * sh[9 + c]; dy_ = -SH_C1 * sh[3 + c]; dz_ = SH_C1 * sh[6 + c];
          if (D > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy_ = x * y, yz = y * z, xz = x * z;
            dsh[12 + c] += SH_C2[0] * xy_ * gr;
            dsh[15 + c] += SH_C2[1] * yz * gr;
            dsh[18 + c] += SH_C2[2] * (2.f * zz - xx - yy) * gr;
            dsh[21 + c] += SH_C2[3] * xz * gr;
            dsh[24 + c] += SH_C2[4] * (xx - yy) * gr;
            dx_ += SH_C2[0] * y * sh[12 + c] + SH_C2[2] * 2.f * -x * sh[18 + c] + SH_C2[3] * z * sh[21 + c] + SH_C2[4] * 2.f * x * sh[24 + c];
            dy_ += SH_C2[0] * x * sh[12 + c] + SH_C2[1] * z * sh[15 + c] + SH_C2[2] * 2.f * -y * sh[18 + c] + SH_C2[4] * 2.f * -y * sh[24 + c];
            dz_ += SH_C2[1] * y * sh[15 + c] + SH_C2[2] * 4.f * z * sh[18 + c] + SH_C2[3] * x * sh[21 + c];
            if (D > 2) {
              dsh[27 + c] += SH_C3[0] * y * (3.f * xx - yy) * gr;
              dsh[30 + c] += SH_C3[1] * xy_ * z * gr;
              dsh[33 + c] += SH_C3[2] * y * (4.f * zz - xx - yy) * gr;
              dsh[36 + c] += SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * gr;
              dsh[39 + c] += SH_C3[4] * x * (4.f * zz - xx - yy) * gr;
              dsh[42 + c] += SH_C3[5] * z * (xx - yy) * gr;
              dsh[45 + c] += SH_C3[6] * x * (xx - 3.f * yy) * gr;
              dx_ += SH_C3[0] * sh[27 + c] * 6.f * xy_ + SH_C3[1] * sh[30 + c] * yz + SH_C3[2] * sh[33 + c] * -2.f * xy_ +
                     SH_C3[3] * sh[36 + c] * -6.f * xz + SH_C3[4] * sh[39 + c] * (4.f * zz - 3.f * xx - yy) +
                     SH_C3[5] * sh[42 + c] * 2.f * xz + SH_C3[6] * sh[45 + c] * 3.f * (xx - yy);
              dy_ += SH_C3[0] * sh[27 + c] * 3.f * (xx - yy) + SH_C3[1] * sh[30 + c] * xz + SH_C3[2] * sh[33 + c] * (4.f * zz - xx - 3.f * yy) +
                     SH_C3[3] * sh[36 + c] * -6.f * yz + SH_C3[4] * sh[39 + c] * -2.f * xy_ + SH_C3[5] * sh[42 + c] * -2.f * yz +
                     SH_C3[6] * sh[45 + c] * -6.f * xy_;
              dz_ += SH_C3[1] * sh[30 + c] * xy_ + SH_C3[2] * sh[33 + c] * 8.f * yz + SH_C3[3] * sh[36 + c] * 3.f * (2.f * zz - xx - yy) +
                     SH_C3[4] * sh[39 + c] * 8.f * xz + SH_C3[5] * sh[42 + c] * (xx - yy);
            }
          }
        }
        ddir[0] += dx_ * gr; ddir[1] += dy_ * gr; ddir[2] += dz_ * gr;
      }
      const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
      dmean[0] += ((sum2 - dorig[0] * dorig[0]) * ddir[0] - dorig[1] * dorig[0] * ddir[1] - dorig[2] * dorig[0] * ddir[2]) * invsum32;
      dmean[1] += (-dorig[0] * dorig[1] * ddir[0] + (sum2 - dorig[1] * dorig[1]) * ddir[1] - dorig[2] * dorig[1] * ddir[2]) * invsum32;
      dmean[2] += (-dorig[0] * dorig[2] * ddir[0] - dorig[1] * dorig[2] * ddir[1] + (sum2 - dorig[2] * dorig[2]) * ddir[2]) * invsum32;
    }
  }

  // ---- sum the quad's view slots (two xor steps inside the DPP quad) ----
#define QUAD_SUM(v) do { (v) = u3d_quad_sum(v); } while (0)
#pragma unroll
  for (int k = 0; k < 3; ++k) { QUAD_SUM(dmean[k]); QUAD_SUM(dcol[k]); }
#pragma unroll
  for (int k = 0; k < 6; ++k) QUAD_SUM(dcov[k]);
  QUAD_SUM(dop);
#pragma unroll
  for (int k = 0; k < K * 3; ++k) QUAD_SUM(dsh[k]);
#undef QUAD_SUM

  // ---- outputs (strided like the source); act != 0 chains through the head activations ----
  float drot[4] = {0.f, 0.f, 0.f, 0.f}, dscale[3] = {0.f, 0.f, 0.f};
  if (src.scales) {
    // Sigma = Mx Mx^T, Mx = R diag(mod*s): exact derivative of the un-normalised quaternion polynomial
    const float Gs[9] = {dcov[0], 0.5f * dcov[1], 0.5f * dcov[2], 0.5f * dcov[1], dcov[3], 0.5f * dcov[4],
                         0.5f * dcov[2], 0.5f * dcov[4], dcov[5]};
    float R[9], Mx[9], sv[3];
    fast::quat_to_R(q, R);
#pragma unroll
    for (int k = 0; k < 3; ++k) sv[k] = mod * s[k];
#pragma unroll
    for (int r_ = 0; r_ < 3; ++r_)
#pragma unroll
      for (int k = 0; k < 3; ++k) Mx[r_ * 3 + k] = R[r_ * 3 + k] * sv[k];
    float dR[9];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float ssum = 0.f;
#pragma unroll
      for (int r_ = 0; r_ < 3; ++r_) {
        const float dM = 2.f * (Gs[r_ * 3] * Mx[k] + Gs[r_ * 3 + 1] * Mx[3 + k] + Gs[r_ * 3 + 2] * Mx[6 + k]);
        ssum += dM * R[r_ * 3 + k];
        dR[r_ * 3 + k] = dM * sv[k];
      }
      dscale[k] = mod * ssum;
    }
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    drot[0] = 2.f * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
    drot[1] = 2.f * (y * dR[1] + z * dR[2] + y * dR[3] - 2.f * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2.f * x * dR[8]);
    drot[2] = 2.f * (-2.f * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2.f * y * dR[8]);
    drot[3] = 2.f * (-2.f * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2.f * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
  }
  if (writer) {
    if (src.act != 0) {
      // tanh, sigmoid, exp(clamp) derivatives (model/gaussian_predictor.py:249-254)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        dmean[k] *= src.offset_scale * (1.f - gin.th[k] * gin.th[k]);
        dscale[k] *= (gin.raw_s[k] >= -1.f && gin.raw_s[k] <= 20.f) ? s[k] : 0.f;
      }
      dop *= op_in * (1.f - op_in);
      if (src.iso) { dscale[0] += dscale[1] + dscale[2]; dscale[1] = dscale[2] = 0.f; }   // one raw channel fed all three axes
      if (src.act == 2) {
        // per-quaternion normalise: d x_j = g_j / n - x_j (g . x) / n^3   (n > eps), g_j / eps otherwise
        const float n2 = gin.raw_q[0] * gin.raw_q[0] + gin.raw_q[1] * gin.raw_q[1] + gin.raw_q[2] * gin.raw_q[2] + gin.raw_q[3] * gin.raw_q[3];
        const float n = sqrtf(n2);
        const float dot = drot[0] * gin.raw_q[0] + drot[1] * gin.raw_q[1] + drot[2] * gin.raw_q[2] + drot[3] * gin.raw_q[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) drot[k] = n > 1e-6f ? drot[k] / n - gin.raw_q[k] * dot / (n2 * n) : drot[k] / 1e-6f;
      } else {
        // across-point normalise (object level): first term here, the column dot product is reduced below and the
        // second term is applied by quat_fixup_kernel
#pragma unroll
        for (int k = 0; k < 4; ++k) { qd[k] = drot[k] * gin.raw_q[k]; drot[k] = drot[k] / gin.qn[k]; }
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) sink.means[gi * src.s_means + k] = dmean[k];
    sink.opac[gi * src.s_opac] = dop;
    if (sink.colors) { sink.colors[gi * 3] = dcol[0]; sink.colors[gi * 3 + 1] = dcol[1]; sink.colors[gi * 3 + 2] = dcol[2]; }
    if (sink.cov) {
#pragma unroll
      for (int k = 0; k < 6; ++k) sink.cov[gi * 6 + k] = dcov[k];
    }
    if (sink.shs) {
      float* o = sink.shs + gi * (size_t)src.s_shs;
      if (sink.shs_rest) {
        float* orr = sink.shs_rest + gi * (size_t)src.s_shs_rest;
#pragma unroll
        for (int k = 0; k < K * 3; ++k) { if (k < 3) o[k] = dsh[k]; else orr[k - 3] = dsh[k]; }
        for (int k = K * 3; k < M * 3; ++k) orr[k - 3] = 0.f;
      } else {
#pragma unroll
        for (int k = 0; k < K * 3; ++k) o[k] = dsh[k];
        for (int k = K * 3; k < M * 3; ++k) o[k] = 0.f;
      }
    }
    if (sink.scales) {
#pragma unroll
      for (int k = 0; k < 3; ++k) sink.scales[gi * src.s_scales + k] = dscale[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) sink.rots[gi * src.s_rots + k] = drot[k];
    }
  }
  }
  if (src.act == 1) {
    // block reduction of sum_i raw_q[i][c] * g[i][c]; one atomic per block and component
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = u3d_wave_sum(qd[k]);
      if (lane == 0) s_qdot[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
      const float v = s_qdot[0][threadIdx.x] + s_qdot[1][threadIdx.x] + s_qdot[2][threadIdx.x] + s_qdot[3][threadIdx.x];
      unsafeAtomicAdd(&sink.qdot[item * 4 + threadIdx.x], v);
    }
  }
}

// across-point quaternion norms: qnorm[item][c] = || raw_rot[item, :, c] ||_2   (F.normalize(x(B,4,N), dim=-1))
constexpr int QN_THREADS = 1024;   // one workgroup per set: 16 waves keep more of the strided loads in flight (7.5 -> ~4 us at P = 2048)
__global__ __launch_bounds__(QN_THREADS) void quat_norms_kernel(U3DSpan span, const float* __restrict__ rots, int s_rots,
                                                                float* __restrict__ qnorm, float* __restrict__ qdot_zero) {
  __shared__ float sm[QN_THREADS / 64][4];
  const int item = blockIdx.x;
  int P;
  size_t gbase;
  u3d_set_span(span, item, P, gbase);
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < P; i += QN_THREADS) {
    const float* r = rots + (gbase + i) * s_rots;
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += r[k] * r[k];
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float v = u3d_wave_sum(a[k]);
    if (lane == 0) sm[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < QN_THREADS / 64; ++w) t += sm[w][threadIdx.x];
    qnorm[item * 4 + threadIdx.x] = sqrtf(t);
    if (qdot_zero) qdot_zero[item * 4 + threadIdx.x] = 0.f;
  }
}

// second term of the across-point normalise backward: d x_i -= x_i * (sum_j x_j g_j) / n^3  when n > eps
__global__ __launch_bounds__(U3D_BLOCK) void quat_fixup_kernel(U3DSpan span, const float* __restrict__ rots, int s_rots,
                                                               const float* __restrict__ qnorm, const float* __restrict__ qdot,
                                                               float* __restrict__ d_rots) {
  const int item = blockIdx.y;
  int P;
  size_t gbase;
  u3d_set_span(span, item, P, gbase);
  const int i = blockIdx.x * U3D_BLOCK + threadIdx.x;
  if (i >= P) return;
  const size_t o = (gbase + i) * s_rots;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float n = qnorm[item * 4 + k];
    if (n > 1e-6f) d_rots[o + k] -= rots[o + k] * qdot[item * 4 + k] / (n * n * n);
  }
}

__global__ void mark_visible_kernel(int P, const float* __restrict__ means3D, const float* __restrict__ V,
                                    uint8_t* __restrict__ present) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const float z = V[2] * means3D[i * 3] + V[6] * means3D[i * 3 + 1] + V[10] * means3D[i * 3 + 2] + V[14];
  present[i] = z > 0.2f ? 1 : 0;
}

}  // namespace

bool u3d_preprocess_sorts(const u3d_raster_desc& d) { return d.P <= U3D_BLOCK; }

void u3d_launch_preprocess_fwd(const u3d_raster_desc& d, const U3DBuffers& b, const U3DSource& src, const float* viewmatrix,
                               const float* projmatrix, const float* campos, int32_t* radii, double* acc_zero, hipStream_t s,
                               uint8_t* visible) {
  const bool fuse_sort = u3d_preprocess_sorts(d);
  const size_t NG = (size_t)d.views_per_item * u3d_total_P(d);
  // enough Gaussians to fill the chip by themselves -> one thread walks several views of its set (the view-independent work --
  // head activations, Sigma, SH fetch -- is done once per thread); otherwise one view per thread.  Four views per thread measured
  // best at scene level (C4: 25.4 us with 1 or 8, 21.2 with 2 or 4; C5: 49.9 / 39.0 / 35.7 / 37.0 us with 1 / 2 / 4 / 8): the thread's
  // view loop is a chain of dependent stores, and 314 workgroups (C4 with all 8 views per thread) leave most of the chip idle
  const int vpt = (u3d_total_P(d) >= 65536) ? (d.views_per_item < 4 ? d.views_per_item : 4) : 1;
  const int chunks = (d.views_per_item + vpt - 1) / vpt;
  dim3 grid((d.P + U3D_BLOCK - 1) / U3D_BLOCK, d.n_items, chunks), block(U3D_BLOCK);
  const size_t lds = src.act != 0 ? (size_t)U3D_BLOCK * src.s_means * sizeof(float) : 0;
  const int D = src.shs ? d.sh_degree : 0;
#define LAUNCH(DEG)                                                                                                   \
  hipLaunchKernelGGL(preprocess_fwd_kernel<DEG>, grid, block, lds, s, u3d_span(d), d.views_per_item, vpt, d.image_height, \
                     d.image_width, d.tanfovx, d.tanfovy, d.scale_modifier, d.flags, src, viewmatrix, projmatrix, campos, \
                     radii, b.depth, b.xy, b.conic_op, b.rgbd, b.rect, b.clamped, b.num_rendered, acc_zero, NG,               \
                     fuse_sort ? b.sorted_id : nullptr, b.sorted_rect, b.n_vis, d.P > U3D_LDS_SORT_MAX ? b.sort_hist : nullptr, u3d_msd_bins(d.P), \
                     u3d_uses_touched_words(d) ? b.touched_words : nullptr, visible,                                      \
                     u3d_uses_touched_words(d) ? b.touched_count : nullptr,                                              \
                     u3d_sparse_bwd(d, src.act) ? 0u : U3D_TOUCHED_NOT_LISTED)
  switch (D) {
    case 0: LAUNCH(0); break;
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    default: LAUNCH(3); break;
  }
#undef LAUNCH
}

void u3d_launch_preprocess_bwd(const u3d_raster_desc& d, const U3DBuffers& b, const U3DSource& src, const float* viewmatrix,
                               const float* projmatrix, const float* campos, const int32_t* radii, const double* acc,
                               const U3DGradSink& sink, hipStream_t s, double* acc_reset, const float* gscale, bool sparse) {
  const size_t NG = (size_t)d.views_per_item * u3d_total_P(d);
  dim3 grid((d.P + U3D_BLOCK / 4 - 1) / (U3D_BLOCK / 4), d.n_items), block(U3D_BLOCK);
  // sparse: one linear grid over the touched list, sized for the worst case (every Gaussian touched); workgroups past the
  // device-side count leave at once
  // (n_items x ceil(P / 64) >= ceil(total_P / 64) blocks: also enough for the dense walk the kernel falls back to when the forward half
  // did not list)
  if (sparse) grid = dim3((unsigned)(grid.x * grid.y), 1);
  const int D = src.shs ? d.sh_degree : 0;
  const int flags = d.flags | (d.P > U3D_LDS_SORT_MAX ? U3D_FLAG_INTERNAL_TRIAGE : 0);   // scene level: most waves only write zeros
  const uint2* sl = sparse ? b.touched_list : nullptr;
  const uint32_t* sc = sparse ? b.touched_count : nullptr;
#define LAUNCH(DEG)                                                                                                    \
  hipLaunchKernelGGL(preprocess_bwd_kernel<DEG>, grid, block, 0, s, u3d_span(d), d.views_per_item, d.sh_coeffs, d.image_height, \
                     d.image_width, d.tanfovx, d.tanfovy, d.scale_modifier, flags, NG, src, viewmatrix, projmatrix,    \
                     campos, radii, b.clamped, acc, acc_reset, sink, gscale, b.touched_words, sl, sc)
  switch (D) {
    case 0: LAUNCH(0); break;
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    default: LAUNCH(3); break;
  }
#undef LAUNCH
}

void u3d_launch_quat_norms(const u3d_raster_desc& d, const float* rots, int s_rots, float* qnorm, float* qdot_zero, hipStream_t s) {
  hipLaunchKernelGGL(quat_norms_kernel, dim3(d.n_items), dim3(QN_THREADS), 0, s, u3d_span(d), rots, s_rots, qnorm, qdot_zero);
}

void u3d_launch_quat_fixup(const u3d_raster_desc& d, const float* rots, int s_rots, const float* qnorm, const float* qdot,
                           float* d_rots, hipStream_t s) {
  hipLaunchKernelGGL(quat_fixup_kernel, dim3((d.P + U3D_BLOCK - 1) / U3D_BLOCK, d.n_items), dim3(U3D_BLOCK), 0, s, u3d_span(d), rots,
                     s_rots, qnorm, qdot, d_rots);
}

void u3d_launch_mark_visible(int P, const float* means3D, const float* viewmatrix, uint8_t* present, hipStream_t s) {
  if (P <= 0) return;
  hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, viewmatrix, present);
}
