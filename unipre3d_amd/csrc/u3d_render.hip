// Tile compositing kernels for gfx950 (wave64, LDS-staged, no instance lists in HBM).
//
// ONE WAVE renders one 16x16 tile, 4 pixels per lane (column lane&15, rows (lane>>4)+4k); a workgroup is four independent
// tiles, LDS regions are wave-private and no kernel here has a workgroup barrier in its main loop.  The wave walks the VIEW's
// depth-sorted Gaussian list (u3d_sort.hip) in batches of 64: every lane tests one sorted entry's tile rectangle against this
// tile, hits are compacted in order into LDS with a ballot + popcount prefix (the "duplicate / sort / range" stages of the
// original operator collapse into this filter), and the batch is blended front to back from LDS broadcast reads (SURVEY.md
// R4 steps 9-10).  The wave leaves as soon as all of its 256 pixels are saturated: with the reference's large, fairly opaque
// splats only ~17 sorted entries are ever touched per pixel.
//
// Backward (R5/R6) re-stages the batches back to front, recovers T by division, accumulates each Gaussian's gradients as
// moments over the lane's 4 pixels, reduces them across the wave with ONE interleaved DPP tree per component, and hands the
// tile's rows to a fixed-order f64 reduction (the original: one fp32 atomic per pixel per component, non-deterministic).
// render_fb_wave_kernel does forward and backward of the fused render-loss training step in a single pass.
#include "u3d_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;

__device__ __forceinline__ bool rect_hits(const uint2 r, int tx, int ty) {
  const int x0 = r.x & 0xffffu, y0 = r.x >> 16, x1 = r.y & 0xffffu, y1 = r.y >> 16;
  return tx >= x0 && tx < x1 && ty >= y0 && ty < y1;
}

// per-pixel loss term and its weight (utils/loss_utils.py:20-45).  torch.isclose(gt, bg, atol=1e-6) uses rtol=1e-5.
__device__ __forceinline__ float focal_weight(const U3DLoss& L, const float* __restrict__ bg, float g0, float g1, float g2) {
  if (L.kind != 2) return 1.f;
  const bool is_bg = fabsf(g0 - bg[0]) <= 1e-6f + 1e-5f * fabsf(bg[0]) && fabsf(g1 - bg[1]) <= 1e-6f + 1e-5f * fabsf(bg[1]) &&
                     fabsf(g2 - bg[2]) <= 1e-6f + 1e-5f * fabsf(bg[2]);
  return is_bg ? L.w_bg : L.w_non;
}
__device__ __forceinline__ float loss_pixel(const U3DLoss& L, const float* __restrict__ bg, float g0, float g1, float g2,
                                            float d0, float d1, float d2) {
  if (L.kind == 3) return fabsf(d0) + fabsf(d1) + fabsf(d2);
  return focal_weight(L, bg, g0, g1, g2) * (d0 * d0 + d1 * d1 + d2 * d2);
}


// ---- lane-mask helpers -------------------------------------------------------------------------------
// Compare results are kept as 64-bit lane masks on the scalar unit (v_cmp -> SGPR pair, s_and/s_or), and applied with
// one v_cndmask; hipcc's own lowering of bool && / ballot costs two extra VALU instructions per use.
#define U3D_FCMP_OGE 3
#define U3D_FCMP_OLT 4
#define U3D_FCMP_OLE 5
#define U3D_ICMP_ULT 36
typedef unsigned long long lanemask_t;
__device__ __forceinline__ float mask_sel0(lanemask_t m, float a) {   // lane in m ? a : 0
  float r;
  asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ float mask_sel(lanemask_t m, float a, float b) {   // lane in m ? a : b
  float r;
  asm("v_cndmask_b32_e64 %0, %2, %1, %3" : "=v"(r) : "v"(a), "v"(b), "s"(m));
  return r;
}
__device__ __forceinline__ float min_099(float a) {   // fminf(0.99f, a) without the canonicalising v_max
  float r;
  asm("v_min_f32_e32 %0, 0x3f7d70a4, %1" : "=v"(r) : "v"(a));
  return r;
}

// ---- forward, wave-per-tile form ---------------------------------------------------------------
// One WAVE renders one 16x16 tile, 4 pixels per lane (column lane&15, rows (lane>>4)+4k): the per-Gaussian LDS
// broadcast reads, loop control and exponent set-up are shared by 4 pixels (4-way ILP), batches are 64 sorted
// entries (the reference's splats saturate a pixel after ~17 entries, so one batch usually suffices), LDS regions
// are wave-private and there is no workgroup barrier.  A workgroup is four independent tiles.
__global__ __launch_bounds__(U3D_BLOCK) void render_fwd_wave_kernel(
    int P, int H, int W, int tiles_x, int T, uint32_t ntiles_total, uint32_t nblocks, const uint32_t* __restrict__ sorted_id,
    const uint2* __restrict__ sorted_rect, const uint32_t* __restrict__ n_vis, const float2* __restrict__ xy,
    const float4* __restrict__ conic_op, const float4* __restrict__ rgbd, const float* __restrict__ bg,
    float* __restrict__ out_color, float* __restrict__ out_invdepth, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, U3DLoss loss) {
  __shared__ float4 sA[4][U3D_WAVE];   // x, y, -0.5*log2e*a, -log2e*b
  __shared__ float4 sB[4][U3D_WAVE];   // -0.5*log2e*c, opacity, 1/depth, pos (bits)
  __shared__ float4 sC[4][U3D_WAVE];   // r, g, b, -

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t lid = u3d_xcd_remap(blockIdx.x, nblocks) * 4u + (uint32_t)wave;
  if (lid >= ntiles_total) return;
  const int view = lid / T, tile = lid - view * T;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int px = tx * U3D_TILE + (lane & 15);
  const int py0 = ty * U3D_TILE + (lane >> 4);
  const float pxf = (float)px;
  const size_t vbase = (size_t)view * P;
  const uint32_t nv = n_vis[view];

  float Tr[4], C0[4], C1[4], C2[4], Dv[4], pyf[4];
  uint32_t last[4];
  bool done[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int py = py0 + 4 * k;
    pyf[k] = (float)py;
    Tr[k] = 1.f; C0[k] = C1[k] = C2[k] = Dv[k] = 0.f; last[k] = 0u;
    done[k] = !(px < W && py < H);
  }
  bool all_done = done[0] && done[1] && done[2] && done[3];

  for (uint32_t base = 0; base < nv; base += U3D_WAVE) {
    if (__ballot(!all_done) == 0ull) break;
    const uint32_t s = base + (uint32_t)lane;
    bool hit = false;
    if (s < nv) hit = rect_hits(sorted_rect[vbase + s], tx, ty);
    const unsigned long long bal = __ballot(hit);
    const int total = __popcll(bal);
    if (total == 0) continue;
    if (hit) {
      const uint32_t o = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      const size_t g = vbase + sorted_id[vbase + s];
      const float2 m = xy[g];
      const float4 co = conic_op[g];
      const float4 cd = rgbd[g];
      sA[wave][o] = make_float4(m.x, m.y, -0.5f * LOG2E * co.x, -LOG2E * co.y);
      sB[wave][o] = make_float4(-0.5f * LOG2E * co.z, co.w, 1.0f / cd.w, __uint_as_float(s + 1u));
      sC[wave][o] = make_float4(cd.x, cd.y, cd.z, 0.f);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int j = 0; j < total; ++j) {
      const float4 A = sA[wave][j];
      const float4 B = sB[wave][j];
      const float4 Cc = sC[wave][j];
      const float dx = A.x - pxf;
      const float adx = A.z * dx, bdx = A.w * dx;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dy = A.y - pyf[k];
        const float pw = fmaf(adx, dx, fmaf(B.x * dy, dy, bdx * dy));
        const float alpha = fminf(0.99f, B.y * __builtin_amdgcn_exp2f(pw));
        const bool ok = !done[k] && pw <= 0.f && alpha >= ALPHA_MIN;
        const float test_T = Tr[k] * (1.f - alpha);
        const bool stop = ok && test_T < T_STOP;
        done[k] = done[k] || stop;
        if (ok && !stop) {
          const float w = alpha * Tr[k];
          C0[k] = fmaf(Cc.x, w, C0[k]);
          C1[k] = fmaf(Cc.y, w, C1[k]);
          C2[k] = fmaf(Cc.z, w, C2[k]);
          Dv[k] = fmaf(B.z, w, Dv[k]);
          Tr[k] = test_T;
          last[k] = __float_as_uint(B.w);
        }
      }
      all_done = done[0] && done[1] && done[2] && done[3];
      if (__ballot(!all_done) == 0ull) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const size_t npix = (size_t)H * W;
  float e = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int py = py0 + 4 * k;
    if (px < W && py < H) {
      const size_t pid = (size_t)py * W + px;
      final_T[(size_t)view * npix + pid] = Tr[k];
      n_contrib[(size_t)view * npix + pid] = last[k];
      const float o0 = fmaf(Tr[k], bg[0], C0[k]), o1 = fmaf(Tr[k], bg[1], C1[k]), o2 = fmaf(Tr[k], bg[2], C2[k]);
      float* oc = out_color + (size_t)view * 3 * npix + pid;
      oc[0] = o0; oc[npix] = o1; oc[2 * npix] = o2;
      if (out_invdepth) out_invdepth[(size_t)view * npix + pid] = Dv[k];
      if (loss.kind != 0) {
        const float* gp = loss.gt + (size_t)view * 3 * npix + pid;
        const float g0 = gp[0], g1 = gp[npix], g2 = gp[2 * npix];
        e += loss_pixel(loss, bg, g0, g1, g2, o0 - g0, o1 - g1, o2 - g2);
      }
    }
  }
  if (loss.kind != 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
    if (lane == 0) loss.partial[lid] = e;
  }
}

// ---- wave64 sum via DPP: result valid in lane 63 ---------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int r = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(r);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror      -> every lane holds its 16-lane row sum
  v = dpp_add<0x142, 0xa>(v);  // row_bcast15 into rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast31 into rows 2,3 -> lane 63 holds the wave sum
  return v;
}

// Per-Gaussian gradient rows travel as RAW moments (mx, my, mxx, mxy, myy, m0, r, g, b, d); they are linear in the pixels, so
// the conversion to accumulator values (which needs the Gaussian's own conic / opacity) is applied once after summation:
//   dL/dmean2D = -(W/2, H/2) * (a mx + b my, c my + b mx),  dL/dconic = -1/2 (mxx, mxy, myy),  dL/dopacity = m0 / opacity.
template <typename T>
__device__ __forceinline__ T moment_to_acc(int k, const T* m, float a, float b, float c, float op, float half_w, float half_h) {
  switch (k) {
    case 0: return -(T)half_w * ((T)a * m[0] + (T)b * m[1]);
    case 1: return -(T)half_h * ((T)c * m[1] + (T)b * m[0]);
    case 2: return (T)-0.5 * m[2];
    case 3: return (T)-0.5 * m[3];
    case 4: return (T)-0.5 * m[4];
    case 5: return m[5] != (T)0 ? m[5] / (T)op : (T)0;
    default: return m[k];
  }
}

// ---- backward, wave-per-tile form --------------------------------------------------------------
// One WAVE owns one 16x16 tile (4 pixels per lane: column lane&15, rows (lane>>4) + 4k); a workgroup is
// BWD_WAVES independent tiles: the batch loop has no workgroup barrier, LDS regions are wave-private.  Per
// Gaussian the gradients are accumulated as MOMENTS of q = dL/dG * G over the pixel offsets,
//     m0 = sum q, mx = sum q dx, my = sum q dy, mxx = sum q dx^2, mxy = sum q dx dy, myy = sum q dy^2,
// first across the lane's 4 pixels (plain FMAs), then ONE DPP tree per component per tile.  Lane 63 turns the
// moments into the accumulator values:
//     dL/dmean2D = -(W/2, H/2) * (a mx + b my, c my + b mx),  dL/dconic = -1/2 (mxx, mxy, myy),
//     dL/dopacity = m0 / opacity.
// Cross-tile accumulation without atomics in the common case: the first 64 positions of the view's sorted list
// -- where the reference's large, fairly opaque splats put essentially all contributions (a pixel saturates after
// ~17 entries) -- are written per tile to a partial buffer part[view][tile][component][position] (plain coalesced
// stores) and summed over the tiles in a FIXED order, in f64, by bwd_reduce_kernel.  Only sorted positions >= 64
// (sparse / semi-transparent scenes) fall back to f64 global atomics, whose ordering does not show at fp32 output
// precision.  The result is therefore run-to-run deterministic, unlike the original's fp32 atomics.
constexpr int BWD_WAVES = 4;
constexpr int BWD_PART_STRIDE = U3D_NACC * U3D_WAVE + 16;   // floats per tile slot (+ row count, 64-B aligned)
constexpr int BWD_REDUCE_SPLIT = 8;                          // workgroups per view in bwd_reduce_kernel

template <bool HAS_INVD>
__global__ __launch_bounds__(BWD_WAVES * U3D_WAVE, 4) void render_bwd_wave_kernel(
    int P, int H, int W, int tiles_x, int T, uint32_t ntiles_total, uint32_t nblocks, size_t NG,
    const uint32_t* __restrict__ sorted_id, const uint2* __restrict__ sorted_rect, const float2* __restrict__ xy,
    const float4* __restrict__ conic_op, const float4* __restrict__ rgbd, const float* __restrict__ bg,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_dinvdepth, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, double* __restrict__ acc, float* __restrict__ part,
    const float* __restrict__ out_color, U3DLoss loss) {
  constexpr int NK = HAS_INVD ? U3D_NACC : U3D_NACC - 1;
  __shared__ float4 sA[BWD_WAVES][U3D_WAVE];   // x, y, a, b
  __shared__ float4 sB[BWD_WAVES][U3D_WAVE];   // c, opacity, 1/depth, pos (bits)
  __shared__ float4 sC[BWD_WAVES][U3D_WAVE];   // r, g, b, id (bits)
  __shared__ float4 sAcc[BWD_WAVES][U3D_WAVE][3];   // per slot: {mx,my,mxx,mxy} {myy,m0,r,g} {b,d,-,-}

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t lid = u3d_xcd_remap(blockIdx.x, nblocks) * (uint32_t)BWD_WAVES + (uint32_t)wave;
  if (lid >= ntiles_total) return;   // whole wave leaves; there is no workgroup barrier below
  const int view = (int)(lid / T);
#pragma unroll
  for (int k = 0; k < 3; ++k) sAcc[wave][lane][k] = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t wmax_all = 0;

  {
    const int tile = (int)lid - view * T;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int px = tx * U3D_TILE + (lane & 15);
    const int py0 = ty * U3D_TILE + (lane >> 4);
    const float pxf = (float)px;
    const size_t vbase = (size_t)view * P;
    const size_t npix = (size_t)H * W;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

    float T_final[4], Tr[4], dp0[4], dp1[4], dp2[4], dinv[4], bg_dot[4], pyf[4];
    // "behind" recurrences  a <- la*l + (1-la)*a  kept as  a <- fma(oml, a, u)  with u = la*l, oml = 1-la
    float ar0[4], ar1[4], ar2[4], u0[4], u1[4], u2[4], oml[4], tfb[4], ainv[4], uinv[4];
    uint32_t last[4];
    uint32_t wmax = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int py = py0 + 4 * k;
      pyf[k] = (float)py;
      const bool inside = px < W && py < H;
      const size_t pid = (size_t)py * W + px;
      T_final[k] = inside ? final_T[(size_t)view * npix + pid] : 0.f;
      last[k] = inside ? n_contrib[(size_t)view * npix + pid] : 0u;
      dp0[k] = dp1[k] = dp2[k] = dinv[k] = 0.f;
      if (inside) {
        if (loss.kind != 0) {
          const float* xp = out_color + (size_t)view * 3 * npix + pid;
          const float* gp = loss.gt + (size_t)view * 3 * npix + pid;
          const float g0 = gp[0], g1 = gp[npix], g2 = gp[2 * npix];
          const float d0 = xp[0] - g0, d1 = xp[npix] - g1, d2 = xp[2 * npix] - g2;
          const float sc = loss.dloss[0] * loss.inv_count;
          if (loss.kind == 3) {
            dp0[k] = sc * (d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f));
            dp1[k] = sc * (d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f));
            dp2[k] = sc * (d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f));
          } else {
            const float w2 = 2.f * sc * focal_weight(loss, bg, g0, g1, g2);
            dp0[k] = w2 * d0; dp1[k] = w2 * d1; dp2[k] = w2 * d2;
          }
        } else {
          const float* dc = dL_dcolor + (size_t)view * 3 * npix + pid;
          dp0[k] = dc[0]; dp1[k] = dc[npix]; dp2[k] = dc[2 * npix];
          if (HAS_INVD) dinv[k] = dL_dinvdepth[(size_t)view * npix + pid];
        }
      }
      bg_dot[k] = bg[0] * dp0[k] + bg[1] * dp1[k] + bg[2] * dp2[k];
      Tr[k] = T_final[k];
      ar0[k] = ar1[k] = ar2[k] = u0[k] = u1[k] = u2[k] = ainv[k] = uinv[k] = 0.f;
      oml[k] = 1.f;
      tfb[k] = T_final[k] * bg_dot[k];
      wmax = max(wmax, last[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, o));

    wmax_all = wmax;
    const int nb = (int)((wmax + U3D_WAVE - 1) / U3D_WAVE);
    for (int b = nb - 1; b >= 0; --b) {
      const uint32_t s = (uint32_t)b * U3D_WAVE + (uint32_t)lane;
      bool hit = false;
      if (s < wmax) hit = rect_hits(sorted_rect[vbase + s], tx, ty);
      const unsigned long long bal = __ballot(hit);
      const int total = __popcll(bal);
      if (total == 0) continue;
      if (hit) {
        const uint32_t o = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        const uint32_t id = sorted_id[vbase + s];
        const size_t g = vbase + id;
        const float2 m = xy[g];
        const float4 co = conic_op[g];
        const float4 cd = rgbd[g];
        sA[wave][o] = make_float4(m.x, m.y, co.x, co.y);
        sB[wave][o] = make_float4(co.z, co.w, 1.0f / cd.w, __uint_as_float(s + 1u));
        sC[wave][o] = make_float4(cd.x, cd.y, cd.z, __uint_as_float(id));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      for (int j = total - 1; j >= 0; --j) {
        const float4 A = sA[wave][j];
        const float4 B = sB[wave][j];
        const float4 Cc = sC[wave][j];
        const uint32_t pos = __float_as_uint(B.w);
        const float dx = A.x - pxf;
        float m0 = 0.f, mx = 0.f, my = 0.f, mxx = 0.f, mxy = 0.f, myy = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f, g_d = 0.f;
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float dy = A.y - pyf[k];
          const float pw = fmaf(-0.5f * LOG2E * A.z * dx, dx, fmaf(-0.5f * LOG2E * B.x * dy, dy, -LOG2E * A.w * dx * dy));
          const float G = __builtin_amdgcn_exp2f(pw);
          const float alpha = fminf(0.99f, B.y * G);
          const bool ok = pos <= last[k] && pw <= 0.f && alpha >= ALPHA_MIN;
          if (ok) {
            any = true;
            const float om = 1.f - alpha;
            const float rc = __builtin_amdgcn_rcpf(om);
            Tr[k] = Tr[k] * rc;
            const float w = alpha * Tr[k];
            ar0[k] = fmaf(oml[k], ar0[k], u0[k]); u0[k] = alpha * Cc.x;
            ar1[k] = fmaf(oml[k], ar1[k], u1[k]); u1[k] = alpha * Cc.y;
            ar2[k] = fmaf(oml[k], ar2[k], u2[k]); u2[k] = alpha * Cc.z;
            float dL_dalpha = (Cc.x - ar0[k]) * dp0[k] + (Cc.y - ar1[k]) * dp1[k] + (Cc.z - ar2[k]) * dp2[k];
            g_r = fmaf(w, dp0[k], g_r); g_g = fmaf(w, dp1[k], g_g); g_b = fmaf(w, dp2[k], g_b);
            if (HAS_INVD) {
              ainv[k] = fmaf(oml[k], ainv[k], uinv[k]); uinv[k] = alpha * B.z;
              dL_dalpha += (B.z - ainv[k]) * dinv[k];
              g_d = fmaf(w, dinv[k], g_d);
            }
            oml[k] = om;
            dL_dalpha = fmaf(dL_dalpha, Tr[k], -(tfb[k] * rc));
            const float q = B.y * dL_dalpha * G;    // dL/dG * G
            const float qdx = q * dx, qdy = q * dy;
            m0 += q; mx += qdx; my += qdy;
            mxx = fmaf(qdx, dx, mxx); mxy = fmaf(qdx, dy, mxy); myy = fmaf(qdy, dy, myy);
          }
        }
        if (__ballot(any) == 0ull) continue;
        // nine interleaved wave reductions, one v_add_f32_dpp per value and level (hipcc does not fuse
        // update_dpp + fadd: -0.0 rule); interleaving keeps dependent DPP ops >= 8 instructions apart, so the
        // VALU-write -> DPP-read hazard needs no wait states inside the asm.
#define U3D_DPP9(CTRL)                                                                                              \
  asm volatile("s_nop 1\n\t"                                                                                        \
               "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\t"                        \
               "v_add_f32_dpp %2, %2, %2 " CTRL "\n\tv_add_f32_dpp %3, %3, %3 " CTRL "\n\t"                        \
               "v_add_f32_dpp %4, %4, %4 " CTRL "\n\tv_add_f32_dpp %5, %5, %5 " CTRL "\n\t"                        \
               "v_add_f32_dpp %6, %6, %6 " CTRL "\n\tv_add_f32_dpp %7, %7, %7 " CTRL "\n\t"                        \
               "v_add_f32_dpp %8, %8, %8 " CTRL "\n\ts_nop 1"                                                       \
               : "+v"(m0), "+v"(mx), "+v"(my), "+v"(mxx), "+v"(mxy), "+v"(myy), "+v"(g_r), "+v"(g_g), "+v"(g_b))
        U3D_DPP9("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
        U3D_DPP9("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
        U3D_DPP9("row_half_mirror row_mask:0xf bank_mask:0xf");
        U3D_DPP9("row_mirror row_mask:0xf bank_mask:0xf");
        U3D_DPP9("row_bcast:15 row_mask:0xa bank_mask:0xf");
        U3D_DPP9("row_bcast:31 row_mask:0xc bank_mask:0xf");
#undef U3D_DPP9
        if (HAS_INVD) g_d = wave_sum_to_lane63(g_d);
        if (lane == 63) {
          // batch 0 is indexed by sorted position (merged across tiles below), later batches by compaction slot
          const int slot = b == 0 ? (int)pos - 1 : j;
          sAcc[wave][slot][0] = make_float4(mx, my, mxx, mxy);
          sAcc[wave][slot][1] = make_float4(myy, m0, g_r, g_g);
          sAcc[wave][slot][2] = make_float4(g_b, g_d, 0.f, 0.f);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (b > 0) {
        if (lane < total) {
          const size_t g = vbase + __float_as_uint(sC[wave][lane].w);
          const float4 m0v = sAcc[wave][lane][0], m1v = sAcc[wave][lane][1], m2v = sAcc[wave][lane][2];
          const float4 Ag = sA[wave][lane], Bg = sB[wave][lane];
          const float m[U3D_NACC] = {m0v.x, m0v.y, m0v.z, m0v.w, m1v.x, m1v.y, m1v.z, m1v.w, m2v.x, m2v.y};
#pragma unroll
          for (int k = 0; k < NK; ++k) {
            const float v = moment_to_acc<float>(k, m, Ag.z, Ag.w, Bg.x, Bg.y, ddelx_dx, ddely_dy);
            if (v != 0.f) unsafeAtomicAdd(&acc[(size_t)k * NG + g], (double)v);
          }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) sAcc[wave][lane][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  // positions 0..63 of this tile: plain coalesced stores, reduced over the tiles by bwd_reduce_kernel
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // (only the rows this tile can have touched: positions < min(wmax, 64); the count is the last word of the slot)
  float* pt = part + (size_t)lid * BWD_PART_STRIDE;
  const uint32_t cnt = min(wmax_all, (uint32_t)U3D_WAVE);
  if ((uint32_t)lane < cnt) {
#pragma unroll
    for (int k = 0; k < NK; ++k) pt[k * U3D_WAVE + lane] = reinterpret_cast<const float*>(&sAcc[wave][lane][0])[k];   // raw moments
  }
  if (lane == 0) reinterpret_cast<uint32_t*>(pt)[U3D_NACC * U3D_WAVE] = cnt;
}

// ---- forward + backward in ONE kernel (training step of the fused render-loss path) -----------------
// The render loss needs nothing but the pixel's own colour and gt, so a tile can blend front to back, evaluate its
// loss term and seed dL/dcolor, and immediately walk the same LDS-resident batch back to front: final_T, n_contrib and
// the colour image never round-trip through HBM, the Gaussian batch is staged once, and one prologue disappears.
// dL/dloss is taken as 1 (the result is linear in it; the host scales the stored gradient).
//
// Lane layout: lane>>2 is the tile ROW, the lane's 4 pixels are the consecutive COLUMNS 4*(lane&3)+k, so image rows
// move as one 16-byte access per lane and channel, and a DPP quad (4 lanes) is one 16-pixel tile row: everything a
// quad sums shares dy.  The kernel is VALU-issue bound, so the arithmetic is arranged for instruction count:
//   * exponent  pw = (a' dx + b' dy) dx + c' dy^2  with the -log2e/2 factors folded in at staging (3 ops/pixel);
//   * the "colour behind" recurrence runs on the scalar  A = sum_c behind_c * dL/dC_c  (dL/dC is constant per pixel),
//     not per channel; the update tolerates alpha = 0, so non-contributing pixels take the same straight-line code
//     with their alpha selected to zero (no per-pixel branches, no copies);
//   * only m0, mx, mxx and the colour gradient are accumulated per pixel; after the two quad levels of the
//     reduction  my = dy m0, mxy = dy mx, myy = dy my;
//   * the half-row and row levels use DPP bank masks to deposit two values into one register per instruction pair
//     (9 values -> 5 -> 3 registers), and the four rows of the wave meet in three LDS float adds.
__global__ __launch_bounds__(BWD_WAVES * U3D_WAVE, 4) void render_fb_wave_kernel(
    int P, int H, int W, int tiles_x, int T, uint32_t ntiles_total, uint32_t nblocks, size_t NG,
    const uint32_t* __restrict__ sorted_id, const uint2* __restrict__ sorted_rect, const uint32_t* __restrict__ n_vis,
    const float2* __restrict__ xy, const float4* __restrict__ conic_op, const float4* __restrict__ rgbd,
    const float* __restrict__ bg, float* __restrict__ out_color, double* __restrict__ acc, float* __restrict__ part,
    U3DLoss loss) {
  constexpr int NK = U3D_NACC - 1;
  __shared__ float4 sA[BWD_WAVES][U3D_WAVE];   // x, y, a' = -log2e/2 a, b' = -log2e b
  __shared__ float4 sB[BWD_WAVES][U3D_WAVE];   // c' = -log2e/2 c, opacity, -, pos (bits)
  __shared__ float4 sC[BWD_WAVES][U3D_WAVE];   // r, g, b, id (bits)
  __shared__ float4 sAcc[BWD_WAVES][U3D_WAVE][3];   // per slot: {mx,my,mxx,mxy} {myy,m0,r,g} {b,-,-,-}

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t lid = u3d_xcd_remap(blockIdx.x, nblocks) * (uint32_t)BWD_WAVES + (uint32_t)wave;
  if (lid >= ntiles_total) return;
  const int view = (int)(lid / T);
  const int tile = (int)lid - view * T;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int py = ty * U3D_TILE + (lane >> 2);
  const int px0 = tx * U3D_TILE + 4 * (lane & 3);
  const float pyf = (float)py;
  const size_t vbase = (size_t)view * P;
  const size_t npix = (size_t)H * W;
  const uint32_t nv = n_vis[view];
#pragma unroll
  for (int k = 0; k < 3; ++k) sAcc[wave][lane][k] = make_float4(0.f, 0.f, 0.f, 0.f);

  // stage sorted entries [b*64, b*64+64) limited to `limit`; returns the hit ballot
  auto stage = [&](int b, uint32_t limit) -> unsigned long long {
    const uint32_t s = (uint32_t)b * U3D_WAVE + (uint32_t)lane;
    bool hit = false;
    if (s < limit) hit = rect_hits(sorted_rect[vbase + s], tx, ty);
    const unsigned long long bal = __ballot(hit);
    if (hit) {
      const uint32_t o = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      const uint32_t id = sorted_id[vbase + s];
      const size_t g = vbase + id;
      const float2 m = xy[g];
      const float4 co = conic_op[g];
      const float4 cd = rgbd[g];
      sA[wave][o] = make_float4(m.x, m.y, (-0.5f * LOG2E) * co.x, -LOG2E * co.y);
      sB[wave][o] = make_float4((-0.5f * LOG2E) * co.z, co.w, 0.f, __uint_as_float(s + 1u));
      sC[wave][o] = make_float4(cd.x, cd.y, cd.z, __uint_as_float(id));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return bal;
  };

  float pxf[4], Tr[4], C0[4], C1[4], C2[4], amin[4];
  uint32_t stop_pos[4];   // sorted position at which the pixel saturated (entries from there on are not blended)
  bool inside[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pxf[k] = (float)(px0 + k);
    inside[k] = px0 + k < W && py < H;
    Tr[k] = 1.f; C0[k] = C1[k] = C2[k] = 0.f;
    amin[k] = inside[k] ? ALPHA_MIN : 2.f;          // a finished pixel accepts no alpha (alpha <= 0.99)
    stop_pos[k] = inside[k] ? 0xffffffffu : 0u;
  }

  // ---------------- forward ----------------
  uint32_t wlast = 0;   // wave-uniform: last sorted position that contributed to any pixel of the tile
  int jlast = 0, blast = -1;
  bool wave_done = false;
  int staged = -1;
  unsigned long long staged_bal = 0ull;
  const int nbf = (int)((nv + U3D_WAVE - 1) / U3D_WAVE);
  for (int b = 0; b < nbf && !wave_done; ++b) {
    const unsigned long long bal = stage(b, nv);
    staged = b; staged_bal = bal;
    const int total = __popcll(bal);
    for (int j = 0; j < total; ++j) {
      const float4 A = sA[wave][j];
      const float4 B = sB[wave][j];
      const float4 Cc = sC[wave][j];
      const float dy = A.y - pyf;
      const float bdy = A.w * dy, cdy2 = (B.x * dy) * dy;
      lanemask_t contrib = 0ull, stopped = 0ull, m_stop[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // one basic block: the four pixels' dependency chains interleave
        const float dx = A.x - pxf[k];
        const float pw = fmaf(fmaf(A.z, dx, bdy), dx, cdy2);
        const float alpha = min_099(B.y * __builtin_amdgcn_exp2f(pw));
        const lanemask_t m_ok = __builtin_amdgcn_fcmpf(pw, 0.f, U3D_FCMP_OLE) & __builtin_amdgcn_fcmpf(alpha, amin[k], U3D_FCMP_OGE);
        const float w = alpha * Tr[k];
        const float test_T = Tr[k] - w;          // T (1 - alpha)
        const lanemask_t m_lt = __builtin_amdgcn_fcmpf(test_T, T_STOP, U3D_FCMP_OLT);
        const lanemask_t m_c = m_ok & ~m_lt;
        m_stop[k] = m_ok & m_lt;
        const float we = mask_sel0(m_c, w);      // blended weight, 0 for pixels that skip this Gaussian
        C0[k] = fmaf(Cc.x, we, C0[k]);
        C1[k] = fmaf(Cc.y, we, C1[k]);
        C2[k] = fmaf(Cc.z, we, C2[k]);
        Tr[k] -= we;
        contrib |= m_c;
        stopped |= m_stop[k];
      }
      if (stopped != 0ull) {   // rare, wave-uniform: pixels saturating at this Gaussian
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          amin[k] = mask_sel(m_stop[k], 2.f, amin[k]);
          stop_pos[k] = __float_as_uint(mask_sel(m_stop[k], B.w, __uint_as_float(stop_pos[k])));
        }
      }
      if (contrib != 0ull) { jlast = j; blast = b; }
      if (stopped != 0ull) {
        const bool all_done = amin[0] > 1.f && amin[1] > 1.f && amin[2] > 1.f && amin[3] > 1.f;
        if (__ballot(!all_done) == 0ull) { wave_done = true; break; }
      }
    }
    if (blast == b) wlast = __builtin_amdgcn_readfirstlane(__float_as_uint(sB[wave][jlast].w));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---------------- loss term, dL/dcolor seed ----------------
  float dp0[4], dp1[4], dp2[4], tfb[4];
  float e = 0.f;
  {
    const float sc = loss.inv_count;   // dL/dloss == 1
    const size_t pid0 = (size_t)py * W + px0;
    const float* gp = loss.gt + (size_t)view * 3 * npix + pid0;
    float* oc = out_color ? out_color + (size_t)view * 3 * npix + pid0 : nullptr;
    float o0[4], o1[4], o2[4], g0[4], g1[4], g2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o0[k] = fmaf(Tr[k], bg[0], C0[k]); o1[k] = fmaf(Tr[k], bg[1], C1[k]); o2[k] = fmaf(Tr[k], bg[2], C2[k]);
      g0[k] = g1[k] = g2[k] = 0.f;
    }
    if ((W & 3) == 0) {
      if (inside[0]) {   // the lane's four pixels are in or out together
        const float4 a = *reinterpret_cast<const float4*>(gp);
        const float4 b4 = *reinterpret_cast<const float4*>(gp + npix);
        const float4 c = *reinterpret_cast<const float4*>(gp + 2 * npix);
        g0[0] = a.x; g0[1] = a.y; g0[2] = a.z; g0[3] = a.w;
        g1[0] = b4.x; g1[1] = b4.y; g1[2] = b4.z; g1[3] = b4.w;
        g2[0] = c.x; g2[1] = c.y; g2[2] = c.z; g2[3] = c.w;
        if (oc) {
          *reinterpret_cast<float4*>(oc) = make_float4(o0[0], o0[1], o0[2], o0[3]);
          *reinterpret_cast<float4*>(oc + npix) = make_float4(o1[0], o1[1], o1[2], o1[3]);
          *reinterpret_cast<float4*>(oc + 2 * npix) = make_float4(o2[0], o2[1], o2[2], o2[3]);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (inside[k]) {
          g0[k] = gp[k]; g1[k] = gp[npix + k]; g2[k] = gp[2 * npix + k];
          if (oc) { oc[k] = o0[k]; oc[npix + k] = o1[k]; oc[2 * npix + k] = o2[k]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dp0[k] = dp1[k] = dp2[k] = 0.f;
      if (inside[k]) {
        const float d0 = o0[k] - g0[k], d1 = o1[k] - g1[k], d2 = o2[k] - g2[k];
        e += loss_pixel(loss, bg, g0[k], g1[k], g2[k], d0, d1, d2);
        if (loss.kind == 3) {
          dp0[k] = sc * (d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f));
          dp1[k] = sc * (d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f));
          dp2[k] = sc * (d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f));
        } else {
          const float w2 = 2.f * sc * focal_weight(loss, bg, g0[k], g1[k], g2[k]);
          dp0[k] = w2 * d0; dp1[k] = w2 * d1; dp2[k] = w2 * d2;
        }
      } else {
        Tr[k] = 0.f;
      }
      tfb[k] = Tr[k] * (bg[0] * dp0[k] + bg[1] * dp1[k] + bg[2] * dp2[k]);   // T_final * (bg . dL/dC)
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) e += __shfl_xor(e, o);
  if (lane == 0) loss.partial[lid] = e;

  // ---------------- backward ----------------
  // With  R_i = sum_{j behind i} w_j (c_j . dL/dC) + T_final (bg . dL/dC)  (everything behind Gaussian i, weighted by dL/dC)
  //   dL/dalpha_i = T_i (c_i . dL/dC) - R_i / (1 - alpha_i),      R_{i-1} = R_i + w_i (c_i . dL/dC),
  // the same quantity as the reference's normalised "accum_rec" form ((c_i - accum_rec) T_i - T_final/(1-alpha_i) bg.dL/dC)
  // in three instructions and one running value per pixel; tfb[] seeds R.
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  const uint32_t wmax = wlast;
  const bool row_lane = (lane & 3) == 0, first_lane = (lane & 15) == 0;
  const int bank = (lane >> 2) & 3;
  const int nb = (int)((wmax + U3D_WAVE - 1) / U3D_WAVE);
  for (int b = nb - 1; b >= 0; --b) {
    unsigned long long bal = staged_bal;
    if (b != staged) { bal = stage(b, wmax); staged = b; staged_bal = bal; }
    // entries of this batch with pos <= wmax (the compaction keeps positions ascending)
    const uint32_t lim = wmax - (uint32_t)b * U3D_WAVE;
    const int total = lim >= U3D_WAVE ? __popcll(bal) : __popcll(bal & ((1ull << lim) - 1ull));
    for (int j = total - 1; j >= 0; --j) {
      const float4 A = sA[wave][j];
      const float4 B = sB[wave][j];
      const float4 Cc = sC[wave][j];
      const uint32_t pos = __float_as_uint(B.w);
      const float dy = A.y - pyf;
      const float bdy = A.w * dy, cdy2 = (B.x * dy) * dy;
      float dx[4], ae[4];
      lanemask_t any = 0ull;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dx[k] = A.x - pxf[k];
        const float pw = fmaf(fmaf(A.z, dx[k], bdy), dx[k], cdy2);
        const float araw = B.y * __builtin_amdgcn_exp2f(pw);   // opacity * G (alpha before the 0.99 clamp)
        const lanemask_t m = __builtin_amdgcn_fcmpf(pw, 0.f, U3D_FCMP_OLE) & __builtin_amdgcn_fcmpf(araw, ALPHA_MIN, U3D_FCMP_OGE) &
                             __builtin_amdgcn_uicmp(pos, stop_pos[k], U3D_ICMP_ULT);
        any |= m;
        ae[k] = mask_sel0(m, araw);
      }
      if (any == 0ull) continue;
      float m0, mx, mxx, g_r, g_g, g_b;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float alpha = min_099(ae[k]);
        const float om = 1.f - alpha;
        const float rc = __builtin_amdgcn_rcpf(om);
        const float Tn = Tr[k] * rc;            // T in front of this Gaussian
        const float w = alpha * Tn;
        Tr[k] = Tn;
        const float cdp = fmaf(Cc.z, dp2[k], fmaf(Cc.y, dp1[k], Cc.x * dp0[k]));
        const float dL_dalpha = fmaf(Tn, cdp, -(tfb[k] * rc));
        tfb[k] = fmaf(w, cdp, tfb[k]);
        const float q = ae[k] * dL_dalpha;    // dL/dG * G
        const float qdx = q * dx[k];
        if (k == 0) {
          g_r = w * dp0[k]; g_g = w * dp1[k]; g_b = w * dp2[k];
          m0 = q; mx = qdx; mxx = qdx * dx[k];
        } else {
          g_r = fmaf(w, dp0[k], g_r); g_g = fmaf(w, dp1[k], g_g); g_b = fmaf(w, dp2[k], g_b);
          m0 += q; mx += qdx;
          mxx = fmaf(qdx, dx[k], mxx);
        }
      }
      // quad levels (one tile row of 16 pixels per quad); v_add_f32_dpp by hand: hipcc does not fuse update_dpp + fadd
      // (-0.0 rule).  Dependent DPP ops stay >= 2 instructions apart (VALU write -> DPP read hazard).
      asm volatile("s_nop 1\n\t"
                   "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %4, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %5, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                   "s_nop 1"
                   : "+v"(m0), "+v"(mx), "+v"(mxx), "+v"(g_r), "+v"(g_g), "+v"(g_b));
      float my = dy * m0, mxy = dy * mx;
      float myy = dy * my;
      // half-row level (lanes i <-> 7-i: banks 0<->1, 2<->3), two values per register: banks {0,2} keep the first
      // operand's sums, banks {1,3} receive the second's; then the row level (i <-> i+8: banks 0<->2, 1<->3) the same way:
      //   mx  <- {mx, my, mxx, mxy}    myy <- {myy, m0, g_r, g_g}    g_b <- g_b     (bank index = component)
      asm volatile("s_nop 1\n\t"
                   "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                   "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                   "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                   "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %0, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                   "v_add_f32_dpp %1, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                   "v_add_f32_dpp %2, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                   "v_add_f32_dpp %3, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                   "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                   "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                   "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                   "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                   "v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                   "s_nop 1"
                   : "+v"(mx), "+v"(mxx), "+v"(myy), "+v"(g_r), "+v"(g_b)
                   : "v"(my), "v"(mxy), "v"(m0), "v"(g_g));
      // the four 16-lane rows meet in LDS: batch 0 is indexed by sorted position (merged across tiles by
      // bwd_reduce_kernel), later batches by compaction slot
      float* sl = reinterpret_cast<float*>(&sAcc[wave][b == 0 ? (int)pos - 1 : j][0]);
      if (row_lane) {
        atomicAdd(sl + bank, mx);
        atomicAdd(sl + 4 + bank, myy);
      }
      if (first_lane) atomicAdd(sl + 8, g_b);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (b > 0) {
      if (lane < total) {
        const size_t g = vbase + __float_as_uint(sC[wave][lane].w);
        const float4 m0v = sAcc[wave][lane][0], m1v = sAcc[wave][lane][1], m2v = sAcc[wave][lane][2];
        const float4 co = conic_op[g];
        const float m[U3D_NACC] = {m0v.x, m0v.y, m0v.z, m0v.w, m1v.x, m1v.y, m1v.z, m1v.w, m2v.x, m2v.y};
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const float v = moment_to_acc<float>(k, m, co.x, co.y, co.z, co.w, ddelx_dx, ddely_dy);
          if (v != 0.f) unsafeAtomicAdd(&acc[(size_t)k * NG + g], (double)v);
        }
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) sAcc[wave][lane][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  float* pt = part + (size_t)lid * BWD_PART_STRIDE;
  const uint32_t cnt = min(wmax, (uint32_t)U3D_WAVE);
  if ((uint32_t)lane < cnt) {
#pragma unroll
    for (int k = 0; k < NK; ++k) pt[k * U3D_WAVE + lane] = reinterpret_cast<const float*>(&sAcc[wave][lane][0])[k];   // raw moments
  }
  if (lane == 0) reinterpret_cast<uint32_t*>(pt)[U3D_NACC * U3D_WAVE] = cnt;
}

// acc[k][view*P + sorted_id[sp]] += sum over a slice of the view's tiles (ascending) of part[tile][k][sp], in f64;
// the BWD_REDUCE_SPLIT slices of a view meet in an f64 atomic (order-insensitive at fp32 output precision).
__global__ __launch_bounds__(U3D_NACC * U3D_WAVE) void bwd_reduce_kernel(int P, int T, int NK, size_t NG, float half_w, float half_h,
                                                                        const uint32_t* __restrict__ sorted_id,
                                                                        const float4* __restrict__ conic_op,
                                                                        const float* __restrict__ part,
                                                                        double* __restrict__ acc, int n_loss,
                                                                        const float* __restrict__ loss_partial, float inv_count,
                                                                        float* __restrict__ loss_out) {
  __shared__ double s_sum[U3D_NACC][U3D_WAVE];
  if (blockIdx.y == BWD_REDUCE_SPLIT) {
    // extra row of the grid: fixed-order sum of the per-tile loss partials (replaces a separate launch)
    if (blockIdx.x != 0) return;
    float* sm = reinterpret_cast<float*>(&s_sum[0][0]);
    constexpr int NT = U3D_NACC * U3D_WAVE;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int i = threadIdx.x;
    for (; i + 3 * NT < n_loss; i += 4 * NT) {
      a0 += loss_partial[i]; a1 += loss_partial[i + NT]; a2 += loss_partial[i + 2 * NT]; a3 += loss_partial[i + 3 * NT];
    }
    for (; i < n_loss; i += NT) a0 += loss_partial[i];
    sm[threadIdx.x] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 64) {
      float v = 0.f;
      for (int j = threadIdx.x; j < NT; j += 64) v += sm[j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (threadIdx.x == 0) loss_out[0] = v * inv_count;
    }
    return;
  }
  const int view = blockIdx.x, k = threadIdx.x >> 6, sp = threadIdx.x & 63;
  double a = 0.0;
  if (k < NK) {
    const int per = (T + BWD_REDUCE_SPLIT - 1) / BWD_REDUCE_SPLIT;
    const int t0 = blockIdx.y * per, t1 = min(T, t0 + per);
    const float* base = part + (size_t)view * T * BWD_PART_STRIDE;
    // loads are unconditional (rows a tile did not write hold stale bytes, discarded by the select) so that a whole group
    // of tiles is in flight at once; accumulation order stays ascending in t within each of the two chains
    double a0 = 0.0, a1 = 0.0;
    int t = t0;
    for (; t + 7 < t1; t += 8) {
      float v[8];
      uint32_t c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float* pu = base + (size_t)(t + u) * BWD_PART_STRIDE;
        c[u] = reinterpret_cast<const uint32_t*>(pu)[U3D_NACC * U3D_WAVE];
        v[u] = pu[k * U3D_WAVE + sp];
      }
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        a0 += (uint32_t)sp < c[u] ? (double)v[u] : 0.0;
        a1 += (uint32_t)sp < c[u + 1] ? (double)v[u + 1] : 0.0;
      }
    }
    for (; t < t1; ++t) {
      const float* p0 = base + (size_t)t * BWD_PART_STRIDE;
      const uint32_t c0 = reinterpret_cast<const uint32_t*>(p0)[U3D_NACC * U3D_WAVE];
      const float v0 = p0[k * U3D_WAVE + sp];
      a0 += (uint32_t)sp < c0 ? (double)v0 : 0.0;
    }
    a = a0 + a1;
  }
  // raw moment sums of this slice -> accumulator values (linear, so slices can be converted independently)
  s_sum[k][sp] = a;
  __syncthreads();
  if (k >= NK) return;
  double m[U3D_NACC];
#pragma unroll
  for (int j = 0; j < U3D_NACC; ++j) m[j] = j < NK ? s_sum[j][sp] : 0.0;
  bool any = false;
#pragma unroll
  for (int j = 0; j < U3D_NACC; ++j) any = any || m[j] != 0.0;
  if (!any) return;
  const size_t g = (size_t)view * P + sorted_id[(size_t)view * P + sp];
  const float4 co = conic_op[g];
  const double outv = moment_to_acc<double>(k, m, co.x, co.y, co.z, co.w, half_w, half_h);
  if (outv != 0.0) unsafeAtomicAdd(&acc[(size_t)k * NG + g], outv);
}

// Fixed-order sum of the per-tile partials (deterministic): 1024 threads, 4 independent accumulators each.
__global__ __launch_bounds__(1024) void loss_reduce_kernel(int n, const float* __restrict__ partial, float inv_count,
                                                           float* __restrict__ loss_out) {
  __shared__ float sm[1024];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int i = threadIdx.x;
  for (; i + 3072 < n; i += 4096) {
    a0 += partial[i]; a1 += partial[i + 1024]; a2 += partial[i + 2048]; a3 += partial[i + 3072];
  }
  for (; i < n; i += 1024) a0 += partial[i];
  sm[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = sm[0] * inv_count;
}

}  // namespace

void u3d_launch_loss_reduce(int n, const float* partial, float inv_count, float* loss_out, hipStream_t s) {
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(1024), 0, s, n, partial, inv_count, loss_out);
}

void u3d_launch_render_fwd(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, float* out_color,
                           float* out_invdepth, const U3DLoss& loss, hipStream_t s) {
  const int tiles_x = (d.image_width + U3D_TILE - 1) / U3D_TILE, tiles_y = (d.image_height + U3D_TILE - 1) / U3D_TILE;
  const int T = tiles_x * tiles_y;
  const uint32_t nblocks = (uint32_t)(d.n_items * d.views_per_item * T);
  if (nblocks == 0) return;
  const uint32_t nwg = (nblocks + 3u) / 4u;
  hipLaunchKernelGGL(render_fwd_wave_kernel, dim3(nwg), dim3(U3D_BLOCK), 0, s, d.P, d.image_height, d.image_width, tiles_x, T,
                     nblocks, nwg, b.sorted_id, b.sorted_rect, b.n_vis, b.xy, b.conic_op, b.rgbd, bg, out_color, out_invdepth,
                     b.final_T, b.n_contrib, loss);

}

void u3d_launch_render_fb(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, float* out_color,
                          const U3DLoss& loss, double* acc, float* part, float* loss_out, hipStream_t s) {
  const int tiles_x = (d.image_width + U3D_TILE - 1) / U3D_TILE, tiles_y = (d.image_height + U3D_TILE - 1) / U3D_TILE;
  const int T = tiles_x * tiles_y;
  const uint32_t nblocks = (uint32_t)(d.n_items * d.views_per_item * T);
  const size_t NG = (size_t)d.n_items * d.views_per_item * d.P;
  if (nblocks == 0 || NG == 0) return;
  const uint32_t nwg = (nblocks + BWD_WAVES - 1u) / BWD_WAVES;
  hipLaunchKernelGGL(render_fb_wave_kernel, dim3(nwg), dim3(BWD_WAVES * U3D_WAVE), 0, s, d.P, d.image_height, d.image_width,
                     tiles_x, T, nblocks, nwg, NG, b.sorted_id, b.sorted_rect, b.n_vis, b.xy, b.conic_op, b.rgbd, bg, out_color,
                     acc, part, loss);
  hipLaunchKernelGGL(bwd_reduce_kernel, dim3(d.n_items * d.views_per_item, BWD_REDUCE_SPLIT + 1), dim3(U3D_NACC * U3D_WAVE), 0, s,
                     d.P, T, U3D_NACC - 1, NG, 0.5f * (float)d.image_width, 0.5f * (float)d.image_height, b.sorted_id, b.conic_op, part,
                     acc, (int)nblocks, loss.partial, loss.inv_count, loss_out);
}

void u3d_launch_render_bwd(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, const float* dL_dcolor,
                           const float* dL_dinvdepth, const float* out_color, const U3DLoss& loss, double* acc,
                           float* part, hipStream_t s) {
  const int tiles_x = (d.image_width + U3D_TILE - 1) / U3D_TILE, tiles_y = (d.image_height + U3D_TILE - 1) / U3D_TILE;
  const int T = tiles_x * tiles_y;
  const uint32_t nblocks = (uint32_t)(d.n_items * d.views_per_item * T);
  const size_t NG = (size_t)d.n_items * d.views_per_item * d.P;
  if (nblocks == 0 || NG == 0) return;
  const uint32_t nwg = (nblocks + BWD_WAVES - 1u) / BWD_WAVES;
  const bool invd = dL_dinvdepth && loss.kind == 0;
  if (invd)
    hipLaunchKernelGGL(render_bwd_wave_kernel<true>, dim3(nwg), dim3(BWD_WAVES * U3D_WAVE), 0, s, d.P, d.image_height, d.image_width,
                       tiles_x, T, nblocks, nwg, NG, b.sorted_id, b.sorted_rect, b.xy, b.conic_op, b.rgbd, bg, dL_dcolor,
                       dL_dinvdepth, b.final_T, b.n_contrib, acc, part, out_color, loss);
  else
    hipLaunchKernelGGL(render_bwd_wave_kernel<false>, dim3(nwg), dim3(BWD_WAVES * U3D_WAVE), 0, s, d.P, d.image_height, d.image_width,
                       tiles_x, T, nblocks, nwg, NG, b.sorted_id, b.sorted_rect, b.xy, b.conic_op, b.rgbd, bg, dL_dcolor,
                       dL_dinvdepth, b.final_T, b.n_contrib, acc, part, out_color, loss);
  hipLaunchKernelGGL(bwd_reduce_kernel, dim3(d.n_items * d.views_per_item, BWD_REDUCE_SPLIT), dim3(U3D_NACC * U3D_WAVE), 0, s,
                     d.P, T, invd ? U3D_NACC : U3D_NACC - 1, NG, 0.5f * (float)d.image_width, 0.5f * (float)d.image_height, b.sorted_id,
                     b.conic_op, part, acc, 0, nullptr, 0.f, nullptr);

}
