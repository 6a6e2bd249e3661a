// Tile compositing kernels for gfx950 (wave64, LDS-staged, no instance lists in HBM).
//
// One 256-thread workgroup (4 waves) renders one 16x16 tile of one view; wave w owns the 16x4 pixel
// strip of rows 4w..4w+3.  The workgroup walks the VIEW's depth-sorted Gaussian list (u3d_sort.hip) in
// batches of 256: every thread tests one sorted entry's tile rectangle against this tile, hits are
// compacted in order into LDS with a wave ballot + popcount prefix (the "duplicate / sort / range"
// stages of the original operator collapse into this filter), and the waves then blend the compacted
// batch front to back.  A wave leaves the batch as soon as all of its 64
// pixels are saturated, the workgroup stops fetching when all four are: with the reference's
// large, fairly opaque splats only the first few dozen sorted entries are ever touched.
//
// Backward (R5/R6) re-stages the same batches back to front, recovers T by division, and reduces each
// Gaussian's 10 screen-space gradient components across the wave with DPP row operations before ONE
// LDS add per wave and ONE global atomic per tile (the original: one atomic per pixel per component).
#include "u3d_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;

__device__ __forceinline__ bool rect_hits(const uint2 r, int tx, int ty) {
  const int x0 = r.x & 0xffffu, y0 = r.x >> 16, x1 = r.y & 0xffffu, y1 = r.y >> 16;
  return tx >= x0 && tx < x1 && ty >= y0 && ty < y1;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(U3D_BLOCK) void render_fwd_kernel(
    int P, int H, int W, int tiles_x, int T, uint32_t nblocks, const uint32_t* __restrict__ sorted_id,
    const uint2* __restrict__ sorted_rect, const uint32_t* __restrict__ n_vis, const float2* __restrict__ xy,
    const float4* __restrict__ conic_op, const float4* __restrict__ rgbd, const float* __restrict__ bg,
    float* __restrict__ out_color, float* __restrict__ out_invdepth, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib) {
  __shared__ float4 sA[U3D_BLOCK];    // x, y, -0.5*log2e*a, -log2e*b
  __shared__ float2 sB[U3D_BLOCK];    // -0.5*log2e*c, opacity
  __shared__ float4 sC[U3D_BLOCK];    // r, g, b, 1/depth
  __shared__ uint32_t sPos[U3D_BLOCK];  // sorted position + 1
  __shared__ uint32_t sCnt[4];

  const uint32_t lid = u3d_xcd_remap(blockIdx.x, nblocks);
  const int view = lid / T, tile = lid - view * T;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int px = tx * U3D_TILE + (lane & 15), py = ty * U3D_TILE + wave * 4 + (lane >> 4);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const size_t vbase = (size_t)view * P;
  const uint32_t nv = n_vis[view];

  float Tr = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dv = 0.f;
  uint32_t last = 0;
  bool done = !inside;

  for (uint32_t base = 0; base < nv; base += U3D_BLOCK) {
    const uint32_t s = base + tid;
    bool hit = false;
    if (s < nv) hit = rect_hits(sorted_rect[vbase + s], tx, ty);
    const unsigned long long bal = __ballot(hit);
    const bool wave_done = __ballot(!done) == 0ull;
    if (lane == 0) sCnt[wave] = (uint32_t)__popcll(bal) | (wave_done ? 0x80000000u : 0u);
    __syncthreads();
    const uint32_t c0 = sCnt[0], c1 = sCnt[1], c2 = sCnt[2], c3 = sCnt[3];
    if ((c0 & c1 & c2 & c3) & 0x80000000u) break;  // every wave saturated: stop fetching
    const uint32_t n0 = c0 & 0x7fffffffu, n1 = c1 & 0x7fffffffu, n2 = c2 & 0x7fffffffu, n3 = c3 & 0x7fffffffu;
    const uint32_t total = n0 + n1 + n2 + n3;
    if (hit) {
      const uint32_t woff = wave == 0 ? 0u : (wave == 1 ? n0 : (wave == 2 ? n0 + n1 : n0 + n1 + n2));
      const uint32_t o = woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      const size_t g = vbase + sorted_id[vbase + s];
      const float2 m = xy[g];
      const float4 co = conic_op[g];
      const float4 cd = rgbd[g];
      sA[o] = make_float4(m.x, m.y, -0.5f * LOG2E * co.x, -LOG2E * co.y);
      sB[o] = make_float2(-0.5f * LOG2E * co.z, co.w);
      sC[o] = make_float4(cd.x, cd.y, cd.z, 1.0f / cd.w);
      sPos[o] = s + 1u;
    }
    __syncthreads();
    if (!wave_done) {
      for (uint32_t j = 0; j < total; ++j) {
        const float4 A = sA[j];
        const float2 B = sB[j];
        const float dx = A.x - pxf, dy = A.y - pyf;
        // log2-domain exponent: -0.5*(a dx^2 + c dy^2) - b dx dy, pre-scaled by log2(e)
        const float pw = fmaf(A.z * dx, dx, fmaf(B.x * dy, dy, A.w * dx * dy));
        const float alpha = fminf(0.99f, B.y * __builtin_amdgcn_exp2f(pw));
        const bool ok = !done && pw <= 0.f && alpha >= ALPHA_MIN;
        const float test_T = Tr * (1.f - alpha);
        const bool stop = ok && test_T < T_STOP;
        done = done || stop;
        if (ok && !stop) {
          const float4 Cc = sC[j];
          const float w = alpha * Tr;
          C0 = fmaf(Cc.x, w, C0);
          C1 = fmaf(Cc.y, w, C1);
          C2 = fmaf(Cc.z, w, C2);
          Dv = fmaf(Cc.w, w, Dv);
          Tr = test_T;
          last = sPos[j];
        }
        if (__ballot(!done) == 0ull) break;  // wave saturated
      }
    }
  }
  if (inside) {
    const size_t npix = (size_t)H * W;
    const size_t pid = (size_t)py * W + px;
    final_T[(size_t)view * npix + pid] = Tr;
    n_contrib[(size_t)view * npix + pid] = last;
    float* oc = out_color + (size_t)view * 3 * npix + pid;
    oc[0] = fmaf(Tr, bg[0], C0);
    oc[npix] = fmaf(Tr, bg[1], C1);
    oc[2 * npix] = fmaf(Tr, bg[2], C2);
    if (out_invdepth) out_invdepth[(size_t)view * npix + pid] = Dv;
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

__global__ __launch_bounds__(U3D_BLOCK) void render_bwd_kernel(
    int P, int H, int W, int tiles_x, int T, uint32_t nblocks, size_t NG, const uint32_t* __restrict__ sorted_id,
    const uint2* __restrict__ sorted_rect, const float2* __restrict__ xy, const float4* __restrict__ conic_op,
    const float4* __restrict__ rgbd, const float* __restrict__ bg, const float* __restrict__ dL_dcolor,
    const float* __restrict__ dL_dinvdepth, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    float* __restrict__ acc) {
  __shared__ float4 sA[U3D_BLOCK];   // x, y, a, b
  __shared__ float4 sB[U3D_BLOCK];   // c, opacity, 1/depth, -
  __shared__ float4 sC[U3D_BLOCK];   // r, g, b, -
  __shared__ uint32_t sPos[U3D_BLOCK];
  __shared__ uint32_t sId[U3D_BLOCK];
  __shared__ float sAcc[U3D_NACC][U3D_BLOCK];
  __shared__ uint32_t sCnt[4];
  __shared__ uint32_t sMax[4];

  const uint32_t lid = u3d_xcd_remap(blockIdx.x, nblocks);
  const int view = lid / T, tile = lid - view * T;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int px = tx * U3D_TILE + (lane & 15), py = ty * U3D_TILE + wave * 4 + (lane >> 4);
  const bool inside = px < W && py < H;
  const float pxf = (float)px, pyf = (float)py;
  const size_t vbase = (size_t)view * P;
  const size_t npix = (size_t)H * W;
  const size_t pid = (size_t)py * W + px;

  const float T_final = inside ? final_T[(size_t)view * npix + pid] : 0.f;
  const uint32_t last = inside ? n_contrib[(size_t)view * npix + pid] : 0u;
  float dp0 = 0.f, dp1 = 0.f, dp2 = 0.f, dinv = 0.f;
  if (inside) {
    const float* dc = dL_dcolor + (size_t)view * 3 * npix + pid;
    dp0 = dc[0]; dp1 = dc[npix]; dp2 = dc[2 * npix];
    if (dL_dinvdepth) dinv = dL_dinvdepth[(size_t)view * npix + pid];
  }
  const float bg_dot = bg[0] * dp0 + bg[1] * dp1 + bg[2] * dp2;
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

  uint32_t wmax = last;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, o));
  if (lane == 0) sMax[wave] = wmax;
  __syncthreads();
  const uint32_t maxpos = max(max(sMax[0], sMax[1]), max(sMax[2], sMax[3]));

  float Tr = T_final, ar0 = 0.f, ar1 = 0.f, ar2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;
  float ainv = 0.f, linv = 0.f;

  const int nb = (int)((maxpos + U3D_BLOCK - 1) / U3D_BLOCK);
  for (int b = nb - 1; b >= 0; --b) {
    const uint32_t base = (uint32_t)b * U3D_BLOCK;
    const uint32_t s = base + tid;
    bool hit = false;
    if (s < maxpos) hit = rect_hits(sorted_rect[vbase + s], tx, ty);
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) sCnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();  // also orders the previous batch's flush before the restage below
    const uint32_t n0 = sCnt[0], n1 = sCnt[1], n2 = sCnt[2], n3 = sCnt[3];
    const uint32_t total = n0 + n1 + n2 + n3;
    if (hit) {
      const uint32_t woff = wave == 0 ? 0u : (wave == 1 ? n0 : (wave == 2 ? n0 + n1 : n0 + n1 + n2));
      const uint32_t o = woff + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      const uint32_t id = sorted_id[vbase + s];
      const size_t g = vbase + id;
      const float2 m = xy[g];
      const float4 co = conic_op[g];
      const float4 cd = rgbd[g];
      sA[o] = make_float4(m.x, m.y, co.x, co.y);
      sB[o] = make_float4(co.z, co.w, 1.0f / cd.w, 0.f);
      sC[o] = make_float4(cd.x, cd.y, cd.z, 0.f);
      sPos[o] = s + 1u;
      sId[o] = id;
    }
    if ((uint32_t)tid < total) {
#pragma unroll
      for (int k = 0; k < U3D_NACC; ++k) sAcc[k][tid] = 0.f;
    }
    __syncthreads();
    if (wmax > base) {
      for (int j = (int)total - 1; j >= 0; --j) {
        const uint32_t pos = sPos[j];
        if (pos > wmax) continue;  // wave-uniform: behind every pixel's last contributor
        const float4 A = sA[j];
        const float4 B = sB[j];
        const float dx = A.x - pxf, dy = A.y - pyf;
        const float pw = fmaf(-0.5f * LOG2E * A.z * dx, dx, fmaf(-0.5f * LOG2E * B.x * dy, dy, -LOG2E * A.w * dx * dy));
        const float G = __builtin_amdgcn_exp2f(pw);
        const float alpha = fminf(0.99f, B.y * G);
        const bool ok = pos <= last && pw <= 0.f && alpha >= ALPHA_MIN;
        if (__ballot(ok) == 0ull) continue;
        float g_mx = 0.f, g_my = 0.f, g_ca = 0.f, g_cb = 0.f, g_cc = 0.f, g_op = 0.f, g_r = 0.f, g_g = 0.f, g_b = 0.f,
              g_d = 0.f;
        if (ok) {
          const float4 Cc = sC[j];
          Tr = Tr * __builtin_amdgcn_rcpf(1.f - alpha);
          const float w = alpha * Tr;
          ar0 = last_alpha * lc0 + (1.f - last_alpha) * ar0; lc0 = Cc.x;
          ar1 = last_alpha * lc1 + (1.f - last_alpha) * ar1; lc1 = Cc.y;
          ar2 = last_alpha * lc2 + (1.f - last_alpha) * ar2; lc2 = Cc.z;
          float dL_dalpha = (Cc.x - ar0) * dp0 + (Cc.y - ar1) * dp1 + (Cc.z - ar2) * dp2;
          g_r = w * dp0; g_g = w * dp1; g_b = w * dp2;
          ainv = last_alpha * linv + (1.f - last_alpha) * ainv; linv = B.z;
          dL_dalpha += (B.z - ainv) * dinv;
          g_d = w * dinv;
          dL_dalpha *= Tr;
          last_alpha = alpha;
          dL_dalpha += (-T_final * __builtin_amdgcn_rcpf(1.f - alpha)) * bg_dot;
          const float dL_dG = B.y * dL_dalpha;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * A.z - gdy * A.w;
          const float dG_ddely = -gdy * B.x - gdx * A.w;
          g_mx = dL_dG * dG_ddelx * ddelx_dx;
          g_my = dL_dG * dG_ddely * ddely_dy;
          g_ca = -0.5f * gdx * dx * dL_dG;
          g_cb = -0.5f * gdx * dy * dL_dG;
          g_cc = -0.5f * gdy * dy * dL_dG;
          g_op = G * dL_dalpha;
        }
        g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my);
        g_ca = wave_sum_to_lane63(g_ca); g_cb = wave_sum_to_lane63(g_cb); g_cc = wave_sum_to_lane63(g_cc);
        g_op = wave_sum_to_lane63(g_op);
        g_r = wave_sum_to_lane63(g_r); g_g = wave_sum_to_lane63(g_g); g_b = wave_sum_to_lane63(g_b);
        g_d = wave_sum_to_lane63(g_d);
        if (lane == 63) {
          atomicAdd(&sAcc[0][j], g_mx); atomicAdd(&sAcc[1][j], g_my);
          atomicAdd(&sAcc[2][j], g_ca); atomicAdd(&sAcc[3][j], g_cb); atomicAdd(&sAcc[4][j], g_cc);
          atomicAdd(&sAcc[5][j], g_op);
          atomicAdd(&sAcc[6][j], g_r); atomicAdd(&sAcc[7][j], g_g); atomicAdd(&sAcc[8][j], g_b);
          atomicAdd(&sAcc[9][j], g_d);
        }
      }
    }
    __syncthreads();
    if ((uint32_t)tid < total) {
      const size_t g = vbase + sId[tid];
#pragma unroll
      for (int k = 0; k < U3D_NACC; ++k) {
        const float v = sAcc[k][tid];
        if (v != 0.f) unsafeAtomicAdd(&acc[(size_t)k * NG + g], v);
      }
    }
  }
}

}  // namespace

void u3d_launch_render_fwd(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, float* out_color,
                           float* out_invdepth, hipStream_t s) {
  const int tiles_x = (d.image_width + U3D_TILE - 1) / U3D_TILE, tiles_y = (d.image_height + U3D_TILE - 1) / U3D_TILE;
  const int T = tiles_x * tiles_y;
  const uint32_t nblocks = (uint32_t)(d.n_items * d.views_per_item * T);
  if (nblocks == 0) return;
  hipLaunchKernelGGL(render_fwd_kernel, dim3(nblocks), dim3(U3D_BLOCK), 0, s, d.P, d.image_height, d.image_width,
                     tiles_x, T, nblocks, b.sorted_id, b.sorted_rect, b.n_vis, b.xy, b.conic_op, b.rgbd, bg, out_color,
                     out_invdepth, b.final_T, b.n_contrib);
}

void u3d_launch_render_bwd(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, const float* dL_dcolor,
                           const float* dL_dinvdepth, float* acc, hipStream_t s) {
  const int tiles_x = (d.image_width + U3D_TILE - 1) / U3D_TILE, tiles_y = (d.image_height + U3D_TILE - 1) / U3D_TILE;
  const int T = tiles_x * tiles_y;
  const uint32_t nblocks = (uint32_t)(d.n_items * d.views_per_item * T);
  const size_t NG = (size_t)d.n_items * d.views_per_item * d.P;
  if (nblocks == 0 || NG == 0) return;
  hipLaunchKernelGGL(render_bwd_kernel, dim3(nblocks), dim3(U3D_BLOCK), 0, s, d.P, d.image_height, d.image_width,
                     tiles_x, T, nblocks, NG, b.sorted_id, b.sorted_rect, b.xy, b.conic_op, b.rgbd, bg, dL_dcolor,
                     dL_dinvdepth, b.final_T, b.n_contrib, acc);
}
