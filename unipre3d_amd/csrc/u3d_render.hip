// Tile compositing kernels for gfx950 (wave64, LDS-staged, no instance lists in HBM).
//
// ONE WAVE (one workgroup) renders one 16x16 tile, 4 pixels per lane (row lane>>2, columns 4*(lane&3)+k).  The wave walks the
// VIEW's depth-sorted Gaussian list (u3d_sort.hip) in batches of 64: every lane tests one sorted entry's tile rectangle
// against this tile, hits are compacted in order into LDS with a ballot + popcount prefix (the "duplicate / sort / range"
// stages of the original operator collapse into this filter), and the batch is blended front to back from LDS broadcast
// reads.  The wave leaves as soon as all of its 256 pixels are saturated.
//
// Backward (R5/R6) re-stages the batches back to front, recovers T by division, accumulates each Gaussian's gradients as
// moments over the lane's 4 pixels, reduces them across the wave (DPP, bank-masked pairs, LDS row merge), and hands the
// tile's rows to a fixed-order f64 reduction (the original: one fp32 atomic per pixel per component, non-deterministic).
// render_fb_wave_kernel does forward and backward of the fused render-loss training step in a single pass.
#include "u3d_common.h"
#if defined(U3D_LPT_EXPERIMENT) || defined(U3D_TIMELINE)
#include <algorithm>
#include <cstdio>
#include <vector>
#endif

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_STOP = 0.0001f;
#ifndef U3D_FWD_HOIST
#define U3D_FWD_HOIST 1      // forward: one wave-uniform saturation test per entry instead of four per-pixel mask chains (round 6)
#endif

__device__ __forceinline__ bool rect_hits(const uint2 r, int tx, int ty) {
  const int x0 = r.x & 0xffffu, y0 = r.x >> 16, x1 = r.y & 0xffffu, y1 = r.y >> 16;
  return tx >= x0 && tx < x1 && ty >= y0 && ty < y1;
}

// Render loss of one pixel (utils/loss_utils.py:17-45), branch-free in the lane: a pixel outside the image gets weight 0.
// torch.isclose(gt, bg, atol=1e-6) uses rtol=1e-5; the per-channel bounds and the background are wave-uniform.
struct LossCtx {
  int kind;
  float w_bg, w_non, b0, b1, b2, t0, t1, t2;
};
__device__ __forceinline__ LossCtx loss_ctx(const U3DLoss& L, const float* __restrict__ bg) {
  const float b0 = bg[0], b1 = bg[1], b2 = bg[2];
  return LossCtx{L.kind, L.w_bg, L.w_non, b0, b1, b2, 1e-6f + 1e-5f * fabsf(b0), 1e-6f + 1e-5f * fabsf(b1), 1e-6f + 1e-5f * fabsf(b2)};
}
__device__ __forceinline__ float loss_weight(const LossCtx& c, bool inside, float g0, float g1, float g2) {
  float w = 1.f;
  if (c.kind == 2) {   // focal: background pixels of the target weigh w_bg, the others w_non
    const bool is_bg = fabsf(g0 - c.b0) <= c.t0 && fabsf(g1 - c.b1) <= c.t1 && fabsf(g2 - c.b2) <= c.t2;
    w = is_bg ? c.w_bg : c.w_non;
  }
  return inside ? w : 0.f;
}
__device__ __forceinline__ float loss_term(const LossCtx& c, float w, float d0, float d1, float d2) {
  if (c.kind == 3) return w * (fabsf(d0) + fabsf(d1) + fabsf(d2));
  return w * (d0 * d0 + d1 * d1 + d2 * d2);
}
// dL/dC of the pixel, sc = dL/dloss / count
__device__ __forceinline__ void loss_seed(const LossCtx& c, float w, float sc, float d0, float d1, float d2, float& dp0, float& dp1, float& dp2) {
  if (c.kind == 3) {
    const float s = sc * w;
    dp0 = s * (d0 > 0.f ? 1.f : (d0 < 0.f ? -1.f : 0.f));
    dp1 = s * (d1 > 0.f ? 1.f : (d1 < 0.f ? -1.f : 0.f));
    dp2 = s * (d2 > 0.f ? 1.f : (d2 < 0.f ? -1.f : 0.f));
  } else {
    const float w2 = 2.f * sc * w;
    dp0 = w2 * d0; dp1 = w2 * d1; dp2 = w2 * d2;
  }
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
// m = a & b (NOT_B: a & ~b);  acc |= m;  returns lane in m ? v : 0.
// (Round 4 tried forming the combined mask IN vcc and selecting with the VOP2 v_cndmask that reads vcc -- 2-cycle class in isolation
// against 4 for the VOP3 form on an SGPR pair: the scalar write -> vector read of vcc serialises each pixel's chain, render_fb at C2
// 191.5 -> 198.5 us, compact 641 -> 682 us; EXPERIMENTS.md.)
template <bool NOT_B>
__device__ __forceinline__ float mask_combine_sel0(lanemask_t a, lanemask_t b, lanemask_t& acc, float v) {
  const lanemask_t m = NOT_B ? (a & ~b) : (a & b);
  acc |= m;
  return mask_sel0(m, v);
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

// ---- tile machinery shared by the three compositing kernels -----------------------------------------
// ONE WAVE owns one 16x16 tile.  Lane layout: lane>>2 is the tile ROW, the lane's 4 pixels are the consecutive COLUMNS
// 4*(lane&3)+k, so image rows move as one 16-byte access per lane and channel, and a DPP quad (4 lanes) is one 16-pixel
// tile row: everything a quad sums shares dy.  LDS regions are wave-private, no kernel here has a workgroup barrier.
// The kernels are VALU-issue bound (no HBM or LDS limit in sight), so the arithmetic is arranged for instruction count
// and for the 2-cycle FMA/MUL/ADD class over the 4-cycle compare/min/select/DPP and 8-cycle exp/rcp classes:
//   * exponent  pw = (a' dx + b' dy) dx + c' dy^2  with the -log2e/2 factors folded in at staging (3 ops/pixel);
//   * compare results live as lane masks on the scalar unit; a pixel that skips a Gaussian takes the same straight-line
//     code with its weight selected to zero (no per-pixel branches, four dependency chains per basic block);
//   * backward: with  R_i = sum_{j behind i} w_j (c_j . dL/dC) + T_final (bg . dL/dC)
//         dL/dalpha_i = T_i (c_i . dL/dC) - R_i / (1 - alpha_i),      R_{i-1} = R_i + w_i (c_i . dL/dC),
//     the same quantity as the reference's normalised "accum_rec" form in three instructions and one running value;
//   * only m0, mx, mxx and the colour (depth) gradients are accumulated per pixel; after the two quad levels of the
//     reduction  my = dy m0, mxy = dy mx, myy = dy my;
//   * the half-row and row levels use DPP bank masks to deposit two values into one register per instruction pair
//     (9-10 values -> 5 -> 3 registers), and the four rows of the wave meet through the LDS crossbar (swizzle + bpermute).
constexpr int BWD_PART_STRIDE = U3D_PART_STRIDE;   // floats per tile: [U3D_PART_BLOCKS][64 positions][10], the LDS rows as they are
// bwd_reduce_kernel: workgroups per view.  The slices of a view meet in f64 atomics (cost ~ slices), the tile chain of a
// slice is latency-bound (cost ~ tiles per slice): ~128 tiles per slice measured best (C2: 10.4 us with 2 slices, 17 with 8).
// With few views (scene level: 8-16) that alone leaves most CUs idle, so the slice count also grows until ~256 workgroups exist.
static inline int bwd_reduce_split(int T, int NV) {
#ifdef U3D_REDUCE_SPLIT_ENV   /* experiment builds: slices per view from the environment */
  if (const char* e = getenv("U3D_REDUCE_SPLIT")) { const int v = atoi(e); if (v > 0 && v <= 32 && v <= T) return v; }
#endif
  int s = (T + 64) / 128;
  const int fill = NV > 0 ? 256 / NV : 1;
  if (s < fill) s = fill;
  if (s > T / 8) s = T / 8;
  return s < 1 ? 1 : (s > 32 ? 32 : s);
}
constexpr int TILE_WAVES = 1;   // tiles per workgroup: one (finer-grained dispatch measured 8 % faster than four)

struct TileLds {   // wave-private; 2560 B of staged entries + 2560 B of gradient rows = 5 KB per wave -> 32 waves per CU
  float4* P0;          // [64] x, y, a' = -log2e/2 a, b' = -log2e b
  float4* P1;          // [64] c' = -log2e/2 c, opacity, r, g
  float2* P2;          // [64] b, pos (bits)
  float* D;            // [64] 1/depth (only the kernels that carry inverse depth)
  float (*acc)[10];    // [64] per slot: mx, my, mxx, mxy, myy, m0, r, g, b, d   (backward only)
  float* S;            // [64], stride sstride: exp2(2 a'), the step ratio of the forward-differenced exponent (alpha_run).  The kernels
  int sstride;         // without depth gradients keep it in the unused tenth float of the gradient rows: the wave stays at 5 KB of LDS
};
struct TileGeom {
  const uint32_t* sorted_id; const uint2* sorted_rect; const float2* xy; const float4* conic_op; const float4* rgbd;
  size_t vbase; int tx, ty;
  int rect_indirect;   // scene level: `sorted_rect` is the per-pair rectangle array, read through sorted_id (the depth sort of 10^5
                       // pairs per view no longer gathers and rewrites 12 bytes per pair for the few dozen positions a tile reads)
  uint32_t* touched;   // the `clamped` words (U3D_TOUCHED_BIT), backward only
  uint32_t* tw;        // per-Gaussian touched bitmap (scene level only, else null) and the first Gaussian of the view's set
  size_t gbase;
  uint2* tlist;        // touched list + count (U3D_FLAG_SPARSE_BWD, else null) and the view's set
  uint32_t* tcount;
  uint32_t item;
};

// stage sorted entries [b*64, b*64+64) limited to `limit`; returns the hit ballot (compaction keeps the order).
// `plain` (wave-uniform) is set when, for EVERY staged entry, two of the per-pixel tests of the blend are provably no-ops, so the
// batch may run the loop variants without them (8 fewer 4-cycle instructions per Gaussian and pass):
//   * opacity <= 0.98: alpha = opacity * exp2(pw) <= 0.99 whenever pw <= 0, i.e. min(0.99, .) changes nothing;
//   * the staged quadratic form is negative semi-definite with margin, a', c' <= 0 and b'^2 <= 4 a' c' (1 - 1e-5): the computed
//     pw = fma(fma(a', dx, b' dy), dx, (c' dy) dy) differs from the exact form by at most 4 * 2^-24 (|a'| dx^2 + |c'| dy^2),
//     while the exact form is <= -(1e-5 / 2) (|a'| dx^2 + |c'| dy^2): the `pw > 0 -> skip` test can never fire (the pixel
//     centre is finite; NaN alphas still fail the alpha >= 1/255 test);
//   * a' >= -2.5 (conic xx <= 3.47: every covariance that went through the +0.3 low-pass has xx <= 3.34), which bounds how fast the
//     exponent can fall along a lane's 4-pixel run -- what alpha_run needs.
// Any entry outside these bounds (opacity above 0.98, a nearly singular or non-finite conic) sends the whole TILE through the loops that
// carry both tests.  Since round 5 the two variants are NOT bit-identical: the PLAIN loops take alpha from the alpha_run recurrence (2 exp2 +
// multiplies), the general ones from exp2(pw) per pixel, which agree to ~1e-6 relative.  The invariant that matters is that FORWARD AND
// BACKWARD OF A TILE RUN THE SAME VARIANT (a 1/255 threshold decided one way by the forward and the other way by the backward would
// desynchronise stop_pos from the recomputed alphas): the single-pass kernel holds it by construction, the two-pass kernels through
// U3D_TILE_PLAIN_BIT in tile_last.  `U3D_FORCE_GENERAL` (experiment / test builds) sends every tile through the general variant, which
// tests/test_gpu_more_parity.py compares against the product build.
template <bool DEPTH>
__device__ __forceinline__ lanemask_t tile_stage(const TileLds& L, const TileGeom& G, int lane, int b, uint32_t limit, bool& plain) {
  const uint32_t s = (uint32_t)b * U3D_WAVE + (uint32_t)lane;
  bool hit = false, ok = true;
  if (s < limit) hit = rect_hits(G.sorted_rect[G.vbase + (G.rect_indirect ? G.sorted_id[G.vbase + s] : s)], G.tx, G.ty);
  const lanemask_t bal = __ballot(hit);
  if (hit) {
    const uint32_t o = (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
    const size_t g = G.vbase + G.sorted_id[G.vbase + s];
    const float2 m = G.xy[g];
    const float4 co = G.conic_op[g];
    const float4 cd = G.rgbd[g];
    const float a1 = (-0.5f * LOG2E) * co.x, b1 = -LOG2E * co.y, c1 = (-0.5f * LOG2E) * co.z;
    L.P0[o] = make_float4(m.x, m.y, a1, b1);
    L.P1[o] = make_float4(c1, co.w, cd.x, cd.y);
    L.P2[o] = make_float2(cd.z, __uint_as_float(s + 1u));
    if (DEPTH) L.D[o] = 1.0f / cd.w;
    L.S[o * L.sstride] = __builtin_amdgcn_exp2f(a1 + a1);
    ok = a1 <= 0.f && a1 >= -2.5f && c1 <= 0.f && b1 * b1 <= (4.f * (1.f - 1e-5f)) * (a1 * c1) && co.w <= 0.98f && fabsf(m.x) < 1e30f && fabsf(m.y) < 1e30f;
  }
#ifdef U3D_FORCE_GENERAL
  ok = false;
#endif
  plain = __ballot(!ok) == 0ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return bal;
}

// PLAIN batches: opacity * exp2(pw) of the lane's four pixels with TWO transcendentals instead of four (round 5).  The lane's pixels are
// dx0, dx0 - 1, dx0 - 2, dx0 - 3 and pw is quadratic in dx, so  pw(dx - 1) - pw(dx) = a'(1 - 2 dx) - b' dy  is linear and its step is the
// constant 2 a':   alpha_0 = o exp2(pw_0),  r_0 = exp2(a'(1 - 2 dx0) - b' dy),  alpha_{k+1} = alpha_k r_k,  r_{k+1} = r_k s,  s = exp2(2 a')
// staged once per entry (tile_stage).  11 multiply-add class instructions + 2 v_exp_f32 per lane and entry against 16 + 4.
// Range: with a' >= -2.5 and the form negative semi-definite, sqrt(-pw) is a seminorm and the run spans sqrt(-pw(3, 0)) <= 4.75, so a
// pixel that matters (o exp2(pw) >= 1/255, -pw <= 8) keeps every pw of its run above -57.3: nothing underflows before it; runs far from
// the splat give 0 * inf = NaN at worst, which fails the alpha >= 1/255 test like the zero it stands for.
__device__ __forceinline__ void alpha_run(float ax, float a1, float bdy, float cdy2, float op, float s2, float px0, float (&araw)[4]) {
  const float dx0 = ax - px0;
  const float pw0 = fmaf(fmaf(a1, dx0, bdy), dx0, cdy2);
  const float dl = fmaf(a1, fmaf(-2.f, dx0, 1.f), -bdy);
  float r = __builtin_amdgcn_exp2f(dl);
  araw[0] = op * __builtin_amdgcn_exp2f(pw0);
  araw[1] = araw[0] * r;
  r *= s2;
  araw[2] = araw[1] * r;
  r *= s2;
  araw[3] = araw[2] * r;
}

struct TileFwd {
  float Tr[4], C0[4], C1[4], C2[4], Dv[4];
  uint32_t stop_pos[4];   // sorted position at which the pixel saturated: entries from there on are not blended
  uint32_t wlast;         // wave-uniform: last sorted position that contributed to any pixel of the tile
  int staged;             // batch left in LDS, and its hit ballot
  lanemask_t staged_bal;
};

// front-to-back blend of the view's sorted list over this tile.  The wave leaves as soon as all
// of its 256 pixels are saturated.
// PLAIN: the loop variant without the clamp and the pw test (see tile_stage); it gives up (returns false, F unusable) at the first
// batch that does not qualify, and the caller runs the tile again with PLAIN = false.
template <bool DEPTH, bool PLAIN>
__device__ __forceinline__ bool tile_forward(const TileLds& L, const TileGeom& G, int lane, uint32_t nv, float pyf,
                                             const float (&pxf)[4], const bool (&inside)[4], TileFwd& F) {
  float amin[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    F.Tr[k] = 1.f; F.C0[k] = F.C1[k] = F.C2[k] = F.Dv[k] = 0.f;
    amin[k] = inside[k] ? ALPHA_MIN : 2.f;          // a finished pixel accepts no alpha (alpha <= 0.99)
    F.stop_pos[k] = inside[k] ? 0xffffffffu : 0u;
  }
  F.wlast = 0; F.staged = -1; F.staged_bal = 0ull;
  int jlast = 0, blast = -1;
  bool wave_done = false;
  const int nbf = (int)((nv + U3D_WAVE - 1) / U3D_WAVE);
  for (int b = 0; b < nbf && !wave_done; ++b) {
    bool plain;
    const lanemask_t bal = tile_stage<DEPTH>(L, G, lane, b, nv, plain);
    if (PLAIN && !plain) return false;
    F.staged = b; F.staged_bal = bal;
    const int total = __popcll(bal);
    for (int j = 0; j < total; ++j) {
      const float4 A = L.P0[j];
      const float4 Q = L.P1[j];
      const float2 R = L.P2[j];
      const float invd = DEPTH ? L.D[j] : 0.f;
      const float dy = A.y - pyf;
      const float bdy = A.w * dy, cdy2 = (Q.x * dy) * dy;
      lanemask_t contrib = 0ull, m_stop[4];
      [[maybe_unused]] lanemask_t stopped = 0ull;
      float ar[4];
      if (PLAIN) alpha_run(A.x, A.z, bdy, cdy2, Q.y, L.S[j * L.sstride], pxf[0], ar);
#if U3D_FWD_HOIST
      // Round 6: the saturation test leaves the per-pixel chains.  A pixel saturates once, and with the reference's large splats the pixels
      // of a tile do so within its last few entries; everywhere else `T (1 - alpha) < 1e-4` is false for all 256 pixels.  So each pixel
      // takes  we = ok ? alpha T : 0,  Tn = T - we  (one compare, one select, no scalar mask algebra), ONE test of the smallest Tn over
      // the lane's pixels decides wave-uniformly whether any pixel stops here, and only then the masks of the general form are built.
      // `Tn < 1e-4` is exactly `ok & (T (1 - alpha) < 1e-4)`: a pixel that skips the entry keeps its T, which is never below 1e-4
      // (live: it has not stopped; finished: the stopping entry was not blended).  Same values, bit for bit.
      float wv[4], tn[4];
      lanemask_t m_okv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pw = 0.f;
        if (!PLAIN) {
          const float dx = A.x - pxf[k];
          pw = fmaf(fmaf(A.z, dx, bdy), dx, cdy2);
          ar[k] = Q.y * __builtin_amdgcn_exp2f(pw);
        }
        const float alpha = PLAIN ? ar[k] : min_099(ar[k]);
        m_okv[k] = PLAIN ? __builtin_amdgcn_fcmpf(alpha, amin[k], U3D_FCMP_OGE)
                         : __builtin_amdgcn_fcmpf(pw, 0.f, U3D_FCMP_OLE) & __builtin_amdgcn_fcmpf(alpha, amin[k], U3D_FCMP_OGE);
        wv[k] = mask_sel0(m_okv[k], alpha * F.Tr[k]);
        tn[k] = F.Tr[k] - wv[k];
      }
      float tmin;
      asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(tn[0]), "v"(tn[1]), "v"(tn[2]));
      asm("v_min_f32_e32 %0, %1, %2" : "=v"(tmin) : "v"(tmin), "v"(tn[3]));
      const bool some_stop = __builtin_amdgcn_fcmpf(tmin, T_STOP, U3D_FCMP_OLT) != 0ull;   // wave-uniform
      contrib = (m_okv[0] | m_okv[1]) | (m_okv[2] | m_okv[3]);
      if (some_stop) {   // the pixels that saturate here do not blend the entry: their weight goes back to 0, their T stays
        contrib = 0ull;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          m_stop[k] = __builtin_amdgcn_fcmpf(tn[k], T_STOP, U3D_FCMP_OLT);
          contrib |= m_okv[k] & ~m_stop[k];
          stopped |= m_stop[k];
          wv[k] = mask_sel(m_stop[k], 0.f, wv[k]);
          tn[k] = F.Tr[k] - wv[k];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        F.C0[k] = fmaf(Q.z, wv[k], F.C0[k]);
        F.C1[k] = fmaf(Q.w, wv[k], F.C1[k]);
        F.C2[k] = fmaf(R.x, wv[k], F.C2[k]);
        if (DEPTH) F.Dv[k] = fmaf(invd, wv[k], F.Dv[k]);
        F.Tr[k] = tn[k];
      }
#else
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // one basic block: the four pixels' dependency chains interleave
        float pw = 0.f;
        if (!PLAIN) {
          const float dx = A.x - pxf[k];
          pw = fmaf(fmaf(A.z, dx, bdy), dx, cdy2);
          ar[k] = Q.y * __builtin_amdgcn_exp2f(pw);
        }
        const float araw = ar[k];
        const float alpha = PLAIN ? araw : min_099(araw);
        const lanemask_t m_ok = PLAIN ? __builtin_amdgcn_fcmpf(alpha, amin[k], U3D_FCMP_OGE)
                                      : __builtin_amdgcn_fcmpf(pw, 0.f, U3D_FCMP_OLE) & __builtin_amdgcn_fcmpf(alpha, amin[k], U3D_FCMP_OGE);
        const float w = alpha * F.Tr[k];
        const float test_T = F.Tr[k] - w;          // T (1 - alpha)
        const lanemask_t m_lt = __builtin_amdgcn_fcmpf(test_T, T_STOP, U3D_FCMP_OLT);
        m_stop[k] = m_ok & m_lt;
        const float we = mask_combine_sel0<true>(m_ok, m_lt, contrib, w);   // blended weight (m_ok & ~m_lt), 0 for pixels that skip this Gaussian
        F.C0[k] = fmaf(Q.z, we, F.C0[k]);
        F.C1[k] = fmaf(Q.w, we, F.C1[k]);
        F.C2[k] = fmaf(R.x, we, F.C2[k]);
        if (DEPTH) F.Dv[k] = fmaf(invd, we, F.Dv[k]);
        F.Tr[k] -= we;
        stopped |= m_stop[k];
      }
#endif
      if (contrib != 0ull) { jlast = j; blast = b; }
#if U3D_FWD_HOIST
      if (some_stop) {         // wave-uniform: pixels saturating at this Gaussian (the last few entries of a tile's walk)
#else
      if (stopped != 0ull) {   // rare, wave-uniform: pixels saturating at this Gaussian
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          amin[k] = mask_sel(m_stop[k], 2.f, amin[k]);
          F.stop_pos[k] = __float_as_uint(mask_sel(m_stop[k], R.y, __uint_as_float(F.stop_pos[k])));
        }
        const bool all_done = amin[0] > 1.f && amin[1] > 1.f && amin[2] > 1.f && amin[3] > 1.f;
        if (__ballot(!all_done) == 0ull) { wave_done = true; break; }
      }
    }
    if (blast == b) F.wlast = __builtin_amdgcn_readfirstlane(__float_as_uint(L.P2[jlast].y));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  return true;
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

// Back-to-front walk over sorted positions [1, wmax] of this tile: recovers T by division and
// accumulates each Gaussian's gradients as MOMENTS of q = dL/dG * G over the pixel offsets,
//     m0 = sum q, mx = sum q dx, my = sum q dy, mxx = sum q dx^2, mxy = sum q dx dy, myy = sum q dy^2.
// Cross-tile accumulation without atomics in the common case: the first 64 (P <= 256) or 128 positions of the view's sorted list -- where
// the reference's large, fairly opaque splats put essentially all contributions -- are written per tile to
// part[tile][position][10] (the LDS rows as they are, only the rows the tile touched) and summed over the tiles in a FIXED order, in f64, by
// bwd_reduce_kernel.  Only sorted positions beyond those blocks (sparse / semi-transparent scenes) fall back to f64 global atomics,
// whose ordering does not show at fp32 output precision (the original: one fp32 atomic per pixel and component).
//   Tr = T_final, Rk = T_final (bg . dL/dC), lim = exclusive sorted-position limit per pixel (0: pixel takes no part).
template <bool HAS_INVD, int PB /* partial-row blocks in use */, bool PLAIN /* every batch of the tile qualified in the forward pass */>
__device__ __forceinline__ void tile_backward(const TileLds& L, const TileGeom& G, int lane, uint32_t wmax, int staged,
                                              lanemask_t staged_bal, float pyf, const float (&pxf)[4],
                                              const uint32_t (&lim)[4], float (&Tr)[4], float (&Rk)[4],
                                              const float (&dp0)[4], const float (&dp1)[4], const float (&dp2)[4],
                                              const float (&dinv)[4], float half_w, float half_h, size_t NG,
                                              double* __restrict__ acc, float* __restrict__ pt, uint32_t* __restrict__ pcnt) {
  constexpr int NK = HAS_INVD ? U3D_NACC : U3D_NACC - 1;
  const int bank = (lane >> 2) & 3;
  const bool writer = (lane & 3) == 0 && lane < 16;   // lanes 0, 4, 8, 12: one per DPP bank
  typedef __attribute__((address_space(3))) float lds_float;
  const uint32_t acc_lane = (uint32_t)(uintptr_t)(lds_float*)&L.acc[0][0] + 4u * (uint32_t)bank;
  const int nb = (int)((wmax + U3D_WAVE - 1) / U3D_WAVE);
  for (int b = nb - 1; b >= 0; --b) {
    lanemask_t bal = staged_bal;
    if (b != staged) { bool plain; bal = tile_stage<HAS_INVD>(L, G, lane, b, wmax, plain); staged = b; staged_bal = bal; }
    // entries of this batch with pos <= wmax (the compaction keeps positions ascending)
    const uint32_t lm = wmax - (uint32_t)b * U3D_WAVE;
    const int total = lm >= U3D_WAVE ? __popcll(bal) : __popcll(bal & ((1ull << lm) - 1ull));
    for (int j = total - 1; j >= 0; --j) {
      const float4 A = L.P0[j];
      const float4 Q = L.P1[j];
      const float2 R = L.P2[j];
      const float invd = HAS_INVD ? L.D[j] : 0.f;
      const uint32_t pos = __float_as_uint(R.y);
      const float dy = A.y - pyf;
      const float bdy = A.w * dy, cdy2 = (Q.x * dy) * dy;
      float dx[4], ae[4], ar[4];
      lanemask_t any = 0ull;
      if (PLAIN) alpha_run(A.x, A.z, bdy, cdy2, Q.y, L.S[j * L.sstride], pxf[0], ar);   // the forward's own values, bit for bit
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        dx[k] = A.x - pxf[k];
        float pw = 0.f;
        if (!PLAIN) {
          pw = fmaf(fmaf(A.z, dx[k], bdy), dx[k], cdy2);
          ar[k] = Q.y * __builtin_amdgcn_exp2f(pw);
        }
        const float araw = ar[k];   // opacity * G (alpha before the 0.99 clamp)
        const lanemask_t m_a = (PLAIN ? ~0ull : __builtin_amdgcn_fcmpf(pw, 0.f, U3D_FCMP_OLE)) & __builtin_amdgcn_fcmpf(araw, ALPHA_MIN, U3D_FCMP_OGE);
        // (Round 6 tried skipping the `pos < lim` compares for entries in front of the tile's EARLIEST limit -- a wave-uniform test per entry,
        //  true for most of the walk: +3 % at C2 / C3, +1.7 % at C5.  The scalar branch costs more than the four compares and s_and it saves.)
        ae[k] = mask_combine_sel0<false>(m_a, __builtin_amdgcn_uicmp(pos, lim[k], U3D_ICMP_ULT), any, araw);
      }
      if (any == 0ull) continue;
      float m0, mx, mxx, g_r, g_g, g_b, g_d = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float alpha = PLAIN ? ae[k] : min_099(ae[k]);
        const float om = 1.f - alpha;
        const float rc = __builtin_amdgcn_rcpf(om);
        const float Tn = Tr[k] * rc;            // T in front of this Gaussian
        const float w = alpha * Tn;
        Tr[k] = Tn;
        float cdp = fmaf(R.x, dp2[k], fmaf(Q.w, dp1[k], Q.z * dp0[k]));
        if (HAS_INVD) cdp = fmaf(invd, dinv[k], cdp);
        const float dL_dalpha = fmaf(Tn, cdp, -(Rk[k] * rc));
        Rk[k] = fmaf(w, cdp, Rk[k]);
        const float q = ae[k] * dL_dalpha;    // dL/dG * G
        const float qdx = q * dx[k];
        if (k == 0) {
          g_r = w * dp0[k]; g_g = w * dp1[k]; g_b = w * dp2[k];
          if (HAS_INVD) g_d = w * dinv[k];
          m0 = q; mx = qdx; mxx = qdx * dx[k];
        } else {
          g_r = fmaf(w, dp0[k], g_r); g_g = fmaf(w, dp1[k], g_g); g_b = fmaf(w, dp2[k], g_b);
          if (HAS_INVD) g_d = fmaf(w, dinv[k], g_d);
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
      if (HAS_INVD)
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1"
                     : "+v"(g_d));
      float my = dy * m0, mxy = dy * mx;
      float myy = dy * my;
      // half-row level (lanes i <-> 7-i: banks 0<->1, 2<->3), two values per register: banks {0,2} keep the first
      // operand's sums, banks {1,3} receive the second's; then the row level (i <-> i+8: banks 0<->2, 1<->3) the same way:
      //   mx  <- {mx, my, mxx, mxy}    myy <- {myy, m0, g_r, g_g}    g_b <- {g_b, g_d, g_b, g_d}     (bank index = component)
      if (HAS_INVD)
        asm volatile("s_nop 1\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %4, %4, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
                     "v_add_f32_dpp %0, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                     "v_add_f32_dpp %1, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                     "v_add_f32_dpp %2, %7, %7 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                     "v_add_f32_dpp %3, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                     "v_add_f32_dpp %4, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
                     "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                     "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                     "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                     "v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                     "s_nop 1"
                     : "+v"(mx), "+v"(mxx), "+v"(myy), "+v"(g_r), "+v"(g_b)
                     : "v"(my), "v"(mxy), "v"(m0), "v"(g_g), "v"(g_d));
      else
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
      const uint32_t slot = b < PB ? pos - 1u - (uint32_t)(b * U3D_WAVE) : (uint32_t)j;
      // the four 16-lane DPP rows meet through the LDS crossbar (no LDS memory traffic, no atomics: a 16-lane ds_add_f32 occupied
      // the CU's LDS unit for ~30 cycles, three of them per entry were 70 % of the kernel's LDS time): rows r <-> r^1 by a
      // swizzle, halves by a bpermute; fixed summation order (r0 + r1) + (r2 + r3).  Every entry of a batch owns its row, so the
      // totals are plain stores.
      {
        const int xaddr = (lane ^ 32) << 2;
        float t0 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(mx), 0x401f));    // lane ^ 16
        float t1 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(myy), 0x401f));
        float t2 = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(g_b), 0x401f));
        mx += t0; myy += t1; g_b += t2;
        t0 = __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(mx)));
        t1 = __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(myy)));
        t2 = __int_as_float(__builtin_amdgcn_ds_bpermute(xaddr, __float_as_int(g_b)));
        mx += t0; myy += t1; g_b += t2;
      }
      if (writer) {   // 32-bit LDS addresses (generic-pointer indexing costs a 64-bit multiply-add and a second address register)
        lds_float* sl = reinterpret_cast<lds_float*>(acc_lane + __umul24(slot, 40u));
        sl[0] = mx;
        sl[4] = myy;
        if (bank < (HAS_INVD ? 2 : 1)) sl[8] = g_b;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (b >= PB) {
      // sorted positions beyond the partial-row blocks (sparse / semi-transparent scenes): f64 atomics, compaction-indexed rows
      if (lane < total) {
        const size_t g = G.vbase + G.sorted_id[G.vbase + __float_as_uint(L.P2[lane].y) - 1u];
        const float4 co = G.conic_op[g];
        float m[U3D_NACC];
#pragma unroll
        for (int k = 0; k < U3D_NACC; ++k) m[k] = L.acc[lane][k];
        bool nz = false;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const float v = moment_to_acc<float>(k, m, co.x, co.y, co.z, co.w, half_w, half_h);
          if (v != 0.f) { unsafeAtomicAdd(&acc[(size_t)k * NG + g], (double)v); nz = true; }
        }
        if (nz) {
          G.touched[g] |= U3D_TOUCHED_BIT;   // (every writer ORs the same bit into an otherwise constant word)
          if (G.tw) u3d_mark_touched_list(G.tw, G.gbase + (g - G.vbase), G.tlist, G.tcount, G.item, (uint32_t)(g - G.vbase));
        }
      }
    } else {
      // position-indexed rows of block b: the LDS rows (raw moments) this tile can have touched, 40 B per lane, plain stores;
      // bwd_reduce_kernel sums them over the tiles in a fixed order
      const uint32_t cnt_b = min(wmax - (uint32_t)b * U3D_WAVE, (uint32_t)U3D_WAVE);
      if ((uint32_t)lane < cnt_b) {
#pragma unroll
        for (int k = 0; k < 5; ++k)
          reinterpret_cast<float2*>(pt + b * (U3D_WAVE * 10))[lane * 5 + k] = reinterpret_cast<const float2*>(&L.acc[lane][0])[k];
      }
    }
    if (b > 0) {
#pragma unroll
      for (int k = 0; k < 5; ++k) reinterpret_cast<float2*>(&L.acc[lane][0])[k] = make_float2(0.f, 0.f);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (lane == 0) *pcnt = min(wmax, (uint32_t)(PB * U3D_WAVE));   // rows of this tile in the partial buffer
}

// image rows: 4 consecutive pixels per lane (one 16-byte access when W % 4 == 0)
__device__ __forceinline__ void load4(const float* __restrict__ p, bool vec, const bool (&inside)[4], float (&v)[4]) {
  v[0] = v[1] = v[2] = v[3] = 0.f;
  if (vec) {
    if (inside[0]) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (inside[k]) v[k] = p[k];
  }
}
__device__ __forceinline__ void store4(float* __restrict__ p, bool vec, const bool (&inside)[4], const float (&v)[4]) {
  if (vec) {
    if (inside[0]) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (inside[k]) p[k] = v[k];
  }
}

// The tile kernels carry two loop variants each (PLAIN and not); left alone, the register allocator takes 75-77 VGPRs for the pair
// (6 waves per SIMD).  The single-pass kernel is PINNED: round 4 measured 8 waves per SIMD (64 VGPRs, a handful of values spilled around the rarely
// taken variant) 3 % ahead of the unpinned build; with round 5's alpha_run and round 6's forward the optimum moved to 7 (72 VGPRs, fewer spills):
// render_fb scope, pins 8 / 7 / 6 / none on one box, two rounds: C2 176.0 / 173.2 - 173.7 / 176.1 - 176.3 / 177.2 - 177.5 us, C3 91.8 - 92.4 / 90.5 - 91.0 /
// 92.1 - 92.2 / 92.3, C5 103.3 - 103.8 / 101.4 - 101.5 / 100.7 - 101.0 / 101.0 - 101.5.  The two-pass kernels measured slower pinned to 8 (forward 97 against
// 87 us at C2); the forward is left to the allocator (61 / 66 VGPRs), the backward's object-level instantiation takes 7 (see U3D_BWD_OCC_PIN).
#ifdef U3D_NO_OCC_PIN   /* tools/pmc_spill_probe.sh: the unpinned build, to attribute the scratch-spill share of the kernel's HBM writes */
#define U3D_FULL_OCCUPANCY
#else
#ifndef U3D_OCC_PIN
#define U3D_OCC_PIN 7
#endif
#define U3D_FULL_OCCUPANCY __attribute__((amdgpu_waves_per_eu(U3D_OCC_PIN, U3D_OCC_PIN)))
#endif
// Grid = (T tiles of a view, views [, view slabs of 65535]): the view index comes from the block id, the tile row from a host-made
// magic multiplier (`tile_magic`, 0 = divide), so the prologue has no integer division.  Workgroups are dispatched to the XCDs
// round-robin in linear order x + T * view, which is what u3d_xcd_chunk_in_view assumes.
static_assert(TILE_WAVES == 1, "one wave = one workgroup = one tile");
#ifdef U3D_LPT_EXPERIMENT   /* tools/lpt_tiles.sh: tiles dispatched in the order of a host-made permutation (cost of the previous, identical step) */
__device__ uint32_t* g_lpt_perm = nullptr;
__device__ uint32_t* g_lpt_cost = nullptr;
#define U3D_LPT_MAP                                                                                     \
  if (g_lpt_perm) { const uint32_t lin_ = g_lpt_perm[blockIdx.x + (uint32_t)T * view_u]; view_u = lin_ / (uint32_t)T; tile = (int)(lin_ - view_u * (uint32_t)T); }
#else
#define U3D_LPT_MAP
#endif
#ifdef U3D_TIMELINE   /* tools/tile_timeline.sh: when and where (XCD, CU, SIMD, wave slot) every tile of a launch ran */
__device__ uint4* g_timeline = nullptr;   // two uint4 per tile
#endif
#define U3D_TILE_PROLOGUE(NWAVES)                                                                       \
  const int lane = threadIdx.x, wave = 0;                                                               \
  uint32_t view_u = blockIdx.y + gridDim.y * blockIdx.z;                                                \
  if (view_u * (uint32_t)T >= ntiles_total) return; /* (partial last slab) */                           \
  int tile = (int)u3d_xcd_chunk_in_view(blockIdx.x, view_u, (uint32_t)T, span.P <= 256 ? view_u : 0u);  \
  U3D_LPT_MAP                                                                                           \
  const int view = (int)view_u;                                                                         \
  const uint32_t lid = view_u * (uint32_t)T + (uint32_t)tile;                                           \
  const int ty = tile_magic ? (int)__umulhi((uint32_t)tile, tile_magic) : tile / tiles_x;               \
  const int tx = tile - ty * tiles_x;                                                                   \
  const int py = ty * U3D_TILE + (lane >> 2);                                                           \
  const int px0 = tx * U3D_TILE + 4 * (lane & 3);                                                       \
  const float pyf = (float)py;                                                                          \
  const size_t npix = (size_t)H * W;                                                                    \
  [[maybe_unused]] const size_t pid0 = (size_t)view * npix + (size_t)py * W + px0; /* scalar images */ \
  const size_t cid0 = (size_t)view * 3 * npix + (size_t)py * W + px0; /* 3-channel images */            \
  const bool vec = (W & 3) == 0;                                                                        \
  float pxf[4];                                                                                         \
  bool inside[4];                                                                                       \
  _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                       \
    pxf[k] = (float)(px0 + k);                                                                          \
    inside[k] = px0 + k < W && py < H;                                                                  \
  }                                                                                                     \
  int Pv_;        /* Gaussians of this view's set; first (view, Gaussian) pair (uniform batch: view * P, no division) */ \
  size_t vb_;                                                                                            \
  u3d_view_span(span, view, Pv_, vb_);                                                                   \
  const TileGeom G{sorted_id, sorted_rect, xy, conic_op, rgbd, vb_, tx, ty, rect_indirect, touched, touched_words, touched_words ? u3d_view_gbase(span, view) : 0, \
                   tlist, tcount, (uint32_t)(view / span.vpi)}

// ---- forward (operator path): colour, inverse depth, and the state the backward kernel restarts from --------
// DEPTH = false: the caller drops the inverse-depth output (the reference does: `rendered_image, radii, _ = rasterizer(...)`,
// gaussian_renderer/__init__.py:89) -- no 1/depth staging, one FMA less per pixel and Gaussian, 4 B less per pixel written.
template <bool DEPTH>
__global__ __launch_bounds__(TILE_WAVES * U3D_WAVE) void render_fwd_wave_kernel(
    U3DSpan span, int H, int W, int tiles_x, int T, uint32_t ntiles_total, uint32_t tile_magic, const uint32_t* __restrict__ sorted_id,
    const uint2* __restrict__ sorted_rect, int rect_indirect, const uint32_t* __restrict__ n_vis, const float2* __restrict__ xy,
    const float4* __restrict__ conic_op, const float4* __restrict__ rgbd, const float* __restrict__ bg,
    float* __restrict__ out_color, float* __restrict__ out_invdepth, float* __restrict__ final_T,
    uint32_t* __restrict__ n_contrib, uint32_t* __restrict__ tile_last, U3DLoss loss) {
  uint32_t* const touched = nullptr;
  uint32_t* const touched_words = nullptr;
  uint2* const tlist = nullptr;
  uint32_t* const tcount = nullptr;
  __shared__ float4 sP0[TILE_WAVES][U3D_WAVE], sP1[TILE_WAVES][U3D_WAVE];
  __shared__ float2 sP2[TILE_WAVES][U3D_WAVE];
  __shared__ float sD[TILE_WAVES][DEPTH ? U3D_WAVE : 1];
  __shared__ float sS[TILE_WAVES][U3D_WAVE];
  U3D_TILE_PROLOGUE(TILE_WAVES);
  const TileLds L{sP0[wave], sP1[wave], sP2[wave], sD[wave], nullptr, sS[wave], 1};
  TileFwd F;
  const uint32_t nv = n_vis[view];
  const bool plain = tile_forward<DEPTH, true>(L, G, lane, nv, pyf, pxf, inside, F);
  if (!plain) tile_forward<DEPTH, false>(L, G, lane, nv, pyf, pxf, inside, F);

  float o0[4], o1[4], o2[4], lim[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o0[k] = fmaf(F.Tr[k], bg[0], F.C0[k]); o1[k] = fmaf(F.Tr[k], bg[1], F.C1[k]); o2[k] = fmaf(F.Tr[k], bg[2], F.C2[k]);
    lim[k] = __uint_as_float(F.stop_pos[k]);
  }
  store4(final_T + pid0, vec, inside, F.Tr);
  store4(reinterpret_cast<float*>(n_contrib) + pid0, vec, inside, lim);   // exclusive position limit of the pixel
  store4(out_color + cid0, vec, inside, o0);
  store4(out_color + cid0 + npix, vec, inside, o1);
  store4(out_color + cid0 + 2 * npix, vec, inside, o2);
  if (DEPTH) store4(out_invdepth + pid0, vec, inside, F.Dv);
  if (lane == 0) tile_last[lid] = F.wlast | (plain ? U3D_TILE_PLAIN_BIT : 0u);   // the backward kernel takes the same loop variant
  if (loss.kind != 0) {
    float g0[4], g1[4], g2[4], e = 0.f;
    load4(loss.gt + cid0, vec, inside, g0);
    load4(loss.gt + cid0 + npix, vec, inside, g1);
    load4(loss.gt + cid0 + 2 * npix, vec, inside, g2);
    const LossCtx lc = loss_ctx(loss, bg);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      e += loss_term(lc, loss_weight(lc, inside[k], g0[k], g1[k], g2[k]), o0[k] - g0[k], o1[k] - g1[k], o2[k] - g2[k]);
    e = u3d_wave_sum(e);
    if (lane == 0) loss.partial[lid] = e;
  }
}

// ---- backward (operator path, and second pass of the two-pass fused loss) ------------------------------------
// Occupancy: the allocator gives this kernel 76 VGPRs = 6 waves per SIMD.  Round 6: the object-level instantiation without inverse-depth gradients (PB == 1, !HAS_INVD: what the reference's call produces) is pinned to 7 (72 VGPRs,
// 20 B of scratch per lane) like the single-pass kernel: two-pass route, render_bwd 161.0 -> 156.2 us at C2 in two alternating pairs; the scene-level
// one (PB == 2) measured -0.8 % at C3 but +1.6 % at C5 pinned and is left to the allocator (1 .. 10 = no constraint).  U3D_BWD_OCC_PIN=0: no pin at all.
#ifndef U3D_BWD_OCC_PIN
#define U3D_BWD_OCC_PIN 7
#endif
#if U3D_BWD_OCC_PIN
#define U3D_BWD_OCCUPANCY __attribute__((amdgpu_waves_per_eu((PB == 1 && !HAS_INVD) ? U3D_BWD_OCC_PIN : 1, (PB == 1 && !HAS_INVD) ? U3D_BWD_OCC_PIN : 10)))
#else
#define U3D_BWD_OCCUPANCY
#endif
template <bool HAS_INVD, int PB>
__global__ __launch_bounds__(TILE_WAVES * U3D_WAVE) U3D_BWD_OCCUPANCY void render_bwd_wave_kernel(
    U3DSpan span, int H, int W, int tiles_x, int T, uint32_t ntiles_total, uint32_t tile_magic, size_t NG,
    const uint32_t* __restrict__ sorted_id, const uint2* __restrict__ sorted_rect, int rect_indirect, const float2* __restrict__ xy,
    const float4* __restrict__ conic_op, const float4* __restrict__ rgbd, const float* __restrict__ bg,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_dinvdepth, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const uint32_t* __restrict__ tile_last, double* __restrict__ acc,
    float* __restrict__ part, const float* __restrict__ out_color, uint32_t* __restrict__ touched, uint32_t* __restrict__ touched_words,
    U3DLoss loss) {
  uint2* const tlist = nullptr;          // (the list is the fused step's: U3D_FLAG_SPARSE_BWD)
  uint32_t* const tcount = nullptr;
  __shared__ float4 sP0[TILE_WAVES][U3D_WAVE], sP1[TILE_WAVES][U3D_WAVE];
  __shared__ float2 sP2[TILE_WAVES][U3D_WAVE];
  __shared__ float sD[TILE_WAVES][HAS_INVD ? U3D_WAVE : 1];
  __shared__ __attribute__((aligned(8))) float sAcc[TILE_WAVES][U3D_WAVE][10];
  __shared__ float sS[TILE_WAVES][HAS_INVD ? U3D_WAVE : 1];   // (with depth gradients the rows' tenth float is in use)
  U3D_TILE_PROLOGUE(TILE_WAVES);
  const TileLds L{sP0[wave], sP1[wave], sP2[wave], sD[wave], sAcc[wave], HAS_INVD ? sS[wave] : &sAcc[wave][0][9], HAS_INVD ? 1 : 10};
#pragma unroll
  for (int k = 0; k < 5; ++k) reinterpret_cast<float2*>(&sAcc[wave][lane][0])[k] = make_float2(0.f, 0.f);

  float Tr[4], Rk[4], dp0[4], dp1[4], dp2[4], dinv[4], limf[4];
  uint32_t lim[4];
  load4(final_T + pid0, vec, inside, Tr);
  load4(reinterpret_cast<const float*>(n_contrib) + pid0, vec, inside, limf);   // 0 for pixels outside the image
  if (loss.kind != 0) {
    float x0[4], x1[4], x2[4], g0[4], g1[4], g2[4];
    load4(out_color + cid0, vec, inside, x0);
    load4(out_color + cid0 + npix, vec, inside, x1);
    load4(out_color + cid0 + 2 * npix, vec, inside, x2);
    load4(loss.gt + cid0, vec, inside, g0);
    load4(loss.gt + cid0 + npix, vec, inside, g1);
    load4(loss.gt + cid0 + 2 * npix, vec, inside, g2);
    const float sc = loss.dloss[0] * loss.inv_count;
    const LossCtx lc = loss_ctx(loss, bg);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dinv[k] = 0.f;
      loss_seed(lc, loss_weight(lc, inside[k], g0[k], g1[k], g2[k]), sc, x0[k] - g0[k], x1[k] - g1[k], x2[k] - g2[k], dp0[k], dp1[k], dp2[k]);
    }
    if (dL_dcolor) {
      // a further image-space term of the caller's objective (the reference adds lambda * LPIPS(rendered) after iteration
      // start_lpips_after, train_network.py:284-300): its dL/dcolor joins the in-kernel loss seed
      float e0[4], e1[4], e2[4];
      load4(dL_dcolor + cid0, vec, inside, e0);
      load4(dL_dcolor + cid0 + npix, vec, inside, e1);
      load4(dL_dcolor + cid0 + 2 * npix, vec, inside, e2);
#pragma unroll
      for (int k = 0; k < 4; ++k) { dp0[k] += e0[k]; dp1[k] += e1[k]; dp2[k] += e2[k]; }
    }
  } else {
    load4(dL_dcolor + cid0, vec, inside, dp0);
    load4(dL_dcolor + cid0 + npix, vec, inside, dp1);
    load4(dL_dcolor + cid0 + 2 * npix, vec, inside, dp2);
    if (HAS_INVD) load4(dL_dinvdepth + pid0, vec, inside, dinv);
    else dinv[0] = dinv[1] = dinv[2] = dinv[3] = 0.f;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    lim[k] = __float_as_uint(limf[k]);
    Rk[k] = Tr[k] * (bg[0] * dp0[k] + bg[1] * dp1[k] + bg[2] * dp2[k]);   // T_final * (bg . dL/dC)
  }
  const uint32_t tl = tile_last[lid];
  if (tl & U3D_TILE_PLAIN_BIT)
    tile_backward<HAS_INVD, PB, true>(L, G, lane, tl & ~U3D_TILE_PLAIN_BIT, -1, 0ull, pyf, pxf, lim, Tr, Rk, dp0, dp1, dp2, dinv, 0.5f * (float)W,
                          0.5f * (float)H, NG, acc, part + (size_t)lid * (PB * U3D_WAVE * 10),
                          reinterpret_cast<uint32_t*>(part + (size_t)ntiles_total * BWD_PART_STRIDE) + lid);
  else
    tile_backward<HAS_INVD, PB, false>(L, G, lane, tl, -1, 0ull, pyf, pxf, lim, Tr, Rk, dp0, dp1, dp2, dinv, 0.5f * (float)W,
                          0.5f * (float)H, NG, acc, part + (size_t)lid * (PB * U3D_WAVE * 10),
                          reinterpret_cast<uint32_t*>(part + (size_t)ntiles_total * BWD_PART_STRIDE) + lid);
}

// ---- forward + backward in ONE kernel (training step of the fused render-loss path) -----------------
// The render loss needs nothing but the pixel's own colour and gt, so a tile can blend front to back, evaluate its
// loss term and seed dL/dcolor, and immediately walk the same LDS-resident batch back to front: final_T, the position
// limits and the colour image never round-trip through HBM, the Gaussian batch is staged once, and one prologue
// disappears.  dL/dloss is taken as 1 (the result is linear in it; the host scales the stored gradient).
template <int PB>
__global__ __launch_bounds__(TILE_WAVES * U3D_WAVE) U3D_FULL_OCCUPANCY void render_fb_wave_kernel(
    U3DSpan span, int H, int W, int tiles_x, int T, uint32_t ntiles_total, uint32_t tile_magic, size_t NG,
    const uint32_t* __restrict__ sorted_id, const uint2* __restrict__ sorted_rect, int rect_indirect, const uint32_t* __restrict__ n_vis,
    const float2* __restrict__ xy, const float4* __restrict__ conic_op, const float4* __restrict__ rgbd,
    const float* __restrict__ bg, float* __restrict__ out_color, double* __restrict__ acc, float* __restrict__ part,
    uint32_t* __restrict__ touched, uint32_t* __restrict__ touched_words, uint2* __restrict__ tlist, uint32_t* __restrict__ tcount,
    U3DLoss loss) {
  __shared__ float4 sP0[TILE_WAVES][U3D_WAVE], sP1[TILE_WAVES][U3D_WAVE];
  __shared__ float2 sP2[TILE_WAVES][U3D_WAVE];
  __shared__ __attribute__((aligned(8))) float sAcc[TILE_WAVES][U3D_WAVE][10];
#ifdef U3D_TIMELINE
  const uint64_t tl_t0 = __builtin_amdgcn_s_memrealtime(), tl_c0 = __builtin_amdgcn_s_memtime();
#endif
#ifdef U3D_PRIO_FWD
  __builtin_amdgcn_s_setprio(U3D_PRIO_FWD);
#endif
  U3D_TILE_PROLOGUE(TILE_WAVES);
  const TileLds L{sP0[wave], sP1[wave], sP2[wave], nullptr, sAcc[wave], &sAcc[wave][0][9], 10};
#pragma unroll
  for (int k = 0; k < 5; ++k) reinterpret_cast<float2*>(&sAcc[wave][lane][0])[k] = make_float2(0.f, 0.f);
  TileFwd F;
  const uint32_t nv = n_vis[view];
  const bool plain = tile_forward<false, true>(L, G, lane, nv, pyf, pxf, inside, F);
  if (!plain) tile_forward<false, false>(L, G, lane, nv, pyf, pxf, inside, F);
#ifdef U3D_TIMELINE
  const uint64_t tl_t1 = __builtin_amdgcn_s_memrealtime();
#endif

  // loss term, dL/dcolor seed
  float dp0[4], dp1[4], dp2[4], dinv[4], Rk[4], o0[4], o1[4], o2[4], g0[4], g1[4], g2[4];
  float e = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o0[k] = fmaf(F.Tr[k], bg[0], F.C0[k]); o1[k] = fmaf(F.Tr[k], bg[1], F.C1[k]); o2[k] = fmaf(F.Tr[k], bg[2], F.C2[k]);
  }
  load4(loss.gt + cid0, vec, inside, g0);
  load4(loss.gt + cid0 + npix, vec, inside, g1);
  load4(loss.gt + cid0 + 2 * npix, vec, inside, g2);
  if (out_color) {
    store4(out_color + cid0, vec, inside, o0);
    store4(out_color + cid0 + npix, vec, inside, o1);
    store4(out_color + cid0 + 2 * npix, vec, inside, o2);
  }
  const LossCtx lc = loss_ctx(loss, bg);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float d0 = o0[k] - g0[k], d1 = o1[k] - g1[k], d2 = o2[k] - g2[k];
    const float wk = loss_weight(lc, inside[k], g0[k], g1[k], g2[k]);
    e += loss_term(lc, wk, d0, d1, d2);
    loss_seed(lc, wk, loss.inv_count, d0, d1, d2, dp0[k], dp1[k], dp2[k]);   // dL/dloss == 1
    dinv[k] = 0.f;
    F.Tr[k] = inside[k] ? F.Tr[k] : 0.f;
    Rk[k] = F.Tr[k] * (bg[0] * dp0[k] + bg[1] * dp1[k] + bg[2] * dp2[k]);   // T_final * (bg . dL/dC)
  }
  e = u3d_wave_sum(e);
  if (lane == 0) loss.partial[lid] = e;
#ifdef U3D_LPT_EXPERIMENT
  if (lane == 0 && g_lpt_cost) g_lpt_cost[lid] = F.wlast;
#endif
#ifdef U3D_TIMELINE
  const uint64_t tl_t2 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef U3D_PRIO_BWD
  __builtin_amdgcn_s_setprio(U3D_PRIO_BWD);
#endif

  if (plain)
    tile_backward<false, PB, true>(L, G, lane, F.wlast, F.staged, F.staged_bal, pyf, pxf, F.stop_pos, F.Tr, Rk, dp0, dp1, dp2, dinv,
                       0.5f * (float)W, 0.5f * (float)H, NG, acc, part + (size_t)lid * (PB * U3D_WAVE * 10),
                          reinterpret_cast<uint32_t*>(part + (size_t)ntiles_total * BWD_PART_STRIDE) + lid);
  else
    tile_backward<false, PB, false>(L, G, lane, F.wlast, F.staged, F.staged_bal, pyf, pxf, F.stop_pos, F.Tr, Rk, dp0, dp1, dp2, dinv,
                       0.5f * (float)W, 0.5f * (float)H, NG, acc, part + (size_t)lid * (PB * U3D_WAVE * 10),
                          reinterpret_cast<uint32_t*>(part + (size_t)ntiles_total * BWD_PART_STRIDE) + lid);
#ifdef U3D_TIMELINE
  if (lane == 0 && g_timeline) {
    const uint64_t tl_t3 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one counter for the whole device
    const uint64_t tl_c1 = __builtin_amdgcn_s_memtime();       // shader clock
    // HW_ID (register 4): wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13];  XCC_ID (register 20): [3:0]
    g_timeline[2 * (size_t)lid] = make_uint4((uint32_t)tl_t0, (uint32_t)tl_t3, __builtin_amdgcn_s_getreg((31 << 11) | 4),
                                             (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u) | (F.wlast << 4));
    // after the forward walk, after the loss epilogue, shader clocks between entry and exit
    g_timeline[2 * (size_t)lid + 1] = make_uint4((uint32_t)tl_t1, (uint32_t)tl_t2, (uint32_t)(tl_c1 - tl_c0), plain ? 1u : 0u);
  }
#endif
}

// acc[k][view*P + sorted_id[sp]] += sum over a slice of the view's tiles (ascending) of part[tile][sp][k], in f64;
// the nsplit slices of a view meet in an f64 atomic (order-insensitive at fp32 output precision).
// One thread per (position, component) element of a tile's [64][10] block, so a tile is one contiguous read; only the
// positions some tile of the slice touched (cmax, usually ~20 of 64) are read at all.
constexpr int REDUCE_THREADS = U3D_WAVE * 10;
// tiles in flight per thread: 32 in the single-block kernel (object level), 16 in the generic one (measured, tools/sweep_ru.sh:
// C3 10.1 -> 9.3 us, C4 17.0 -> 15.5, C5 13.0 -> 11.8 with 16; C2 11.3 -> 11.6; 64 is slower everywhere)
#define RU 16
#define RU1 32
template <int PB>
__global__ __launch_bounds__(REDUCE_THREADS) void bwd_reduce_kernel(U3DSpan span, int T, int NK, int nsplit, size_t NG, float half_w, float half_h,
                                                                   const uint32_t* __restrict__ sorted_id,
                                                                   const float4* __restrict__ conic_op,
                                                                   const float* __restrict__ part,
                                                                   const uint32_t* __restrict__ part_cnt,
                                                                   double* __restrict__ acc, uint32_t* __restrict__ touched,
                                                                   uint32_t* __restrict__ touched_words, int n_loss,
                                                                   const float* __restrict__ loss_partial, float inv_count,
                                                                   float* __restrict__ loss_out, uint2* __restrict__ tlist,
                                                                   uint32_t* __restrict__ tcount, float* __restrict__ zero_fill,
                                                                   size_t zero_floats) {
  __shared__ double s_sum[U3D_WAVE][10];
  __shared__ uint32_t s_cmax;
  if ((int)blockIdx.y > nsplit) {
    // rows beyond the loss row (U3D_FLAG_SPARSE_BWD): zero-fill the gradient buffer the backward half will scatter into -- 16-byte
    // stores from workgroups that run beside the latency-bound tile chains of the reduction, i.e. off the step's critical path
    const size_t nb = (size_t)gridDim.x * (gridDim.y - nsplit - 1), bid = (size_t)(blockIdx.y - nsplit - 1) * gridDim.x + blockIdx.x;
    const size_t n4 = zero_floats >> 2;
    float4* z4 = reinterpret_cast<float4*>(zero_fill);
    for (size_t e = bid * REDUCE_THREADS + threadIdx.x; e < n4; e += nb * REDUCE_THREADS) z4[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bid == 0 && threadIdx.x < (zero_floats & 3)) zero_fill[(n4 << 2) + threadIdx.x] = 0.f;
    return;
  }
  if ((int)blockIdx.y == nsplit) {
    // extra row of the grid: fixed-order sum of the per-tile loss partials (replaces a separate launch)
    if (blockIdx.x != 0) return;
    float* sm = reinterpret_cast<float*>(&s_sum[0][0]);
    constexpr int NT = REDUCE_THREADS;
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
  const int view = blockIdx.x, sp = threadIdx.x / 10, k = threadIdx.x - sp * 10;
  const int per = (T + nsplit - 1) / nsplit;
  const int t0 = blockIdx.y * per, t1 = min(T, t0 + per);
  const uint32_t* cnt = part_cnt + (size_t)view * T;
  if (threadIdx.x == 0) s_cmax = 0u;
  __syncthreads();
  {
    uint32_t c = 0u;
    for (int t = t0 + (int)threadIdx.x; t < t1; t += REDUCE_THREADS) c = max(c, cnt[t]);
    if (c != 0u) atomicMax(&s_cmax, c);
  }
  __syncthreads();
  const uint32_t cmax = s_cmax;
  if (cmax == 0u) return;
  constexpr size_t tstride = (size_t)PB * (U3D_WAVE * 10);   // floats per tile
#pragma unroll
  for (int h = 0; h < PB; ++h) {   // block of 64 sorted positions (PB == 1: exactly the single-block kernel)
    if (h > 0 && cmax <= (uint32_t)(h * U3D_WAVE)) break;
    const uint32_t spos = (uint32_t)(h * U3D_WAVE + sp);
    double a = 0.0;
    if (k < NK && spos < cmax) {
      const float* base = part + (size_t)view * T * tstride + h * (U3D_WAVE * 10) + threadIdx.x;
      // loads are unconditional below cmax (rows a tile did not write hold stale bytes, discarded by the select) so that a
      // whole group of tiles is in flight at once; accumulation order stays ascending in t within each of the two chains
      double a0 = 0.0, a1 = 0.0;
      int t = t0;
      for (; t + RU - 1 < t1; t += RU) {
        float v[RU];
        uint32_t c[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
          c[u] = cnt[t + u];
          v[u] = base[(size_t)(t + u) * tstride];
        }
#pragma unroll
        for (int u = 0; u < RU; u += 2) {
          a0 += spos < c[u] ? (double)v[u] : 0.0;
          a1 += spos < c[u + 1] ? (double)v[u + 1] : 0.0;
        }
      }
      for (; t < t1; ++t) {
        const float v0 = base[(size_t)t * tstride];
        a0 += spos < cnt[t] ? (double)v0 : 0.0;
      }
      a = a0 + a1;
    }
    // raw moment sums of this slice -> accumulator values (linear, so slices can be converted independently)
    if (h > 0) __syncthreads();   // the previous block's readers of s_sum are done
    s_sum[sp][k] = a;
    __syncthreads();
    const bool mine = k < NK && spos < cmax;
    if (h == PB - 1 && !mine) return;
    if (mine) {
      double m[U3D_NACC];
#pragma unroll
      for (int j = 0; j < U3D_NACC; ++j) m[j] = j < NK ? s_sum[sp][j] : 0.0;
      bool any = false;
#pragma unroll
      for (int j = 0; j < U3D_NACC; ++j) any = any || m[j] != 0.0;
      if (any) {
        int Pv;
        size_t vb;
        u3d_view_span(span, view, Pv, vb);
        const size_t g = vb + sorted_id[vb + spos];
        const float4 co = conic_op[g];
        const double outv = moment_to_acc<double>(k, m, co.x, co.y, co.z, co.w, half_w, half_h);
        if (outv != 0.0) unsafeAtomicAdd(&acc[(size_t)k * NG + g], outv);
        if (k == 0) {
          touched[g] |= U3D_TOUCHED_BIT;   // this (view, Gaussian) has a non-zero row
          if (touched_words)
            u3d_mark_touched_list(touched_words, u3d_view_gbase(span, view) + (g - vb), tlist, tcount, (uint32_t)(view / span.vpi), (uint32_t)(g - vb));
        }
      }
    }
  }
}

// Single-block form (object level: every tile's rows fit the first 64 positions): the same reduction with the block loop
// and the per-tile stride folded away (the generic instantiation measured 2 us slower at C2).
__global__ __launch_bounds__(REDUCE_THREADS) void bwd_reduce1_kernel(U3DSpan span, int T, int NK, int nsplit, size_t NG, float half_w, float half_h,
                                                                   const uint32_t* __restrict__ sorted_id,
                                                                   const float4* __restrict__ conic_op,
                                                                   const float* __restrict__ part,
                                                                   const uint32_t* __restrict__ part_cnt,
                                                                   double* __restrict__ acc, uint32_t* __restrict__ touched,
                                                                   uint32_t* __restrict__ touched_words, int n_loss,
                                                                   const float* __restrict__ loss_partial, float inv_count,
                                                                   float* __restrict__ loss_out) {
  __shared__ double s_sum[U3D_WAVE][10];
  __shared__ uint32_t s_cmax;
  if ((int)blockIdx.y == nsplit) {
    // extra row of the grid: fixed-order sum of the per-tile loss partials (replaces a separate launch)
    if (blockIdx.x != 0) return;
    float* sm = reinterpret_cast<float*>(&s_sum[0][0]);
    constexpr int NT = REDUCE_THREADS;
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
  const int view = blockIdx.x, sp = threadIdx.x / 10, k = threadIdx.x - sp * 10;
  const int per = (T + nsplit - 1) / nsplit;
  const int t0 = blockIdx.y * per, t1 = min(T, t0 + per);
  const uint32_t* cnt = part_cnt + (size_t)view * T;
  if (threadIdx.x == 0) s_cmax = 0u;
  __syncthreads();
  {
    uint32_t c = 0u;
    for (int t = t0 + (int)threadIdx.x; t < t1; t += REDUCE_THREADS) c = max(c, cnt[t]);
    if (c != 0u) atomicMax(&s_cmax, c);
  }
  __syncthreads();
  const uint32_t cmax = s_cmax;
  if (cmax == 0u) return;
  double a = 0.0;
  if (k < NK && (uint32_t)sp < cmax) {
    const float* base = part + (size_t)view * T * (U3D_WAVE * 10) + threadIdx.x;
    // loads are unconditional below cmax (rows a tile did not write hold stale bytes, discarded by the select) so that a
    // whole group of tiles is in flight at once; accumulation order stays ascending in t within each of the two chains
    double a0 = 0.0, a1 = 0.0;
    int t = t0;
    for (; t + RU1 - 1 < t1; t += RU1) {
      float v[RU1];
      uint32_t c[RU1];
#pragma unroll
      for (int u = 0; u < RU1; ++u) {
        c[u] = cnt[t + u];
        v[u] = base[(size_t)(t + u) * (U3D_WAVE * 10)];
      }
#pragma unroll
      for (int u = 0; u < RU1; u += 2) {
        a0 += (uint32_t)sp < c[u] ? (double)v[u] : 0.0;
        a1 += (uint32_t)sp < c[u + 1] ? (double)v[u + 1] : 0.0;
      }
    }
    for (; t < t1; ++t) {
      const float v0 = base[(size_t)t * (U3D_WAVE * 10)];
      a0 += (uint32_t)sp < cnt[t] ? (double)v0 : 0.0;
    }
    a = a0 + a1;
  }
  // raw moment sums of this slice -> accumulator values (linear, so slices can be converted independently)
  s_sum[sp][k] = a;
  __syncthreads();
  if (k >= NK || (uint32_t)sp >= cmax) return;
  double m[U3D_NACC];
#pragma unroll
  for (int j = 0; j < U3D_NACC; ++j) m[j] = j < NK ? s_sum[sp][j] : 0.0;
  bool any = false;
#pragma unroll
  for (int j = 0; j < U3D_NACC; ++j) any = any || m[j] != 0.0;
  if (!any) return;
  int Pv;
  size_t vb;
  u3d_view_span(span, view, Pv, vb);
  const size_t g = vb + sorted_id[vb + sp];
  const float4 co = conic_op[g];
  const double outv = moment_to_acc<double>(k, m, co.x, co.y, co.z, co.w, half_w, half_h);
  if (outv != 0.0) unsafeAtomicAdd(&acc[(size_t)k * NG + g], outv);
  if (k == 0) {
    touched[g] |= U3D_TOUCHED_BIT;   // this (view, Gaussian) has a non-zero row
    if (touched_words) u3d_mark_touched(touched_words, u3d_view_gbase(span, view) + (g - vb));
  }
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

// launch shape of the tile kernels (see U3D_TILE_PROLOGUE)
struct TileGrid { dim3 grid; uint32_t magic; };
static inline TileGrid tile_grid(const u3d_raster_desc& d, int tiles_x, int T) {
  const uint32_t NV = (uint32_t)(d.n_items * d.views_per_item);
  const uint32_t gy = NV < 65535u ? NV : 65535u, gz = (NV + gy - 1u) / gy;
  // ty = (tile * magic) >> 32 is exact for every tile < T when T * tiles_x < 2^32 (error term (magic * tiles_x - 2^32) < tiles_x)
  uint32_t magic = 0u;
  if (tiles_x > 1 && (uint64_t)T * (uint64_t)tiles_x < (1ull << 32)) magic = (uint32_t)(((1ull << 32) + (uint64_t)tiles_x - 1ull) / (uint64_t)tiles_x);
  return TileGrid{dim3((uint32_t)T, gy, gz), magic};
}

void u3d_launch_render_fwd(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, float* out_color,
                           float* out_invdepth, const U3DLoss& loss, hipStream_t s) {
  const int tiles_x = (d.image_width + U3D_TILE - 1) / U3D_TILE, tiles_y = (d.image_height + U3D_TILE - 1) / U3D_TILE;
  const int T = tiles_x * tiles_y;
  const uint32_t ntiles = (uint32_t)(d.n_items * d.views_per_item * T);
  if (ntiles == 0) return;
  const TileGrid tg = tile_grid(d, tiles_x, T);
  auto* kern = out_invdepth ? render_fwd_wave_kernel<true> : render_fwd_wave_kernel<false>;
  hipLaunchKernelGGL(kern, tg.grid, dim3(TILE_WAVES * U3D_WAVE), 0, s, u3d_span(d), d.image_height, d.image_width,
                     tiles_x, T, ntiles, tg.magic, b.sorted_id, u3d_rect_indirect(d) ? b.rect : b.sorted_rect, u3d_rect_indirect(d), b.n_vis, b.xy, b.conic_op, b.rgbd, bg, out_color,
                     out_invdepth, b.final_T, b.n_contrib, b.tile_last, loss);
}

void u3d_launch_render_fb(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, float* out_color,
                          const U3DLoss& loss, double* acc, float* part, float* loss_out, hipStream_t s,
                          float* zero_fill, size_t zero_floats, bool list_touched) {
  const int tiles_x = (d.image_width + U3D_TILE - 1) / U3D_TILE, tiles_y = (d.image_height + U3D_TILE - 1) / U3D_TILE;
  const int T = tiles_x * tiles_y;
  const uint32_t ntiles = (uint32_t)(d.n_items * d.views_per_item * T);
  const size_t NG = (size_t)d.views_per_item * u3d_total_P(d);
  if (ntiles == 0 || NG == 0) return;
  const TileGrid tg = tile_grid(d, tiles_x, T);
  uint32_t* tw = u3d_uses_touched_words(d) ? b.touched_words : nullptr;
  uint2* tl = (list_touched && tw) ? b.touched_list : nullptr;
  uint32_t* tc = (list_touched && tw) ? b.touched_count : nullptr;
#ifdef U3D_LPT_EXPERIMENT
  {
    static int launches = 0;
    static uint32_t *d_cost = nullptr, *d_perm = nullptr;
    static const int mode = getenv("U3D_LPT_MODE") ? atoi(getenv("U3D_LPT_MODE")) : 0;   // 1 heavy first, 2 light first, 3 random (control), 4 heavy first within views interleaved
    if (mode > 0 && u3d_part_blocks(d) == 1) {
      ++launches;
      if (launches == 1) {
        (void)hipMalloc(&d_cost, sizeof(uint32_t) * ntiles); (void)hipMalloc(&d_perm, sizeof(uint32_t) * ntiles);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lpt_cost), &d_cost, sizeof(d_cost));
      }
      if (launches == 30) {
        (void)hipStreamSynchronize(s);
        std::vector<uint32_t> cost(ntiles), perm(ntiles);
        (void)hipMemcpy(cost.data(), d_cost, sizeof(uint32_t) * ntiles, hipMemcpyDeviceToHost);
        for (uint32_t i = 0; i < ntiles; ++i) perm[i] = i;
        if (mode == 1) std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
        if (mode == 2) std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return cost[a] < cost[b]; });
        if (mode == 3) { uint32_t st = 12345u; for (uint32_t i = ntiles - 1; i > 0; --i) { st = st * 1664525u + 1013904223u; std::swap(perm[i], perm[st % (i + 1)]); } }
        if (mode == 4) {   // per view: its tiles heavy first; block b -> (view b % NV, rank b / NV): what a production form could compute per view
          const uint32_t NV = ntiles / (uint32_t)T;
          std::vector<uint32_t> within(ntiles);
          for (uint32_t v = 0; v < NV; ++v) {
            std::vector<uint32_t> t((size_t)T);
            for (int j = 0; j < T; ++j) t[j] = v * (uint32_t)T + (uint32_t)j;
            std::stable_sort(t.begin(), t.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
            for (int j = 0; j < T; ++j) within[v * (uint32_t)T + (uint32_t)j] = t[j];
          }
          for (uint32_t b2 = 0; b2 < ntiles; ++b2) perm[b2] = within[(b2 % NV) * (uint32_t)T + b2 / NV];
        }
        double mean = 0; uint32_t mx = 0; for (auto c2 : cost) { mean += c2; mx = c2 > mx ? c2 : mx; }
        fprintf(stderr, "[lpt] mode %d: %u tiles, cost mean %.2f max %u; first %u last %u\n", mode, ntiles, mean / ntiles, mx, cost[perm[0]], cost[perm[ntiles - 1]]);
        (void)hipMemcpy(d_perm, perm.data(), sizeof(uint32_t) * ntiles, hipMemcpyHostToDevice);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lpt_perm), &d_perm, sizeof(d_perm));
        uint32_t* nul = nullptr;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lpt_cost), &nul, sizeof(nul));
      }
    }
  }
#endif
#ifdef U3D_TIMELINE
  {
    static int launches = 0;
    static uint4* d_tl = nullptr;
    static const char* out = getenv("U3D_TIMELINE_OUT");
    static const int at = getenv("U3D_TIMELINE_AT") ? atoi(getenv("U3D_TIMELINE_AT")) : 150;   // (the first ~100 launches of a process run below the operating clock)
    if (out) {
      ++launches;
      if (launches == at) {   // launches at .. at + 5 record (back to back with their steps); the last one is kept
        (void)hipMalloc(&d_tl, 2 * sizeof(uint4) * ntiles);
        (void)hipMemset(d_tl, 0, 2 * sizeof(uint4) * ntiles);
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &d_tl, sizeof(d_tl));
      }
      if (launches == at + 6) {
        (void)hipStreamSynchronize(s);
        std::vector<uint4> h(2 * (size_t)ntiles);
        (void)hipMemcpy(h.data(), d_tl, 2 * sizeof(uint4) * ntiles, hipMemcpyDeviceToHost);
        uint4* nul = nullptr;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &nul, sizeof(nul));
        if (FILE* f = fopen(out, "wb")) { fwrite(h.data(), sizeof(uint4), 2 * (size_t)ntiles, f); fclose(f); }
        fprintf(stderr, "[timeline] %u tiles of launch %d -> %s\n", ntiles, at + 5, out);
      }
    }
  }
#endif
  if (u3d_part_blocks(d) == 1)
    hipLaunchKernelGGL(render_fb_wave_kernel<1>, tg.grid, dim3(TILE_WAVES * U3D_WAVE), 0, s, u3d_span(d), d.image_height, d.image_width,
                       tiles_x, T, ntiles, tg.magic, NG, b.sorted_id, u3d_rect_indirect(d) ? b.rect : b.sorted_rect, u3d_rect_indirect(d), b.n_vis, b.xy, b.conic_op, b.rgbd, bg, out_color,
                       acc, part, b.clamped, tw, tl, tc, loss);
  else
    hipLaunchKernelGGL(render_fb_wave_kernel<U3D_PART_BLOCKS>, tg.grid, dim3(TILE_WAVES * U3D_WAVE), 0, s, u3d_span(d), d.image_height,
                       d.image_width, tiles_x, T, ntiles, tg.magic, NG, b.sorted_id, u3d_rect_indirect(d) ? b.rect : b.sorted_rect, u3d_rect_indirect(d), b.n_vis, b.xy, b.conic_op, b.rgbd, bg,
                       out_color, acc, part, b.clamped, tw, tl, tc, loss);
  const int nsplit = bwd_reduce_split(T, d.n_items * d.views_per_item);
  const int NVi = d.n_items * d.views_per_item;
  if (u3d_part_blocks(d) == 1) {
    hipLaunchKernelGGL(bwd_reduce1_kernel, dim3(NVi, nsplit + 1), dim3(REDUCE_THREADS), 0, s,
                       u3d_span(d), T, U3D_NACC - 1, nsplit, NG, 0.5f * (float)d.image_width, 0.5f * (float)d.image_height, b.sorted_id, b.conic_op, part,
                       reinterpret_cast<const uint32_t*>(part + (size_t)ntiles * BWD_PART_STRIDE), acc, b.clamped, tw, (int)ntiles, loss.partial,
                       loss.inv_count, loss_out);
  } else {
    // zero-fill rows (U3D_FLAG_SPARSE_BWD): ~256 extra workgroups of 640 threads, at most one per 64 KB to fill
    int zrows = 0;
    if (zero_fill && zero_floats > 0) {
      const size_t want = (zero_floats * sizeof(float) + 65535) / 65536;
      zrows = (int)((want < 256 ? want : 256) + (size_t)NVi - 1) / NVi;
      if (zrows < 1) zrows = 1;
    }
    hipLaunchKernelGGL(bwd_reduce_kernel<U3D_PART_BLOCKS>, dim3(NVi, nsplit + 1 + zrows), dim3(REDUCE_THREADS), 0, s,
                       u3d_span(d), T, U3D_NACC - 1, nsplit, NG, 0.5f * (float)d.image_width, 0.5f * (float)d.image_height, b.sorted_id, b.conic_op, part,
                       reinterpret_cast<const uint32_t*>(part + (size_t)ntiles * BWD_PART_STRIDE), acc, b.clamped, tw, (int)ntiles, loss.partial,
                       loss.inv_count, loss_out, tl, tc, zero_fill, zero_floats);
  }
}

void u3d_launch_render_bwd(const u3d_raster_desc& d, const U3DBuffers& b, const float* bg, const float* dL_dcolor,
                           const float* dL_dinvdepth, const float* out_color, const U3DLoss& loss, double* acc,
                           float* part, hipStream_t s) {
  const int tiles_x = (d.image_width + U3D_TILE - 1) / U3D_TILE, tiles_y = (d.image_height + U3D_TILE - 1) / U3D_TILE;
  const int T = tiles_x * tiles_y;
  const uint32_t ntiles = (uint32_t)(d.n_items * d.views_per_item * T);
  const size_t NG = (size_t)d.views_per_item * u3d_total_P(d);
  if (ntiles == 0 || NG == 0) return;
  const TileGrid tg = tile_grid(d, tiles_x, T);
  const bool invd = dL_dinvdepth && loss.kind == 0;
  uint32_t* tw = u3d_uses_touched_words(d) ? b.touched_words : nullptr;
#define LAUNCH(INVD, PBV)                                                                                                   \
  hipLaunchKernelGGL((render_bwd_wave_kernel<INVD, PBV>), tg.grid, dim3(TILE_WAVES * U3D_WAVE), 0, s, u3d_span(d), d.image_height, \
                     d.image_width, tiles_x, T, ntiles, tg.magic, NG, b.sorted_id, u3d_rect_indirect(d) ? b.rect : b.sorted_rect, u3d_rect_indirect(d), b.xy, b.conic_op, b.rgbd, bg,    \
                     dL_dcolor, dL_dinvdepth, b.final_T, b.n_contrib, b.tile_last, acc, part, out_color, b.clamped, tw, loss)
  if (u3d_part_blocks(d) == 1) { if (invd) LAUNCH(true, 1); else LAUNCH(false, 1); }
  else { if (invd) LAUNCH(true, U3D_PART_BLOCKS); else LAUNCH(false, U3D_PART_BLOCKS); }
#undef LAUNCH
  const int nsplit = bwd_reduce_split(T, d.n_items * d.views_per_item);
  if (u3d_part_blocks(d) == 1)
    hipLaunchKernelGGL(bwd_reduce1_kernel, dim3(d.n_items * d.views_per_item, nsplit), dim3(REDUCE_THREADS), 0, s,
                       u3d_span(d), T, invd ? U3D_NACC : U3D_NACC - 1, nsplit, NG, 0.5f * (float)d.image_width, 0.5f * (float)d.image_height, b.sorted_id,
                       b.conic_op, part, reinterpret_cast<const uint32_t*>(part + (size_t)ntiles * BWD_PART_STRIDE), acc, b.clamped, tw, 0, nullptr, 0.f,
                       nullptr);
  else
    hipLaunchKernelGGL(bwd_reduce_kernel<U3D_PART_BLOCKS>, dim3(d.n_items * d.views_per_item, nsplit), dim3(REDUCE_THREADS), 0, s,
                       u3d_span(d), T, invd ? U3D_NACC : U3D_NACC - 1, nsplit, NG, 0.5f * (float)d.image_width, 0.5f * (float)d.image_height, b.sorted_id,
                       b.conic_op, part, reinterpret_cast<const uint32_t*>(part + (size_t)ntiles * BWD_PART_STRIDE), acc, b.clamped, tw, 0, nullptr, 0.f,
                       nullptr, (uint2*)nullptr, (uint32_t*)nullptr, (float*)nullptr, (size_t)0);
}
