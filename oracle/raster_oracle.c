/*
 * oracle/raster_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the differentiable Gaussian-splat rasterizer that UniPre3D
 * calls through `diff_gaussian_rasterization`
 * (reference call site: gaussian_renderer/__init__.py:8,45-61,89-97).
 *
 * PARITY UNPINNED: the arithmetic lives in a third-party CUDA extension
 * (graphdeco-inria/diff-gaussian-rasterization, installed from an unpinned
 * `git clone --recursive` of gaussian-splatting@main, docs/INSTALLATION.md:51-61)
 * that is NOT in /root/reference and not installed in the build container.  The
 * reference ships no golden image / known-answer vector for it.  This file
 * restates the published algorithm ("3D Gaussian Splatting for Real-Time Radiance
 * Field Rendering", Kerbl et al. 2023, plus the anti-aliasing / inverse-depth
 * revision the reference's call site requires: 13th settings field
 * `antialiasing`, 3-tuple return) as listed in SURVEY.md section 8a rows R4-R6.
 * What IS pinned against reference code (tests/golden, tests/test_oracle_golden.py):
 *   - SH basis constants and degree-1..3 polynomials   (utils/sh_utils.py:26-27,57-116)
 *   - row-vector matrix convention / +1e-7 homogeneous divide
 *                                                      (utils/graphics_utils.py:22-30,38-84)
 *   - quaternion (r,x,y,z) -> rotation matrix formula  (utils/general_utils.py:171-194)
 *   - Sigma = (R S)(R S)^T                              (utils/general_utils.py:197-206)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product path (unipre3d_amd/) never does.
 *
 * The pipeline deliberately follows the ORIGINAL structure (per-Gaussian
 * preprocess -> prefix sum of tiles touched -> duplicate with (tile,depth) keys
 * -> stable sort -> per-tile ranges -> per-tile front-to-back blend), which is
 * different from the GPU implementation (one depth sort per view + per-tile
 * stream compaction); agreement of the two is therefore a real check.
 *
 * Compiled twice: REAL=float (prefix orf_) mirrors fp32 arithmetic,
 * REAL=double (prefix ord_) is the high-precision arbiter.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef ORACLE_DOUBLE
typedef float REAL;
#define FN(name) orf_##name
#define R_EXP expf
#define R_SQRT sqrtf
#define R_CEIL ceilf
#else
typedef double REAL;
#define FN(name) ord_##name
#define R_EXP exp
#define R_SQRT sqrt
#define R_CEIL ceil
#endif
#define C(x) ((REAL)(x))
#define RMIN(a, b) ((a) < (b) ? (a) : (b))
#define RMAX(a, b) ((a) > (b) ? (a) : (b))

#define TILE 16

/* flags shared with include/unipre3d_rasterizer.h */
#define FLAG_PREFILTERED 1
#define FLAG_ANTIALIASING 2
#define FLAG_DEBUG 4
#define FLAG_EXACT_AA_GRAD 8

/* Real spherical-harmonic basis constants; same values as utils/sh_utils.py:26-43 */
static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                0.3731763325901154,  -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435};

typedef struct {
  uint32_t tile;
  REAL depth;
  uint32_t idx;
} inst_t;

typedef struct oracle_state {
  int P, D, M, W, H, tiles_x, tiles_y, flags;
  REAL tan_fovx, tan_fovy, scale_modifier;
  REAL view[16], proj[16], campos[3], bg[3];
  /* borrowed input pointers are NOT kept: inputs are copied so that backward is self-contained */
  REAL *means3D, *shs, *colors_precomp, *opacities, *scales, *rotations, *cov3D_precomp;
  /* per-Gaussian forward state */
  REAL *depths, *means2D, *cov3D, *conic_opacity, *rgb;
  int *radii, *clamped;
  uint32_t *tiles_touched;
  /* binning */
  int64_t num_rendered;
  uint32_t *point_list;  /* [R] Gaussian ids, sorted by (tile, depth), stable */
  uint32_t *ranges;      /* [tiles*2] */
  /* per-pixel */
  REAL *final_T;
  uint32_t *n_contrib;
} oracle_state;

static REAL *dup_arr(const REAL *src, size_t n) {
  if (!src || n == 0) return NULL;
  REAL *d = (REAL *)malloc(n * sizeof(REAL));
  memcpy(d, src, n * sizeof(REAL));
  return d;
}

/* p (row vector, w=1) times a row-major 4x4: the convention of
 * utils/graphics_utils.py:22-30 (points_hom @ transf_matrix). */
static void xform4x4(const REAL *p, const REAL *m, REAL *out4) {
  for (int j = 0; j < 4; ++j) out4[j] = m[0 + j] * p[0] + m[4 + j] * p[1] + m[8 + j] * p[2] + m[12 + j];
}
static void xform4x3(const REAL *p, const REAL *m, REAL *out3) {
  for (int j = 0; j < 3; ++j) out3[j] = m[0 + j] * p[0] + m[4 + j] * p[1] + m[8 + j] * p[2] + m[12 + j];
}

/* Rotation matrix of an UN-normalised quaternion (r,x,y,z): same polynomial as
 * utils/general_utils.py:185-193 but without the division by |q| at :172-176
 * (the rasterizer must not normalise). Row-major R[i*3+j]. */
static void quat_to_R(const REAL *q, REAL *R) {
  REAL r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = C(1) - C(2) * (y * y + z * z);
  R[1] = C(2) * (x * y - r * z);
  R[2] = C(2) * (x * z + r * y);
  R[3] = C(2) * (x * y + r * z);
  R[4] = C(1) - C(2) * (x * x + z * z);
  R[5] = C(2) * (y * z - r * x);
  R[6] = C(2) * (x * z - r * y);
  R[7] = C(2) * (y * z + r * x);
  R[8] = C(1) - C(2) * (x * x + y * y);
}

/* Sigma = (R S)(R S)^T, utils/general_utils.py:197-206; upper triangle out. */
static void compute_cov3D(const REAL *scale, REAL mod, const REAL *rot, REAL *cov6) {
  REAL R[9], Mx[9];
  quat_to_R(rot, R);
  for (int i = 0; i < 3; ++i)
    for (int k = 0; k < 3; ++k) Mx[i * 3 + k] = R[i * 3 + k] * (mod * scale[k]);
  REAL S[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      REAL a = 0;
      for (int k = 0; k < 3; ++k) a += Mx[i * 3 + k] * Mx[j * 3 + k];
      S[i * 3 + j] = a;
    }
  cov6[0] = S[0]; cov6[1] = S[1]; cov6[2] = S[2];
  cov6[3] = S[4]; cov6[4] = S[5]; cov6[5] = S[8];
}

typedef struct {
  REAL t[3];        /* view-space point with clamped x,y */
  REAL xmask, ymask;
  REAL J[6];        /* 2x3 */
  REAL Wm[9];       /* Wm[j*3+i] = view[i*4+j]   (t = Wm p + trans) */
  REAL M2[6];       /* J * Wm, 2x3 */
} proj_lin_t;

static void ewa_setup(const REAL *mean, REAL fx, REAL fy, REAL tan_fovx, REAL tan_fovy,
                      const REAL *view, proj_lin_t *L) {
  xform4x3(mean, view, L->t);
  REAL limx = C(1.3) * tan_fovx, limy = C(1.3) * tan_fovy;
  REAL txtz = L->t[0] / L->t[2], tytz = L->t[1] / L->t[2];
  L->xmask = (txtz < -limx || txtz > limx) ? C(0) : C(1);
  L->ymask = (tytz < -limy || tytz > limy) ? C(0) : C(1);
  L->t[0] = RMIN(limx, RMAX(-limx, txtz)) * L->t[2];
  L->t[1] = RMIN(limy, RMAX(-limy, tytz)) * L->t[2];
  REAL tz = L->t[2];
  L->J[0] = fx / tz; L->J[1] = 0; L->J[2] = -(fx * L->t[0]) / (tz * tz);
  L->J[3] = 0; L->J[4] = fy / tz; L->J[5] = -(fy * L->t[1]) / (tz * tz);
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) L->Wm[j * 3 + i] = view[i * 4 + j];
  for (int k = 0; k < 2; ++k)
    for (int i = 0; i < 3; ++i) {
      REAL a = 0;
      for (int j = 0; j < 3; ++j) a += L->J[k * 3 + j] * L->Wm[j * 3 + i];
      L->M2[k * 3 + i] = a;
    }
}

static void cov6_to_sym(const REAL *c, REAL *S) {
  S[0] = c[0]; S[1] = c[1]; S[2] = c[2];
  S[3] = c[1]; S[4] = c[3]; S[5] = c[4];
  S[6] = c[2]; S[7] = c[4]; S[8] = c[5];
}

/* EWA splat covariance  cov2D = (J W) Sigma (J W)^T ; returns (a,b,c). */
static void compute_cov2D(const proj_lin_t *L, const REAL *cov6, REAL *abc) {
  REAL S[9], MS[6];
  cov6_to_sym(cov6, S);
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 3; ++j) {
      REAL a = 0;
      for (int i = 0; i < 3; ++i) a += L->M2[k * 3 + i] * S[i * 3 + j];
      MS[k * 3 + j] = a;
    }
  abc[0] = MS[0] * L->M2[0] + MS[1] * L->M2[1] + MS[2] * L->M2[2];
  abc[1] = MS[0] * L->M2[3] + MS[1] * L->M2[4] + MS[2] * L->M2[5];
  abc[2] = MS[3] * L->M2[3] + MS[4] * L->M2[4] + MS[5] * L->M2[5];
}

static void sh_basis_dir(const REAL *mean, const REAL *campos, REAL *dir, REAL *dir_orig) {
  for (int i = 0; i < 3; ++i) dir_orig[i] = mean[i] - campos[i];
  REAL len = R_SQRT(dir_orig[0] * dir_orig[0] + dir_orig[1] * dir_orig[1] + dir_orig[2] * dir_orig[2]);
  for (int i = 0; i < 3; ++i) dir[i] = dir_orig[i] / len;
}

/* Colour from SH coefficients laid out (P, M, 3); polynomials as utils/sh_utils.py:74-116 */
static void color_from_sh(int deg, int M, const REAL *mean, const REAL *campos, const REAL *sh /*[M][3]*/,
                          int *clamped3, REAL *rgb) {
  REAL dir[3], dorig[3];
  sh_basis_dir(mean, campos, dir, dorig);
  REAL x = dir[0], y = dir[1], z = dir[2];
  (void)M;
  for (int c = 0; c < 3; ++c) {
    REAL res = C(SH_C0) * sh[0 * 3 + c];
    if (deg > 0) {
      res = res - C(SH_C1) * y * sh[1 * 3 + c] + C(SH_C1) * z * sh[2 * 3 + c] - C(SH_C1) * x * sh[3 * 3 + c];
      if (deg > 1) {
        REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        res = res + C(SH_C2[0]) * xy * sh[4 * 3 + c] + C(SH_C2[1]) * yz * sh[5 * 3 + c] +
              C(SH_C2[2]) * (C(2) * zz - xx - yy) * sh[6 * 3 + c] + C(SH_C2[3]) * xz * sh[7 * 3 + c] +
              C(SH_C2[4]) * (xx - yy) * sh[8 * 3 + c];
        if (deg > 2) {
          res = res + C(SH_C3[0]) * y * (C(3) * xx - yy) * sh[9 * 3 + c] + C(SH_C3[1]) * xy * z * sh[10 * 3 + c] +
                C(SH_C3[2]) * y * (C(4) * zz - xx - yy) * sh[11 * 3 + c] +
                C(SH_C3[3]) * z * (C(2) * zz - C(3) * xx - C(3) * yy) * sh[12 * 3 + c] +
                C(SH_C3[4]) * x * (C(4) * zz - xx - yy) * sh[13 * 3 + c] +
                C(SH_C3[5]) * z * (xx - yy) * sh[14 * 3 + c] + C(SH_C3[6]) * x * (xx - C(3) * yy) * sh[15 * 3 + c];
        }
      }
    }
    res += C(0.5);
    clamped3[c] = (res < 0);
    rgb[c] = RMAX(res, C(0));
  }
}

static REAL ndc2pix(REAL v, int S) { return ((v + C(1)) * (REAL)S - C(1)) * C(0.5); }

static void get_rect(const REAL *p, int max_radius, int gx, int gy, int *rmin, int *rmax) {
  int v;
  v = (int)((p[0] - (REAL)max_radius) / (REAL)TILE); rmin[0] = v < 0 ? 0 : (v > gx ? gx : v);
  v = (int)((p[1] - (REAL)max_radius) / (REAL)TILE); rmin[1] = v < 0 ? 0 : (v > gy ? gy : v);
  v = (int)((p[0] + (REAL)max_radius + (REAL)(TILE - 1)) / (REAL)TILE); rmax[0] = v < 0 ? 0 : (v > gx ? gx : v);
  v = (int)((p[1] + (REAL)max_radius + (REAL)(TILE - 1)) / (REAL)TILE); rmax[1] = v < 0 ? 0 : (v > gy ? gy : v);
}

/* stable merge sort on (tile, depth); ties keep emission (= Gaussian index) order */
static int inst_le(const inst_t *a, const inst_t *b) {
  if (a->tile != b->tile) return a->tile < b->tile;
  return a->depth <= b->depth;
}
static void merge_sort(inst_t *a, inst_t *tmp, int64_t n) {
  for (int64_t w = 1; w < n; w *= 2) {
#pragma omp parallel for schedule(static)
    for (int64_t lo = 0; lo < n; lo += 2 * w) {
      int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      int64_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) tmp[k++] = inst_le(&a[i], &a[j]) ? a[i++] : a[j++];
      while (i < mid) tmp[k++] = a[i++];
      while (j < hi) tmp[k++] = a[j++];
    }
    memcpy(a, tmp, (size_t)n * sizeof(inst_t));
  }
}

void FN(free)(oracle_state *s) {
  if (!s) return;
  free(s->means3D); free(s->shs); free(s->colors_precomp); free(s->opacities); free(s->scales);
  free(s->rotations); free(s->cov3D_precomp);
  free(s->depths); free(s->means2D); free(s->cov3D); free(s->conic_opacity); free(s->rgb);
  free(s->radii); free(s->clamped); free(s->tiles_touched); free(s->point_list); free(s->ranges);
  free(s->final_T); free(s->n_contrib);
  free(s);
}

/* ---- FORWARD ------------------------------------------------------------------------------
 * Mirrors the operator called at gaussian_renderer/__init__.py:89-97.
 * shs: (P,M,3) or NULL; colors_precomp: (P,3) or NULL; scales (P,3)+rotations (P,4) or
 * cov3D_precomp (P,6).  out_color (3,H,W), out_invdepth (H,W), radii (P).
 * Returns an opaque state for backward / inspection (free with *_free). */
/* disc_*: the DISCRETE per-Gaussian decisions of another evaluation (NULL: decide here).  The reference operator is fp32: which
 * Gaussians survive the culls, their integer radius and tile rectangle, and the ORDER its (tile | fp32 depth bits) key with
 * stable index ties puts them in are integer facts of fp32 arithmetic.  An fp64 evaluation that re-decides them in fp64 is an
 * arbiter of a DIFFERENT discrete problem whenever two fp32 depths tie (routine at 10^5 Gaussians per view) or a radius sits on a
 * ceil boundary.  With the overrides the fp64 build keeps every continuous quantity in fp64 and takes visibility / radius /
 * rectangle / sort key from the fp32 restatement (oracle.forward(..., discrete_from=r32)): the arbiter the parity rule uses. */
static oracle_state *forward_impl(int P, int D, int M, const REAL *bg, int W, int H, const REAL *means3D, const REAL *shs,
                                  const REAL *colors_precomp, const REAL *opacities, const REAL *scales,
                                  REAL scale_modifier, const REAL *rotations, const REAL *cov3D_precomp,
                                  const REAL *viewmatrix, const REAL *projmatrix, const REAL *campos, REAL tan_fovx,
                                  REAL tan_fovy, int flags, REAL *out_color, REAL *out_invdepth, int *radii_out,
                                  const int *disc_radii, const int *disc_rect, const float *disc_depth) {
  oracle_state *s = (oracle_state *)calloc(1, sizeof(oracle_state));
  s->P = P; s->D = D; s->M = M; s->W = W; s->H = H; s->flags = flags;
  s->tiles_x = (W + TILE - 1) / TILE; s->tiles_y = (H + TILE - 1) / TILE;
  s->tan_fovx = tan_fovx; s->tan_fovy = tan_fovy; s->scale_modifier = scale_modifier;
  memcpy(s->view, viewmatrix, 16 * sizeof(REAL));
  memcpy(s->proj, projmatrix, 16 * sizeof(REAL));
  memcpy(s->campos, campos, 3 * sizeof(REAL));
  memcpy(s->bg, bg, 3 * sizeof(REAL));
  s->means3D = dup_arr(means3D, (size_t)P * 3);
  s->shs = dup_arr(shs, (size_t)P * M * 3);
  s->colors_precomp = dup_arr(colors_precomp, (size_t)P * 3);
  s->opacities = dup_arr(opacities, (size_t)P);
  s->scales = dup_arr(scales, (size_t)P * 3);
  s->rotations = dup_arr(rotations, (size_t)P * 4);
  s->cov3D_precomp = dup_arr(cov3D_precomp, (size_t)P * 6);
  size_t Pn = P > 0 ? (size_t)P : 1;
  s->depths = (REAL *)calloc(Pn, sizeof(REAL));
  s->means2D = (REAL *)calloc(Pn * 2, sizeof(REAL));
  s->cov3D = (REAL *)calloc(Pn * 6, sizeof(REAL));
  s->conic_opacity = (REAL *)calloc(Pn * 4, sizeof(REAL));
  s->rgb = (REAL *)calloc(Pn * 3, sizeof(REAL));
  s->radii = (int *)calloc(Pn, sizeof(int));
  s->clamped = (int *)calloc(Pn * 3, sizeof(int));
  s->tiles_touched = (uint32_t *)calloc(Pn, sizeof(uint32_t));
  const int antialiasing = (flags & FLAG_ANTIALIASING) != 0;
  const REAL focal_y = (REAL)H / (C(2) * tan_fovy), focal_x = (REAL)W / (C(2) * tan_fovx);
  const int gx = s->tiles_x, gy = s->tiles_y;

  /* (1) per-Gaussian preprocess */
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; ++idx) {
    const REAL *p = means3D + 3 * idx;
    REAL p_view[3], p_hom[4];
    xform4x3(p, viewmatrix, p_view);
    if (disc_radii) { if (disc_radii[idx] <= 0) continue; }   /* every cull below was decided by the fp32 evaluation */
    else if (p_view[2] <= C(0.2)) continue;                  /* near cull (R4 step 1) */
    xform4x4(p, projmatrix, p_hom);
    REAL p_w = C(1) / (p_hom[3] + C(0.0000001));             /* R4 step 2 */
    REAL p_proj[2] = {p_hom[0] * p_w, p_hom[1] * p_w};
    REAL *cov6 = s->cov3D + 6 * idx;
    if (cov3D_precomp) memcpy(cov6, cov3D_precomp + 6 * idx, 6 * sizeof(REAL));
    else compute_cov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, cov6);
    proj_lin_t L;
    ewa_setup(p, focal_x, focal_y, tan_fovx, tan_fovy, viewmatrix, &L);
    REAL abc[3];
    compute_cov2D(&L, cov6, abc);
    const REAL h_var = C(0.3);                                /* R4 step 5 */
    REAL det_cov = abc[0] * abc[2] - abc[1] * abc[1];
    abc[0] += h_var; abc[2] += h_var;
    REAL det_plus = abc[0] * abc[2] - abc[1] * abc[1];
    REAL h_scaling = C(1);
    if (antialiasing) h_scaling = R_SQRT(RMAX(C(0.000025), det_cov / det_plus));
    REAL det = det_plus;
    if (det == C(0) && !disc_radii) continue;
    REAL det_inv = C(1) / det;
    REAL conic[3] = {abc[2] * det_inv, -abc[1] * det_inv, abc[0] * det_inv};
    REAL mid = C(0.5) * (abc[0] + abc[2]);                    /* R4 step 6 */
    REAL lambda1 = mid + R_SQRT(RMAX(C(0.1), mid * mid - det));
    REAL lambda2 = mid - R_SQRT(RMAX(C(0.1), mid * mid - det));
    REAL my_radius = R_CEIL(C(3) * R_SQRT(RMAX(lambda1, lambda2)));
    REAL point_image[2] = {ndc2pix(p_proj[0], W), ndc2pix(p_proj[1], H)};
    int rmin[2], rmax[2];
    if (disc_radii) {
      my_radius = (REAL)disc_radii[idx];
      rmin[0] = disc_rect[4 * idx]; rmin[1] = disc_rect[4 * idx + 1]; rmax[0] = disc_rect[4 * idx + 2]; rmax[1] = disc_rect[4 * idx + 3];
    } else {
      get_rect(point_image, (int)my_radius, gx, gy, rmin, rmax);
      if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
    }
    if (colors_precomp) {
      for (int c = 0; c < 3; ++c) s->rgb[3 * idx + c] = colors_precomp[3 * idx + c];
    } else {
      color_from_sh(D, M, p, campos, shs + (size_t)idx * M * 3, s->clamped + 3 * idx, s->rgb + 3 * idx);
    }
    s->depths[idx] = p_view[2];
    s->radii[idx] = (int)my_radius;
    s->means2D[2 * idx] = point_image[0]; s->means2D[2 * idx + 1] = point_image[1];
    s->conic_opacity[4 * idx + 0] = conic[0]; s->conic_opacity[4 * idx + 1] = conic[1];
    s->conic_opacity[4 * idx + 2] = conic[2]; s->conic_opacity[4 * idx + 3] = opacities[idx] * h_scaling;
    s->tiles_touched[idx] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
  }
  if (radii_out) memcpy(radii_out, s->radii, (size_t)P * sizeof(int));

  /* (2) inclusive prefix sum -> offsets, (3) duplicate with keys */
  int64_t *offsets = (int64_t *)malloc(Pn * sizeof(int64_t));
  int64_t run = 0;
  for (int i = 0; i < P; ++i) { run += s->tiles_touched[i]; offsets[i] = run; }
  int64_t Rn = run;
  s->num_rendered = Rn;
  inst_t *inst = (inst_t *)malloc((size_t)(Rn > 0 ? Rn : 1) * sizeof(inst_t));
  inst_t *tmp = (inst_t *)malloc((size_t)(Rn > 0 ? Rn : 1) * sizeof(inst_t));
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; ++idx) {
    if (s->radii[idx] <= 0) continue;
    int64_t off = idx == 0 ? 0 : offsets[idx - 1];
    int rmin[2], rmax[2];
    if (disc_radii) {
      rmin[0] = disc_rect[4 * idx]; rmin[1] = disc_rect[4 * idx + 1]; rmax[0] = disc_rect[4 * idx + 2]; rmax[1] = disc_rect[4 * idx + 3];
    } else {
      get_rect(s->means2D + 2 * idx, s->radii[idx], gx, gy, rmin, rmax);
    }
    for (int y = rmin[1]; y < rmax[1]; ++y)
      for (int x = rmin[0]; x < rmax[0]; ++x) {
        inst[off].tile = (uint32_t)(y * gx + x);
        inst[off].depth = disc_depth ? (REAL)disc_depth[idx] : s->depths[idx];   /* sort key: the fp32 depth, ties -> index (R4 step 8) */
        inst[off].idx = (uint32_t)idx;
        ++off;
      }
  }
  /* (4) stable sort by (tile, depth) (R4 step 8) */
  merge_sort(inst, tmp, Rn);
  free(tmp);
  s->point_list = (uint32_t *)malloc((size_t)(Rn > 0 ? Rn : 1) * sizeof(uint32_t));
  for (int64_t i = 0; i < Rn; ++i) s->point_list[i] = inst[i].idx;
  /* (5) tile ranges */
  int ntiles = gx * gy;
  s->ranges = (uint32_t *)calloc((size_t)ntiles * 2, sizeof(uint32_t));
  for (int64_t i = 0; i < Rn; ++i) {
    uint32_t t = inst[i].tile;
    if (i == 0 || inst[i - 1].tile != t) s->ranges[2 * t] = (uint32_t)i;
    if (i == Rn - 1 || inst[i + 1].tile != t) s->ranges[2 * t + 1] = (uint32_t)(i + 1);
  }
  free(inst);
  free(offsets);

  /* (6) per-tile front-to-back blend (R4 steps 9-10) */
  s->final_T = (REAL *)calloc((size_t)W * H, sizeof(REAL));
  s->n_contrib = (uint32_t *)calloc((size_t)W * H, sizeof(uint32_t));
#pragma omp parallel for schedule(dynamic, 1)
  for (int tile = 0; tile < ntiles; ++tile) {
    int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
    uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
    for (int py = ty0; py < ty0 + TILE && py < H; ++py)
      for (int px = tx0; px < tx0 + TILE && px < W; ++px) {
        REAL T = C(1), Cc[3] = {0, 0, 0}, inv_d = 0;
        uint32_t contributor = 0, last = 0;
        for (uint32_t k = r0; k < r1; ++k) {
          ++contributor;
          uint32_t g = s->point_list[k];
          REAL dx = s->means2D[2 * g] - (REAL)px, dy = s->means2D[2 * g + 1] - (REAL)py;
          const REAL *co = s->conic_opacity + 4 * g;
          REAL power = C(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > C(0)) continue;
          REAL alpha = RMIN(C(0.99), co[3] * R_EXP(power));
          if (alpha < C(1.0) / C(255.0)) continue;
          REAL test_T = T * (C(1) - alpha);
          if (test_T < C(0.0001)) break;
          for (int c = 0; c < 3; ++c) Cc[c] += s->rgb[3 * g + c] * alpha * T;
          inv_d += (C(1) / s->depths[g]) * alpha * T;
          T = test_T;
          last = contributor;
        }
        size_t pid = (size_t)py * W + px;
        s->final_T[pid] = T;
        s->n_contrib[pid] = last;
        for (int c = 0; c < 3; ++c) out_color[(size_t)c * H * W + pid] = Cc[c] + T * bg[c];
        if (out_invdepth) out_invdepth[pid] = inv_d;
      }
  }
  return s;
}

/* accessors for per-stage parity tests */
oracle_state *FN(forward)(int P, int D, int M, const REAL *bg, int W, int H, const REAL *means3D, const REAL *shs,
                          const REAL *colors_precomp, const REAL *opacities, const REAL *scales,
                          REAL scale_modifier, const REAL *rotations, const REAL *cov3D_precomp,
                          const REAL *viewmatrix, const REAL *projmatrix, const REAL *campos, REAL tan_fovx,
                          REAL tan_fovy, int flags, REAL *out_color, REAL *out_invdepth, int *radii_out) {
  return forward_impl(P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                      viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, flags, out_color, out_invdepth, radii_out, NULL, NULL, NULL);
}

/* The same forward under ANOTHER evaluation's discrete decisions (see forward_impl): radii [P], rectangles [P][4] =
 * (xmin, ymin, xmax, ymax) in tiles, fp32 depths [P] (the sort key). */
oracle_state *FN(forward_discrete)(int P, int D, int M, const REAL *bg, int W, int H, const REAL *means3D, const REAL *shs,
                                   const REAL *colors_precomp, const REAL *opacities, const REAL *scales,
                                   REAL scale_modifier, const REAL *rotations, const REAL *cov3D_precomp,
                                   const REAL *viewmatrix, const REAL *projmatrix, const REAL *campos, REAL tan_fovx,
                                   REAL tan_fovy, int flags, REAL *out_color, REAL *out_invdepth, int *radii_out,
                                   const int *disc_radii, const int *disc_rect, const float *disc_depth) {
  return forward_impl(P, D, M, bg, W, H, means3D, shs, colors_precomp, opacities, scales, scale_modifier, rotations, cov3D_precomp,
                      viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, flags, out_color, out_invdepth, radii_out, disc_radii,
                      disc_rect, disc_depth);
}

/* The discrete decisions of THIS evaluation, in the form forward_discrete takes them. */
void FN(state_discrete)(const oracle_state *s, int *rect_out /*[P][4]*/, float *depth_out /*[P]*/) {
  for (int idx = 0; idx < s->P; ++idx) {
    int rmin[2] = {0, 0}, rmax[2] = {0, 0};
    if (s->radii[idx] > 0) get_rect(s->means2D + 2 * idx, s->radii[idx], s->tiles_x, s->tiles_y, rmin, rmax);
    rect_out[4 * idx] = rmin[0]; rect_out[4 * idx + 1] = rmin[1]; rect_out[4 * idx + 2] = rmax[0]; rect_out[4 * idx + 3] = rmax[1];
    depth_out[idx] = (float)s->depths[idx];
  }
}

int64_t FN(num_rendered)(const oracle_state *s) { return s->num_rendered; }
const REAL *FN(state_depths)(const oracle_state *s) { return s->depths; }
const REAL *FN(state_means2D)(const oracle_state *s) { return s->means2D; }
const REAL *FN(state_cov3D)(const oracle_state *s) { return s->cov3D; }
const REAL *FN(state_conic_opacity)(const oracle_state *s) { return s->conic_opacity; }
const REAL *FN(state_rgb)(const oracle_state *s) { return s->rgb; }
const uint32_t *FN(state_tiles_touched)(const oracle_state *s) { return s->tiles_touched; }
const REAL *FN(state_final_T)(const oracle_state *s) { return s->final_T; }
const uint32_t *FN(state_n_contrib)(const oracle_state *s) { return s->n_contrib; }
const uint32_t *FN(state_point_list)(const oracle_state *s) { return s->point_list; }
const uint32_t *FN(state_ranges)(const oracle_state *s) { return s->ranges; }

/* ---- BACKWARD -----------------------------------------------------------------------------
 * SURVEY R5/R6.  dL_dpix (3,H,W), dL_dinvdepth (H,W) or NULL.
 * Outputs (caller-zeroed not required): dL_dmeans3D (P,3), dL_dmeans2D (P,3), dL_dshs (P,M,3),
 * dL_dcolors (P,3), dL_dopacity (P), dL_dscales (P,3), dL_drots (P,4), dL_dcov3D (P,6).
 * Deliberate deviations from the true derivative (R6 i-v) are marked DEV. */
void FN(backward)(const oracle_state *s, const REAL *dL_dpix, const REAL *dL_dinvdepth_pix, REAL *dL_dmeans3D,
                  REAL *dL_dmeans2D, REAL *dL_dshs, REAL *dL_dcolors, REAL *dL_dopacity, REAL *dL_dscales,
                  REAL *dL_drots, REAL *dL_dcov3D) {
  const int P = s->P, W = s->W, H = s->H, M = s->M, D = s->D;
  const int gx = s->tiles_x, ntiles = s->tiles_x * s->tiles_y;
  size_t Pn = P > 0 ? (size_t)P : 1;
  memset(dL_dmeans3D, 0, (size_t)P * 3 * sizeof(REAL));
  memset(dL_dmeans2D, 0, (size_t)P * 3 * sizeof(REAL));
  if (dL_dshs && M > 0) memset(dL_dshs, 0, (size_t)P * M * 3 * sizeof(REAL));
  memset(dL_dcolors, 0, (size_t)P * 3 * sizeof(REAL));
  memset(dL_dopacity, 0, (size_t)P * sizeof(REAL));
  memset(dL_dscales, 0, (size_t)P * 3 * sizeof(REAL));
  memset(dL_drots, 0, (size_t)P * 4 * sizeof(REAL));
  memset(dL_dcov3D, 0, (size_t)P * 6 * sizeof(REAL));
  REAL *g_conic = (REAL *)calloc(Pn * 3, sizeof(REAL));   /* (dA, dB_half, dC) */
  REAL *g_invd = (REAL *)calloc(Pn, sizeof(REAL));

  /* (1) per-tile back-to-front pass.  Thread-private accumulators, reduced afterwards.  Tiles are dealt to the threads round-robin
   * (schedule(static, 1)): for a given thread count the fp32 summation grouping -- and with it the restatement's own distance from
   * the fp64 arbiter, which the parity tests use as their yardstick on ill-conditioned scenes -- is the same in every run. */
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  const int NG = 10; /* mean2D.xy, conic(3), opacity, rgb(3), invdepth */
  REAL *priv = (REAL *)calloc((size_t)nthreads * Pn * NG, sizeof(REAL));
  const REAL ddelx_dx = C(0.5) * (REAL)W, ddely_dy = C(0.5) * (REAL)H;
#pragma omp parallel for schedule(static, 1)
  for (int tile = 0; tile < ntiles; ++tile) {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    REAL *acc = priv + (size_t)tid * Pn * NG;
    int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
    uint32_t r0 = s->ranges[2 * tile], r1 = s->ranges[2 * tile + 1];
    for (int py = ty0; py < ty0 + TILE && py < H; ++py)
      for (int px = tx0; px < tx0 + TILE && px < W; ++px) {
        size_t pid = (size_t)py * W + px;
        const REAL T_final = s->final_T[pid];
        REAL T = T_final;
        uint32_t contributor = r1 - r0;
        const uint32_t last_contributor = s->n_contrib[pid];
        REAL accum_rec[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
        REAL accum_invd = 0, last_invd = 0;
        REAL dpix[3] = {dL_dpix[pid], dL_dpix[(size_t)H * W + pid], dL_dpix[(size_t)2 * H * W + pid]};
        REAL dinv = dL_dinvdepth_pix ? dL_dinvdepth_pix[pid] : C(0);
        REAL bg_dot_dpixel = s->bg[0] * dpix[0] + s->bg[1] * dpix[1] + s->bg[2] * dpix[2];
        for (uint32_t k = r1; k-- > r0;) {
          --contributor;
          if (contributor >= last_contributor) continue;
          uint32_t g = s->point_list[k];
          REAL dx = s->means2D[2 * g] - (REAL)px, dy = s->means2D[2 * g + 1] - (REAL)py;
          const REAL *co = s->conic_opacity + 4 * g;
          REAL power = C(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > C(0)) continue;
          REAL G = R_EXP(power);
          REAL alpha = RMIN(C(0.99), co[3] * G);
          if (alpha < C(1.0) / C(255.0)) continue;
          T = T / (C(1) - alpha);
          REAL dchannel_dcolor = alpha * T;
          REAL dL_dalpha = 0;
          REAL *a = acc + (size_t)g * NG;
          for (int c = 0; c < 3; ++c) {
            REAL col = s->rgb[3 * g + c];
            accum_rec[c] = last_alpha * last_color[c] + (C(1) - last_alpha) * accum_rec[c];
            last_color[c] = col;
            dL_dalpha += (col - accum_rec[c]) * dpix[c];
            a[6 + c] += dchannel_dcolor * dpix[c];
          }
          REAL invd = C(1) / s->depths[g];
          accum_invd = last_alpha * last_invd + (C(1) - last_alpha) * accum_invd;
          last_invd = invd;
          dL_dalpha += (invd - accum_invd) * dinv;
          a[9] += dchannel_dcolor * dinv;
          dL_dalpha *= T;
          last_alpha = alpha;
          dL_dalpha += (-T_final / (C(1) - alpha)) * bg_dot_dpixel;
          /* DEV(i): no mask for the min(0.99, .) clamp */
          REAL dL_dG = co[3] * dL_dalpha;
          REAL gdx = G * dx, gdy = G * dy;
          REAL dG_ddelx = -gdx * co[0] - gdy * co[1];
          REAL dG_ddely = -gdy * co[2] - gdx * co[1];
          a[0] += dL_dG * dG_ddelx * ddelx_dx;
          a[1] += dL_dG * dG_ddely * ddely_dy;
          a[2] += C(-0.5) * gdx * dx * dL_dG;
          a[3] += C(-0.5) * gdx * dy * dL_dG;
          a[4] += C(-0.5) * gdy * dy * dL_dG;
          a[5] += G * dL_dalpha;
        }
      }
  }
  for (int t = 0; t < nthreads; ++t) {
    const REAL *acc = priv + (size_t)t * Pn * NG;
    for (int g = 0; g < P; ++g) {
      const REAL *a = acc + (size_t)g * NG;
      dL_dmeans2D[3 * g] += a[0]; dL_dmeans2D[3 * g + 1] += a[1];
      g_conic[3 * g] += a[2]; g_conic[3 * g + 1] += a[3]; g_conic[3 * g + 2] += a[4];
      dL_dopacity[g] += a[5];
      dL_dcolors[3 * g] += a[6]; dL_dcolors[3 * g + 1] += a[7]; dL_dcolors[3 * g + 2] += a[8];
      g_invd[g] += a[9];
    }
  }
  free(priv);

  const int antialiasing = (s->flags & FLAG_ANTIALIASING) != 0;
  const int exact_aa = (s->flags & FLAG_EXACT_AA_GRAD) != 0;
  const REAL focal_y = (REAL)H / (C(2) * s->tan_fovy), focal_x = (REAL)W / (C(2) * s->tan_fovx);

  /* (2) per-Gaussian: conic -> cov2D -> (cov3D, mean3D); mean2D -> mean3D; SH; cov3D -> scale/rot */
#pragma omp parallel for schedule(static)
  for (int idx = 0; idx < P; ++idx) {
    if (!(s->radii[idx] > 0)) {
      /* Gaussians that never reached the image get zero gradient everywhere */
      dL_dopacity[idx] = 0;
      continue;
    }
    const REAL *mean = s->means3D + 3 * idx;
    const REAL *cov6 = s->cov3D + 6 * idx;
    proj_lin_t L;
    ewa_setup(mean, focal_x, focal_y, s->tan_fovx, s->tan_fovy, s->view, &L);
    REAL abc[3];
    compute_cov2D(&L, cov6, abc);
    REAL c_xx = abc[0], c_xy = abc[1], c_yy = abc[2];
    const REAL h_var = C(0.3);
    REAL d_inside_root = 0;
    REAL x0 = c_xx, y0 = c_yy; /* pre-filter values */
    if (antialiasing) {
      REAL det_cov = c_xx * c_yy - c_xy * c_xy;
      c_xx += h_var; c_yy += h_var;
      REAL det_plus = c_xx * c_yy - c_xy * c_xy;
      REAL h_scaling = R_SQRT(RMAX(C(0.000025), det_cov / det_plus));
      REAL dL_dop_v = dL_dopacity[idx];
      REAL d_h_scaling = dL_dop_v * s->opacities[idx];
      dL_dopacity[idx] = dL_dop_v * h_scaling;                 /* R6(v) */
      d_inside_root = (det_cov / det_plus) <= C(0.000025) ? C(0) : d_h_scaling / (C(2) * h_scaling);
    } else {
      c_xx += h_var; c_yy += h_var;
    }
    REAL dL_dc_xx = 0, dL_dc_xy = 0, dL_dc_yy = 0;
    if (antialiasing) {
      /* d/d{x,y,z} of (x y - z^2)/((x+w)(y+w) - z^2).
       * DEV(vi) [UPSTREAM-RECALL]: the original evaluates these closed forms with x,y taken
       * AFTER the +w low-pass; FLAG_EXACT_AA_GRAD selects the mathematically exact variant
       * (x,y before the low-pass). */
      REAL x = exact_aa ? x0 : c_xx, y = exact_aa ? y0 : c_yy, z = c_xy, w = h_var;
      REAL dn = w * w + w * (x + y) + x * y - z * z;
      REAL denom_f = d_inside_root / (dn * dn);
      dL_dc_xx = w * (w * y + y * y + z * z) * denom_f;
      dL_dc_yy = w * (w * x + x * x + z * z) * denom_f;
      dL_dc_xy = C(-2) * w * z * (w + x + y) * denom_f;
    }
    REAL dA = g_conic[3 * idx], dBh = g_conic[3 * idx + 1], dCc = g_conic[3 * idx + 2];
    REAL denom = c_xx * c_yy - c_xy * c_xy;
    REAL denom2inv = C(1) / ((denom * denom) + C(0.0000001));
    REAL dL_dtv[3] = {0, 0, 0};
    if (denom2inv != 0) {
      dL_dc_xx += denom2inv * (-c_yy * c_yy * dA + C(2) * c_xy * c_yy * dBh + (denom - c_xx * c_yy) * dCc);
      dL_dc_yy += denom2inv * (-c_xx * c_xx * dCc + C(2) * c_xx * c_xy * dBh + (denom - c_xx * c_yy) * dA);
      dL_dc_xy += denom2inv * C(2) * (c_xy * c_yy * dA - (denom + C(2) * c_xy * c_xy) * dBh + c_xx * c_xy * dCc);
      /* cov2D = M2 Sigma M2^T with Gc = [[da, db/2],[db/2, dc]] */
      REAL Gc[4] = {dL_dc_xx, C(0.5) * dL_dc_xy, C(0.5) * dL_dc_xy, dL_dc_yy};
      REAL GM[6]; /* Gc * M2 (2x3) */
      for (int k = 0; k < 2; ++k)
        for (int i = 0; i < 3; ++i) GM[k * 3 + i] = Gc[k * 2] * L.M2[i] + Gc[k * 2 + 1] * L.M2[3 + i];
      REAL dS[9]; /* M2^T Gc M2 */
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) dS[i * 3 + j] = L.M2[i] * GM[j] + L.M2[3 + i] * GM[3 + j];
      REAL *dc = dL_dcov3D + 6 * idx;
      dc[0] = dS[0]; dc[3] = dS[4]; dc[5] = dS[8];
      dc[1] = C(2) * dS[1]; dc[2] = C(2) * dS[2]; dc[4] = C(2) * dS[5];
      /* dL/dM2 = 2 Gc M2 Sigma */
      REAL S[9], dM2[6];
      cov6_to_sym(cov6, S);
      for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 3; ++j) {
          REAL a = 0;
          for (int i = 0; i < 3; ++i) a += GM[k * 3 + i] * S[i * 3 + j];
          dM2[k * 3 + j] = C(2) * a;
        }
      /* dL/dJ = dL/dM2 * Wm^T */
      REAL dJ[6];
      for (int k = 0; k < 2; ++k)
        for (int l = 0; l < 3; ++l) {
          REAL a = 0;
          for (int m = 0; m < 3; ++m) a += dM2[k * 3 + m] * L.Wm[l * 3 + m];
          dJ[k * 3 + l] = a;
        }
      REAL tz = C(1) / L.t[2], tz2 = tz * tz, tz3 = tz2 * tz;
      /* DEV(ii): clamp masks only the direct t.x / t.y path */
      dL_dtv[0] = L.xmask * -focal_x * tz2 * dJ[2];
      dL_dtv[1] = L.ymask * -focal_y * tz2 * dJ[5];
      dL_dtv[2] = -focal_x * tz2 * dJ[0] - focal_y * tz2 * dJ[4] + (C(2) * focal_x * L.t[0]) * tz3 * dJ[2] +
                  (C(2) * focal_y * L.t[1]) * tz3 * dJ[5];
    }
    /* inverse-depth output: d(1/tz)/dtz */
    dL_dtv[2] -= g_invd[idx] / (L.t[2] * L.t[2]);
    REAL dmean[3];
    for (int i = 0; i < 3; ++i)
      dmean[i] = s->view[i * 4 + 0] * dL_dtv[0] + s->view[i * 4 + 1] * dL_dtv[1] + s->view[i * 4 + 2] * dL_dtv[2];

    /* screen-space mean -> 3D mean */
    REAL m_hom[4];
    xform4x4(mean, s->proj, m_hom);
    REAL m_w = C(1) / (m_hom[3] + C(0.0000001));
    REAL mul1 = m_hom[0] * m_w * m_w, mul2 = m_hom[1] * m_w * m_w;
    REAL d2x = dL_dmeans2D[3 * idx], d2y = dL_dmeans2D[3 * idx + 1];
    for (int i = 0; i < 3; ++i)
      dmean[i] += (s->proj[i * 4 + 0] * m_w - s->proj[i * 4 + 3] * mul1) * d2x +
                  (s->proj[i * 4 + 1] * m_w - s->proj[i * 4 + 3] * mul2) * d2y;

    /* SH */
    if (s->shs) {
      REAL dir[3], dorig[3];
      sh_basis_dir(mean, s->campos, dir, dorig);
      REAL x = dir[0], y = dir[1], z = dir[2];
      const REAL *sh = s->shs + (size_t)idx * M * 3;
      REAL *dsh = dL_dshs + (size_t)idx * M * 3;
      REAL dRGB[3];
      for (int c = 0; c < 3; ++c) dRGB[c] = s->clamped[3 * idx + c] ? C(0) : dL_dcolors[3 * idx + c]; /* R6(iii) */
      REAL ddir[3] = {0, 0, 0};
      for (int c = 0; c < 3; ++c) {
        REAL g = dRGB[c];
        dsh[0 * 3 + c] = C(SH_C0) * g;
        REAL dx_ = 0, dy_ = 0, dz_ = 0;
        if (D > 0) {
          dsh[1 * 3 + c] = -C(SH_C1) * y * g;
          dsh[2 * 3 + c] = C(SH_C1) * z * g;
          dsh[3 * 3 + c] = -C(SH_C1) * x * g;
          dx_ = -C(SH_C1) * sh[3 * 3 + c];
          dy_ = -C(SH_C1) * sh[1 * 3 + c];
          dz_ = C(SH_C1) * sh[2 * 3 + c];
          if (D > 1) {
            REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            dsh[4 * 3 + c] = C(SH_C2[0]) * xy * g;
            dsh[5 * 3 + c] = C(SH_C2[1]) * yz * g;
            dsh[6 * 3 + c] = C(SH_C2[2]) * (C(2) * zz - xx - yy) * g;
            dsh[7 * 3 + c] = C(SH_C2[3]) * xz * g;
            dsh[8 * 3 + c] = C(SH_C2[4]) * (xx - yy) * g;
            dx_ += C(SH_C2[0]) * y * sh[4 * 3 + c] + C(SH_C2[2]) * C(2) * -x * sh[6 * 3 + c] +
                   C(SH_C2[3]) * z * sh[7 * 3 + c] + C(SH_C2[4]) * C(2) * x * sh[8 * 3 + c];
            dy_ += C(SH_C2[0]) * x * sh[4 * 3 + c] + C(SH_C2[1]) * z * sh[5 * 3 + c] +
                   C(SH_C2[2]) * C(2) * -y * sh[6 * 3 + c] + C(SH_C2[4]) * C(2) * -y * sh[8 * 3 + c];
            dz_ += C(SH_C2[1]) * y * sh[5 * 3 + c] + C(SH_C2[2]) * C(4) * z * sh[6 * 3 + c] +
                   C(SH_C2[3]) * x * sh[7 * 3 + c];
            if (D > 2) {
              dsh[9 * 3 + c] = C(SH_C3[0]) * y * (C(3) * xx - yy) * g;
              dsh[10 * 3 + c] = C(SH_C3[1]) * xy * z * g;
              dsh[11 * 3 + c] = C(SH_C3[2]) * y * (C(4) * zz - xx - yy) * g;
              dsh[12 * 3 + c] = C(SH_C3[3]) * z * (C(2) * zz - C(3) * xx - C(3) * yy) * g;
              dsh[13 * 3 + c] = C(SH_C3[4]) * x * (C(4) * zz - xx - yy) * g;
              dsh[14 * 3 + c] = C(SH_C3[5]) * z * (xx - yy) * g;
              dsh[15 * 3 + c] = C(SH_C3[6]) * x * (xx - C(3) * yy) * g;
              dx_ += C(SH_C3[0]) * sh[9 * 3 + c] * C(6) * xy + C(SH_C3[1]) * sh[10 * 3 + c] * yz +
                     C(SH_C3[2]) * sh[11 * 3 + c] * -C(2) * xy + C(SH_C3[3]) * sh[12 * 3 + c] * -C(6) * xz +
                     C(SH_C3[4]) * sh[13 * 3 + c] * (C(4) * zz - C(3) * xx - yy) +
                     C(SH_C3[5]) * sh[14 * 3 + c] * C(2) * xz + C(SH_C3[6]) * sh[15 * 3 + c] * C(3) * (xx - yy);
              dy_ += C(SH_C3[0]) * sh[9 * 3 + c] * C(3) * (xx - yy) + C(SH_C3[1]) * sh[10 * 3 + c] * xz +
                     C(SH_C3[2]) * sh[11 * 3 + c] * (C(4) * zz - xx - C(3) * yy) +
                     C(SH_C3[3]) * sh[12 * 3 + c] * -C(6) * yz + C(SH_C3[4]) * sh[13 * 3 + c] * -C(2) * xy +
                     C(SH_C3[5]) * sh[14 * 3 + c] * -C(2) * yz + C(SH_C3[6]) * sh[15 * 3 + c] * -C(6) * xy;
              dz_ += C(SH_C3[1]) * sh[10 * 3 + c] * xy + C(SH_C3[2]) * sh[11 * 3 + c] * C(8) * yz +
                     C(SH_C3[3]) * sh[12 * 3 + c] * C(3) * (C(2) * zz - xx - yy) +
                     C(SH_C3[4]) * sh[13 * 3 + c] * C(8) * xz + C(SH_C3[5]) * sh[14 * 3 + c] * (xx - yy);
            }
          }
        }
        ddir[0] += dx_ * g; ddir[1] += dy_ * g; ddir[2] += dz_ * g;
      }
      /* derivative of v/|v| */
      REAL sum2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
      REAL invsum32 = C(1) / R_SQRT(sum2 * sum2 * sum2);
      dmean[0] += ((sum2 - dorig[0] * dorig[0]) * ddir[0] - dorig[1] * dorig[0] * ddir[1] - dorig[2] * dorig[0] * ddir[2]) * invsum32;
      dmean[1] += (-dorig[0] * dorig[1] * ddir[0] + (sum2 - dorig[1] * dorig[1]) * ddir[1] - dorig[2] * dorig[1] * ddir[2]) * invsum32;
      dmean[2] += (-dorig[0] * dorig[2] * ddir[0] - dorig[1] * dorig[2] * ddir[1] + (sum2 - dorig[2] * dorig[2]) * ddir[2]) * invsum32;
    }
    for (int i = 0; i < 3; ++i) dL_dmeans3D[3 * idx + i] = dmean[i];

    /* cov3D -> scale, rotation (exact derivative of the un-normalised polynomial) */
    if (s->scales) {
      const REAL *dc = dL_dcov3D + 6 * idx;
      REAL Gs[9] = {dc[0], C(0.5) * dc[1], C(0.5) * dc[2], C(0.5) * dc[1], dc[3], C(0.5) * dc[4],
                    C(0.5) * dc[2], C(0.5) * dc[4], dc[5]};
      REAL Rm[9], Mx[9], sv[3];
      const REAL *q = s->rotations + 4 * idx;
      quat_to_R(q, Rm);
      for (int k = 0; k < 3; ++k) sv[k] = s->scale_modifier * s->scales[3 * idx + k];
      for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) Mx[i * 3 + k] = Rm[i * 3 + k] * sv[k];
      REAL dM[9]; /* 2 Gs Mx */
      for (int i = 0; i < 3; ++i)
        for (int k = 0; k < 3; ++k) {
          REAL a = 0;
          for (int j = 0; j < 3; ++j) a += Gs[i * 3 + j] * Mx[j * 3 + k];
          dM[i * 3 + k] = C(2) * a;
        }
      REAL dR[9];
      for (int k = 0; k < 3; ++k) {
        REAL a = 0;
        for (int i = 0; i < 3; ++i) { a += dM[i * 3 + k] * Rm[i * 3 + k]; dR[i * 3 + k] = dM[i * 3 + k] * sv[k]; }
        dL_dscales[3 * idx + k] = s->scale_modifier * a;
      }
      REAL r = q[0], x = q[1], y = q[2], z = q[3];
      dL_drots[4 * idx + 0] = C(2) * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
      dL_drots[4 * idx + 1] = C(2) * (y * dR[1] + z * dR[2] + y * dR[3] - C(2) * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - C(2) * x * dR[8]);
      dL_drots[4 * idx + 2] = C(2) * (-C(2) * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - C(2) * y * dR[8]);
      dL_drots[4 * idx + 3] = C(2) * (-C(2) * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - C(2) * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    }
  }
  free(g_conic);
  free(g_invd);
}

/* visibility test only (operator `mark_visible`; no caller in the reference tree) */
void FN(mark_visible)(int P, const REAL *means3D, const REAL *viewmatrix, const REAL *projmatrix, uint8_t *present) {
  (void)projmatrix;
  for (int i = 0; i < P; ++i) {
    REAL pv[3];
    xform4x3(means3D + 3 * i, viewmatrix, pv);
    present[i] = pv[2] > C(0.2);
  }
}

/* OpenMP threads of the CALLING thread's next parallel regions (per-thread ICV): bench.py's cpu_baseline leg renders one view per
 * host thread with the inner regions serial, which is how a CPU renders 128 small views -- not 128 threads forking inside each. */
void FN(set_num_threads)(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}

int FN(num_threads)(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
