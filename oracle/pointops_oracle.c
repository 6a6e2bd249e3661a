/*
 * oracle/pointops_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's `pointnet2_batch_cuda` operators, following the in-tree CUDA
 * sources line by line -- INCLUDING the order in which the CUDA block reduction resolves ties, because furthest point
 * sampling is an integer-valued, chaotic selection (one different pick changes every later one):
 *   furthest_point_sampling   openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:94-216 (+ launcher :218-253,
 *                             block size opt_n_threads(n) from cuda_utils.h:10-14, temp = 1e10 from
 *                             openpoints/models/layers/subsample.py:93)
 *   ball_query                src/ball_query_gpu.cu:15-51   (idx zero-initialised: layers/group.py:192)
 *   group_points (+grad)      src/group_points_gpu.cu:53-72, 14-31
 *   gather_points (+grad)     src/sampling_gpu.cu:15-31, 53-70
 *   three_nn                  src/interpolate_gpu.cu:16-59
 *   three_interpolate (+grad) src/interpolate_gpu.cu:84-103, 127-148
 * The reference sources are CUDA (no nvcc, no NVIDIA device here) and ship no test vectors, so this restatement is
 * "parity unpinned" against the reference BINARY; it is pinned to the reference SOURCE by construction: the
 * simulated thread loop / shared-memory tree below is the kernel's own control flow executed sequentially.
 * One arithmetic choice cannot be read off the source: how nvcc (-fmad=true) contracts the three-term sums.  All three candidates
 * are implemented (po_set_contraction; include/unipre3d_pointops.h explains why mode 0, LLVM's combiner order, is the default) and the
 * HIP kernels are bit-exact against this file in each of them (tests/test_gpu_pointops.py); tests/test_pointops_oracle.py counts
 * where the modes disagree.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* a0*a1 + b0*b1 + c0*c1 in the three contractions of include/unipre3d_pointops.h (this file is built -ffp-contract=off, so the
 * plain products and sums below are rounded one by one): 0 = LLVM / NVVM combiner order, 1 = left-to-right chain, 2 = no fma */
static int g_mode = 0;
void po_set_contraction(int mode) { g_mode = mode; }
int po_get_contraction(void) { return g_mode; }
static float sum3(float a0, float a1, float b0, float b1, float c0, float c1) {
  if (g_mode == 0) return fmaf(c0, c1, fmaf(a0, a1, b0 * b1));
  if (g_mode == 1) return fmaf(c0, c1, fmaf(b0, b1, a0 * a1));
  return (a0 * a1 + b0 * b1) + c0 * c1;
}
static float dist2(const float* a, const float* b) {
  const float dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
  return sum3(dx, dx, dy, dy, dz, dz);
}

/* cuda_utils.h:10-14 */
int po_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 1024) v = 1024;
  if (v < 1) v = 1;
  return v;
}

/* sampling_gpu.cu:100-216: one simulated block per batch element */
void po_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp, int* idxs) {
  if (m <= 0) return;
  const int bs = po_opt_n_threads(n);
  float* dists = (float*)malloc(sizeof(float) * bs);
  int* dists_i = (int*)malloc(sizeof(int) * bs);
  for (int bi = 0; bi < b; ++bi) {
    const float* ds = dataset + (size_t)bi * n * 3;
    float* tp = temp + (size_t)bi * n;
    int* out = idxs + (size_t)bi * m;
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      for (int tid = 0; tid < bs; ++tid) {       /* :131-150 per-thread strided scan */
        int besti = 0;
        float best = -1.f;
        for (int k = tid; k < n; k += bs) {
          const float d = dist2(ds + old * 3, ds + k * 3);
          const float d2 = d < tp[k] ? d : tp[k];
          tp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1)      /* :152-211 shared-memory tree, __update :88-97 */
        for (int tid = 0; tid < s; ++tid) {
          const float v1 = dists[tid], v2 = dists[tid + s];
          const int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2;
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/* ball_query_gpu.cu:15-51 ; idx must be zero-initialised by the caller like layers/group.py:192 */
void po_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx) {
  const float radius2 = radius * radius;
  for (int bi = 0; bi < b; ++bi)
    for (int p = 0; p < m; ++p) {
      const float* q = new_xyz + ((size_t)bi * m + p) * 3;
      const float* pts = xyz + (size_t)bi * n * 3;
      int* o = idx + ((size_t)bi * m + p) * nsample;
      int cnt = 0;
      for (int k = 0; k < n; ++k) {
        /* (new_x - x)^2 + ... : same value as dist2(pts+k, q) term by term */
        const float dx = q[0] - pts[k * 3], dy = q[1] - pts[k * 3 + 1], dz = q[2] - pts[k * 3 + 2];
        const float d2 = sum3(dx, dx, dy, dy, dz, dz);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[l] = k;
          o[cnt] = k;
          ++cnt;
          if (cnt >= nsample) break;
        }
      }
    }
}

/* group_points_gpu.cu:53-72 */
void po_group_points(int b, int c, int n, int npoints, int nsample, const float* points, const int* idx, float* out) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int p = 0; p < npoints; ++p)
        for (int s = 0; s < nsample; ++s)
          out[(((size_t)bi * c + ci) * npoints + p) * nsample + s] =
              points[((size_t)bi * c + ci) * n + idx[((size_t)bi * npoints + p) * nsample + s]];
}

/* group_points_gpu.cu:14-31 (atomicAdd scatter); grad_points must be zero-initialised (layers/group.py:113) */
void po_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out, const int* idx,
                          float* grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int p = 0; p < npoints; ++p)
        for (int s = 0; s < nsample; ++s)
          grad_points[((size_t)bi * c + ci) * n + idx[((size_t)bi * npoints + p) * nsample + s]] +=
              grad_out[(((size_t)bi * c + ci) * npoints + p) * nsample + s];
}

/* sampling_gpu.cu:15-31 */
void po_gather_points(int b, int c, int n, int m, const float* points, const int* idx, float* out) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int p = 0; p < m; ++p)
        out[((size_t)bi * c + ci) * m + p] = points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + p]];
}

/* sampling_gpu.cu:53-70 */
void po_gather_points_grad(int b, int c, int n, int m, const float* grad_out, const int* idx, float* grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int p = 0; p < m; ++p)
        grad_points[((size_t)bi * c + ci) * n + idx[(size_t)bi * m + p]] += grad_out[((size_t)bi * c + ci) * m + p];
}

/* interpolate_gpu.cu:16-59: one simulated thread per (batch, unknown point); `double` bests initialised to 1e40, strict `<`
 * insertions while scanning the known points in index order; results stored as float / int. */
void po_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2v, int* idx) {
  for (int bi = 0; bi < b; ++bi)
    for (int p = 0; p < n; ++p) {
      const float* u = unknown + ((size_t)bi * n + p) * 3;
      const float* kn = known + (size_t)bi * m * 3;
      const float ux = u[0], uy = u[1], uz = u[2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
        /* (ux-x)*(ux-x) + (uy-y)*(uy-y) + (uz-z)*(uz-z), contracted like the other distances of this file */
        const float d = sum3(ux - x, ux - x, uy - y, uy - y, uz - z, uz - z);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d; besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d; besti2 = k;
        } else if (d < best3) {
          best3 = d; besti3 = k;
        }
      }
      float* od = dist2v + ((size_t)bi * n + p) * 3;
      int* oi = idx + ((size_t)bi * n + p) * 3;
      od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
}

/* interpolate_gpu.cu:84-103 ; weight[0]*p[idx0] + weight[1]*p[idx1] + weight[2]*p[idx2] in the selected contraction */
void po_three_interpolate(int b, int c, int m, int n, const float* points, const int* idx, const float* weight, float* out) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int p = 0; p < n; ++p) {
        const float* w = weight + ((size_t)bi * n + p) * 3;
        const int* ix = idx + ((size_t)bi * n + p) * 3;
        const float* row = points + ((size_t)bi * c + ci) * m;
        out[((size_t)bi * c + ci) * n + p] = sum3(w[0], row[ix[0]], w[1], row[ix[1]], w[2], row[ix[2]]);
      }
}

/* interpolate_gpu.cu:127-148 (three atomicAdds per thread); grad_points zero-initialised by the caller (upsampling.py:86) */
void po_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int* idx, const float* weight,
                               float* grad_points) {
  for (int bi = 0; bi < b; ++bi)
    for (int ci = 0; ci < c; ++ci)
      for (int p = 0; p < n; ++p) {
        const float g = grad_out[((size_t)bi * c + ci) * n + p];
        const float* w = weight + ((size_t)bi * n + p) * 3;
        const int* ix = idx + ((size_t)bi * n + p) * 3;
        float* row = grad_points + ((size_t)bi * c + ci) * m;
        row[ix[0]] += g * w[0];
        row[ix[1]] += g * w[1];
        row[ix[2]] += g * w[2];
      }
}
