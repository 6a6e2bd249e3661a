/*
 * unipre3d_pointops.h -- C-ABI of the MI355X (gfx950) point sampling / grouping operators.
 *
 * Drop-in for the reference's `pointnet2_batch_cuda` extension (openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24),
 * which the transformer / pointmlp backbones and the ShapeNet loader need before anything can feed the render-loss path
 * (openpoints/models/layers/subsample.py:77-107, group.py:76-203; dataset/shapenet.py:368).  Each function takes the
 * reference wrapper's integer arguments in the same order, raw DEVICE pointers instead of at::Tensor, and a HIP stream.
 * fp32 data, int32 indices, contiguous row-major.  Returns 0 on success, 1 invalid argument, 3 launch failure.
 *
 *   u3d_furthest_point_sampling  <- furthest_point_sampling_wrapper(b, n, m, points (B,N,3), temp (B,N), idx (B,M))
 *        Start index 0; ties in the arg-max resolve exactly as the reference's block reduction does for its block size
 *        opt_n_threads(n) (cuda_utils.h:10-14), see oracle/pointops_oracle.c.  `temp` is unused scratch kept for
 *        signature parity (may be NULL): minimum distances live in registers.
 *   u3d_ball_query               <- ball_query_wrapper(b, n, m, radius, nsample, new_xyz (B,M,3), xyz (B,N,3), idx (B,M,nsample))
 *        First `nsample` support points with d^2 < r^2 in index order, padded with the first hit; rows without any hit
 *        are written as zeros (the reference relies on the caller's zero_()).
 *   u3d_group_points[_grad]      <- group_points[_grad]_wrapper(b, c, n, npoints, nsample, points (B,C,N), idx, out (B,C,npoints,nsample))
 *   u3d_gather_points[_grad]     <- gather_points[_grad]_wrapper(b, c, n, npoints, points (B,C,N), idx (B,npoints), out (B,C,npoints))
 *        The *_grad forms ACCUMULATE into grad_points (B,C,N) with float atomics; zero it first like the reference does.
 *   u3d_three_nn                 <- three_nn_wrapper(b, n, m, unknown (B,N,3), known (B,M,3), dist2 (B,N,3), idx (B,N,3))
 *        (pointnet2_api.cpp:21, interpolate_gpu.cu:16-59) squared distances and indices of the three nearest known points of
 *        every unknown point, ascending; ties keep the lower index (the reference's strict `<` insertions in index order);
 *        with fewer than three known points the unfilled slots hold the reference's initial 1e40 (stored as +inf) and index 0.
 *   u3d_three_interpolate        <- three_interpolate_wrapper(b, c, m, n, points (B,C,M), idx (B,N,3), weight (B,N,3), out (B,C,N))
 *        (:22, interpolate_gpu.cu:84-103) out[b][c][i] = sum_k weight[b][i][k] * points[b][c][idx[b][i][k]].
 *   u3d_three_interpolate_grad   <- three_interpolate_grad_wrapper(b, c, n, m, grad_out (B,C,N), idx, weight, grad_points (B,C,M))
 *        (:23, interpolate_gpu.cu:127-148) ACCUMULATES grad_out * weight into grad_points with float atomics; zero it first
 *        (openpoints/models/layers/upsampling.py:86).
 */
#ifndef UNIPRE3D_POINTOPS_H
#define UNIPRE3D_POINTOPS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/*
 * Floating-point contraction of the reference's three-term sums.  `dx*dx + dy*dy + dz*dz` (sampling_gpu.cu:140, ball_query_gpu.cu:38,
 * interpolate_gpu.cu:40) and `w0*p0 + w1*p1 + w2*p2` (interpolate_gpu.cu:103) are built by nvcc with its default -fmad=true, i.e. as
 * one rounded product and two fused multiply-adds; WHICH product is the rounded one is the compiler's choice, it changes results in
 * the last bit, and furthest point sampling turns a last-bit difference into a different index set (ties between equidistant
 * points).  The reference binary cannot be built here (no nvcc), so the choice is a process-wide mode of this library, each
 * mode bit-exact against oracle/pointops_oracle.c built the same way:
 *   U3D_PO_FMA_LLVM   fma(c, c, fma(a, a, b*b))  -- DEFAULT.  NVVM is LLVM: its DAG combiner folds the multiply of the FIRST operand of an
 *                     fadd into an fma before it looks at the second (visitFADDForFMACombine: "fold (fadd (fmul x, y), z) -> (fma x, y, z)"),
 *                     so ((a*a + b*b) + c*c) becomes fma(c, c, fma(a, a, b*b)) with b*b the lone rounded product.
 *   U3D_PO_FMA_CHAIN  fma(c, c, fma(b, b, a*a))  -- the left-to-right reading rounds 1-3 of this build used (first product rounded).
 *   U3D_PO_NO_FMA     every product rounded, sums left to right: what nvcc emits under -fmad=false.
 * tests/test_pointops_oracle.py reports on which clouds the three differ (random clouds: FPS never; lattices with exact ties: often).
 */
#define U3D_PO_FMA_LLVM 0
#define U3D_PO_FMA_CHAIN 1
#define U3D_PO_NO_FMA 2
int u3d_pointops_set_contraction(int mode); /* 0 ok, 1 unknown mode; applies to the launches that follow */
int u3d_pointops_get_contraction(void);

int u3d_furthest_point_sampling(int b, int n, int m, const float* points, float* temp, int32_t* idx, void* stream);
int u3d_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int32_t* idx,
                   void* stream);
int u3d_group_points(int b, int c, int n, int npoints, int nsample, const float* points, const int32_t* idx, float* out,
                     void* stream);
int u3d_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out, const int32_t* idx,
                          float* grad_points, void* stream);
int u3d_gather_points(int b, int c, int n, int npoints, const float* points, const int32_t* idx, float* out, void* stream);
int u3d_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out, const int32_t* idx, float* grad_points,
                           void* stream);
int u3d_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2, int32_t* idx, void* stream);
int u3d_three_interpolate(int b, int c, int m, int n, const float* points, const int32_t* idx, const float* weight, float* out,
                          void* stream);
int u3d_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int32_t* idx, const float* weight,
                               float* grad_points, void* stream);

#ifdef __cplusplus
}
#endif
#endif
