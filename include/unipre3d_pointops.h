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
 */
#ifndef UNIPRE3D_POINTOPS_H
#define UNIPRE3D_POINTOPS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

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

#ifdef __cplusplus
}
#endif
#endif
