/*
 * unipre3d_fusion.h -- C-ABI of the 2D->3D feature lookup of UniPre3D's object-level fusion.
 *
 * Replaces the body of FeatureFusion.__call__ between the projection and the concatenation
 * (fusion/feat_fusion.py:86-131): ~20 PyTorch ops and TWO host synchronisations (torch.nonzero at :100,
 * unique_ids.max().item() at :109) become three kernels with none:
 *   1. pixel = round((cam.x * fx) / cam.z + cx, (cam.y * fy) / cam.z + cy)   (round-half-even, fp32, :46-54)
 *      inside = 0 <= px < H and 0 <= py < W and depth >= 0                  (the reference's H/W swap kept, :91-97)
 *      z-buffer: zbuf[b*H*W + py*H + px] = min(depth)                       (scatter_reduce amin, :106-114)
 *   2. mapped[b][n][:] = image_features[b][:, px, py] for every point whose depth equals the pixel's minimum
 *      (ties keep ALL tied points, :117-131), zeros otherwise; sel[b][n] = px*W + py or -1.
 *   3. backward: grad_features[b][:, px, py] += grad_mapped[b][n][:] for selected points.
 * camera_points [B][N][4] are the points already transformed by the world-to-camera matrix (the reference's own
 * torch.matmul at :42-45 stays in PyTorch so that pixel rounding is bit-identical).
 * zbuf: B*H*W uint32 scratch.  grad_features must be zeroed by the caller (accumulated with float atomics).
 * Returns 0 ok, 1 invalid argument, 3 launch failure.
 */
#ifndef UNIPRE3D_FUSION_H
#define UNIPRE3D_FUSION_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int u3d_zbuffer_fusion_forward(int B, int N, int C, int H, int W, float fx, float fy, float cx, float cy,
                               const float* camera_points, const float* image_features, float* mapped, int32_t* sel,
                               uint32_t* zbuf, void* stream);
int u3d_zbuffer_fusion_backward(int B, int N, int C, int H, int W, const float* grad_mapped, const int32_t* sel,
                                float* grad_features, void* stream);
#ifdef __cplusplus
}
#endif
#endif
