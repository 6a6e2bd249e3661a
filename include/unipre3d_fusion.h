/*
 * unipre3d_fusion.h -- C-ABI of the 2D->3D feature lookup of UniPre3D's object-level fusion.
 *
 * Replaces the body of FeatureFusion.__call__ between the projection and the concatenation
 * (fusion/feat_fusion.py:86-131): ~20 PyTorch ops and TWO host synchronisations (torch.nonzero at :100,
 * unique_ids.max().item() at :109) become three kernels with none:
 *   1. pixel = round((cam.x * fx) / cam.z + cx, (cam.y * fy) / cam.z + cy)   (round-half-even, fp32, :46-54)
 *      inside = 0 <= px < H and 0 <= py < W and depth >= 0                  (the reference's H/W swap kept, :91-97)
 *      z-buffer: zbuf[b][px*W + py] = min over the pixel's points of (depth bits << 32 | point index)
 *      (scatter_reduce amin, :106-114; the low half names the pixel's first winner for the backward)
 *   2. mapped[b][n][:] = image_features[b][:, px, py] for every point whose depth equals the pixel's minimum
 *      (ties keep ALL tied points, :117-131), zeros otherwise; sel[b][n] = px*W + py or -1.
 *   3. backward, gather form: grad_features[b][:, px, py] = grad_mapped[b][first winner of the pixel][:] or 0 -- every element of the
 *      (B,C,H,W) gradient written exactly ONCE (no zero-fill + scatter) -- then the rows of tied points are added.
 * camera_points [B][N][4] are the points already transformed by the world-to-camera matrix (the reference's own
 * torch.matmul at :42-45 stays in PyTorch so that pixel rounding is bit-identical).
 * zbuf: u3d_zbuffer_fusion_zbuf_bytes(B, H, W) bytes (B*H*W uint64 winner words), written by the forward and READ by the backward
 *       (keep it with sel); N < 2^32.
 * grad_features need NOT be initialised (ABI 2; ABI 1 accumulated with float atomics into a caller-zeroed buffer).
 * Returns 0 ok, 1 invalid argument, 2 unsupported shape, 3 launch failure.
 */
#ifndef UNIPRE3D_FUSION_H
#define UNIPRE3D_FUSION_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
int u3d_zbuffer_fusion_forward(int B, int N, int C, int H, int W, float fx, float fy, float cx, float cy,
                               const float* camera_points, const float* image_features, float* mapped, int32_t* sel,
                               uint64_t* zbuf, void* stream);
int u3d_zbuffer_fusion_backward(int B, int N, int C, int H, int W, const float* grad_mapped, const int32_t* sel,
                                const uint64_t* zbuf, float* grad_features, void* stream);
size_t u3d_zbuffer_fusion_zbuf_bytes(int B, int H, int W);
#define U3D_FUSION_ABI_VERSION 2
int u3d_fusion_abi_version(void);
#ifdef __cplusplus
}
#endif
#endif
