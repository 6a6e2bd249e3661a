// Projection + z-buffer + feature gather of the object-level 2D->3D fusion; see include/unipre3d_fusion.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "unipre3d_fusion.h"

namespace {

// pixel of one point exactly as fusion/feat_fusion.py:46-54 computes it; key < 0 when outside (:91-97)
__device__ __forceinline__ int project(const float* __restrict__ cp, int H, int W, float fx, float fy, float cx, float cy, int& px,
                                       int& py, float& depth) {
  const float x = cp[0], y = cp[1], z = cp[2];
  depth = z;
  const float u = rintf((x * fx) / z + cx), v = rintf((y * fy) / z + cy);   // torch.round = half to even
  if (!(fabsf(u) < 1e9f) || !(fabsf(v) < 1e9f)) return -1;                  // NaN / inf / beyond any image
  px = (int)u; py = (int)v;
  if (px < 0 || py < 0 || px >= H || py >= W || !(z >= 0.f)) return -1;
  return py * H + px;
}

__global__ void zbuf_min_kernel(int N, int H, int W, int total, float fx, float fy, float cx, float cy,
                                const float* __restrict__ camera_points, uint32_t* __restrict__ zbuf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int b = t / N;
  int px, py; float d;
  const int key = project(camera_points + (size_t)t * 4, H, W, fx, fy, cx, cy, px, py, d);
  if (key >= 0) atomicMin(&zbuf[(size_t)b * H * W + key], __float_as_uint(d == 0.f ? 0.f : d));   // d >= 0: bits are ordered
}

__global__ void gather_kernel(int N, int C, int H, int W, int total, float fx, float fy, float cx, float cy,
                              const float* __restrict__ camera_points, const float* __restrict__ feat,
                              const uint32_t* __restrict__ zbuf, float* __restrict__ mapped, int32_t* __restrict__ sel) {
  // one wave per point: lanes stride over the C channels (mapped rows are contiguous)
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= total) return;
  const int b = t / N;
  int px = 0, py = 0; float d;
  const int key = project(camera_points + (size_t)t * 4, H, W, fx, fy, cx, cy, px, py, d);
  const bool win = key >= 0 && zbuf[(size_t)b * H * W + key] == __float_as_uint(d == 0.f ? 0.f : d);
  if (lane == 0) sel[t] = win ? px * W + py : -1;
  float* o = mapped + (size_t)t * C;
  const float* f = feat + (size_t)b * C * H * W + (size_t)px * W + py;
  for (int c = lane; c < C; c += 64) o[c] = win ? f[(size_t)c * H * W] : 0.f;
}

__global__ void scatter_grad_kernel(int N, int C, int H, int W, int total, const float* __restrict__ grad_mapped,
                                    const int32_t* __restrict__ sel, float* __restrict__ grad_feat) {
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= total) return;
  const int s = sel[t];
  if (s < 0) return;
  const int b = t / N;
  float* g = grad_feat + (size_t)b * C * H * W + s;
  const float* go = grad_mapped + (size_t)t * C;
  for (int c = lane; c < C; c += 64) unsafeAtomicAdd(&g[(size_t)c * H * W], go[c]);
}

}  // namespace

extern "C" {

int u3d_zbuffer_fusion_forward(int B, int N, int C, int H, int W, float fx, float fy, float cx, float cy,
                               const float* camera_points, const float* image_features, float* mapped, int32_t* sel,
                               uint32_t* zbuf, void* stream) {
  if (B < 0 || N < 0 || C < 0 || H <= 0 || W <= 0) return 1;
  if (B == 0 || N == 0) return 0;
  if (!camera_points || !image_features || !mapped || !sel || !zbuf) return 1;
  hipStream_t s = (hipStream_t)stream;
  const int total = B * N;
  (void)hipMemsetAsync(zbuf, 0xFF, sizeof(uint32_t) * (size_t)B * H * W, s);
  hipLaunchKernelGGL(zbuf_min_kernel, dim3((total + 255) / 256), dim3(256), 0, s, N, H, W, total, fx, fy, cx, cy, camera_points, zbuf);
  hipLaunchKernelGGL(gather_kernel, dim3((total + 3) / 4), dim3(256), 0, s, N, C, H, W, total, fx, fy, cx, cy, camera_points,
                     image_features, zbuf, mapped, sel);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_zbuffer_fusion_backward(int B, int N, int C, int H, int W, const float* grad_mapped, const int32_t* sel,
                                float* grad_features, void* stream) {
  if (B < 0 || N < 0 || C < 0 || H <= 0 || W <= 0) return 1;
  if (B == 0 || N == 0 || C == 0) return 0;
  if (!grad_mapped || !sel || !grad_features) return 1;
  const int total = B * N;
  hipLaunchKernelGGL(scatter_grad_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, C, H, W, total, grad_mapped, sel,
                     grad_features);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

}  // extern "C"
