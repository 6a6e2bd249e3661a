// Projection + z-buffer + feature gather of the object-level 2D->3D fusion; see include/unipre3d_fusion.h.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

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

// Winner table (ABI 2): one 64-bit word per pixel, (depth bits << 32) | point index, indexed by sel = px*W + py.  One atomic min gives the
// z-test (high half: depths are >= 0, so their bits order like unsigned integers) AND, in the low half, the smallest index among the points
// that tie at the minimum -- the pixel's "first winner", which the backward gathers from.  Empty pixels keep the 0xFF fill.
__global__ void zbuf_min_kernel(int N, int H, int W, int total, float fx, float fy, float cx, float cy,
                                const float* __restrict__ camera_points, unsigned long long* __restrict__ zbuf) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  const int b = t / N;
  int px, py; float d;
  const int key = project(camera_points + (size_t)t * 4, H, W, fx, fy, cx, cy, px, py, d);
  if (key >= 0)
    atomicMin(&zbuf[(size_t)b * H * W + (size_t)px * W + py],
              ((unsigned long long)__float_as_uint(d == 0.f ? 0.f : d) << 32) | (unsigned)(t - b * N));
}

__global__ void gather_kernel(int N, int C, int H, int W, int total, float fx, float fy, float cx, float cy,
                              const float* __restrict__ camera_points, const float* __restrict__ feat,
                              const unsigned long long* __restrict__ zbuf, float* __restrict__ mapped, int32_t* __restrict__ sel) {
  // one wave per point: lanes stride over the C channels (mapped rows are contiguous)
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= total) return;
  const int b = t / N;
  int px = 0, py = 0; float d;
  const int key = project(camera_points + (size_t)t * 4, H, W, fx, fy, cx, cy, px, py, d);
  const bool win = key >= 0 && (uint32_t)(zbuf[(size_t)b * H * W + (size_t)px * W + py] >> 32) == __float_as_uint(d == 0.f ? 0.f : d);
  if (lane == 0) sel[t] = win ? px * W + py : -1;
  float* o = mapped + (size_t)t * C;
  const float* f = feat + (size_t)b * C * H * W + (size_t)px * W + py;
  for (int c = lane; c < C; c += 64) o[c] = win ? f[(size_t)c * H * W] : 0.f;
}

// Backward in GATHER form: every element of the (B,C,H,W) gradient is written exactly once --
// grad_feat[b][c][s] = grad_mapped[b][first winner of pixel s][c], or 0 -- instead of a zero-fill of the 805 MB followed by a scatter.
// One workgroup writes ONE channel plane front to back (64 KB contiguous at the reference's size) with NONTEMPORAL 16-byte stores and reads the
// item's winner words (B*H*W*8 bytes in all, L2-resident, shared by the C workgroups of an item) as it goes; fewer than 1 % of the pixels have
// a winner at the reference's sizes.  Measured forms (MI355X, 32 x 384 x 128 x 128, a plain zero-fill of the same bytes takes 117 us):
//   this one 134 - 137 us (5.9 TB/s);  the same with ordinary stores 253;  2 / 4 planes per workgroup 150;  a thread owning 4 pixels of 8 planes
//   (the winner words read once, but eight 1 KB store streams 64 KB apart per wave) 207 with either store kind;  a 1-bit-per-pixel "empty"
//   bitmap in front of the winner words, 8 pixels per thread 375 (32-byte lane stride: half-line nontemporal stores).
typedef float f32x4 __attribute__((ext_vector_type(4)));   // (a clang vector: __builtin_nontemporal_store takes no HIP float4 struct)

__global__ __launch_bounds__(256) void grad_plane_kernel(int N, int C, int HW, const float* __restrict__ grad_mapped,
                                                         const unsigned long long* __restrict__ zbuf, float* __restrict__ grad_feat) {
  const int HW4 = HW >> 2;
  const int b = blockIdx.y, c = blockIdx.x;
  const ulonglong2* zb = reinterpret_cast<const ulonglong2*>(zbuf + (size_t)b * HW);
  f32x4* o = reinterpret_cast<f32x4*>(grad_feat + ((size_t)b * C + c) * HW);
  const float* g = grad_mapped + (size_t)b * N * C + c;
  for (int q = threadIdx.x; q < HW4; q += 256) {
    const ulonglong2 z01 = zb[2 * (size_t)q], z23 = zb[2 * (size_t)q + 1];
    const uint32_t w0 = (uint32_t)z01.x, w1 = (uint32_t)z01.y, w2 = (uint32_t)z23.x, w3 = (uint32_t)z23.y;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((w0 & w1 & w2 & w3) != 0xffffffffu) {
      v.x = w0 != 0xffffffffu ? g[(size_t)w0 * C] : 0.f;
      v.y = w1 != 0xffffffffu ? g[(size_t)w1 * C] : 0.f;
      v.z = w2 != 0xffffffffu ? g[(size_t)w2 * C] : 0.f;
      v.w = w3 != 0xffffffffu ? g[(size_t)w3 * C] : 0.f;
    }
    __builtin_nontemporal_store(v, o + q);
  }
}

// H*W not a multiple of 4: one pixel per thread
template <int KC>
__global__ __launch_bounds__(256) void grad_dense1_kernel(int N, int C, int HW, const float* __restrict__ grad_mapped,
                                                          const unsigned long long* __restrict__ zbuf, float* __restrict__ grad_feat) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= HW) return;
  const int c0 = blockIdx.y * KC, b = blockIdx.z;
  const uint32_t w = (uint32_t)zbuf[(size_t)b * HW + q];
  float* o = grad_feat + ((size_t)b * C + c0) * HW + q;
  const float* g = grad_mapped + (size_t)b * N * C + c0;
  const int kc = C - c0 < KC ? C - c0 : KC;
  for (int k = 0; k < kc; ++k) o[(size_t)k * HW] = w != 0xffffffffu ? g[(size_t)w * C + k] : 0.f;
}

// Points that TIE with their pixel's first winner (same depth bits, larger index) add their rows afterwards: the reference keeps all tied
// points (fusion/feat_fusion.py:117-131).  One wave per point; nearly every wave leaves at once.
__global__ void tie_add_kernel(int N, int C, int HW, int total, const float* __restrict__ grad_mapped, const int32_t* __restrict__ sel,
                               const unsigned long long* __restrict__ zbuf, float* __restrict__ grad_feat) {
  const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= total) return;
  const int s = sel[t];
  if (s < 0) return;
  const int b = t / N;
  if ((uint32_t)zbuf[(size_t)b * HW + s] == (uint32_t)(t - b * N)) return;
  float* g = grad_feat + (size_t)b * C * HW + s;
  const float* go = grad_mapped + (size_t)t * C;
  for (int c = lane; c < C; c += 64) unsafeAtomicAdd(&g[(size_t)c * HW], go[c]);
}

}  // namespace

extern "C" {

int u3d_fusion_abi_version(void) { return U3D_FUSION_ABI_VERSION; }

size_t u3d_zbuffer_fusion_zbuf_bytes(int B, int H, int W) {
  const size_t npix = (size_t)(B > 0 ? B : 0) * (size_t)(H > 0 ? H : 0) * (size_t)(W > 0 ? W : 0);
  return npix * sizeof(uint64_t);
}

int u3d_zbuffer_fusion_forward(int B, int N, int C, int H, int W, float fx, float fy, float cx, float cy,
                               const float* camera_points, const float* image_features, float* mapped, int32_t* sel,
                               uint64_t* zbuf, void* stream) {
  if (B < 0 || N < 0 || C < 0 || H <= 0 || W <= 0) return 1;
  if (B == 0 || N == 0) return 0;
  if (!camera_points || !image_features || !mapped || !sel || !zbuf) return 1;
  hipStream_t s = (hipStream_t)stream;
  const int total = B * N;
  (void)hipMemsetAsync(zbuf, 0xFF, u3d_zbuffer_fusion_zbuf_bytes(B, H, W), s);   // 0xFF: no winner
  hipLaunchKernelGGL(zbuf_min_kernel, dim3((total + 255) / 256), dim3(256), 0, s, N, H, W, total, fx, fy, cx, cy, camera_points,
                     (unsigned long long*)zbuf);
  hipLaunchKernelGGL(gather_kernel, dim3((total + 3) / 4), dim3(256), 0, s, N, C, H, W, total, fx, fy, cx, cy, camera_points,
                     image_features, (const unsigned long long*)zbuf, mapped, sel);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_zbuffer_fusion_backward(int B, int N, int C, int H, int W, const float* grad_mapped, const int32_t* sel, const uint64_t* zbuf,
                                float* grad_features, void* stream) {
  if (B < 0 || N < 0 || C < 0 || H <= 0 || W <= 0) return 1;
  if (B == 0 || C == 0) return 0;
  if (!grad_features) return 1;
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  if (N == 0) {   // no points: the gradient is all zeros
    (void)hipMemsetAsync(grad_features, 0, sizeof(float) * (size_t)B * C * HW, s);
    return hipGetLastError() == hipSuccess ? 0 : 3;
  }
  if (!grad_mapped || !sel || !zbuf) return 1;
  if (B > 65535 || C > 65535) return 2;
  const unsigned long long* z = (const unsigned long long*)zbuf;
  if ((HW & 3) == 0) {
    hipLaunchKernelGGL(grad_plane_kernel, dim3(C, B), dim3(256), 0, s, N, C, HW, grad_mapped, z, grad_features);
  } else {
    constexpr int KC = 8;
    hipLaunchKernelGGL(grad_dense1_kernel<KC>, dim3((HW + 255) / 256, (C + KC - 1) / KC, B), dim3(256), 0, s, N, C, HW, grad_mapped, z,
                       grad_features);
  }
  const int total = B * N;
  hipLaunchKernelGGL(tie_add_kernel, dim3((total + 3) / 4), dim3(256), 0, s, N, C, HW, total, grad_mapped, sel, z, grad_features);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

}  // extern "C"
