// Gradient validity check + global-norm clip as multi-tensor gfx950 kernels (include/unipre3d_gradclip.h; SURVEY.md N4,
// train_network.py:368-390).  HBM-bound byte work: pass 1 reads every gradient once (4 B per element), pass 2 reads and writes
// it once more and only runs when the norm exceeds max_norm.  One workgroup = one chunk of one tensor, found by a binary search
// of the chunk-prefix table (a few hundred tensors: 9 scalar steps); 16-byte loads on the aligned middle of the chunk, scalar
// head / tail (DDP bucket views sit at arbitrary 4-byte offsets).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "unipre3d_gradclip.h"

namespace {

constexpr int GC_THREADS = 512;
struct __attribute__((aligned(16))) Partial { double sumsq; float amax; uint32_t bad; };
static_assert(sizeof(Partial) == U3D_GC_PARTIAL_BYTES, "partial record size is part of the ABI");
struct State { double total_norm, amax; float coef, grad_scale, found_inf, reserved; };
static_assert(sizeof(State) == U3D_GC_STATE_BYTES, "state block size is part of the ABI");

__device__ __forceinline__ int find_tensor(const int32_t* __restrict__ first, int n, int chunk) {
  int lo = 0, hi = n;   // largest t with first[t] <= chunk
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (first[mid] <= chunk) lo = mid; else hi = mid;
  }
  return lo;
}

struct Acc {
  double s = 0.0;
  float m = 0.f;
  uint32_t bad = 0u;
  __device__ __forceinline__ void add(float x) {
    const float a = fabsf(x);
    const bool fin = a <= 3.402823466e38f;   // NaN and +-Inf both fail the comparison (train_network.py:377: isnan(...) or isinf(...))
    bad |= !fin;
    m = fmaxf(m, fin ? a : 0.f);        // largest FINITE |g| (the flag carries the rest)
    s += (double)x * (double)x;         // f64: a finite fp32 gradient cannot overflow the sum of squares
  }
};

__global__ __launch_bounds__(GC_THREADS) void gc_stats_kernel(const float* const* __restrict__ ptrs, const int64_t* __restrict__ numel,
                                                              const int32_t* __restrict__ first, int n, Partial* __restrict__ part) {
  __shared__ double s_s[GC_THREADS / 64];
  __shared__ float s_m[GC_THREADS / 64];
  __shared__ uint32_t s_b[GC_THREADS / 64];
  const int chunk = blockIdx.x;
  const int t = find_tensor(first, n, chunk);
  const int64_t off = (int64_t)(chunk - first[t]) * U3D_GC_CHUNK;
  const int64_t left = numel[t] - off;
  const int cnt = (int)(left < U3D_GC_CHUNK ? left : U3D_GC_CHUNK);
  const float* __restrict__ p = ptrs[t] + off;
  int head = (int)(((16u - (uint32_t)((uintptr_t)p & 15u)) & 15u) >> 2);
  if (head > cnt) head = cnt;
  const int nvec = (cnt - head) >> 2, tail0 = head + (nvec << 2);
  Acc a;
  const float4* __restrict__ pv = reinterpret_cast<const float4*>(p + head);
  int v = threadIdx.x;
  for (; v + 3 * GC_THREADS < nvec; v += 4 * GC_THREADS) {   // four independent 16-byte loads in flight per thread
    const float4 x0 = pv[v], x1 = pv[v + GC_THREADS], x2 = pv[v + 2 * GC_THREADS], x3 = pv[v + 3 * GC_THREADS];
    a.add(x0.x); a.add(x0.y); a.add(x0.z); a.add(x0.w);
    a.add(x1.x); a.add(x1.y); a.add(x1.z); a.add(x1.w);
    a.add(x2.x); a.add(x2.y); a.add(x2.z); a.add(x2.w);
    a.add(x3.x); a.add(x3.y); a.add(x3.z); a.add(x3.w);
  }
  for (; v < nvec; v += GC_THREADS) {
    const float4 x0 = pv[v];
    a.add(x0.x); a.add(x0.y); a.add(x0.z); a.add(x0.w);
  }
  if ((int)threadIdx.x < head) a.add(p[threadIdx.x]);
  if (tail0 + (int)threadIdx.x < cnt) a.add(p[tail0 + threadIdx.x]);
  // wave butterfly, then the waves of the block in a fixed order
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a.s += __shfl_xor(a.s, o);
    a.m = fmaxf(a.m, __shfl_xor(a.m, o));
    a.bad |= __shfl_xor(a.bad, o);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { s_s[wave] = a.s; s_m[wave] = a.m; s_b[wave] = a.bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0; float m = 0.f; uint32_t b = 0u;
#pragma unroll
    for (int w = 0; w < GC_THREADS / 64; ++w) { s += s_s[w]; m = fmaxf(m, s_m[w]); b |= s_b[w]; }
    part[chunk] = Partial{s, m, b};
  }
}

constexpr int FIN_THREADS = 1024;
__global__ __launch_bounds__(FIN_THREADS) void gc_finalize_kernel(const Partial* __restrict__ part, int n_chunks, float max_norm,
                                                                  State* __restrict__ state) {
  __shared__ double s_s[FIN_THREADS];
  __shared__ float s_m[FIN_THREADS];
  __shared__ uint32_t s_b[FIN_THREADS];
  double s = 0.0; float m = 0.f; uint32_t b = 0u;
  for (int i = threadIdx.x; i < n_chunks; i += FIN_THREADS) { const Partial q = part[i]; s += q.sumsq; m = fmaxf(m, q.amax); b |= q.bad; }
  s_s[threadIdx.x] = s; s_m[threadIdx.x] = m; s_b[threadIdx.x] = b;
  __syncthreads();
  for (int o = FIN_THREADS / 2; o > 0; o >>= 1) {   // fixed-order tree: the same partials give the same bits every run
    if ((int)threadIdx.x < o) {
      s_s[threadIdx.x] += s_s[threadIdx.x + o];
      s_m[threadIdx.x] = fmaxf(s_m[threadIdx.x], s_m[threadIdx.x + o]);
      s_b[threadIdx.x] |= s_b[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const bool bad = s_b[0] != 0u;
    const double total = sqrt(s_s[0]);
    // torch.nn.utils.clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    double coef = (double)max_norm / (total + 1e-6);
    if (!(coef < 1.0) || bad) coef = 1.0;
    State st;
    st.total_norm = bad ? __longlong_as_double(0x7ff8000000000000ll) : total;
    st.amax = (double)s_m[0];
    st.coef = (float)coef;
    st.grad_scale = (float)(1.0 / coef);
    st.found_inf = bad ? 1.f : 0.f;
    st.reserved = 0.f;
    *state = st;
  }
}

__global__ __launch_bounds__(GC_THREADS) void gc_scale_kernel(float* const* __restrict__ ptrs, const int64_t* __restrict__ numel,
                                                              const int32_t* __restrict__ first, int n, const State* __restrict__ state) {
  const float coef = state->coef;
  if (coef == 1.f || state->found_inf != 0.f) return;   // wave-uniform: nothing to clip (or the step is skipped anyway)
  const int chunk = blockIdx.x;
  const int t = find_tensor(first, n, chunk);
  const int64_t off = (int64_t)(chunk - first[t]) * U3D_GC_CHUNK;
  const int64_t left = numel[t] - off;
  const int cnt = (int)(left < U3D_GC_CHUNK ? left : U3D_GC_CHUNK);
  float* __restrict__ p = ptrs[t] + off;
  int head = (int)(((16u - (uint32_t)((uintptr_t)p & 15u)) & 15u) >> 2);
  if (head > cnt) head = cnt;
  const int nvec = (cnt - head) >> 2, tail0 = head + (nvec << 2);
  float4* __restrict__ pv = reinterpret_cast<float4*>(p + head);
  for (int v = threadIdx.x; v < nvec; v += GC_THREADS) {
    float4 x = pv[v];
    x.x *= coef; x.y *= coef; x.z *= coef; x.w *= coef;
    pv[v] = x;
  }
  if ((int)threadIdx.x < head) p[threadIdx.x] *= coef;
  if (tail0 + (int)threadIdx.x < cnt) p[tail0 + threadIdx.x] *= coef;
}

int launched() { return hipGetLastError() == hipSuccess ? 0 : 3; }

}  // namespace

extern "C" {

int u3d_gradclip_stats(const void* const* grad_ptrs, const int64_t* numel, const int32_t* first_chunk, int32_t n_tensors,
                       int32_t n_chunks, void* partials, void* stream) {
  if (n_tensors < 0 || n_chunks < 0) return 1;
  if (n_tensors == 0 || n_chunks == 0) return 0;
  if (!grad_ptrs || !numel || !first_chunk || !partials) return 1;
  hipLaunchKernelGGL(gc_stats_kernel, dim3((uint32_t)n_chunks), dim3(GC_THREADS), 0, (hipStream_t)stream,
                     reinterpret_cast<const float* const*>(grad_ptrs), numel, first_chunk, n_tensors, (Partial*)partials);
  return launched();
}

int u3d_gradclip_finalize(const void* partials, int32_t n_chunks, float max_norm, void* state, void* stream) {
  if (n_chunks < 0 || !state || !(max_norm > 0.f)) return 1;
  if (n_chunks > 0 && !partials) return 1;
  hipLaunchKernelGGL(gc_finalize_kernel, dim3(1), dim3(FIN_THREADS), 0, (hipStream_t)stream, (const Partial*)partials, n_chunks, max_norm,
                     (State*)state);
  return launched();
}

int u3d_gradclip_scale(void* const* grad_ptrs, const int64_t* numel, const int32_t* first_chunk, int32_t n_tensors, int32_t n_chunks,
                       const void* state, void* stream) {
  if (n_tensors < 0 || n_chunks < 0) return 1;
  if (n_tensors == 0 || n_chunks == 0) return 0;
  if (!grad_ptrs || !numel || !first_chunk || !state) return 1;
  hipLaunchKernelGGL(gc_scale_kernel, dim3((uint32_t)n_chunks), dim3(GC_THREADS), 0, (hipStream_t)stream,
                     reinterpret_cast<float* const*>(grad_ptrs), numel, first_chunk, n_tensors, (const State*)state);
  return launched();
}

}  // extern "C"
