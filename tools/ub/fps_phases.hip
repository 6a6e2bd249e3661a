// Where does a furthest-point-sampling selection's time go?  A copy of fps_kernel (u3d_pointops.hip; default contraction) with s_memtime stamps of
// wave 0 at the phase boundaries, averaged over the selections: centre read, distance scan, wave maximum, holder search, LDS atomic, barrier, read back.
// usage: fps_phases   (prints cycles per phase for 256 x 4 @ 1024 points, 512 x 4 @ 2048, 1024 x 8 @ 8192)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ uint32_t tie_key(uint32_t k, int lg, uint32_t bs) {
  const uint32_t r = lg ? (__brev(k & (bs - 1)) >> (32 - lg)) : 0u;
  return (r << 22) | (k >> lg);
}
__device__ __forceinline__ int decode(uint32_t tk, int lg) {
  const uint32_t r = tk >> 22, q = tk & 0x3FFFFFu;
  return (int)((q << lg) + (lg ? (__brev(r) >> (32 - lg)) : 0u));
}
template <int CTRL> __device__ __forceinline__ uint32_t dmax(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
  return o > v ? o : v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = dmax<0xB1>(v); v = dmax<0x4E>(v); v = dmax<0x141>(v); v = dmax<0x140>(v);
  uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
  for (int q = 1; q < 4; ++q) { const uint32_t u = (uint32_t)__builtin_amdgcn_readlane((int)v, q * 16); r = u > r ? u : r; }
  return r;
}
#define STAMP(i) do { if (wave0) { const unsigned long long now = __builtin_amdgcn_s_memtime(); acc[i] += now - last; last = now; } } while (0)
template <int NT, int PPT>
__global__ __launch_bounds__(NT) void fps(int n, int m, int lg, const float* __restrict__ dataset, int* __restrict__ idxs, unsigned long long* __restrict__ phases) {
  extern __shared__ float s_xyz[];
  __shared__ unsigned long long s_best[3];
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const bool wave0 = tid < 64 && bi == 0;
  const uint32_t wave_base = (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63);
  const float* ds = dataset + (size_t)bi * n * 3;
  int* out = idxs + (size_t)bi * m;
  for (int i = tid; i < n * 3; i += NT) s_xyz[i] = ds[i];
  if (tid < 3) s_best[tid] = 0ull;
  __syncthreads();
  const uint32_t bs = 1u << lg;
  float x[PPT], y[PPT], z[PPT]; uint32_t t[PPT];
  for (int i = 0; i < PPT; ++i) { const int k = tid + i * NT; const bool v = k < n; x[i] = v ? s_xyz[k * 3] : 0.f; y[i] = v ? s_xyz[k * 3 + 1] : 0.f; z[i] = v ? s_xyz[k * 3 + 2] : 0.f; t[i] = v ? __float_as_uint(1e10f) : 0u; }
  int old = 0, cur = 1, nxt = 2;
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = __builtin_amdgcn_s_memtime();
  for (int j = 1; j < m; ++j) {
    const float ox = s_xyz[old * 3], oy = s_xyz[old * 3 + 1], oz = s_xyz[old * 3 + 2];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    STAMP(0);
    uint32_t lmax = 0u;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float dx = x[i] - ox, dy = y[i] - oy, dz = z[i] - oz;
      const float d = fmaf(dz, dz, fmaf(dx, dx, dy * dy));
      t[i] = min(t[i], __float_as_uint(d));
      lmax = max(lmax, t[i]);
    }
    asm volatile("" : "+v"(lmax));
    STAMP(1);
    const uint32_t wmax = wave_max_u32(lmax);
    STAMP(2);
    uint32_t btk = 0u;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      unsigned long long mk = __ballot(t[i] == wmax);
      while (mk) {
        const uint32_t k = wave_base + (uint32_t)__builtin_ctzll(mk) + (uint32_t)(i * NT);
        mk &= mk - 1ull;
        if (k < (uint32_t)n) { const uint32_t tk = ~tie_key(k, lg, bs); btk = tk > btk ? tk : btk; }
      }
    }
    STAMP(3);
    if (lane == 0) atomicMax(&s_best[cur], ((unsigned long long)wmax << 32) | btk);
    if (tid == 0) s_best[nxt] = 0ull;
    STAMP(4);
    __syncthreads();
    STAMP(5);
    const unsigned long long w = s_best[cur];
    old = decode(~(uint32_t)w, lg);
    if (tid == 0) out[j] = old;
    cur = nxt; nxt = nxt == 2 ? 0 : nxt + 1;
    old = __builtin_amdgcn_readfirstlane(old);
    STAMP(6);
  }
  if (wave0 && lane == 0) for (int i = 0; i < 8; ++i) phases[i] = acc[i];
}
template <int NT, int PPT> void run(int n, int m, int lg, const char* name) {
  const int B = 16;
  std::vector<float> h((size_t)B * n * 3);
  srand(1); for (auto& v : h) v = (float)rand() / RAND_MAX;
  float* d; int* idx; unsigned long long* ph;
  (void)hipMalloc(&d, h.size() * 4); (void)hipMalloc(&idx, (size_t)B * m * 4); (void)hipMalloc(&ph, 64);
  (void)hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  if ((size_t)n * 12 > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fps<NT, PPT>), hipFuncAttributeMaxDynamicSharedMemorySize, n * 12);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int r = 0; r < 3; ++r) {
    (void)hipEventRecord(e0); hipLaunchKernelGGL((fps<NT, PPT>), dim3(B), dim3(NT), (size_t)n * 12, 0, n, m, lg, d, idx, ph); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  unsigned long long p[8]; (void)hipMemcpy(p, ph, 64, hipMemcpyDeviceToHost);
  const char* nm[7] = {"centre read (LDS)", "distance scan", "wave maximum", "holder search", "LDS atomic issue", "barrier", "read back + decode"};
  double tot = 0; for (int i = 0; i < 7; ++i) tot += (double)p[i];
  printf("%s: %d threads x %d points per lane, n = %d, m = %d: %.1f us = %.3f us per selection (with stamps); s_memtime ticks per selection %.0f\n", name, NT, PPT, n, m, best * 1e3, best * 1e3 / (m - 1), tot / (m - 1));
  for (int i = 0; i < 7; ++i) printf("    %-22s %7.1f ticks  %4.1f %%\n", nm[i], (double)p[i] / (m - 1), 100.0 * p[i] / tot);
  (void)hipFree(d); (void)hipFree(idx); (void)hipFree(ph);
}
int main() {
  run<256, 4>(1024, 512, 10, "1024 -> 512");
  run<512, 4>(2048, 1024, 10, "2048 -> 1024");
  run<1024, 8>(8192, 1024, 10, "8192 -> 1024");
  run<64, 4>(256, 128, 8, "256 -> 128 (one wave, multi-wave code)");
  return 0;
}
