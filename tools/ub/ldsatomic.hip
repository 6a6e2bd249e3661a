// LDS atomic rates on gfx950: lane-updates per clock and CU of ds_add_f32 / ds_add_u32 / ds_add_rtn_u32 / plain ds_write_b32 under three
// address patterns (distinct consecutive, random in 4096 words, all lanes of a wave on 2 addresses).  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ idx, float* __restrict__ out, int iters) {
  __shared__ float s_f[4096];
  unsigned* s_u = reinterpret_cast<unsigned*>(s_f);
  for (int i = threadIdx.x; i < 4096; i += 256) s_f[i] = 0.f;
  __syncthreads();
  int a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = idx[(blockIdx.x * 8 + u) * 256 + threadIdx.x] & 4095;
  unsigned acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) unsafeAtomicAdd(&s_f[a[u]], 1.0f);
      if (MODE == 1) atomicAdd(&s_u[a[u]], 1u);
      if (MODE == 2) acc += atomicAdd(&s_u[a[u]], 1u);
      if (MODE == 3) s_f[a[u]] = (float)i;
      if (MODE == 4) {   // float add as a compare-and-swap loop on the word (integer LDS atomics)
        unsigned old = s_u[a[u]], assumed;
        do {
          assumed = old;
          old = atomicCAS(&s_u[a[u]], assumed, __float_as_uint(__uint_as_float(assumed) + 1.0f));
        } while (old != assumed);
      }
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = s_f[threadIdx.x] + (float)acc;
}
int main() {
  const int nb = 1024, iters = 200;
  int* h = (int*)malloc(sizeof(int) * nb * 8 * 256);
  int* d; float* o;
  hipMalloc(&d, sizeof(int) * nb * 8 * 256); hipMalloc(&o, sizeof(float) * nb * 256);
  const char* pat[3] = {"consecutive", "random", "2 addresses per wave"};
  const char* mode[5] = {"ds_add_f32", "ds_add_u32", "ds_add_rtn_u32", "ds_write_b32", "CAS-loop float add"};
  for (int p = 0; p < 3; ++p) {
    unsigned r = 12345u;
    for (int i = 0; i < nb * 8 * 256; ++i) {
      r = r * 1664525u + 1013904223u;
      const int lane = i & 63;
      h[i] = p == 0 ? (i & 4095) : p == 1 ? (int)(r >> 8) : ((i >> 6) * 131 + (lane < 24 ? 0 : lane < 48 ? 7 : lane));
    }
    hipMemcpy(d, h, sizeof(int) * nb * 8 * 256, hipMemcpyHostToDevice);
    for (int m = 0; m < 5; ++m) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto run = [&]() {
        if (m == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, d, o, iters);
        if (m == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, d, o, iters);
        if (m == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, d, o, iters);
        if (m == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(256), 0, 0, d, o, iters);
        if (m == 4) hipLaunchKernelGGL(k<4>, dim3(nb), dim3(256), 0, 0, d, o, iters);
      };
      run(); hipDeviceSynchronize();
      hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double ups = (double)nb * 256 * 8 * iters;
      printf("%-22s %-16s %8.3f ms  %7.1f G updates/s  %5.2f per clock and CU (2.4 GHz, 256 CUs)\n", pat[p], mode[m], ms, ups / ms / 1e6, ups / (ms * 1e-3) / 2.4e9 / 256);
    }
  }
  return 0;
}
