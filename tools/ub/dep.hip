// Dependent-issue behaviour of gfx950 VALU: cycles per instruction of C independent v_fma_f32 chains per wave at W resident
// waves per SIMD (W limited through the LDS allocation of 256-thread workgroups).  Time by HIP events at an assumed 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define CMP(i) "v_cmp_le_f32_e64 s[20:21], %" #i ", %8\n"
#define ARGS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
template <int C, int LDS, int MIX>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  extern __shared__ unsigned char dyn[];
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MIX == 0) {
        if (C == 1) asm volatile(FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) FMA(0) ARGS);
        if (C == 2) asm volatile(FMA(0) FMA(1) FMA(0) FMA(1) FMA(0) FMA(1) FMA(0) FMA(1) ARGS);
        if (C == 4) asm volatile(FMA(0) FMA(1) FMA(2) FMA(3) FMA(0) FMA(1) FMA(2) FMA(3) ARGS);
        if (C == 8) asm volatile(FMA(0) FMA(1) FMA(2) FMA(3) FMA(4) FMA(5) FMA(6) FMA(7) ARGS);
      } else {   // every 4th instruction a compare on the chain's value
        if (C == 1) asm volatile(CMP(0) FMA(0) FMA(0) FMA(0) CMP(0) FMA(0) FMA(0) FMA(0) ARGS : "s20", "s21");
        if (C == 2) asm volatile(CMP(0) FMA(0) CMP(1) FMA(1) FMA(0) FMA(1) FMA(0) FMA(1) ARGS : "s20", "s21");
        if (C == 4) asm volatile(CMP(0) FMA(0) FMA(1) FMA(2) CMP(3) FMA(3) FMA(0) FMA(1) ARGS : "s20", "s21");
        if (C == 8) asm volatile(CMP(0) FMA(0) FMA(1) FMA(2) CMP(3) FMA(3) FMA(4) FMA(5) ARGS : "s20", "s21");
      }
    }
  }
  if (LDS < 0) dyn[0] = 1;
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int C, int MIX> void run(float* d, int waves) {
  // 160 KB LDS per CU: a 256-thread workgroup (one wave per SIMD) with 160/waves KB leaves `waves` resident per SIMD
  const int lds = waves >= 8 ? 0 : (160 * 1024 / waves) - 1024;
  hipFuncSetAttribute((const void*)k<C, 0, MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  const int iters = 200, grid = 4096;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<C, 0, MIX>), dim3(grid), dim3(256), lds, 0, d, 1.0001f, 0.5f, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double instr_per_simd = (double)grid * 4 * iters * 64 / 1024.0;
  printf("  %s chains %d waves/SIMD %d : %.2f cycles per instruction per SIMD  (%.1f per wave)\n", MIX ? "cmp+3fma" : "fma     ", C, waves,
         best * 1e-3 * 2.4e9 / instr_per_simd, best * 1e-3 * 2.4e9 / instr_per_simd * waves);
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4096 * 4);
  for (int w : {8, 4, 2, 1}) { run<1, 0>(d, w); run<2, 0>(d, w); run<4, 0>(d, w); run<8, 0>(d, w); }
  for (int w : {8, 4, 2, 1}) { run<1, 1>(d, w); run<2, 1>(d, w); run<4, 1>(d, w); run<8, 1>(d, w); }
  return 0;
}
