// Does a scalar instruction cost VALU issue time on gfx950?  8 waves per SIMD, 8 independent v_fma_f32 chains, with 0/1/2/3 SALU
// instructions (s_and_b64 on private SGPRs), an s_nop, or a v_cmp -> s_and -> v_cndmask ladder between them.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define ARGS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define SA "s_and_b64 s[20:21], s[22:23], s[24:25]\n"
#define SB "s_or_b64 s[26:27], s[22:23], s[24:25]\n"
#define SC "s_andn2_b64 s[28:29], s[22:23], s[24:25]\n"
#define F1(i) FMA(i) SA
#define F2(i) FMA(i) SA SB
#define F3(i) FMA(i) SA SB SC
#define FN(i) FMA(i) "s_nop 0\n"
#define LAD(i) "v_cmp_le_f32_e64 s[22:23], %" #i ", %8\n" "s_and_b64 s[20:21], s[22:23], s[24:25]\n" "v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n" FMA(i)
#define CLOB : "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29"
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile(OP8(FMA) ARGS);
      if (MODE == 1) asm volatile(OP8(F1) ARGS CLOB);
      if (MODE == 2) asm volatile(OP8(F2) ARGS CLOB);
      if (MODE == 3) asm volatile(OP8(F3) ARGS CLOB);
      if (MODE == 4) asm volatile(OP8(FN) ARGS);
      if (MODE == 5) asm volatile(OP8(LAD) ARGS CLOB);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
static float t0 = 0;
template <int MODE> void run(float* d, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(8192), dim3(256), 0, 0, d, 1.0001f, 0.5f, 300); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  if (MODE == 0) t0 = best;
  printf("%-44s %.3f ms = %.2f x the pure-fma stream\n", name, best, best / t0);
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8192 * 4);
  run<0>(d, "8 fma"); run<1>(d, "8 x (fma + 1 salu)"); run<2>(d, "8 x (fma + 2 salu)"); run<3>(d, "8 x (fma + 3 salu)"); run<4>(d, "8 x (fma + s_nop)");
  run<5>(d, "8 x (v_cmp, s_and, v_cndmask, fma)");
  return 0;
}
