// Issue cost of VALU instructions on gfx950 when MIXED with full-rate fp32 work (tools/ub/ops.hip measures them back to back).
// 8 independent chains per lane; time by HIP events, reported relative to a pure v_fma_f32 stream of the same length.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define ARGS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
#define F "v_fma_f32 %" 
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define FMA3(i) FMA(i) FMA(i) FMA(i)
#define MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define CMP64(i) "v_cmp_le_f32_e64 s[20:21], %" #i ", %8\n"
#define CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[22:23]\n"
#define MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define DPPQ(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define SWZ(i) "ds_swizzle_b32 %" #i ", %" #i " offset:0x80b1\n"   /* quad-perm mode [1,0,3,2] */
#define ADD(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define CMP_3(i) CMP64(i) FMA3(i)
#define MIN_3(i) MIN(i) FMA3(i)
#define CND_3(i) CND64(i) FMA3(i)
#define DPP_3(i) DPPQ(i) FMA3(i)
#define DPP_1(i) DPPQ(i) FMA(i)
#define CMP_1(i) CMP64(i) FMA(i)
#define MIN_1(i) MIN(i) FMA(i)
#define EXP_7(i) EXP(i) FMA3(i) FMA3(i) FMA(i)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile(OP8(FMA) ARGS);
      if (MODE == 1) asm volatile(OP8(MAX3) ARGS);
      if (MODE == 2) asm volatile(OP8(MED3) ARGS);
      if (MODE == 3) asm volatile(OP8(MAXF) ARGS);
      if (MODE == 4) asm volatile(OP8(AND) ARGS);
      if (MODE == 5) asm volatile(OP8(OR3) ARGS);
      if (MODE == 6) asm volatile(OP8(ADDU) ARGS);
      if (MODE == 7) asm volatile(OP8(CMP_3) ARGS : "s20", "s21");
      if (MODE == 8) asm volatile(OP8(MIN_3) ARGS);
      if (MODE == 9) asm volatile(OP8(CND_3) ARGS : "s22", "s23");
      if (MODE == 10) asm volatile(OP8(DPP_3) ARGS);
      if (MODE == 11) asm volatile(OP8(DPP_1) ARGS);
      if (MODE == 12) asm volatile(OP8(CMP_1) ARGS : "s20", "s21");
      if (MODE == 13) asm volatile(OP8(MIN_1) ARGS);
      if (MODE == 14) asm volatile(OP8(EXP_7) ARGS);
      if (MODE == 15) asm volatile(OP8(SWZ) "s_waitcnt lgkmcnt(0)\n" OP8(ADD) ARGS);
      if (MODE == 16) asm volatile(OP8(ADD) ARGS);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
static float t_fma = 0.f;
template <int MODE> void run(float* d, const char* name, int per) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(8192), dim3(256), 0, 0, d, 1.0001f, 0.5f, 300);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  if (MODE == 0) t_fma = best / 8;   // per instruction slot
  printf("%-28s %.3f ms  = %.2f fma-slots per group of %d instructions\n", name, best, best / t_fma, per);
}
int main() {
  float* d; hipMalloc(&d, 256 * 8192 * 4);
  run<0>(d, "8 fma", 8); run<16>(d, "8 add", 8); run<1>(d, "8 max3", 8); run<2>(d, "8 med3", 8); run<3>(d, "8 max", 8); run<4>(d, "8 and_b32", 8);
  run<5>(d, "8 or3_b32", 8); run<6>(d, "8 add_u32", 8);
  run<7>(d, "8 x (cmp_e64 + 3 fma)", 32); run<8>(d, "8 x (min + 3 fma)", 32); run<9>(d, "8 x (cndmask_e64 + 3 fma)", 32);
  run<10>(d, "8 x (add_dpp + 3 fma)", 32); run<11>(d, "8 x (add_dpp + fma)", 16); run<12>(d, "8 x (cmp_e64 + fma)", 16);
  run<13>(d, "8 x (min + fma)", 16); run<14>(d, "8 x (exp + 7 fma)", 64); run<15>(d, "8 swizzle + wait + 8 add", 16);
  return 0;
}
