#include <hip/hip_runtime.h>
#include <cstdio>
// 8 independent chains, 8x unrolled -> 64 instr per loop iteration
#define OP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define ARGS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b)
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_mirror row_mask:0xf bank_mask:0xf\n"
#define DPPQ(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define CND64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define CNDFMA(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define CMP64(i) "v_cmp_le_f32_e64 s[20:21], %" #i ", %8\n"
#define MIN(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define CNDB(i) "v_cndmask_b32 %" #i ", %9, %8, vcc\n"
#define MOV(i) "v_mov_b32 %" #i ", %8\n"
#define MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define CMP(i) "v_cmp_le_f32 vcc, %" #i ", %8\n"
#define EXPFMA(i) "v_exp_f32 %" #i ", %" #i "\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile(OP8(FMA) ARGS);
      if (MODE == 1) asm volatile(OP8(EXP) ARGS);
      if (MODE == 2) asm volatile(OP8(RCP) ARGS);
      if (MODE == 3) asm volatile(OP8(DPP) ARGS);
      if (MODE == 4) asm volatile(OP8(DPPQ) ARGS);
      if (MODE == 5) asm volatile(OP8(CND) ARGS : "vcc");
      if (MODE == 6) asm volatile(OP8(MOV) ARGS);
      if (MODE == 7) asm volatile(OP8(MUL) ARGS);
      if (MODE == 8) asm volatile(OP8(CMP) ARGS : "vcc");
      if (MODE == 9) asm volatile(OP8(EXPFMA) ARGS);   // 1 exp + 3 fma per slot = 32 instr
      if (MODE == 12) asm volatile(OP8(CND64) ARGS : "s20", "s21");
      if (MODE == 13) asm volatile(OP8(CNDFMA) ARGS : "vcc");
      if (MODE == 14) asm volatile(OP8(CMP64) ARGS : "s20", "s21");
      if (MODE == 15) asm volatile(OP8(MIN) ARGS);
      if (MODE == 16) asm volatile(OP8(CNDB) ARGS : "vcc");
      if (MODE == 10) asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                                   "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n" ARGS);
      if (MODE == 11) asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                                   "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n" ARGS);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE> void run(float* d, const char* name, int per) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(8192), dim3(256), 0, 0, d, 1.0001f, 0.5f, 500);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double instr = 8192.0 * 4 * 500 * 8 * per;
  printf("%-12s %.3f ms  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, best, best * 1e-3 * 2.4e9 / (instr / 1024.0));
}
int main() {
  float* d; hipMalloc(&d, 256 * 8192 * 4);
  run<0>(d, "fma", 8); run<1>(d, "exp", 8); run<2>(d, "rcp", 8); run<3>(d, "dpp_rowmir", 8); run<4>(d, "dpp_quad", 8);
  run<5>(d, "cndmask", 8); run<6>(d, "mov", 8); run<7>(d, "mul", 8); run<8>(d, "cmp", 8); run<9>(d, "exp+3fma", 32);
  run<12>(d, "cnd_e64", 8); run<13>(d, "cnd+3fma", 32); run<14>(d, "cmp_e64", 8); run<15>(d, "min", 8); run<16>(d, "cnd_indep", 8);
  run<10>(d, "perm32swap", 8); run<11>(d, "perm16swap", 8);
  return 0;
}
