// Issue cost of the non-FMA opcodes of render_fb_wave_kernel's loop bodies on gfx950, relative to a v_fma_f32 stream
// timed in the same process: 8192 workgroups x 256 threads (8 waves per SIMD), 8 independent chains per lane, 64 slots per loop
// iteration.  cycles = 2 x (time / time of the pure v_fma_f32 stream): the anchor is MI355X_MICROARCH.md's "v_fma_f32 (wave64) 2 cyc".
// Prints one JSON object per line; tools/ub_opcost.sh collects them into profiles/r06/opcode_issue_costs.json.
#include <hip/hip_runtime.h>
#include <cstdio>
#define OP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define ARGS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b), "v"(addr)
#define FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define FMAC(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
#define MUL(i) "v_mul_f32_e32 %" #i ", %" #i ", %8\n"
#define MUL64(i) "v_mul_f32_e64 %" #i ", %" #i ", %8\n"
#define SUB(i) "v_sub_f32_e32 %" #i ", %" #i ", %8\n"
#define MOV(i) "v_mov_b32_e32 %" #i ", %8\n"
#define EXP(i) "v_exp_f32_e32 %" #i ", %" #i "\n"
#define RCP(i) "v_rcp_f32_e32 %" #i ", %" #i "\n"
#define MIN(i) "v_min_f32_e32 %" #i ", %" #i ", %8\n"
#define MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define CMPVCC(i) "v_cmp_ge_f32_e32 vcc, %" #i ", %8\n"
#define CMPS(i) "v_cmp_gt_f32_e64 s[20:21], %" #i ", %8\n"
#define CMPU(i) "v_cmp_lt_u32_e64 s[20:21], %" #i ", %8\n"
#define CND64(i) "v_cndmask_b32_e64 %" #i ", 0, %" #i ", s[22:23]\n"
#define DPPQ(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define DPPH(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_half_mirror row_mask:0xf bank_mask:0x5\n"
#define DPPR(i) "v_add_f32_dpp %" #i ", %" #i ", %" #i " row_ror:8 row_mask:0xf bank_mask:0x3\n"
#define FMACDPP(i) "v_fmac_f32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define RFL(i) "v_readfirstlane_b32 s20, %" #i "\n"
#define BPERM(i) "ds_bpermute_b32 %" #i ", %10, %" #i "\n"
#define RD128 "ds_read_b128 v[40:43], %10\n"
#define SAND "s_and_b64 s[24:25], s[26:27], s[28:29]\n"
#define FS1(i) FMA(i) SAND
#define FS2(i) FMA(i) SAND "s_or_b64 s[30:31], s[26:27], s[28:29]\n"
#define FNOP(i) FMA(i) "s_nop 1\n"
#define LAD(i) "v_cmp_ge_f32_e64 s[26:27], %" #i ", %8\n" "s_and_b64 s[24:25], s[26:27], s[28:29]\n" "v_cndmask_b32_e64 %" #i ", 0, %" #i ", s[24:25]\n"
#define CLOB : "vcc", "scc", "s20", "s21", "s24", "s25", "s26", "s27", "s30", "s31", "v40", "v41", "v42", "v43", "memory"

// packed fp32 (v_pk_*_f32: two fp32 operations per lane in one instruction, operands in aligned register pairs)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void kpk(float* out, float a, float b, int iters) {
  f32x2 x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
  const f32x2 pa = {a, a}, pb = {b, b};
#define PKARGS : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(pa), "v"(pb)
#define PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define PKMUL(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) asm volatile(OP8(PKFMA) PKARGS);
      if (OP == 1) asm volatile(OP8(PKMUL) PKARGS);
      if (OP == 2) asm volatile(OP8(PKADD) PKARGS);
    }
  }
  const f32x2 t = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;
}
template <int OP> void runpk(float* d, const char* name, float t_fma) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kpk<OP>, dim3(8192), dim3(256), 0, 0, d, 1.0001f, 0.5f, 300);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("{\"op\": \"%s\", \"ms\": %.4f, \"instructions_per_slot\": 1, \"time_over_fma_stream\": %.3f, \"cycles_per_slot_at_2_per_fma\": %.2f, \"note\": \"two fp32 operations per lane and instruction\"}\n",
         name, best, best / t_fma, 2.0 * best / t_fma);
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b;
  __syncthreads();
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  const int addr = ((threadIdx.x & 63) ^ 32) << 2;
  asm volatile("s_mov_b64 s[22:23], exec\n s_mov_b64 s[26:27], exec\n s_mov_b64 s[28:29], exec" ::: "s22", "s23", "s26", "s27", "s28", "s29");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile(OP8(FMA) ARGS CLOB);
      if (MODE == 1) asm volatile(OP8(FMAC) ARGS CLOB);
      if (MODE == 2) asm volatile(OP8(MUL) ARGS CLOB);
      if (MODE == 3) asm volatile(OP8(MUL64) ARGS CLOB);
      if (MODE == 4) asm volatile(OP8(SUB) ARGS CLOB);
      if (MODE == 5) asm volatile(OP8(MOV) ARGS CLOB);
      if (MODE == 6) asm volatile(OP8(EXP) ARGS CLOB);
      if (MODE == 7) asm volatile(OP8(RCP) ARGS CLOB);
      if (MODE == 8) asm volatile(OP8(MIN) ARGS CLOB);
      if (MODE == 9) asm volatile(OP8(MIN3) ARGS CLOB);
      if (MODE == 10) asm volatile(OP8(CMPVCC) ARGS CLOB);
      if (MODE == 11) asm volatile(OP8(CMPS) ARGS CLOB);
      if (MODE == 12) asm volatile(OP8(CMPU) ARGS CLOB);
      if (MODE == 13) asm volatile(OP8(CND64) ARGS CLOB);
      if (MODE == 14) asm volatile(OP8(DPPQ) ARGS CLOB);
      if (MODE == 15) asm volatile(OP8(DPPH) ARGS CLOB);
      if (MODE == 16) asm volatile(OP8(DPPR) ARGS CLOB);
      if (MODE == 17) asm volatile(OP8(FMACDPP) ARGS CLOB);
      if (MODE == 18) asm volatile(OP8(RFL) ARGS CLOB);
      if (MODE == 19) asm volatile(OP8(BPERM) "s_waitcnt lgkmcnt(0)\n" ARGS CLOB);
      if (MODE == 20) asm volatile(RD128 RD128 RD128 RD128 RD128 RD128 RD128 RD128 "s_waitcnt lgkmcnt(0)\n" ARGS CLOB);
      if (MODE == 21) asm volatile(OP8(FS1) ARGS CLOB);
      if (MODE == 22) asm volatile(OP8(FS2) ARGS CLOB);
      if (MODE == 23) asm volatile(OP8(FNOP) ARGS CLOB);
      if (MODE == 24) asm volatile(OP8(LAD) ARGS CLOB);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + lds[(threadIdx.x * 7) & 1023];
}
static float t_fma = 0;
template <int MODE> void run(float* d, const char* name, int per, const char* note) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(8192), dim3(256), 0, 0, d, 1.0001f, 0.5f, 300);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  if (MODE == 0) t_fma = best;
  // `per` instructions per slot (8 slots x 8 unrolls per loop iteration)
  printf("{\"op\": \"%s\", \"ms\": %.4f, \"instructions_per_slot\": %d, \"time_over_fma_stream\": %.3f, \"cycles_per_slot_at_2_per_fma\": %.2f, \"note\": \"%s\"}\n", name, best, per,
         best / t_fma, 2.0 * best / t_fma, note);
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 8192 * 4);
  run<0>(d, "v_fma_f32", 1, "anchor: 2 cycles per wave64 instruction per SIMD (MI355X_MICROARCH.md)");
  run<0>(d, "v_fma_f32 (repeat)", 1, "clock ramp check");
  run<1>(d, "v_fmac_f32_e32", 1, ""); run<2>(d, "v_mul_f32_e32", 1, ""); run<3>(d, "v_mul_f32_e64", 1, "VOP3 encoding");
  run<4>(d, "v_sub_f32_e32", 1, ""); run<5>(d, "v_mov_b32_e32", 1, ""); run<6>(d, "v_exp_f32_e32", 1, "transcendental");
  run<7>(d, "v_rcp_f32_e32", 1, "transcendental"); run<8>(d, "v_min_f32_e32", 1, ""); run<9>(d, "v_min3_f32", 1, "");
  run<10>(d, "v_cmp_ge_f32_e32 -> vcc", 1, ""); run<11>(d, "v_cmp_gt_f32_e64 -> sgpr pair", 1, ""); run<12>(d, "v_cmp_lt_u32_e64 -> sgpr pair", 1, "");
  run<13>(d, "v_cndmask_b32_e64 (sgpr pair)", 1, ""); run<14>(d, "v_add_f32_dpp quad_perm", 1, ""); run<15>(d, "v_add_f32_dpp row_half_mirror bank_mask:0x5", 1, "");
  run<16>(d, "v_add_f32_dpp row_ror:8 bank_mask:0x3", 1, ""); run<17>(d, "v_fmac_f32_dpp row_shr:1", 1, "");
  run<18>(d, "v_readfirstlane_b32", 1, "writes an SGPR"); run<19>(d, "ds_bpermute_b32", 1, "8 in flight, then s_waitcnt");
  run<20>(d, "ds_read_b128 (one address per wave)", 1, "8 in flight, then s_waitcnt");
  run<21>(d, "v_fma_f32 + s_and_b64", 2, "slot = 1 VALU + 1 SALU: the excess over 2 cycles is what a scalar instruction costs the SIMD");
  run<22>(d, "v_fma_f32 + s_and_b64 + s_or_b64", 3, "slot = 1 VALU + 2 SALU");
  run<23>(d, "v_fma_f32 + s_nop 1", 2, "the DPP hazard fences of the reduction");
  run<24>(d, "v_cmp_e64 + s_and_b64 + v_cndmask_e64", 3, "one mask ladder of the blend (compare -> combine -> select)");
  runpk<0>(d, "v_pk_fma_f32", t_fma); runpk<1>(d, "v_pk_mul_f32", t_fma); runpk<2>(d, "v_pk_add_f32", t_fma);
  return 0;
}
