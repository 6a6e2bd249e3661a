// How many single-wave workgroups does an MI355X CU really keep resident?  Each wave sleeps a fixed time (s_sleep), so
// kernel time = waves * t_wave / resident capacity.  Capacity is reported relative to the device's 256 CUs x 4 SIMDs.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDS, int NT>
__global__ __launch_bounds__(NT) void k(unsigned* out, int loops) {
  __shared__ unsigned char s[LDS > 0 ? LDS : 4];
  if (LDS > 0) s[threadIdx.x] = (unsigned char)threadIdx.x;
  for (int i = 0; i < loops; ++i) __builtin_amdgcn_s_sleep(127);
  if (LDS > 0 && s[(threadIdx.x * 7) % (LDS > 0 ? LDS : 4)] == 255 && loops < 0) out[0] = 1;
}
template <int LDS, int NT> void run(unsigned* d, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int loops = 8;
  // t_wave: one workgroup alone
  float t1 = 1e9, tn = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL((k<LDS, NT>), dim3(1), dim3(NT), 0, 0, d, loops); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < t1) t1 = ms;
  }
  const int N = 262144 / (NT / 64);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL((k<LDS, NT>), dim3(N), dim3(NT), 0, 0, d, loops); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < tn) tn = ms;
  }
  const double waves = (double)N * (NT / 64);
  printf("%-26s alone %.4f ms, %d workgroups %.3f ms -> resident waves per SIMD ~ %.2f (if a wave takes the 'alone' time)\n", name, t1, N, tn,
         waves * t1 / tn / 1024.0);
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 4096);
  run<0, 64>(d, "64 thr, no LDS"); run<2560, 64>(d, "64 thr, 2.5 KB LDS"); run<5120, 64>(d, "64 thr, 5 KB LDS"); run<10240, 64>(d, "64 thr, 10 KB LDS");
  run<0, 256>(d, "256 thr, no LDS"); run<20480, 256>(d, "256 thr, 20 KB LDS"); run<0, 128>(d, "128 thr, no LDS"); run<10240, 128>(d, "128 thr, 10 KB LDS");
  return 0;
}
