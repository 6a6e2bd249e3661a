// Point sampling / grouping operators for gfx950; semantics: oracle/pointops_oracle.c, which restates
// openpoints/cpp/pointnet2_batch/src/{sampling,ball_query,group_points}_gpu.cu.
//
//  * furthest point sampling: one 256- or 1024-thread workgroup per cloud, coordinates staged once in LDS, up to 32 points per
//    thread with their running minimum distances in REGISTERS (the reference round-trips a (B,N) temp array through
//    global memory every iteration), arg-max over (distance, inverted tie key): DPP row steps + 4 readlanes within the
//    wave, 4 wave results through double-buffered LDS => ONE barrier per selected point.  The tie key reproduces the order in which the
//    reference's shared-memory tree resolves equal distances for ITS block size, so the selection is bit-identical.
//  * ball query: one WAVE per query (the reference: one thread per query scanning all N points serially): 64 candidates
//    per step, ballot + popcount prefix keeps index order, early exit at nsample.
//  * group / gather (+grad): one thread per output element, coalesced on the output side.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "unipre3d_pointops.h"

namespace {

// `a*a + b*b + c*c` and `w0*p0 + w1*p1 + w2*p2` as the reference's kernels write them (sampling_gpu.cu:140, ball_query_gpu.cu:38,
// interpolate_gpu.cu:40, :103) are compiled by nvcc with -fmad=true: WHICH product is rounded on its own before the two fused
// multiply-adds decides ties, and FPS is a chaotic integer selection -- so the contraction is a call-time mode
// (u3d_pointops_set_contraction), bit-exact against oracle/pointops_oracle.c in every mode:
//   U3D_PO_FMA_LLVM (default)  fma(c, c, fma(a, a, b*b))  what LLVM's DAG combiner -- NVVM is LLVM -- makes of fadd(fadd(fmul, fmul), fmul):
//                              the FIRST operand's multiply is folded into the inner add, the second product stays a rounded fmul
//   U3D_PO_FMA_CHAIN           fma(c, c, fma(b, b, a*a))  rounds 1-3's reading (left-to-right chain, first product rounded)
//   U3D_PO_NO_FMA              ((a*a + b*b) + c*c) with every product rounded: -fmad=false
__device__ __forceinline__ float sum3(float a0, float a1, float b0, float b1, float c0, float c1, int cm) {   // a0*a1 + b0*b1 + c0*c1
#pragma clang fp contract(off)   // (only the fmaf calls below fuse; HIP's __fmul_rn / __fadd_rn are plain operators the optimiser may contract)
  if (cm == U3D_PO_FMA_LLVM) return fmaf(c0, c1, fmaf(a0, a1, b0 * b1));
  if (cm == U3D_PO_FMA_CHAIN) return fmaf(c0, c1, fmaf(b0, b1, a0 * a1));
  return (a0 * a1 + b0 * b1) + c0 * c1;
}
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz, int cm) {
  const float dx = bx - ax, dy = by - ay, dz = bz - az;
  return sum3(dx, dx, dy, dy, dz, dz, cm);
}
int g_contraction = U3D_PO_FMA_LLVM;   // host: mode of the launches that follow (process-wide, like a build flag of the reference)

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long w = __shfl_xor(v, o);
    v = w > v ? w : v;
  }
  return v;
}

// (distance, inverted tie key) arg-max across the wave without the LDS crossbar: four DPP steps leave every lane with
// its 16-lane row's winner, the four row winners are read into SGPRs and compared on the scalar unit.
template <int CTRL>
__device__ __forceinline__ void dpp_max_step(float& d, uint32_t& t) {
  const float od = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), CTRL, 0xf, 0xf, false));
  const uint32_t ot = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, CTRL, 0xf, 0xf, false);
  const bool take = od > d || (od == d && ot > t);
  d = take ? od : d;
  t = take ? ot : t;
}
__device__ __forceinline__ unsigned long long wave_argmax(float d, uint32_t t) {
  dpp_max_step<0xB1>(d, t);    // quad_perm [1,0,3,2]
  dpp_max_step<0x4E>(d, t);    // quad_perm [2,3,0,1]
  dpp_max_step<0x141>(d, t);   // row_half_mirror
  dpp_max_step<0x140>(d, t);   // row_mirror
  unsigned long long best = 0ull;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const unsigned long long k = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(__float_as_int(d), r * 16) << 32) |
                                 (uint32_t)__builtin_amdgcn_readlane((int)t, r * 16);
    best = k > best ? k : best;
  }
  return best;
}

// tie key of point k for the reference's block size bs = 2^lg (lg <= 10): equal distances are won by the smaller
// (bit-reversed (k mod bs), k div bs)  [per-thread strided scan keeps the first maximum; the tree keeps the lower slot].
// Packed as r << 22 | q (q = k div bs < 2^22 for any n < 2^32 / ... in practice n < 4M * bs).
__device__ __forceinline__ uint32_t fps_tie_key(uint32_t k, int lg, uint32_t bs) {
  const uint32_t r = lg ? (__brev(k & (bs - 1)) >> (32 - lg)) : 0u;
  return (r << 22) | (k >> lg);
}
__device__ __forceinline__ int fps_decode(uint32_t tk, int lg) {
  const uint32_t r = tk >> 22, q = tk & 0x3FFFFFu;
  return (int)((q << lg) + (lg ? (__brev(r) >> (32 - lg)) : 0u));
}

template <int FPS_THREADS, int PPT>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(int n, int m, int lg, int cm, const float* __restrict__ dataset,
                                                          int32_t* __restrict__ idxs) {
  constexpr int FPS_WAVES = FPS_THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) float s_xyz[];   // [n][3]
  __shared__ unsigned long long s_key[2][FPS_WAVES];
  const int bi = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* ds = dataset + (size_t)bi * n * 3;
  int32_t* out = idxs + (size_t)bi * m;
  for (int i = tid; i < n * 3; i += FPS_THREADS) s_xyz[i] = ds[i];
  __syncthreads();
  const uint32_t bs = 1u << lg;
  float x[PPT], y[PPT], z[PPT], t[PPT];
  uint32_t inv_tk[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * FPS_THREADS;
    const bool v = k < n;
    x[i] = v ? s_xyz[k * 3] : 0.f; y[i] = v ? s_xyz[k * 3 + 1] : 0.f; z[i] = v ? s_xyz[k * 3 + 2] : 0.f;
    t[i] = v ? 1e10f : 0.f;                        // padding slots: distance 0 and the lowest key, never beat a real point
    inv_tk[i] = v ? 0xFFFFFFFFu - fps_tie_key((uint32_t)k, lg, bs) : 0u;
  }
  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float ox = s_xyz[old * 3], oy = s_xyz[old * 3 + 1], oz = s_xyz[old * 3 + 2];
    float bd = -1.f;
    uint32_t bt = 0u;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float d = dist2(ox, oy, oz, x[i], y[i], z[i], cm);
      t[i] = fminf(d, t[i]);
      const bool take = t[i] > bd || (t[i] == bd && inv_tk[i] > bt);
      bd = take ? t[i] : bd;
      bt = take ? inv_tk[i] : bt;
    }
    const unsigned long long best = wave_argmax(bd, bt);
    if (lane == 0) s_key[j & 1][wave] = best;
    __syncthreads();
    unsigned long long w = s_key[j & 1][0];
#pragma unroll
    for (int q = 1; q < FPS_WAVES; ++q) {
      const unsigned long long u = s_key[j & 1][q];
      w = u > w ? u : w;
    }
    old = fps_decode(0xFFFFFFFFu - (uint32_t)w, lg);
    if (tid == 0) out[j] = old;
  }
}

// generic path for clouds larger than FPS_THREADS*8 points: minimum distances in global scratch
__global__ __launch_bounds__(1024) void fps_kernel_large(int n, int m, int lg, int cm, const float* __restrict__ dataset,
                                                         float* __restrict__ temp, int32_t* __restrict__ idxs) {
  constexpr int FPS_THREADS = 1024, FPS_WAVES = 16;
  __shared__ unsigned long long s_key[2][FPS_WAVES];
  const int bi = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* ds = dataset + (size_t)bi * n * 3;
  float* tp = temp + (size_t)bi * n;
  int32_t* out = idxs + (size_t)bi * m;
  const uint32_t bs = 1u << lg;
  for (int k = tid; k < n; k += FPS_THREADS) tp[k] = 1e10f;
  int old = 0;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float ox = ds[old * 3], oy = ds[old * 3 + 1], oz = ds[old * 3 + 2];
    unsigned long long best = 0ull;
    for (int k = tid; k < n; k += FPS_THREADS) {
      const float d = dist2(ox, oy, oz, ds[k * 3], ds[k * 3 + 1], ds[k * 3 + 2], cm);
      const float d2 = fminf(d, tp[k]);
      tp[k] = d2;
      const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (0xFFFFFFFFu - fps_tie_key((uint32_t)k, lg, bs));
      best = key > best ? key : best;
    }
    best = wave_max_u64(best);
    if (lane == 0) s_key[j & 1][wave] = best;
    __syncthreads();
    unsigned long long w = s_key[j & 1][0];
#pragma unroll
    for (int q = 1; q < FPS_WAVES; ++q) {
      const unsigned long long u = s_key[j & 1][q];
      w = u > w ? u : w;
    }
    old = fps_decode(0xFFFFFFFFu - (uint32_t)w, lg);
    if (tid == 0) out[j] = old;
  }
}

__global__ __launch_bounds__(256) void ball_query_kernel(int n, int m, int total_q, float radius2, int nsample, int cm,
                                                         const float* __restrict__ new_xyz, const float* __restrict__ xyz,
                                                         int32_t* __restrict__ idx) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + wave;          // global query = bi * m + p
  if (qi >= total_q) return;
  const int bi = qi / m;
  const float* q = new_xyz + (size_t)qi * 3;
  const float* pts = xyz + (size_t)bi * n * 3;
  int32_t* o = idx + (size_t)qi * nsample;
  const float qx = q[0], qy = q[1], qz = q[2];
  int cnt = 0, first = 0;
  for (int base = 0; base < n && cnt < nsample; base += 64) {
    const int k = base + lane;
    bool in = false;
    if (k < n) {
      const float dx = qx - pts[k * 3], dy = qy - pts[k * 3 + 1], dz = qz - pts[k * 3 + 2];
      in = sum3(dx, dx, dy, dy, dz, dz, cm) < radius2;
    }
    const unsigned long long bal = __ballot(in);
    if (bal) {
      if (cnt == 0) first = base + __ffsll((long long)bal) - 1;
      const int pos = cnt + (int)__popcll(bal & ((1ull << lane) - 1ull));
      if (in && pos < nsample) o[pos] = k;
      cnt += (int)__popcll(bal);
    }
  }
  if (cnt > nsample) cnt = nsample;
  const int pad = cnt > 0 ? first : 0;
  for (int l = cnt + lane; l < nsample; l += 64) o[l] = pad;
}

__global__ void group_points_kernel(int c, int n, int npoints, int nsample, const float* __restrict__ points,
                                    const int32_t* __restrict__ idx, float* __restrict__ out) {
  const int bi = blockIdx.z, ci = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npoints * nsample) return;
  const int src = idx[(size_t)bi * npoints * nsample + e];
  out[((size_t)bi * c + ci) * npoints * nsample + e] = points[((size_t)bi * c + ci) * n + src];
}

__global__ void group_points_grad_kernel(int c, int n, int npoints, int nsample, const float* __restrict__ grad_out,
                                         const int32_t* __restrict__ idx, float* __restrict__ grad_points) {
  const int bi = blockIdx.z, ci = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= npoints * nsample) return;
  const int dst = idx[(size_t)bi * npoints * nsample + e];
  unsafeAtomicAdd(&grad_points[((size_t)bi * c + ci) * n + dst], grad_out[((size_t)bi * c + ci) * npoints * nsample + e]);
}

// ---- three_nn / three_interpolate (+grad): interpolate_gpu.cu:16-140 ------------------------------------------------------
// three_nn: the reference gives every query a thread that streams all m known points from global memory.  Here a workgroup
// stages the known cloud through LDS in tiles (one coalesced copy per tile, then every lane reads the SAME LDS word per step:
// a broadcast, no bank conflicts) and each lane keeps its query's running three best in registers.  Scanning in index order
// with the reference's strict `<` insertions gives the three smallest (distance, index) pairs in lexicographic order, ties to
// the lower index -- reproduced exactly; an unfilled slot keeps the reference's initial 1e40 (stored as +inf) / index 0.
constexpr int NN_THREADS = 128;
constexpr int NN_TILE = 2048;   // known points per LDS tile: 24 KB
__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(int n, int m, int cm, const float* __restrict__ unknown,
                                                              const float* __restrict__ known, float* __restrict__ dist2,
                                                              int32_t* __restrict__ idx) {
  __shared__ float s_k[NN_TILE * 3];
  const int bi = blockIdx.y;
  const int pt = blockIdx.x * NN_THREADS + threadIdx.x;
  const bool live = pt < n;
  const float* u = unknown + ((size_t)bi * n + (live ? pt : 0)) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float* kb = known + (size_t)bi * m * 3;
  double best1 = 1e40, best2 = 1e40, best3 = 1e40;
  int i1 = 0, i2 = 0, i3 = 0;
  for (int k0 = 0; k0 < m; k0 += NN_TILE) {
    const int cnt = min(NN_TILE, m - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += NN_THREADS) s_k[e] = kb[(size_t)k0 * 3 + e];
    __syncthreads();
    for (int j = 0; j < cnt; ++j) {
      const float dx = ux - s_k[j * 3], dy = uy - s_k[j * 3 + 1], dz = uz - s_k[j * 3 + 2];
      const float d = sum3(dx, dx, dy, dy, dz, dz, cm);   // (ux-x)^2 + (uy-y)^2 + (uz-z)^2 in the selected contraction
      const int k = k0 + j;
      if (d < best1) { best3 = best2; i3 = i2; best2 = best1; i2 = i1; best1 = d; i1 = k; }
      else if (d < best2) { best3 = best2; i3 = i2; best2 = d; i2 = k; }
      else if (d < best3) { best3 = d; i3 = k; }
    }
  }
  if (live) {
    float* od = dist2 + ((size_t)bi * n + pt) * 3;
    int32_t* oi = idx + ((size_t)bi * n + pt) * 3;
    od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
  }
}

// three_interpolate: the reference launches one thread per (batch, channel, point), so every channel re-reads the point's three
// indices and weights.  Here a lane owns a point, loads them once and walks IC_CH channels: three gathers from the channel's
// row of m values (cache-resident) and one coalesced store per channel.  out = w0 p[i0] + w1 p[i1] + w2 p[i2], contracted
// left to right like nvcc does (fma(w2, p2, fma(w1, p1, w0 p0))).
constexpr int IC_CH = 8;
__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n, int cm, const float* __restrict__ points,
                                                                const int32_t* __restrict__ idx, const float* __restrict__ weight,
                                                                float* __restrict__ out) {
  const int bi = blockIdx.z, c0 = blockIdx.y * IC_CH;
  const int pt = blockIdx.x * 256 + threadIdx.x;
  if (pt >= n) return;
  const int32_t* ip = idx + ((size_t)bi * n + pt) * 3;
  const float* wp = weight + ((size_t)bi * n + pt) * 3;
  const int i0 = ip[0], i1 = ip[1], i2 = ip[2];
  const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
  const int c1 = min(c, c0 + IC_CH);
  for (int ci = c0; ci < c1; ++ci) {
    const float* row = points + ((size_t)bi * c + ci) * m;
    out[((size_t)bi * c + ci) * n + pt] = sum3(w0, row[i0], w1, row[i1], w2, row[i2], cm);
  }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(int c, int n, int m, const float* __restrict__ grad_out,
                                                                     const int32_t* __restrict__ idx,
                                                                     const float* __restrict__ weight,
                                                                     float* __restrict__ grad_points) {
  const int bi = blockIdx.z, c0 = blockIdx.y * IC_CH;
  const int pt = blockIdx.x * 256 + threadIdx.x;
  if (pt >= n) return;
  const int32_t* ip = idx + ((size_t)bi * n + pt) * 3;
  const float* wp = weight + ((size_t)bi * n + pt) * 3;
  const int i0 = ip[0], i1 = ip[1], i2 = ip[2];
  const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
  const int c1 = min(c, c0 + IC_CH);
  for (int ci = c0; ci < c1; ++ci) {
    const float g = grad_out[((size_t)bi * c + ci) * n + pt];
    float* row = grad_points + ((size_t)bi * c + ci) * m;
    unsafeAtomicAdd(row + i0, g * w0);
    unsafeAtomicAdd(row + i1, g * w1);
    unsafeAtomicAdd(row + i2, g * w2);
  }
}

}  // namespace

extern "C" {

int u3d_furthest_point_sampling(int b, int n, int m, const float* points, float* temp, int32_t* idx, void* stream) {
  if (b < 0 || n < 0 || m < 0) return 1;
  if (b == 0 || m == 0) return 0;
  if (n <= 0 || !points || !idx) return 1;
  int lg = 0;
  {  // opt_n_threads(n): 2^floor(log2 n) capped at 1024 (cuda_utils.h:10-14, evaluated like the reference, in double)
    const int pow_2 = (int)(log((double)n) / log(2.0));
    lg = pow_2 > 10 ? 10 : (pow_2 < 0 ? 0 : pow_2);
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)n * 3 * sizeof(float);
#define FPS(T, P) hipLaunchKernelGGL((fps_kernel<T, P>), dim3(b), dim3(T), lds, s, n, m, lg, g_contraction, points, idx)
  // small clouds: 4 waves keep the per-sample dependency chain short; larger ones spread over 16 waves
  if (n <= 256) FPS(256, 1);
  else if (n <= 512) FPS(256, 2);
  else if (n <= 1024) FPS(256, 4);
  else if (n <= 2048) FPS(1024, 2);
  else if (n <= 4096) FPS(1024, 4);
  else if (n <= 8192) FPS(1024, 8);
  else {
    if (!temp) return 1;
    hipLaunchKernelGGL(fps_kernel_large, dim3(b), dim3(1024), 0, s, n, m, lg, g_contraction, points, temp, idx);
  }
#undef FPS
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int32_t* idx,
                   void* stream) {
  if (b < 0 || n < 0 || m < 0 || nsample < 0) return 1;
  if (b == 0 || m == 0 || nsample == 0) return 0;
  if (!new_xyz || !xyz || !idx) return 1;
  const int total = b * m;
  hipLaunchKernelGGL(ball_query_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, n, m, total, radius * radius,
                     nsample, g_contraction, new_xyz, xyz, idx);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_group_points(int b, int c, int n, int npoints, int nsample, const float* points, const int32_t* idx, float* out,
                     void* stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return 1;
  if (b == 0 || c == 0 || npoints * nsample == 0) return 0;
  if (!points || !idx || !out) return 1;
  hipLaunchKernelGGL(group_points_kernel, dim3((npoints * nsample + 255) / 256, c, b), dim3(256), 0, (hipStream_t)stream, c, n,
                     npoints, nsample, points, idx, out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out, const int32_t* idx,
                          float* grad_points, void* stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return 1;
  if (b == 0 || c == 0 || npoints * nsample == 0) return 0;
  if (!grad_out || !idx || !grad_points) return 1;
  hipLaunchKernelGGL(group_points_grad_kernel, dim3((npoints * nsample + 255) / 256, c, b), dim3(256), 0, (hipStream_t)stream,
                     c, n, npoints, nsample, grad_out, idx, grad_points);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

// gather == group with nsample = 1
int u3d_gather_points(int b, int c, int n, int npoints, const float* points, const int32_t* idx, float* out, void* stream) {
  return u3d_group_points(b, c, n, npoints, 1, points, idx, out, stream);
}

int u3d_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out, const int32_t* idx, float* grad_points,
                           void* stream) {
  return u3d_group_points_grad(b, c, n, npoints, 1, grad_out, idx, grad_points, stream);
}

int u3d_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2, int32_t* idx, void* stream) {
  if (b < 0 || n < 0 || m < 0) return 1;
  if (b == 0 || n == 0) return 0;
  if (b > 65535) return 1;
  if (!unknown || (m > 0 && !known) || !dist2 || !idx) return 1;
  hipLaunchKernelGGL(three_nn_kernel, dim3((n + NN_THREADS - 1) / NN_THREADS, b), dim3(NN_THREADS), 0, (hipStream_t)stream, n, m, g_contraction, unknown,
                     known, dist2, idx);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_three_interpolate(int b, int c, int m, int n, const float* points, const int32_t* idx, const float* weight, float* out,
                          void* stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return 1;
  if (b == 0 || c == 0 || n == 0) return 0;
  if (b > 65535 || (c + IC_CH - 1) / IC_CH > 65535) return 1;
  if (!points || !idx || !weight || !out) return 1;
  hipLaunchKernelGGL(three_interpolate_kernel, dim3((n + 255) / 256, (c + IC_CH - 1) / IC_CH, b), dim3(256), 0, (hipStream_t)stream, c, m,
                     n, g_contraction, points, idx, weight, out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int32_t* idx, const float* weight,
                               float* grad_points, void* stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return 1;
  if (b == 0 || c == 0 || n == 0) return 0;
  if (b > 65535 || (c + IC_CH - 1) / IC_CH > 65535) return 1;
  if (!grad_out || !idx || !weight || !grad_points) return 1;
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3((n + 255) / 256, (c + IC_CH - 1) / IC_CH, b), dim3(256), 0, (hipStream_t)stream,
                     c, n, m, grad_out, idx, weight, grad_points);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_pointops_set_contraction(int mode) {
  if (mode != U3D_PO_FMA_LLVM && mode != U3D_PO_FMA_CHAIN && mode != U3D_PO_NO_FMA) return 1;
  g_contraction = mode;
  return 0;
}
int u3d_pointops_get_contraction(void) { return g_contraction; }

}  // extern "C"
