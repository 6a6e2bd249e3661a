// Point sampling / grouping operators for gfx950; semantics: oracle/pointops_oracle.c, which restates
// openpoints/cpp/pointnet2_batch/src/{sampling,ball_query,group_points}_gpu.cu.
//
//  * furthest point sampling: one workgroup per cloud, coordinates staged once in LDS, up to 8 points per thread with their running
//    minimum distances in REGISTERS (the reference round-trips a (B,N) temp array through global memory every iteration); the
//    selection's critical path is described at fps_kernel.  The tie key reproduces the order in which the reference's shared-memory
//    tree resolves equal distances for ITS block size, so the selection is bit-identical.
//  * ball query: one WAVE per query (the reference: one thread per query scanning all N points serially): 64 candidates
//    per step, ballot + popcount prefix keeps index order, early exit at nsample.
//  * group / gather: four outputs per thread, indices in registers across 8 channels; gradients: channel rows accumulated in LDS.
//  * three_nn: a DPP quad per query over an LDS-staged known cloud; three_interpolate (+grad): indices / weights loaded once per point.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "unipre3d_pointops.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// workgroup sizes of the furthest-point-sampling kernel (experiments: make EXTRA=-DU3D_FPS_BIG_THREADS=512 LIBDIR=../lib_x)
#ifndef U3D_FPS_SMALL_THREADS
#define U3D_FPS_SMALL_THREADS 256    // clouds of <= 1024 points
#endif
#ifndef U3D_FPS_PK
#define U3D_FPS_PK 1                 // two points per v_pk_*_f32 instruction in the distance update
#endif
#ifndef U3D_FPS_BIG_THREADS
#define U3D_FPS_BIG_THREADS 1024     // clouds of <= 8192 points
#endif

// `a*a + b*b + c*c` and `w0*p0 + w1*p1 + w2*p2` as the reference's kernels write them (sampling_gpu.cu:140, ball_query_gpu.cu:38,
// interpolate_gpu.cu:40, :103) are compiled by nvcc with -fmad=true: WHICH product is rounded on its own before the two fused
// multiply-adds decides ties, and FPS is a chaotic integer selection -- so the contraction is a call-time mode
// (u3d_pointops_set_contraction; a template parameter of every kernel, so the hot loops carry no mode branch), bit-exact against
// oracle/pointops_oracle.c in every mode:
//   U3D_PO_FMA_LLVM (default)  fma(c, c, fma(a, a, b*b))  what LLVM's DAG combiner -- NVVM is LLVM -- makes of fadd(fadd(fmul, fmul), fmul):
//                              the FIRST operand's multiply is folded into the inner add, the second product stays a rounded fmul
//   U3D_PO_FMA_CHAIN           fma(c, c, fma(b, b, a*a))  rounds 1-3's reading (left-to-right chain, first product rounded)
//   U3D_PO_NO_FMA              ((a*a + b*b) + c*c) with every product rounded: -fmad=false
template <int CM>
__device__ __forceinline__ float sum3(float a0, float a1, float b0, float b1, float c0, float c1) {   // a0*a1 + b0*b1 + c0*c1
#pragma clang fp contract(off)   // (only the fmaf calls below fuse; HIP's __fmul_rn / __fadd_rn are plain operators the optimiser may contract)
  if (CM == U3D_PO_FMA_LLVM) return fmaf(c0, c1, fmaf(a0, a1, b0 * b1));
  if (CM == U3D_PO_FMA_CHAIN) return fmaf(c0, c1, fmaf(b0, b1, a0 * a1));
  return (a0 * a1 + b0 * b1) + c0 * c1;
}
// the same sum for TWO points at once: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 round each half exactly like the scalar forms
template <int CM>
__device__ __forceinline__ f32x2 sum3_pk(f32x2 a, f32x2 b, f32x2 c) {   // a*a + b*b + c*c
#pragma clang fp contract(off)
  if (CM == U3D_PO_FMA_LLVM) return __builtin_elementwise_fma(c, c, __builtin_elementwise_fma(a, a, b * b));
  if (CM == U3D_PO_FMA_CHAIN) return __builtin_elementwise_fma(c, c, __builtin_elementwise_fma(b, b, a * a));
  return (a * a + b * b) + c * c;
}
template <int CM>
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = bx - ax, dy = by - ay, dz = bz - az;
  return sum3<CM>(dx, dx, dy, dy, dz, dz);
}
bool g_interp_lds = getenv("U3D_INTERP_GLOBAL") == nullptr;   // (experiment switch: U3D_INTERP_GLOBAL=1 keeps round 5's global-gather kernel)
int g_contraction = U3D_PO_FMA_LLVM;   // host: mode of the launches that follow (process-wide, like a build flag of the reference)

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long w = __shfl_xor(v, o);
    v = w > v ? w : v;
  }
  return v;
}

// maximum of a 32-bit key over the wave, returned in a scalar register: four DPP steps leave every lane with its 16-lane row's maximum
// (v_max_u32 with a DPP operand: one instruction per step), the four row results are read into SGPRs and combined on the scalar unit
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_max_step(uint32_t v) {
  const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
  return o > v ? o : v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
  v = dpp_max_step<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_max_step<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_max_step<0x141>(v);   // row_half_mirror
  v = dpp_max_step<0x140>(v);   // row_mirror
  uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
#pragma unroll
  for (int q = 1; q < 4; ++q) {
    const uint32_t u = (uint32_t)__builtin_amdgcn_readlane((int)v, q * 16);
    r = u > r ? u : r;
  }
  return r;
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

// Furthest point sampling is m - 1 DEPENDENT selections, so what is optimised is the length of one selection's critical path (round 5;
// rounds 1-4: 0.7 - 2.7 us per selection, a (distance, key) pair carried through every compare and a mode branch in the loop):
//   * squared distances are >= +0, so their IEEE bit patterns order like unsigned integers: the running minimum is v_min_u32, the
//     per-lane and per-wave maxima v_max3_u32 / v_max_u32 with a DPP operand -- no (distance, tie key) pair in the scan at all;
//   * two points per instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) for the distance itself;
//   * only once the wave's maximum is known does the wave ask WHICH of its points holds it: one v_cmp_eq per register slot gives a
//     lane mask in SGPRs, and the tie key of every set bit is computed from (wave, lane, slot) on the SCALAR unit (usually one bit);
//   * the waves' (maximum, inverted tie key) pairs meet in ONE ds_max_u64 per wave on a triple-buffered LDS word (slot j mod 3 is
//     used by selection j and cleared during selection j - 1), one barrier per selection, every lane reads the winner back.
// The tie key reproduces the order in which the reference's shared-memory tree resolves equal distances for ITS block size.
template <int FPS_THREADS, int PPT, int CM>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(int n, int m, int lg, const float* __restrict__ dataset,
                                                          int32_t* __restrict__ idxs) {
  extern __shared__ __attribute__((aligned(16))) float s_xyz[];   // [n][3]
  __shared__ unsigned long long s_best[3];
  const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave_base = (uint32_t)__builtin_amdgcn_readfirstlane(tid & ~63);
  const float* ds = dataset + (size_t)bi * n * 3;
  int32_t* out = idxs + (size_t)bi * m;
  for (int i = tid; i < n * 3; i += FPS_THREADS) s_xyz[i] = ds[i];
  if (tid < 3) s_best[tid] = 0ull;
  __syncthreads();
  const uint32_t bs = 1u << lg;
  float x[PPT], y[PPT], z[PPT];
  uint32_t t[PPT];                                   // running minimum squared distance, as its bit pattern
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * FPS_THREADS;
    const bool v = k < n;
    x[i] = v ? s_xyz[k * 3] : 0.f; y[i] = v ? s_xyz[k * 3 + 1] : 0.f; z[i] = v ? s_xyz[k * 3 + 2] : 0.f;
    t[i] = v ? __float_as_uint(1e10f) : 0u;          // padding slots: distance 0, and never a candidate below (k >= n)
  }
  int old = 0, cur = 1, nxt = 2;
  if (tid == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float ox = s_xyz[old * 3], oy = s_xyz[old * 3 + 1], oz = s_xyz[old * 3 + 2];
    uint32_t lmax = 0u;
    if constexpr (PPT == 1 || PPT > 4 || !U3D_FPS_PK) {   // (packed math measured 4 % ahead at 4 points per lane, 3 % behind at 8)
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        t[i] = min(t[i], __float_as_uint(dist2<CM>(ox, oy, oz, x[i], y[i], z[i])));
        lmax = max(lmax, t[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < PPT; i += 2) {
#pragma clang fp contract(off)
        const f32x2 dx = f32x2{x[i], x[i + 1]} - ox, dy = f32x2{y[i], y[i + 1]} - oy, dz = f32x2{z[i], z[i + 1]} - oz;
        const f32x2 d = sum3_pk<CM>(dx, dy, dz);
        t[i] = min(t[i], __float_as_uint(d.x));
        t[i + 1] = min(t[i + 1], __float_as_uint(d.y));
        lmax = max(lmax, max(t[i], t[i + 1]));
      }
    }
    const uint32_t wmax = wave_max_u32(lmax);
    uint32_t btk = 0u;                               // largest inverted tie key among this wave's points at distance wmax
    // (round 6 tried the single-wave kernel's vector count here -- one tie key on the scalar unit when exactly one point of the wave is at its maximum:
    //  +3 % at 2048 points, +4.5 % at 8192: with several waves per SIMD the per-slot ballots hide behind the other waves, the extra VALU work does not)
    {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {                // (all compares first + one combined test measured slower: 61 vs 56 us at 1024 -> 128)
        unsigned long long mk = __ballot(t[i] == wmax);
        while (mk) {
          const uint32_t k = wave_base + (uint32_t)__builtin_ctzll(mk) + (uint32_t)(i * FPS_THREADS);
          mk &= mk - 1ull;
          if (k < (uint32_t)n) {
            const uint32_t tk = ~fps_tie_key(k, lg, bs);
            btk = tk > btk ? tk : btk;
          }
        }
      }
    }
    if (lane == 0) atomicMax(&s_best[cur], ((unsigned long long)wmax << 32) | btk);
    if (tid == 0) s_best[nxt] = 0ull;
    __syncthreads();
    const unsigned long long w = s_best[cur];
    old = fps_decode(~(uint32_t)w, lg);
    if (tid == 0) out[j] = old;
    cur = nxt;
    nxt = nxt == 2 ? 0 : nxt + 1;
  }
}

// (Round 6 also rebuilt the multi-wave selection around `tools/ub/fps_phases.hip`'s stamps -- centre read from LDS ~200 clocks, scan ~210, wave maximum ~150,
//  holder search ~350, atomic ~190, barrier ~95 (660 with 16 waves), read back + decode ~260 --: per-lane running (maximum, slot, coordinates) in the scan so that one
//  ballot and v_readlanes replace the per-slot search, every wave's candidate coordinates posted beside its key so that the next centre needs ONE LDS round trip after
//  the barrier instead of two.  Bit-identical selections, and SLOWER: 1024 points 0.445 -> 0.578 us per selection, 2048: 0.498 -> 0.616, 8192: 0.753 -> 1.236.  The
//  selects ride on the scan's dependent chain and the scalar lane reads serialise behind the wave maximum; the stamped phases overstate what the search costs.)
// Round 6: ONE WAVE per cloud (n <= 64 * PPT).  A selection of the multi-wave kernel above is ~45 instructions and ~950 clocks: most of it the
// waves meeting -- an LDS atomic, a barrier, the read back -- not arithmetic.  A lone wave needs none of that: the wave maximum and the winner's tie key are
// already wave-uniform (SGPRs) after the DPP reduction and the per-slot ballots, so the next centre is decoded on the scalar unit and its coordinates come
// from one broadcast LDS read.  The price is PPT points per lane in the distance scan (independent chains: they pipeline), which is why more points per lane
// LOST in the multi-wave form (128 threads: +18 %) and wins here.  Same distances, same tie order: bit-identical selections.
template <int PPT, int CM>
__global__ __launch_bounds__(64) void fps_wave_kernel(int n, int m, int lg, const float* __restrict__ dataset, int32_t* __restrict__ idxs) {
  extern __shared__ __attribute__((aligned(16))) float s_xyz[];   // [n][3]
  const int bi = blockIdx.x, lane = threadIdx.x;
  const float* ds = dataset + (size_t)bi * n * 3;
  int32_t* out = idxs + (size_t)bi * m;
  for (int i = lane; i < n * 3; i += 64) s_xyz[i] = ds[i];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const uint32_t bs = 1u << lg;
  float x[PPT], y[PPT], z[PPT];
  uint32_t t[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = lane + i * 64;
    const bool v = k < n;
    x[i] = v ? s_xyz[k * 3] : 0.f; y[i] = v ? s_xyz[k * 3 + 1] : 0.f; z[i] = v ? s_xyz[k * 3 + 2] : 0.f;
    t[i] = v ? __float_as_uint(1e10f) : 0u;          // padding slots: distance 0, and never a candidate below (k >= n)
  }
  int old = 0;
  if (lane == 0) out[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float ox = s_xyz[old * 3], oy = s_xyz[old * 3 + 1], oz = s_xyz[old * 3 + 2];
    uint32_t lmax = 0u;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      t[i] = min(t[i], __float_as_uint(dist2<CM>(ox, oy, oz, x[i], y[i], z[i])));
      lmax = max(lmax, t[i]);
    }
    const uint32_t wmax = wave_max_u32(lmax);
    // WHICH point holds the maximum.  Equal distances are rare, so the common case is decided on the vector unit: every lane counts its slots at
    // the maximum and remembers one of them (compare + select + add per slot, pipelined -- a scalar ballot-and-branch per slot costs ~56 clocks of a lone
    // wave's time each); one ballot then says whether exactly ONE point of the cloud is at the maximum, and if so its index is lane + 64 slot -- no
    // tie key at all.  Only a genuine tie walks the slots' ballots for the reference's tie order.
    uint32_t cnt = 0u, slot = 0u;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const bool e = t[i] == wmax;
      cnt += e ? 1u : 0u;
      slot = e ? (uint32_t)i : slot;
    }
    const unsigned long long has = __ballot(cnt != 0u), many = __ballot(cnt > 1u);
    const uint32_t L = (uint32_t)__builtin_ctzll(has);
    const uint32_t k1 = L + 64u * (uint32_t)__builtin_amdgcn_readlane((int)slot, (int)L);
    if (many == 0ull && (has & (has - 1ull)) == 0ull && k1 < (uint32_t)n) {
      old = (int)k1;
    } else {
      uint32_t btk = 0u;                             // largest inverted tie key among the points at distance wmax
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        unsigned long long mk = __ballot(t[i] == wmax);
        while (mk) {
          const uint32_t k = (uint32_t)__builtin_ctzll(mk) + (uint32_t)(i * 64);
          mk &= mk - 1ull;
          if (k < (uint32_t)n) {
            const uint32_t tk = ~fps_tie_key(k, lg, bs);
            btk = tk > btk ? tk : btk;
          }
        }
      }
      old = fps_decode(~btk, lg);
    }
    if (lane == 0) out[j] = old;
  }
}

// generic path for clouds larger than FPS_THREADS*8 points: minimum distances in global scratch
template <int CM>
__global__ __launch_bounds__(1024) void fps_kernel_large(int n, int m, int lg, const float* __restrict__ dataset,
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
      const float d = dist2<CM>(ox, oy, oz, ds[k * 3], ds[k * 3 + 1], ds[k * 3 + 2]);
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

template <int CM>
__global__ __launch_bounds__(256) void ball_query_kernel(int n, int m, int total_q, float radius2, int nsample,
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
      in = sum3<CM>(dx, dx, dy, dy, dz, dz) < radius2;
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

// group / gather: out[b][c][e] = points[b][c][idx[b][e]].  A thread owns FOUR consecutive outputs and walks GP_CH channels with their
// indices in registers (one 16-byte index load, then per channel four gathers from a row that sits in cache and one 16-byte store);
// the scalar kernel serves shapes / pointers the vector form cannot (total % 4 != 0, unaligned bases).
constexpr int GP_CH = 8;
__global__ __launch_bounds__(256) void group_points_vec_kernel(int c, int n, int total, const float* __restrict__ points,
                                                               const int32_t* __restrict__ idx, float* __restrict__ out) {
  // grid = (channel blocks, output chunks, clouds): workgroups are dealt to the 8 XCDs round-robin in linear order, so the workgroups that gather
  // from the SAME channel rows (same cloud and channel block, different output chunk) sit a whole row of channel blocks apart -- on one XCD,
  // one L2 -- whenever that row's length is a multiple of 8 (384 channels: 48 blocks); with the chunks in x every XCD fetched the rows itself
  // (PMC: 404 MB moved for 252 MB of algorithmic bytes)
  const int bi = blockIdx.z, c0 = blockIdx.x * GP_CH;
  const int e0 = (blockIdx.y * 256 + threadIdx.x) * 4;
  if (e0 >= total) return;
  const int4 id = *reinterpret_cast<const int4*>(idx + (size_t)bi * total + e0);
  const int c1 = min(c, c0 + GP_CH);
#pragma unroll 4
  for (int ci = c0; ci < c1; ++ci) {
    const float* row = points + ((size_t)bi * c + ci) * n;
    *reinterpret_cast<float4*>(out + ((size_t)bi * c + ci) * total + e0) = make_float4(row[id.x], row[id.y], row[id.z], row[id.w]);
  }
}
__global__ void group_points_kernel(int c, int n, int total, const float* __restrict__ points, const int32_t* __restrict__ idx,
                                    float* __restrict__ out) {
  const int bi = blockIdx.z, ci = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int src = idx[(size_t)bi * total + e];
  out[((size_t)bi * c + ci) * total + e] = points[((size_t)bi * c + ci) * n + src];
}

// Float accumulation into LDS.  gfx950's ds_add_f32 retires ~0.3 lane-updates per clock and CU whatever the addresses (tools/ub/ldsatomic.hip:
// 204 G/s chip-wide; integer LDS atomics run at 7 - 14 per clock), a compare-and-swap on the word at 2.3 - 4.6 -- unless many lanes of a wave
// hit ONE address, where the retry loop collapses (0.06).  So: one compare-and-swap attempt, and only the lanes that lost it (another lane of
// the wave, or another wave, touched the word in between) take the hardware float atomic.
__device__ __forceinline__ void lds_add_f32(float* addr, float v) {
  unsigned* w = reinterpret_cast<unsigned*>(addr);
  const unsigned old = *w;
  if (atomicCAS(w, old, __float_as_uint(__uint_as_float(old) + v)) != old) unsafeAtomicAdd(addr, v);
}

// group / gather gradient: grad_points[b][c][idx[b][e]] += grad_out[b][c][e].  The reference (and rounds 1-4 here) issue one GLOBAL float
// atomic per element; ball-query index sets repeat their first hit as padding, so thousands of them land on one address (3.0 ms at
// the transformer shape).  Here a workgroup owns `cb` channel rows of one cloud in LDS ([cb][n] floats), accumulates with LDS atomics
// (an index is loaded once and reused by the cb channels) and adds the finished rows to grad_points with plain coalesced
// read-modify-writes: every output element belongs to exactly one workgroup.
template <int CB>
__global__ __launch_bounds__(256) void group_points_grad_lds_kernel(int c, int n, int total, const float* __restrict__ grad_out,
                                                                    const int32_t* __restrict__ idx, float* __restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float s_acc[];   // [CB][n]
  const int bi = blockIdx.y, c0 = blockIdx.x * CB;
  const int nc = min(CB, c - c0);
  for (int i = threadIdx.x; i < nc * n; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  const int32_t* ip = idx + (size_t)bi * total;
  const float* go = grad_out + ((size_t)bi * c + c0) * total;
  const int lane = threadIdx.x & 63;
  // (software pipeline: the loads of step s + 1 are issued before step s is processed, so a wave's own HBM latency overlaps its scan and its
  //  LDS updates instead of waiting for another wave to fill the gap)
  bool active = (int)threadIdx.x < total;
  int dst = active ? ip[threadIdx.x] : 0;
  float g[CB];
#pragma unroll
  for (int cc = 0; cc < CB; ++cc) g[cc] = (active && cc < nc) ? go[(size_t)cc * total + threadIdx.x] : 0.f;
  for (int e0 = 0; e0 < total; e0 += 256) {         // (wave-uniform trip count: the run merge below works across the wave)
    const int en = e0 + 256 + threadIdx.x;
    const bool an = en < total;
    const int dstn = an ? ip[en] : 0;
    float gn[CB];
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) gn[cc] = (an && cc < nc) ? go[(size_t)cc * total + en] : 0.f;
    // Ball-query index sets pad every group with copies of its first hit: runs of EQUAL CONSECUTIVE destinations, which would meet in one LDS
    // word (the compare-and-swap collapses there).  A wave that holds such a run sums it first -- segmented inclusive scan over the run, the
    // run's last lane keeps the total -- and only the tails update LDS.  Waves without adjacent duplicates (gather, three_nn-style indices)
    // skip this.  Round 6: the scan runs on DPP (row_shr 1/2/4/8 inside the 16-lane rows, then the two GFX9 row broadcasts) instead of
    // 6 ds_bpermute round trips per channel, the run structure comes from one ballot (run start = highest head bit at or below the lane), and
    // the per-step conditions are 0/1 multipliers of a v_fmac_f32_dpp: ONE VALU instruction per step and channel (it was shuffle + compare +
    // add: ~55 LDS operations and ~200 VALU per step of 8 channels, which -- not HBM -- bounded the kernel at 99 us).
    const int key = active ? dst : -1 - lane;
    const int prev = __builtin_amdgcn_update_dpp(key, key, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    const bool head = lane == 0 || prev != key;
    const unsigned long long hm = __ballot(head);
    bool tail = true;
    if (hm != ~0ull) {
      const int rs = 63 - __clzll(hm & (~0ull >> (63 - lane)));          // first lane of this lane's run
      const int lo = max(rs, lane & ~15);
      const float m1 = lane - 1 >= lo ? 1.f : 0.f, m2 = lane - 2 >= lo ? 1.f : 0.f, m4 = lane - 4 >= lo ? 1.f : 0.f, m8 = lane - 8 >= lo ? 1.f : 0.f;
      const float mb15 = ((lane & 16) && rs < (lane & ~15)) ? 1.f : 0.f;  // rows 1 and 3: the run began in an earlier row
      const float mb31 = (lane >= 32 && rs < 32) ? 1.f : 0.f;             // rows 2 and 3: ... before lane 32
      // 0 x inf = NaN would leak a non-finite value into a NEIGHBOURING run: such waves take the select form
      float chk = 0.f;
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) chk += fabsf(g[cc]);
      if (__ballot(!(chk < __builtin_inff())) == 0ull) {
#pragma unroll
        for (int cc = 0; cc < CB; ++cc)
          asm volatile("s_nop 1\n\t"
                       "v_fmac_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_dpp %0, %0, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_dpp %0, %0, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_dpp %0, %0, %4 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_dpp %0, %0, %5 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                       "s_nop 1\n\t"
                       "v_fmac_f32_dpp %0, %0, %6 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                       "s_nop 1"
                       : "+v"(g[cc])
                       : "v"(m1), "v"(m2), "v"(m4), "v"(m8), "v"(mb15), "v"(mb31));
      } else {
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) {
#pragma unroll
          for (int st = 1; st < 64; st <<= 1) {
            const float o = __shfl_up(g[cc], st);
            if (lane - st >= rs) g[cc] += o;
          }
        }
      }
      tail = lane == 63 || ((hm >> (lane + 1)) & 1ull);
    }
    if (active && tail) {
#pragma unroll
      for (int cc = 0; cc < CB; ++cc)
        if (cc < nc) lds_add_f32(&s_acc[cc * n + dst], g[cc]);
    }
    active = an; dst = dstn;
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) g[cc] = gn[cc];
  }
  __syncthreads();
  float* gp = grad_points + ((size_t)bi * c + c0) * n;
  const int len = nc * n;
  for (int i0 = threadIdx.x; i0 < len; i0 += 256 * 4) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = i0 + q * 256 < len ? gp[i0 + q * 256] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (i0 + q * 256 < len) gp[i0 + q * 256] = v[q] + s_acc[i0 + q * 256];
  }
}
__global__ void group_points_grad_kernel(int c, int n, int total, const float* __restrict__ grad_out,
                                         const int32_t* __restrict__ idx, float* __restrict__ grad_points) {
  const int bi = blockIdx.z, ci = blockIdx.y;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int dst = idx[(size_t)bi * total + e];
  unsafeAtomicAdd(&grad_points[((size_t)bi * c + ci) * n + dst], grad_out[((size_t)bi * c + ci) * total + e]);
}

// ---- three_nn / three_interpolate (+grad): interpolate_gpu.cu:16-140 ------------------------------------------------------
// three_nn: the reference gives every query a thread that streams all m known points from global memory.  Here a workgroup
// stages the known cloud through LDS in tiles and FOUR lanes (a DPP quad) share one query: lane s of the quad scans the known points
// s, s + 4, s + 8, ... of the tile (the quad reads 12 consecutive LDS words per step: no bank conflict, 16 quads broadcast), keeps
// its running three best in registers behind a single `d < best3` guard, and the four sorted triples are merged at the end with two
// quad_perm exchanges.  The reference's scan in index order with strict `<` insertions yields the three smallest (distance, index)
// pairs in lexicographic order; the merge compares exactly that pair, so any partition of the indices gives the same answer --
// bit-exact, ties to the lower index; an unfilled slot keeps the reference's initial 1e40 (stored as +inf) / index 0.
constexpr int NN_THREADS = 256, NN_SPLIT = 4, NN_Q = NN_THREADS / NN_SPLIT;
constexpr int NN_TILE = 2048;   // known points per LDS tile: 24 KB
struct Best3 { float d1, d2, d3; int i1, i2, i3; };
__device__ __forceinline__ void nn_insert_lex(Best3& b, float d, int k) {   // (d, k) into the sorted triple, lexicographic order
  const bool l1 = d < b.d1 || (d == b.d1 && k < b.i1);
  const bool l2 = d < b.d2 || (d == b.d2 && k < b.i2);
  const bool l3 = d < b.d3 || (d == b.d3 && k < b.i3);
  b.d3 = l2 ? b.d2 : (l3 ? d : b.d3); b.i3 = l2 ? b.i2 : (l3 ? k : b.i3);
  b.d2 = l1 ? b.d1 : (l2 ? d : b.d2); b.i2 = l1 ? b.i1 : (l2 ? k : b.i2);
  b.d1 = l1 ? d : b.d1;               b.i1 = l1 ? k : b.i1;
}
template <int CTRL>
__device__ __forceinline__ void nn_merge_step(Best3& b) {
  const auto xf = [](float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); };
  const auto xi = [](int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); };
  const float o1 = xf(b.d1), o2 = xf(b.d2), o3 = xf(b.d3);
  const int j1 = xi(b.i1), j2 = xi(b.i2), j3 = xi(b.i3);
  nn_insert_lex(b, o1, j1);
  nn_insert_lex(b, o2, j2);
  nn_insert_lex(b, o3, j3);
}
template <int CM>
__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(int n, int m, const float* __restrict__ unknown,
                                                              const float* __restrict__ known, float* __restrict__ dist2,
                                                              int32_t* __restrict__ idx) {
  __shared__ float s_k[NN_TILE * 3];
  const int bi = blockIdx.y;
  const int pt = blockIdx.x * NN_Q + (threadIdx.x >> 2), sub = threadIdx.x & 3;
  const bool live = pt < n;
  const float* u = unknown + ((size_t)bi * n + (live ? pt : 0)) * 3;
  const float ux = u[0], uy = u[1], uz = u[2];
  const float* kb = known + (size_t)bi * m * 3;
  const float inf = __builtin_inff();                  // (float)1e40: what the reference's `double best = 1e40` becomes when stored
  Best3 b{inf, inf, inf, 0, 0, 0};
  for (int k0 = 0; k0 < m; k0 += NN_TILE) {
    const int cnt = min(NN_TILE, m - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < cnt * 3; e += NN_THREADS) s_k[e] = kb[(size_t)k0 * 3 + e];
    __syncthreads();
    for (int j = sub; j < cnt; j += NN_SPLIT) {
      const float dx = ux - s_k[j * 3], dy = uy - s_k[j * 3 + 1], dz = uz - s_k[j * 3 + 2];
      const float d = sum3<CM>(dx, dx, dy, dy, dz, dz);   // (ux-x)^2 + (uy-y)^2 + (uz-z)^2 in the selected contraction
      if (d < b.d3) {                                     // ascending indices within a lane: the reference's strict `<` insertions
        const int k = k0 + j;
        const bool l1 = d < b.d1, l2 = d < b.d2;
        b.d3 = l2 ? b.d2 : d;                 b.i3 = l2 ? b.i2 : k;
        b.d2 = l1 ? b.d1 : (l2 ? d : b.d2);   b.i2 = l1 ? b.i1 : (l2 ? k : b.i2);
        b.d1 = l1 ? d : b.d1;                 b.i1 = l1 ? k : b.i1;
      }
    }
  }
  nn_merge_step<0xB1>(b);   // quad_perm [1,0,3,2]
  nn_merge_step<0x4E>(b);   // quad_perm [2,3,0,1]
  if (live && sub == 0) {
    float* od = dist2 + ((size_t)bi * n + pt) * 3;
    int32_t* oi = idx + ((size_t)bi * n + pt) * 3;
    od[0] = b.d1; od[1] = b.d2; od[2] = b.d3;
    oi[0] = b.i1; oi[1] = b.i2; oi[2] = b.i3;
  }
}

// three_interpolate: the reference launches one thread per (batch, channel, point), so every channel re-reads the point's three
// indices and weights.  Here a lane owns a point, loads them once and walks IC_CH channels: three gathers from the channel's
// row of m values (cache-resident) and one coalesced store per channel.  out = w0 p[i0] + w1 p[i1] + w2 p[i2] in the selected contraction.
constexpr int IC_CH = 8;
template <int CM>
__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n, const float* __restrict__ points,
                                                                const int32_t* __restrict__ idx, const float* __restrict__ weight,
                                                                float* __restrict__ out) {
  const int bi = blockIdx.z, c0 = blockIdx.y * IC_CH;
  const int pt = blockIdx.x * 256 + threadIdx.x;
  if (pt >= n) return;
  const int32_t* ip = idx + ((size_t)bi * n + pt) * 3;
  const float* wp = weight + ((size_t)bi * n + pt) * 3;
  const int i0 = ip[0], i1 = ip[1], i2 = ip[2];
  const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
  const int nc = min(IC_CH, c - c0);
  float p0[IC_CH], p1[IC_CH], p2[IC_CH];              // every gather of the IC_CH channels in flight before the first store
#pragma unroll
  for (int cc = 0; cc < IC_CH; ++cc) {
    const float* row = points + ((size_t)bi * c + c0 + (cc < nc ? cc : 0)) * m;
    p0[cc] = row[i0]; p1[cc] = row[i1]; p2[cc] = row[i2];
  }
#pragma unroll
  for (int cc = 0; cc < IC_CH; ++cc)
    if (cc < nc) out[((size_t)bi * c + c0 + cc) * n + pt] = sum3<CM>(w0, p0[cc], w1, p1[cc], w2, p2[cc]);
}

// Round 6: the global-gather form above is bound by its gather LATENCY (PMC r05: 1.6 TB/s, no byte wasted): a wave's 24 gathers wait on its index
// loads, then its stores on the gathers, and 16 K such waves run in two rounds.  Here a workgroup stages IL_CB channel rows of the known
// cloud in LDS (they are consecutive in memory: one coalesced copy), and every thread interpolates IL_PT points from LDS -- index / weight
// loads of all its points in flight at once, LDS gathers (~50 clocks instead of an L2 round trip), coalesced stores.  Same sum3, bit-exact.
#ifndef U3D_IL_CB
#define U3D_IL_CB 4
#endif
#ifndef U3D_IL_PT
#define U3D_IL_PT 4
#endif
constexpr int IL_CB = U3D_IL_CB, IL_PT = U3D_IL_PT;
template <int CM>
__global__ __launch_bounds__(256) void three_interpolate_lds_kernel(int c, int m, int n, const float* __restrict__ points,
                                                                    const int32_t* __restrict__ idx, const float* __restrict__ weight,
                                                                    float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float s_rows[];   // [IL_CB][m]
  const int bi = blockIdx.z, c0 = blockIdx.y * IL_CB;
  const int nc = min(IL_CB, c - c0);
  const int p0 = blockIdx.x * (256 * IL_PT) + threadIdx.x;
  int i0[IL_PT], i1[IL_PT], i2[IL_PT];
  float w0[IL_PT], w1[IL_PT], w2[IL_PT];
#pragma unroll
  for (int j = 0; j < IL_PT; ++j) {                 // (issued before the staging copy: both are in flight together)
    const int pt = min(p0 + j * 256, n - 1);
    const int32_t* ip = idx + ((size_t)bi * n + pt) * 3;
    const float* wp = weight + ((size_t)bi * n + pt) * 3;
    i0[j] = ip[0]; i1[j] = ip[1]; i2[j] = ip[2];
    w0[j] = wp[0]; w1[j] = wp[1]; w2[j] = wp[2];
  }
  const float* src = points + ((size_t)bi * c + c0) * m;
  const int len = nc * m;
  if (((uintptr_t)src & 15u) == 0 && (len & 3) == 0) {
    for (int i = threadIdx.x * 4; i < len; i += 1024) *reinterpret_cast<float4*>(s_rows + i) = *reinterpret_cast<const float4*>(src + i);
  } else {
    for (int i = threadIdx.x; i < len; i += 256) s_rows[i] = src[i];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < IL_PT; ++j) {
    const int pt = p0 + j * 256;
    if (pt < n) {
#pragma unroll
      for (int cc = 0; cc < IL_CB; ++cc)
        if (cc < nc) {
          const float* row = s_rows + cc * m;
          out[((size_t)bi * c + c0 + cc) * n + pt] = sum3<CM>(w0[j], row[i0[j]], w1[j], row[i1[j]], w2[j], row[i2[j]]);
        }
    }
  }
}

// gradient: grad_points[b][c][idx[b][i][k]] += grad_out[b][c][i] * weight[b][i][k].  Like the grouping gradient: `cb` channel rows of one
// cloud accumulate in LDS ([cb][m] floats, LDS atomics; a point's three indices and weights are loaded once for the cb channels) and are
// added to grad_points by coalesced read-modify-writes; the global-atomic form stays for known clouds that do not fit.
template <int CB>
__global__ __launch_bounds__(256) void three_interpolate_grad_lds_kernel(int c, int n, int m, const float* __restrict__ grad_out,
                                                                         const int32_t* __restrict__ idx, const float* __restrict__ weight,
                                                                         float* __restrict__ grad_points) {
  extern __shared__ __attribute__((aligned(16))) float s_acc[];   // [CB][m]
  const int bi = blockIdx.y, c0 = blockIdx.x * CB;
  const int nc = min(CB, c - c0);
  for (int i = threadIdx.x; i < nc * m; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  const float* go = grad_out + ((size_t)bi * c + c0) * n;
  for (int pt = threadIdx.x; pt < n; pt += 256) {
    const int32_t* ip = idx + ((size_t)bi * n + pt) * 3;
    const float* wp = weight + ((size_t)bi * n + pt) * 3;
    const int i0 = ip[0], i1 = ip[1], i2 = ip[2];
    const float w0 = wp[0], w1 = wp[1], w2 = wp[2];
    float g[CB];                                    // all loads of the step in flight before the first LDS atomic
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) g[cc] = cc < nc ? go[(size_t)cc * n + pt] : 0.f;
#pragma unroll
    for (int cc = 0; cc < CB; ++cc)
      if (cc < nc) {
        lds_add_f32(&s_acc[cc * m + i0], g[cc] * w0);
        lds_add_f32(&s_acc[cc * m + i1], g[cc] * w1);
        lds_add_f32(&s_acc[cc * m + i2], g[cc] * w2);
      }
  }
  __syncthreads();
  float* gp = grad_points + ((size_t)bi * c + c0) * m;
  const int len = nc * m;
  for (int i0 = threadIdx.x; i0 < len; i0 += 256 * 4) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = i0 + q * 256 < len ? gp[i0 + q * 256] : 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (i0 + q * 256 < len) gp[i0 + q * 256] = v[q] + s_acc[i0 + q * 256];
  }
}
// (Round 6 also built this gradient in GATHER form -- per-destination lists of the cloud's index set made in LDS by count / scan / placement with
//  integer atomics, then every thread walking the lists of its known points with CB gathers and multiply-adds per entry, no float atomics: bit-equal
//  to the tolerance, and 112 / 121 / 174 us with 8 / 4 / 16 channels per workgroup against 44 us for the accumulate form below: 52 KB of lists per
//  workgroup leave 3 workgroups per CU, every one of the 32 workgroups of a cloud rebuilds the same lists, and the lists' lengths diverge.)
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

// channel rows per workgroup of the LDS-accumulating gradient kernels: a power of two <= 16 such that the rows of `len` floats fit
// 64 KB of LDS and the launch has on the order of three thousand workgroups (b * c rows in all); 0: one row does not fit -> global atomics.
// (Round 6 sweep at the reference's shapes: grouping gradient 32 x 384 rows of 1024: 1 / 2 / 4 / 8 / 16 rows -> 87 / 72 / 70 / 85 / 129 us;
//  three_interpolate gradient 16 x 256 rows of 512: 48 / 44 / 49 / 50 / 72 us.  Fewer rows = more, smaller workgroups per CU (LDS-limited
//  residency: 8 rows of 1024 floats allow 5 workgroups per CU = 1280 slots for 1536 workgroups, i.e. a second round one fifth full),
//  more rows = fewer re-reads of the index array; rounds 1-5 aimed at a thousand workgroups.)
inline int lds_rows(int len, int b, int c) {
  const int fit = 16384 / (len > 0 ? len : 1);
  if (fit < 1) return 0;
  const long long want = ((long long)b * c + 3071) / 3072;
  int cb = 1;
  while (cb * 2 <= 16 && cb * 2 <= fit && cb * 2 <= want) cb *= 2;
  return cb;
}
inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// launch KERNEL<..., CM> for the process-wide contraction mode
#define U3D_PO_LAUNCH_CM(KERNEL, ...)                                                                 \
  do {                                                                                                \
    if (g_contraction == U3D_PO_FMA_LLVM) hipLaunchKernelGGL((KERNEL<U3D_PO_FMA_LLVM>), __VA_ARGS__);  \
    else if (g_contraction == U3D_PO_FMA_CHAIN) hipLaunchKernelGGL((KERNEL<U3D_PO_FMA_CHAIN>), __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<U3D_PO_NO_FMA>), __VA_ARGS__);                                    \
  } while (0)

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
  // (clouds beyond 5461 points stage more than 64 KB: opt in to the larger dynamic LDS size.  The attribute belongs to the (device, function)
  // pair, so it is set on every such launch -- a host-side table lookup -- and checked: if the runtime refuses it the launch goes to
  // fps_kernel_large when the caller brought `temp`, and fails loudly otherwise)
#define FPS_CM(T, P, CM)                                                                                                             \
  do {                                                                                                                               \
    if (lds > 65536 &&                                                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<T, P, CM>), hipFuncAttributeMaxDynamicSharedMemorySize,          \
                            8192 * 3 * (int)sizeof(float)) != hipSuccess) {                                                          \
      (void)hipGetLastError();                                                                                                       \
      big_lds_refused = true;                                                                                                        \
    } else {                                                                                                                         \
      hipLaunchKernelGGL((fps_kernel<T, P, CM>), dim3(b), dim3(T), lds, s, n, m, lg, points, idx);                                   \
    }                                                                                                                                \
  } while (0)
#define FPS(T, P)                                                    \
  do {                                                               \
    if (g_contraction == U3D_PO_FMA_LLVM) FPS_CM(T, P, U3D_PO_FMA_LLVM);        \
    else if (g_contraction == U3D_PO_FMA_CHAIN) FPS_CM(T, P, U3D_PO_FMA_CHAIN); \
    else FPS_CM(T, P, U3D_PO_NO_FMA);                                \
  } while (0)
  // small clouds: U3D_FPS_SMALL_THREADS / 64 waves keep the per-selection chain short; larger ones spread over U3D_FPS_BIG_THREADS / 64
  constexpr int ST = U3D_FPS_SMALL_THREADS, BT = U3D_FPS_BIG_THREADS;
  bool big_lds_refused = false;
  // one wave per cloud up to 512 points (measured, us per selection, one wave / multi-wave: 256 points 0.273 / 0.393, 512: 0.329 / 0.391, 1024: 0.46 - 0.485 /
  // 0.42 - 0.44, 2048: 0.98 / 0.50: a lone wave issues one VALU instruction per ~3 clocks, 11 per point and selection).  U3D_FPS_WAVE_MAX: experiment switch
  static const int wave_max = getenv("U3D_FPS_WAVE_MAX") ? atoi(getenv("U3D_FPS_WAVE_MAX")) : 512;
#define FPSW(P)                                                                                                        \
  do {                                                                                                                 \
    if (g_contraction == U3D_PO_FMA_LLVM) hipLaunchKernelGGL((fps_wave_kernel<P, U3D_PO_FMA_LLVM>), dim3(b), dim3(64), lds, s, n, m, lg, points, idx);        \
    else if (g_contraction == U3D_PO_FMA_CHAIN) hipLaunchKernelGGL((fps_wave_kernel<P, U3D_PO_FMA_CHAIN>), dim3(b), dim3(64), lds, s, n, m, lg, points, idx); \
    else hipLaunchKernelGGL((fps_wave_kernel<P, U3D_PO_NO_FMA>), dim3(b), dim3(64), lds, s, n, m, lg, points, idx);    \
  } while (0)
  if (n <= wave_max && n <= 2048) {
    if (n <= 128) FPSW(2);
    else if (n <= 256) FPSW(4);
    else if (n <= 512) FPSW(8);
    else if (n <= 1024) FPSW(16);
    else FPSW(32);
  } else
  if (n <= 256) FPS(256, 1);
  else if (n <= 512) FPS(ST, 512 / ST);
  else if (n <= 1024) FPS(ST, 1024 / ST);
  else if (n <= 2048) FPS(512, 4);                   // (8 waves measured 5 % ahead of 16 here, 9 % behind at 8192)
  else if (n <= 4096) FPS(BT, 4096 / BT);
  else if (n <= 8192) FPS(BT, 8192 / BT);
  else {
    if (!temp) return 1;
    U3D_PO_LAUNCH_CM(fps_kernel_large, dim3(b), dim3(1024), 0, s, n, m, lg, points, temp, idx);
  }
#undef FPS
#undef FPS_CM
#undef FPSW
  if (big_lds_refused) {
    if (!temp) return 3;
    U3D_PO_LAUNCH_CM(fps_kernel_large, dim3(b), dim3(1024), 0, s, n, m, lg, points, temp, idx);
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int32_t* idx,
                   void* stream) {
  if (b < 0 || n < 0 || m < 0 || nsample < 0) return 1;
  if (b == 0 || m == 0 || nsample == 0) return 0;
  if (!new_xyz || !xyz || !idx) return 1;
  const int total = b * m;
  U3D_PO_LAUNCH_CM(ball_query_kernel, dim3((total + 3) / 4), dim3(256), 0, (hipStream_t)stream, n, m, total, radius * radius, nsample,
                   new_xyz, xyz, idx);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_group_points(int b, int c, int n, int npoints, int nsample, const float* points, const int32_t* idx, float* out,
                     void* stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return 1;
  if (b == 0 || c == 0 || npoints * nsample == 0) return 0;
  if (b > 65535 || c > 65535) return 1;
  if (!points || !idx || !out) return 1;
  const int total = npoints * nsample;
  if (total % 4 == 0 && aligned16(idx) && aligned16(out) && (total / 4 + 255) / 256 <= 65535)
    hipLaunchKernelGGL(group_points_vec_kernel, dim3((c + GP_CH - 1) / GP_CH, (total / 4 + 255) / 256, b), dim3(256), 0, (hipStream_t)stream,
                       c, n, total, points, idx, out);
  else
    hipLaunchKernelGGL(group_points_kernel, dim3((total + 255) / 256, c, b), dim3(256), 0, (hipStream_t)stream, c, n, total, points, idx, out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_group_points_grad(int b, int c, int n, int npoints, int nsample, const float* grad_out, const int32_t* idx,
                          float* grad_points, void* stream) {
  if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return 1;
  if (b == 0 || c == 0 || npoints * nsample == 0) return 0;
  if (b > 65535 || c > 65535) return 1;
  if (!grad_out || !idx || !grad_points) return 1;
  const int total = npoints * nsample;
  int cb = lds_rows(n, b, c);
  static const int cb_env = getenv("U3D_GG_CB") ? atoi(getenv("U3D_GG_CB")) : 0;   // (experiment switch)
  if (cb_env > 0 && cb > 0 && (size_t)cb_env * n * sizeof(float) <= 65536) cb = cb_env;
#define GG(CB) hipLaunchKernelGGL((group_points_grad_lds_kernel<CB>), dim3((c + CB - 1) / CB, b), dim3(256), sizeof(float) * (size_t)CB * n, \
                                  (hipStream_t)stream, c, n, total, grad_out, idx, grad_points)
  if (cb == 16) GG(16);
  else if (cb == 8) GG(8);
  else if (cb == 4) GG(4);
  else if (cb == 2) GG(2);
  else if (cb == 1) GG(1);
#undef GG
  else
    hipLaunchKernelGGL(group_points_grad_kernel, dim3((total + 255) / 256, c, b), dim3(256), 0, (hipStream_t)stream, c, n, total, grad_out, idx,
                       grad_points);
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
  U3D_PO_LAUNCH_CM(three_nn_kernel, dim3((n + NN_Q - 1) / NN_Q, b), dim3(NN_THREADS), 0, (hipStream_t)stream, n, m, unknown, known, dist2, idx);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_three_interpolate(int b, int c, int m, int n, const float* points, const int32_t* idx, const float* weight, float* out,
                          void* stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return 1;
  if (b == 0 || c == 0 || n == 0) return 0;
  if (b > 65535 || (c + IC_CH - 1) / IC_CH > 65535) return 1;
  if (!points || !idx || !weight || !out) return 1;
  if (m > 0 && (size_t)m * IL_CB * sizeof(float) <= 32768 && (c + IL_CB - 1) / IL_CB <= 65535 && g_interp_lds) {
    U3D_PO_LAUNCH_CM(three_interpolate_lds_kernel, dim3((n + 256 * IL_PT - 1) / (256 * IL_PT), (c + IL_CB - 1) / IL_CB, b), dim3(256),
                     sizeof(float) * (size_t)m * IL_CB, (hipStream_t)stream, c, m, n, points, idx, weight, out);
  } else {
    U3D_PO_LAUNCH_CM(three_interpolate_kernel, dim3((n + 255) / 256, (c + IC_CH - 1) / IC_CH, b), dim3(256), 0, (hipStream_t)stream, c, m, n, points,
                     idx, weight, out);
  }
  return hipGetLastError() == hipSuccess ? 0 : 3;
}

int u3d_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out, const int32_t* idx, const float* weight,
                               float* grad_points, void* stream) {
  if (b < 0 || c < 0 || m < 0 || n < 0) return 1;
  if (b == 0 || c == 0 || n == 0) return 0;
  if (b > 65535 || (c + IC_CH - 1) / IC_CH > 65535) return 1;
  if (!grad_out || !idx || !weight || !grad_points) return 1;
  int cb = m > 0 ? lds_rows(m, b, c) : 0;
  static const int ig_env = getenv("U3D_IG_CB") ? atoi(getenv("U3D_IG_CB")) : 0;   // (experiment switch)
  if (ig_env > 0 && cb > 0 && (size_t)ig_env * m * sizeof(float) <= 65536) cb = ig_env;
#define IG(CB) hipLaunchKernelGGL((three_interpolate_grad_lds_kernel<CB>), dim3((c + CB - 1) / CB, b), dim3(256), sizeof(float) * (size_t)CB * m, \
                                  (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points)
  if (cb == 16) IG(16);
  else if (cb == 8) IG(8);
  else if (cb == 4) IG(4);
  else if (cb == 2) IG(2);
  else if (cb == 1) IG(1);
#undef IG
  else
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
