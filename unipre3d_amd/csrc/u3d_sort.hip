// Front-to-back ordering: ONE depth sort per view (not one per tile instance).
//
// The original operator duplicates every Gaussian once per touched tile, sorts the R = sum(tiles
// touched) 64-bit keys (tile << 32 | depth bits) with a stable radix sort and then finds per-tile
// ranges.  Within a tile that order is "ascending depth bits, ties by ascending
// Gaussian index".  Sorting the P Gaussians of a view once by (depth bits, index) and letting every
// tile filter that list by its rectangle (u3d_render.hip) yields exactly the same per-tile sequence
// while moving 12*P instead of 36*R bytes (R ~ P*T for the reference's large splats).
//
//  * P <= 256: fused into preprocess_fwd (bitonic network over 256 (depth bits << 32 | index) keys in LDS).
//  * P <= 4096: one workgroup per view, 4-pass LSD radix sort entirely in LDS (64 KiB of the CU's 160 KiB), one launch.
//  * larger P: one most-significant-digit partition into 512 depth buckets (histogram + stable scatter), then one workgroup per
//    bucket sorts it in LDS and writes the sorted ids and tile rectangles: three launches (see below).
#include "u3d_common.h"

namespace {

// ---- 256 < P <= 4096: block radix sort -----------------------------------------------------------
// One workgroup per view runs all four 8-bit LSD passes in one launch; keys and payload ping-pong between two LDS buffers
// (64 KiB at N = 4096).  Per pass: LDS histogram, wave-shuffle scan of the 256 digit totals, then rounds of NT keys ranked
// with the ballot multi-split of radix_scatter_kernel (stable: element order = round, wave, lane); the per-wave digit counts
// become destinations through one prefix over the waves.  1024 threads (a lone 4-wave workgroup per CU cannot hide its own
// LDS and barrier latency: 34 us); measured at C3 (P = 2048, 64 views): 21 us against 32 us for a 66-stage bitonic network.
template <int NT>
__global__ __launch_bounds__(NT) void depth_sort_block_radix_kernel(U3DSpan span, int N, const float* __restrict__ depth,
                                                                    const int32_t* __restrict__ radii,
                                                                    const uint2* __restrict__ rect,
                                                                    uint32_t* __restrict__ sorted_id,
                                                                    uint2* __restrict__ sorted_rect,
                                                                    uint32_t* __restrict__ n_vis) {
  constexpr int NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // keys[2][N], vals[2][N]
  __shared__ uint32_t digit_base[256];
  __shared__ uint32_t wave_cnt[2][NW][256];   // double-buffered per round: counts, then exclusive prefixes over the waves
  __shared__ uint32_t wave_tot[4];
  uint32_t* keys[2] = {lds, lds + N};
  uint32_t* vals[2] = {lds + 2 * N, lds + 3 * N};
  const int view = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const uint32_t lane = u3d_lane_id();
  int P;          // Gaussians of this view's set (<= N)
  size_t base;    // its first (view, Gaussian) pair
  u3d_view_span(span, view, P, base);
  const int rounds = N / NT;
  for (int i = tid; i < N; i += NT) {
    keys[0][i] = (i < P && radii[base + i] > 0) ? __float_as_uint(depth[base + i]) : 0xFFFFFFFFu;
    vals[0][i] = (uint32_t)i;
  }
  __syncthreads();
  for (int pass = 0; pass < 4; ++pass) {
    const uint32_t* kin = keys[pass & 1];
    const uint32_t* vin = vals[pass & 1];
    uint32_t* kout = keys[(pass + 1) & 1];
    uint32_t* vout = vals[(pass + 1) & 1];
    const int sh = 8 * pass;
    if (tid < 256) digit_base[tid] = 0;
    for (int e = tid; e < 2 * NW * 256; e += NT) (&wave_cnt[0][0][0])[e] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += NT) atomicAdd(&digit_base[(kin[i] >> sh) & 255u], 1u);
    __syncthreads();
    {   // exclusive scan of the 256 totals (first four waves): inclusive scan inside each wave by shuffles, then the wave totals
      uint32_t tot = 0, inc = 0;
      if (tid < 256) {
        tot = digit_base[tid];
        inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
          if ((int)lane >= o) inc += v;
        }
        if (lane == 63) wave_tot[wave] = inc;
      }
      __syncthreads();
      if (tid < 256) {
        uint32_t off = 0;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        digit_base[tid] = off + inc - tot;
      }
      __syncthreads();
    }
    for (int r = 0; r < rounds; ++r) {
      const int idx = r * NT + tid;
      const uint32_t k = kin[idx], v = vin[idx];
      const uint32_t digit = (k >> sh) & 255u;
      unsigned long long same = ~0ull;
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        same &= bit ? m : ~m;
      }
      const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull));
      uint32_t (*cnt)[256] = wave_cnt[r & 1];
      if (rank == 0) cnt[wave][digit] = (uint32_t)__popcll(same);
      __syncthreads();
      if (tid < 256) {   // counts -> destination of each wave's first key of this digit; digit_base moves past the round
        uint32_t run = digit_base[tid];
#pragma unroll
        for (int w = 0; w < NW; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
        digit_base[tid] = run;
      }
      __syncthreads();
      const uint32_t dst = cnt[wave][digit] + rank;
      kout[dst] = k;
      vout[dst] = v;
      for (int e = tid; e < NW * 256; e += NT) (&wave_cnt[(r + 1) & 1][0][0])[e] = 0;   // the other buffer, for the next round
      __syncthreads();
    }
  }
  const uint32_t* kf = keys[0];
  const uint32_t* vf = vals[0];
  if (tid == 0 && kf[0] == 0xFFFFFFFFu) n_vis[view] = 0;
  for (int i = tid; i < P; i += NT) {
    const uint32_t k = kf[i];
    const bool vis = k != 0xFFFFFFFFu;
    if (vis && (i == N - 1 || kf[i + 1] == 0xFFFFFFFFu)) n_vis[view] = (uint32_t)(i + 1);
    const uint32_t id = vis ? vf[i] : 0u;
    sorted_id[base + i] = id;
    sorted_rect[base + i] = vis ? rect[base + id] : make_uint2(0u, 0u);
  }
}

// ---- large P: one most-significant-digit partition, then every bucket sorted by its own workgroup -------------------------
// Key = depth bits - bits(0.2f): a visible Gaussian has depth > 0.2 (the near cull), so the difference is >= 1 and order-preserving.
// Bucket = min(key >> 18, 511): 32 buckets per octave of depth up to 13107.2, everything beyond in the last one.
//   msd_hist     per-block bucket histogram (culled Gaussians are dropped here: they are neither counted nor moved);
//   msd_scatter  stable partition into bucket order (ballot multi-split ranking, per-block offsets from the histograms), bucket
//                start table, n_vis;
//   bucket_sort  one 256- or 512-thread workgroup per (view, bucket): LSD radix sort (6-bit digits) of the bucket's keys on their low 18 bits (the
//                whole key in the last bucket), in LDS when the bucket fits (<= 1024 keys), through the global ping-pong buffers
//                otherwise (an unusually dense or degenerate bucket: correct, slower); writes the sorted ids and rectangles.
// Three launches instead of the nine of a four-pass LSD sort over all keys (each of which is latency-bound at these sizes).
// Stable throughout and the initial order is index order, so depth ties resolve by ascending Gaussian index.
constexpr uint32_t MSD_KEY_BASE = 0x3E4CCCCDu;   // bits of 0.2f
constexpr int MSD_SHIFT = 18, MSD_BITS = 9, MSD_BINS = 1 << MSD_BITS;
constexpr int BUCKET_LDS_CAP = 1024;

__device__ __forceinline__ uint32_t msd_key(int P, int idx, size_t base, const float* depth, const int32_t* radii) {
  return radii[base + idx] > 0 ? __float_as_uint(depth[base + idx]) - MSD_KEY_BASE : 0xFFFFFFFFu;
}
__device__ __forceinline__ uint32_t msd_bucket(uint32_t k) { return min(k >> MSD_SHIFT, (uint32_t)(MSD_BINS - 1)); }

template <int NT, int ITEMS>
__global__ __launch_bounds__(NT) void msd_hist_kernel(U3DSpan span, int nblk, const float* __restrict__ depth, const int32_t* __restrict__ radii,
                                                      uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[MSD_BINS];
  const int view = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  int P;
  size_t base;
  u3d_view_span(span, view, P, base);
  if (tid < MSD_BINS) h[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int idx = blk * (ITEMS * NT) + r * NT + tid;
    if (idx < P) {
      const uint32_t k = msd_key(P, idx, base, depth, radii);
      if (k != 0xFFFFFFFFu) atomicAdd(&h[msd_bucket(k)], 1u);
    }
  }
  __syncthreads();
  if (tid < MSD_BINS) hist[((size_t)view * nblk + blk) * MSD_BINS + tid] = h[tid];   // [view][block][bucket]: coalesced
}

template <int NT, int ITEMS>
__global__ __launch_bounds__(NT) void msd_scatter_kernel(U3DSpan span, int nblk, const float* __restrict__ depth, const int32_t* __restrict__ radii,
                                                         uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                         const uint32_t* __restrict__ hist, uint32_t* __restrict__ n_vis,
                                                         uint32_t* __restrict__ bucket_off) {
  constexpr int NW = NT / 64;
  static_assert(NT >= MSD_BINS, "one thread per bucket in the prologue");
  extern __shared__ uint32_t s_dyn[];                                   // wave_cnt[2][NW][MSD_BINS], double-buffered per round:
  uint32_t (*wave_cnt)[NW][MSD_BINS] = reinterpret_cast<uint32_t (*)[NW][MSD_BINS]>(s_dyn);   // counts, then prefixes over the waves
  __shared__ uint32_t digit_base[MSD_BINS];
  __shared__ uint32_t wave_tot[MSD_BINS / 64];
  const int view = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6;
  const uint32_t lane = u3d_lane_id();
  int P;
  size_t base;
  u3d_view_span(span, view, P, base);
  for (int e = tid; e < 2 * NW * MSD_BINS; e += NT) (&wave_cnt[0][0][0])[e] = 0;
  {
    // global offset of (bucket tid, this block) in bucket-major / block-minor order, from the per-block counts hist[view][b][bucket]
    // (every block redoes this small scan: no separate scan launch; 8 loads in flight: the column walk is latency-bound)
    uint32_t tot = 0, before = 0, inc = 0;
    if (tid < MSD_BINS) {
      const uint32_t* col = hist + (size_t)view * nblk * MSD_BINS + tid;
      int b = 0;
      for (; b + 7 < nblk; b += 8) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = col[(size_t)(b + u) * MSD_BINS];
#pragma unroll
        for (int u = 0; u < 8; ++u) { tot += c[u]; before += b + u < blk ? c[u] : 0u; }
      }
      for (; b < nblk; ++b) { const uint32_t c0 = col[(size_t)b * MSD_BINS]; tot += c0; before += b < blk ? c0 : 0u; }
      inc = tot;   // inclusive scan of the bucket totals: shuffles inside each of the first waves, then the wave totals
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
        if ((int)lane >= o) inc += v;
      }
      if (lane == 63) wave_tot[wave] = inc;
    }
    __syncthreads();
    if (tid < MSD_BINS) {
      uint32_t off = 0;
      for (int w = 0; w < wave; ++w) off += wave_tot[w];
      const uint32_t start = off + inc - tot;
      if (blk == 0) {
        bucket_off[(size_t)view * (MSD_BINS + 1) + tid] = start;
        if (tid == MSD_BINS - 1) { bucket_off[(size_t)view * (MSD_BINS + 1) + MSD_BINS] = off + inc; n_vis[view] = off + inc; }
      }
      digit_base[tid] = start + before;
    }
    __syncthreads();
  }
  for (int r = 0; r < ITEMS; ++r) {
    const int idx = blk * (ITEMS * NT) + r * NT + tid;
    bool valid = idx < P;
    uint32_t k = 0, digit = 0;
    if (valid) {
      k = msd_key(P, idx, base, depth, radii);
      digit = msd_bucket(k);
      valid = k != 0xFFFFFFFFu;
    }
    // lanes of this wave holding the same bucket (stable multi-split, one ballot per bit)
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < MSD_BITS; ++b) {
      const bool bit = (digit >> b) & 1u;
      const unsigned long long m = __ballot(bit && valid);
      same &= bit ? m : ~m;
    }
    const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull));
    uint32_t (*cnt)[MSD_BINS] = wave_cnt[r & 1];
    if (valid && rank == 0) cnt[wave][digit] = (uint32_t)__popcll(same);
    __syncthreads();
    if (tid < MSD_BINS) {   // counts -> destination of each wave's first key of this bucket; digit_base moves past the round
      uint32_t run = digit_base[tid];
#pragma unroll
      for (int w = 0; w < NW; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
      digit_base[tid] = run;
    }
    __syncthreads();
    if (valid) {
      const uint32_t dst = cnt[wave][digit] + rank;
      keys_out[base + dst] = k;
      vals_out[base + dst] = (uint32_t)idx;
    }
    for (int e = tid; e < NW * MSD_BINS; e += NT) (&wave_cnt[(r + 1) & 1][0][0])[e] = 0;   // the other buffer, for the next round
    __syncthreads();
  }
}

// One stable LSD pass of a workgroup over n keys (kin/vin -> kout/vout; LDS or global arrays), digit = (key >> sh) & (2^BITS - 1).
// 6-bit digits: the 64 digit totals are scanned by one wave and the per-wave count table is 4x smaller than with 8 bits -- the fixed
// cost of a pass matters more than the pass count for buckets of a few hundred keys.
template <int NT, int BITS>
__device__ __forceinline__ void wg_radix_pass(const uint32_t* kin, const uint32_t* vin, uint32_t* kout, uint32_t* vout, int n, int sh,
                                              uint32_t* digit_base, uint32_t (*wave_cnt)[NT / 64][1 << BITS]) {
  constexpr int NW = NT / 64, BINS = 1 << BITS;
  static_assert(BINS == 64, "the digit totals are scanned by one wave");
  const int tid = threadIdx.x, wave = tid >> 6;
  const uint32_t lane = u3d_lane_id();
  if (tid < BINS) digit_base[tid] = 0;
  for (int e = tid; e < 2 * NW * BINS; e += NT) (&wave_cnt[0][0][0])[e] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += NT) atomicAdd(&digit_base[(kin[i] >> sh) & (BINS - 1)], 1u);
  __syncthreads();
  if (tid < BINS) {   // exclusive scan of the 64 totals inside the first wave
    const uint32_t tot = digit_base[tid];
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
      if ((int)lane >= o) inc += v;
    }
    digit_base[tid] = inc - tot;
  }
  __syncthreads();
  const int rounds = (n + NT - 1) / NT;
  for (int r = 0; r < rounds; ++r) {
    const int idx = r * NT + tid;
    const bool valid = idx < n;
    const uint32_t k = valid ? kin[idx] : 0u, v = valid ? vin[idx] : 0u;
    const uint32_t digit = (k >> sh) & (BINS - 1);
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const bool bit = (digit >> b) & 1u;
      const unsigned long long m = __ballot(bit && valid);
      same &= bit ? m : ~m;
    }
    const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull));
    uint32_t (*cnt)[BINS] = wave_cnt[r & 1];
    if (valid && rank == 0) cnt[wave][digit] = (uint32_t)__popcll(same);
    __syncthreads();
    if (tid < BINS) {
      uint32_t run = digit_base[tid];
#pragma unroll
      for (int w = 0; w < NW; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
      digit_base[tid] = run;
    }
    __syncthreads();
    if (valid) {
      const uint32_t dst = cnt[wave][digit] + rank;
      kout[dst] = k;
      vout[dst] = v;
    }
    for (int e = tid; e < NW * BINS; e += NT) (&wave_cnt[(r + 1) & 1][0][0])[e] = 0;
    __threadfence_block();   // (global ping-pong: this pass's stores are read by other threads of the workgroup in the next one)
    __syncthreads();
  }
}

template <int BUCKET_NT>   // 256 threads per bucket up to 64 k Gaussians per view, 512 beyond (denser buckets: C5 sort 97 -> 85 us, C4 49 -> 52)
__global__ __launch_bounds__(BUCKET_NT) void bucket_sort_kernel(U3DSpan span, int lds_cap, uint32_t* __restrict__ keys0, uint32_t* __restrict__ vals0,
                                                                uint32_t* __restrict__ keys1, uint32_t* __restrict__ vals1,
                                                                const uint32_t* __restrict__ bucket_off, const uint2* __restrict__ rect,
                                                                uint32_t* __restrict__ sorted_id, uint2* __restrict__ sorted_rect) {
  constexpr int NT = BUCKET_NT, NW = NT / 64;
  extern __shared__ __attribute__((aligned(16))) uint32_t s_data[];   // keys[2][cap], vals[2][cap]
  constexpr int BITS = 6;
  __shared__ uint32_t digit_base[1 << BITS];
  __shared__ uint32_t wave_cnt[2][NW][1 << BITS];
  const int view = blockIdx.y, bucket = blockIdx.x, tid = threadIdx.x;
  int P;
  size_t base;
  u3d_view_span(span, view, P, base);
  const uint32_t start = bucket_off[(size_t)view * (MSD_BINS + 1) + bucket], end = bucket_off[(size_t)view * (MSD_BINS + 1) + bucket + 1];
  const int n = (int)(end - start);
  if (n == 0) {
    // (positions past the visible keys read id 0 / empty rectangle: the last bucket's workgroup fills them)
  } else if (n == 1) {
    if (tid == 0) { const uint32_t v = vals0[base + start]; sorted_id[base + start] = v; sorted_rect[base + start] = rect[base + v]; }
  } else {
    // bits that can differ inside a bucket: the low 18 (three 6-bit digits), the whole key in the last bucket (six)
    const int passes = bucket == MSD_BINS - 1 ? 6 : 3;
    const uint32_t* kf;
    const uint32_t* vf;
    if (n <= lds_cap) {
      uint32_t* lk[2] = {s_data, s_data + lds_cap};
      uint32_t* lv[2] = {s_data + 2 * lds_cap, s_data + 3 * lds_cap};
      for (int i = tid; i < n; i += NT) { lk[0][i] = keys0[base + start + i]; lv[0][i] = vals0[base + start + i]; }
      __syncthreads();
      for (int p = 0; p < passes; ++p)
        wg_radix_pass<NT, BITS>(lk[p & 1], lv[p & 1], lk[(p + 1) & 1], lv[(p + 1) & 1], n, BITS * p, digit_base, wave_cnt);
      kf = lk[passes & 1]; vf = lv[passes & 1];
    } else {
      uint32_t* gk[2] = {keys0 + base + start, keys1 + base + start};
      uint32_t* gv[2] = {vals0 + base + start, vals1 + base + start};
      for (int p = 0; p < passes; ++p)
        wg_radix_pass<NT, BITS>(gk[p & 1], gv[p & 1], gk[(p + 1) & 1], gv[(p + 1) & 1], n, BITS * p, digit_base, wave_cnt);
      kf = gk[passes & 1]; vf = gv[passes & 1];
    }
    (void)kf;
    for (int i = tid; i < n; i += NT) {
      const uint32_t v = vf[i];
      sorted_id[base + start + i] = v;
      sorted_rect[base + start + i] = rect[base + v];
    }
  }
  if (bucket == MSD_BINS - 1) {
    const uint32_t nv = bucket_off[(size_t)view * (MSD_BINS + 1) + MSD_BINS];
    for (int i = (int)nv + tid; i < P; i += NT) { sorted_id[base + i] = 0u; sorted_rect[base + i] = make_uint2(0u, 0u); }
  }
}

}  // namespace

void u3d_launch_depth_sort(const u3d_raster_desc& d, const U3DBuffers& b, const int32_t* radii, hipStream_t s) {
  const int NV = d.n_items * d.views_per_item;
  if (d.P <= U3D_LDS_SORT_MAX) {
    {
      // 1024 threads once there is more than one round of them (16 waves hide the LDS / barrier latency of a lone workgroup)
      const int NT = d.P > 1024 ? 1024 : 256;
      const int N = (d.P + NT - 1) / NT * NT;
      static bool attr_set = false;
      if (!attr_set) {   // up to 64 KiB of dynamic LDS (N = 4096) on top of the static arrays
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(depth_sort_block_radix_kernel<1024>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 4 * U3D_LDS_SORT_MAX * (int)sizeof(uint32_t));
        attr_set = true;
      }
      if (NT == 1024)
        hipLaunchKernelGGL(depth_sort_block_radix_kernel<1024>, dim3(NV), dim3(1024), (size_t)4 * N * sizeof(uint32_t), s, u3d_span(d), N, b.depth,
                           radii, b.rect, b.sorted_id, b.sorted_rect, b.n_vis);
      else
        hipLaunchKernelGGL(depth_sort_block_radix_kernel<256>, dim3(NV), dim3(256), (size_t)4 * N * sizeof(uint32_t), s, u3d_span(d), N, b.depth,
                           radii, b.rect, b.sorted_id, b.sorted_rect, b.n_vis);
    }
    return;
  }
  const int tile = u3d_radix_tile(d.P);
  const int nblk = (d.P + tile - 1) / tile;
  // LDS per bucket workgroup decides how many buckets a CU sorts at once against how many take the global route: 1024 keys measured
  // best up to 64 k Gaussians per view (C4), 4096 beyond (C5: denser buckets)
  static const int lds_cap_env = getenv("U3D_BUCKET_LDS_CAP") ? atoi(getenv("U3D_BUCKET_LDS_CAP")) : 0;   // (tests force the global route)
  const int lds_cap = lds_cap_env ? lds_cap_env : (d.P <= 65536 ? BUCKET_LDS_CAP : 4096);
#define LAUNCH(NT, IT)                                                                                                       \
  do {                                                                                                                       \
    constexpr size_t lds = (size_t)2 * (NT / 64) * MSD_BINS * sizeof(uint32_t);                                              \
    static bool attr = false;                                                                                                \
    if (!attr) {                                                                                                             \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(msd_scatter_kernel<NT, IT>),                                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sort_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                4 * 4096 * (int)sizeof(uint32_t));                                                          \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sort_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                4 * 4096 * (int)sizeof(uint32_t));                                                          \
      attr = true;                                                                                                           \
    }                                                                                                                        \
    hipLaunchKernelGGL((msd_hist_kernel<NT, IT>), dim3(nblk, NV), dim3(NT), 0, s, u3d_span(d), nblk, b.depth, radii, b.sort_hist);   \
    hipLaunchKernelGGL((msd_scatter_kernel<NT, IT>), dim3(nblk, NV), dim3(NT), lds, s, u3d_span(d), nblk, b.depth, radii,            \
                       b.sort_keys[0], b.sort_vals[0], b.sort_hist, b.n_vis, b.sort_over);                                   \
  } while (0)
  if (d.P <= 65536) LAUNCH(U3D_RADIX_NT_SMALL, U3D_RADIX_IT_SMALL); else LAUNCH(U3D_RADIX_NT_LARGE, U3D_RADIX_IT_LARGE);
#undef LAUNCH
  const int cap = lds_cap < 1 ? 1 : (lds_cap > 4096 ? 4096 : lds_cap);
  if (d.P <= 65536)
    hipLaunchKernelGGL(bucket_sort_kernel<256>, dim3(MSD_BINS, NV), dim3(256), (size_t)4 * cap * sizeof(uint32_t), s, u3d_span(d), cap, b.sort_keys[0],
                       b.sort_vals[0], b.sort_keys[1], b.sort_vals[1], b.sort_over, b.rect, b.sorted_id, b.sorted_rect);
  else
    hipLaunchKernelGGL(bucket_sort_kernel<512>, dim3(MSD_BINS, NV), dim3(512), (size_t)4 * cap * sizeof(uint32_t), s, u3d_span(d), cap, b.sort_keys[0],
                       b.sort_vals[0], b.sort_keys[1], b.sort_vals[1], b.sort_over, b.rect, b.sorted_id, b.sorted_rect);
}
