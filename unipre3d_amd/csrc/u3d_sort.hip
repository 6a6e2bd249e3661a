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
//  * larger P: one most-significant-digit partition into 512 depth buckets (histogram + unordered scatter), then one workgroup per
//    bucket RANKS its (key, index) pairs in LDS and writes the sorted ids and tile rectangles: three launches (see below).
#include "u3d_common.h"

namespace {

// ---- 256 < P <= 4096: block radix sort -----------------------------------------------------------
// One workgroup per view runs all four 8-bit LSD passes in one launch; keys and payload ping-pong between two LDS buffers
// (64 KiB at N = 4096).  Per pass: LDS histogram, wave-shuffle scan of the 256 digit totals, then rounds of NT keys ranked
// with the ballot multi-split of radix_scatter_kernel (stable: element order = round, wave, lane); the per-wave digit counts
// become destinations through one prefix over the waves.  1024 threads (a lone 4-wave workgroup per CU cannot hide its own
// LDS and barrier latency: 34 us); measured at C3 (P = 2048, 64 views): 21 us against 32 us for a 66-stage bitonic network.
template <int NT>
__device__ __forceinline__ void block_radix_sort(U3DSpan span, int N, const float* __restrict__ depth, const int32_t* __restrict__ radii,
                                                 const uint2* __restrict__ rect, uint32_t* __restrict__ sorted_id,
                                                 uint2* __restrict__ sorted_rect, uint32_t* __restrict__ n_vis, uint32_t* lds) {
  constexpr int NW = NT / 64;
  // lds: keys[2][N], vals[2][N]
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

// ---- 256 < P <= 4096, first attempt: rank inside linear depth bins (round 3) ----------------------------------------------------
// The four radix passes above cost ~50 workgroup barriers (C3: 19 us for 2048 keys).  Object-level depths sit within an octave or
// two, so 1024 bins LINEAR in the key (depth bits) between the view's smallest and largest visible key hold a handful of keys each:
// bin by an LDS atomic, scan, scatter the (key, index) pairs to LDS, and every pair counts the smaller pairs of its own bin -- the
// rank gives "ascending depth bits, ties by ascending index" whatever order the atomics produced (six barriers).  Returns false
// (workgroup-uniform, nothing written) when some bin holds more than 256 pairs -- clustered depths -- and the radix sort runs.
template <int NT>
__device__ __forceinline__ bool block_rank_sort(U3DSpan span, int N, const float* __restrict__ depth, const int32_t* __restrict__ radii,
                                                const uint2* __restrict__ rect, uint32_t* __restrict__ sorted_id,
                                                uint2* __restrict__ sorted_rect, uint32_t* __restrict__ n_vis, uint32_t* lds) {
  constexpr int BINS = 1024, ITEMS = U3D_LDS_SORT_MAX / NT, BIN_MAX = 256;
  static_assert(NT == BINS, "one thread per bin in the scan");
  unsigned long long* pairs = reinterpret_cast<unsigned long long*>(lds);   // [N] (the radix sort's buffers, 16 N bytes, hold them)
  __shared__ uint32_t s_bin[BINS + 1];
  __shared__ uint32_t s_wv[NT / 64];
  __shared__ uint32_t s_min, s_max, s_big;
  const int view = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const uint32_t lane = u3d_lane_id();
  int P;
  size_t base;
  u3d_view_span(span, view, P, base);
  uint32_t k[ITEMS], slot[ITEMS];
  uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int i = r * NT + tid;
    k[r] = (i < P && radii[base + i] > 0) ? __float_as_uint(depth[base + i]) : 0xFFFFFFFFu;
    if (k[r] != 0xFFFFFFFFu) { lo = min(lo, k[r]); hi = max(hi, k[r]); }
  }
  s_bin[tid] = 0;
  if (tid == 0) { s_min = 0xFFFFFFFFu; s_max = 0u; s_big = 0u; s_bin[BINS] = 0; }
  __syncthreads();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); }
  if (lane == 0 && lo != 0xFFFFFFFFu) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
  __syncthreads();
  const uint32_t kmin = s_min, kmax = s_max;
  if (kmin == 0xFFFFFFFFu) {   // nothing visible
    if (tid == 0) n_vis[view] = 0;
    for (int i = tid; i < P; i += NT) { sorted_id[base + i] = 0u; sorted_rect[base + i] = make_uint2(0u, 0u); }
    return true;
  }
  const uint32_t range = kmax - kmin;
  const int shift = range < (uint32_t)BINS ? 0 : (32 - __clz((int)range)) - 10;   // (range >> shift) < 1024
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if (k[r] != 0xFFFFFFFFu) slot[r] = atomicAdd(&s_bin[(k[r] - kmin) >> shift], 1u);
  __syncthreads();
  {   // exclusive scan of the 1024 bin counts (one per thread) + the oversize test
    const uint32_t tot = s_bin[tid];
    uint32_t inc = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
      if ((int)lane >= o) inc += t;
    }
    if (lane == 63) s_wv[wave] = inc;
    if (tot > (uint32_t)BIN_MAX) s_big = 1u;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += s_wv[w];
    s_bin[tid] = off + inc - tot;
    if (tid == NT - 1) s_bin[BINS] = off + inc;
    __syncthreads();
  }
  if (s_big != 0u) { __syncthreads(); return false; }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if (k[r] != 0xFFFFFFFFu) pairs[s_bin[(k[r] - kmin) >> shift] + slot[r]] = ((unsigned long long)k[r] << 32) | (uint32_t)(r * NT + tid);
  __syncthreads();
  const uint32_t nv = s_bin[BINS];
  if (tid == 0) n_vis[view] = nv;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (k[r] != 0xFFFFFFFFu) {
      const uint32_t i = (uint32_t)(r * NT + tid);
      const unsigned long long mine = ((unsigned long long)k[r] << 32) | i;
      const uint32_t bin = (k[r] - kmin) >> shift;
      const uint32_t s0 = s_bin[bin], s1 = s_bin[bin + 1];
      uint32_t rank = s0;
      for (uint32_t j = s0; j < s1; ++j) rank += pairs[j] < mine ? 1u : 0u;
      sorted_id[base + rank] = i;
      sorted_rect[base + rank] = rect[base + i];
    }
  }
  for (int i = (int)nv + tid; i < P; i += NT) { sorted_id[base + i] = 0u; sorted_rect[base + i] = make_uint2(0u, 0u); }
  return true;
}

template <int NT, bool TRY_RANK>
__global__ __launch_bounds__(NT) void depth_sort_block_radix_kernel(U3DSpan span, int N, const float* __restrict__ depth,
                                                                    const int32_t* __restrict__ radii,
                                                                    const uint2* __restrict__ rect,
                                                                    uint32_t* __restrict__ sorted_id,
                                                                    uint2* __restrict__ sorted_rect,
                                                                    uint32_t* __restrict__ n_vis) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // keys[2][N], vals[2][N]  (or the rank path's pairs[N])
  if constexpr (TRY_RANK) {
    if (block_rank_sort<NT>(span, N, depth, radii, rect, sorted_id, sorted_rect, n_vis, lds)) return;
  }
  block_radix_sort<NT>(span, N, depth, radii, rect, sorted_id, sorted_rect, n_vis, lds);
}

// ---- large P: a coarse most-significant-digit partition, then every bucket sub-binned and ranked by its own workgroup --------
// Key = depth bits - bits(0.2f): a visible Gaussian has depth > 0.2 (the near cull), so the difference is >= 1 and order-preserving.
// Bucket = min(key >> 18, 511): 32 buckets per octave of depth up to 13107.2, everything beyond in the last one (64 per octave, 1024
// buckets, beyond 64 k Gaussians per view).
// No step below is stable and none needs to be: the final position of a pair is its RANK by (key, index) inside its bucket, so
// "ascending depth bits, ties by ascending Gaussian index" comes out whatever order the atomics produced (bit-identical runs).
//   msd_hist     per-workgroup bucket histogram in LDS (culled pairs are neither counted nor moved); each non-empty (workgroup,
//                bucket) count reserves its slice of the bucket with ONE global atomic on the view's totals (zeroed by
//                preprocess_fwd) and keeps the slice offset -- ~170 atomics per 4096 keys, not one per key (a global atomic per
//                key measured 5 x the whole old sort);
//   msd_scatter  bucket starts = scan of the 512 totals (every workgroup redoes it), destination = start + slice offset + the
//                key's arrival number inside the workgroup's slice (an LDS atomic): no column walk over per-workgroup tables, no
//                ballot multi-split;
//   bucket_sort  one workgroup per (view, bucket): the bucket's pairs go to 256 sub-bins of the next 8 key bits in LDS (LDS
//                atomics), then every pair counts the smaller pairs of its own sub-bin (a dozen of them): four barriers instead
//                of the ~30 of three LSD radix passes.  The last bucket (keys differ above bit 18), a bucket beyond the LDS
//                capacity and a bucket with a sub-bin of more than 512 pairs (thousands of near-equal depths) take stable LSD
//                radix passes over (index bits, then key bits) -- in LDS up to 1024 keys, through the global ping-pong buffers
//                beyond: correct for any input, slower.
// Round 2 (stable ballot multi-split scatter with a column walk, three 6-bit radix passes per bucket): C5 5.3 + 27.3 + 50.3 us,
// C4 5.0 + 16.0 + 20.3 us.  Sorted positions at and beyond n_vis[view] are NOT written: nothing reads them (tile_stage stops there).
constexpr uint32_t MSD_KEY_BASE = U3D_MSD_KEY_BASE;   // bits of 0.2f
// SHIFT = 18: 512 buckets, 32 per octave (up to 64 k Gaussians per view); SHIFT = 17: 1024 buckets, 64 per octave (beyond: halves the
// buckets, so that a 256-thread workgroup with 16 KB of LDS still holds one and all of them are resident at once); SHIFT = 16: 2048 buckets
// beyond 256 k per set (hist / scatter threads then own two buckets each)
template <int SHIFT> struct Msd { static constexpr int BINS = 16 << (23 - SHIFT); static_assert(BINS <= U3D_MSD_BINS_MAX, "scratch is carved for this many"); };
constexpr int SUB_BITS = 8, SUB_BINS = 1 << SUB_BITS;
constexpr int SUB_MAX = 512;                          // largest sub-bin ranked by all-pairs comparison
constexpr int BUCKET_LDS_CAP = 1024;                  // largest bucket the radix fallback sorts in LDS

template <int SHIFT> __device__ __forceinline__ uint32_t msd_bucket(uint32_t k) { return min(k >> SHIFT, (uint32_t)(Msd<SHIFT>::BINS - 1)); }

template <int NT, int ITEMS, int SHIFT>
__global__ __launch_bounds__(NT) void msd_hist_kernel(U3DSpan span, int nblk, const float* __restrict__ depth, uint32_t* __restrict__ total,
                                                      uint32_t* __restrict__ slice_off) {
  constexpr int MSD_BINS = Msd<SHIFT>::BINS;
  __shared__ uint32_t h[MSD_BINS];
  const int view = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  int P;
  size_t base;
  u3d_view_span(span, view, P, base);
  for (int t = tid; t < MSD_BINS; t += NT) h[t] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int idx = blk * (ITEMS * NT) + r * NT + tid;
    if (idx < P) {
      const float z = depth[base + idx];        // 0 for a culled pair, > 0.2 otherwise (preprocess_fwd)
      if (z > 0.f) atomicAdd(&h[msd_bucket<SHIFT>(__float_as_uint(z) - MSD_KEY_BASE)], 1u);
    }
  }
  __syncthreads();
  for (int t = tid; t < MSD_BINS; t += NT) {
    const uint32_t c = h[t];
    slice_off[((size_t)view * nblk + blk) * MSD_BINS + t] = c ? atomicAdd(&total[(size_t)view * MSD_BINS + t], c) : 0u;
  }
}

template <int NT, int ITEMS, int SHIFT>
__global__ __launch_bounds__(NT) void msd_scatter_kernel(U3DSpan span, int nblk, const float* __restrict__ depth,
                                                         uint2* __restrict__ pairs_out,
                                                         const uint32_t* __restrict__ total, const uint32_t* __restrict__ slice_off,
                                                         uint32_t* __restrict__ n_vis, uint32_t* __restrict__ bucket_off) {
  constexpr int MSD_BINS = Msd<SHIFT>::BINS;
  constexpr int BPT = MSD_BINS > NT ? MSD_BINS / NT : 1;   // consecutive buckets per thread of the prologue's scan
  constexpr int SCAN_T = MSD_BINS / BPT;                    // threads taking part in it (a multiple of 64)
  static_assert(MSD_BINS % BPT == 0 && SCAN_T <= NT && SCAN_T % 64 == 0, "the scan covers the buckets with whole waves");
  __shared__ uint32_t s_base[MSD_BINS], s_cnt[MSD_BINS];
  __shared__ uint32_t s_wave[SCAN_T / 64];
  const int view = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const uint32_t lane = u3d_lane_id();
  int P;
  size_t base;
  u3d_view_span(span, view, P, base);
  {
    uint32_t tot[BPT], sum = 0, inc = 0;
    if (tid < SCAN_T) {
#pragma unroll
      for (int q = 0; q < BPT; ++q) {
        tot[q] = total[(size_t)view * MSD_BINS + tid * BPT + q];
        sum += tot[q];
        s_cnt[tid * BPT + q] = 0;
      }
      inc = sum;   // inclusive scan of the threads' bucket totals: shuffles inside each of the first waves, then the wave totals
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, o);
        if ((int)lane >= o) inc += v;
      }
      if (lane == 63) s_wave[wave] = inc;
    }
    __syncthreads();
    if (tid < SCAN_T) {
      uint32_t off = 0;
      for (int w = 0; w < wave; ++w) off += s_wave[w];
      uint32_t start = off + inc - sum;
#pragma unroll
      for (int q = 0; q < BPT; ++q) {
        const int bkt = tid * BPT + q;
        if (blk == 0) bucket_off[(size_t)view * (MSD_BINS + 1) + bkt] = start;
        s_base[bkt] = start + slice_off[((size_t)view * nblk + blk) * MSD_BINS + bkt];
        start += tot[q];
      }
      if (blk == 0 && tid == SCAN_T - 1) { bucket_off[(size_t)view * (MSD_BINS + 1) + MSD_BINS] = start; n_vis[view] = start; }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int idx = blk * (ITEMS * NT) + r * NT + tid;
    if (idx < P) {
      const float z = depth[base + idx];
      if (z > 0.f) {
        const uint32_t k = __float_as_uint(z) - MSD_KEY_BASE;
        const uint32_t bkt = msd_bucket<SHIFT>(k);
        const uint32_t dst = s_base[bkt] + atomicAdd(&s_cnt[bkt], 1u);
        pairs_out[base + dst] = make_uint2(k, (uint32_t)idx);   // one 8-byte store per pair
      }
    }
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

// (key, index) order as one 64-bit compare
__device__ __forceinline__ unsigned long long pair64(uint32_t k, uint32_t v) { return ((unsigned long long)k << 32) | v; }

template <int NT, int ITEMS, int SHIFT>   // 256 x 8 = 2048 pairs in LDS
__global__ __launch_bounds__(NT) void bucket_sort_kernel(U3DSpan span, const uint2* __restrict__ pairs, uint32_t* __restrict__ keys0,
                                                         uint32_t* __restrict__ vals0, uint32_t* __restrict__ keys1, uint32_t* __restrict__ vals1,
                                                         const uint32_t* __restrict__ bucket_off, const uint2* __restrict__ rect,
                                                         uint32_t* __restrict__ sorted_id, uint2* __restrict__ sorted_rect) {
  constexpr int BITS = 6, CAP = NT * ITEMS, MSD_BINS = Msd<SHIFT>::BINS, MSD_SHIFT = SHIFT, SUB_SHIFT = SHIFT - SUB_BITS;
  static_assert(NT >= SUB_BINS && 2 * CAP >= 4 * BUCKET_LDS_CAP, "sub-bin scan by one thread each; the radix fallback reuses the pair array");
  __shared__ __attribute__((aligned(16))) unsigned long long s_pair[CAP];     // 16 / 32 KB
  __shared__ uint32_t s_sub[SUB_BINS + 1];
  __shared__ uint32_t s_wave[SUB_BINS / 64];
  __shared__ uint32_t s_max;
  __shared__ uint32_t digit_base[1 << BITS];
  __shared__ uint32_t wave_cnt[2][NT / 64][1 << BITS];
  const int view = blockIdx.y, bucket = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const uint32_t lane = u3d_lane_id();
  int P;
  size_t base;
  u3d_view_span(span, view, P, base);
  const uint32_t start = bucket_off[(size_t)view * (MSD_BINS + 1) + bucket], end = bucket_off[(size_t)view * (MSD_BINS + 1) + bucket + 1];
  const int n = (int)(end - start);
  if (n == 0) return;
  if (n == 1) {
    if (tid == 0) sorted_id[base + start] = pairs[base + start].y;
    return;
  }
  bool fallback = n > CAP || bucket == MSD_BINS - 1;
  if (!fallback) {
    // sub-bin = the next 8 key bits below the bucket's; arrival number inside the sub-bin from an LDS atomic
    uint32_t k[ITEMS], v[ITEMS], slot[ITEMS];
    if (tid < SUB_BINS) s_sub[tid] = 0;
    if (tid == 0) s_max = 0;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const int i = r * NT + tid;
      if (i < n) {
        const uint2 kv = pairs[base + start + i];
        k[r] = kv.x;
        v[r] = kv.y;
        slot[r] = atomicAdd(&s_sub[(k[r] >> SUB_SHIFT) & (SUB_BINS - 1)], 1u);
      }
    }
    __syncthreads();
    {   // exclusive scan of the 256 sub-bin counts (first four waves) + the largest sub-bin
      uint32_t tot = 0, inc = 0;
      if (tid < SUB_BINS) {
        tot = s_sub[tid];
        inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t t = (uint32_t)__shfl_up((int)inc, o);
          if ((int)lane >= o) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        if (tot > (uint32_t)SUB_MAX) atomicMax(&s_max, tot);
      }
      __syncthreads();
      if (tid < SUB_BINS) {
        uint32_t off = 0;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        s_sub[tid] = off + inc - tot;
        if (tid == SUB_BINS - 1) s_sub[SUB_BINS] = off + inc;
      }
      __syncthreads();
    }
    fallback = s_max != 0u;     // (workgroup-uniform) a sub-bin too large to rank by comparison: near-equal depths by the thousand
    if (!fallback) {
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        const int i = r * NT + tid;
        if (i < n) s_pair[s_sub[(k[r] >> SUB_SHIFT) & (SUB_BINS - 1)] + slot[r]] = pair64(k[r], v[r]);
      }
      __syncthreads();
      for (int p = tid; p < n; p += NT) {
        const unsigned long long mine = s_pair[p];
        const uint32_t sub = ((uint32_t)(mine >> 32) >> SUB_SHIFT) & (SUB_BINS - 1);
        const uint32_t s0 = s_sub[sub], s1 = s_sub[sub + 1];
        uint32_t rank = s0;
        for (uint32_t j = s0; j < s1; ++j) rank += s_pair[j] < mine ? 1u : 0u;
        sorted_id[base + start + rank] = (uint32_t)mine;   // (the tile kernels read the rectangle through the id: u3d_rect_indirect)
      }
      return;
    }
    __syncthreads();
  }
  // ---- fallback: stable LSD radix passes, first over the index bits, then over the key bits that can differ inside the bucket ----
  {
    uint32_t* const s_data = reinterpret_cast<uint32_t*>(s_pair);
    int idx_bits = 1;
    while ((1 << idx_bits) < P) ++idx_bits;
    const int idx_passes = (idx_bits + BITS - 1) / BITS;
    const int key_passes = bucket == MSD_BINS - 1 ? (32 + BITS - 1) / BITS : (MSD_SHIFT + BITS - 1) / BITS;
    const int passes = idx_passes + key_passes;
    // ping-pong buffers: LDS up to BUCKET_LDS_CAP keys, the global key / value arrays beyond (pointers formed by arithmetic: an
    // array of LDS addresses would be a constant initialiser the backend cannot express)
    const bool in_lds = n <= BUCKET_LDS_CAP;
    uint32_t* const k0 = in_lds ? s_data : keys0 + base + start;
    uint32_t* const k1 = in_lds ? s_data + BUCKET_LDS_CAP : keys1 + base + start;
    uint32_t* const v0 = in_lds ? s_data + 2 * BUCKET_LDS_CAP : vals0 + base + start;
    uint32_t* const v1 = in_lds ? s_data + 3 * BUCKET_LDS_CAP : vals1 + base + start;
    for (int i = tid; i < n; i += NT) { const uint2 kv = pairs[base + start + i]; k0[i] = kv.x; v0[i] = kv.y; }   // un-zip (LDS or global)
    __threadfence_block();
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
      uint32_t* ki = (p & 1) ? k1 : k0; uint32_t* ko = (p & 1) ? k0 : k1;
      uint32_t* vi = (p & 1) ? v1 : v0; uint32_t* vo = (p & 1) ? v0 : v1;
      if (p < idx_passes)   // sort by an index digit: key and payload swap roles
        wg_radix_pass<NT, BITS>(vi, ki, vo, ko, n, BITS * p, digit_base, wave_cnt);
      else
        wg_radix_pass<NT, BITS>(ki, vi, ko, vo, n, BITS * (p - idx_passes), digit_base, wave_cnt);
    }
    const uint32_t* vf = (passes & 1) ? v1 : v0;
    for (int i = tid; i < n; i += NT) sorted_id[base + start + i] = vf[i];
  }
}

}  // namespace

void u3d_launch_depth_sort(const u3d_raster_desc& d, const U3DBuffers& b, const int32_t* radii, hipStream_t s) {
  const int NV = d.n_items * d.views_per_item;
  if (d.P <= U3D_LDS_SORT_MAX) {
    {
      // 1024 threads once there is more than one round of them (16 waves hide the LDS / barrier latency of a lone workgroup)
      const int NT = 1024;   // (one thread per linear depth bin of the rank path; the radix fallback wants 16 waves anyway)
      const int N = (d.P + NT - 1) / NT * NT;
      static bool attr_set = false;
      if (!attr_set) {   // up to 64 KiB of dynamic LDS (N = 4096) on top of the static arrays
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(depth_sort_block_radix_kernel<1024, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 4 * U3D_LDS_SORT_MAX * (int)sizeof(uint32_t));
        attr_set = true;
      }
      if (NT == 1024)
        hipLaunchKernelGGL((depth_sort_block_radix_kernel<1024, true>), dim3(NV), dim3(1024), (size_t)4 * N * sizeof(uint32_t), s, u3d_span(d), N, b.depth,
                           radii, b.rect, b.sorted_id, b.sorted_rect, b.n_vis);
      else
        hipLaunchKernelGGL((depth_sort_block_radix_kernel<256, false>), dim3(NV), dim3(256), (size_t)4 * N * sizeof(uint32_t), s, u3d_span(d), N, b.depth,
                           radii, b.rect, b.sorted_id, b.sorted_rect, b.n_vis);
    }
    return;
  }
  // (the views' bucket totals b.sort_hist[NV][bins] were zeroed by preprocess_fwd)
  const int tile = u3d_radix_tile(d.P);
  const int nblk = (d.P + tile - 1) / tile;
  const int bins = u3d_msd_bins(d.P);
  uint32_t* total = b.sort_hist;
  uint32_t* slice_off = b.sort_hist + (size_t)NV * bins;
#define LAUNCH(NT, IT, SH, BIT)                                                                                                               \
  do {                                                                                                                                        \
    hipLaunchKernelGGL((msd_hist_kernel<NT, IT, SH>), dim3(nblk, NV), dim3(NT), 0, s, u3d_span(d), nblk, b.depth, total, slice_off);          \
    hipLaunchKernelGGL((msd_scatter_kernel<NT, IT, SH>), dim3(nblk, NV), dim3(NT), 0, s, u3d_span(d), nblk, b.depth, b.sort_pairs,            \
                       total, slice_off, b.n_vis, b.sort_over);                                                                               \
    hipLaunchKernelGGL((bucket_sort_kernel<256, BIT, SH>), dim3(bins, NV), dim3(256), 0, s, u3d_span(d), b.sort_pairs, b.sort_keys[0], b.sort_vals[0], \
                       b.sort_keys[1], b.sort_vals[1], b.sort_over, b.rect, b.sorted_id, b.sorted_rect);                                      \
  } while (0)
  // beyond 256 k Gaussians per set the fuller depth buckets exceed the 2048 pairs a bucket's workgroup ranks in LDS (C5 + fused
  // pixel-Gaussians, 350 k: the radix fallback through global memory took 110 us): 2048 buckets there, 128 per octave (44.7 us; 1024 buckets
  // with 4096-pair workgroups measured 49.9)
  if (d.P <= 65536) LAUNCH(U3D_RADIX_NT_SMALL, U3D_RADIX_IT_SMALL, 18, 8);
  else if (d.P <= 262144) LAUNCH(U3D_RADIX_NT_LARGE, U3D_RADIX_IT_LARGE, 17, 8);
  else LAUNCH(U3D_RADIX_NT_LARGE, U3D_RADIX_IT_LARGE, 16, 8);
#undef LAUNCH
}
