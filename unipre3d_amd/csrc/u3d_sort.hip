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
//  * larger P: 4-pass LSD radix sort (8-bit digits) on the depth bits with the index as payload;
//    LSD passes are stable and the initial order is index order, so ties resolve by index.  Per pass: per-block digit
//    histogram, then a scatter whose prologue turns the histograms into its own offsets (no separate scan launch) and
//    ranks keys with ballot multi-split (8 ballots per key, stable within the wave, waves ordered through LDS); the last
//    pass writes the sorted ids and their tile rectangles directly.
#include "u3d_common.h"

namespace {

// ---- 256 < P <= 4096: block radix sort -----------------------------------------------------------
// One workgroup per view runs all four 8-bit LSD passes in one launch; keys and payload ping-pong between two LDS buffers
// (64 KiB at N = 4096).  Per pass: LDS histogram, wave-shuffle scan of the 256 digit totals, then rounds of NT keys ranked
// with the ballot multi-split of radix_scatter_kernel (stable: element order = round, wave, lane); the per-wave digit counts
// become destinations through one prefix over the waves.  1024 threads (a lone 4-wave workgroup per CU cannot hide its own
// LDS and barrier latency: 34 us); measured at C3 (P = 2048, 64 views): 21 us against 32 us for a 66-stage bitonic network.
template <int NT>
__global__ __launch_bounds__(NT) void depth_sort_block_radix_kernel(int P, int N, const float* __restrict__ depth,
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
  const size_t base = (size_t)view * P;
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

// ---- large P: LSD radix sort, 8-bit digits, ITEMS*256 keys per workgroup (u3d_radix_tile) -------

__device__ __forceinline__ uint32_t radix_key(int pass, int P, int idx, size_t base, const float* depth,
                                              const int32_t* radii, const uint32_t* keys_in) {
  if (pass == 0) return radii[base + idx] > 0 ? __float_as_uint(depth[base + idx]) : 0xFFFFFFFFu;
  return keys_in[base + idx];
}

// Pass 0 drops the culled Gaussians (key 0xFFFFFFFF): they are neither counted nor scattered, so passes 1-3 and the
// finalize step only see the n_vis[view] visible keys (53 % of the keys at C5); workgroups past that count leave at once.
// NT threads x ITEMS keys per workgroup and pass.
template <int NT, int ITEMS>
__global__ __launch_bounds__(NT) void radix_hist_kernel(int pass, int P, int nblk, const float* __restrict__ depth,
                                                        const int32_t* __restrict__ radii, const uint32_t* __restrict__ keys_in,
                                                        const uint32_t* __restrict__ n_vis, uint32_t* __restrict__ hist) {
  __shared__ uint32_t h[256];
  const int view = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const size_t base = (size_t)view * P;
  const int limit = pass == 0 ? P : (int)n_vis[view];
  if (tid < 256) h[tid] = 0;
  __syncthreads();
  if (blk * (ITEMS * NT) < limit) {
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const int idx = blk * (ITEMS * NT) + r * NT + tid;
      if (idx < limit) {
        const uint32_t k = radix_key(pass, P, idx, base, depth, radii, keys_in);
        if (k != 0xFFFFFFFFu) atomicAdd(&h[(k >> (8 * pass)) & 255u], 1u);
      }
    }
    __syncthreads();
  }
  if (tid < 256) hist[((size_t)view * nblk + blk) * 256 + tid] = h[tid];   // [view][block][digit]: coalesced
}

template <int NT, int ITEMS>
__global__ __launch_bounds__(NT) void radix_scatter_kernel(int pass, int P, int nblk, const float* __restrict__ depth,
                                                           const int32_t* __restrict__ radii, const uint32_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
                                                           uint32_t* __restrict__ vals_out, const uint32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ n_vis, const uint2* __restrict__ rect,
                                                           uint32_t* __restrict__ sorted_id, uint2* __restrict__ sorted_rect) {
  constexpr int NW = NT / 64;
  __shared__ uint32_t digit_base[256];
  __shared__ uint32_t wave_cnt[2][NW][256];   // double-buffered per round: counts, then exclusive prefixes over the waves
  __shared__ uint32_t wave_tot[4];
  const int view = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
  const int wave = tid >> 6;
  const uint32_t lane = u3d_lane_id();
  const size_t base = (size_t)view * P;
  const int limit = pass == 0 ? P : (int)n_vis[view];
  if (pass == 3) {
    // the last pass writes the sorted list itself (ids and their tile rectangles; no separate gather launch); positions past the
    // visible keys read id 0 / empty rectangle
    for (int idx = blk * (ITEMS * NT) + tid; idx < min(P, (blk + 1) * (ITEMS * NT)); idx += NT)
      if (idx >= limit) { sorted_id[base + idx] = 0u; sorted_rect[base + idx] = make_uint2(0u, 0u); }
  }
  if (blk * (ITEMS * NT) >= limit) return;   // whole workgroup past the visible keys (uniform)
  for (int e = tid; e < 2 * NW * 256; e += NT) (&wave_cnt[0][0][0])[e] = 0;
  {
    // global offset of (digit tid, this block) in digit-major / block-minor order, from the per-block counts
    // hist[view][b][digit] (every block redoes this small scan: no separate scan launch between histogram and scatter;
    // 8 loads in flight: the column walk is latency-bound and was most of this kernel's time)
    uint32_t tot = 0, before = 0, inc = 0;
    if (tid < 256) {
      const uint32_t* col = hist + (size_t)view * nblk * 256 + tid;
      int b = 0;
      for (; b + 7 < nblk; b += 8) {
        uint32_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = col[(size_t)(b + u) * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) { tot += c[u]; before += b + u < blk ? c[u] : 0u; }
      }
      for (; b < nblk; ++b) { const uint32_t c0 = col[(size_t)b * 256]; tot += c0; before += b < blk ? c0 : 0u; }
      inc = tot;   // inclusive scan of the 256 digit totals: shuffles inside each of the four waves, then the wave totals
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
      if (pass == 0 && blk == 0 && tid == 255) n_vis[view] = off + inc;   // total of the visible keys of this view
      digit_base[tid] = off + inc - tot + before;
    }
    __syncthreads();
  }
  for (int r = 0; r < ITEMS; ++r) {
    const int idx = blk * (ITEMS * NT) + r * NT + tid;
    bool valid = idx < limit;
    uint32_t k = 0, v = 0, digit = 0;
    if (valid) {
      k = radix_key(pass, P, idx, base, depth, radii, keys_in);
      v = pass == 0 ? (uint32_t)idx : vals_in[base + idx];
      digit = (k >> (8 * pass)) & 255u;
      valid = k != 0xFFFFFFFFu;          // (only pass 0 meets culled entries)
    }
    // lanes of this wave holding the same digit (stable multi-split via 8 ballots)
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (digit >> b) & 1u;
      const unsigned long long m = __ballot(bit && valid);
      same &= bit ? m : ~m;
    }
    const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull));
    uint32_t (*cnt)[256] = wave_cnt[r & 1];
    if (valid && rank == 0) cnt[wave][digit] = (uint32_t)__popcll(same);
    __syncthreads();
    if (tid < 256) {   // counts -> destination of each wave's first key of this digit; digit_base moves past the round
      uint32_t run = digit_base[tid];
#pragma unroll
      for (int w = 0; w < NW; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = run; run += c; }
      digit_base[tid] = run;
    }
    __syncthreads();
    if (valid) {
      const uint32_t dst = cnt[wave][digit] + rank;
      if (pass == 3) {
        sorted_id[base + dst] = v;
        sorted_rect[base + dst] = rect[base + v];
      } else {
        keys_out[base + dst] = k;
        vals_out[base + dst] = v;
      }
    }
    for (int e = tid; e < NW * 256; e += NT) (&wave_cnt[(r + 1) & 1][0][0])[e] = 0;   // the other buffer, for the next round
    __syncthreads();
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
        hipLaunchKernelGGL(depth_sort_block_radix_kernel<1024>, dim3(NV), dim3(1024), (size_t)4 * N * sizeof(uint32_t), s, d.P, N, b.depth,
                           radii, b.rect, b.sorted_id, b.sorted_rect, b.n_vis);
      else
        hipLaunchKernelGGL(depth_sort_block_radix_kernel<256>, dim3(NV), dim3(256), (size_t)4 * N * sizeof(uint32_t), s, d.P, N, b.depth,
                           radii, b.rect, b.sorted_id, b.sorted_rect, b.n_vis);
    }
    return;
  }
  const int tile = u3d_radix_tile(d.P);
  const int nblk = (d.P + tile - 1) / tile;
  for (int pass = 0; pass < 4; ++pass) {
    const uint32_t* kin = pass == 0 ? nullptr : b.sort_keys[(pass + 1) & 1];
    const uint32_t* vin = pass == 0 ? nullptr : b.sort_vals[(pass + 1) & 1];
    uint32_t* kout = b.sort_keys[pass & 1];
    uint32_t* vout = b.sort_vals[pass & 1];
#define LAUNCH(NT, IT)                                                                                                       \
  do {                                                                                                                       \
    hipLaunchKernelGGL((radix_hist_kernel<NT, IT>), dim3(nblk, NV), dim3(NT), 0, s, pass, d.P, nblk, b.depth, radii, kin,    \
                       b.n_vis, b.sort_hist);                                                                                \
    hipLaunchKernelGGL((radix_scatter_kernel<NT, IT>), dim3(nblk, NV), dim3(NT), 0, s, pass, d.P, nblk, b.depth, radii, kin, \
                       vin, kout, vout, b.sort_hist, b.n_vis, b.rect, b.sorted_id, b.sorted_rect);                           \
  } while (0)
    if (d.P <= 65536) LAUNCH(U3D_RADIX_NT_SMALL, U3D_RADIX_IT_SMALL); else LAUNCH(U3D_RADIX_NT_LARGE, U3D_RADIX_IT_LARGE);
#undef LAUNCH
  }
}
