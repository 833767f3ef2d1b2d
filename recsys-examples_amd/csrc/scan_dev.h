// Block / wave scan helpers shared by index_ops.hip and fused_fwd.hip.
#pragma once
#include "common.h"

namespace mi355 {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanThreads * kScanItems;  // 1024

__device__ __forceinline__ int wave_incl_scan(int v) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int o = __shfl_up(v, off, 64);
    if (lane_id() >= off) v += o;
  }
  return v;
}

// exclusive scan of one int per thread across a 256-thread block; returns the block total in `total`
__device__ __forceinline__ int block_excl_scan(int v, int& total) {
  __shared__ int s_w[kScanThreads / 64 + 1];
  const int w = threadIdx.x >> 6;
  int incl = wave_incl_scan(v);
  if (lane_id() == 63) s_w[w] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < kScanThreads / 64; ++k) {
    int x = s_w[k];
    if (k < w) base += x;
    tot += x;
  }
  __syncthreads();
  total = tot;
  return base + incl - v;
}

// Tile prefix without a scan launch: the block sums the per-tile counts of all earlier tiles itself.  With a few
// hundred tiles that is one or two loads per thread, cheaper than the ~5 us a dependent one-block kernel costs in the
// launch chain.  (Callers fall back to scan_partials_kernel above kSelfPrefixMaxTiles.)
constexpr int64_t kSelfPrefixMaxTiles = 4096;
__device__ __forceinline__ int self_prefix(const int* __restrict__ partial, int b) {
  int s = 0;
  for (int j = threadIdx.x; j < b; j += kScanThreads) s += partial[j];
  int tot;
  block_excl_scan(s, tot);
  return tot;
}

__device__ __forceinline__ int upper_bound_i64(const int64_t* __restrict__ a, int n, int64_t x) {
  int lo = 0, hi = n;  // first index with a[idx] > x
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (a[mid] <= x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

}  // namespace mi355
