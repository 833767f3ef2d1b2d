// Block / wave scan helpers shared by index_ops.hip and fused_fwd.hip.
#pragma once
#include "common.h"

namespace mi355 {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 4;
constexpr int kScanTile = kScanThreads * kScanItems;  // 1024

// Inclusive scan over the 64 lanes of a wave on the DPP path of the VALU (round 5): row_shr 1 / 2 / 4 / 8 inside the 16-lane rows,
// then row_bcast:15 / :31 across them -- six v_add_u32_dpp instructions.  The __shfl_up form it replaces compiled to six dependent
// ds_bpermute_b32 round trips through the LDS crossbar (~100+ cycles each): the block scans of the index kernels were their
// longest phases (profiles/r03_index_phase_stamps.txt: "entry scan + publish sums" 11 K cycles).
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1 (lanes without a source add 0)
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
  return v;
}
// running maximum over the lanes of a wave, same instruction pattern
__device__ __forceinline__ int wave_incl_max(int v) {
  constexpr int kNeg = -2147483647 - 1;
  int o;
  o = __builtin_amdgcn_update_dpp(kNeg, v, 0x111, 0xf, 0xf, false); v = o > v ? o : v;
  o = __builtin_amdgcn_update_dpp(kNeg, v, 0x112, 0xf, 0xf, false); v = o > v ? o : v;
  o = __builtin_amdgcn_update_dpp(kNeg, v, 0x114, 0xf, 0xf, false); v = o > v ? o : v;
  o = __builtin_amdgcn_update_dpp(kNeg, v, 0x118, 0xf, 0xf, false); v = o > v ? o : v;
  o = __builtin_amdgcn_update_dpp(kNeg, v, 0x142, 0xa, 0xf, false); v = o > v ? o : v;
  o = __builtin_amdgcn_update_dpp(kNeg, v, 0x143, 0xc, 0xf, false); v = o > v ? o : v;
  return v;
}
// sum over the wave, returned to every lane
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_incl_scan(v), 63); }

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
