// Round 5: the index stage of batches beyond 1 M keys (the 16x batch of SURVEY 8(d): 5.8 M keys) on the partitioned path.
// Included by fused_fwd.hip.  Restates (reference, corelib/dynamicemb/): segmented_unique_cuda (src/unique_op.cu:484-714) -- whose
// one code path serves any batch size.
//
// Why the lists of path (c) do not scale as they are: a tile reserves its records in the partitions' lists with one returning device
// atomic per (tile, partition) it touches -- tiles x min(P, pairs per tile) of them.  At 5.8 M keys that is 2 816 tiles x ~1 100
// partitions = 3.1 M atomics, as many as the per-slot counters of path (b) cost (they resolve at the memory side, ~10 G/s: 350 us).
// Here (DESIGN 6b of round 3, built):
//   1. probe_c_kernel<kStage> writes a tile's records TILE-MAJOR -- dense behind the tile's first position -- and the tile's record
//      count: no histogram, no reservation, no atomic;
//   2. split_records_kernel: a block takes a run of consecutive tiles (~16 K records), counts them per partition in LDS, reserves
//      with ONE atomic per (block, partition) -- blocks x P, a tenth of the per-tile count -- and moves the records into the
//      partitions' lists (L2-hot second pass); every staged record leaves a forwarding entry, which is what the per-occurrence
//      references of the probe kernel (late rows of the gather, lazily materialised reverse indices) go through;
//   3. fused_part3_kernel<4096>: the partition kernel of path (c) over lists of 4 096 records (P = keys / 2 816, up to 4 096).
#pragma once

namespace mi355 {

constexpr int kSplitThreads = 1024;

__device__ __forceinline__ int part_of_code(const FusedArgs& a, int z, int cshift) {
  // slot code of a record: a global slot, S (no slot: the last partition), or -(bucket + 2) (bucket full: deferred to the partition kernel)
  if (z >= 0) return z < a.S ? (int)((uint32_t)(z >> cshift) / (uint32_t)(a.spp >> cshift)) : a.P - 1;
  return (int)((uint32_t)(-z - 2) / (uint32_t)(a.spp >> cshift));
}

__global__ void __launch_bounds__(kSplitThreads) split_records_kernel(FusedArgs a, int tiles_per_block, int ntiles) {
  __shared__ int s_h[kPartMaxBig];      // records of this block per partition, then the running position inside the block's share
  __shared__ int s_base[kPartMaxBig];   // start of the block's share in the partition's sub-list
  const int tid = (int)threadIdx.x;
  const int t0 = (int)blockIdx.x * tiles_per_block;
  const int t1 = t0 + tiles_per_block < ntiles ? t0 + tiles_per_block : ntiles;
  const int cshift = __builtin_ctzll((unsigned long long)a.t.C);
  const int sub = (int)blockIdx.x % kPartSub;
  const int subcap = a.cap / kPartSub;
  for (int p = tid; p < a.P; p += kSplitThreads) s_h[p] = 0;
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int c = a.tile_cnt[t];
    const int64_t base = (int64_t)t * a.tl;
    for (int r = tid; r < c; r += kSplitThreads) atomicAdd(&s_h[part_of_code(a, (int)a.stage_rec[base + r].z, cshift)], 1);
  }
  __syncthreads();
  for (int p = tid; p < a.P; p += kSplitThreads) {
    const int c = s_h[p];
    s_base[p] = c ? atomicAdd(&a.pcount[p * kPartSub + sub], c) : 0;
    s_h[p] = 0;
  }
  __syncthreads();
  for (int t = t0; t < t1; ++t) {
    const int c = a.tile_cnt[t];
    const int64_t base = (int64_t)t * a.tl;
    for (int r = tid; r < c; r += kSplitThreads) {
      const uint4 rec = a.stage_rec[base + r];
      const int pk = part_of_code(a, (int)rec.z, cshift);
      const int idx = s_base[pk] + atomicAdd(&s_h[pk], 1);
      int ref = -1;
      if (idx < subcap) {
        ref = pk * a.cap + sub * subcap + idx;
        a.rec[ref] = rec;
      } else {
        a.hdr[a.ovf_word] = a.ovf_val;   // a partition received more records than it can hold: the step is flagged (see the module)
      }
      a.fwd[base + r] = ref;
    }
  }
}

}  // namespace mi355
