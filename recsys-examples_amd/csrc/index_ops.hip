// Index plumbing of the DynamicEmb lookup path for gfx950: segmented unique (dedup),
// table-id expansion, table ranges, order-preserving stream compaction, the block-aggregated
// counting sort that groups the keys of a batch by unique row (CSR for the backward pass) and
// the row-wise key -> rank bucketize that precedes the all-to-all.
//
// Replaces (reference, corelib/dynamicemb/src/): segmented_unique_cuda + expand_table_ids_cuda
// (unique_op.cu:209-750), get_table_range / flagged_compact (index_calculation.cu:77-232),
// generate_gather_ids + cub radix sort of reduce_grads (dynamic_emb_op.cu:140-263),
// block_bucketize_sparse_features (sparse_block_bucketize_features.cu:220-350).
//
// MI355X design notes
//  * Dedup is a per-table open-addressing set of int32 input positions in HBM (2 slots per
//    key).  The representative of a key is made the MINIMUM input position with atomicMin, so
//    the unique order is first-occurrence order: deterministic, unlike the schedule-dependent
//    order of the reference.  Hot (Zipf) keys cost no atomic traffic: a lane only issues the
//    atomicMin when its position is below the value it just read (values only decrease).
//  * unique ids come from an exclusive scan of the "is first occurrence" flags (block scan with
//    wave64 shuffles + one small partial pass), so no counts cross to the host.
//  * grouping keys by unique row uses LDS-privatised counters: each 1024-key tile counts its
//    rows in an LDS hash table and issues ONE global atomic per distinct row per tile, so the
//    hottest row of a Zipf stream costs (#tiles) atomics instead of (#occurrences).
#include "common.h"
#include "hot.h"
#include "stamps.h"
#include "../../include/recsys_amd.h"
#include "internal.h"
#include "scan_dev.h"

namespace mi355 {

// ---------------------------------------------------------------------------------------------
// segmented unique
// ---------------------------------------------------------------------------------------------
struct UniqWs {
  int* slots;    // [2n]   open-addressing set, value = representative (smallest) input position, -1 empty
  int* gcnt;     // [2n]   per slot: occurrences - 1 (starts at -1, same fill byte as slots); later the unique id
  int* rep;      // [n]    slot of key i
  int* partial;  // [nb+1] per-tile counts of first occurrences, then exclusive offsets
  int* total;    // [1]
};

// Clears the scratch set / counters (0xFF bytes) and, when asked, derives the table ranges from the bag offsets
// (get_table_range_kernel's job) in the same launch.
__global__ void __launch_bounds__(256)
uniq_prepare_kernel(uint4* __restrict__ fill, int64_t n16, const int64_t* __restrict__ offsets,
                    const int64_t* __restrict__ feature_offsets, int T, int64_t feature_x_batch, int64_t* __restrict__ range) {
  const uint4 ff = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) fill[i] = ff;
  if (blockIdx.x == 0 && range) {
    const int64_t nfeat = feature_offsets[T];
    const int64_t B = nfeat > 0 ? feature_x_batch / nfeat : 0;
    for (int t = threadIdx.x; t <= T; t += blockDim.x) range[t] = offsets[feature_offsets[t] * B];
  }
}

// One kUniqTile-key tile per block.  Keys are first de-duplicated INSIDE the tile in an LDS hash set (block
// representative = smallest position), then only the tile representatives touch the global set.  Under a
// Zipf stream this turns (#occurrences) same-address CAS/atomicMin operations on a hot key -- which
// serialise at ~12 ns each in the L2 atomic unit -- into at most (#tiles).
#ifndef UNIQ_TILE
#define UNIQ_TILE 2048
#endif
#ifndef UNIQ_THREADS
#define UNIQ_THREADS 512
#endif
constexpr int kUniqTile = UNIQ_TILE;        // keys per block: the hottest key costs one serialised global atomic per TILE
constexpr int kUniqLds = 2 * kUniqTile;     // LDS set slots
constexpr int kUniqThreads = UNIQ_THREADS;
constexpr int kUniqPer = kUniqTile / kUniqThreads;
__global__ void __launch_bounds__(kUniqThreads)
uniq_insert_kernel(const uint64_t* __restrict__ keys, int64_t n, const int64_t* __restrict__ seg, int T, UniqWs ws,
                   int* __restrict__ csr_rank) {
  __shared__ uint64_t s_key[kUniqTile];
  __shared__ int s_tab[kUniqLds];   // tile-local position of the representative (-1 empty); after the global insert:
                                    // the key's global slot
  __shared__ int s_cnt[kUniqLds];   // occurrences of the key inside the tile; after the global insert: occurrences of
                                    // the key in the tiles that got to the global counter first
  __shared__ uint16_t s_t[kUniqTile];  // table of each key of the tile (T <= 65535, checked by the launcher)
  const int64_t tile0 = (int64_t)blockIdx.x * kUniqTile;
  for (int s = threadIdx.x; s < kUniqLds; s += kUniqThreads) { s_tab[s] = -1; s_cnt[s] = 0; }
  int hh[kUniqPer], rk[kUniqPer];
#pragma unroll
  for (int q = 0; q < kUniqPer; ++q) {
    const int li = q * kUniqThreads + threadIdx.x;
    const int64_t i = tile0 + li;
    if (i < n) { s_key[li] = keys[i]; s_t[li] = (uint16_t)(upper_bound_i64(seg, T + 1, i) - 1); }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kUniqPer; ++q) {
    const int li = q * kUniqThreads + threadIdx.x;
    hh[q] = -1;
    if (tile0 + li < n) {
      const uint64_t key = s_key[li];
      const int t = s_t[li];
      int h = (int)(fmix64(key + 0x9E3779B97F4A7C15ull * (uint64_t)t) >> 40) & (kUniqLds - 1);
      while (true) {
        int cur = atomicCAS(&s_tab[h], -1, li);
        if (cur == -1) break;
        if (s_key[cur] == key && s_t[cur] == t) { atomicMin(&s_tab[h], li); break; }
        h = (h + 1) & (kUniqLds - 1);
      }
      hh[q] = h;
      rk[q] = atomicAdd(&s_cnt[h], 1);   // rank of this occurrence inside the tile (arbitrary but unique)
    }
  }
  __syncthreads();
  bool isrep[kUniqPer];
#pragma unroll
  for (int q = 0; q < kUniqPer; ++q) isrep[q] = hh[q] >= 0 && s_tab[hh[q]] == q * kUniqThreads + (int)threadIdx.x;
  __syncthreads();   // s_tab / s_cnt change meaning below
#pragma unroll
  for (int q = 0; q < kUniqPer; ++q) {
    const int li = q * kUniqThreads + threadIdx.x;
    if (isrep[q]) {  // tile representative: insert into the global per-table set
      const int64_t i = tile0 + li;
      const uint64_t key = s_key[li];
      const int t = s_t[li];
      const int64_t lo = seg[t], len = seg[t + 1] - lo;
      const uint64_t range = 2ull * (uint64_t)len;
      const int64_t base = 2 * lo;
      const uint64_t h = fmix64(key);
      int64_t off = (int64_t)((((h >> 32) ^ (h & 0xffffffffull)) * range) >> 32);
      int64_t pos = base + off;
      const int me = (int)i;
      while (true) {
        int cur = ws.slots[pos];
        if (cur == -1) {
          int old = atomicCAS(&ws.slots[pos], -1, me);
          if (old == -1) break;
          cur = old;
        }
        if (keys[cur] == key) {
          if (me < cur) atomicMin(&ws.slots[pos], me);
          break;
        }
        ++off;
        if ((uint64_t)off == range) off = 0;
        pos = base + off;
      }
      // one counter update per distinct key per tile; the value before it places this tile's occurrences in
      // the key's list (counter starts at -1: see UniqWs::gcnt)
      const int cnt = s_cnt[hh[q]];
      s_tab[hh[q]] = (int)pos;
      s_cnt[hh[q]] = atomicAdd(&ws.gcnt[pos], cnt) + 1;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kUniqPer; ++q) {
    const int li = q * kUniqThreads + threadIdx.x;
    if (hh[q] >= 0) {
      ws.rep[tile0 + li] = s_tab[hh[q]];
      if (csr_rank) csr_rank[tile0 + li] = s_cnt[hh[q]] + rk[q];
    }
  }
}

// count first occurrences (slots[rep[i]] == i) per 1024-tile
__global__ void __launch_bounds__(kScanThreads)
uniq_flag_kernel(int64_t n, UniqWs ws) {
  const int64_t tile0 = (int64_t)blockIdx.x * kScanTile;
  int c = 0;
  int rp[kScanItems], sl[kScanItems];   // branch-free (clamped) hops: the items' chains overlap
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    rp[k] = ws.rep[i < n ? i : n - 1];
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) sl[k] = ws.slots[rp[k]];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    c += (i < n) && (sl[k] == (int)i);
  }
  int tot;
  block_excl_scan(c, tot);
  if (threadIdx.x == 0) ws.partial[blockIdx.x] = tot;
}

// single block: exclusive scan of `partial[0..nb)` in place, total -> *total
__device__ __forceinline__ void scan_partials_body(int* partial, int64_t nb, int* total) {
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < nb; b0 += kScanThreads) {
    int64_t b = b0 + threadIdx.x;
    int v = b < nb ? partial[b] : 0;
    int tot;
    int ex = block_excl_scan(v, tot);
    int carry = s_carry;
    if (b < nb) partial[b] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_carry;
}
__global__ void __launch_bounds__(kScanThreads) scan_partials_kernel(int* partial, int64_t nb, int* total) {
  scan_partials_body(partial, nb, total);
}
template <bool kFreq, bool kSelf>
__global__ void __launch_bounds__(kScanThreads)
uniq_emit_kernel(const uint64_t* __restrict__ keys, int64_t n, const int64_t* __restrict__ seg, int T, UniqWs ws,
                 uint64_t* __restrict__ unique_keys, int64_t* __restrict__ table_offsets, int64_t* __restrict__ freq,
                 const int64_t* __restrict__ in_freq, int* __restrict__ csr_cnt) {
  const int64_t tile0 = (int64_t)blockIdx.x * kScanTile;
  int f[kScanItems];
  int c = 0;
  int rp[kScanItems], sl[kScanItems];   // branch-free (clamped) hops: the items' chains overlap
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    rp[k] = ws.rep[i < n ? i : n - 1];
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) sl[k] = ws.slots[rp[k]];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    f[k] = (i < n) && (sl[k] == (int)i);
    c += f[k];
  }
  int tot;
  const int pre = kSelf ? self_prefix(ws.partial, blockIdx.x) : ws.partial[blockIdx.x];
  int ex = block_excl_scan(c, tot) + pre;
  if (kSelf && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {   // the last tile knows the number of uniques
    const int64_t total = pre + tot;
    *ws.total = (int)total;
    for (int t = T; t >= 0 && seg[t] >= n; --t) table_offsets[t] = total;
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    if (i < n) {
      // table boundaries: every table t with seg[t] == i starts at the exclusive count here
      int t = upper_bound_i64(seg, T + 1, i) - 1;
      while (t >= 0 && seg[t] == i) { table_offsets[t] = ex; --t; }
      if (f[k]) {
        unique_keys[ex] = keys[i];
        const int slot = ws.rep[i];
        const int occ = ws.gcnt[slot] + 1;   // only this thread reads the slot's counter: it may now hold the id
        ws.gcnt[slot] = ex;
        if (csr_cnt) csr_cnt[ex] = occ;
        if (kFreq) freq[ex] = in_freq ? 0 : occ;
        ++ex;
      }
    }
  }
  if (!kSelf && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t total = *ws.total;
    for (int t = T; t >= 0 && seg[t] >= n; --t) table_offsets[t] = total;
  }
}

template <bool kFreq>
__global__ void __launch_bounds__(256)
uniq_finish_kernel(int64_t n, UniqWs ws, const int64_t* __restrict__ in_freq, int64_t* __restrict__ output_indices,
                   int64_t* __restrict__ freq, const int64_t* __restrict__ table_offsets, int T, int64_t* __restrict__ table_ids) {
  const int64_t nu = table_ids ? *ws.total : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < nu) table_ids[i] = upper_bound_i64(table_offsets, T + 1, i) - 1;   // fused expand_table_ids
    const int u = ws.gcnt[ws.rep[i]];
    output_indices[i] = u;
    if (kFreq && in_freq) atomicAdd((unsigned long long*)&freq[u], (unsigned long long)in_freq[i]);
  }
}

__global__ void zero_offsets_kernel(int64_t* p, int64_t n) {
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}

// expand_table_ids_kernel (unique_op.cu:471-480): table_ids[i] = upper_bound(offsets, i) - 1
__global__ void __launch_bounds__(256)
expand_table_ids_kernel(const int64_t* __restrict__ offsets, int T, int64_t n, const int64_t* __restrict__ n_dev,
                        int64_t* __restrict__ table_ids) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    table_ids[i] = upper_bound_i64(offsets, T + 1, i) - 1;
}

// get_table_range_kernel (index_calculation.cu:78-91)
__global__ void get_table_range_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ feature_offsets,
                                       int T, int64_t feature_x_batch, int64_t* __restrict__ range) {
  const int64_t nfeat = feature_offsets[T];
  const int64_t B = nfeat > 0 ? feature_x_batch / nfeat : 0;
  for (int t = threadIdx.x; t <= T; t += blockDim.x) range[t] = offsets[feature_offsets[t] * B];
}

// ---------------------------------------------------------------------------------------------
// order-preserving compaction (flagged_compact, index_calculation.cu:129-232) without the host
// sync on the count: count stays in `*count_out` on the device.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kScanThreads)
compact_count_kernel(const uint8_t* __restrict__ flags, int64_t n, const int64_t* __restrict__ n_dev, int* partial) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int64_t tile0 = (int64_t)blockIdx.x * kScanTile;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    c += (i < n) && flags[i];
  }
  int tot;
  block_excl_scan(c, tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

struct CompactArrays {
  const int64_t* in[6];
  int64_t* out[6];
  int num;
};

__global__ void __launch_bounds__(kScanThreads)
compact_emit_kernel(const uint8_t* __restrict__ flags, int64_t n, const int64_t* __restrict__ n_dev,
                    const int* __restrict__ partial, const int* __restrict__ total, int64_t* __restrict__ count_out,
                    int64_t* __restrict__ out_index, CompactArrays arrs) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int64_t tile0 = (int64_t)blockIdx.x * kScanTile;
  int f[kScanItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    f[k] = (i < n) && flags[i];
    c += f[k];
  }
  int tot;
  int ex = block_excl_scan(c, tot) + partial[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    if (f[k]) {
      if (out_index) out_index[ex] = i;
      for (int a = 0; a < arrs.num; ++a) arrs.out[a][ex] = arrs.in[a][i];
      ++ex;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *count_out = *total;
}

// ---------------------------------------------------------------------------------------------
// group the keys of a pooled / sequence batch by unique row: histogram + exclusive scan + scatter
// (a counting sort keyed by the reverse index).  csr_src[p] = source grad row of the p-th entry:
// pooled: slot f*B+b -> b*F+f is NOT used here; we store the BAG id (f*B+b) and let the consumer
// derive (b, f); sequence: the key position j.
// ---------------------------------------------------------------------------------------------
constexpr int kHistTile = 1024;  // keys per block
constexpr int kHistSlots = 2048; // LDS hash slots

struct LdsEntry { int u; int cnt; int base; };

__device__ __forceinline__ int lds_find_or_add(LdsEntry* tab, int u) {
  int h = (int)(((uint32_t)u * 2654435761u) >> 21) & (kHistSlots - 1);
  while (true) {
    int cur = atomicCAS(&tab[h].u, -1, u);
    if (cur == -1 || cur == u) return h;
    h = (h + 1) & (kHistSlots - 1);
  }
}

// pass 1: cnt[u] += occurrences (one global atomic per distinct row per tile)
__global__ void __launch_bounds__(256)
csr_hist_kernel(const int64_t* __restrict__ rev, int64_t n, int* __restrict__ cnt) {
  __shared__ LdsEntry tab[kHistSlots];
  for (int s = threadIdx.x; s < kHistSlots; s += blockDim.x) { tab[s].u = -1; tab[s].cnt = 0; }
  __syncthreads();
  const int64_t tile0 = (int64_t)blockIdx.x * kHistTile;
  for (int k = threadIdx.x; k < kHistTile; k += blockDim.x) {
    int64_t j = tile0 + k;
    if (j < n) { int h = lds_find_or_add(tab, (int)rev[j]); atomicAdd(&tab[h].cnt, 1); }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < kHistSlots; s += blockDim.x)
    if (tab[s].u >= 0) atomicAdd(&cnt[tab[s].u], tab[s].cnt);
}

// exclusive scan of cnt[0..nu) -> ptr[0..nu], nu read from the device
__global__ void __launch_bounds__(kScanThreads)
scan_reduce_kernel(const int* __restrict__ in, int64_t n, const int64_t* __restrict__ n_dev, int* partial, int* c0 = nullptr,
                   int* c1 = nullptr) {
  if (c0 && blockIdx.x == 0 && threadIdx.x == 0) { *c0 = 0; *c1 = 0; c0[4] = 0; }   // hot-list counters n_hot, n_tasks, n_wave (hot.h header)
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int64_t tile0 = (int64_t)blockIdx.x * kScanTile;
  int c = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    if (i < n) c += in[i];
  }
  int tot;
  block_excl_scan(c, tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
template <bool kSelf>
__global__ void __launch_bounds__(kScanThreads)
scan_down_kernel(const int* __restrict__ in, int64_t n, const int64_t* __restrict__ n_dev, const int* __restrict__ partial,
                 int* __restrict__ total, int* __restrict__ out, HotList hot, bool build_hot) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int64_t tile0 = (int64_t)blockIdx.x * kScanTile;
  int v[kScanItems];
  int c = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    v[k] = i < n ? in[i] : 0;
    c += v[k];
  }
  int tot;
  const int pre = kSelf ? self_prefix(partial, blockIdx.x) : partial[blockIdx.x];
  int ex = block_excl_scan(c, tot) + pre;
  if (kSelf && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { *total = pre + tot; out[n] = pre + tot; }
  // hot rows of this tile: ids and task ranges are reserved with ONE atomic pair per block
  // (rows with khot < count <= kwave go to the one-wave list, longer ones to the chunked block tasks)
  int nh_local = 0, nt_local = 0, nw_local = 0;
  if (build_hot) {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (v[k] > hot.khot && v[k] <= hot.kwave) ++nw_local;
      else if (v[k] > hot.khot) { ++nh_local; nt_local += (v[k] + hot.kchunk - 1) / hot.kchunk; }
    }
  }
  __shared__ int s_hbase[3];
  int h_ex = 0, t_ex = 0, w_ex = 0;
  if (build_hot) {
    int th, tt, tw;
    h_ex = block_excl_scan(nh_local, th);
    t_ex = block_excl_scan(nt_local, tt);
    w_ex = block_excl_scan(nw_local, tw);
    if (threadIdx.x == 0) {
      s_hbase[0] = th ? atomicAdd(hot.n_hot, th) : 0;
      s_hbase[1] = tt ? atomicAdd(hot.n_tasks, tt) : 0;
      s_hbase[2] = tw ? atomicAdd(hot.n_wave, tw) : 0;
    }
    __syncthreads();
    h_ex += s_hbase[0];
    t_ex += s_hbase[1];
    w_ex += s_hbase[2];
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    int64_t i = tile0 + threadIdx.x * kScanItems + k;
    if (i < n) {
      out[i] = ex;
      if (build_hot && v[k] > hot.khot && v[k] <= hot.kwave) {
        const int w = w_ex++;
        if (w < hot.max_hot) { hot.wave_u[w] = (int)i; hot.wave_lo[w] = ex; hot.wave_cnt[w] = v[k]; }
      } else if (build_hot && v[k] > hot.khot) {
        const int nch = (v[k] + hot.kchunk - 1) / hot.kchunk;
        const int h = h_ex++, t0 = t_ex;
        t_ex += nch;
        if (h < hot.max_hot && t0 + nch <= hot.max_tasks) {  // tasks are written out by csr_fill_kernel
          hot.hot_done[h] = 0;
          hot.hot_nchunks[h] = nch;
          hot.hot_u[h] = (int)i;
          hot.hot_lo[h] = ex;
          hot.hot_cnt[h] = v[k];
          hot.hot_t0[h] = t0;
        }
      }
      ex += v[k];
    }
  }
  if (!kSelf && blockIdx.x == 0 && threadIdx.x == 0) out[n] = *total;
}

// pass 2: scatter.  src id of key j: pooled -> bag id, sequence -> j.  The bag of every key of the tile is
// resolved in LDS: two binary searches per TILE (first / last key), the heads of the bags that start
// inside the tile are marked, an inclusive max-scan spreads them -- instead of a 16-step global binary
// search per key.
__global__ void __launch_bounds__(256)
csr_fill_kernel(const int64_t* __restrict__ rev, int64_t n, const int64_t* __restrict__ offsets, int64_t num_bags,
                const int* __restrict__ ptr, int* __restrict__ cursor, int* __restrict__ csr_src, HotList hot, bool build_hot) {
  __shared__ LdsEntry tab[kHistSlots];
  if (build_hot) {
    // expand the hot rows registered by scan_down_kernel into wave tasks and clear their accumulators
    // (one wave per hot row, riding on this launch)
    int nh = *hot.n_hot;
    nh = nh < hot.max_hot ? nh : hot.max_hot;
    for (int h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); h < nh; h += gridDim.x * (blockDim.x >> 6)) {
      const int nch = hot.hot_nchunks[h], t0 = hot.hot_t0[h], lo = hot.hot_lo[h], cnt = hot.hot_cnt[h], u = hot.hot_u[h];
      if (t0 + nch > hot.max_tasks) continue;
      for (int cc = lane_id(); cc < nch; cc += 64) {
        hot.task_u[t0 + cc] = u;
        hot.task_h[t0 + cc] = h;
        hot.task_lo[t0 + cc] = lo + cc * hot.kchunk;
        const int hi = lo + (cc + 1) * hot.kchunk;
        hot.task_hi[t0 + cc] = hi < lo + cnt ? hi : lo + cnt;
      }
      for (int e = lane_id(); e < hot.dim; e += 64) hot.hot_acc[(int64_t)h * hot.dim + e] = 0.f;
    }
  }
  __shared__ int s_bag[kHistTile];
  __shared__ int s_range[2];
  __shared__ int s_wmax[4];
  for (int s = threadIdx.x; s < kHistSlots; s += blockDim.x) { tab[s].u = -1; tab[s].cnt = 0; }
  const int64_t tile0 = (int64_t)blockIdx.x * kHistTile;
  const int64_t tile_end = tile0 + kHistTile < n ? tile0 + kHistTile : n;
  if (offsets) {
    for (int k = threadIdx.x; k < kHistTile; k += blockDim.x) s_bag[k] = -1;
    if (threadIdx.x == 0 || threadIdx.x == 64) {
      const int64_t key = threadIdx.x == 0 ? tile0 : tile_end - 1;
      int lo = 0, hi = (int)num_bags;  // first idx with offsets[idx] > key
      while (lo < hi) { int mid = (lo + hi) >> 1; if (offsets[mid] <= key) lo = mid + 1; else hi = mid; }
      s_range[threadIdx.x == 0 ? 0 : 1] = lo - 1;
    }
  }
  __syncthreads();
  if (offsets) {
    const int b_lo = s_range[0], b_hi = s_range[1];
    for (int b = b_lo + threadIdx.x; b <= b_hi; b += blockDim.x) {
      const int64_t o0 = offsets[b], o1 = offsets[b + 1];
      if (o1 > o0) { const int64_t p = o0 > tile0 ? o0 - tile0 : 0; if (p < kHistTile) s_bag[p] = b; }
    }
    __syncthreads();
    // inclusive max-scan over the tile positions (4 consecutive positions per thread)
    int v[4];
    int m = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = s_bag[threadIdx.x * 4 + k]; m = v[k] > m ? v[k] : m; v[k] = m; }
    int incl = m;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int o = __shfl_up(incl, off, 64); if (lane_id() >= off) incl = o > incl ? o : incl; }
    if (lane_id() == 63) s_wmax[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = -1;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base = s_wmax[w] > base ? s_wmax[w] : base;
    int prev = __shfl_up(incl, 1, 64);
    if (lane_id() == 0) prev = -1;
    prev = prev > base ? prev : base;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_bag[threadIdx.x * 4 + k] = v[k] > prev ? v[k] : prev;
  }
  __syncthreads();
  int hh[kHistTile / 256], rk[kHistTile / 256];
#pragma unroll
  for (int q = 0; q < kHistTile / 256; ++q) {
    int64_t j = tile0 + q * 256 + threadIdx.x;
    hh[q] = -1;
    if (j < n) { hh[q] = lds_find_or_add(tab, (int)rev[j]); rk[q] = atomicAdd(&tab[hh[q]].cnt, 1); }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < kHistSlots; s += blockDim.x)
    if (tab[s].u >= 0) tab[s].base = ptr[tab[s].u] + atomicAdd(&cursor[tab[s].u], tab[s].cnt);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kHistTile / 256; ++q) {
    int64_t j = tile0 + q * 256 + threadIdx.x;
    if (hh[q] >= 0) csr_src[tab[hh[q]].base + rk[q]] = offsets ? s_bag[q * 256 + threadIdx.x] : (int)j;
  }
}

// pass 2 when the forward already ranked every occurrence inside its unique row (mi355_segmented_unique_csr):
// csr_src[ptr[rev[j]] + rank[j]] = src id of key j.  No atomics, no LDS hash; the bag resolution and the hot-row task
// expansion are those of csr_fill_kernel.
// kMode 1: the unique id of key j is uidmap[2 * slot[j]] (fused forward, fused_fwd.hip: pairs per slot) and is also written
// to rev_out[j].  kMode 2 (partitioned fused forward): slot[j] is the key's (tile, key) record; the record knows the unique
// id and the rank base of its tile inside the row's list (PartRefs); rank[j] is the rank inside the tile and is replaced by
// the full rank; keys whose slot was found late (deferred eviction) get their row address here.
STAMP_ARRAY(g_st_scatter, 2048, 6)
#define SST(ph) STAMP(g_st_scatter, 2048, 6, ph)
template <int kMode>
__global__ void __launch_bounds__(256)
csr_scatter_kernel(const int64_t* __restrict__ rev, int* __restrict__ rank, int64_t n, const int64_t* __restrict__ offsets,
                   int64_t num_bags, const int* __restrict__ ptr, int* __restrict__ csr_src, HotList hot, bool build_hot,
                   const int* __restrict__ slot, const int* __restrict__ uidmap, int64_t* __restrict__ rev_out,
                   int* __restrict__ hdr_reset = nullptr, PartRefs pr = PartRefs{}, const int* __restrict__ gate = nullptr,
                   int gate_val = 0, int* __restrict__ mark = nullptr) {
  constexpr bool kSlot = kMode == 1;
  if (gate && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gate_val) return;   // (grid uniform)
  if (mark && blockIdx.x == 0 && threadIdx.x == 0) *mark = 1;
  SST(0);
  // fused forward: the deferred-key count, barrier words and release flag of the table's aux header are cleared for the
  // next step here, behind the numbering kernel that read them
  if (hdr_reset && blockIdx.x == 0 && threadIdx.x < 64 && threadIdx.x != 5 && threadIdx.x != 6) hdr_reset[threadIdx.x] = 0;   // ([5]: sticky error flag, [6]: epoch of the last overflow under MI355_FUSED_OVERFLOW_RERUN)
  if (build_hot) {
    int nh = *hot.n_hot;
    nh = nh < hot.max_hot ? nh : hot.max_hot;
    for (int h = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); h < nh; h += gridDim.x * (blockDim.x >> 6)) {
      const int nch = hot.hot_nchunks[h], t0 = hot.hot_t0[h], lo = hot.hot_lo[h], cnt = hot.hot_cnt[h], u = hot.hot_u[h];
      if (t0 + nch > hot.max_tasks) continue;
      for (int cc = lane_id(); cc < nch; cc += 64) {
        hot.task_u[t0 + cc] = u;
        hot.task_h[t0 + cc] = h;
        hot.task_lo[t0 + cc] = lo + cc * hot.kchunk;
        const int hi = lo + (cc + 1) * hot.kchunk;
        hot.task_hi[t0 + cc] = hi < lo + cnt ? hi : lo + cnt;
      }
      for (int e = lane_id(); e < hot.dim; e += 64) hot.hot_acc[(int64_t)h * hot.dim + e] = 0.f;
    }
  }
  __shared__ int s_bag[kHistTile];
  __shared__ int s_range[2];
  __shared__ int s_wmax[4];
  const int64_t tile0 = (int64_t)blockIdx.x * kHistTile;
  const int64_t tile_end = tile0 + kHistTile < n ? tile0 + kHistTile : n;
  // The kernel is a chain of dependent loads, so everything that can start at once does: the index hops of the keys
  // (slot -> unique id -> row pointer; clamped, branch-free, the four chains of a thread overlap) are issued first and
  // run under the bag search below, which itself is 64-ary (one probe per lane: 3 rounds for 64 K bags, not 16).
  constexpr int NQ = kHistTile / 256;
  int64_t r[NQ];
  int rk[NQ], p[NQ];
  int rec[NQ], late[NQ];   // kMode 2
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    int64_t j = tile0 + q * 256 + threadIdx.x;
    j = j < n ? j : n - 1;
    if constexpr (kMode != 0) r[q] = slot[j]; else r[q] = rev[j];
    rk[q] = rank[j];
  }
  if constexpr (kMode == 2) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      rec[q] = (int)r[q];
      const int2 ro = pr.rec_out[rec[q] >= 0 ? rec[q] : 0];
      r[q] = ro.x;
      late[q] = ro.y < 0;
      rk[q] += late[q] ? ~ro.y : ro.y;
    }
  }
  if (offsets) {
    for (int k = threadIdx.x; k < kHistTile; k += blockDim.x) s_bag[k] = -1;
    if (threadIdx.x < 128) {
      const int64_t key = threadIdx.x < 64 ? tile0 : tile_end - 1;
      int lo = 0, hi = (int)num_bags;  // first idx with offsets[idx] > key, in [lo, hi]
      while (hi > lo) {
        const int step = (hi - lo + 63) >> 6;
        const int64_t pi = (int64_t)lo + (int64_t)(lane_id() + 1) * step - 1;
        const bool gt = pi >= hi ? true : offsets[pi] > key;
        const uint64_t gm = __ballot(gt);
        if (!gm) { lo = hi; break; }                     // every probe <= key: the answer is the upper end
        const int first = __ffsll((unsigned long long)gm) - 1;
        const int nlo = first ? lo + first * step : lo;
        const int64_t nhi = (int64_t)lo + (int64_t)(first + 1) * step - 1;
        lo = nlo;
        hi = nhi < hi ? (int)nhi : hi;
      }
      if (lane_id() == 0) s_range[threadIdx.x < 64 ? 0 : 1] = lo - 1;
    }
  }
  if constexpr (kSlot) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) r[q] = uidmap[2 * r[q]];   // {counter, unique id} pairs per slot
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) p[q] = ptr[r[q]];
  SST(1);
  if (offsets) {
    __syncthreads();
    SST(2);
    const int b_lo = s_range[0], b_hi = s_range[1];
    for (int b = b_lo + threadIdx.x; b <= b_hi; b += blockDim.x) {
      const int64_t o0 = offsets[b], o1 = offsets[b + 1];
      if (o1 > o0) { const int64_t pp = o0 > tile0 ? o0 - tile0 : 0; if (pp < kHistTile) s_bag[pp] = b; }
    }
    __syncthreads();
    int v[4];
    int m = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = s_bag[threadIdx.x * 4 + k]; m = v[k] > m ? v[k] : m; v[k] = m; }
    int incl = m;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { int o = __shfl_up(incl, off, 64); if (lane_id() >= off) incl = o > incl ? o : incl; }
    if (lane_id() == 63) s_wmax[threadIdx.x >> 6] = incl;
    __syncthreads();
    int base = -1;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base = s_wmax[w] > base ? s_wmax[w] : base;
    int prev = __shfl_up(incl, 1, 64);
    if (lane_id() == 0) prev = -1;
    prev = prev > base ? prev : base;
#pragma unroll
    for (int k = 0; k < 4; ++k) s_bag[threadIdx.x * 4 + k] = v[k] > prev ? v[k] : prev;
    __syncthreads();
  }
  SST(3);
  int keep_alive = 0;
#pragma unroll
  for (int q = 0; q < NQ; ++q) keep_alive += p[q] + rk[q];
  asm volatile("" :: "v"(keep_alive));
  SST(4);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int64_t j = tile0 + q * 256 + threadIdx.x;
    if (j < n) {
      if constexpr (kMode == 2) {
        if (rec[q] >= 0) {
          csr_src[p[q] + rk[q]] = offsets ? s_bag[q * 256 + threadIdx.x] : (int)j;
          rank[j] = rk[q];
          if (late[q]) pr.occ_addr[j] = pr.row_addr[r[q]];
        }
        rev_out[j] = r[q];
      } else {
        csr_src[p[q] + rk[q]] = offsets ? s_bag[q * 256 + threadIdx.x] : (int)j;
        if constexpr (kSlot) rev_out[j] = r[q];
      }
    }
  }
  SST(5);
}

__global__ void __launch_bounds__(256)
rev_from_slots_kernel(const int* __restrict__ slot, const int* __restrict__ uidmap, int64_t n, int64_t* __restrict__ rev,
                      int* __restrict__ hdr_reset = nullptr) {
  if (hdr_reset && blockIdx.x == 0 && threadIdx.x < 64 && threadIdx.x != 5 && threadIdx.x != 6) hdr_reset[threadIdx.x] = 0;   // ([5]: sticky error flag, [6]: epoch of the last overflow under MI355_FUSED_OVERFLOW_RERUN)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) rev[i] = uidmap[2 * (int64_t)slot[i]];
}

// partitioned fused forward without a backward workspace: reverse indices, full ranks and late row addresses only
__global__ void __launch_bounds__(256)
rev_from_records_kernel(const int* __restrict__ slot, PartRefs pr, int* __restrict__ rank, int64_t n, int64_t* __restrict__ rev,
                        int* __restrict__ hdr_reset) {
  if (hdr_reset && blockIdx.x == 0 && threadIdx.x < 64 && threadIdx.x != 5 && threadIdx.x != 6) hdr_reset[threadIdx.x] = 0;   // ([5]: sticky error flag, [6]: epoch of the last overflow under MI355_FUSED_OVERFLOW_RERUN)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int rc = slot[i];
    if (rc < 0) { rev[i] = -1; continue; }     // (no record: its list overflowed, the step reports no uniques)
    const int2 ro = pr.rec_out[rc];
    rev[i] = ro.x;
    rank[i] += ro.y < 0 ? ~ro.y : ro.y;
    if (ro.y < 0) pr.occ_addr[i] = pr.row_addr[ro.x];
  }
}

// ---------------------------------------------------------------------------------------------
// row-wise key -> rank routing before the key all-to-all (sparse_block_bucketize_features.cu:220-350)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void route(uint64_t idx, int dist, uint64_t blk, uint64_t W, uint64_t& p, uint64_t& nw) {
  if (dist == 1) { p = idx % W; nw = idx; }
  else if (dist == 2) { p = fmix64(idx) % W; nw = idx; }
  else if (idx < blk * W) { p = idx / blk; nw = idx % blk; }
  else { p = idx % W; nw = idx / W; }
}

// FBGEMM's two generalisations of the op (sparse_block_bucketize_features.cu:194-211, 262-292, 341-347): features with DIFFERENT
// batch sizes (fstart [F + 1]: first bag of every feature; the reference materialises a length -> feature array, here a bag
// finds its feature by a binary search of F + 1 cached words) and uneven shard boundaries (pos / pos_off: the sorted boundaries
// of every feature, concatenated; the rank is the last boundary <= idx, the dist types do not apply).
struct BktEx {
  const int64_t* fstart = nullptr;
  int F = 0;
  const int64_t* pos = nullptr;
  const int64_t* pos_off = nullptr;
};
__device__ __forceinline__ int64_t bkt_feature(const BktEx& x, int64_t bag, int64_t B) {
  if (!x.fstart) return bag / B;
  int lo = 0, hi = x.F;      // first f with fstart[f + 1] > bag
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (x.fstart[mid + 1] <= bag) lo = mid + 1; else hi = mid; }
  return lo < x.F ? lo : x.F - 1;
}
__device__ __forceinline__ void route_ex(const BktEx& x, int64_t f, uint64_t idx, int dist, uint64_t blk, uint64_t W, uint64_t& p, uint64_t& nw) {
  if (!x.pos) { route(idx, dist, blk, W, p, nw); return; }
  int64_t first = x.pos_off[f], last = x.pos_off[f + 1];
  while (first < last) { const int64_t mid = first + ((last - first) >> 1); if ((uint64_t)x.pos[mid] <= idx) first = mid + 1; else last = mid; }
  const uint64_t lb = (uint64_t)(first - x.pos_off[f] - 1);
  if (lb < W) { p = lb; nw = idx - (uint64_t)x.pos[x.pos_off[f] + (int64_t)lb]; }
  else { p = idx % W; nw = idx / W; }
}

// one wave per bag (HSTU bags are whole sequences: few bags, thousands of keys each): lane r owns
// the counter of destination rank r; ranks of a 64-key chunk are tallied with one ballot per rank.
__global__ void __launch_bounds__(256)
bucketize_count_kernel(int64_t FB, int64_t B, int W, const int64_t* __restrict__ offsets, const uint64_t* __restrict__ indices,
                       const int64_t* __restrict__ block_sizes, const int* __restrict__ dist_type, int64_t* __restrict__ new_lengths, BktEx x) {
  const int lane = lane_id();
  for (int64_t bag = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); bag < FB; bag += (int64_t)gridDim.x * (blockDim.x / 64)) {
    const int64_t f = bkt_feature(x, bag, B);
    const int dist = dist_type ? dist_type[f] : 0;
    const uint64_t blk = (uint64_t)block_sizes[f];
    const int64_t lo = offsets[bag], hi = offsets[bag + 1];
    for (int p0 = 0; p0 < W; p0 += 64) {
      int64_t mycnt = 0;
      for (int64_t j0 = lo; j0 < hi; j0 += 64) {
        const int64_t j = j0 + lane;
        uint64_t p = ~0ull, nw;
        if (j < hi) route_ex(x, f, indices[j], dist, blk, (uint64_t)W, p, nw);
        const int nr = W - p0 < 64 ? W - p0 : 64;
        for (int r = 0; r < nr; ++r) {
          int c = __popcll(__ballot(p == (uint64_t)(p0 + r)));
          if (lane == r) mycnt += c;
        }
      }
      if (p0 + lane < W) new_lengths[(int64_t)(p0 + lane) * FB + bag] = mycnt;
    }
  }
}

// one wave per bag; order of keys inside a (rank, bag) segment = order inside the bag, as the
// reference's one-thread-per-bag kernel2 produces.
__global__ void __launch_bounds__(256)
bucketize_scatter_kernel(int64_t FB, int64_t B, int W, const int64_t* __restrict__ offsets, const uint64_t* __restrict__ indices,
                         const int64_t* __restrict__ block_sizes, const int* __restrict__ dist_type,
                         const int64_t* __restrict__ new_offsets, uint64_t* __restrict__ new_indices,
                         int64_t* __restrict__ unbucketize_permute, const float* __restrict__ weights,
                         float* __restrict__ new_weights, BktEx x) {
  const int lane = lane_id();
  for (int64_t bag = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); bag < FB; bag += (int64_t)gridDim.x * (blockDim.x / 64)) {
    const int64_t f = bkt_feature(x, bag, B);
    const int dist = dist_type ? dist_type[f] : 0;
    const uint64_t blk = (uint64_t)block_sizes[f];
    const int64_t lo = offsets[bag], hi = offsets[bag + 1];
    for (int p0 = 0; p0 < W; p0 += 64) {
      int64_t cursor = p0 + lane < W ? new_offsets[(int64_t)(p0 + lane) * FB + bag] : 0;  // lane r: write head of rank p0+r
      const int nr = W - p0 < 64 ? W - p0 : 64;
      for (int64_t j0 = lo; j0 < hi; j0 += 64) {
        const int64_t j = j0 + lane;
        uint64_t p = ~0ull, nw = 0;
        if (j < hi) route_ex(x, f, indices[j], dist, blk, (uint64_t)W, p, nw);
        int64_t dst = -1;
        for (int r = 0; r < nr; ++r) {
          const uint64_t m = __ballot(p == (uint64_t)(p0 + r));
          const uint32_t clo = __shfl((int)(uint32_t)cursor, r, 64), chi = __shfl((int)(uint32_t)((uint64_t)cursor >> 32), r, 64);
          const int64_t head = (int64_t)(((uint64_t)chi << 32) | clo);
          if (p == (uint64_t)(p0 + r)) dst = head + __popcll(m & ((1ull << lane) - 1));
          if (lane == r) cursor += __popcll(m);
        }
        if (dst >= 0) {
          new_indices[dst] = nw;
          if (unbucketize_permute) unbucketize_permute[j] = dst;
          if (weights) new_weights[dst] = weights[j];
        }
      }
    }
  }
}

// Short-bag forms (many bags of a few keys: the C2 shape has 65 536 bags of ~5 keys): GL lanes per bag, 64 / GL bags per
// wave, same counting by ballot -- the group's bits of the wave ballot -- and the same key order inside a (rank, bag)
// segment as the wave-per-bag kernels above.
template <int GL>
__global__ void __launch_bounds__(256)
bucketize_count_short_kernel(int64_t FB, int64_t B, int W, const int64_t* __restrict__ offsets, const uint64_t* __restrict__ indices,
                             const int64_t* __restrict__ block_sizes, const int* __restrict__ dist_type,
                             int64_t* __restrict__ new_lengths, BktEx x) {
  constexpr int NG = 64 / GL;
  const int lane = lane_id(), grp = lane / GL, gl = lane % GL;
  const uint64_t gmask = ((1ull << GL) - 1);
  const int64_t bag = ((int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * NG + grp;
  const bool live = bag < FB;
  const int64_t bg = live ? bag : FB - 1;
  const int64_t f = bkt_feature(x, bg, B);
  const int dist = dist_type ? dist_type[f] : 0;
  const uint64_t blk = (uint64_t)block_sizes[f];
  const int64_t lo = offsets[bg], hi = live ? offsets[bg + 1] : lo;
  int64_t len = hi - lo, maxlen = len;
  for (int o = GL; o < 64; o <<= 1) { const int64_t t = __shfl_xor((int)maxlen, o, 64); maxlen = t > maxlen ? t : maxlen; }   // bag lengths fit 31 bits
  for (int p0 = 0; p0 < W; p0 += GL) {
    int64_t mycnt = 0;
    const int nr = W - p0 < GL ? W - p0 : GL;
    for (int64_t j0 = 0; j0 < maxlen; j0 += GL) {
      const int64_t j = lo + j0 + gl;
      uint64_t p = ~0ull, nw;
      if (j < hi) route_ex(x, f, indices[j], dist, blk, (uint64_t)W, p, nw);
      for (int r = 0; r < nr; ++r) {
        const int c = __popcll((__ballot(p == (uint64_t)(p0 + r)) >> (grp * GL)) & gmask);
        if (gl == r) mycnt += c;
      }
    }
    if (live && p0 + gl < W) new_lengths[(int64_t)(p0 + gl) * FB + bag] = mycnt;
  }
}

template <int GL>
__global__ void __launch_bounds__(256)
bucketize_scatter_short_kernel(int64_t FB, int64_t B, int W, const int64_t* __restrict__ offsets, const uint64_t* __restrict__ indices,
                               const int64_t* __restrict__ block_sizes, const int* __restrict__ dist_type,
                               const int64_t* __restrict__ new_offsets, uint64_t* __restrict__ new_indices,
                               int64_t* __restrict__ unbucketize_permute, const float* __restrict__ weights,
                               float* __restrict__ new_weights, BktEx x) {
  constexpr int NG = 64 / GL;
  const int lane = lane_id(), grp = lane / GL, gl = lane % GL;
  const uint64_t gmask = ((1ull << GL) - 1);
  const int64_t bag = ((int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * NG + grp;
  const bool live = bag < FB;
  const int64_t bg = live ? bag : FB - 1;
  const int64_t f = bkt_feature(x, bg, B);
  const int dist = dist_type ? dist_type[f] : 0;
  const uint64_t blk = (uint64_t)block_sizes[f];
  const int64_t lo = offsets[bg], hi = live ? offsets[bg + 1] : lo;
  int64_t maxlen = hi - lo;
  for (int o = GL; o < 64; o <<= 1) { const int64_t t = __shfl_xor((int)maxlen, o, 64); maxlen = t > maxlen ? t : maxlen; }
  for (int p0 = 0; p0 < W; p0 += GL) {
    int64_t cursor = (live && p0 + gl < W) ? new_offsets[(int64_t)(p0 + gl) * FB + bag] : 0;   // lane r of the group: head of rank p0+r
    const int nr = W - p0 < GL ? W - p0 : GL;
    for (int64_t j0 = 0; j0 < maxlen; j0 += GL) {
      const int64_t j = lo + j0 + gl;
      uint64_t p = ~0ull, nw = 0;
      if (j < hi) route_ex(x, f, indices[j], dist, blk, (uint64_t)W, p, nw);
      int64_t dst = -1;
      for (int r = 0; r < nr; ++r) {
        const uint64_t m = (__ballot(p == (uint64_t)(p0 + r)) >> (grp * GL)) & gmask;
        const int src = grp * GL + r;
        const uint32_t clo = __shfl((int)(uint32_t)cursor, src, 64), chi = __shfl((int)(uint32_t)((uint64_t)cursor >> 32), src, 64);
        const int64_t head = (int64_t)(((uint64_t)chi << 32) | clo);
        if (p == (uint64_t)(p0 + r)) dst = head + __popcll(m & ((1ull << gl) - 1));
        if (gl == r) cursor += __popcll(m);
      }
      if (dst >= 0) {
        new_indices[dst] = nw;
        if (unbucketize_permute) unbucketize_permute[j] = dst;
        if (weights) new_weights[dst] = weights[j];
      }
    }
  }
}

// compute_dedup_lengths (get_new_length_and_offsets_kernel, lookup_kernel.cuh:1049-1090): spread the Nu_t unique keys
// of table t evenly over its (features of t) x local_batch pseudo-bags; first `remainder` bags get one more.
__global__ void __launch_bounds__(256)
dedup_lengths_kernel(const int64_t* __restrict__ unique_offsets, const int64_t* __restrict__ table_offsets_in_feature,
                     int num_tables, int64_t n, int64_t local_batch, int64_t* __restrict__ new_lengths,
                     int64_t* __restrict__ new_offsets) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t feature = i / local_batch;
  int t = 0;
  while (t + 1 < num_tables && table_offsets_in_feature[t + 1] <= feature) ++t;  // upper_bound - 1
  const int64_t f0 = table_offsets_in_feature[t];
  const int64_t buckets = (table_offsets_in_feature[t + 1] - f0) * local_batch;
  const int64_t bid = i - f0 * local_batch;
  const uint64_t nu = (uint64_t)(unique_offsets[t + 1] - unique_offsets[t]);
  const uint64_t base = nu / (uint64_t)buckets, rem = nu % (uint64_t)buckets;
  const uint64_t len = base + ((uint64_t)bid < rem ? 1 : 0);
  const uint64_t off = (uint64_t)unique_offsets[t] + (uint64_t)bid * base + ((uint64_t)bid < rem ? (uint64_t)bid : rem);
  new_lengths[i] = (int64_t)len;
  new_offsets[i] = (int64_t)off;
  if (i == n - 1) new_offsets[n] = (int64_t)(off + base);  // as the reference writes it (== off + len: the last bag never has the +1)
}

// segmented_sum_cuda (index_calculation.cu:38-75): out[s] = sum(int32 data[offsets[s] .. offsets[s+1])) as int64
__global__ void __launch_bounds__(256)
segmented_sum_kernel(const int32_t* __restrict__ data, const int64_t* __restrict__ offsets, int64_t* __restrict__ out) {
  __shared__ int64_t part[4];
  const int64_t b = offsets[blockIdx.x], e = offsets[blockIdx.x + 1];
  int64_t acc = 0;
  for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) acc += data[i];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// Re-order the bags of a received key stream from (source rank, feature, batch) to (feature, source rank,
// batch) order -- what TorchRec's KJTAllToAll does with permute_2D_sparse_data after the key all-to-all
// (third-party; call site corelib/dynamicemb/dynamicemb/input_dist.py:239-285).  One wave per output bag.
__global__ void __launch_bounds__(256)
permute_bags_kernel(int64_t S, int64_t F, int64_t B, int64_t W8, const int64_t* __restrict__ in_offsets,
                    const int64_t* __restrict__ out_offsets, const uint64_t* __restrict__ in_keys,
                    uint64_t* __restrict__ out_keys) {
  const int lane = lane_id();
  const int64_t nb = S * F * B;
  for (int64_t o = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); o < nb; o += (int64_t)gridDim.x * (blockDim.x >> 6)) {
    const int64_t f = o / (S * B), s = (o / B) % S, b = o % B;
    const int64_t i = s * (F * B) + f * B + b;
    const int64_t src = in_offsets[i] * W8, len = (in_offsets[i + 1] - in_offsets[i]) * W8, dst = out_offsets[o] * W8;
    for (int64_t k = lane; k < len; k += 64) out_keys[dst + k] = in_keys[src + k];
  }
}
// Same permutation for long bags (a few bags of many rows: the dedup'd exchange of sharded.py): blockIdx.y
// walks the bags, the blocks of one grid row share a bag.
__global__ void __launch_bounds__(256)
permute_long_bags_kernel(int64_t S, int64_t F, int64_t B, int64_t W8, const int64_t* __restrict__ in_offsets,
                         const int64_t* __restrict__ out_offsets, const uint64_t* __restrict__ in_keys,
                         uint64_t* __restrict__ out_keys) {
  const int64_t nb = S * F * B;
  for (int64_t o = blockIdx.y; o < nb; o += gridDim.y) {
    const int64_t f = o / (S * B), s = (o / B) % S, b = o % B;
    const int64_t i = s * (F * B) + f * B + b;
    const int64_t src = in_offsets[i] * W8, len = (in_offsets[i + 1] - in_offsets[i]) * W8, dst = out_offsets[o] * W8;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < len; k += (int64_t)gridDim.x * 256)
      out_keys[dst + k] = in_keys[src + k];
  }
}

__global__ void __launch_bounds__(256)
permute_lengths_kernel(int64_t S, int64_t F, int64_t B, const int64_t* __restrict__ in_lengths, int64_t* __restrict__ out_lengths) {
  const int64_t nb = S * F * B;
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < nb; o += (int64_t)gridDim.x * blockDim.x) {
    const int64_t f = o / (S * B), s = (o / B) % S, b = o % B;
    out_lengths[o] = in_lengths[s * (F * B) + f * B + b];
  }
}

// exclusive scan of int64 lengths -> offsets (single block; W*F*B is small relative to the keys)
__global__ void __launch_bounds__(1024) scan_i64_kernel(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ out) {
  __shared__ int64_t s_w[17];
  __shared__ int64_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int w = threadIdx.x >> 6, lane = lane_id();
  for (int64_t b0 = 0; b0 < n; b0 += 1024) {
    int64_t i = b0 + threadIdx.x;
    int64_t v = i < n ? in[i] : 0, incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int lo = __shfl_up((int)(uint32_t)incl, off, 64), hi = __shfl_up((int)(uint32_t)((uint64_t)incl >> 32), off, 64);
      int64_t o = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int64_t base = s_carry, tot = 0;
    for (int k = 0; k < 16; ++k) { if (k < w) base += s_w[k]; tot += s_w[k]; }
    if (i < n) out[i] = base + incl - v;
    __syncthreads();
    if (threadIdx.x == 0) s_carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[n] = s_carry;
}

// multi-block variant for long inputs (the W*F*B bag lengths of the sharded path): tile sums, the single-block scan
// above over the tile sums, then a per-tile scan that starts from its tile's prefix
constexpr int kScan64Tile = 1024 * 4;
__device__ __forceinline__ int64_t block_incl_scan_i64(int64_t v, int64_t& total) {
  __shared__ int64_t s_w64[16];
  const int w = threadIdx.x >> 6, lane = lane_id();
  int64_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int lo = __shfl_up((int)(uint32_t)incl, off, 64), hi = __shfl_up((int)(uint32_t)((uint64_t)incl >> 32), off, 64);
    int64_t o = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    if (lane >= off) incl += o;
  }
  __syncthreads();
  if (lane == 63) s_w64[w] = incl;
  __syncthreads();
  int64_t base = 0, tot = 0;
  for (int k = 0; k < 16; ++k) { if (k < w) base += s_w64[k]; tot += s_w64[k]; }
  total = tot;
  return base + incl;
}
__global__ void __launch_bounds__(1024) scan64_tile_sums_kernel(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ sums) {
  const int64_t i0 = (int64_t)blockIdx.x * kScan64Tile + threadIdx.x * 4;
  int64_t v = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) v += i0 + k < n ? in[i0 + k] : 0;
  int64_t tot;
  block_incl_scan_i64(v, tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) scan64_tiles_kernel(const int64_t* __restrict__ in, int64_t n, const int64_t* __restrict__ tile_prefix,
                                                            int64_t* __restrict__ out) {
  const int64_t i0 = (int64_t)blockIdx.x * kScan64Tile + threadIdx.x * 4;
  int64_t x[4], v = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { x[k] = i0 + k < n ? in[i0 + k] : 0; v += x[k]; }
  int64_t tot;
  int64_t run = tile_prefix[blockIdx.x] + block_incl_scan_i64(v, tot) - v;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (i0 + k < n) out[i0 + k] = run; run += x[k]; }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = tile_prefix[gridDim.x];
}

// Single-pass form (round 3): ONE launch instead of three.  Every block scans its tile, publishes the tile's sum in one
// 64-bit word (status in the top bits; the value travels IN the word, so no fence), and the first wave walks back over its
// predecessors' words -- 64 at a time, all polled at once -- until it meets one that already knows its inclusive prefix
// (decoupled look-back; blocks are dispatched in index order, so a predecessor is always resident or finished).  The status
// words are all-zero between calls: the block that finishes LAST (ticket) clears them and the ticket.
constexpr unsigned long long kS64Agg = 1ull << 62, kS64Pre = 2ull << 62, kS64Mask = 3ull << 62;
__global__ void __launch_bounds__(1024) scan64_chained_kernel(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ out,
                                                              unsigned long long* __restrict__ state, int* __restrict__ ticket) {
  __shared__ unsigned long long s_pre;
  __shared__ int s_last;
  const int t = blockIdx.x;
  const int64_t i0 = (int64_t)t * kScan64Tile + threadIdx.x * 4;
  int64_t x[4], v = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { x[k] = i0 + k < n ? in[i0 + k] : 0; v += x[k]; }
  int64_t tot;
  const int64_t incl = block_incl_scan_i64(v, tot);
  if (threadIdx.x == 0)
    __hip_atomic_store(state + t, (t == 0 ? kS64Pre : kS64Agg) | (unsigned long long)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < 64) {
    unsigned long long run = 0;
    for (int pos = t - 1; t > 0;) {
      const int idx = pos - (int)threadIdx.x;
      unsigned long long w = idx >= 0 ? __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kS64Pre;
      while ((w & kS64Mask) == 0) {
        __builtin_amdgcn_s_sleep(1);
        w = __hip_atomic_load(state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const unsigned long long pre = __ballot((w & kS64Mask) == kS64Pre);
      const int first = pre ? __builtin_ctzll(pre) : 64;     // nearest predecessor of this round with an inclusive prefix
      unsigned long long val = (int)threadIdx.x <= first ? (w & ~kS64Mask) : 0ull;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_xor((int)(uint32_t)val, off, 64), hi = __shfl_xor((int)(uint32_t)(val >> 32), off, 64);
        val += ((unsigned long long)hi << 32) | lo;
      }
      run += val;
      if (first < 64) break;
      pos -= 64;
    }
    if (threadIdx.x == 0) {
      if (t > 0) __hip_atomic_store(state + t, kS64Pre | (run + (unsigned long long)tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_pre = run;
    }
  }
  __syncthreads();
  int64_t run = (int64_t)s_pre + incl - v;
#pragma unroll
  for (int k = 0; k < 4; ++k) { if (i0 + k < n) out[i0 + k] = run; run += x[k]; }
  if (t == (int)gridDim.x - 1 && threadIdx.x == 0) out[n] = (int64_t)s_pre + tot;
  // ---- the last block to get here has no reader left behind it: it clears the words for the next call
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(ticket, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (s_last) {
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 1024) state[i] = 0ull;
    if (threadIdx.x == 0) *ticket = 0;
  }
}

}  // namespace mi355

using namespace mi355;
STAMP_EXPORT(mi355_debug_stamps_scatter, g_st_scatter)

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Library-owned look-back words of the long int64 scan (+ its ticket in the word in front of them): one buffer per host
// thread, grown on demand and ZERO between calls (the kernel's last block clears what the call used; a fresh buffer is
// cleared here).  hipFree synchronises the device, so a buffer still in use is never released under a running kernel.
// Callers on ONE thread must not run the scans of two streams concurrently; the sharded path issues them on a single
// stream.  Not hipGraph-capturable while it grows -- warm the path up once before capturing.
static int64_t* scan64_scratch(int64_t words) {
  static thread_local int64_t* p = nullptr;
  static thread_local int64_t cap = 0;
  if (words > cap) {
    if (p) (void)hipFree(p);
    cap = words < 8192 ? 8192 : 2 * words;
    if (hipMalloc(&p, cap * sizeof(int64_t)) != hipSuccess || hipMemset(p, 0, cap * sizeof(int64_t)) != hipSuccess) {
      p = nullptr; cap = 0;
    }
  }
  return p;
}
// Pseudo-bags over per-table unique-key lists (sharded rows-back mode, dynamicemb/sharded.py): table t's
// unique_offsets[t+1]-unique_offsets[t] keys are cut into `num_chunks` bags of at most `chunk` keys.
__global__ void __launch_bounds__(256)
chunk_bags_kernel(const int64_t* __restrict__ uoff, int64_t T, int64_t chunk, int64_t nchunk, int64_t* __restrict__ lengths,
                  int64_t* __restrict__ offsets) {
  const int64_t total = T * nchunk;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= total; i += (int64_t)gridDim.x * blockDim.x) {
    if (i == total) { offsets[i] = uoff[T]; continue; }
    const int64_t t = i / nchunk, c = i - t * nchunk;
    const int64_t lo = uoff[t], cnt = uoff[t + 1] - lo;
    const int64_t a = c * chunk < cnt ? c * chunk : cnt, b = (c + 1) * chunk < cnt ? (c + 1) * chunk : cnt;
    offsets[i] = lo + a;
    lengths[i] = b - a;
  }
}

// keys sent to / received from every peer = differences of the send / receive offsets at the peer boundaries;
// `splits` [2 * W] may live in pinned host memory (the one host read of the exchange)
__global__ void peer_splits_kernel(const int64_t* __restrict__ send_off, const int64_t* __restrict__ recv_off, int64_t per_peer,
                                   int64_t W, int64_t* __restrict__ splits) {
  for (int64_t p = threadIdx.x; p < W; p += blockDim.x) {
    splits[p] = send_off[(p + 1) * per_peer] - send_off[p * per_peer];
    splits[W + p] = recv_off[(p + 1) * per_peer] - recv_off[p * per_peer];
  }
}

static int scan_i64(const int64_t* in, int64_t n, int64_t* out, hipStream_t stream) {
  if (n <= 4 * kScan64Tile) {
    hipLaunchKernelGGL(scan_i64_kernel, dim3(1), dim3(1024), 0, stream, in, n, out);
    return MI355_OK;
  }
  const int64_t nt = ceil_div(n, kScan64Tile);
  int64_t* sc = scan64_scratch(nt + 2);
  if (!sc) { mi355_set_error("scan scratch allocation failed"); return MI355_ELAUNCH; }
  constexpr int three = 0;   // (the three-launch form: kept for batches beyond the one-launch scan's capacity)
  if (three) {
    static thread_local int64_t* sc3 = nullptr;
    static thread_local int64_t cap3 = 0;
    if (2 * nt + 2 > cap3) {
      if (sc3) (void)hipFree(sc3);
      cap3 = 4 * nt + 8192;
      if (hipMalloc(&sc3, cap3 * sizeof(int64_t)) != hipSuccess) { sc3 = nullptr; cap3 = 0; mi355_set_error("scan scratch allocation failed"); return MI355_ELAUNCH; }
    }
    hipLaunchKernelGGL(scan64_tile_sums_kernel, dim3((unsigned)nt), dim3(1024), 0, stream, in, n, sc3);
    hipLaunchKernelGGL(scan_i64_kernel, dim3(1), dim3(1024), 0, stream, sc3, nt, sc3 + nt);
    hipLaunchKernelGGL(scan64_tiles_kernel, dim3((unsigned)nt), dim3(1024), 0, stream, in, n, sc3 + nt, out);
    return MI355_OK;
  }
  hipLaunchKernelGGL(scan64_chained_kernel, dim3((unsigned)nt), dim3(1024), 0, stream, in, n, out, (unsigned long long*)(sc + 2),
                     (int*)sc);
  return MI355_OK;
}

extern "C" {

int64_t mi355_segmented_unique_workspace_bytes(int64_t n) {
  int64_t nb = ceil_div(n > 0 ? n : 1, kScanTile);
  return 2 * align_up(8 * n, 256) + align_up(4 * n, 256) + align_up(4 * (nb + 1), 256) + 256;
}

int mi355_segmented_unique(const void* keys, int64_t n, const int64_t* segmented_range, int64_t num_tables,
                           const int64_t* input_frequencies, int count_freq, void* unique_keys,
                           int64_t* output_indices, int64_t* table_offsets, int64_t* freq, void* workspace,
                           int64_t workspace_bytes, hipStream_t stream) {
  return mi355_segmented_unique_csr(keys, n, segmented_range, num_tables, input_frequencies, count_freq, unique_keys,
                                    output_indices, table_offsets, freq, nullptr, nullptr, workspace, workspace_bytes, stream);
}

int mi355_segmented_unique_csr(const void* keys, int64_t n, const int64_t* segmented_range, int64_t num_tables,
                               const int64_t* input_frequencies, int count_freq, void* unique_keys,
                               int64_t* output_indices, int64_t* table_offsets, int64_t* freq, int32_t* csr_cnt,
                               int32_t* csr_rank, void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  return mi355i_segmented_unique(keys, n, segmented_range, num_tables, input_frequencies, count_freq, unique_keys,
                                 output_indices, table_offsets, freq, csr_cnt, csr_rank, nullptr, nullptr, 0, nullptr, nullptr,
                                 workspace, workspace_bytes, stream);
}

int mi355i_segmented_unique(const void* keys, int64_t n, const int64_t* segmented_range, int64_t num_tables,
                            const int64_t* input_frequencies, int count_freq, void* unique_keys,
                            int64_t* output_indices, int64_t* table_offsets, int64_t* freq, int32_t* csr_cnt,
                            int32_t* csr_rank, const int64_t* offsets, const int64_t* feature_offsets,
                            int64_t feature_x_batch, int64_t* table_range_out, int64_t* table_ids_out, void* workspace,
                            int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(num_tables > 0 && num_tables <= 65535, "num_tables must be in [1, 65535]");
  MI355_CHECK_ARG(n < 0x7fffffffLL / 2, "num_keys must be < 2^30");
  MI355_CHECK_ARG(segmented_range || (offsets && feature_offsets && table_range_out), "table ranges or bag offsets required");
  if (n == 0) {
    hipLaunchKernelGGL(zero_offsets_kernel, dim3(1), dim3(64), 0, stream, table_offsets, num_tables + 1);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  }
  MI355_CHECK_ARG(!count_freq || freq, "freq output required when counting frequencies");
  MI355_CHECK_ARG(workspace && workspace_bytes >= mi355_segmented_unique_workspace_bytes(n), "workspace too small");
  const int64_t nb = ceil_div(n, kScanTile);
  uint8_t* w = (uint8_t*)workspace;
  UniqWs ws;
  ws.slots = (int*)w; w += align_up(8 * n, 256);
  ws.gcnt = (int*)w; w += align_up(8 * n, 256);
  ws.rep = (int*)w; w += align_up(4 * n, 256);
  ws.partial = (int*)w; w += align_up(4 * (nb + 1), 256);
  ws.total = (int*)w;
  const int T = (int)num_tables;
  // slots and counters are adjacent and share the fill byte: one fill (plus the table ranges when they are derived here)
  const int64_t n16 = 2 * align_up(8 * n, 256) / 16;
  hipLaunchKernelGGL(uniq_prepare_kernel, dim3(grid_for(n16, 256, 2048)), dim3(256), 0, stream, (uint4*)ws.slots, n16, offsets,
                     feature_offsets, T, feature_x_batch, segmented_range ? (int64_t*)nullptr : table_range_out);
  if (!segmented_range) segmented_range = table_range_out;
  const uint64_t* k = (const uint64_t*)keys;
  hipLaunchKernelGGL(uniq_insert_kernel, dim3((unsigned)ceil_div(n, kUniqTile)), dim3(kUniqThreads), 0, stream, k, n, segmented_range, T, ws,
                     csr_rank);
  hipLaunchKernelGGL(uniq_flag_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, n, ws);
  const bool self = nb <= kSelfPrefixMaxTiles;   // emit blocks sum the earlier tiles' counts themselves: one launch less
  if (!self) hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(kScanThreads), 0, stream, ws.partial, nb, ws.total);
#define MI355_EMIT(FREQ, SELF)                                                                                          \
  hipLaunchKernelGGL((uniq_emit_kernel<FREQ, SELF>), dim3((unsigned)nb), dim3(kScanThreads), 0, stream, k, n, segmented_range, T, \
                     ws, (uint64_t*)unique_keys, table_offsets, freq, input_frequencies, csr_cnt)
  if (count_freq) {
    if (self) MI355_EMIT(true, true); else MI355_EMIT(true, false);
    hipLaunchKernelGGL(uniq_finish_kernel<true>, dim3(grid_for(n, 256, 256 * 32)), dim3(256), 0, stream, n, ws, input_frequencies,
                       output_indices, freq, table_offsets, T, table_ids_out);
  } else {
    if (self) MI355_EMIT(false, true); else MI355_EMIT(false, false);
    hipLaunchKernelGGL(uniq_finish_kernel<false>, dim3(grid_for(n, 256, 256 * 32)), dim3(256), 0, stream, n, ws, input_frequencies,
                       output_indices, freq, table_offsets, T, table_ids_out);
  }
#undef MI355_EMIT
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_expand_table_ids(const int64_t* offsets, int64_t num_tables, int64_t n, const int64_t* n_dev,
                           int64_t* table_ids, hipStream_t stream) {
  if (n == 0) return MI355_OK;
  hipLaunchKernelGGL(expand_table_ids_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, offsets, (int)num_tables, n, n_dev, table_ids);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_get_table_range(const int64_t* offsets, const int64_t* feature_offsets, int64_t num_tables, int64_t feature_x_batch,
                          int64_t* table_range, hipStream_t stream) {
  hipLaunchKernelGGL(get_table_range_kernel, dim3(1), dim3(128), 0, stream, offsets, feature_offsets, (int)num_tables, feature_x_batch, table_range);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int64_t mi355_flagged_compact_workspace_bytes(int64_t n) {
  return align_up(4 * (ceil_div(n > 0 ? n : 1, kScanTile) + 1), 256) + 256;
}

// out_index / inputs are int64 arrays (keys, table ids, scores are all 8-byte words on this path)
int mi355_flagged_compact(const uint8_t* flags, int64_t n, const int64_t* n_dev, int64_t* count_out, int64_t* out_index,
                          int num_arrays, const void* const* inputs, void* const* outputs, void* workspace,
                          int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(num_arrays >= 0 && num_arrays <= 6, "at most 6 arrays");
  if (n == 0) {
    hipLaunchKernelGGL(zero_offsets_kernel, dim3(1), dim3(64), 0, stream, count_out, (int64_t)1);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  }
  MI355_CHECK_ARG(workspace && workspace_bytes >= mi355_flagged_compact_workspace_bytes(n), "workspace too small");
  const int64_t nb = ceil_div(n, kScanTile);
  int* partial = (int*)workspace;
  int* total = (int*)((uint8_t*)workspace + align_up(4 * (nb + 1), 256));
  CompactArrays arrs;
  arrs.num = num_arrays;
  for (int a = 0; a < num_arrays; ++a) { arrs.in[a] = (const int64_t*)inputs[a]; arrs.out[a] = (int64_t*)outputs[a]; }
  hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, flags, n, n_dev, partial);
  hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(kScanThreads), 0, stream, partial, nb, total);
  hipLaunchKernelGGL(compact_emit_kernel, dim3((unsigned)nb), dim3(kScanThreads), 0, stream, flags, n, n_dev, partial, total,
                     count_out, out_index, arrs);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// CSR of the batch keyed by unique row.  cnt/cursor: int32[max_unique] scratch, ptr: int32[max_unique+1],
// csr_src: int32[n].  num_unique may live on the device (nu_dev).
int64_t mi355_group_by_unique_workspace_bytes(int64_t n, int64_t max_unique) {
  return 2 * align_up(4 * (max_unique + 1), 256) + align_up(4 * (ceil_div(max_unique + 1, kScanTile) + 1), 256) + 256;
}
int64_t mi355_hot_rows_workspace_bytes(int64_t num_keys, int64_t dim) { return hot_bytes(num_keys, dim); }

int mi355_group_by_unique(const int64_t* reverse_indices, int64_t n, const int64_t* offsets, int64_t num_bags,
                          int64_t max_unique, const int64_t* nu_dev, int32_t* ptr, int32_t* csr_src, void* workspace,
                          int64_t workspace_bytes, void* hot_workspace, int64_t hot_workspace_bytes, int64_t dim,
                          hipStream_t stream) {
  MI355_CHECK_ARG(n < 0x7fffffffLL, "too many keys");
  MI355_CHECK_ARG(workspace && workspace_bytes >= mi355_group_by_unique_workspace_bytes(n, max_unique), "workspace too small");
  MI355_CHECK_ARG(!hot_workspace || hot_workspace_bytes >= hot_bytes(n, dim), "hot workspace too small");
  uint8_t* w = (uint8_t*)workspace;
  int* cnt = (int*)w; w += align_up(4 * (max_unique + 1), 256);
  int* cursor = (int*)w; w += align_up(4 * (max_unique + 1), 256);
  const int64_t nbu = ceil_div(max_unique + 1, kScanTile);
  int* partial = (int*)w; w += align_up(4 * (nbu + 1), 256);
  int* total = (int*)w;
  HotList hot{};
  if (hot_workspace) {
    hot = hot_carve(hot_workspace, n, dim);
    if (hipMemsetAsync(hot_workspace, 0, 256, stream) != hipSuccess) { mi355_set_error("memset failed"); return MI355_ELAUNCH; }
  }
  if (hipMemsetAsync(cnt, 0, 2 * align_up(4 * (max_unique + 1), 256), stream) != hipSuccess) { mi355_set_error("memset failed"); return MI355_ELAUNCH; }
  if (n > 0) hipLaunchKernelGGL(csr_hist_kernel, dim3((unsigned)ceil_div(n, kHistTile)), dim3(256), 0, stream, reverse_indices, n, cnt);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nbu), dim3(kScanThreads), 0, stream, cnt, max_unique, nu_dev, partial, (int*)nullptr,
                     (int*)nullptr);
  hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(kScanThreads), 0, stream, partial, nbu, total);
  hipLaunchKernelGGL(scan_down_kernel<false>, dim3((unsigned)nbu), dim3(kScanThreads), 0, stream, cnt, max_unique, nu_dev, partial, total, ptr,
                     hot, hot_workspace != nullptr);
  if (n > 0) hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)ceil_div(n, kHistTile)), dim3(256), 0, stream, reverse_indices, n, offsets,
                                num_bags, ptr, cursor, csr_src, hot, hot_workspace != nullptr);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// CSR from the counts / ranks produced by mi355_segmented_unique_csr: scan + scatter, no atomics.
int64_t mi355_group_by_unique_csr_workspace_bytes(int64_t max_unique) {
  return align_up(4 * (ceil_div(max_unique + 1, kScanTile) + 1), 256) + 256;
}
int mi355_group_by_unique_csr(const int32_t* csr_cnt, const int32_t* csr_rank, const int64_t* reverse_indices, int64_t n,
                              const int64_t* offsets, int64_t num_bags, int64_t max_unique, const int64_t* nu_dev,
                              int32_t* ptr, int32_t* csr_src, void* workspace, int64_t workspace_bytes,
                              void* hot_workspace, int64_t hot_workspace_bytes, int64_t dim, hipStream_t stream) {
  MI355_CHECK_ARG(n < 0x7fffffffLL && max_unique < 0x7fffffffLL, "n must be < 2^31");
  MI355_CHECK_ARG(workspace && workspace_bytes >= mi355_group_by_unique_csr_workspace_bytes(max_unique), "workspace too small");
  MI355_CHECK_ARG(csr_cnt && csr_rank, "counts and ranks required");
  const int64_t nbu = ceil_div(max_unique + 1, kScanTile);
  int* partial = (int*)workspace;
  int* total = (int*)((uint8_t*)workspace + align_up(4 * (nbu + 1), 256));
  HotList hot{};
  if (hot_workspace) {
    MI355_CHECK_ARG(hot_workspace_bytes >= hot_bytes(n, dim), "hot workspace too small");
    hot = hot_carve(hot_workspace, n, dim);
  }
  // the hot-list counters are cleared by the reduce pass; the down pass sums the earlier tiles' counts itself when
  // there are few tiles (no one-block scan launch in between)
  hipLaunchKernelGGL(scan_reduce_kernel, dim3((unsigned)nbu), dim3(kScanThreads), 0, stream, csr_cnt, max_unique, nu_dev, partial,
                     hot_workspace ? hot.n_hot : (int*)nullptr, hot_workspace ? hot.n_tasks : (int*)nullptr);
  if (nbu <= kSelfPrefixMaxTiles) {
    hipLaunchKernelGGL(scan_down_kernel<true>, dim3((unsigned)nbu), dim3(kScanThreads), 0, stream, csr_cnt, max_unique, nu_dev, partial, total,
                       ptr, hot, hot_workspace != nullptr);
  } else {
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(kScanThreads), 0, stream, partial, nbu, total);
    hipLaunchKernelGGL(scan_down_kernel<false>, dim3((unsigned)nbu), dim3(kScanThreads), 0, stream, csr_cnt, max_unique, nu_dev, partial, total,
                       ptr, hot, hot_workspace != nullptr);
  }
  if (n > 0) hipLaunchKernelGGL(csr_scatter_kernel<0>, dim3((unsigned)ceil_div(n, kHistTile)), dim3(256), 0, stream, reverse_indices, const_cast<int32_t*>(csr_rank), n,
                                offsets, num_bags, ptr, csr_src, hot, hot_workspace != nullptr, (const int*)nullptr, (const int*)nullptr,
                                (int64_t*)nullptr);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// Last index kernel of the fused forward (fused_fwd.hip): its numbering kernel left the unique id of every occurrence's
// slot pair (or (tile, key) record: `part`), the CSR row pointers in `ptr` and the hot rows registered; this scatters the
// bag of every key into its row's list, expands the hot rows' chunk tasks and writes the reverse indices.
// ptr == NULL (no backward workspace): only the reverse indices (and full ranks) are produced.
int mi355i_csr_from_slots(const int32_t* csr_rank, const int32_t* occ_slot, const int32_t* uidmap, int64_t* reverse_indices,
                          int64_t n, const int64_t* offsets, int64_t num_bags, int32_t* ptr, int32_t* csr_src,
                          void* hot_workspace, int64_t hot_workspace_bytes, int64_t dim, int32_t* hdr_reset,
                          const PartRefs* part, hipStream_t stream, const int* gate, int gate_val, int* mark) {
  MI355_CHECK_ARG(n < 0x7fffffffLL, "n must be < 2^31");
  if (n == 0) return MI355_OK;
  HotList hot{};
  if (hot_workspace) {
    MI355_CHECK_ARG(hot_workspace_bytes >= hot_bytes(n, dim), "hot workspace too small");
    hot = hot_carve(hot_workspace, n, dim);
  }
  if (ptr) {
    if (part && part->rec_out)
      hipLaunchKernelGGL(csr_scatter_kernel<2>, dim3((unsigned)ceil_div(n, kHistTile)), dim3(256), 0, stream, (const int64_t*)nullptr,
                         const_cast<int32_t*>(csr_rank), n, offsets, num_bags, ptr, csr_src, hot, hot_workspace != nullptr, occ_slot,
                         uidmap, reverse_indices, hdr_reset, *part);
    else
      hipLaunchKernelGGL(csr_scatter_kernel<1>, dim3((unsigned)ceil_div(n, kHistTile)), dim3(256), 0, stream, (const int64_t*)nullptr,
                         const_cast<int32_t*>(csr_rank), n, offsets, num_bags, ptr, csr_src, hot, hot_workspace != nullptr, occ_slot,
                         uidmap, reverse_indices, hdr_reset, PartRefs{}, gate, gate_val, mark);
  } else if (part && part->rec_out) {
    hipLaunchKernelGGL(rev_from_records_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, occ_slot, *part,
                       const_cast<int32_t*>(csr_rank), n, reverse_indices, hdr_reset);
  } else {
    hipLaunchKernelGGL(rev_from_slots_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, occ_slot, uidmap, n, reverse_indices,
                       hdr_reset);
  }
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_compute_dedup_lengths(const int64_t* unique_offsets, const int64_t* table_offsets_in_feature, int64_t num_tables,
                                int64_t local_batch_size, int64_t new_lengths_size, int64_t* new_lengths,
                                int64_t* new_offsets, hipStream_t stream) {
  MI355_CHECK_ARG(local_batch_size > 0 || new_lengths_size == 0, "local_batch_size must be positive");
  if (new_lengths_size == 0) return MI355_OK;
  hipLaunchKernelGGL(dedup_lengths_kernel, dim3((unsigned)ceil_div(new_lengths_size, 256)), dim3(256), 0, stream,
                     unique_offsets, table_offsets_in_feature, (int)num_tables, new_lengths_size, local_batch_size,
                     new_lengths, new_offsets);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_segmented_sum(const int32_t* data, const int64_t* offsets, int64_t num_segments, int64_t* out, hipStream_t stream) {
  MI355_CHECK_ARG(num_segments > 0, "offsets size must be at least 2 (num_segments >= 1)");
  hipLaunchKernelGGL(segmented_sum_kernel, dim3((unsigned)num_segments), dim3(256), 0, stream, data, offsets, out);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_permute_lengths(int64_t num_sources, int64_t num_features, int64_t batch_size, const int64_t* in_lengths,
                          int64_t* out_lengths, hipStream_t stream) {
  const int64_t nb = num_sources * num_features * batch_size;
  if (nb == 0) return MI355_OK;
  hipLaunchKernelGGL(permute_lengths_kernel, dim3(grid_for(nb, 256)), dim3(256), 0, stream, num_sources, num_features, batch_size,
                     in_lengths, out_lengths);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_permute_bags(int64_t num_sources, int64_t num_features, int64_t batch_size, int64_t elem_bytes,
                       int64_t num_elements, const int64_t* in_offsets, const int64_t* out_offsets, const void* in_keys,
                       void* out_keys, hipStream_t stream) {
  const int64_t nb = num_sources * num_features * batch_size;
  MI355_CHECK_ARG(elem_bytes > 0 && elem_bytes % 8 == 0, "element size must be a multiple of 8 bytes");
  if (nb == 0 || num_elements == 0) return MI355_OK;
  const int64_t words_per_bag = num_elements * (elem_bytes / 8) / nb;
  if (words_per_bag > 512) {
    const int64_t gx = std::min<int64_t>((words_per_bag + 1023) / 1024, 256);
    const int64_t gy = std::min<int64_t>(nb, 65535);
    hipLaunchKernelGGL(permute_long_bags_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, stream, num_sources,
                       num_features, batch_size, elem_bytes / 8, in_offsets, out_offsets, (const uint64_t*)in_keys,
                       (uint64_t*)out_keys);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  }
  hipLaunchKernelGGL(permute_bags_kernel, dim3(grid_for(nb, 4, 1 << 16)), dim3(256), 0, stream, num_sources, num_features, batch_size,
                     elem_bytes / 8, in_offsets, out_offsets, (const uint64_t*)in_keys, (uint64_t*)out_keys);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_exclusive_offsets(const int64_t* lengths, int64_t n, int64_t* offsets, hipStream_t stream) {
  MI355_CHECK_ARG(n >= 0 && offsets, "offsets output required");
  if (n == 0) {
    hipLaunchKernelGGL(zero_offsets_kernel, dim3(1), dim3(64), 0, stream, offsets, (int64_t)1);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  }
  if (scan_i64(lengths, n, offsets, stream) != MI355_OK) return MI355_ELAUNCH;
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_chunk_bags(const int64_t* unique_offsets, int64_t num_tables, int64_t chunk, int64_t num_chunks, int64_t* lengths,
                     int64_t* offsets, hipStream_t stream) {
  MI355_CHECK_ARG(num_tables > 0 && chunk > 0 && num_chunks > 0, "tables, chunk and num_chunks must be positive");
  hipLaunchKernelGGL(chunk_bags_kernel, dim3(grid_for(num_tables * num_chunks + 1, 256)), dim3(256), 0, stream, unique_offsets,
                     num_tables, chunk, num_chunks, lengths, offsets);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_peer_splits(const int64_t* send_offsets, const int64_t* recv_offsets, int64_t bags_per_peer, int64_t world_size,
                      int64_t* splits, hipStream_t stream) {
  MI355_CHECK_ARG(world_size > 0 && bags_per_peer >= 0, "bad world size / bags per peer");
  hipLaunchKernelGGL(peer_splits_kernel, dim3(1), dim3(64), 0, stream, send_offsets, recv_offsets, bags_per_peer, world_size, splits);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_block_bucketize_ex(int64_t world_size, int64_t num_bags, int64_t batch_size, const int64_t* offsets,
                             const void* indices, const int64_t* block_sizes, const int32_t* dist_type_per_feature,
                             const float* weights, int64_t* new_lengths, int64_t* new_offsets, void* new_indices,
                             float* new_weights, int64_t* unbucketize_permute, const int64_t* feature_bag_starts,
                             int64_t num_features, const int64_t* block_bucketize_pos_concat,
                             const int64_t* block_bucketize_pos_offsets, hipStream_t stream) {
  MI355_CHECK_ARG(world_size >= 1 && world_size <= 4096, "bad world size");
  MI355_CHECK_ARG(!feature_bag_starts || (num_features >= 1 && num_features < (1 << 30)), "feature_bag_starts needs num_features");
  MI355_CHECK_ARG(feature_bag_starts || batch_size > 0 || num_bags == 0, "batch_size must be positive");
  MI355_CHECK_ARG((block_bucketize_pos_concat == nullptr) == (block_bucketize_pos_offsets == nullptr), "block_bucketize_pos needs its offsets");
  if (num_bags == 0) return MI355_OK;
  const int W = (int)world_size;
  BktEx x;
  x.fstart = feature_bag_starts; x.F = (int)num_features; x.pos = block_bucketize_pos_concat; x.pos_off = block_bucketize_pos_offsets;
  // many bags (embedding-bag batches: tens of thousands of bags of a few keys) -> 8 lanes per bag; few bags (HSTU
  // sequences, the pseudo-bags of the rows-back exchange) -> a wave per bag.  Same results either way.
  const bool short_bags = num_bags >= 8192;
  constexpr int GL = 8;
  const int grid_s = grid_for(num_bags, 4 * (64 / GL), 1 << 20);
  if (short_bags)
    hipLaunchKernelGGL(bucketize_count_short_kernel<GL>, dim3(grid_s), dim3(256), 0, stream, num_bags, batch_size, W, offsets,
                       (const uint64_t*)indices, block_sizes, dist_type_per_feature, new_lengths, x);
  else
    hipLaunchKernelGGL(bucketize_count_kernel, dim3(grid_for(num_bags, 4)), dim3(256), 0, stream, num_bags, batch_size, W, offsets,
                       (const uint64_t*)indices, block_sizes, dist_type_per_feature, new_lengths, x);
  if (scan_i64(new_lengths, world_size * num_bags, new_offsets, stream) != MI355_OK) return MI355_ELAUNCH;
  if (short_bags)
    hipLaunchKernelGGL(bucketize_scatter_short_kernel<GL>, dim3(grid_s), dim3(256), 0, stream, num_bags, batch_size, W, offsets,
                       (const uint64_t*)indices, block_sizes, dist_type_per_feature, new_offsets, (uint64_t*)new_indices,
                       unbucketize_permute, weights, new_weights, x);
  else
    hipLaunchKernelGGL(bucketize_scatter_kernel, dim3(grid_for(num_bags, 4)), dim3(256), 0, stream, num_bags, batch_size, W, offsets,
                       (const uint64_t*)indices, block_sizes, dist_type_per_feature, new_offsets, (uint64_t*)new_indices,
                       unbucketize_permute, weights, new_weights, x);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_block_bucketize(int64_t world_size, int64_t num_bags, int64_t batch_size, const int64_t* offsets,
                          const void* indices, const int64_t* block_sizes, const int32_t* dist_type_per_feature,
                          const float* weights, int64_t* new_lengths, int64_t* new_offsets, void* new_indices,
                          float* new_weights, int64_t* unbucketize_permute, hipStream_t stream) {
  return mi355_block_bucketize_ex(world_size, num_bags, batch_size, offsets, indices, block_sizes, dist_type_per_feature, weights,
                                  new_lengths, new_offsets, new_indices, new_weights, unbucketize_permute, nullptr, 0, nullptr, nullptr,
                                  stream);
}

}  // extern "C"
