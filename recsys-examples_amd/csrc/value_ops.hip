// Embedding-row kernels of the DynamicEmb forward path for gfx950: fused gather + pooling,
// sequence gather, flat-table row load/store, first-touch row initialisers.
//
// Replaces (reference, corelib/dynamicemb/src/): multi_to_one_warp_per_ev_vec4_kernel /
// multi_to_one_cta_per_ev_kernel (lookup_kernel.cuh:859-998, descriptor lookup_forward.cu:30-104),
// one_to_one_warp_per_ev kernels (lookup_kernel.cuh:811-857), load_from_flat_table_* /
// store_to_flat_table_* (dynamic_emb_op.cu:294-684), initialize_with_index_addressor_kernel
// (initializer.cu:64-83, initializer.cuh:158-176).
//
// MI355X design
//  * The reference gathers table rows into a dense [Nu, D] staging tensor (load_from_flat) and
//    pools from that.  Here the pooling kernel can read the table rows IN PLACE through a
//    per-unique row-address array, so every unique row is read from HBM once (duplicates hit
//    L2/MALL) and the staging pass disappears.  The dense-source form of the reference op is kept.
//  * One wave64 per bag.  A row of D elements is covered by LPR = D/4 lanes (16 B per lane for
//    fp32 rows), so a 128-D fp32 row is half a wave and TWO rows move per wave instruction
//    (1 KiB per dwordx4 load instruction); the bag loop is unrolled 4x so up to 8 independent
//    512-B row reads are in flight per wave.  Partial sums of the row groups are folded with
//    xor-shuffles; fp32 accumulation, one rounding at the store.
//  * HBM-bound: no LDS, no MFMA.  Occupancy (small VGPR count) supplies the latency hiding.
#include "common.h"
#include "internal.h"
#include "init_dev.h"
#include "stamps.h"
#include "gather_dev.h"
#include <stdlib.h>

namespace mi355 {

template <int SDT>
__device__ __forceinline__ const void* src_row(const PoolArgs& a, int64_t u) {
  if (a.row_addr) return reinterpret_cast<const void*>(a.row_addr[u]);
  return reinterpret_cast<const typename Elem<SDT>::T*>(a.src) + u * a.src_stride;
}

// vectorised: D_f % 4 == 0, rows 16-B (fp32) / 8-B (16-bit) aligned.  NCOL = ceil(D / (4*LPR)).
//
// The kernel is LATENCY bound, not issue bound: a bag is a chain of dependent loads
// (offsets -> reverse index -> row address -> row) of ~1.5 us per hop under load, so the number of
// independent chains in flight decides the bandwidth.  Each LPR-lane group therefore owns NB whole
// bags at a time (64/LPR groups per wave -> NB*64/LPR bags per wave) and walks them in lock step,
// four rows per bag per round; all index loads of a round are issued before the row loads, all row
// loads before the adds.  A bag never leaves its lane group, so there is no cross-lane reduction and
// the fp32 sum runs in bag order (bit-identical to a sequential sum).
// (PoolArgs, the zero row, ld4g and the flat-stream gather live in gather_dev.h)
template <int SDT, int DDT, int NCOL, int NB, bool kAddr, int RPR = 4>
__global__ void __launch_bounds__(256) gather_pooled_vec_kernel(PoolArgs a, int lpr_log2) {
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2;
  const int NSUB = 64 >> lpr_log2;
  const int sub = lane >> lpr_log2;
  const int c = lane & (LPR - 1);
  const int64_t wpb = blockDim.x >> 6;
  const int64_t sg = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * NSUB + sub;
  const int64_t total_sg = (int64_t)gridDim.x * wpb * NSUB;
  const gptr_t zero = (gptr_t)(uintptr_t)g_zero_row;
  constexpr int EB = SDT == kF32 ? 4 : 2;
  for (int64_t bag0 = sg * NB; bag0 < a.FB; bag0 += total_sg * NB) {
    int64_t lo[NB], hi[NB];
    int Dfb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int64_t bag = bag0 + b < a.FB ? bag0 + b : a.FB - 1;
      lo[b] = a.offsets[bag];
      hi[b] = bag0 + b < a.FB ? a.offsets[bag + 1] : lo[b];
      Dfb[b] = a.D;
      if (a.D_offsets) { const int f = (int)(bag / a.B); Dfb[b] = a.D_offsets[f + 1] - a.D_offsets[f]; }
    }
    float4 acc[NB][NCOL];
    int64_t maxlen = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int k = 0; k < NCOL; ++k) acc[b][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      maxlen = hi[b] - lo[b] > maxlen ? hi[b] - lo[b] : maxlen;
    }
    for (int64_t r = 0; r < maxlen; r += RPR) {
      // hop 1: reverse indices (clamped to the bag's last key: the duplicate is masked below)
      int64_t u[NB][RPR];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < RPR; ++q) {
          int64_t j = lo[b] + r + q;
          j = j < hi[b] ? j : hi[b] - 1;
          j = j < lo[b] ? lo[b] : j;           // empty bag: any in-range key (a.n > 0 here)
          j = j < a.n ? j : a.n - 1;
          u[b][q] = a.rev ? a.rev[j] : j;
        }
      // hop 2: row addresses
      uintptr_t rp[NB][RPR];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < RPR; ++q) {
          uintptr_t p;
          if constexpr (kAddr) p = (uintptr_t)a.row_addr[u[b][q]];
          else p = (uintptr_t)a.src + (uintptr_t)(u[b][q] * a.src_stride * EB);
          rp[b][q] = (lo[b] + r + q < hi[b]) ? p : 0;
        }
      // hop 3: rows
      float4 v[NB][RPR][NCOL];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < RPR; ++q)
#pragma unroll
          for (int k = 0; k < NCOL; ++k) {
            const int e = 4 * (c + k * LPR);
            const gptr_t p = (rp[b][q] != 0 && e < Dfb[b]) ? (gptr_t)(rp[b][q] + (uintptr_t)(e * EB)) : zero;
            v[b][q][k] = ld4g<SDT>(p);
          }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < RPR; ++q)
#pragma unroll
          for (int k = 0; k < NCOL; ++k) add4(acc[b][k], v[b][q][k]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int64_t bag = bag0 + b;
      if (bag >= a.FB) continue;
      const int f = (int)(bag / a.B), bb = (int)(bag % a.B);
      int d0, Df;
      if (a.D_offsets) { d0 = a.D_offsets[f]; Df = a.D_offsets[f + 1] - d0; } else { d0 = f * a.D; Df = a.D; }
      const int64_t L = hi[b] - lo[b];
      if (a.combiner == 1 && L > 0) {
        const float fl = (float)L;
#pragma unroll
        for (int k = 0; k < NCOL; ++k) { acc[b][k].x /= fl; acc[b][k].y /= fl; acc[b][k].z /= fl; acc[b][k].w /= fl; }
      }
#pragma unroll
      for (int k = 0; k < NCOL; ++k) {
        const int e = 4 * (c + k * LPR);
        if (e < Df) st4<DDT>(a.dst, (int64_t)bb * a.total_D + d0 + e, acc[b][k]);
      }
    }
  }
}

// Software-pipelined form (gather_dev.h: gather_pooled_pipe): the default of one-column-group rows
STAMP_ARRAY(g_st_gather, 16384, 2)
template <int SDT, int DDT, int kAddr, int UNR, int KIT>
__global__ void __launch_bounds__(256) gather_pooled_pipe_kernel(PoolArgs a, int lpr_log2) {
  STAMP(g_st_gather, 16384, 2, 0);
  const int64_t sg = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 >> lpr_log2) + (lane_id() >> lpr_log2);
  gather_pooled_pipe<SDT, DDT, kAddr, UNR, KIT>(a, LateRefs{}, lpr_log2, sg);
  STAMP(g_st_gather, 16384, 2, 1);
}


// scalar fallback (odd dims such as 7 / 11 / 13): lane handles elements lane, lane+64, ...
template <int SDT, int DDT>
__global__ void __launch_bounds__(256) gather_pooled_scalar_kernel(PoolArgs a) {
  const int lane = lane_id();
  const int64_t wpb = blockDim.x >> 6;
  constexpr int kMaxCol = 16;  // D <= 1024
  for (int64_t bag = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); bag < a.FB; bag += (int64_t)gridDim.x * wpb) {
    const int f = (int)(bag / a.B), b = (int)(bag % a.B);
    int d0, Df;
    if (a.D_offsets) { d0 = a.D_offsets[f]; Df = a.D_offsets[f + 1] - d0; } else { d0 = f * a.D; Df = a.D; }
    const int64_t lo = a.offsets[bag], hi = a.offsets[bag + 1];
    float acc[kMaxCol];
#pragma unroll
    for (int k = 0; k < kMaxCol; ++k) acc[k] = 0.f;
    for (int64_t j = lo; j < hi; ++j) {
      const void* rp = src_row<SDT>(a, a.rev ? a.rev[j] : j);
      if (!rp) continue;
#pragma unroll
      for (int k = 0; k < kMaxCol; ++k) {
        const int e = lane + 64 * k;
        if (e < Df) acc[k] += ld1<SDT>(rp, e);
      }
    }
    const int64_t L = hi - lo;
#pragma unroll
    for (int k = 0; k < kMaxCol; ++k) {
      const int e = lane + 64 * k;
      if (e < Df) {
        float v = acc[k];
        if (a.combiner == 1 && L > 0) v /= (float)L;
        st1<DDT>(a.dst, (int64_t)b * a.total_D + d0 + e, v);
      }
    }
  }
}

// sequence gather: out[i, :D] = row(rev[i]) (cast).  LPR lanes per row, 64/LPR rows per wave step.
template <int SDT, int DDT, bool kVec>
__global__ void __launch_bounds__(256)
gather_rows_kernel(const void* src, int64_t src_stride, const int64_t* __restrict__ row_addr,
                   const int64_t* __restrict__ index, int64_t n, const int64_t* __restrict__ n_dev, int D,
                   void* dst, int64_t dst_stride, int lpr_log2) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2, R = 64 >> lpr_log2;
  const int sub = lane >> lpr_log2, c = lane & (LPR - 1);
  const int64_t rows_per_block = (int64_t)(blockDim.x >> 6) * R;
  for (int64_t r0 = (int64_t)blockIdx.x * rows_per_block + (int64_t)(threadIdx.x >> 6) * R; r0 < n;
       r0 += (int64_t)gridDim.x * rows_per_block) {
    const int64_t i = r0 + sub;
    if (i >= n) continue;
    const int64_t u = index ? index[i] : i;
    const void* rp = nullptr;
    if (u >= 0) {
      if (row_addr) rp = reinterpret_cast<const void*>(row_addr[u]);
      else rp = reinterpret_cast<const typename Elem<SDT>::T*>(src) + u * src_stride;
    }
    if (kVec) {
      for (int e = 4 * c; e < D; e += 4 * LPR) {
        float4 v = rp ? ld4<SDT>(rp, e) : make_float4(0.f, 0.f, 0.f, 0.f);
        st4<DDT>(dst, i * dst_stride + e, v);
      }
    } else {
      for (int e = c; e < D; e += LPR) st1<DDT>(dst, i * dst_stride + e, rp ? ld1<SDT>(rp, e) : 0.f);
    }
  }
}

// flat-table row copy, both directions (dynamic_emb_op.cu:294-490).  region: 0 contiguous
// (min(vdim, dim) elements), 1 embedding only, 2 embedding + optimizer state re-padded to
// max_emb_dim in the dense buffer.  kLoad: table -> dense, else dense -> table.  idx < 0 skipped.
template <int DT, bool kLoad>
__global__ void __launch_bounds__(256)
flat_table_copy_kernel(int region, int64_t n, const int64_t* __restrict__ n_dev, void* dense, int64_t dense_dim,
                       int64_t dense_stride, const int64_t* __restrict__ indices, const int64_t* __restrict__ table_ids,
                       int64_t scalar_table_id, const int64_t* __restrict__ table_ptrs,
                       const int64_t* __restrict__ table_value_dims, const int64_t* __restrict__ table_emb_dims,
                       int64_t max_emb_dim) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  using T = typename Elem<DT>::T;
  const int lane = lane_id();
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t i = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); i < n; i += (int64_t)gridDim.x * wpb) {
    const int64_t idx = indices[i];
    if (idx < 0) continue;
    const int64_t t = region == 0 ? scalar_table_id : table_ids[i];
    const int64_t vdim = table_value_dims[t];
    T* trow = reinterpret_cast<T*>(table_ptrs[t]) + idx * vdim;
    T* drow = reinterpret_cast<T*>(dense) + i * dense_stride;
    auto copy = [&](T* tp, T* dp, int64_t len) {
      for (int64_t e = lane; e < len; e += 64) { if (kLoad) dp[e] = tp[e]; else tp[e] = dp[e]; }
    };
    if (region == 0) copy(trow, drow, vdim < dense_dim ? vdim : dense_dim);
    else if (region == 1) { int64_t ed = table_emb_dims[t]; copy(trow, drow, ed < dense_dim ? ed : dense_dim); }
    else {
      int64_t ed = table_emb_dims[t];
      copy(trow, drow, ed);
      if (vdim > ed) copy(trow + ed, drow + max_emb_dim, vdim - ed);
    }
  }
}

// per-unique absolute row address: addr[u] = table_ptrs[tid[u]] + slot[u] * vdim * elem_bytes, 0 if slot < 0
__global__ void __launch_bounds__(256)
row_addr_kernel(int64_t n, const int64_t* __restrict__ n_dev, const int64_t* __restrict__ slots,
                const int64_t* __restrict__ table_ids, const int64_t* __restrict__ table_ptrs,
                const int64_t* __restrict__ table_value_dims, int elem_bytes, int64_t* __restrict__ addr) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = slots[i];
    const int64_t t = table_ids ? table_ids[i] : 0;
    addr[i] = s < 0 ? 0 : table_ptrs[t] + s * table_value_dims[t] * elem_bytes;
  }
}

// rows[i] (by address or dense) <- initializer(key_i); only where mask[i] != 0 (mask nullable) and,
// when `results` is given, where the insert result says the slot is NEW (Insert/Reclaim/Evict).
// Optional first half of the kernel (fused forward): the unlock pass of the hash-table insert (table.hip
// table_unlock_kernel, kernels.cuh:569-585) -- publish the real key into every slot the insert took and write the row
// address of EVERY key -- so that unlock and first-touch initialisation are one launch.  storage == nullptr: off.
struct UnlockArgs {
  uint8_t* storage;               // hash-table arena
  const int64_t* tbo;             // table bucket offsets
  int64_t C, stride;              // slots per bucket, bytes per bucket
  const int64_t* indices;         // slot of every key (-1: none)
  const int64_t* table_ptrs;      // value-table base addresses
  int elem_bytes;
  int64_t* row_addr_out;          // [n]
};

template <int DT>
__global__ void __launch_bounds__(256)
init_rows_kernel(InitArgs a, int64_t n, const int64_t* __restrict__ n_dev, const uint64_t* __restrict__ keys,
                 const int64_t* __restrict__ sel, const int64_t* __restrict__ row_addr, void* dense, int64_t dense_stride,
                 int emb_dim, int value_dim, const uint8_t* __restrict__ results, const uint8_t* __restrict__ skip,
                 const int64_t* __restrict__ table_ids, const int64_t* __restrict__ table_emb_dims,
                 const int64_t* __restrict__ table_value_dims, UnlockArgs u) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int lane = lane_id();
  const int64_t wpb = blockDim.x >> 6;
  // each wave owns 64 candidate rows: one lane per row tests the flags, then the whole wave
  // initialises the rows that need it, one after the other with coalesced stores.  In steady state
  // (nothing new) this is a single flag read per row.
  for (int64_t q0 = ((int64_t)blockIdx.x * wpb + (threadIdx.x >> 6)) * 64; q0 < n; q0 += (int64_t)gridDim.x * wpb * 64) {
    const int64_t q = q0 + lane;
    bool need = q < n;
    int64_t i = 0;
    int64_t my_addr = 0;   // unlock mode: the row address of my key, handed to the wave by shuffle below
    if (need) {
      i = sel ? sel[q] : q;
      const bool skipped = skip && skip[i];
      if (u.storage) {
        const int64_t idx = u.indices[i];
        const int64_t tid = table_ids ? table_ids[i] : 0;
        my_addr = idx < 0 ? 0 : u.table_ptrs[tid] + idx * table_value_dims[tid] * u.elem_bytes;
        u.row_addr_out[i] = my_addr;
        if (!skipped && idx >= 0) {
          const int64_t b = u.tbo[tid] + idx / u.C;
          __hip_atomic_store(reinterpret_cast<uint64_t*>(u.storage + b * u.stride) + idx % u.C, keys[i], __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (skipped) need = false;
      if (need && results) { uint8_t r = results[i]; need = (r == 0 || r == 1 || r == 3); }
    }
    uint64_t todo = __ballot(need);
    while (todo) {
      const int src = __ffsll((unsigned long long)todo) - 1;
      todo &= todo - 1;
      const uint32_t ilo = __shfl((int)(uint32_t)i, src, 64), ihi = __shfl((int)(uint32_t)((uint64_t)i >> 32), src, 64);
      const int64_t ii = (int64_t)(((uint64_t)ihi << 32) | ilo);
      void* rp;
      if (u.storage) {
        const uint32_t alo = __shfl((int)(uint32_t)my_addr, src, 64), ahi = __shfl((int)(uint32_t)((uint64_t)my_addr >> 32), src, 64);
        rp = reinterpret_cast<void*>((uintptr_t)(((uint64_t)ahi << 32) | alo));
      } else {
        rp = row_addr ? reinterpret_cast<void*>(row_addr[ii])
                      : (void*)(reinterpret_cast<typename Elem<DT>::T*>(dense) + ii * dense_stride);
      }
      if (!rp) continue;
      const uint64_t key = keys[ii];
      int ed = emb_dim, vd = value_dim;
      if (table_ids && table_emb_dims) { const int64_t t = table_ids[ii]; ed = (int)table_emb_dims[t]; vd = (int)table_value_dims[t]; }
      for (int e = lane; e < vd; e += 64)
        st1<DT>(rp, e, e < ed ? init_value(a, key, (uint32_t)e) : a.state_init);
    }
  }
}

// out[r, :] = sum_c in[c, r, :] (fp32 in, any out dtype): local reduction of the partial pooled sums that
// arrive from the W shards (the reduce half of the pooled output dist: TorchRec reduce-scatter in the
// reference, all-to-all + this sum here because xGMI is a point-to-point mesh).
template <int DDT>
__global__ void __launch_bounds__(256)
sum_chunks_kernel(const float* __restrict__ in, int64_t chunks, int64_t n, void* out) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t c = 0; c < chunks; ++c) add4(acc, *reinterpret_cast<const float4*>(in + c * n + i));
    st4<DDT>(out, i, acc);
  }
}
// the same for chunks that crossed the fabric in a 16-bit wire type (fp32 accumulation, one rounding at the end)
template <int SDT, int DDT>
__global__ void __launch_bounds__(256)
sum_chunks_typed_kernel(const void* __restrict__ in, int64_t chunks, int64_t n, void* out) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t c = 0; c < chunks; ++c) add4(acc, ld4<SDT>(in, c * n + i));
    st4<DDT>(out, i, acc);
  }
}

// the same with ONE chunk read in place: the rank's own block of partial sums never goes through the all-to-all (its
// send buffer IS the chunk); same sum order, chunk 0 .. W-1
template <int SDT, int DDT>
__global__ void __launch_bounds__(256)
sum_chunks_self_kernel(const void* __restrict__ in, int64_t chunks, int64_t n, const void* __restrict__ self_chunk, int64_t self_index,
                       void* out) {
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t c = 0; c < chunks; ++c) add4(acc, c == self_index ? ld4<SDT>(self_chunk, i) : ld4<SDT>(in, c * n + i));
    st4<DDT>(out, i, acc);
  }
}

}  // namespace mi355

using namespace mi355;
STAMP_EXPORT(mi355_debug_stamps_gather, g_st_gather)

static int lpr_log2_for(int D) {
  int l = 3;  // at least 8 lanes
  while ((4 << l) < D && l < 6) ++l;
  return l;
}

template <int SDT, int DDT>
static int launch_pooled(PoolArgs a, bool vec, hipStream_t stream) {
  if (!vec) {
    const int grid = grid_for(a.FB, 4, 1 << 20);
    hipLaunchKernelGGL((gather_pooled_scalar_kernel<SDT, DDT>), dim3(grid), dim3(256), 0, stream, a);
  } else {
    const int l = lpr_log2_for(a.D);
    const int ncol = (a.D + (4 << l) - 1) / (4 << l);
    const int nsub = 64 >> l;
    // bags per block = 4 waves x nsub groups x NB
    if (ncol <= 1) {
      // the software-pipelined kernel (gather_dev.h: gather_pooled_pipe).  Rounds 1-3 kept eight lock-step variants and a flat
      // double-buffered stream next to it (measured equal at best: 30.2 vs 29.7 us at C2); removed in round 6 with their knob.
      constexpr int KIT = 4;
      const int grid = grid_for(a.FB, 4 * nsub * KIT, 1 << 20);
#define MI355_POOL_P(UNRV)                                                                                                   \
  do {                                                                                                                       \
    if (a.row_addr && !a.rev) hipLaunchKernelGGL((gather_pooled_pipe_kernel<SDT, DDT, 2, UNRV, KIT>), dim3(grid), dim3(256), 0, stream, a, l); \
    else if (a.row_addr) hipLaunchKernelGGL((gather_pooled_pipe_kernel<SDT, DDT, 1, UNRV, KIT>), dim3(grid), dim3(256), 0, stream, a, l); \
    else hipLaunchKernelGGL((gather_pooled_pipe_kernel<SDT, DDT, 0, UNRV, KIT>), dim3(grid), dim3(256), 0, stream, a, l); \
  } while (0)
      // rows per load batch: short bags (C2: 1..10 keys) run best with 4 (8 waves / SIMD), long bags with 8 (2, the optimum of the
      // late-row gather of path (c) at C2, loses 0.5-1 % here at the 4x / 16x batches: profiles/r06_gather_variants.txt)
      if (a.n <= 8 * a.FB) MI355_POOL_P(4); else MI355_POOL_P(8);
#undef MI355_POOL_P
    } else if (ncol <= 2) {
      constexpr int NB = 2;
      if (a.row_addr) hipLaunchKernelGGL((gather_pooled_vec_kernel<SDT, DDT, 2, NB, true>), dim3(grid_for(a.FB, 4 * nsub * NB, 1 << 20)), dim3(256), 0, stream, a, l);
      else hipLaunchKernelGGL((gather_pooled_vec_kernel<SDT, DDT, 2, NB, false>), dim3(grid_for(a.FB, 4 * nsub * NB, 1 << 20)), dim3(256), 0, stream, a, l);
    } else {
      constexpr int NB = 1;
      if (a.row_addr) hipLaunchKernelGGL((gather_pooled_vec_kernel<SDT, DDT, 4, NB, true>), dim3(grid_for(a.FB, 4 * nsub * NB, 1 << 20)), dim3(256), 0, stream, a, l);
      else hipLaunchKernelGGL((gather_pooled_vec_kernel<SDT, DDT, 4, NB, false>), dim3(grid_for(a.FB, 4 * nsub * NB, 1 << 20)), dim3(256), 0, stream, a, l);
    }
  }
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

extern "C" {

// gather_embedding_pooled (dynamic_emb_op.cu:106-133).  Source is EITHER the dense unique-row
// tensor `src` (reference form) OR the table rows themselves through `row_addr` (fused form).
int mi355_gather_pooled(const void* src, int64_t src_stride, const int64_t* row_addr, int src_dtype,
                        const int64_t* reverse_indices, int64_t num_keys, const int64_t* offsets, int64_t num_bags, int64_t batch_size,
                        int combiner, int64_t dim, const int32_t* D_offsets, int64_t total_D, void* dst, int dst_dtype,
                        int aligned16, hipStream_t stream) {
  MI355_CHECK_ARG(src || row_addr, "src or row_addr required");
  MI355_CHECK_ARG(reverse_indices || row_addr, "reverse_indices may only be NULL with per-key row addresses");
  MI355_CHECK_ARG(batch_size > 0 && num_bags % batch_size == 0, "num_bags must be a multiple of batch_size");
  MI355_CHECK_ARG(dim > 0 && dim <= 1024, "embedding dim must be in (0, 1024]");
  MI355_CHECK_ARG(combiner == 0 || combiner == 1, "combiner must be 0 (sum) or 1 (mean)");
  if (num_bags == 0) return MI355_OK;
  PoolArgs a;
  a.src = src; a.src_stride = src_stride; a.row_addr = row_addr; a.rev = reverse_indices; a.offsets = offsets;
  a.D_offsets = D_offsets; a.dst = dst; a.FB = num_bags; a.n = num_keys; a.B = (int)batch_size; a.D = (int)dim;
  a.total_D = (int)total_D; a.combiner = combiner;
  if (num_keys == 0) {  // every bag is empty: the pooled output is all zeros
    if (hipMemsetAsync(dst, 0, (size_t)batch_size * total_D * dtype_bytes(dst_dtype), stream) != hipSuccess) {
      mi355_set_error("memset failed"); return MI355_ELAUNCH;
    }
    return MI355_OK;
  }
  const bool vec = aligned16 != 0;
  return MI355_DISPATCH_DTYPE(src_dtype, S, [&] {
    return MI355_DISPATCH_DTYPE(dst_dtype, Dd, [&] { return launch_pooled<S, Dd>(a, vec, stream); });
  });
}

// gather_embedding (dynamic_emb_op.cu:79-104), dense or row-address source; index may be null (identity)
int mi355_gather_rows(const void* src, int64_t src_stride, const int64_t* row_addr, int src_dtype, const int64_t* index,
                      int64_t n, const int64_t* n_dev, int64_t dim, void* dst, int64_t dst_stride, int dst_dtype,
                      int aligned16, hipStream_t stream) {
  MI355_CHECK_ARG(src || row_addr, "src or row_addr required");
  if (n == 0) return MI355_OK;
  const int l = lpr_log2_for((int)dim);
  const int R = 64 >> l;
  const int grid = grid_for(n, 4 * R, 1 << 20);
  return MI355_DISPATCH_DTYPE(src_dtype, S, [&] {
    return MI355_DISPATCH_DTYPE(dst_dtype, Dd, [&] {
      if (aligned16)
        hipLaunchKernelGGL((gather_rows_kernel<S, Dd, true>), dim3(grid), dim3(256), 0, stream, src, src_stride, row_addr, index, n,
                           n_dev, (int)dim, dst, dst_stride, l);
      else
        hipLaunchKernelGGL((gather_rows_kernel<S, Dd, false>), dim3(grid), dim3(256), 0, stream, src, src_stride, row_addr, index, n,
                           n_dev, (int)dim, dst, dst_stride, l);
      MI355_LAUNCH_CHECK();
      return MI355_OK;
    });
  });
}

// load_from_flat_table_{contiguous,emb,value} / store_to_flat_table_{contiguous,value}
int mi355_flat_table_copy(int is_load, int region, int64_t n, const int64_t* n_dev, void* dense, int64_t dense_dim,
                          int64_t dense_stride, int dtype, const int64_t* indices, const int64_t* table_ids,
                          int64_t scalar_table_id, const int64_t* table_ptrs, const int64_t* table_value_dims,
                          const int64_t* table_emb_dims, int64_t max_emb_dim, hipStream_t stream) {
  MI355_CHECK_ARG(region >= 0 && region <= 2, "region must be 0, 1 or 2");
  MI355_CHECK_ARG(region == 0 || table_ids, "table_ids required for region 1/2");
  if (n == 0) return MI355_OK;
  const int grid = grid_for(n, 4, 1 << 20);
  return MI355_DISPATCH_DTYPE(dtype, DT, [&] {
    if (is_load)
      hipLaunchKernelGGL((flat_table_copy_kernel<DT, true>), dim3(grid), dim3(256), 0, stream, region, n, n_dev, dense, dense_dim,
                         dense_stride, indices, table_ids, scalar_table_id, table_ptrs, table_value_dims, table_emb_dims, max_emb_dim);
    else
      hipLaunchKernelGGL((flat_table_copy_kernel<DT, false>), dim3(grid), dim3(256), 0, stream, region, n, n_dev, dense, dense_dim,
                         dense_stride, indices, table_ids, scalar_table_id, table_ptrs, table_value_dims, table_emb_dims, max_emb_dim);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  });
}

int mi355_sum_chunks(const float* in, int64_t chunks, int64_t numel_per_chunk, void* out, int out_dtype, hipStream_t stream) {
  MI355_CHECK_ARG(numel_per_chunk % 4 == 0, "chunk size must be a multiple of 4 elements");
  if (numel_per_chunk == 0) return MI355_OK;
  return MI355_DISPATCH_DTYPE(out_dtype, Dd, [&] {
    hipLaunchKernelGGL((sum_chunks_kernel<Dd>), dim3(grid_for(numel_per_chunk, 1024)), dim3(256), 0, stream, in, chunks,
                       numel_per_chunk, out);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  });
}

int mi355_sum_chunks_typed(const void* in, int in_dtype, int64_t chunks, int64_t numel_per_chunk, void* out, int out_dtype,
                           hipStream_t stream) {
  MI355_CHECK_ARG(numel_per_chunk % 4 == 0, "chunk size must be a multiple of 4 elements");
  if (numel_per_chunk == 0) return MI355_OK;
  return MI355_DISPATCH_DTYPE(in_dtype, Sd, [&] {
    return MI355_DISPATCH_DTYPE(out_dtype, Dd, [&] {
      hipLaunchKernelGGL((sum_chunks_typed_kernel<Sd, Dd>), dim3(grid_for(numel_per_chunk, 1024)), dim3(256), 0, stream, in,
                         chunks, numel_per_chunk, out);
      MI355_LAUNCH_CHECK();
      return MI355_OK;
    });
  });
}

int mi355_sum_chunks_self(const void* in, int in_dtype, int64_t chunks, int64_t numel_per_chunk, const void* self_chunk,
                          int64_t self_index, void* out, int out_dtype, hipStream_t stream) {
  MI355_CHECK_ARG(numel_per_chunk % 4 == 0, "chunk size must be a multiple of 4 elements");
  MI355_CHECK_ARG(self_chunk && self_index >= 0 && self_index < chunks, "self chunk");
  if (numel_per_chunk == 0) return MI355_OK;
  return MI355_DISPATCH_DTYPE(in_dtype, Sd, [&] {
    return MI355_DISPATCH_DTYPE(out_dtype, Dd, [&] {
      hipLaunchKernelGGL((sum_chunks_self_kernel<Sd, Dd>), dim3(grid_for(numel_per_chunk, 1024)), dim3(256), 0, stream, in,
                         chunks, numel_per_chunk, self_chunk, self_index, out);
      MI355_LAUNCH_CHECK();
      return MI355_OK;
    });
  });
}

int mi355_row_addresses(int64_t n, const int64_t* n_dev, const int64_t* slots, const int64_t* table_ids,
                        const int64_t* table_ptrs, const int64_t* table_value_dims, int elem_bytes, int64_t* row_addr,
                        hipStream_t stream) {
  if (n == 0) return MI355_OK;
  hipLaunchKernelGGL(row_addr_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, n, n_dev, slots, table_ids, table_ptrs,
                     table_value_dims, elem_bytes, row_addr);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// initializers (initializer.cu:64-212): mode 0 uniform(p0,p1) 1 normal(p0,p1) 2 trunc_normal(p0,p1,p2,p3)
// 3 const(p0) 4 debug(key % 100000).  Rows by address (row_addr) or dense buffer.
int mi355_init_rows(int mode, float p0, float p1, float p2, float p3, uint64_t seed, float state_init, int64_t n,
                    const int64_t* n_dev, const void* keys, const int64_t* sel, const int64_t* row_addr, void* dense,
                    int64_t dense_stride, int dtype, int64_t emb_dim, int64_t value_dim, const uint8_t* results,
                    const uint8_t* skip, const int64_t* table_ids, const int64_t* table_emb_dims,
                    const int64_t* table_value_dims, hipStream_t stream) {
  MI355_CHECK_ARG(mode >= 0 && mode <= 4, "bad initializer mode");
  MI355_CHECK_ARG(row_addr || dense, "row_addr or dense required");
  if (n == 0) return MI355_OK;
  InitArgs a{mode, p0, p1, p2, p3, seed, state_init};
  const int grid = grid_for(n, 4 * 64, 1 << 20);
  return MI355_DISPATCH_DTYPE(dtype, DT, [&] {
    hipLaunchKernelGGL((init_rows_kernel<DT>), dim3(grid), dim3(256), 0, stream, a, n, n_dev, (const uint64_t*)keys, sel, row_addr,
                       dense, dense_stride, (int)emb_dim, (int)value_dim, results, skip, table_ids, table_emb_dims,
                       table_value_dims, UnlockArgs{});
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  });
}

// unlock pass of mi355_table_insert + row addresses of all keys + first-touch initialisation of the new rows: one launch
int mi355i_unlock_init_rows(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores,
                            const int64_t* indices, const int64_t* table_ptrs, int elem_bytes, int64_t* row_addr_out, int mode,
                            float p0, float p1, float p2, float p3, uint64_t seed, float state_init, int64_t n,
                            const int64_t* n_dev, const void* keys, int dtype, int64_t emb_dim, int64_t value_dim,
                            const uint8_t* results, const uint8_t* skip, const int64_t* table_ids,
                            const int64_t* table_emb_dims, const int64_t* table_value_dims, hipStream_t stream) {
  MI355_CHECK_ARG(mode >= 0 && mode <= 4, "bad initializer mode");
  MI355_CHECK_ARG(storage && indices && table_ptrs && table_value_dims && row_addr_out, "unlock arguments required");
  if (n == 0) return MI355_OK;
  InitArgs a{mode, p0, p1, p2, p3, seed, state_init};
  UnlockArgs u{(uint8_t*)storage, table_bucket_offsets, C, (9 + 8 * num_scores) * C, indices, table_ptrs, elem_bytes, row_addr_out};
  const int grid = grid_for(n, 4 * 64, 1 << 20);
  return MI355_DISPATCH_DTYPE(dtype, DT, [&] {
    hipLaunchKernelGGL((init_rows_kernel<DT>), dim3(grid), dim3(256), 0, stream, a, n, n_dev, (const uint64_t*)keys,
                       (const int64_t*)nullptr, (const int64_t*)nullptr, (void*)nullptr, (int64_t)0, (int)emb_dim, (int)value_dim,
                       results, skip, table_ids, table_emb_dims, table_value_dims, u);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  });
}

}  // extern "C"
