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

namespace mi355 {

struct PoolArgs {
  const void* src;             // dense source [*, src_stride] (row_addr == nullptr)
  int64_t src_stride;          // elements
  const int64_t* row_addr;     // per-unique absolute row address (0 = missing row -> contributes 0)
  const int64_t* rev;          // [Nt] key -> unique
  const int64_t* offsets;      // [FB+1] feature-major bag offsets
  const int32_t* D_offsets;    // [F+1] or nullptr (uniform D)
  void* dst;                   // [B, total_D]
  int64_t FB;
  int B;
  int D;                       // uniform dim, or max_D when D_offsets != nullptr
  int total_D;
  int combiner;                // 0 sum, 1 mean
};

template <int SDT>
__device__ __forceinline__ const void* src_row(const PoolArgs& a, int64_t u) {
  if (a.row_addr) return reinterpret_cast<const void*>(a.row_addr[u]);
  return reinterpret_cast<const typename Elem<SDT>::T*>(a.src) + u * a.src_stride;
}

__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// vectorised: D_f % 4 == 0, rows 16-B (fp32) / 8-B (16-bit) aligned.  NCOL = ceil(D / (4*LPR)).
template <int SDT, int DDT, int NCOL>
__global__ void __launch_bounds__(256) gather_pooled_vec_kernel(PoolArgs a, int lpr_log2) {
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2;
  const int R = 64 >> lpr_log2;         // rows in flight per wave instruction
  const int sub = lane >> lpr_log2;
  const int c = lane & (LPR - 1);
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t bag = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); bag < a.FB; bag += (int64_t)gridDim.x * wpb) {
    const int f = (int)(bag / a.B), b = (int)(bag % a.B);
    int d0, Df;
    if (a.D_offsets) { d0 = a.D_offsets[f]; Df = a.D_offsets[f + 1] - d0; } else { d0 = f * a.D; Df = a.D; }
    const int64_t lo = a.offsets[bag], hi = a.offsets[bag + 1];
    float4 acc[NCOL];
#pragma unroll
    for (int k = 0; k < NCOL; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t j0 = lo + sub; j0 < hi; j0 += 4 * R) {
      const void* rp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t j = j0 + (int64_t)q * R;
        rp[q] = j < hi ? src_row<SDT>(a, a.rev[j]) : nullptr;
      }
      float4 v[4][NCOL];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < NCOL; ++k) {
          const int e = 4 * (c + k * LPR);
          v[q][k] = (rp[q] && e < Df) ? ld4<SDT>(rp[q], e) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < NCOL; ++k) add4(acc[k], v[q][k]);
    }
    // fold the R row groups
    for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < NCOL; ++k) {
        acc[k].x += __shfl_xor(acc[k].x, off, 64); acc[k].y += __shfl_xor(acc[k].y, off, 64);
        acc[k].z += __shfl_xor(acc[k].z, off, 64); acc[k].w += __shfl_xor(acc[k].w, off, 64);
      }
    }
    if (sub == 0) {
      const int64_t L = hi - lo;
      if (a.combiner == 1 && L > 0) {
        const float fl = (float)L;
#pragma unroll
        for (int k = 0; k < NCOL; ++k) { acc[k].x /= fl; acc[k].y /= fl; acc[k].z /= fl; acc[k].w /= fl; }
      }
#pragma unroll
      for (int k = 0; k < NCOL; ++k) {
        const int e = 4 * (c + k * LPR);
        if (e < Df) st4<DDT>(a.dst, (int64_t)b * a.total_D + d0 + e, acc[k]);
      }
    }
  }
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
      const void* rp = src_row<SDT>(a, a.rev[j]);
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

// ---- counter-based RNG (Philox4x32-10): value depends only on (seed, row key, element), never on
// the launch geometry, so first-touch initialisation is reproducible on any device layout ----
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(M0, ctr.x), lo0 = M0 * ctr.x;
    uint32_t hi1 = __umulhi(M1, ctr.z), lo1 = M1 * ctr.z;
    ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
    key.x += W0; key.y += W1;
  }
  return ctr;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

enum InitMode : int { kInitUniform = 0, kInitNormal = 1, kInitTruncNormal = 2, kInitConst = 3, kInitDebug = 4 };

struct InitArgs {
  int mode;
  float p0, p1, p2, p3;  // uniform: lower, upper; normal: mean, std; trunc: mean, std, lower, upper; const: value
  uint64_t seed;
  float state_init;      // initial optimizer-state value for elements [emb_dim, value_dim)
};

__device__ __forceinline__ float init_value(const InitArgs& a, uint64_t key, uint32_t e) {
  if (a.mode == kInitConst) return a.p0;
  if (a.mode == kInitDebug) return (float)(key % 100000ull);  // initializer.cuh:158-176
  uint4 r = philox4x32(make_uint4((uint32_t)key, (uint32_t)(key >> 32), e, 0u),
                       make_uint2((uint32_t)a.seed, (uint32_t)(a.seed >> 32)));
  if (a.mode == kInitUniform) return a.p0 + (a.p1 - a.p0) * u01(r.x);
  // Box-Muller; truncated normal by rejection over the 2 x 2 draws, then clamp (initializer.cuh)
  float n0 = sqrtf(-2.f * __logf(u01(r.x))) * __cosf(6.28318530718f * u01(r.y));
  float n1 = sqrtf(-2.f * __logf(u01(r.z))) * __cosf(6.28318530718f * u01(r.w));
  float v = a.p0 + a.p1 * n0;
  if (a.mode == kInitTruncNormal) {
    if (v < a.p2 || v > a.p3) v = a.p0 + a.p1 * n1;
    v = fminf(fmaxf(v, a.p2), a.p3);
  }
  return v;
}

// rows[i] (by address or dense) <- initializer(key_i); only where mask[i] != 0 (mask nullable) and,
// when `results` is given, where the insert result says the slot is NEW (Insert/Reclaim/Evict).
template <int DT>
__global__ void __launch_bounds__(256)
init_rows_kernel(InitArgs a, int64_t n, const int64_t* __restrict__ n_dev, const uint64_t* __restrict__ keys,
                 const int64_t* __restrict__ sel, const int64_t* __restrict__ row_addr, void* dense, int64_t dense_stride,
                 int emb_dim, int value_dim, const uint8_t* __restrict__ results, const uint8_t* __restrict__ skip,
                 const int64_t* __restrict__ table_ids, const int64_t* __restrict__ table_emb_dims,
                 const int64_t* __restrict__ table_value_dims) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int lane = lane_id();
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t q = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); q < n; q += (int64_t)gridDim.x * wpb) {
    const int64_t i = sel ? sel[q] : q;
    if (skip && skip[i]) continue;
    if (results) { uint8_t r = results[i]; if (!(r == 0 || r == 1 || r == 3)) continue; }
    void* rp = row_addr ? reinterpret_cast<void*>(row_addr[i])
                        : (void*)(reinterpret_cast<typename Elem<DT>::T*>(dense) + i * dense_stride);
    if (!rp) continue;
    const uint64_t key = keys[i];
    int ed = emb_dim, vd = value_dim;
    if (table_ids && table_emb_dims) { const int64_t t = table_ids[i]; ed = (int)table_emb_dims[t]; vd = (int)table_value_dims[t]; }
    for (int e = lane; e < vd; e += 64)
      st1<DT>(rp, e, e < ed ? init_value(a, key, (uint32_t)e) : a.state_init);
  }
}

}  // namespace mi355

using namespace mi355;

static int lpr_log2_for(int D) {
  int l = 3;  // at least 8 lanes
  while ((4 << l) < D && l < 6) ++l;
  return l;
}

template <int SDT, int DDT>
static int launch_pooled(PoolArgs a, bool vec, hipStream_t stream) {
  const int grid = grid_for(a.FB, 4, 1 << 20);
  if (!vec) {
    hipLaunchKernelGGL((gather_pooled_scalar_kernel<SDT, DDT>), dim3(grid), dim3(256), 0, stream, a);
  } else {
    const int l = lpr_log2_for(a.D);
    const int ncol = (a.D + (4 << l) - 1) / (4 << l);
    if (ncol <= 1) hipLaunchKernelGGL((gather_pooled_vec_kernel<SDT, DDT, 1>), dim3(grid), dim3(256), 0, stream, a, l);
    else if (ncol <= 2) hipLaunchKernelGGL((gather_pooled_vec_kernel<SDT, DDT, 2>), dim3(grid), dim3(256), 0, stream, a, l);
    else hipLaunchKernelGGL((gather_pooled_vec_kernel<SDT, DDT, 4>), dim3(grid), dim3(256), 0, stream, a, l);
  }
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

extern "C" {

// gather_embedding_pooled (dynamic_emb_op.cu:106-133).  Source is EITHER the dense unique-row
// tensor `src` (reference form) OR the table rows themselves through `row_addr` (fused form).
int mi355_gather_pooled(const void* src, int64_t src_stride, const int64_t* row_addr, int src_dtype,
                        const int64_t* reverse_indices, const int64_t* offsets, int64_t num_bags, int64_t batch_size,
                        int combiner, int64_t dim, const int32_t* D_offsets, int64_t total_D, void* dst, int dst_dtype,
                        int aligned16, hipStream_t stream) {
  MI355_CHECK_ARG(src || row_addr, "src or row_addr required");
  MI355_CHECK_ARG(batch_size > 0 && num_bags % batch_size == 0, "num_bags must be a multiple of batch_size");
  MI355_CHECK_ARG(dim > 0 && dim <= 1024, "embedding dim must be in (0, 1024]");
  MI355_CHECK_ARG(combiner == 0 || combiner == 1, "combiner must be 0 (sum) or 1 (mean)");
  if (num_bags == 0) return MI355_OK;
  PoolArgs a;
  a.src = src; a.src_stride = src_stride; a.row_addr = row_addr; a.rev = reverse_indices; a.offsets = offsets;
  a.D_offsets = D_offsets; a.dst = dst; a.FB = num_bags; a.B = (int)batch_size; a.D = (int)dim;
  a.total_D = (int)total_D; a.combiner = combiner;
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
  const int grid = grid_for(n, 4, 1 << 20);
  return MI355_DISPATCH_DTYPE(dtype, DT, [&] {
    hipLaunchKernelGGL((init_rows_kernel<DT>), dim3(grid), dim3(256), 0, stream, a, n, n_dev, (const uint64_t*)keys, sel, row_addr,
                       dense, dense_stride, (int)emb_dim, (int)value_dim, results, skip, table_ids, table_emb_dims,
                       table_value_dims);
    MI355_LAUNCH_CHECK();
    return MI355_OK;
  });
}

}  // extern "C"
