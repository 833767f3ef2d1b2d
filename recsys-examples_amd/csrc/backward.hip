// Backward of the DynamicEmb lookup for gfx950: per-unique-row gradient reduction and the fused
// in-place optimizer update.
//
// Replaces (reference, corelib/dynamicemb/src/): LocalReduce two-stage kernels
// (lookup_backward.cu:32-374,555-628) behind reduce_grads (dynamic_emb_op.cu:159-285), and the
// fused optimizers update4_with_index_flat_table_kernel / update4_padded_buffer_kernel with
// Sgd/Adam/AdaGrad/RowWiseAdaGrad functors (optimizer_kernel.cuh:40-512, optimizer.cu:34-476).
//
// MI355X design
//  * Input is the CSR produced by mi355_group_by_unique (keys of the batch grouped by unique row),
//    NOT a radix-sorted (reverse_idx, gather_id) pair stream.
//  * One wave64 per unique row: LPR = D/4 lanes cover a gradient row, 64/LPR rows are summed per
//    wave step, bag loop unrolled 4x, fp32 accumulation.  The finished sum goes straight into a
//    "sink": either the dense unique_grads tensor (reference op `reduce_grads`) or the optimizer,
//    which reads-modifies-writes the table row in place -- the [Nu, D] gradient tensor never
//    touches HBM in the fused path.
//  * Zipf-hot rows (more than kHot occurrences) would serialise one wave for hundreds of
//    microseconds.  They are cut into 256-entry chunks; each chunk is summed by one block and
//    added to a per-hot-row fp32 accumulator with agent-scope atomics, a ticket counter elects the
//    last chunk, which applies the sink after an agent-scope acquire (release/acquire pattern of
//    the CDNA hand-off recipe).
#include "common.h"

namespace mi355 {

constexpr int kHot = 256;       // rows with more occurrences than this take the chunked path
constexpr int kChunk = 256;     // occurrences per hot chunk (one block)

enum Opt : int { kOptStore = 0, kOptSgd = 1, kOptAdam = 2, kOptAdagrad = 3, kOptRowwiseAdagrad = 4 };

struct OptArgs {
  int kind;
  float lr, beta1, beta2, eps, weight_decay;
  float bias1, bias2;   // 1 - beta^t (Adam), computed on the host
  int state_offset;     // elements from the row start to the first optimizer state (emb_dim, or max_emb_dim);
                        // < 0: the row's own embedding dim (flat tables with per-table dims)
  // kOptStore: write the reduced gradient to a dense tensor
  void* out; int64_t out_stride;
};

struct BwdArgs {
  const int32_t* ptr;       // [Nu+1]
  const int32_t* csr_src;   // [Nt] bag id (pooled) or key position (sequence)
  const void* grads;        // pooled: [B, total_D]; sequence: [Nt, D]
  int64_t grad_stride;      // elements per grad row
  const int64_t* offsets;   // bag offsets (pooled) or nullptr
  const int32_t* D_offsets; // per-feature column offsets or nullptr
  int B, D, combiner;       // combiner: -1 sequence, 0 sum, 1 mean
  const int64_t* row_addr;  // [Nu] absolute table row address, 0 = skip (kind != store)
  int64_t max_unique;
  const int64_t* nu_dev;    // number of uniques on the device (nullable)
  int round_grad;           // round the reduced gradient to the grad dtype first (what the reference's
                            // unique_grads tensor does, batched_dynamicemb_function.py:1242)
  // hot-row machinery
  int* n_hot; int* n_tasks; int* hot_u; int* hot_done; int* hot_nchunks; int* task_h; int* task_c; float* hot_acc;
  int max_hot, max_tasks;
};

__device__ __forceinline__ void add4s(float4& a, const float4& b, float s) {
  a.x += b.x * s; a.y += b.y * s; a.z += b.z * s; a.w += b.w * s;
}

// gradient row of one CSR entry
template <int GDT>
__device__ __forceinline__ void grad_src(const BwdArgs& a, int src, const void*& base, int64_t& off, int& Df, float& scale) {
  scale = 1.f;
  if (a.combiner < 0) { base = a.grads; off = (int64_t)src * a.grad_stride; Df = a.D; return; }
  const int f = src / a.B, b = src % a.B;
  int d0;
  if (a.D_offsets) { d0 = a.D_offsets[f]; Df = a.D_offsets[f + 1] - d0; } else { d0 = f * a.D; Df = a.D; }
  if (a.combiner == 1) {
    const int64_t L = a.offsets[src + 1] - a.offsets[src];
    if (L > 0) scale = 1.0f / (float)L;
  }
  base = a.grads; off = (int64_t)b * a.grad_stride + d0;
}

// sum of the gradient rows of CSR entries [lo, hi) -> acc (all R row groups folded, every lane of a
// column group holds the full sum).  kVec: 4 elements per lane; else 1 element per lane per column.
template <int GDT, int NCOL, bool kVec>
__device__ __forceinline__ void reduce_entries(const BwdArgs& a, int lo, int hi, int lpr_log2, float (&acc)[NCOL][4]) {
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2, R = 64 >> lpr_log2;
  const int sub = lane >> lpr_log2, c = lane & (LPR - 1);
  constexpr int W = kVec ? 4 : 1;
#pragma unroll
  for (int k = 0; k < NCOL; ++k)
#pragma unroll
    for (int w = 0; w < 4; ++w) acc[k][w] = 0.f;
  for (int p0 = lo + sub; p0 < hi; p0 += 4 * R) {
    const void* base[4]; int64_t off[4]; int Df[4]; float sc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = p0 + q * R;
      base[q] = nullptr; off[q] = 0; Df[q] = 0; sc[q] = 0.f;
      if (p < hi) grad_src<GDT>(a, a.csr_src[p], base[q], off[q], Df[q], sc[q]);
    }
    float4 v[4][NCOL];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < NCOL; ++k) {
        const int e = W * (c + k * LPR);
        v[q][k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (base[q] && e < Df[q]) {
          if (kVec) v[q][k] = ld4<GDT>(base[q], off[q] + e);
          else v[q][k].x = ld1<GDT>(base[q], off[q] + e);
        }
      }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < NCOL; ++k) {
        acc[k][0] += v[q][k].x * sc[q];
        if (kVec) { acc[k][1] += v[q][k].y * sc[q]; acc[k][2] += v[q][k].z * sc[q]; acc[k][3] += v[q][k].w * sc[q]; }
      }
  }
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int k = 0; k < NCOL; ++k)
#pragma unroll
      for (int w = 0; w < W; ++w) acc[k][w] += __shfl_xor(acc[k][w], o, 64);
}

// sum over the LPR lanes of a column group (every lane gets the total)
__device__ __forceinline__ float group_sum(float v, int lpr_log2) {
  for (int o = 1; o < (1 << lpr_log2); o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Apply the sink to one row.  g holds the full reduced gradient of the row in the column layout
// (element W*(c + k*LPR) + w).  Called by ALL 64 lanes (row-wise AdaGrad needs a group reduction);
// only row group 0 (sub == 0) touches memory.
template <int WDT, int GDT, int NCOL, bool kVec>
__device__ __forceinline__ void apply_sink(const OptArgs& o_in, int64_t u, void* row, int D, int lpr_log2, bool round_grad,
                                           float (&g)[NCOL][4]) {
  OptArgs o = o_in;
  if (o.state_offset < 0) o.state_offset = D;
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2;
  const int sub = lane >> lpr_log2, c = lane & (LPR - 1);
  constexpr int W = kVec ? 4 : 1;
  if (round_grad) {
#pragma unroll
    for (int k = 0; k < NCOL; ++k)
#pragma unroll
      for (int w = 0; w < W; ++w) g[k][w] = Elem<GDT>::rnd(g[k][w]);
  }
  if (o.kind == kOptStore) {
    if (sub != 0) return;
#pragma unroll
    for (int k = 0; k < NCOL; ++k) {
      const int e = W * (c + k * LPR);
      if (e < D) {
        if (kVec) st4<GDT>(o.out, u * o.out_stride + e, make_float4(g[k][0], g[k][1], g[k][2], g[k][3]));
        else st1<GDT>(o.out, u * o.out_stride + e, g[k][0]);
      }
    }
    return;
  }
  float rw_gt = 0.f;
  if (o.kind == kOptRowwiseAdagrad) {
    // G += mean(g^2) over the row (optimizer_kernel.cuh:320-346); state = ONE element at state_offset
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NCOL; ++k)
#pragma unroll
      for (int w = 0; w < W; ++w) { const int e = W * (c + k * LPR) + w; if (e < D) s += g[k][w] * g[k][w]; }
    s = group_sum(s, lpr_log2);
    if (row) {
      rw_gt = ld1<WDT>(row, o.state_offset) + s / (float)D;
      // every lane of the group has read the old state before anyone overwrites it
    }
  }
  if (sub != 0 || !row) return;
  if (o.kind == kOptRowwiseAdagrad && c == 0) st1<WDT>(row, o.state_offset, rw_gt);
#pragma unroll
  for (int k = 0; k < NCOL; ++k) {
    const int e0 = W * (c + k * LPR);
    if (e0 >= D) continue;
    float wv[4] = {0, 0, 0, 0}, m[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
    if (kVec) { float4 t = ld4<WDT>(row, e0); wv[0] = t.x; wv[1] = t.y; wv[2] = t.z; wv[3] = t.w; }
    else wv[0] = ld1<WDT>(row, e0);
    if (o.kind == kOptAdam || o.kind == kOptAdagrad) {
      if (kVec) { float4 t = ld4<WDT>(row, o.state_offset + e0); m[0] = t.x; m[1] = t.y; m[2] = t.z; m[3] = t.w; }
      else m[0] = ld1<WDT>(row, o.state_offset + e0);
    }
    if (o.kind == kOptAdam) {
      if (kVec) { float4 t = ld4<WDT>(row, o.state_offset + D + e0); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
      else v[0] = ld1<WDT>(row, o.state_offset + D + e0);
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const float gr = g[k][w];
      switch (o.kind) {
        case kOptSgd: wv[w] -= gr * o.lr; break;                                   // optimizer_kernel.cuh:44-79
        case kOptAdam: {                                                          // :81-213
          m[w] = o.beta1 * m[w] + (1.0f - o.beta1) * gr;
          v[w] = o.beta2 * v[w] + (1.0f - o.beta2) * gr * gr;
          const float mh = m[w] / o.bias1, vh = v[w] / o.bias2;
          wv[w] -= o.lr * (mh / (sqrtf(vh) + o.eps) + o.weight_decay * wv[w]);
        } break;
        case kOptAdagrad: {                                                       // :215-292
          m[w] += gr * gr;
          wv[w] -= o.lr * gr / (sqrtf(m[w]) + o.eps);
        } break;
        default: wv[w] -= o.lr * gr / (sqrtf(rw_gt) + o.eps); break;              // row-wise, :315-411
      }
    }
    if (kVec) st4<WDT>(row, e0, make_float4(wv[0], wv[1], wv[2], wv[3])); else st1<WDT>(row, e0, wv[0]);
    if (o.kind == kOptAdam || o.kind == kOptAdagrad) {
      if (kVec) st4<WDT>(row, o.state_offset + e0, make_float4(m[0], m[1], m[2], m[3])); else st1<WDT>(row, o.state_offset + e0, m[0]);
    }
    if (o.kind == kOptAdam) {
      if (kVec) st4<WDT>(row, o.state_offset + D + e0, make_float4(v[0], v[1], v[2], v[3])); else st1<WDT>(row, o.state_offset + D + e0, v[0]);
    }
  }
}

template <int WDT, int GDT, int NCOL, bool kVec>
__global__ void __launch_bounds__(256) bwd_rows_kernel(BwdArgs a, OptArgs o, int lpr_log2) {
  int64_t nu = a.max_unique;
  if (a.nu_dev) { int64_t m = *a.nu_dev; nu = m < nu ? m : nu; }
  const int lane = lane_id();
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t u = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); u < nu; u += (int64_t)gridDim.x * wpb) {
    const int lo = a.ptr[u], hi = a.ptr[u + 1];
    const int cnt = hi - lo;
    void* row = o.kind == kOptStore ? nullptr : reinterpret_cast<void*>(a.row_addr[u]);
    if (cnt > kHot && a.n_hot) {
      // register the hot row and its chunks; the chunk kernel finishes it
      const int nchunks = (cnt + kChunk - 1) / kChunk;
      int h = 0, t0 = 0;
      if (lane == 0) { h = atomicAdd(a.n_hot, 1); t0 = atomicAdd(a.n_tasks, nchunks); }
      h = __shfl(h, 0, 64); t0 = __shfl(t0, 0, 64);
      if (h < a.max_hot && t0 + nchunks <= a.max_tasks) {
        if (lane == 0) { a.hot_u[h] = (int)u; a.hot_done[h] = 0; a.hot_nchunks[h] = nchunks; }
        for (int e = lane; e < a.D; e += 64) a.hot_acc[(int64_t)h * a.D + e] = 0.f;
        for (int cc = lane; cc < nchunks; cc += 64) { a.task_h[t0 + cc] = h; a.task_c[t0 + cc] = cc; }
        continue;
      }
      // (cannot happen with the documented workspace sizes) fall through to the serial path
    }
    float g[NCOL][4];
    reduce_entries<GDT, NCOL, kVec>(a, lo, hi, lpr_log2, g);
    if (cnt == 0 && o.kind != kOptStore) continue;  // unique without occurrences: nothing to apply
    int Drow = a.D;  // mixed dims: every occurrence of a row belongs to the same table, hence the same width
    if (a.D_offsets && a.combiner >= 0 && cnt > 0 && o.kind != kOptStore) {
      const int f = a.csr_src[lo] / a.B;
      Drow = a.D_offsets[f + 1] - a.D_offsets[f];
    }
    apply_sink<WDT, GDT, NCOL, kVec>(o, u, row, Drow, lpr_log2, a.round_grad != 0, g);
  }
}

template <int WDT, int GDT, int NCOL, bool kVec>
__global__ void __launch_bounds__(256) bwd_hot_kernel(BwdArgs a, OptArgs o, int lpr_log2) {
  extern __shared__ __attribute__((aligned(16))) float s_part[];  // [4 waves][D]
  __shared__ int s_last;
  const int ntasks = *a.n_tasks < a.max_tasks ? *a.n_tasks : a.max_tasks;
  const int lane = lane_id(), wv = threadIdx.x >> 6;
  const int LPR = 1 << lpr_log2;
  const int sub = lane >> lpr_log2, c = lane & (LPR - 1);
  constexpr int W = kVec ? 4 : 1;
  for (int task = blockIdx.x; task < ntasks; task += gridDim.x) {
    const int h = a.task_h[task], cc = a.task_c[task];
    const int u = a.hot_u[h];
    const int lo = a.ptr[u] + cc * kChunk;
    int hi = lo + kChunk; if (hi > a.ptr[u + 1]) hi = a.ptr[u + 1];
    const int per = (hi - lo + 3) / 4;
    int wlo = lo + wv * per, whi = wlo + per; if (whi > hi) whi = hi; if (wlo > hi) wlo = hi;
    float g[NCOL][4];
    reduce_entries<GDT, NCOL, kVec>(a, wlo, whi, lpr_log2, g);
    if (sub == 0) {
#pragma unroll
      for (int k = 0; k < NCOL; ++k)
#pragma unroll
        for (int w = 0; w < W; ++w) { const int e = W * (c + k * LPR) + w; if (e < a.D) s_part[wv * a.D + e] = g[k][w]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < a.D; e += blockDim.x) {
      float s = s_part[e] + s_part[a.D + e] + s_part[2 * a.D + e] + s_part[3 * a.D + e];
      __hip_atomic_fetch_add(&a.hot_acc[(int64_t)h * a.D + e], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // publish: drain this block's atomics, then take a ticket (CDNA hand-off recipe, counter form)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      int t = __hip_atomic_fetch_add(&a.hot_done[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (t == a.hot_nchunks[h] - 1);
      if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (s_last && wv == 0) {
      float gg[NCOL][4];
#pragma unroll
      for (int k = 0; k < NCOL; ++k)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const int e = W * (c + k * LPR) + w;
          gg[k][w] = (w < W && e < a.D)
                         ? __hip_atomic_load(&a.hot_acc[(int64_t)h * a.D + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                         : 0.f;
        }
      void* row = o.kind == kOptStore ? nullptr : reinterpret_cast<void*>(a.row_addr[u]);
      int Drow = a.D;
      if (a.D_offsets && a.combiner >= 0 && o.kind != kOptStore) {
        const int f = a.csr_src[a.ptr[u]] / a.B;
        Drow = a.D_offsets[f + 1] - a.D_offsets[f];
      }
      apply_sink<WDT, GDT, NCOL, kVec>(o, u, row, Drow, lpr_log2, a.round_grad != 0, gg);
    }
    __syncthreads();
  }
}

// optimizer on dense unique gradients ({sgd,adam,adagrad,rowwise_adagrad}_update_for_flat_table /
// ..._for_padded_buffer): one wave per row, same arithmetic as the fused path.
template <int WDT, int GDT, int NCOL, bool kVec>
__global__ void __launch_bounds__(256)
opt_rows_kernel(OptArgs o, const void* grads, int64_t grad_stride, int64_t n, const int64_t* __restrict__ n_dev,
                const int64_t* __restrict__ row_addr, void* dense_rows, int64_t dense_stride, int D, int lpr_log2) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2;
  const int c = lane & (LPR - 1);
  constexpr int W = kVec ? 4 : 1;
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t u = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); u < n; u += (int64_t)gridDim.x * wpb) {
    void* row = row_addr ? reinterpret_cast<void*>(row_addr[u])
                         : (void*)(reinterpret_cast<typename Elem<WDT>::T*>(dense_rows) + u * dense_stride);
    float g[NCOL][4];
#pragma unroll
    for (int k = 0; k < NCOL; ++k) {
      const int e = W * (c + k * LPR);
      g[k][0] = g[k][1] = g[k][2] = g[k][3] = 0.f;
      if (e < D) {
        if (kVec) { float4 t = ld4<GDT>(grads, u * grad_stride + e); g[k][0] = t.x; g[k][1] = t.y; g[k][2] = t.z; g[k][3] = t.w; }
        else g[k][0] = ld1<GDT>(grads, u * grad_stride + e);
      }
    }
    apply_sink<WDT, GDT, NCOL, kVec>(o, u, row, D, lpr_log2, false, g);
  }
}

}  // namespace mi355

using namespace mi355;

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static int lpr_log2_for(int D, bool vec) {
  int l = 3;
  const int per = vec ? 4 : 1;
  while ((per << l) < D && l < 6) ++l;
  return l;
}

struct HotWs { int max_hot, max_tasks; int64_t bytes; };
static HotWs hot_ws(int64_t n, int64_t D) {
  HotWs w;
  w.max_hot = (int)(n / kHot + 1);
  w.max_tasks = (int)(n / kChunk + w.max_hot + 1);
  w.bytes = 256 + 3 * align_up(4LL * w.max_hot, 256) + 2 * align_up(4LL * w.max_tasks, 256) + align_up(4LL * w.max_hot * D, 256);
  return w;
}

template <int WDT, int GDT>
static int launch_bwd(BwdArgs a, OptArgs o, bool vec, hipStream_t stream) {
  const int l = lpr_log2_for(a.D, vec);
  const int per = vec ? 4 : 1;
  const int ncol = (a.D + (per << l) - 1) / (per << l);
  const int grid = grid_for(a.max_unique, 4, 1 << 20);
  const int hgrid = a.n_hot ? (a.max_tasks < 4096 ? a.max_tasks : 4096) : 0;
  const size_t smem = 4 * (size_t)a.D * sizeof(float);
#define MI355_BWD_LAUNCH(NC, V)                                                                                         \
  do {                                                                                                                  \
    hipLaunchKernelGGL((bwd_rows_kernel<WDT, GDT, NC, V>), dim3(grid), dim3(256), 0, stream, a, o, l);                   \
    if (hgrid) hipLaunchKernelGGL((bwd_hot_kernel<WDT, GDT, NC, V>), dim3(hgrid), dim3(256), smem, stream, a, o, l);     \
  } while (0)
  if (vec) {
    if (ncol <= 1) MI355_BWD_LAUNCH(1, true); else if (ncol <= 2) MI355_BWD_LAUNCH(2, true); else MI355_BWD_LAUNCH(4, true);
  } else {
    if (ncol <= 1) MI355_BWD_LAUNCH(1, false); else if (ncol <= 2) MI355_BWD_LAUNCH(2, false);
    else if (ncol <= 4) MI355_BWD_LAUNCH(4, false); else MI355_BWD_LAUNCH(16, false);
  }
#undef MI355_BWD_LAUNCH
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

template <int WDT, int GDT>
static int launch_opt(OptArgs o, const void* grads, int64_t grad_stride, int64_t n, const int64_t* n_dev,
                      const int64_t* row_addr, void* dense_rows, int64_t dense_stride, int D, bool vec, hipStream_t stream) {
  const int l = lpr_log2_for(D, vec);
  const int per = vec ? 4 : 1;
  const int ncol = (D + (per << l) - 1) / (per << l);
  const int grid = grid_for(n, 4, 1 << 20);
#define MI355_OPT_LAUNCH(NC, V)                                                                                    \
  hipLaunchKernelGGL((opt_rows_kernel<WDT, GDT, NC, V>), dim3(grid), dim3(256), 0, stream, o, grads, grad_stride, n, \
                     n_dev, row_addr, dense_rows, dense_stride, D, l)
  if (vec) {
    if (ncol <= 1) MI355_OPT_LAUNCH(1, true); else if (ncol <= 2) MI355_OPT_LAUNCH(2, true); else MI355_OPT_LAUNCH(4, true);
  } else {
    if (ncol <= 1) MI355_OPT_LAUNCH(1, false); else if (ncol <= 2) MI355_OPT_LAUNCH(2, false);
    else if (ncol <= 4) MI355_OPT_LAUNCH(4, false); else MI355_OPT_LAUNCH(16, false);
  }
#undef MI355_OPT_LAUNCH
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

extern "C" {

int64_t mi355_backward_workspace_bytes(int64_t num_keys, int64_t dim) { return hot_ws(num_keys, dim).bytes; }

// Fused backward over the CSR of mi355_group_by_unique.
//   opt_kind 0: store the reduced gradients to `out` [max_unique, out_stride] in the grad dtype (reduce_grads)
//   opt_kind 1..4: SGD / Adam / AdaGrad / row-wise AdaGrad applied in place on the table rows `row_addr`
int mi355_backward_fused(const int32_t* ptr, const int32_t* csr_src, int64_t num_keys, int64_t max_unique,
                         const int64_t* nu_dev, const void* grads, int64_t grad_stride, int grad_dtype,
                         const int64_t* offsets, const int32_t* D_offsets, int64_t batch_size, int64_t dim, int combiner,
                         const int64_t* row_addr, int weight_dtype, int opt_kind, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int64_t iter_num, int64_t state_offset, int round_grad,
                         void* out, int64_t out_stride, int aligned16, void* workspace, int64_t workspace_bytes,
                         hipStream_t stream) {
  MI355_CHECK_ARG(opt_kind >= 0 && opt_kind <= 4, "bad optimizer kind");
  MI355_CHECK_ARG(opt_kind != kOptStore || out, "out required for opt_kind 0");
  MI355_CHECK_ARG(opt_kind == kOptStore || row_addr, "row_addr required for optimizer kinds");
  MI355_CHECK_ARG(dim > 0 && dim <= 1024, "embedding dim must be in (0, 1024]");
  MI355_CHECK_ARG(combiner < 0 || (offsets && batch_size > 0), "pooled mode needs offsets and batch_size");
  if (max_unique == 0) return MI355_OK;
  BwdArgs a{};
  a.ptr = ptr; a.csr_src = csr_src; a.grads = grads; a.grad_stride = grad_stride; a.offsets = offsets; a.D_offsets = D_offsets;
  a.B = (int)batch_size; a.D = (int)dim; a.combiner = combiner; a.row_addr = row_addr; a.max_unique = max_unique;
  a.nu_dev = nu_dev; a.round_grad = round_grad;
  if (workspace) {
    HotWs hw = hot_ws(num_keys, dim);
    MI355_CHECK_ARG(workspace_bytes >= hw.bytes, "workspace too small");
    uint8_t* w = (uint8_t*)workspace;
    a.n_hot = (int*)w; a.n_tasks = (int*)(w + 8); w += 256;
    a.hot_u = (int*)w; w += align_up(4LL * hw.max_hot, 256);
    a.hot_done = (int*)w; w += align_up(4LL * hw.max_hot, 256);
    a.hot_nchunks = (int*)w; w += align_up(4LL * hw.max_hot, 256);
    a.task_h = (int*)w; w += align_up(4LL * hw.max_tasks, 256);
    a.task_c = (int*)w; w += align_up(4LL * hw.max_tasks, 256);
    a.hot_acc = (float*)w;
    a.max_hot = hw.max_hot; a.max_tasks = hw.max_tasks;
    if (hipMemsetAsync(workspace, 0, 256, stream) != hipSuccess) { mi355_set_error("memset failed"); return MI355_ELAUNCH; }
  }
  OptArgs o{};
  o.kind = opt_kind; o.lr = lr; o.beta1 = beta1; o.beta2 = beta2; o.eps = eps; o.weight_decay = weight_decay;
  o.bias1 = (float)(1.0 - pow((double)beta1, (double)iter_num));
  o.bias2 = (float)(1.0 - pow((double)beta2, (double)iter_num));
  o.state_offset = (int)state_offset; o.out = out; o.out_stride = out_stride;
  const bool vec = aligned16 != 0;
  return MI355_DISPATCH_DTYPE(weight_dtype, Wd, [&] {
    return MI355_DISPATCH_DTYPE(grad_dtype, Gd, [&] { return launch_bwd<Wd, Gd>(a, o, vec, stream); });
  });
}

// optimizer step on dense unique gradients [n, grad_stride]; rows by address (flat table) or in a
// dense padded buffer [n, dense_stride] (update_for_padded_buffer, optimizer.cu:242-413).
int mi355_optimizer_update(int opt_kind, const void* grads, int64_t grad_stride, int grad_dtype, int64_t n,
                           const int64_t* n_dev, const int64_t* row_addr, void* dense_rows, int64_t dense_stride,
                           int weight_dtype, int64_t dim, int64_t state_offset, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int64_t iter_num, int aligned16, hipStream_t stream) {
  MI355_CHECK_ARG(opt_kind >= 1 && opt_kind <= 4, "bad optimizer kind");
  MI355_CHECK_ARG(row_addr || dense_rows, "row_addr or dense_rows required");
  MI355_CHECK_ARG(dim > 0 && dim <= 1024, "embedding dim must be in (0, 1024]");
  if (n == 0) return MI355_OK;
  OptArgs o{};
  o.kind = opt_kind; o.lr = lr; o.beta1 = beta1; o.beta2 = beta2; o.eps = eps; o.weight_decay = weight_decay;
  o.bias1 = (float)(1.0 - pow((double)beta1, (double)iter_num));
  o.bias2 = (float)(1.0 - pow((double)beta2, (double)iter_num));
  o.state_offset = (int)state_offset;
  const bool vec = aligned16 != 0;
  return MI355_DISPATCH_DTYPE(weight_dtype, Wd, [&] {
    return MI355_DISPATCH_DTYPE(grad_dtype, Gd, [&] {
      return launch_opt<Wd, Gd>(o, grads, grad_stride, n, n_dev, row_addr, dense_rows, dense_stride, (int)dim, vec, stream);
    });
  });
}

}  // extern "C"
