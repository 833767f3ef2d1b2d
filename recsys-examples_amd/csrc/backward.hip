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
//  * One LPR-lane group (LPR = D/4: half a wave at D = 128) per unique row; 4 CSR entries per round,
//    every load of a round unconditional (padding entries read a zero row) so the round is three
//    back-to-back batches of independent loads, fp32 accumulation in CSR order.  The finished sum goes
//    straight into a "sink": either the dense unique_grads tensor (reference op `reduce_grads`) or
//    the optimizer, which reads-modifies-writes the table row in place -- the [Nu, D] gradient
//    tensor never touches HBM in the fused path.
//  * Zipf-hot rows (more than kHot occurrences) would serialise one lane group for hundreds of
//    microseconds.  The CSR builder cuts them into kChunk-entry tasks (hot.h); the FIRST blocks of
//    the same launch take the tasks (one wave each, so the long work starts first), add their partial
//    sums to a per-hot-row fp32 accumulator with agent-scope atomics, and a ticket counter elects the
//    last task, which applies the sink after an agent-scope acquire (CDNA hand-off recipe).
#include "common.h"
#include "hot.h"
#include "internal.h"
#include "stamps.h"

namespace mi355 {

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
  int n_entries;            // number of CSR entries (= num_keys)
  HotList hot;              // hot-row task list built by mi355_group_by_unique (hot.n_tasks == nullptr: none)
  int hot_blocks;           // leading blocks of the launch that serve the hot tasks
  int wave_blocks;          // blocks after them whose waves serve the one-wave rows (hot.wave_*)
  // consecutive row groups one lane group walks (<= PIPE_KIT).  Round 6 (profiles/r06_bwd_variants.txt): the walk COMPILED for 2 groups
  // beats the one for 4 by 1.5 us at C2 (47.7 -> 46.1 us; the 4-group code stopped after 2 groups at run time does not: 47.5), 4 beat
  // 2 by 1-1.5 % from the 4x batch on (fewer, longer walks keep more index loads ahead of the rows); 1 and 3 lose at C2, and so do
  // 1 or 3 entries per round and 1, 3 or 4 rows per group.  Hence two instantiations of the SGD kernel, chosen by the batch size.
  int kit;
  int one_feature;          // pooled, one feature (num_bags == batch): a source id IS the gradient row -- no division by the batch
  const int32_t* tile_bags; // nullable.  Round 3 (fused forward, csrc/fused_fwd.hip): a CSR entry e < 0 is a REFERENCE -- the
                            // source id is tile_bags[~e] (occurrences of one key inside one 2048-key tile are listed there by
                            // the probe kernel; the partition kernel only stores the references, nobody copies the lists)
};

// source id of a CSR entry value (see BwdArgs::tile_bags); the reference form costs one more dependent load
__device__ __forceinline__ int entry_src(const BwdArgs& a, int sv) {
  if (a.tile_bags && sv < 0) sv = a.tile_bags[~sv];
  return sv;
}

#ifndef PIPE_NB
#define PIPE_NB 2
#endif
#ifndef PIPE_RPR
#define PIPE_RPR 2
#endif
#ifndef HOT_UNR
#define HOT_UNR 8
#endif
#ifndef PIPE_KIT
#define PIPE_KIT 4      // groups per lane group the walk is compiled for; BwdArgs::kit says how many a launch uses
#endif
#ifndef BWD_KIT_SMALL
#define BWD_KIT_SMALL 2  // ... and for batches of <= 720 K keys (SGD, one column group)
#endif
constexpr int kBwdGroupsPerLaneGroup = PIPE_KIT;  // consecutive row groups walked by one lane group (regular rows)

__device__ __attribute__((aligned(16))) float g_zero_grad[1024];  // see g_zero_row in value_ops.hip

typedef const __attribute__((address_space(1))) char* gptr_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int DT>
__device__ __forceinline__ void ldNg(gptr_t p, bool vec, float (&o)[4]) {
  if (vec) {
    if constexpr (DT == kF32) {
      const f32x4_t t = *reinterpret_cast<const __attribute__((address_space(1))) f32x4_t*>(p);
      o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    } else {
      const u32x2_t r = *reinterpret_cast<const __attribute__((address_space(1))) u32x2_t*>(p);
      if constexpr (DT == kBF16) {
        o[0] = __uint_as_float(r.x << 16); o[1] = __uint_as_float(r.x & 0xffff0000u);
        o[2] = __uint_as_float(r.y << 16); o[3] = __uint_as_float(r.y & 0xffff0000u);
      } else {
        o[0] = f16_to_f32((uint16_t)(r.x & 0xffff)); o[1] = f16_to_f32((uint16_t)(r.x >> 16));
        o[2] = f16_to_f32((uint16_t)(r.y & 0xffff)); o[3] = f16_to_f32((uint16_t)(r.y >> 16));
      }
    }
  } else {
    if constexpr (DT == kF32) o[0] = *reinterpret_cast<const __attribute__((address_space(1))) float*>(p);
    else {
      const uint16_t h = *reinterpret_cast<const __attribute__((address_space(1))) uint16_t*>(p);
      o[0] = DT == kBF16 ? bf16_to_f32(h) : f16_to_f32(h);
    }
    o[1] = o[2] = o[3] = 0.f;
  }
}

// Sums of the gradient rows of NB CSR ranges [lo[b], hi[b]) by ONE lane group, RPR entries of every
// range per round; every lane owns the columns W*(c + k*LPR) .. +W-1 of the rows.  Branch free (see the
// kernel comment in value_ops.hip): an empty or exhausted range keeps reading a zero row.
template <int GDT, int NCOL, bool kVec, int NB, int RPR>
__device__ __forceinline__ void reduce_multi(const BwdArgs& a, const int (&lo)[NB], const int (&hi)[NB], int lpr_log2,
                                             float (&acc)[NB][NCOL][4]) {
  const int LPR = 1 << lpr_log2;
  const int c = lane_id() & (LPR - 1);
  constexpr int W = kVec ? 4 : 1;
  constexpr int EB = GDT == kF32 ? 4 : 2;
  const gptr_t zero = (gptr_t)(uintptr_t)g_zero_grad;
  int maxcnt = 0;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int k = 0; k < NCOL; ++k)
#pragma unroll
      for (int w = 0; w < 4; ++w) acc[b][k][w] = 0.f;
    maxcnt = hi[b] - lo[b] > maxcnt ? hi[b] - lo[b] : maxcnt;
  }
  for (int r = 0; r < maxcnt; r += RPR) {
    int src[NB][RPR];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < RPR; ++q) {
        int p = lo[b] + r + q;
        p = p < hi[b] ? p : hi[b] - 1;
        p = p < lo[b] ? lo[b] : p;
        p = p < a.n_entries ? p : a.n_entries - 1;
        src[b][q] = entry_src(a, a.csr_src[p]);
      }
    uintptr_t base[NB][RPR]; int Df[NB][RPR]; float sc[NB][RPR];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < RPR; ++q) {
        const bool ok = lo[b] + r + q < hi[b];
        Df[b][q] = a.D; sc[b][q] = 1.f;
        int64_t off;
        const int sv = src[b][q];
        if (a.combiner < 0) {
          off = (int64_t)sv * a.grad_stride;
        } else {
          const int f = sv / a.B, bb = sv - f * a.B;
          int d0 = f * a.D;
          if (a.D_offsets) { d0 = a.D_offsets[f]; Df[b][q] = a.D_offsets[f + 1] - d0; }
          if (a.combiner == 1) {
            const int64_t L = a.offsets[sv + 1] - a.offsets[sv];
            sc[b][q] = L > 0 ? 1.0f / (float)L : 1.f;
          }
          off = (int64_t)bb * a.grad_stride + d0;
        }
        base[b][q] = ok ? (uintptr_t)a.grads + (uintptr_t)(off * EB) : 0;
      }
    float v[NB][RPR][NCOL][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < RPR; ++q)
#pragma unroll
        for (int k = 0; k < NCOL; ++k) {
          const int e = W * (c + k * LPR);
          const gptr_t p = (base[b][q] != 0 && e < Df[b][q]) ? (gptr_t)(base[b][q] + (uintptr_t)(e * EB)) : zero;
          ldNg<GDT>(p, kVec, v[b][q][k]);
        }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < RPR; ++q)
#pragma unroll
        for (int k = 0; k < NCOL; ++k)
#pragma unroll
          for (int w = 0; w < W; ++w) acc[b][k][w] += v[b][q][k][w] * sc[b][q];
  }
}

// Sum of the gradient rows of the CSR entries [slo, shi) by ONE lane group, for slices of up to `per` entries
// (`per` is wave uniform).  The entry indices are fetched LPR at a time -- one per lane, coalesced -- and every
// lane turns ITS entry into a gradient-row address; the addresses are then handed round the group with
// shuffles, so the UNR gradient rows of a batch are independent loads instead of a csr -> address -> row chain
// per round (a hot task used to be 16 dependent latencies long).  Entries are added in CSR order.
template <int GDT, int NCOL, bool kVec, int UNR>
__device__ __forceinline__ void reduce_chunk(const BwdArgs& a, int slo, int shi, int per, int lpr_log2, float (&acc)[NCOL][4]) {
  const int LPR = 1 << lpr_log2;
  const int c = lane_id() & (LPR - 1);
  constexpr int W = kVec ? 4 : 1;
  constexpr int EB = GDT == kF32 ? 4 : 2;
  const gptr_t zero = (gptr_t)(uintptr_t)g_zero_grad;
#pragma unroll
  for (int k = 0; k < NCOL; ++k)
#pragma unroll
    for (int w = 0; w < 4; ++w) acc[k][w] = 0.f;
  const bool need_df = a.D_offsets != nullptr && a.combiner >= 0;
  const bool need_sc = a.combiner == 1;
  for (int j0 = 0; j0 < per; j0 += LPR) {
    // my entry of this sub-chunk -> address of its gradient row (0: none), its width and scale
    int e = slo + j0 + c;
    const bool mine = e < shi;
    e = e < a.n_entries ? e : a.n_entries - 1;
    int sv = a.csr_src[e < 0 ? 0 : e];
    if (a.tile_bags) {   // (wave uniform) references are the rule in hot rows: the second load is unconditional
      const int t = a.tile_bags[sv < 0 ? ~sv : 0];
      sv = sv < 0 ? t : sv;
    }
    int myDf = a.D;
    float mysc = 1.f;
    int64_t off;
    if (a.combiner < 0) {
      off = (int64_t)sv * a.grad_stride;
    } else {
      const int f = a.one_feature ? 0 : sv / a.B, bb = sv - f * a.B;
      int d0 = f * a.D;
      if (need_df) { d0 = a.D_offsets[f]; myDf = a.D_offsets[f + 1] - d0; }
      if (need_sc) {
        const int64_t L = a.offsets[sv + 1] - a.offsets[sv];
        mysc = L > 0 ? 1.0f / (float)L : 1.f;
      }
      off = (int64_t)bb * a.grad_stride + d0;
    }
    const uintptr_t mybase = mine ? (uintptr_t)a.grads + (uintptr_t)(off * EB) : 0;
    const int blo = (int)(mybase & 0xffffffffu), bhi = (int)(mybase >> 32);
    int left = shi - (slo + j0);
    left = left < 0 ? 0 : left;
    for (int q0 = 0; q0 < LPR; q0 += UNR) {
      if (__ballot(q0 < left) == 0) break;  // wave uniform
      float v[UNR][NCOL][4], sc[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const uintptr_t base = (uintptr_t)(unsigned)__shfl(blo, q0 + q, LPR) | ((uintptr_t)(unsigned)__shfl(bhi, q0 + q, LPR) << 32);
        const int Df = need_df ? __shfl(myDf, q0 + q, LPR) : a.D;
        sc[q] = need_sc ? __shfl(mysc, q0 + q, LPR) : 1.f;
#pragma unroll
        for (int k = 0; k < NCOL; ++k) {
          const int el = W * (c + k * LPR);
          const gptr_t p = (base != 0 && el < Df) ? (gptr_t)(base + (uintptr_t)(el * EB)) : zero;
          ldNg<GDT>(p, kVec, v[q][k]);
        }
      }
#pragma unroll
      for (int q = 0; q < UNR; ++q)
#pragma unroll
        for (int k = 0; k < NCOL; ++k)
#pragma unroll
          for (int w = 0; w < W; ++w) acc[k][w] += v[q][k][w] * sc[q];
    }
  }
}

// sum over the LPR lanes of a column group (every lane gets the total)
__device__ __forceinline__ float group_sum(float v, int lpr_log2) {
  for (int o = 1; o < (1 << lpr_log2); o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Apply the sink to one row.  g holds the full reduced gradient of the row in the column layout
// (element W*(c + k*LPR) + w).  Called by whole waves (row-wise AdaGrad needs a lane-group reduction);
// lane groups with `active` == false run the shuffles but touch no memory.
template <int WDT, int GDT, int NCOL, bool kVec>
__device__ __forceinline__ void apply_sink(const OptArgs& o_in, int64_t u, void* row, int D, int lpr_log2, bool round_grad,
                                           float (&g)[NCOL][4], bool active) {
  OptArgs o = o_in;
  if (o.state_offset < 0) o.state_offset = D;
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2;
  const int c = lane & (LPR - 1);
  const int sub = active ? 0 : 1;  // inactive lane groups run the shuffles but touch no memory
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
        if (kVec) st4<WDT>(o.out, u * o.out_stride + e, make_float4(g[k][0], g[k][1], g[k][2], g[k][3]));
        else st1<WDT>(o.out, u * o.out_stride + e, g[k][0]);
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

// Regular rows, SGD on vector rows (the common case): a software pipeline over KIT consecutive groups of NB
// unique rows per lane group.  The walk of one group is a chain of dependent loads (ptr -> CSR entries ->
// gradient rows; row_addr -> table row); run group by group a wave spends most of its life waiting on the small
// index loads with nothing heavy in flight.  Here the index loads of group i+1 (CSR entries) and i+2 (ptr,
// row_addr) are issued BEFORE the heavy loads of group i, so each iteration has one wait with everything in
// flight (vmcnt retires in order: the index loads must be older than the heavy ones).  The first RPR entries of
// every row ride the pipeline; the few rows with more entries (<= hot threshold) take dependent extra rounds.
// kSgd: the table row is prefetched with the gradients and updated in registers; otherwise the reduced gradient goes
// through apply_sink (dense store, Adam, AdaGrad, row-wise AdaGrad), which fetches the row and its state itself.
template <int WDT, int GDT, int NCOL, int NB, int RPR, int KIT, bool kSgd>
__device__ __forceinline__ void rows_pipelined(const BwdArgs& a, const OptArgs& o, int lpr_log2, int64_t nu, int64_t sg) {
  const int LPR = 1 << lpr_log2;
  const int c = lane_id() & (LPR - 1);
  constexpr int EB = GDT == kF32 ? 4 : 2;
  constexpr int WB = WDT == kF32 ? 4 : 2;
  const gptr_t zero = (gptr_t)(uintptr_t)g_zero_grad;
  const bool hot_on = a.hot.n_tasks != nullptr;
  const int kit = a.kit < KIT ? a.kit : KIT;
  const int64_t ubase = sg * (int64_t)(kit * NB);
  if (ubase >= nu) return;  // no cross-lane operation below
  int pA[NB + 1];
  int64_t rA[NB];
  struct Idx { int lo[NB], cnt[NB], n[NB]; uintptr_t rowp[NB]; int src[NB][RPR]; };
  auto stageP = [&](int it) {
    const int64_t u0 = ubase + (int64_t)it * NB;
#pragma unroll
    for (int b = 0; b <= NB; ++b) { int64_t u = u0 + b; u = u < nu ? u : nu; pA[b] = a.ptr[u]; }
#pragma unroll
    for (int b = 0; b < NB; ++b) { int64_t u = u0 + b; u = u < nu ? u : nu - 1; rA[b] = (kSgd || a.row_addr) ? a.row_addr[u] : 0; }
  };
  auto entry = [&](int lo, int cnt, int q) {
    int e = lo + (q < cnt ? q : 0);
    e = e < a.n_entries ? e : a.n_entries - 1;
    return a.csr_src[e < 0 ? 0 : e];
  };
  auto stageS = [&](Idx& x) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = pA[b + 1] - pA[b];
      const bool work = n > 0 && !(hot_on && n > a.hot.khot);
      x.lo[b] = pA[b];
      x.cnt[b] = work ? n : 0;
      x.rowp[b] = (work && o.kind != kOptStore) ? (uintptr_t)rA[b] : 0;
      x.n[b] = n;
#pragma unroll
      for (int q = 0; q < RPR; ++q) x.src[b][q] = entry(pA[b], x.cnt[b], q);
    }
  };
  // one round of gradient rows: entries r .. r+RPR-1 of every row, indices already in `src`
  auto grads_round = [&](const Idx& x, const int (&src)[NB][RPR], int r, float (&acc)[NB][NCOL][4]) {
    uintptr_t base[NB][RPR];
    float sc[NB][RPR];
    int64_t L0[NB][RPR], L1[NB][RPR];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < RPR; ++q) {
        const int sv = src[b][q];
        int64_t off;
        if (a.combiner < 0) {
          off = (int64_t)sv * a.grad_stride;
        } else {
          // (gfx950 has no integer divide: `sv / B` by a run-time B is ~40 instructions per gathered gradient row)
          const int f = a.one_feature ? 0 : sv / a.B, bb = sv - f * a.B;
          off = (int64_t)bb * a.grad_stride + (int64_t)f * a.D;
          if (a.combiner == 1) { L0[b][q] = a.offsets[sv]; L1[b][q] = a.offsets[sv + 1]; }
        }
        base[b][q] = r + q < x.cnt[b] ? (uintptr_t)a.grads + (uintptr_t)(off * EB) : 0;
      }
    float v[NB][RPR][NCOL][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < RPR; ++q)
#pragma unroll
        for (int k = 0; k < NCOL; ++k) {
          const int e = 4 * (c + k * LPR);
          const gptr_t p = (base[b][q] != 0 && e < a.D) ? (gptr_t)(base[b][q] + (uintptr_t)(e * EB)) : zero;
          ldNg<GDT>(p, true, v[b][q][k]);
        }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < RPR; ++q) {
        sc[b][q] = 1.f;
        if (a.combiner == 1) { const int64_t L = L1[b][q] - L0[b][q]; sc[b][q] = L > 0 ? 1.0f / (float)L : 1.f; }
#pragma unroll
        for (int k = 0; k < NCOL; ++k)
#pragma unroll
          for (int w = 0; w < 4; ++w) acc[b][k][w] += v[b][q][k][w] * sc[b][q];
      }
  };
  Idx cur, nxt;
  stageP(0);
  stageS(cur);
  stageP(1);
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    if (it >= kit || ubase + (int64_t)it * NB >= nu) break;
    stageS(nxt);       // CSR entries of group it+1 (its ptr / row_addr arrived with the previous wait)
    stageP(it + 2);    // ptr / row_addr of group it+2
    if (a.tile_bags) {   // reference entries (two occurrences of a cold key in one tile: ~1 % of these rows): a rare dependent hop
      bool neg = false;
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < RPR; ++q) neg |= cur.src[b][q] < 0;
      if (__ballot(neg)) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int q = 0; q < RPR; ++q) cur.src[b][q] = entry_src(a, cur.src[b][q]);
      }
    }
    float wrow[NB][NCOL][4], acc[NB][NCOL][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int k = 0; k < NCOL; ++k) {
        if constexpr (kSgd) {
          const int e = 4 * (c + k * LPR);
          const gptr_t p = (cur.rowp[b] != 0 && e < a.D) ? (gptr_t)(cur.rowp[b] + (uintptr_t)(e * WB)) : zero;
          ldNg<WDT>(p, true, wrow[b][k]);
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) acc[b][k][w] = 0.f;
      }
    grads_round(cur, cur.src, 0, acc);
    int maxcnt = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) maxcnt = cur.cnt[b] > maxcnt ? cur.cnt[b] : maxcnt;
    for (int r = RPR; r < maxcnt; r += RPR) {
      int src[NB][RPR];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < RPR; ++q) src[b][q] = entry(cur.lo[b], cur.cnt[b], r + q);
      if (a.tile_bags) {
        bool neg = false;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int q = 0; q < RPR; ++q) neg |= src[b][q] < 0;
        if (__ballot(neg)) {
#pragma unroll
          for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < RPR; ++q) src[b][q] = entry_src(a, src[b][q]);
        }
      }
      grads_round(cur, src, r, acc);
    }
    if constexpr (!kSgd) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int64_t u = ubase + (int64_t)it * NB + b;
        // dense store: a row without occurrences is written as zeros; a hot row is written by its tasks
        const bool empty_row = o.kind == kOptStore && u < nu && cur.n[b] == 0;
        apply_sink<WDT, GDT, NCOL, true>(o, u < nu ? u : 0, reinterpret_cast<void*>(cur.rowp[b]), a.D, lpr_log2,
                                         a.round_grad != 0, acc[b], cur.cnt[b] > 0 || empty_row);
      }
      cur = nxt;
      continue;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (cur.rowp[b] == 0) continue;
#pragma unroll
      for (int k = 0; k < NCOL; ++k) {
        const int e = 4 * (c + k * LPR);
        if (e < a.D) {
          float r4[4];
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float gr = a.round_grad ? Elem<GDT>::rnd(acc[b][k][w]) : acc[b][k][w];
            r4[w] = wrow[b][k][w] - gr * o.lr;
          }
          st4<WDT>(reinterpret_cast<void*>(cur.rowp[b]), e, make_float4(r4[0], r4[1], r4[2], r4[3]));
        }
      }
    }
    cur = nxt;
  }
}

// kSgd: instantiation for SGD on vector rows without per-feature dims (its own register budget: 93 VGPRs)
STAMP_ARRAY(g_st_bwd, 16384, 2)
#if MI355_STAMPS
struct BwdEndStamp { __device__ ~BwdEndStamp() { STAMP(g_st_bwd, 16384, 2, 1); } };
#endif
template <int WDT, int GDT, int NCOL, bool kVec, bool kSgd, int KITT = kBwdGroupsPerLaneGroup>
__global__ void __launch_bounds__(256) bwd_kernel(BwdArgs a, OptArgs o, int lpr_log2) {
  STAMP(g_st_bwd, 16384, 2, 0);
#if MI355_STAMPS
  BwdEndStamp end_stamp__;
#endif
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2, NSUB = 64 >> lpr_log2;
  const int sub = lane >> lpr_log2, c = lane & (LPR - 1);
  const int wpb = blockDim.x >> 6;
  constexpr int W = kVec ? 4 : 1;
  if ((int)blockIdx.x < a.hot_blocks) {
    // ------------------------------------------------ hot tasks: one BLOCK per chunk of CSR entries
    // The chunk (1024 entries by default, hot.h) is split over the block's 4*NSUB lane groups, each slice summed by
    // reduce_chunk, folded inside the wave with shuffles and across the 4 waves through LDS.  A row that fits one
    // chunk is finished right here; longer rows add one partial per chunk into the row's fp32 accumulator
    // (agent-scope atomics: ~0.1-0.2 us each when thousands hit one address, hence block-sized chunks)
    // and the chunk that draws the last ticket applies the sink.
    extern __shared__ __attribute__((aligned(16))) float s_part[];  // [4 waves][D]
    const int wv = threadIdx.x >> 6;
    int ntasks = *a.hot.n_tasks;
    ntasks = ntasks < a.hot.max_tasks ? ntasks : a.hot.max_tasks;
    for (int task = blockIdx.x; task < ntasks; task += a.hot_blocks) {
      const int u = a.hot.task_u[task], h = a.hot.task_h[task];
      const int lo = a.hot.task_lo[task], hi = a.hot.task_hi[task];
      const int ngroups = wpb * NSUB;
      const int per = (hi - lo + ngroups - 1) / ngroups;
      int slo = lo + (wv * NSUB + sub) * per, shi = slo + per;
      shi = shi < hi ? shi : hi;
      float g[NCOL][4];
      reduce_chunk<GDT, NCOL, kVec, HOT_UNR>(a, slo, shi, per, lpr_log2, g);
      for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < NCOL; ++k)
#pragma unroll
          for (int w = 0; w < W; ++w) g[k][w] += __shfl_xor(g[k][w], off, 64);
      if (sub == 0) {
#pragma unroll
        for (int k = 0; k < NCOL; ++k)
#pragma unroll
          for (int w = 0; w < W; ++w) { const int e = W * (c + k * LPR) + w; if (e < a.D) s_part[wv * a.D + e] = g[k][w]; }
      }
      __syncthreads();
      if (wv == 0) {
        int Drow = a.D;
        if (a.D_offsets && a.combiner >= 0) { const int f = entry_src(a, a.csr_src[lo]) / a.B; Drow = a.D_offsets[f + 1] - a.D_offsets[f]; }
        float gg[NCOL][4];
#pragma unroll
        for (int k = 0; k < NCOL; ++k)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const int e = W * (c + k * LPR) + w;
            gg[k][w] = (w < W && e < a.D) ? s_part[e] + s_part[a.D + e] + s_part[2 * a.D + e] + s_part[3 * a.D + e] : 0.f;
          }
        const int nch = a.hot.hot_nchunks[h];
        int last = 1;
        if (nch > 1) {
          if (sub == 0) {
#pragma unroll
            for (int k = 0; k < NCOL; ++k)
#pragma unroll
              for (int w = 0; w < W; ++w) {
                const int e = W * (c + k * LPR) + w;
                if (e < Drow)
                  __hip_atomic_fetch_add(&a.hot.hot_acc[(int64_t)h * a.hot.dim + e], gg[k][w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              }
          }
          // The accumulator is only ever touched with agent-scope atomics (performed at the coherence
          // point), so no release/acquire fence is needed -- a release fence would write back the XCD's
          // whole dirty L2 (the table rows other waves are updating) once per task.  Drain this wave's
          // atomics (vmcnt counts them), then take a ticket; the last ticket owns the row.
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          last = 0;
          if (lane == 0) {
            const int t = __hip_atomic_fetch_add(&a.hot.hot_done[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (t == nch - 1);
          }
          last = __shfl(last, 0, 64);
          if (last) {
#pragma unroll
            for (int k = 0; k < NCOL; ++k)
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                const int e = W * (c + k * LPR) + w;
                gg[k][w] = (w < W && e < Drow)
                               ? __hip_atomic_load(&a.hot.hot_acc[(int64_t)h * a.hot.dim + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : 0.f;
              }
          }
        }
        if (last) {
          void* row = o.kind == kOptStore ? nullptr : reinterpret_cast<void*>(a.row_addr[u]);
          apply_sink<WDT, GDT, NCOL, kVec>(o, u, row, Drow, lpr_log2, a.round_grad != 0, gg, sub == 0);
        }
      }
      __syncthreads();
    }
    return;
  }
  // ---------------------------------------------------- one-wave rows (khot < occurrences <= kwave)
  // The wave's lane groups split the row's entries, reduce_chunk sums them with independent loads, an xor-shuffle
  // folds the groups and group 0 applies the sink: no LDS, no atomics, four rows per block at a time.
  if ((int)blockIdx.x < a.hot_blocks + a.wave_blocks) {
    int nw = *a.hot.n_wave;
    nw = nw < a.hot.max_hot ? nw : a.hot.max_hot;
    for (int t = ((int)blockIdx.x - a.hot_blocks) * wpb + (int)(threadIdx.x >> 6); t < nw; t += a.wave_blocks * wpb) {
      const int u = a.hot.wave_u[t], lo = a.hot.wave_lo[t], cnt = a.hot.wave_cnt[t];
      const int per = (cnt + NSUB - 1) / NSUB;
      const int slo = lo + sub * per;
      int shi = slo + per;
      shi = shi < lo + cnt ? shi : lo + cnt;
      float g[NCOL][4];
      reduce_chunk<GDT, NCOL, kVec, HOT_UNR>(a, slo, shi, per, lpr_log2, g);
      for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < NCOL; ++k)
#pragma unroll
          for (int w = 0; w < W; ++w) g[k][w] += __shfl_xor(g[k][w], off, 64);
      int Drow = a.D;
      if (a.D_offsets && a.combiner >= 0) { const int f = entry_src(a, a.csr_src[lo]) / a.B; Drow = a.D_offsets[f + 1] - a.D_offsets[f]; }
      void* row = o.kind == kOptStore ? nullptr : reinterpret_cast<void*>(a.row_addr[u]);
      apply_sink<WDT, GDT, NCOL, kVec>(o, u, row, Drow, lpr_log2, a.round_grad != 0, g, sub == 0);
    }
    return;
  }
  // ---------------------------------------------------- regular rows: NB unique rows per lane group
  // Most rows of a batch occur once or twice, so one row per lane group would leave a wave with ~1.5 KB
  // in flight per ~5 us dependent chain (ptr -> CSR entry -> gradient row).  Each lane group therefore
  // takes NB consecutive unique rows and walks their entries in lock step, RPR entries of each per round;
  // the table rows themselves are fetched up front (SGD) since their address only needs row_addr[u].
  constexpr int NB = NCOL == 1 ? PIPE_NB : (NCOL == 2 ? 2 : 1);
  constexpr int RPR = 2;
  constexpr int KIT = KITT;      // (the SGD walk of one-column-group rows is also compiled for 2 groups: see BwdArgs::kit)
  int64_t nu = a.max_unique;
  if (a.nu_dev) { int64_t m = *a.nu_dev; nu = m < nu ? m : nu; }
  const int64_t sg = ((int64_t)(blockIdx.x - a.hot_blocks - a.wave_blocks) * wpb + (threadIdx.x >> 6)) * NSUB + sub;
  if constexpr (kVec) {
    if constexpr (kSgd) {
      rows_pipelined<WDT, GDT, NCOL, NB, PIPE_RPR, KIT, true>(a, o, lpr_log2, nu, sg);
      return;
    } else if (a.D_offsets == nullptr) {
      rows_pipelined<WDT, GDT, NCOL, NB, PIPE_RPR, KIT, false>(a, o, lpr_log2, nu, sg);
      return;
    }
  }
  const int kit_g = a.kit < KIT ? a.kit : KIT;
  for (int it = 0; it < kit_g; ++it) {   // wave-uniform trip count (apply_sink shuffles inside)
    const int64_t u0 = (sg * kit_g + it) * NB;
    int lo[NB], hi[NB];
    bool work[NB], have[NB], hotrow[NB];
    uintptr_t rowp[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int64_t u = u0 + b;
      have[b] = u < nu;
      const int64_t uc = have[b] ? u : (nu > 0 ? nu - 1 : 0);
      const int l = a.ptr[uc], h = a.ptr[uc + 1];
      hotrow[b] = a.hot.n_tasks != nullptr && (h - l) > a.hot.khot;
      work[b] = have[b] && h > l && !hotrow[b];
      lo[b] = work[b] ? l : 0;
      hi[b] = work[b] ? h : 0;
      rowp[b] = (work[b] && o.kind != kOptStore) ? (uintptr_t)a.row_addr[uc] : 0;
    }
    float g[NB][NCOL][4];
    reduce_multi<GDT, NCOL, kVec, NB, RPR>(a, lo, hi, lpr_log2, g);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      int Drow = a.D;
      if (work[b] && a.D_offsets && a.combiner >= 0 && o.kind != kOptStore) {
        const int f = entry_src(a, a.csr_src[lo[b]]) / a.B;
        Drow = a.D_offsets[f + 1] - a.D_offsets[f];
      }
      // rows without occurrences: the reference's reduce_grads leaves them unwritten; store zeros
      const bool act = work[b] || (have[b] && !hotrow[b] && o.kind == kOptStore);
      apply_sink<WDT, GDT, NCOL, kVec>(o, have[b] ? u0 + b : 0, reinterpret_cast<void*>(rowp[b]), Drow, lpr_log2,
                                       a.round_grad != 0, g[b], act);
    }
  }
}

// optimizer on dense unique gradients ({sgd,adam,adagrad,rowwise_adagrad}_update_for_flat_table /
// ..._for_padded_buffer): one wave per row, same arithmetic as the fused path.
template <int WDT, int GDT, int NCOL, bool kVec>
__global__ void __launch_bounds__(256)
opt_rows_kernel(OptArgs o, const void* grads, int64_t grad_stride, int64_t n, const int64_t* __restrict__ n_dev,
                const int64_t* __restrict__ row_addr, void* dense_rows, int64_t dense_stride, int D, int lpr_log2,
                const int64_t* __restrict__ table_ids, const int64_t* __restrict__ table_emb_dims) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2;
  const int c = lane & (LPR - 1);
  constexpr int W = kVec ? 4 : 1;
  const int64_t wpb = blockDim.x >> 6;
  for (int64_t u = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); u < n; u += (int64_t)gridDim.x * wpb) {
    void* row = row_addr ? reinterpret_cast<void*>(row_addr[u])
                         : (void*)(reinterpret_cast<typename Elem<WDT>::T*>(dense_rows) + u * dense_stride);
    // rows of tables with different widths in one padded buffer (update4_padded_buffer_kernel, optimizer.cu:242-280: the row's
    // own width from table_emb_dims[table_ids[row]], its state behind the WIDEST embedding)
    const int Dr = table_ids ? (int)table_emb_dims[table_ids[u]] : D;
    float g[NCOL][4];
#pragma unroll
    for (int k = 0; k < NCOL; ++k) {
      const int e = W * (c + k * LPR);
      g[k][0] = g[k][1] = g[k][2] = g[k][3] = 0.f;
      if (e < Dr) {
        if (kVec) { float4 t = ld4<GDT>(grads, u * grad_stride + e); g[k][0] = t.x; g[k][1] = t.y; g[k][2] = t.z; g[k][3] = t.w; }
        else g[k][0] = ld1<GDT>(grads, u * grad_stride + e);
      }
    }
    apply_sink<WDT, GDT, NCOL, kVec>(o, u, row, Dr, lpr_log2, false, g, (lane >> lpr_log2) == 0);
  }
}

}  // namespace mi355

using namespace mi355;
STAMP_EXPORT(mi355_debug_stamps_bwd, g_st_bwd)

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static int lpr_log2_for(int D, bool vec) {
  int l = 3;
  const int per = vec ? 4 : 1;
  while ((per << l) < D && l < 6) ++l;
  return l;
}

template <int WDT, int GDT>
static int launch_bwd(BwdArgs a, OptArgs o, bool vec, hipStream_t stream) {
  const int l = lpr_log2_for(a.D, vec);
  const int per = vec ? 4 : 1;
  const int ncol = (a.D + (per << l) - 1) / (per << l);
  const int nsub = 64 >> l;
  constexpr int hot_cap = 2048;    // (swept in rounds 2-5, profiles/r05_bwd_hot_sweep.txt: the optimum at C2 and at the 16x batch)
  a.hot_blocks = a.hot.n_tasks ? (a.hot.max_tasks < hot_cap ? a.hot.max_tasks : hot_cap) : 0;
  const size_t smem = a.hot.n_tasks ? 4 * (size_t)a.D * sizeof(float) : 0;
  const int nb = ncol <= 1 ? PIPE_NB : (ncol <= 2 ? 2 : 1);
  a.wave_blocks = 0;
  constexpr int wave_cap = 1024;
  if (a.hot.n_tasks && a.hot.kwave > a.hot.khot) a.wave_blocks = a.hot.max_hot / 4 + 1 < wave_cap ? a.hot.max_hot / 4 + 1 : wave_cap;
  // (max_unique = the key count of the batch: C2 360 K, its 4x batch 1.44 M)
  // (round 6, stateful optimizers: the same rule -- C2 with Adam 0.212 -> 0.198 ms per step with the walk compiled for 2 groups, 1 the same,
  //  3 and 4 slower, one block per CU slower; profiles/r06_bwd_variants.txt)
  a.kit = (a.max_unique <= 720 * 1024 && vec && a.D_offsets == nullptr && ncol <= 1) ? BWD_KIT_SMALL : kBwdGroupsPerLaneGroup;
  const int grid = a.hot_blocks + a.wave_blocks + grid_for(a.max_unique, 4 * nsub * nb * a.kit, 1 << 20);
#define MI355_BWD_LAUNCH(NC, V) hipLaunchKernelGGL((bwd_kernel<WDT, GDT, NC, V, false>), dim3(grid), dim3(256), smem, stream, a, o, l)
#define MI355_BWD_LAUNCH_SGD(NC) hipLaunchKernelGGL((bwd_kernel<WDT, GDT, NC, true, true>), dim3(grid), dim3(256), smem, stream, a, o, l)
#define MI355_BWD_LAUNCH_SGD2() hipLaunchKernelGGL((bwd_kernel<WDT, GDT, 1, true, true, BWD_KIT_SMALL>), dim3(grid), dim3(256), smem, stream, a, o, l)
  if (vec && o.kind == kOptSgd && a.D_offsets == nullptr) {
    if (ncol <= 1 && a.kit == BWD_KIT_SMALL) MI355_BWD_LAUNCH_SGD2();
    else if (ncol <= 1) MI355_BWD_LAUNCH_SGD(1); else if (ncol <= 2) MI355_BWD_LAUNCH_SGD(2); else MI355_BWD_LAUNCH_SGD(4);
  } else if (vec) {
    if (ncol <= 1 && a.kit == BWD_KIT_SMALL) hipLaunchKernelGGL((bwd_kernel<WDT, GDT, 1, true, false, BWD_KIT_SMALL>), dim3(grid), dim3(256), smem, stream, a, o, l);
    else if (ncol <= 1) MI355_BWD_LAUNCH(1, true); else if (ncol <= 2) MI355_BWD_LAUNCH(2, true); else MI355_BWD_LAUNCH(4, true);
  } else {
    if (ncol <= 1) MI355_BWD_LAUNCH(1, false); else if (ncol <= 2) MI355_BWD_LAUNCH(2, false);
    else if (ncol <= 4) MI355_BWD_LAUNCH(4, false); else MI355_BWD_LAUNCH(16, false);
  }
#undef MI355_BWD_LAUNCH
#undef MI355_BWD_LAUNCH_SGD
#undef MI355_BWD_LAUNCH_SGD2
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

template <int WDT, int GDT>
static int launch_opt(OptArgs o, const void* grads, int64_t grad_stride, int64_t n, const int64_t* n_dev,
                      const int64_t* row_addr, void* dense_rows, int64_t dense_stride, int D, bool vec, hipStream_t stream,
                      const int64_t* table_ids = nullptr, const int64_t* table_emb_dims = nullptr) {
  const int l = lpr_log2_for(D, vec);
  const int per = vec ? 4 : 1;
  const int ncol = (D + (per << l) - 1) / (per << l);
  const int grid = grid_for(n, 4, 1 << 20);
#define MI355_OPT_LAUNCH(NC, V)                                                                                    \
  hipLaunchKernelGGL((opt_rows_kernel<WDT, GDT, NC, V>), dim3(grid), dim3(256), 0, stream, o, grads, grad_stride, n, \
                     n_dev, row_addr, dense_rows, dense_stride, D, l, table_ids, table_emb_dims)
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

int64_t mi355_backward_workspace_bytes(int64_t num_keys, int64_t dim) { return hot_bytes(num_keys, dim); }

// Fused backward over the CSR of mi355_group_by_unique.
//   opt_kind 0: store the reduced gradients to `out` [max_unique, out_stride]; `weight_dtype` is then the dtype
//               of `out` (reduce_grads uses the grad dtype; the sharded backward keeps fp32)
//   opt_kind 1..4: SGD / Adam / AdaGrad / row-wise AdaGrad applied in place on the table rows `row_addr`
int mi355_backward_fused(const int32_t* ptr, const int32_t* csr_src, int64_t num_keys, int64_t max_unique,
                         const int64_t* nu_dev, const void* grads, int64_t grad_stride, int grad_dtype,
                         const int64_t* offsets, const int32_t* D_offsets, int64_t batch_size, int64_t dim, int combiner,
                         const int64_t* row_addr, int weight_dtype, int opt_kind, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int64_t iter_num, int64_t state_offset, int round_grad,
                         void* out, int64_t out_stride, int aligned16, void* workspace, int64_t workspace_bytes,
                         hipStream_t stream) {
  return mi355i_backward_fused(ptr, csr_src, num_keys, max_unique, nu_dev, grads, grad_stride, grad_dtype, offsets, D_offsets,
                               batch_size, dim, combiner, row_addr, weight_dtype, opt_kind, lr, beta1, beta2, eps, weight_decay,
                               iter_num, state_offset, round_grad, out, out_stride, aligned16, workspace, workspace_bytes,
                               nullptr, 0, stream);
}

// the same with the tile lists CSR reference entries point into (internal.h; only the fused forward produces references)
int mi355i_backward_fused(const int32_t* ptr, const int32_t* csr_src, int64_t num_keys, int64_t max_unique,
                          const int64_t* nu_dev, const void* grads, int64_t grad_stride, int grad_dtype,
                          const int64_t* offsets, const int32_t* D_offsets, int64_t batch_size, int64_t dim, int combiner,
                          const int64_t* row_addr, int weight_dtype, int opt_kind, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int64_t iter_num, int64_t state_offset, int round_grad,
                          void* out, int64_t out_stride, int aligned16, void* workspace, int64_t workspace_bytes,
                          const int32_t* tile_bags, int one_feature, hipStream_t stream) {
  MI355_CHECK_ARG(opt_kind >= 0 && opt_kind <= 4, "bad optimizer kind");
  MI355_CHECK_ARG(opt_kind != kOptStore || out, "out required for opt_kind 0");
  MI355_CHECK_ARG(opt_kind == kOptStore || row_addr, "row_addr required for optimizer kinds");
  MI355_CHECK_ARG(dim > 0 && dim <= 1024, "embedding dim must be in (0, 1024]");
  MI355_CHECK_ARG(combiner < 0 || (offsets && batch_size > 0), "pooled mode needs offsets and batch_size");
  if (max_unique == 0) return MI355_OK;
  BwdArgs a{};
  a.ptr = ptr; a.csr_src = csr_src; a.grads = grads; a.grad_stride = grad_stride; a.offsets = offsets; a.D_offsets = D_offsets;
  a.B = (int)batch_size; a.D = (int)dim; a.combiner = combiner; a.row_addr = row_addr; a.max_unique = max_unique;
  a.nu_dev = nu_dev; a.round_grad = round_grad; a.n_entries = (int)num_keys; a.tile_bags = tile_bags;
  a.one_feature = combiner >= 0 && one_feature;
  if (workspace) {  // the hot-row task list filled by mi355_group_by_unique(..., hot_workspace = workspace, dim)
    MI355_CHECK_ARG(workspace_bytes >= hot_bytes(num_keys, dim), "workspace too small");
    a.hot = hot_carve(workspace, num_keys, dim);
  }
  OptArgs o{};
  o.kind = opt_kind; o.lr = lr; o.beta1 = beta1; o.beta2 = beta2; o.eps = eps; o.weight_decay = weight_decay;
  o.bias1 = (float)(1.0 - pow((double)beta1, (double)iter_num));
  o.bias2 = (float)(1.0 - pow((double)beta2, (double)iter_num));
  o.state_offset = (int)state_offset; o.out = out; o.out_stride = out_stride;
  const bool vec = aligned16 != 0;
  mi355i_prof_mark(1, 0, stream);
  const int rc = MI355_DISPATCH_DTYPE(weight_dtype, Wd, [&] {
    return MI355_DISPATCH_DTYPE(grad_dtype, Gd, [&] { return launch_bwd<Wd, Gd>(a, o, vec, stream); });
  });
  mi355i_prof_mark(1, 1, stream);
  return rc;
}

// optimizer step on dense unique gradients [n, grad_stride]; rows by address (flat table) or in a
// dense padded buffer [n, dense_stride] (update_for_padded_buffer, optimizer.cu:242-413).
int mi355_optimizer_update_tables(int opt_kind, const void* grads, int64_t grad_stride, int grad_dtype, int64_t n,
                                  const int64_t* n_dev, const int64_t* row_addr, void* dense_rows, int64_t dense_stride,
                                  int weight_dtype, int64_t dim, int64_t state_offset, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int64_t iter_num, int aligned16, const int64_t* table_ids,
                                  const int64_t* table_emb_dims, hipStream_t stream) {
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
      return launch_opt<Wd, Gd>(o, grads, grad_stride, n, n_dev, row_addr, dense_rows, dense_stride, (int)dim, vec, stream,
                                table_ids, table_emb_dims);
    });
  });
}

int mi355_optimizer_update(int opt_kind, const void* grads, int64_t grad_stride, int grad_dtype, int64_t n,
                           const int64_t* n_dev, const int64_t* row_addr, void* dense_rows, int64_t dense_stride,
                           int weight_dtype, int64_t dim, int64_t state_offset, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int64_t iter_num, int aligned16, hipStream_t stream) {
  return mi355_optimizer_update_tables(opt_kind, grads, grad_stride, grad_dtype, n, n_dev, row_addr, dense_rows, dense_stride,
                                       weight_dtype, dim, state_offset, lr, beta1, beta2, eps, weight_decay, iter_num, aligned16,
                                       nullptr, nullptr, stream);
}

}  // extern "C"
