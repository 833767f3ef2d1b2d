// Device side of the pooled gather shared by value_ops.hip (mi355_gather_pooled) and fused_fwd.hip (the fused forward's
// launches): argument block, row loads, and the FLAT-STREAM gather -- the round-3 form of the bandwidth kernel.
//
// Replaces (reference, corelib/dynamicemb/src/): multi_to_one_warp_per_ev_vec4_kernel / multi_to_one_cta_per_ev_kernel
// (lookup_kernel.cuh:859-998, descriptor lookup_forward.cu:30-104).
//
// Why a flat stream.  Measured on the C2 shape (tools/ubench_gather.hip, one MI355X): one lane group per bag, however its
// index hops are pipelined, spends most of a wave's life with no row bytes in flight -- a bag of 1..10 keys is 1..3 dependent
// rounds of 4 rows, and the next bag's rows wait for the current bag's adds (33 us for 361 K rows; the library's 3-hop
// software pipeline: 31 us).  Here a lane group owns KB CONSECUTIVE bags, i.e. one contiguous run of rows, and walks it in
// chunks of U rows whatever the bag boundaries are: the rows of chunk k+1 are issued BEFORE chunk k is accumulated (two
// register buffers), the address words of chunk k+2 before that, so U .. 2U rows per lane group are in flight for the whole
// life of the wave (27 us, same output bit for bit: rows are still added in key order, one rounding at the store).
#pragma once
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
  int64_t n;                   // number of keys (= offsets[FB])
  int B;
  int D;                       // uniform dim, or max_D when D_offsets != nullptr
  int total_D;
  int combiner;                // 0 sum, 1 mean
};

// 4 KiB of zeros every lane may read: padding rows, missing rows (address 0) and lanes beyond a row's width load from here,
// so EVERY load of a round is unconditional -- hipcc otherwise wraps each predicated load in its own exec-masked branch with
// an s_waitcnt vmcnt(0), which serialises the whole round into one long dependent chain (measured: 3x slower).
static __device__ __attribute__((aligned(16))) float g_zero_row[1024];

typedef const __attribute__((address_space(1))) char* gptr_t;   // explicit GLOBAL pointers: keeps the
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) u32x2_t* gptr2_t;  // row loads global_load, not flat_load
typedef const __attribute__((address_space(1))) f32x4_t* gptr4_t;

__device__ __forceinline__ void add4(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

template <int DT>
__device__ __forceinline__ float4 ld4g(gptr_t p) {
  if constexpr (DT == kF32) {
    const f32x4_t t = *reinterpret_cast<gptr4_t>(p);
    return make_float4(t.x, t.y, t.z, t.w);
  } else {
    const u32x2_t r = *reinterpret_cast<gptr2_t>(p);
    float4 o;
    if constexpr (DT == kBF16) {
      o.x = __uint_as_float(r.x << 16); o.y = __uint_as_float(r.x & 0xffff0000u);
      o.z = __uint_as_float(r.y << 16); o.w = __uint_as_float(r.y & 0xffff0000u);
    } else {
      o.x = f16_to_f32((uint16_t)(r.x & 0xffff)); o.y = f16_to_f32((uint16_t)(r.x >> 16));
      o.z = f16_to_f32((uint16_t)(r.y & 0xffff)); o.w = f16_to_f32((uint16_t)(r.y >> 16));
    }
    return o;
  }
}

// Flat-stream pooled gather of the bags [b0, b0 + bn) by ONE lane group of LPR = 1 << lpr_log2 lanes (rows of one column
// group: D <= 4 * LPR; uniform D: a.D_offsets == nullptr; bn <= KB < LPR).  Called by whole lane groups; groups of a wave are
// independent (shuffles stay inside the group).
// kAddr: 0 dense source through rev, 1 row addresses per UNIQUE key through rev (one more index hop, one more pipeline stage),
//        2 row addresses per OCCURRENCE (fused forward).
template <int SDT, int DDT, int kAddr, int U, int KB>
__device__ __forceinline__ void gather_pooled_flat(const PoolArgs& a, int lpr_log2, int64_t b0) {
  const int LPR = 1 << lpr_log2;
  const int c = lane_id() & (LPR - 1);
  const gptr_t zero = (gptr_t)(uintptr_t)g_zero_row;
  constexpr int EB = SDT == kF32 ? 4 : 2;
  constexpr bool kTwoHop = kAddr == 1;
  int bn = (int)(a.FB - b0 < (int64_t)KB ? a.FB - b0 : (int64_t)KB);
  // offsets of my bags: lane i of the group holds offsets[b0 + i], i <= bn (KB < LPR)
  const int64_t myoff = a.offsets[b0 + (c <= bn ? c : bn)];
  const int olo = (int)(myoff & 0xffffffff), ohi = (int)(myoff >> 32);
  auto off = [&](int i) -> int64_t {
    i = i <= bn ? i : bn;
    return (int64_t)(((uint64_t)(unsigned)__shfl(ohi, i, LPR) << 32) | (uint64_t)(unsigned)__shfl(olo, i, LPR));
  };
  const int64_t lo = off(0);
  int64_t hi = off(bn);
  hi = hi < a.n ? hi : a.n;
  // index word of MY row (lane c < U) of the chunk at r: the per-occurrence row address (kAddr 2) or the reverse index
  auto pre = [&](int64_t r) -> int64_t {
    int64_t j = r + c;
    j = j < hi ? j : hi - 1;
    j = j < 0 ? 0 : j;
    if constexpr (kAddr == 2) return a.row_addr[j]; else return a.rev[j];
  };
  // ... to the row address (0: no such row in this chunk / missing row)
  auto fin = [&](int64_t w, int64_t r) -> uintptr_t {
    uintptr_t p;
    if constexpr (kAddr == 2) p = (uintptr_t)w;
    else if constexpr (kAddr == 1) p = (uintptr_t)a.row_addr[w];
    else p = (uintptr_t)a.src + (uintptr_t)(w * a.src_stride * EB);
    return (r + c < hi && c < U) ? p : 0;
  };
  const bool col = 4 * c < a.D;
  auto issue = [&](uintptr_t ad, float4 (&v)[U]) {
    const int alo = (int)(ad & 0xffffffffu), ahi = (int)(ad >> 32);
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const uintptr_t base = (uintptr_t)(unsigned)__shfl(alo, q, LPR) | ((uintptr_t)(unsigned)__shfl(ahi, q, LPR) << 32);
      const gptr_t p = (base != 0 && col) ? (gptr_t)(base + (uintptr_t)(4 * c * EB)) : zero;
      v[q] = ld4g<SDT>(p);
    }
  };
  // consume cursor: current bag (relative), its first row and the row behind its last
  int b = 0;
  int64_t bbeg = lo, bend = off(1);
  // (feature, sample) of the current bag, advanced incrementally (one 32-bit division per lane group)
  int f = (int)((uint32_t)b0 / (uint32_t)a.B), bb = (int)((uint32_t)b0 - (uint32_t)f * (uint32_t)a.B);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto flush = [&]() {   // store the current bag's sum and move to the next bag
    const int64_t L = bend - bbeg;
    if (a.combiner == 1 && L > 0) { const float fl = (float)L; acc.x /= fl; acc.y /= fl; acc.z /= fl; acc.w /= fl; }
    if (col) st4<DDT>(a.dst, (int64_t)bb * a.total_D + (int64_t)f * a.D + 4 * c, acc);
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
    ++b;
    if (++bb == a.B) { bb = 0; ++f; }
    bbeg = bend;
    bend = off(b + 1);
  };
  auto consume = [&](int64_t r, const float4 (&v)[U]) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t j = r + q;
      if (j < hi) {
        add4(acc, v[q]);
        while (b < bn && j + 1 >= bend) flush();    // (>=: empty bags behind this row are flushed as zeros)
      }
    }
  };
  while (b < bn && bend <= lo) flush();              // leading empty bags
  if (lo >= hi) {                                    // no rows at all (every bag empty, or offsets beyond n)
    while (b < bn) flush();
    return;
  }
  float4 va[U], vb[U];
  uintptr_t a1;
  int64_t w2 = 0;
  if constexpr (kTwoHop) {
    const int64_t w0 = pre(lo), w1 = pre(lo + U);
    w2 = pre(lo + 2 * U);
    const uintptr_t a0 = fin(w0, lo);
    a1 = fin(w1, lo + U);
    issue(a0, va);
  } else {
    const uintptr_t a0 = fin(pre(lo), lo);
    a1 = fin(pre(lo + U), lo + U);
    issue(a0, va);
  }
  for (int64_t r = lo; r < hi; r += 2 * U) {
    uintptr_t a2, a3;
    int64_t w3 = 0, w4 = 0;
    if constexpr (kTwoHop) { w3 = pre(r + 3 * U); a2 = fin(w2, r + 2 * U); } else { a2 = fin(pre(r + 2 * U), r + 2 * U); }
    issue(a1, vb);
    consume(r, va);
    if (__ballot(r + U < hi) == 0) break;            // wave uniform: the other lane group of the wave may still have rows
    if constexpr (kTwoHop) { w4 = pre(r + 4 * U); a3 = fin(w3, r + 3 * U); } else { a3 = fin(pre(r + 3 * U), r + 3 * U); }
    issue(a2, va);
    consume(r + U, vb);
    a1 = a3; w2 = w4;
  }
  while (b < bn) flush();                             // trailing empty bags (and nothing else: every row has been consumed)
}

}  // namespace mi355
