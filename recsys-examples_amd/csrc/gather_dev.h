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
#include "table_dev.h"

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

// Late rows of the fused forward (fused_fwd.hip): a key whose bucket had no free slot gets its slot from the eviction kernel,
// after the probe kernel wrote the per-occurrence addresses.  Such an occurrence carries the address word 1 and finds its row
// through its (tile, key) record: occ_slot[j] = record, whose third word is the slot by then (>= S: no row this step).
struct LateRefs {
  const int32_t* occ_slot = nullptr;
  const uint4* rec = nullptr;
  const int64_t* table_ptrs = nullptr;        // rows of the (single) table: base address, elements per row, first bucket
  const int64_t* table_value_dims = nullptr;
  const int64_t* tbo = nullptr;
  int64_t C = 0;
  int elem_bytes = 0;
  int S = 0;
  int T = 1;                                  // tables (global slots are table-major: tbo[t] C <= slot < tbo[t + 1] C)
  const int32_t* ready = nullptr;             // partition blocks in the SAME launch (round 5): ready[p] != 0 once partition p's evictions
  int cap = 2048;                             // are in the records (records per partition: ref / cap = p)
  unsigned long long* notice = nullptr;       // the step's host-visible notice word (fused_fwd.hip): bit 33 = a wait on `ready` was abandoned
};
// kReady: the caller's launch holds the partition blocks itself (part3_lean.h) and the lane waits for its partition's flag.  A
// template parameter on purpose: compiled into the pooled gather -- where the flag is never set -- the rarely taken spin loop
// cost the kernel's hot path 3.5 us in round 5 (28.9 -> 32.4 us) and 9 us once the wait was bounded (round 6: 31 -> 40 us).  What is
// left of the rare path costs nothing (a stub in its place: same time) and must stay inlined (noinline: 31.6 -> 42 us;
// profiles/r06_gather_variants.txt).
template <bool kReady>
__device__ __forceinline__ uintptr_t late_row(const LateRefs& L, int64_t j) {
  const int ref = L.occ_slot[j];
  if (ref < 0) return 0;
  if constexpr (kReady) {   // (rare: a key whose bucket was full) the partition block that evicts for it runs in this very launch, ahead of us
    // The partition blocks have the lowest block ids of the launch and are dispatched first -- in practice, not by any guarantee
    // of the programming model (CU masks, a pre-empting queue, a serialising profiler): the wait is BOUNDED (~0.5 s).  Abandoned,
    // the occurrence gets a zero row and the step's notice word says so: the host raises when it settles the step.
    const int32_t* rp = L.ready + ref / L.cap;
    unsigned spins = 0;
    while (__hip_atomic_load(rp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1u << 22)) {
        if (L.notice) __hip_atomic_fetch_or(L.notice, 1ull << 33, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return 0;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  const int z = (int)L.rec[ref].z;
  if (z < 0 || z >= L.S) return 0;
  int t = 0;
  while (t + 1 < L.T && L.tbo[t + 1] * L.C <= (int64_t)z) ++t;   // (rare path: keys that went through an eviction)
  return (uintptr_t)(L.table_ptrs[t] + ((int64_t)z - L.tbo[t] * L.C) * L.table_value_dims[t] * L.elem_bytes);
}

// Software-pipelined form for rows of one column group (NCOL == 1, the C2 shape): an LPR-lane group owns KIT
// consecutive bags.  The bag offsets are fetched once (one per lane), a bag's keys one per lane (coalesced): every
// lane resolves ITS key to a row address, and the addresses are handed round the group with shuffles, so the rows
// of a bag are independent loads instead of RPR-wide rounds of a three-hop chain.  The index hops of bag i+1 (row
// addresses) and i+2 (reverse indices) are issued BEFORE the row loads of bag i -- one wait per bag with everything
// in flight.  Rows are added in key order (bit-identical to the sequential sum).
// kAddr: 0 dense source, 1 row addresses per UNIQUE key (through rev), 2 row addresses per OCCURRENCE (fused forward: the
// reverse-index hop does not exist), 3 = 2 with late rows: an address word of 1 stands for a key whose row was resolved
// by the partition kernel (bucket full -> eviction) and is looked up through its record (LateRefs)
template <int SDT, int DDT, int kAddr, int UNR, int KIT>
__device__ __forceinline__ void gather_pooled_pipe(const PoolArgs& a, const LateRefs& late, int lpr_log2, int64_t sg) {
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2;
  const int c = lane & (LPR - 1);
  const gptr_t zero = (gptr_t)(uintptr_t)g_zero_row;
  constexpr int EB = SDT == kF32 ? 4 : 2;
  const int64_t bag0 = sg * KIT;
  if (bag0 >= a.FB) return;  // lane groups are independent below (shuffles stay inside the group)
  // offsets of my KIT bags: lane i of the group holds offsets[bag0 + i], i <= KIT (KIT < 8 <= LPR)
  int64_t myoff;
  {
    int64_t b = bag0 + (c < KIT ? c : KIT);
    b = b < a.FB ? b : a.FB;
    myoff = a.offsets[b];
  }
  const int off_lo = (int)(myoff & 0xffffffff), off_hi = (int)(myoff >> 32);
  auto bag_off = [&](int i) -> int64_t {
    i = i <= KIT ? i : KIT;
    return (int64_t)(((uint64_t)(unsigned)__shfl(off_hi, i, LPR) << 32) | (uint64_t)(unsigned)__shfl(off_lo, i, LPR));
  };
  // stage 1: the reverse index of MY key of the sub-chunk [lo + r, ...) of a bag
  auto my_index = [&](int64_t lo, int64_t hi, int64_t r) -> int64_t {
    int64_t j = lo + r + c;
    j = j < hi ? j : hi - 1;
    j = j < 0 ? 0 : j;
    j = j < a.n ? j : a.n - 1;
    if constexpr (kAddr >= 2) return j; else return a.rev[j];
  };
  // stage 2: its row address (0: no such key in this sub-chunk / missing row)
  auto my_row = [&](int64_t u, int64_t lo, int64_t hi, int64_t r) -> uintptr_t {
    uintptr_t p;
    if constexpr (kAddr) p = (uintptr_t)a.row_addr[u];
    else p = (uintptr_t)a.src + (uintptr_t)(u * a.src_stride * EB);
    return lo + r + c < hi ? p : 0;
  };
  auto add_rows = [&](uintptr_t rp, int nq, int Df, float4& acc) {
    const int rlo = (int)(rp & 0xffffffffu), rhi = (int)(rp >> 32);
    for (int q0 = 0; q0 < LPR; q0 += UNR) {
      if (__ballot(q0 < nq) == 0) break;
      float4 v[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const uintptr_t base = (uintptr_t)(unsigned)__shfl(rlo, q0 + q, LPR) | ((uintptr_t)(unsigned)__shfl(rhi, q0 + q, LPR) << 32);
        const int e = 4 * c;
        const gptr_t p = (base != 0 && e < Df) ? (gptr_t)(base + (uintptr_t)(e * EB)) : zero;
        v[q] = ld4g<SDT>(p);
      }
#pragma unroll
      for (int q = 0; q < UNR; ++q) add4(acc, v[q]);
    }
  };
  int64_t lo_c = bag_off(0), hi_c = bag_off(1);          // current bag
  int64_t lo_n = hi_c, hi_n = bag_off(2);                // next bag
  int64_t u_n;
  uintptr_t rp_c;
  {
    const int64_t u0 = my_index(lo_c, hi_c, 0);
    u_n = my_index(lo_n, hi_n, 0);
    rp_c = my_row(u0, lo_c, hi_c, 0);
  }
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int64_t bag = bag0 + it;
    if (bag >= a.FB) break;
    const int64_t lo_nn = hi_n, hi_nn = bag_off(it + 3);
    const uintptr_t rp_n = my_row(u_n, lo_n, hi_n, 0);    // row addresses of bag it+1
    const int64_t u_nn = my_index(lo_nn, hi_nn, 0);        // reverse indices of bag it+2
    const int f = (int)(bag / a.B), bb = (int)(bag - (int64_t)f * a.B);
    int d0 = f * a.D, Df = a.D;
    if (a.D_offsets) { d0 = a.D_offsets[f]; Df = a.D_offsets[f + 1] - d0; }
    const int64_t L = hi_c - lo_c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (kAddr == 3) { if (__ballot(rp_c == 1)) { if (rp_c == 1) rp_c = late_row<false>(late, lo_c + c); } }
    add_rows(rp_c, (int)(L < LPR ? L : LPR), Df, acc);
    for (int64_t r = LPR; __ballot(r < L) != 0; r += LPR) {   // bags longer than a lane group: dependent hops
      const int64_t u = my_index(lo_c, hi_c, r);
      uintptr_t rp = my_row(u, lo_c, hi_c, r);
      if constexpr (kAddr == 3) { if (__ballot(rp == 1)) { if (rp == 1) rp = late_row<false>(late, lo_c + r + c); } }
      const int64_t left = L - r;
      add_rows(rp, (int)(left < 0 ? 0 : (left < LPR ? left : LPR)), Df, acc);
    }
    if (a.combiner == 1 && L > 0) {
      const float fl = (float)L;
      acc.x /= fl; acc.y /= fl; acc.z /= fl; acc.w /= fl;
    }
    if (4 * c < Df) st4<DDT>(a.dst, (int64_t)bb * a.total_D + d0 + 4 * c, acc);
    lo_c = lo_n; hi_c = hi_n; rp_c = rp_n;
    lo_n = lo_nn; hi_n = hi_nn; u_n = u_nn;
  }
}


// ---- eval / inference forward as ONE kernel (round 3): no dedup, no unique numbering, no per-occurrence address array -- the
// lanes probe the scored hash table themselves (reference: table_lookup_kernel, kernels.cuh:81-187, and the eval path of
// BatchedDynamicEmbeddingTablesV2, batched_dynamicemb_tables.py:1140-1218; unknown keys contribute zero rows,
// test_batched_dynamic_embedding_tables_v2.py:1517-1591), then pool the rows exactly as gather_pooled_pipe does.
// Single table, non-counting score policies (a found key's score is an idempotent store: every occurrence writes the same
// value), bucket capacity a power of two (the bucket comes out of a multiply-high, not a 64-bit division).
// Measured at C2 (rocprofv3): 41.5 us against 49-52 us for the probe + gather pair.  Forms tried on the way: KIT keys per lane
// probed up front (84 registers, five waves per SIMD: 44 us); two phases per block through the address array (58 registers
// but fifteen dependent hops per block: 46 us).
struct ProbeRefs {
  const uint64_t* keys = nullptr;
  Table t{};
  const int64_t* tbo = nullptr;               // [T + 1] bucket ranges of the tables
  const int64_t* table_ptrs = nullptr;        // [T] rows: base address
  const int64_t* table_value_dims = nullptr;  // [T] elements per row
  int elem_bytes = 0;
  int find_policy = 0;                        // kConst / kAssign / kGlobalTimer
  uint64_t score_value = 0, timer = 0;        // timer == 0: the device clock
  // several tables (round 4): table t owns the keys [offsets[feature_offsets[t] B], offsets[feature_offsets[t + 1] B])
  int T = 1;
  const int64_t* feature_offsets = nullptr;   // [T + 1]
};

// Per-table scalars of the multi-table eval kernels, staged in LDS once per block (one two-hop load, then a barrier): a key's
// table is a binary search of its position in `seg`, everything else an LDS read -- against three to four more dependent
// global hops per key if the lanes searched the offsets themselves.
constexpr int kEvalMaxT = 128;
struct EvalTabs {
  int64_t seg[kEvalMaxT + 1];     // first key of every table (seg[T] = n)
  int64_t tbo[kEvalMaxT + 1];
  uint64_t magic[kEvalMaxT];      // floor((2^64 - 1) / buckets)
  int64_t tptr[kEvalMaxT];
  int rowb[kEvalMaxT];            // bytes per row
};
// called by the whole block; ends with a barrier
__device__ __forceinline__ void eval_tabs_load(EvalTabs& L, const ProbeRefs& pr, const int64_t* offsets, int B, int64_t n) {
  for (int t = threadIdx.x; t <= pr.T; t += blockDim.x) {
    L.seg[t] = t == pr.T ? n : offsets[pr.feature_offsets[t] * B];
    L.tbo[t] = pr.tbo[t];
    if (t < pr.T) {
      const uint64_t nb = (uint64_t)(pr.tbo[t + 1] - pr.tbo[t]);
      L.magic[t] = nb ? ~0ull / nb : 0ull;
      L.tptr[t] = pr.table_ptrs[t];
      L.rowb[t] = (int)pr.table_value_dims[t] * pr.elem_bytes;
    }
  }
  __syncthreads();
}

// the scalars of ONE table a probe needs
struct EvalTab {
  int64_t bkt0, tp0, rowb;
  uint64_t nb, magic;
};
__device__ __forceinline__ EvalTab eval_tab_of(const EvalTabs& L, int T, int64_t j) {
  int lo = 0, hi = T;      // first t with seg[t + 1] > j
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (L.seg[mid + 1] <= j) lo = mid + 1; else hi = mid; }
  const int t = lo < T ? lo : T - 1;
  EvalTab e;
  e.bkt0 = L.tbo[t]; e.nb = (uint64_t)(L.tbo[t + 1] - e.bkt0); e.magic = L.magic[t]; e.tp0 = L.tptr[t]; e.rowb = L.rowb[t];
  return e;
}

// Row address of `key` in table `e` (0: unknown key / !have): digest vector, key word, row -- three dependent hops; a found key's
// score is refreshed (score.cuh:72-96; idempotent across the key's occurrences).
__device__ __forceinline__ uintptr_t eval_probe_key(const ProbeRefs& pr, const EvalTab& e, uint64_t key, bool have, int C, int cshift) {
  const int64_t hash = (int64_t)(fmix64(key) & 0x7FFFFFFFFFFFFFFFull);
  const uint64_t x = (uint64_t)hash >> cshift;
  uint64_t rr = x - __umul64hi(x, e.magic) * e.nb;
  if (rr >= e.nb) rr -= e.nb;
  if (rr >= e.nb) rr -= e.nb;
  const int b = (int)(e.bkt0 + (int64_t)(e.nb ? rr : 0ull));
  const int st = (int)((uint64_t)hash & (uint64_t)(C - 1)) & ~15;
  have = have && is_valid(key) && e.nb > 0;
  const uint4 dv = *reinterpret_cast<const uint4*>(pr.t.dig(b) + st);
  const uint32_t m0 = eq_mask16(dv, digest_of(hash));
  int slot = m0 ? st + __ffs(m0) - 1 : -1;
  const uint64_t kw = pr.t.keys(b)[slot >= 0 ? slot : 0];
  slot = (slot >= 0 && kw == key) ? slot : -2;
  if (__ballot(have && slot == -2)) {   // the first candidate was another key, the key sits beyond the first vector, or is unknown
    if (have && slot == -2) {
      slot = -1;
      const uint32_t d = digest_of(hash);
      const uint64_t* ks = pr.t.keys(b);
      const uint8_t* dg = pr.t.dig(b);
      for (int gi = 0; gi < (C >> 4) && slot < 0; ++gi) {
        int p0 = st + (gi << 4);
        if (p0 >= C) p0 -= C;
        uint32_t m = eq_mask16(*reinterpret_cast<const uint4*>(dg + p0), d);
        while (m) {
          const int bit = __ffs(m) - 1;
          m &= m - 1;
          if (ks[p0 + bit] == key) { slot = p0 + bit; break; }
        }
      }
    }
  }
  if (!have || slot < 0) return 0;
  uint64_t* sc = pr.t.scores(b) + (int64_t)slot * pr.t.ns;
  if (pr.find_policy == kGlobalTimer) *sc = pr.timer;
  else if (pr.find_policy == kAssign) *sc = pr.score_value;
  return (uintptr_t)(e.tp0 + (((int64_t)b - e.bkt0) * C + slot) * e.rowb);
}

// one-launch eval forward, lane-group form: the KIT bags of a lane group are ONE contiguous run of keys, so lane c probes the
// run's key c (one key per lane: ~12 registers of probe state instead of KIT times that) and the bag loop picks its rows'
// addresses out of the lanes with shuffles.  Keys beyond the first LPR of the run (a few per cent of the groups at C2) are
// probed when their bag is reached (dependent hops).
// kMT: several tables -- `tabs` (LDS, eval_tabs_load) holds their scalars, a key's table follows from its position.
template <int SDT, int DDT, int UNR, int KIT, bool kMT = false>
__device__ __forceinline__ void gather_pooled_eval(const PoolArgs& a, ProbeRefs pr, int lpr_log2, int64_t sg, const EvalTabs* tabs = nullptr) {
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2;
  const int c = lane & (LPR - 1);
  const gptr_t zero = (gptr_t)(uintptr_t)g_zero_row;
  constexpr int EB = SDT == kF32 ? 4 : 2;
  const int64_t bag0 = sg * KIT;
  if (bag0 >= a.FB) return;  // lane groups are independent below (shuffles stay inside the group)
  int64_t myoff;
  {
    int64_t b = bag0 + (c < KIT ? c : KIT);
    b = b < a.FB ? b : a.FB;
    myoff = a.offsets[b];
  }
  EvalTab one{};
  if constexpr (!kMT) {
    one.bkt0 = pr.tbo[0];
    one.nb = (uint64_t)(pr.tbo[1] - one.bkt0);
    one.tp0 = pr.table_ptrs[0]; one.rowb = pr.table_value_dims[0] * pr.elem_bytes;
    one.magic = one.nb ? ~0ull / one.nb : 0ull;
  }
  if (!pr.timer) pr.timer = device_clock();
  const int C = (int)pr.t.C;
  const int cshift = __builtin_ctz((unsigned)C);
  const int off_lo = (int)(myoff & 0xffffffff), off_hi = (int)(myoff >> 32);
  auto bag_off = [&](int i) -> int64_t {
    i = i <= KIT ? i : KIT;
    return (int64_t)(((uint64_t)(unsigned)__shfl(off_hi, i, LPR) << 32) | (uint64_t)(unsigned)__shfl(off_lo, i, LPR));
  };
  // row address of the key at position j (0: unknown key / j >= jend): three dependent hops
  auto probe = [&](int64_t j, int64_t jend) -> uintptr_t {
    const bool have = j < jend;
    int64_t jc = have ? j : jend - 1;
    jc = jc < 0 ? 0 : jc;
    jc = jc < a.n ? jc : a.n - 1;
    const uint64_t key = pr.keys[jc];
    if constexpr (kMT) return eval_probe_key(pr, eval_tab_of(*tabs, pr.T, jc), key, have, C, cshift);
    else return eval_probe_key(pr, one, key, have, C, cshift);
  };
  // rows whose addresses sit in lanes base .. base + nq - 1 of `rp`
  auto add_rows = [&](uintptr_t rp, int base, int nq, int Df, float4& acc) {
    const int rlo = (int)(rp & 0xffffffffu), rhi = (int)(rp >> 32);
    for (int q0 = 0; q0 < LPR; q0 += UNR) {
      if (__ballot(q0 < nq) == 0) break;
      float4 v[UNR];
#pragma unroll
      for (int q = 0; q < UNR; ++q) {
        const int src = base + q0 + q;
        uintptr_t ad = (uintptr_t)(unsigned)__shfl(rlo, src & (LPR - 1), LPR) | ((uintptr_t)(unsigned)__shfl(rhi, src & (LPR - 1), LPR) << 32);
        ad = q0 + q < nq ? ad : 0;
        const int e = 4 * c;
        const gptr_t p = (ad != 0 && e < Df) ? (gptr_t)(ad + (uintptr_t)(e * EB)) : zero;
        v[q] = ld4g<SDT>(p);
      }
#pragma unroll
      for (int q = 0; q < UNR; ++q) add4(acc, v[q]);
    }
  };
  const int64_t run_lo = bag_off(0);
  int64_t run_hi = bag_off(KIT);
  run_hi = run_hi < a.n ? run_hi : a.n;
  const uintptr_t rp0 = probe(run_lo + c, run_hi);      // the run's first LPR keys, one per lane
#pragma unroll
  for (int it = 0; it < KIT; ++it) {
    const int64_t bag = bag0 + it;
    if (bag >= a.FB) break;
    const int64_t lo_c = bag_off(it), hi_c = bag_off(it + 1);
    const int f = (int)(bag / a.B), bb = (int)(bag - (int64_t)f * a.B);
    int d0 = f * a.D, Df = a.D;
    if (a.D_offsets) { d0 = a.D_offsets[f]; Df = a.D_offsets[f + 1] - d0; }
    const int64_t L = hi_c - lo_c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // keys of this bag inside the first LPR of the run come out of rp0, the rest is probed now, LPR at a time
    int64_t r = 0;
    const int64_t in_first = run_lo + LPR - lo_c;      // keys of the bag covered by rp0 (may be <= 0 or >= L)
    if (__ballot(in_first > 0 && L > 0)) {
      const int64_t nq = in_first < L ? (in_first > 0 ? in_first : 0) : L;
      add_rows(rp0, (int)(lo_c - run_lo), (int)nq, Df, acc);
      r = nq;
    }
    for (; __ballot(r < L) != 0; r += LPR) {
      const uintptr_t rr = probe(lo_c + r + c, hi_c);
      const int64_t left = L - r;
      add_rows(rr, 0, (int)(left < 0 ? 0 : (left < LPR ? left : LPR)), Df, acc);
    }
    if (a.combiner == 1 && L > 0) {
      const float fl = (float)L;
      acc.x /= fl; acc.y /= fl; acc.z /= fl; acc.w /= fl;
    }
    if (4 * c < Df) st4<DDT>(a.dst, (int64_t)bb * a.total_D + d0 + 4 * c, acc);
  }
}


}  // namespace mi355
