// Fused index stage of the DynamicEmb forward for gfx950: ONE kernel de-duplicates a tile of the batch in LDS, probes the
// scored hash table with the tile's distinct keys, counts / ranks the occurrences of every table slot, inserts unseen keys
// into free slots and initialises their rows -- the work of segmented_unique + table_lookup + table_insert + unlock +
// init_rows + row_addresses of the unfused chain (five dedup launches, lookup, insert, unlock/init: 10 launches, ~120 us
// at C2) in one launch (~35 us).
//
// Restates (reference, corelib/dynamicemb/): segmented_unique_cuda (src/unique_op.cu:484-714), table_lookup_kernel /
// table_insert_kernel / table_unlock_kernel (src/table_operation/kernels.cuh:81-585), the first-touch initialisation and
// the hit-slot pinning of _prefetch_hbm_direct_path (dynamicemb/batched_dynamicemb_function.py:559-696).
//
// Design (MI355X-first, not a translation):
//  * In steady state every key of the batch already owns a table slot, and the slot IS the identity of the unique row.
//    So the batch is de-duplicated BY SLOT: a persistent int32 counter per slot (`occ`, all zero between steps) receives
//    one atomicAdd per (tile, distinct key) pair; the value it returns ranks the tile's occurrences inside the row's
//    list (the backward's CSR needs exactly that), and the occurrence that draws rank 0 is the row's representative.
//    No scratch hash set, no clear pass, no separate lookup over the uniques.
//  * The pooled gather takes the row address of every OCCURRENCE (written here), so it does not depend on the unique
//    numbering at all.
//  * Unseen keys are inserted in place when their bucket has a free slot: CAS Empty -> Locked, digest, score, publish
//    the key.  A prober that meets a Locked slot re-reads until the key is published, so two tiles inserting the same key
//    agree on one slot.  Full buckets are deferred to the head of the next kernel (skipped when the list is empty), which
//    evicts the minimum score under a per-bucket lock; slots with occ > 0 (hit or inserted by THIS batch) and pinned
//    slots are never candidates -- the reference's increment_counter / decrement_counter bracket comes for free.
//  * ONE more kernel (fused_mid_kernel) numbers the uniques, scans the occurrence counts into the backward's CSR row
//    pointers (decoupled look-back across its blocks) and registers the hot rows; csr_scatter_kernel then writes the
//    CSR entries and the reverse indices.  Chain of a training forward: probe, numbering, scatter, gather.
//  * Probing is one lane per key with 16-B digest vectors (the first vector resolves the probe at normal load factors):
//    64 independent probes per wave in flight instead of 8 with the 8-lane groups of the per-op kernels.
#include "common.h"
#include "hot.h"
#include "../../include/recsys_amd.h"
#include "internal.h"
#include "scan_dev.h"
#include "table_dev.h"
#include "init_dev.h"
#include "roctx.h"
#include "stamps.h"
#include "gather_dev.h"
#include <pthread.h>
#include <chrono>
#include <stdlib.h>

namespace mi355 {

STAMP_ARRAY(g_st_probe, 1024, 12)
STAMP_ARRAY(g_st_part, 1024, 10)
STAMP_ARRAY(g_st_evict, 1024, 10)
STAMP_ARRAY(g_st_fgather, 16384, 2)
#define PST(ph) STAMP(g_st_probe, 1024, 12, ph)
#define QST(ph) STAMP(g_st_part, 1024, 10, ph)
#define EST(ph) STAMP(g_st_evict, 1024, 10, ph)      // the first deferred key of a partition block (thread 0 leads its lane group)

constexpr int kFusedMaxT = 128;     // tables per fused launch (per-table metadata lives in LDS)
constexpr int kPartMax = 1024;      // partitions of the partitioned index stage (their counters live in the aux header)
constexpr int kPartCap = 2048;      // (tile, key) records per partition
constexpr int kPartSub = 4;         // sub-lists per partition (tile % kPartSub picks one): same-address atomics serialise at
                                    // ~40 ns each, 176 tiles on ONE counter per partition is a 7 us chain, on four 1.8 us
constexpr int kSubCap = kPartCap / kPartSub;
constexpr int kAuxHdr = 64 + 4096 * 4;     // (room for 4 096 partitions' sub-list counters: a layout constant of the aux buffer)
// constexpr int kAuxHdr_r4 = 64 + kPartMax * 4;   // ints in front of the occ array: [0] deferred keys, [1..] grid barrier, [5] sticky
                                    // error flag of the partitioned stage, [64 + 4 p + r] records of sub-list r of partition p (zero between steps)

struct FusedArgs {
  Table t;
  const int64_t* tbo;             // [T+1] bucket offsets
  int32_t* bucket_sizes;
  const int32_t* counter;         // pin counters per slot (nullable)
  int32_t* hdr;                   // aux header
  int32_t* occ;                   // [S+1] pairs {occurrences of the slot in this batch, unique id of the slot}: occ[2 s], occ[2 s + 1]
                                  // (one 8-byte line-local pair per slot; the last entry serves keys without a slot)
  int32_t* locks;                 // [num_buckets] eviction lock
  int64_t S;                      // total slots
  const int64_t* table_ptrs;
  const int64_t* table_value_dims;
  const int64_t* table_emb_dims;
  int elem_bytes, value_dtype;
  const uint64_t* keys;
  int64_t n;
  const int64_t* offsets;
  const int64_t* feature_offsets;
  int64_t num_bags;
  int64_t batch;                  // bags per feature
  int T;
  int find_policy, insert_policy, use_count;
  int dbg;                        // MI355_FUSED_DBG (profiling only): 1 skip the slot-counter atomic, 2 skip the probe
  uint64_t score_value, timer;
  InitArgs init;
  // per-step outputs
  int64_t* occ_addr;              // [n] row address of every occurrence (0: no row)
  int32_t* occ_slot;              // [n] global slot (S: none; <= -2: deferred entry -(e+2))
  int32_t* csr_rank;              // [n]
  int32_t* partial;               // [ceil(n/1024)] representatives per 1024 occurrences
  int64_t* seg_out;               // [T+1] table ranges
  uint64_t* d_key; int32_t* d_tid; int32_t* d_cnt; int32_t* d_slot; int32_t* d_base;   // deferred (bucket full) keys
  // partitioned index stage (single table, 64 K .. 1 M keys): records of the (tile, key) pairs, grouped by slot range
  int P;                          // partitions (0: not partitioned)
  int spp;                        // slots per partition (multiple of the bucket capacity)
  int32_t* pcount;                // [P * kPartSub] records per sub-list (aux header; zero between steps)
  uint4* rec;                     // [P * kPartCap] {key lo, key hi, slot code, occurrences of the key in the record's tile}: ONE
                                  // 16-byte store per record; slot code = global slot, S (no slot), or -(bucket + 2) (bucket
                                  // full: deferred)
  int2* rec_out;                  // [P * kPartCap] out: {unique id, rank base of the tile inside the row's list; < 0: ~base of a
                                  // record whose row address was resolved late}
  unsigned long long* tstat;      // [ceil(n/1024)] look-back words of the merged numbering kernel (zeroed by the probe)
  int* hot_counters;              // hot-list header of the backward (cleared by the probe; nullable)
  // round 3, path (c): the probe kernel also lists, per tile, the bag ids of the keys that occur more than once in the tile
  // (grouped by key), so that the partition kernel can write the backward's CSR itself -- no scatter kernel
  int32_t* tile_bags;             // [n] tile t owns [t * TILE, ...): bag ids of its multi-occurrence keys, key after key
  int32_t* occ_trank;             // [n] rank of every occurrence among its key's occurrences in the tile (lazy reverse indices)
  uint64_t magic0;                // floor((2^64 - 1) / buckets of table 0): the multiply-high modulus of the one-table paths
  int4* rec_out4;                 // [P * kPartCap] out of path (c): {unique id (~id: row resolved late), rank base, CSR position, 0}
  // round 4: several tables on path (c).  Partitions are aligned to table boundaries -- table t owns 1 + floor((P - T) n_t / n) of
  // them (n_t = its keys in this batch), each a range of the table's BUCKETS -- so the unique order stays table-major; one record
  // list per partition (a table's keys come from a few neighbouring tiles: no same-address chain to split, and a sub-list could
  // overflow on one tile's records alone)
  int mt;                         // 1: multi-table partitions
  int32_t* ptab;                  // [P] table of every partition (written by block 0 of the probe kernel)
  // opt-in re-run of an overflowed step on the per-slot-counter path inside the same call (MI355_FUSED_OVERFLOW_RERUN=1): the
  // partitioned probe reports an overflow as hdr[ovf_word] = ovf_val (the call's epoch instead of the sticky 1), and the kernels
  // of the re-run chain, queued behind the gather on every step, return at once unless *gate == gate_val
  const int* gate;                // nullable: run only if *gate == gate_val
  int gate_val;
  int ovf_word, ovf_val;          // where / what the partitioned probe writes when a record list overflows
  int* rerun_mark;                // nullable: cleared by block 0 of the partitioned probe, set by the re-run chain's last kernel
  // round 6 (the default): the partition kernel's first block tells the HOST whether a record list of this step overflowed -- one
  // 8-byte system-scope store {epoch, flooded} into pinned memory -- and the step's backward (or whoever reads its unique
  // numbering first) re-runs the index stage on the per-slot-counter path before it uses the CSR (mi355_demb_fused_step_flooded,
  // mi355_demb_forward_fused_rerun): no update is ever skipped and the steady state pays no launch for it
  unsigned long long* notice;     // nullable: pinned, host-coherent
  // round 6, the prefetch pipeline on path (c): the index stage of batch k + 1 runs under the backward of batch k, so an eviction
  // must not take a slot an in-flight step uses.  With recency scores (STEP, TIMESTAMP) every key of an in-flight step carries a
  // score >= the score of the oldest of them: victims must score BELOW `protect` (~0: no limit) -- the pin of the reference's
  // prefetch (increment_counter, batched_dynamicemb_function.py:559-696) without a counter atomic per key
  uint64_t protect;
  int tl;                         // probe_c_kernel (round 5): keys of a tile, a run-time value (<= the kernel's capacity, multiple of 64)
  int cap;                        // records per partition of `rec` (kPartCap)
  int32_t* part_ready;            // [P] (round 5, the partition blocks riding in the gather's launch): 1 once partition p's deferred keys
                                  // have their slots in the records (zeroed by the probe kernel; nullable)
};

__device__ __forceinline__ void store_digest(uint8_t* p, uint8_t d) {
  __hip_atomic_store(p, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// score of a key that is already stored (score.cuh:72-96); `cnt` = occurrences of the key in the caller's tile
__device__ __forceinline__ void score_found(const FusedArgs& a, uint64_t* sc, int cnt) {
  const uint64_t v = a.use_count ? (uint64_t)cnt : a.score_value;
  switch (a.find_policy) {
    // Assign / timer scores are idempotent and nobody reads a score before the next kernel: plain (write-back cached)
    // stores.  Device-scope (sc1) stores resolve at the memory side like atomics, ~10 G/s on this part, and were half of
    // the kernel when every (tile, key) pair issued one.
    case kConst: break;
    case kAccumulate: atomicAdd((unsigned long long*)sc, (unsigned long long)v); break;
    case kLruLfu: *sc = a.timer; atomicAdd((unsigned long long*)(sc + 1), (unsigned long long)v); break;
    case kGlobalTimer: *sc = a.timer; break;
    default: *sc = v; break;
  }
}
// score of a key placed in a fresh (or evicted and cleared) slot: Accumulate acts as Assign (key_value_table.py:881-925)
__device__ __forceinline__ void score_new(const FusedArgs& a, uint64_t* sc, int cnt) {
  const uint64_t v = a.use_count ? (uint64_t)cnt : a.score_value;
  switch (a.insert_policy) {
    case kConst: ast64(sc, 0); break;
    case kLruLfu: ast64(sc, a.timer); ast64(sc + 1, v); break;
    case kGlobalTimer: ast64(sc, a.timer); break;
    default: ast64(sc, v); break;
  }
}

// One lane probes one key (types.cuh:308-396 order: 16-aligned start, wrap around; inside a 16-slot vector first the
// slots whose digest matches, then the first Empty one).  kInsert: an absent key takes the first Empty slot.
// Returns the slot, -1 (absent, lookup only) or -2 (no Empty slot in the bucket / gave up on a stuck lock).
template <bool kInsert>
__device__ __forceinline__ int thread_probe(const FusedArgs& a, int64_t b, uint64_t key, int64_t hash, int cnt, bool& inserted) {
  const Table& t = a.t;
  const int C = (int)t.C;
  const uint32_t d = digest_of(hash);
  const uint32_t ed = digest_of((int64_t)(fmix64(kEmptyKey) & 0x7FFFFFFFFFFFFFFFull));
  const int start = (int)(((C & (C - 1)) == 0 ? ((uint64_t)hash & (uint64_t)(C - 1)) : ((uint64_t)hash % (uint64_t)C))) & ~15;
  uint64_t* ks = t.keys(b);
  uint8_t* dg = t.dig(b);
  inserted = false;
  const int ngroups = C >> 4;
  int gi = 0, guard = 0;
  bool fresh = false;
  while (gi < ngroups) {
    int p0 = start + (gi << 4);
    if (p0 >= C) p0 -= C;
    const uint4 dv = load_dig16(dg + p0, fresh);
    bool again = false;
    uint32_t m = eq_mask16(dv, d);
    while (m) {
      const int bit = __ffs(m) - 1;
      m &= m - 1;
      const uint64_t k = fresh ? ald64(ks + p0 + bit) : ks[p0 + bit];
      if (k == key) return p0 + bit;
      // Locked: being inserted right now -- possibly this very key.  Empty behind a non-empty digest: the digest is
      // current but the key word comes from a stale cache line (the two live in different lines) -- look again with
      // device-scope loads, or the key would be inserted a second time further down the probe order.
      if (kInsert && (k == kLockedKey || (k == kEmptyKey && d != ed))) again = true;
    }
    if (!again) {
      uint32_t me = eq_mask16(dv, ed);
      while (me) {
        const int bit = __ffs(me) - 1;
        me &= me - 1;
        uint64_t k;
        if constexpr (kInsert) {
          // (round 5) the compare-and-swap IS the look at the slot: it takes an Empty word, and says what is there otherwise -- one
          // dependent round trip instead of load-then-CAS on the path every new key of a step goes down
          k = kEmptyKey;
          if (cas64(ks + p0 + bit, k, kLockedKey)) {
            store_digest(dg + p0 + bit, (uint8_t)d);
            score_new(a, t.scores(b) + (int64_t)(p0 + bit) * t.ns, cnt);
            atomicAdd(&a.bucket_sizes[b], 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ast64(ks + p0 + bit, key);     // publish: probers waiting on the Locked word now see the key
            inserted = true;
            return p0 + bit;
          }
        } else {
          k = fresh ? ald64(ks + p0 + bit) : ks[p0 + bit];
          if (k == kEmptyKey) return -1;
        }
        if (k == key) return p0 + bit;     // published after the digest snapshot was taken
        if (kInsert && k == kLockedKey) { again = true; break; }
        // a tombstone or another key behind a stale digest: keep scanning
      }
    }
    if (again) {
      fresh = true;
      if (++guard > (1 << 16)) return -2;  // a slot stuck in Locked (foreign writer): leave it to the deferred pass
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    ++gi;
  }
  return kInsert ? -2 : -1;
}

// whole wave initialises one row (initializer.cu:64-83 semantics, counter-based generator of init_dev.h)
__device__ __forceinline__ void wave_init_row(const FusedArgs& a, void* rp, uint64_t key, int ed, int vd) {
  const int lane = lane_id();
  for (int e = lane; e < vd; e += 64) {
    const float v = e < ed ? init_value(a.init, key, (uint32_t)e) : a.init.state_init;
    if (a.value_dtype == kF32) st1<kF32>(rp, e, v);
    else if (a.value_dtype == kBF16) st1<kBF16>(rp, e, v);
    else st1<kF16>(rp, e, v);
  }
}

// kFast (opt-in, MI355_FUSED_FASTMOD=1; bucket capacity a power of two): the key's bucket without 64-bit divisions.  gfx950
// has no 64-bit divide -- `(hash % (buckets * C)) / C` expands to ~400 instructions per key ahead of the digest load.
// floor((h mod n C) / C) = floor(h / C) mod n; h / C is a shift, and x mod n = x - mulhi64(x, M) n with M = floor((2^64 - 1) / n)
// leaves a quotient that is at most one short: two conditional subtractions make it exact.  M is formed once per table and
// block (s_magic).  The partition of a slot, slot / spp with spp a multiple of C, is bucket / (spp / C): 32-bit.
// kBags (path (c) of round 3, with kPart): the tile also resolves the BAG of every occurrence (64-ary search of the tile's ends in
// the offsets, bag starts marked in LDS, max-scan) and groups the bag ids of the keys that occur more than once in the tile --
// key after key, an exclusive scan over the representatives' counts gives the starts -- into tile_bags; a record carries
// {position of the representative, bag id | start of the key's list, slot code, occurrences}.
// kMT (round 4, with kPart + kBags): table-aligned partitions of a multi-table batch (FusedArgs::mt).
// kSeq (round 4, with kBags): sequence lookups -- the "bag" of occurrence j is j itself (the backward's CSR lists gradient rows),
// so the bag search and the running maximum over the bag marks fall away.
template <int TILE, int THREADS, bool kTrain, bool kPart = false, bool kFast = false, bool kBags = false, bool kMT = false, bool kSeq = false>
__global__ void __launch_bounds__(THREADS) fused_probe_kernel(FusedArgs a) {
  constexpr bool kOne = kPart && !kMT;             // the one-table forms of the partitioned paths keep the table's scalars in registers
  __shared__ int s_pt[kMT ? kFusedMaxT : 1];       // kMT: partitions of every table
  __shared__ int s_pb[kMT ? kFusedMaxT + 1 : 1];   //      first partition of every table
  __shared__ uint64_t s_psc[kMT ? kFusedMaxT : 1]; //      floor(2^32 partitions / buckets) of every table
  __shared__ int s_bag[kBags ? TILE : 1];          // bag of every occurrence of the tile
  __shared__ uint16_t s_sm[kBags ? 2 * TILE : 1];  // per dedup entry: start of the key's list in the tile | multi flag << 15
  __shared__ int s_brange[2], s_wmax[THREADS / 64], s_wsum[THREADS / 64];
  __shared__ int s_hist[kPart ? kPartMax : 1];    // kPart: records of this tile per partition, then their base in the partition
  __shared__ uint64_t s_magic[kFast ? kFusedMaxT : 1];
  if (a.gate && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.gate_val) return;   // (block uniform)
  PST(0);
  if (!a.timer) a.timer = device_clock();
  if (kPart && a.rerun_mark && blockIdx.x == 0 && threadIdx.x == 0) *a.rerun_mark = 0;
  constexpr int PER = TILE / THREADS;
  constexpr int LDS = 2 * TILE;
  constexpr int HALVES = TILE / 1024;
  __shared__ uint64_t s_key[TILE];
  __shared__ int s_tab[LDS];        // dedup: tile position of the key's representative; after the probe: its global slot
  __shared__ int s_cnt[LDS];        // dedup: occurrences inside the tile; after the probe: rank base of the tile
  __shared__ uint16_t s_t[TILE];    // table of every key (bit 15: representative whose row is new)
  __shared__ int s_gs[TILE];        // global slot of new rows (by representative position)
  __shared__ int64_t s_seg[kFusedMaxT + 1];
  __shared__ int64_t s_tbo[kFusedMaxT + 1];
  __shared__ int64_t s_tptr[kFusedMaxT];
  __shared__ int s_rowb[kFusedMaxT];
  __shared__ int s_nrep[HALVES];
  const int T = a.T;
  const int64_t tile0 = (int64_t)blockIdx.x * TILE;
  // The kernel is a chain of dependent memory round trips (keys -> digest vector -> key word -> slot counter), every block
  // of the grid is resident at once, so its duration is the latency of ONE chain: the key loads go out first, the
  // per-table metadata next (one hop, or none for a single table), and the first digest vector of EVERY key is fetched
  // before the LDS dedup -- speculative for the duplicates, but the dedup then runs in the shadow of that load.
  uint64_t kreg[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int64_t i = tile0 + q * THREADS + threadIdx.x;
    kreg[q] = a.keys[i < a.n ? i : a.n - 1];
  }
  // kPart (one table): the table's scalars come straight from memory with uniform loads issued here and used two phases later --
  // staged through LDS by threads 0 / 1 they were a global round trip in front of the FIRST barrier of every block
  int64_t m_tbo0 = 0, m_tbo1 = 0, m_tptr0 = 0;
  int m_rowb0 = 0;
  if constexpr (kOne) { m_tbo0 = a.tbo[0]; m_tbo1 = a.tbo[1]; m_tptr0 = a.table_ptrs[0]; m_rowb0 = (int)a.table_value_dims[0] * a.elem_bytes; }
  auto tbo_of = [&](int t) -> int64_t { if constexpr (kOne) return t == 0 ? m_tbo0 : m_tbo1; else return s_tbo[t]; };
  auto tptr_of = [&](int t) -> int64_t { if constexpr (kOne) return m_tptr0; else return s_tptr[t]; };
  auto rowb_of = [&](int t) -> int { if constexpr (kOne) return m_rowb0; else return s_rowb[t]; };
  for (int t = threadIdx.x; t <= T && !kOne; t += THREADS) {
    s_seg[t] = T == 1 ? (t == 0 ? 0 : a.n) : a.offsets[a.feature_offsets[t] * a.batch];
    s_tbo[t] = a.tbo[t];
    if (t < T) { s_tptr[t] = a.table_ptrs[t]; s_rowb[t] = (int)a.table_value_dims[t] * a.elem_bytes; }
    if constexpr (kFast) {
      if (t < T) { const uint64_t nb = (uint64_t)(a.tbo[t + 1] - a.tbo[t]); s_magic[t] = nb ? ~0ull / nb : 0ull; }
    }
  }
  const int cshift = kFast ? __builtin_ctzll((unsigned long long)a.t.C) : 0;
  if (blockIdx.x == 0 && kTrain) {
    if (a.hot_counters && threadIdx.x < 3) a.hot_counters[2 * threadIdx.x] = 0;   // n_hot, n_tasks, n_wave
  }
  if constexpr (kBags && !kSeq) {
    for (int k = threadIdx.x; k < TILE; k += THREADS) s_bag[k] = -1;
    if (threadIdx.x < 128) {   // waves 0 / 1: bags of the tile's first / last occurrence (first idx with offsets[idx] > key, minus 1)
      // 64-ary rounds, one probe per lane: three dependent round trips for 64 K bags.  Measured against wider rounds -- 4 probes
      // per lane (two rounds) took this phase from 8.6 K to 10.9 K cycles, 16 per lane (1024 scattered lines per wave and
      // round) the whole kernel from 28 to 38 us: the scattered lines cost more than the round they save.
      const int64_t tile_end = tile0 + TILE < a.n ? tile0 + TILE : a.n;
      const int64_t key = threadIdx.x < 64 ? tile0 : tile_end - 1;
      int lo = 0, hi = (int)a.num_bags;
      while (hi > lo) {
        const int step = (hi - lo + 63) >> 6;
        const int64_t pi = (int64_t)lo + (int64_t)(lane_id() + 1) * step - 1;
        const bool gt = pi >= hi ? true : a.offsets[pi] > key;
        const uint64_t gm = __ballot(gt);
        if (!gm) { lo = hi; break; }
        const int first = __ffsll((unsigned long long)gm) - 1;
        const int nlo = first ? lo + first * step : lo;
        const int64_t nhi = (int64_t)lo + (int64_t)(first + 1) * step - 1;
        lo = nlo;
        hi = nhi < hi ? (int)nhi : hi;
      }
      if (lane_id() == 0) s_brange[threadIdx.x < 64 ? 0 : 1] = lo - 1;
    }
  }
  if (kTrain && a.tstat && threadIdx.x < HALVES) {
    const int64_t pt = (int64_t)blockIdx.x * HALVES + threadIdx.x;
    if (pt * 1024 < a.n) { a.tstat[pt] = 0ull; if (kPart) a.tstat[a.P + pt] = 0ull; }
  }
  if constexpr (kPart && kTrain) {   // more partitions than 1024-key tile halves (small batches with thin partitions): the rest of the 2 P words
    const int have = (int)((a.n + 1023) >> 10);
    if (a.P > have && a.tstat) {
      const int per = (2 * a.P + (int)gridDim.x - 1) / (int)gridDim.x;
      for (int k = threadIdx.x; k < per; k += THREADS) {
        const int wd = (int)blockIdx.x * per + k;
        if (wd < 2 * a.P) a.tstat[wd] = 0ull;
      }
    }
  }
  for (int s = threadIdx.x; s < LDS; s += THREADS) { s_tab[s] = -1; s_cnt[s] = 0; }
  if (threadIdx.x < HALVES) s_nrep[threadIdx.x] = 0;
  if constexpr (kPart) for (int p = threadIdx.x; p < a.P; p += THREADS) s_hist[p] = 0;
  __syncthreads();
  PST(1);
  // kBags: the offsets of "my" bag of the tile's range are fetched now and used behind the next barrier
  int mb = 0;
  int64_t mo0 = 0, mo1 = 0;
  if constexpr (kBags && !kSeq) {
    mb = s_brange[0] + (int)threadIdx.x;
    const int bc = mb <= s_brange[1] ? mb : s_brange[1];
    mo0 = a.offsets[bc < 0 ? 0 : bc];
    mo1 = a.offsets[(bc < 0 ? 0 : bc) + 1];
  }
  if constexpr (kMT) {   // partitions per table: one each, the rest in proportion to the tables' keys in this batch
    if ((int)threadIdx.x < T) {
      const uint64_t nt_ = (uint64_t)(s_seg[threadIdx.x + 1] - s_seg[threadIdx.x]);
      s_pt[threadIdx.x] = 1 + (a.n > 0 ? (int)((uint64_t)(a.P - T) * nt_ / (uint64_t)a.n) : 0);
    }
  }
  int hh[PER], rk[PER];
  int64_t bq[PER], hq[PER];      // bucket and hash of my keys (bucket -1: key without a home)
  uint4 dvq[PER];                // first digest vector of the probe
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int li = q * THREADS + threadIdx.x;
    const int64_t i = tile0 + li;
    uint16_t tt = 0;
    if (T > 1) {
      int lo = 0, hi = T + 1;  // first t with seg[t] > i
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_seg[mid] <= i) lo = mid + 1; else hi = mid; }
      tt = (uint16_t)(lo - 1 < 0 ? 0 : (lo - 1 >= T ? T - 1 : lo - 1));
    }
    s_key[li] = kreg[q];
    s_t[li] = tt;
    const uint64_t key = kreg[q];
    const int64_t hash = (int64_t)(fmix64(key) & 0x7FFFFFFFFFFFFFFFull);
    const int64_t bb = tbo_of(tt);
    bool ok;
    int64_t b;
    if constexpr (kFast) {
      const uint64_t nb = (uint64_t)(tbo_of(tt + 1) - bb);
      const uint64_t x = (uint64_t)hash >> cshift;
      uint64_t r = x - __umul64hi(x, kOne ? a.magic0 : s_magic[tt]) * nb;
      if (r >= nb) r -= nb;
      if (r >= nb) r -= nb;
      ok = i < a.n && is_valid(key) && nb > 0;
      b = bb + (int64_t)(nb ? r : 0ull);
    } else {
      const int64_t cap = (tbo_of(tt + 1) - bb) * a.t.C;
      ok = i < a.n && is_valid(key) && cap > 0;
      const uint64_t local = (uint64_t)hash % (uint64_t)(cap > 0 ? cap : 1);
      b = bb + (int64_t)(local / (uint64_t)a.t.C);
    }
    hq[q] = hash;
    bq[q] = ok ? b : -1;
    const int C = (int)a.t.C;
    const int start = (int)(((C & (C - 1)) == 0 ? ((uint64_t)hash & (uint64_t)(C - 1)) : ((uint64_t)hash % (uint64_t)C))) & ~15;
    dvq[q] = *reinterpret_cast<const uint4*>(a.t.dig(ok ? b : 0) + start);   // unconditional: bucket 0 for homeless keys
  }
  PST(2);
  __syncthreads();
  PST(3);
  if constexpr (kMT) {
    if (threadIdx.x < 64) {      // (T <= 128: two tables per lane of wave 0)
      const int l = threadIdx.x;
      const int v0 = l < T ? s_pt[l] : 0, v1 = l + 64 < T ? s_pt[l + 64] : 0;
      const int i0 = wave_incl_scan(v0);
      const int i1 = wave_incl_scan(v1) + __shfl(i0, 63, 64);
      if (l < T) {
        s_pb[l] = i0 - v0;
        const uint64_t nb = (uint64_t)(s_tbo[l + 1] - s_tbo[l]);
        s_psc[l] = nb ? ((uint64_t)v0 << 32) / nb : 0ull;
      }
      if (l + 64 < T) {
        s_pb[l + 64] = i1 - v1;
        const uint64_t nb = (uint64_t)(s_tbo[l + 65] - s_tbo[l + 64]);
        s_psc[l + 64] = nb ? ((uint64_t)v1 << 32) / nb : 0ull;
      }
      if (l == 63) s_pb[T] = i1;
    }
  }
  if constexpr (kBags && !kSeq) {
    const int bhi = s_brange[1];
    if (mb >= 0 && mb <= bhi && mo1 > mo0) { const int64_t pp = mo0 > tile0 ? mo0 - tile0 : 0; if (pp < TILE) s_bag[pp] = mb; }
    for (int b = mb + THREADS; b <= bhi; b += THREADS) {      // (more bags than threads in the tile's range: empty / one-key bags)
      const int64_t o0 = a.offsets[b], o1 = a.offsets[b + 1];
      if (o1 > o0) { const int64_t pp = o0 > tile0 ? o0 - tile0 : 0; if (pp < TILE) s_bag[pp] = b; }
    }
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int li = q * THREADS + threadIdx.x;
    hh[q] = -1;
    rk[q] = 0;
    if (tile0 + li < a.n) {
      const uint64_t key = kreg[q];
      const int t = s_t[li];
      int h = (int)(fmix64(key + 0x9E3779B97F4A7C15ull * (uint64_t)t) >> 40) & (LDS - 1);
      while (true) {
        const int cur = atomicCAS(&s_tab[h], -1, li);
        if (cur == -1) break;
        if (s_key[cur] == key && s_t[cur] == t) { atomicMin(&s_tab[h], li); break; }
        h = (h + 1) & (LDS - 1);
      }
      hh[q] = h;
      rk[q] = atomicAdd(&s_cnt[h], 1);
    }
  }
  PST(4);
  __syncthreads();
  PST(5);
  bool isrep[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) isrep[q] = hh[q] >= 0 && s_tab[hh[q]] == q * THREADS + (int)threadIdx.x;
  // kBags, first half of two block scans (their second halves sit behind the next barrier): running maximum of the bag marks
  // (thread t owns the occurrences t * PER ..), and the list starts of the keys that occur more than once
  int bagv[PER], bag_incl = -1, mcnt[PER], m_incl = 0, m_mine = 0;
  if constexpr (kBags) {
    if constexpr (!kSeq) {
      int m = -1;
#pragma unroll
      for (int k = 0; k < PER; ++k) { const int v = s_bag[threadIdx.x * PER + k]; m = v > m ? v : m; bagv[k] = m; }
      bag_incl = m;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(bag_incl, off, 64); if (lane_id() >= off) bag_incl = o > bag_incl ? o : bag_incl; }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) { const int c = isrep[q] ? s_cnt[hh[q]] : 0; mcnt[q] = c > 1 ? c : 0; m_mine += mcnt[q]; }
    m_incl = m_mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(m_incl, off, 64); if (lane_id() >= off) m_incl += o; }
    if (lane_id() == 63) { s_wmax[threadIdx.x >> 6] = bag_incl; s_wsum[threadIdx.x >> 6] = m_incl; }
  }
  // kPart: the partition of a pair follows from its BUCKET (keys without a home: the last partition), which is known
  // before the probe -- so the tile's histogram and the one global atomic per (tile, partition) go out here and their
  // round trip runs under the probe; the bases are read when the records are written.
  int lpq[PER], my_base = 0;
  if constexpr (kPart) {
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      lpq[q] = 0;
      if (isrep[q]) {
        int pk;
        if constexpr (kMT) {   // the table's partitions split its bucket range evenly; keys without a home: its last partition
          const int t = s_t[q * THREADS + threadIdx.x];
          const int p0 = s_pb[t], p1 = s_pb[t + 1];
          const int x = bq[q] < 0 ? p1 - p0 - 1 : (int)(((uint64_t)(bq[q] - s_tbo[t]) * s_psc[t]) >> 32);
          pk = p0 + (x < p1 - p0 ? x : p1 - p0 - 1);
        } else
        pk = bq[q] < 0 ? a.P - 1
                       : kFast ? (int)((uint32_t)bq[q] / (uint32_t)(a.spp >> cshift)) : (int)(bq[q] * a.t.C / a.spp);
        lpq[q] = pk * 4096 + atomicAdd(&s_hist[pk], 1);
      }
    }
    __syncthreads();
    if ((int)threadIdx.x < a.P) {     // (P <= kPartMax = THREADS)
      const int c = s_hist[threadIdx.x];
      if (c) my_base = atomicAdd(&a.pcount[threadIdx.x * kPartSub + (kMT ? 0 : (int)blockIdx.x % kPartSub)], c);
      if constexpr (kMT) {
        if (blockIdx.x == 0) {   // the partition kernel learns its table from here
          int t = 0;
          while (t + 1 < T && s_pb[t + 1] <= (int)threadIdx.x) ++t;
          a.ptab[threadIdx.x] = t;
        }
      }
    }
    if constexpr (kBags) {
      const int w = threadIdx.x >> 6;
      int bbase = -1, sbase = 0;
      for (int k = 0; k < w; ++k) { bbase = s_wmax[k] > bbase ? s_wmax[k] : bbase; sbase += s_wsum[k]; }
      if constexpr (kSeq) {
#pragma unroll
        for (int k = 0; k < PER; ++k) s_bag[threadIdx.x * PER + k] = (int)tile0 + (int)threadIdx.x * PER + k;
      } else {
        int prev = __shfl_up(bag_incl, 1, 64);
        if (lane_id() == 0) prev = -1;
        prev = prev > bbase ? prev : bbase;
#pragma unroll
        for (int k = 0; k < PER; ++k) s_bag[threadIdx.x * PER + k] = bagv[k] > prev ? bagv[k] : prev;
      }
      int st = sbase + m_incl - m_mine;
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        if (isrep[q]) s_sm[hh[q]] = (uint16_t)(mcnt[q] ? (st | 0x8000) : 0);
        st += mcnt[q];
      }
    }
  }
  // first candidate of the prefetched vector: its key word is loaded by every lane (clamped address), so that the loads
  // of a thread's keys overlap; everything the fast path cannot decide goes to thread_probe
  uint64_t kc[PER];
  int cand[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const uint32_t m = eq_mask16(dvq[q], digest_of(hq[q]));
    const int C = (int)a.t.C;
    const int start = (int)(((C & (C - 1)) == 0 ? ((uint64_t)hq[q] & (uint64_t)(C - 1)) : ((uint64_t)hq[q] % (uint64_t)C))) & ~15;
    cand[q] = m ? start + __ffs(m) - 1 : -1;
    kc[q] = a.t.keys(bq[q] >= 0 ? bq[q] : 0)[cand[q] >= 0 ? cand[q] : 0];
  }
  PST(6);
  __syncthreads();   // s_tab / s_cnt change meaning below
  PST(7);
  // ---- probe: one lane per distinct key of the tile
  int cnt_tile[PER];   // (kPart) occurrences of my distinct keys inside the tile
#pragma unroll
  for (int q = 0; q < PER; ++q) cnt_tile[q] = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    if (!isrep[q]) continue;
    const int li = q * THREADS + threadIdx.x;
    const uint64_t key = kreg[q];
    const int t = s_t[li];
    const int cnt = s_cnt[hh[q]];
    if constexpr (kPart) cnt_tile[q] = cnt;
    int gslot = (int)a.S, base = 0;     // default: no slot
    bool defer = false;
    uint64_t* found_sc = nullptr;
    const bool idem = !a.use_count && (a.find_policy == kAssign || a.find_policy == kGlobalTimer);
    if (bq[q] >= 0) {
      const int64_t b = bq[q];
      bool inserted = false;
      int slot;
      if (a.dbg & 2) slot = (int)((uint64_t)hq[q] & (uint64_t)(a.t.C - 1));
      else if (cand[q] >= 0 && kc[q] == key) slot = cand[q];
      else slot = thread_probe<kTrain>(a, b, key, hq[q], cnt, inserted);
      if (slot >= 0) {
        gslot = (int)(b * a.t.C + slot);
        if (inserted) { s_t[li] = (uint16_t)(t | 0x8000); s_gs[li] = gslot; }
        else if (!idem) score_found(a, a.t.scores(b) + (int64_t)slot * a.t.ns, cnt);
        else found_sc = a.t.scores(b) + (int64_t)slot * a.t.ns;
      } else if (slot == -2) {
        defer = true;
      }
    }
    if constexpr (kPart) {
      // no global counter per slot: the pair becomes a record of the partition that owns its slot range; the partition's
      // block (fused_part_kernel) merges the records of a slot and ranks the tiles.  One LDS atomic here, one global
      // atomic per (tile, partition) below.
      if (found_sc) score_found(a, found_sc, cnt);
      s_tab[hh[q]] = lpq[q];              // (partition, position among the tile's records of the partition)
      s_cnt[hh[q]] = defer ? -(int)bq[q] - 2 : gslot;
    } else {
    if (kTrain) {
      if (defer) {
        const int e = atomicAdd(&a.hdr[0], 1);
        a.d_key[e] = key; a.d_tid[e] = t; a.d_cnt[e] = cnt;
        gslot = -(e + 2);
      } else {
        base = (a.dbg & 1) ? 0 : atomicAdd(&a.occ[2 * (int64_t)gslot], cnt);
        // Assign / timer scores are the same value from every tile: only the row's representative writes it
        if (found_sc && base == 0) score_found(a, found_sc, cnt);
      }
    } else if (found_sc) {
      score_found(a, found_sc, cnt);
    }
    s_tab[hh[q]] = gslot;
    s_cnt[hh[q]] = base;
    }
  }
  PST(8);
  __syncthreads();
  PST(9);
  if constexpr (kPart) {
    const int sub = kMT ? 0 : (int)blockIdx.x % kPartSub;
    constexpr int kListCap = kMT ? kPartCap : kSubCap;
    if ((int)threadIdx.x < a.P) s_hist[threadIdx.x] = my_base;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (!isrep[q]) continue;
      const int v = s_tab[hh[q]];
      const int pk = v >> 12, idx = s_hist[pk] + (v & 4095);
      int ref = -1;
      if (idx < kListCap) {
        ref = pk * kPartCap + sub * kSubCap + idx;
        if constexpr (kBags) {
          const int li = q * THREADS + threadIdx.x;
          const int sm = s_sm[hh[q]];
          a.rec[ref] = make_uint4((uint32_t)(tile0 + li), (uint32_t)((sm & 0x8000) ? (int)tile0 + (sm & 0x7fff) : s_bag[li]),
                                  (uint32_t)s_cnt[hh[q]], (uint32_t)cnt_tile[q]);
        } else
        a.rec[ref] = make_uint4((uint32_t)kreg[q], (uint32_t)(kreg[q] >> 32), (uint32_t)s_cnt[hh[q]], (uint32_t)cnt_tile[q]);
      } else {
        a.hdr[a.ovf_word] = a.ovf_val;     // a partition received more records than it can hold: the step is flagged (see the module)
      }
      s_tab[hh[q]] = ref;
    }
    __syncthreads();
  }
  PST(10);
  int nrep[HALVES];
#pragma unroll
  for (int hq = 0; hq < HALVES; ++hq) nrep[hq] = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int li = q * THREADS + threadIdx.x;
    if (hh[q] >= 0) {
      const int64_t i = tile0 + li;
      const int t = s_t[li] & 0x7fff;
      if constexpr (kPart) {
        const int g = s_cnt[hh[q]];           // slot code of the key's record
        a.occ_slot[i] = s_tab[hh[q]];         // the record
        if constexpr (kBags) {
          a.occ_trank[i] = rk[q];
          const int sm = s_sm[hh[q]];
          if (sm & 0x8000) a.tile_bags[tile0 + (sm & 0x7fff) + rk[q]] = s_bag[li];
          // address word 1: the row comes out of the partition kernel's eviction (gather_dev.h: LateRefs)
          a.occ_addr[i] = (g >= 0 && g < a.S) ? tptr_of(t) + ((int64_t)g - tbo_of(t) * a.t.C) * rowb_of(t) : (g <= -2 ? 1 : 0);
          continue;
        }
        a.csr_rank[i] = rk[q];                // rank inside the tile (completed by the scatter kernel)
        a.occ_addr[i] = (g >= 0 && g < a.S) ? tptr_of(t) + ((int64_t)g - tbo_of(t) * a.t.C) * rowb_of(t) : 0;
        continue;
      }
      const int g = s_tab[hh[q]];
      const int r = s_cnt[hh[q]] + rk[q];
      a.occ_slot[i] = g;
      if (kTrain) a.csr_rank[i] = r;
      a.occ_addr[i] = (g >= 0 && g < a.S) ? tptr_of(t) + ((int64_t)g - tbo_of(t) * a.t.C) * rowb_of(t) : 0;
      if (kTrain && g >= 0 && r == 0) ++nrep[li >> 10];
    }
  }
  if (kTrain) {
#pragma unroll
    for (int hq = 0; hq < HALVES; ++hq) {
      int v = nrep[hq];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (lane_id() == 0 && v) atomicAdd(&s_nrep[hq], v);
    }
    // first-touch initialisation of the rows this block inserted: a wave per row, coalesced stores
    const int wave = threadIdx.x >> 6, nw = THREADS >> 6;
    for (int base = wave * 64; base < TILE; base += nw * 64) {
      const int li = base + lane_id();
      uint64_t todo = __ballot((s_t[li] & 0x8000) != 0);
      while (todo) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const int l2 = base + src;
        const int t = s_t[l2] & 0x7fff;
        const int g = s_gs[l2];
        void* rp = reinterpret_cast<void*>((uintptr_t)(tptr_of(t) + ((int64_t)g - tbo_of(t) * a.t.C) * rowb_of(t)));
        wave_init_row(a, rp, s_key[l2], (int)a.table_emb_dims[t], (int)a.table_value_dims[t]);
      }
    }
    __syncthreads();
    if (threadIdx.x < HALVES) {
      const int64_t pt = (int64_t)blockIdx.x * HALVES + threadIdx.x;
      if (pt * 1024 < a.n) a.partial[pt] = s_nrep[threadIdx.x];
    }
  }
  if (blockIdx.x == 0)
    for (int t = threadIdx.x; t <= T; t += THREADS) a.seg_out[t] = kOne ? (t == 0 ? 0 : a.n) : s_seg[t];
  PST(11);
}

}  // namespace mi355
#include "probe_c.h"
namespace mi355 {

// ---- deferred keys: bucket without a free slot -> evict the minimum score (kernels.cuh:226-287, types.cuh:398-512) ----
// Barrier across a grid whose blocks are all resident.  Two levels -- 16 blocks share a counter, the last arrival of a group
// bumps the top counter, the last group publishes the generation -- because same-address device atomics serialise at
// ~40 ns each: 352 arrivals on ONE word (plus the pollers) cost ~50 us per barrier, 16 + 22 cost ~1.5 us.
// hdr: [1] top counter, [3] generation, [8 + g] group counters; all monotonic within a launch, cleared by its last block.
__device__ __forceinline__ void grid_sync(int* hdr, int k /* 1-based barrier number */, int nblk) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int ngroups = (nblk + 15) >> 4;
    const int g = (int)blockIdx.x >> 4;
    const int gsize = nblk - (g << 4) < 16 ? nblk - (g << 4) : 16;
    if (atomicAdd(&hdr[8 + g], 1) == k * gsize - 1) {
      if (atomicAdd(&hdr[1], 1) == k * ngroups - 1) {
        __threadfence();
        __hip_atomic_store(&hdr[3], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // (bounded: a grid that is not fully resident would otherwise hang the device; ~1 s, then the step is garbage)
    int spins = 0;
    while (__hip_atomic_load(&hdr[3], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < k && ++spins < (1 << 23)) __builtin_amdgcn_s_sleep(2);
    __threadfence();
  }
  __syncthreads();
}
__device__ __forceinline__ void grid_sync_reset(int* hdr) {
  hdr[1] = 0; hdr[3] = 0;
  const int ngroups = ((int)gridDim.x + 15) >> 4;
  for (int g = 0; g < ngroups; ++g) hdr[8 + g] = 0;
}

__device__ __forceinline__ void evict_phase(const FusedArgs& a, int nd, int nblk) {
  const int g = lane_id() & (G - 1);
  const int gpb = blockDim.x / G;
  const int C = (int)a.t.C;
  for (int e0 = blockIdx.x * gpb; e0 < nd; e0 += nblk * gpb) {
    const int e = e0 + threadIdx.x / G;
    const bool act = e < nd;
    const uint64_t key = act ? a.d_key[e] : kEmptyKey;
    const int tid = act ? a.d_tid[e] : 0;
    const int cnt = act ? a.d_cnt[e] : 0;
    Located L = locate(key, tid, a.tbo, a.t.C);
    int gslot = (int)a.S;
    bool done = !(act && L.ok);
    int guard = 0;
    while (__ballot(!done)) {
      if (!done) {
        int got = 0;
        if (g == 0) got = atomicCAS(&a.locks[L.bucket], 0, 1) == 0 ? 1 : 0;
        got = group_bcast(got, 0);
        if (got) {
          uint64_t* ks = a.t.keys(L.bucket);
          int found_slot, empty_slot;
          group_probe(a.t, L.bucket, key, L.hash, true, true, found_slot, empty_slot);
          int slot = -1;
          bool fresh_row = false;
          if (found_slot >= 0) {             // another deferred entry of the same key got here first
            slot = found_slot;
            if (g == 0) score_found(a, a.t.scores(L.bucket) + (int64_t)slot * a.t.ns, cnt);
          } else if (empty_slot >= 0) {      // a slot was freed meanwhile
            slot = empty_slot;
            fresh_row = true;
            if (g == 0) {
              store_digest(a.t.dig(L.bucket) + slot, digest_of(L.hash));
              score_new(a, a.t.scores(L.bucket) + (int64_t)slot * a.t.ns, cnt);
              atomicAdd(&a.bucket_sizes[L.bucket], 1);
            }
          } else {
            uint64_t best = ~0ull, bkey = 0;
            int bslot = -1;
            const uint64_t* sc = a.t.scores(L.bucket);
            const int32_t* pin = a.counter ? a.counter + L.bucket * a.t.C : nullptr;
            const int32_t* oc = a.occ + 2 * (L.bucket * a.t.C);
            for (int s0 = 2 * g; s0 < C; s0 += 2 * G) {
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const int s = s0 + u;
                const uint64_t v = ald64(sc + (int64_t)s * a.t.ns + (a.t.ns - 1));
                if (v < best) {
                  const uint64_t k = ald64(ks + s);
                  if (k == kLockedKey || k == kEmptyKey) continue;
                  if (pin && __hip_atomic_load(pin + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) continue;
                  if (__hip_atomic_load(oc + 2 * s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) continue;  // used by this batch
                  best = v; bslot = s; bkey = k;
                }
              }
            }
            group_argmin(best, bslot, bkey);
            if (bslot >= 0) {
              slot = bslot;
              fresh_row = true;
              if (g == 0) {
                ast64(ks + slot, kLockedKey);
                store_digest(a.t.dig(L.bucket) + slot, digest_of(L.hash));
                for (int64_t w = 0; w < a.t.ns; ++w) ast64((uint64_t*)sc + (int64_t)slot * a.t.ns + w, 0);
                score_new(a, a.t.scores(L.bucket) + (int64_t)slot * a.t.ns, cnt);
                if (bkey == kReclaimKey) atomicAdd(&a.bucket_sizes[L.bucket], 1);
              }
            }
          }
          if (slot >= 0) {
            gslot = (int)(L.bucket * a.t.C + slot);
            if (fresh_row) {
              void* rp = reinterpret_cast<void*>((uintptr_t)(a.table_ptrs[tid] + ((int64_t)gslot - a.tbo[tid] * a.t.C) *
                                                                                a.table_value_dims[tid] * a.elem_bytes));
              const int ed = (int)a.table_emb_dims[tid], vd = (int)a.table_value_dims[tid];
              for (int el = g; el < vd; el += G) {
                const float v = el < ed ? init_value(a.init, key, (uint32_t)el) : a.init.state_init;
                if (a.value_dtype == kF32) st1<kF32>(rp, el, v);
                else if (a.value_dtype == kBF16) st1<kBF16>(rp, el, v);
                else st1<kF16>(rp, el, v);
              }
              if (g == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                ast64(ks + slot, key);
              }
            }
          }
          if (g == 0) {
            // count the key's occurrences BEFORE the lock is released: occ > 0 is what protects the slot from the next
            // eviction in this bucket
            if (slot >= 0) { a.d_slot[e] = gslot; a.d_base[e] = atomicAdd(&a.occ[2 * (int64_t)gslot], cnt); }
            __threadfence();
            __hip_atomic_store(&a.locks[L.bucket], 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          }
          done = true;
        } else if (++guard > (1 << 22)) {
          done = true;   // give up: the key gets no slot this step
        }
      }
    }
    if (act && g == 0 && gslot == (int)a.S) {   // no slot could be had: the key is served without a row this step
      a.d_slot[e] = gslot;
      a.d_base[e] = atomicAdd(&a.occ[2 * (int64_t)gslot], cnt);
    }
  }
}

// the occurrences of the deferred keys learn their slot / rank / row address
__device__ __forceinline__ void patch_phase(const FusedArgs& a, int nblk) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)nblk * blockDim.x) {
    const int s = a.occ_slot[i];
    if (s <= -2) {
      const int e = -(s + 2);
      const int gs = a.d_slot[e];
      const int r = a.csr_rank[i] + a.d_base[e];
      a.occ_slot[i] = gs;
      a.csr_rank[i] = r;
      int64_t addr = 0;
      if (gs < a.S) {
        const int t = a.d_tid[e];
        addr = a.table_ptrs[t] + ((int64_t)gs - a.tbo[t] * a.t.C) * a.table_value_dims[t] * a.elem_bytes;
      }
      a.occ_addr[i] = addr;
      if (r == 0) atomicAdd(&a.partial[i >> 10], 1);
    }
  }
}

// ---- unique numbering: the occurrence with rank 0 represents its slot -------------------------------------------------
struct EmitOut {
  uint64_t* unique_keys;
  int64_t* table_offsets;   // [T+1]
  int64_t* table_ids;       // [n] (nullable)
  int64_t* slots;           // [n] table-relative slot of every unique key (-1: none)
  int64_t* row_addr;        // [n] row address of every unique key
  int64_t* freq;            // [n] nullable
  int32_t* csr_cnt;         // [n]
  int32_t* total;           // [1]
};

// ---- merged numbering kernel: deferred keys (if any) + unique numbering + CSR row pointers + hot-row registration --------
// One launch instead of three (evict, emit, scan_down).  Unique ids need only the per-tile representative counts the probe
// left in `partial` (every block sums its predecessors itself).  The row pointers are a second scan in unique order --
// over the occurrence counts, which are final only now -- and that one is chained across blocks by decoupled look-back:
// a block publishes the sum of its tile, then its inclusive prefix, in ONE 64-bit word per tile (status in the top bits)
// with device-scope stores; successors poll those words.  Nothing but the word itself crosses blocks, so no fence (= no
// L2 write-back) is involved; blocks are dispatched in index order, so a predecessor is always resident or finished.
// Deferred keys: the first `resident` blocks run the eviction and patch phases between two barriers of their own while
// the later blocks wait for the release flag -- the steady state (no deferred key) takes none of it.
constexpr unsigned long long kStatAgg = 1ull << 62, kStatPre = 2ull << 62, kStatMask = 3ull << 62;

__device__ __forceinline__ unsigned long long stat_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void stat_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// exclusive prefix of the tile sums in front of tile `t` (whose own sum is already published); called by the whole
// block, every thread returns the result.
// All predecessors of a round (256 per round) are polled at once -- one memory round trip for batches up to 256 K keys --
// and the walk stops at the nearest tile that already knows its inclusive prefix.
__device__ __forceinline__ unsigned long long lookback_prefix64(unsigned long long* tstat, int t, unsigned long long my_sum) {
  __shared__ int s_first;
  __shared__ unsigned long long s_sum;
  if (t == 0) return 0;   // (the caller published the tile's sum -- tile 0: as its inclusive prefix -- earlier)
  unsigned long long run = 0;
  for (int pos = t - 1;; pos -= kScanThreads) {
    if (threadIdx.x == 0) { s_first = kScanThreads; s_sum = 0; }
    __syncthreads();
    const int idx = pos - (int)threadIdx.x;
    unsigned long long v = idx >= 0 ? stat_load(tstat + idx) : kStatPre;   // in front of tile 0: prefix 0
    while ((v & kStatMask) == 0) { __builtin_amdgcn_s_sleep(1); v = stat_load(tstat + idx); }
    if ((v & kStatMask) == kStatPre) atomicMin(&s_first, (int)threadIdx.x);
    __syncthreads();
    const int first = s_first;      // nearest predecessor of this round that knows its inclusive prefix (256: none)
    unsigned long long val = (int)threadIdx.x <= first ? (v & ~kStatMask) : 0ull;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t lo = __shfl_xor((int)(uint32_t)val, off, 64), hi = __shfl_xor((int)(uint32_t)(val >> 32), off, 64);
      val += ((unsigned long long)hi << 32) | lo;
    }
    if (lane_id() == 0 && val) atomicAdd(&s_sum, val);
    __syncthreads();
    run += s_sum;
    __syncthreads();
    if (first < kScanThreads) break;
  }
  if (threadIdx.x == 0) stat_store(tstat + t, kStatPre | (run + my_sum));
  return run;
}
// Sums of two packed words over ALL predecessors of partition `t` (t < kPartMax): the words only ever hold a partition's
// own sums (status kStatAgg), every block adds up its predecessors itself -- up to 4 per thread, all polled at once, so
// the whole look-back is one memory round trip once the predecessors have published.
__device__ __forceinline__ void lookback_sum2(const unsigned long long* ta, const unsigned long long* tb, int t,
                                              unsigned long long& pre_a, unsigned long long& pre_b) {
  __shared__ unsigned long long s_sa, s_sb;
  if (threadIdx.x == 0) { s_sa = 0; s_sb = 0; }
  __syncthreads();
  constexpr int NJ = kPartMax / kScanThreads;
  unsigned long long va[NJ], vb[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int idx = t - 1 - ((int)threadIdx.x + j * kScanThreads);
    va[j] = idx >= 0 ? stat_load(ta + idx) : kStatAgg;
    vb[j] = idx >= 0 ? stat_load(tb + idx) : kStatAgg;
  }
  unsigned long long xa = 0, xb = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int idx = t - 1 - ((int)threadIdx.x + j * kScanThreads);
    while ((va[j] & kStatMask) == 0) { __builtin_amdgcn_s_sleep(1); va[j] = stat_load(ta + idx); }
    while ((vb[j] & kStatMask) == 0) { __builtin_amdgcn_s_sleep(1); vb[j] = stat_load(tb + idx); }
    xa += va[j] & ~kStatMask;
    xb += vb[j] & ~kStatMask;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t lo = __shfl_xor((int)(uint32_t)xa, off, 64), hi = __shfl_xor((int)(uint32_t)(xa >> 32), off, 64);
    xa += ((unsigned long long)hi << 32) | lo;
    lo = __shfl_xor((int)(uint32_t)xb, off, 64); hi = __shfl_xor((int)(uint32_t)(xb >> 32), off, 64);
    xb += ((unsigned long long)hi << 32) | lo;
  }
  if (lane_id() == 0) { if (xa) atomicAdd(&s_sa, xa); if (xb) atomicAdd(&s_sb, xb); }
  __syncthreads();
  pre_a = s_sa; pre_b = s_sb;
}
__device__ __forceinline__ int lookback_prefix(unsigned long long* tstat, int t, int my_sum) {
  return (int)lookback_prefix64(tstat, t, (unsigned long long)(unsigned)my_sum);
}

template <bool kSelf>
__global__ void __launch_bounds__(kScanThreads)
fused_mid_kernel(FusedArgs a, EmitOut o, int* __restrict__ ptr, HotList hot, bool build_hot, int resident) {
  __shared__ int s_ex[kScanTile + 1];       // exclusive representative count in front of every item of the tile
  __shared__ int64_t s_seg[kFusedMaxT + 1];
  __shared__ int s_hbase[3];
  if (a.gate && __hip_atomic_load(a.gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.gate_val) return;   // (grid uniform)
  const int64_t tile0 = (int64_t)blockIdx.x * kScanTile;
  const int64_t n = a.n;
  const int T = a.T;
  int sl[kScanItems], rk[kScanItems], f[kScanItems], oc[kScanItems];
  uint64_t ky[kScanItems];
  int64_t ad[kScanItems];
  // every load is unconditional (clamped) and issued up front, in front of the deferred-key count as well: in the steady
  // state (no deferred key) nothing changes them any more, otherwise they are read again behind the eviction
  auto load_items = [&]() {
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      const int64_t i = tile0 + threadIdx.x * kScanItems + k;
      const int64_t ic = i < n ? i : n - 1;
      sl[k] = a.occ_slot[ic];
      rk[k] = a.csr_rank[ic];
      ky[k] = a.keys[ic];
      ad[k] = a.occ_addr[ic];
    }
  };
  load_items();
  // representatives of the earlier tiles (self_prefix of scan_dev.h, its loads hoisted to the top of the chain)
  int pre_part = 0;
  if (kSelf) for (int j = threadIdx.x; j < (int)blockIdx.x; j += kScanThreads) pre_part += a.partial[j];
  else pre_part = a.partial[blockIdx.x];
  for (int t = threadIdx.x; t <= T; t += kScanThreads) s_seg[t] = a.seg_out[t];
  // ---- deferred keys (bucket without a free slot) ----
  int nd = __hip_atomic_load(&a.hdr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (nd > 0) {
    if (!a.timer) a.timer = device_clock();
    if ((int64_t)nd > a.n) nd = (int)a.n;
    const int R = (int)gridDim.x < resident ? (int)gridDim.x : resident;
    if ((int)blockIdx.x < R) {
      evict_phase(a, nd, R);
      grid_sync(a.hdr, 1, R);
      patch_phase(a, R);
      grid_sync(a.hdr, 2, R);
      if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&a.hdr[4], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(&a.hdr[4], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0 && ++spins < (1 << 24)) __builtin_amdgcn_s_sleep(4);
        __threadfence();
      }
      __syncthreads();
    }
    load_items();
    pre_part = 0;   // (patch_phase adds the deferred keys' representatives to `partial`)
    if (kSelf) for (int j = threadIdx.x; j < (int)blockIdx.x; j += kScanThreads) pre_part += a.partial[j];
    else pre_part = a.partial[blockIdx.x];
  }
  // ---- unique numbering: the occurrence with rank 0 represents its slot ----
  int c = 0, cs = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    const int64_t i = tile0 + threadIdx.x * kScanItems + k;
    f[k] = (i < n) && (rk[k] == 0) && (sl[k] >= 0);
    oc[k] = a.occ[2 * (int64_t)(f[k] ? sl[k] : 0)];     // random 4-byte gathers: only the representatives' (the others hit one line)
    c += f[k];
  }
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) cs += f[k] ? oc[k] : 0;
  // the tile's occurrence sum goes out first: the successors' look-back waits for it
  int tot, tot2 = 0;
  int ex2 = ptr ? block_excl_scan(cs, tot2) : 0;     // occurrences of the tile's earlier representatives
  if (ptr && threadIdx.x == 0)
    stat_store(a.tstat + blockIdx.x, (blockIdx.x == 0 ? kStatPre : kStatAgg) | (unsigned)tot2);
  int pre = pre_part;
  if (kSelf) block_excl_scan(pre_part, pre);
  int ex = block_excl_scan(c, tot) + pre;
  const int ex_first = ex;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    s_ex[threadIdx.x * kScanItems + k] = ex;
    if (f[k]) {
      const int s = sl[k];
      *reinterpret_cast<int2*>(a.occ + 2 * (int64_t)s) = make_int2(0, ex);   // counter cleared, unique id set: one store
      o.unique_keys[ex] = ky[k];
      o.csr_cnt[ex] = oc[k];
      if (o.freq) o.freq[ex] = oc[k];
      o.row_addr[ex] = ad[k];
      int ti = 0;
      if (T > 1) {
        const int64_t i = tile0 + threadIdx.x * kScanItems + k;
        int lo = 0, hi = T + 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_seg[mid] <= i) lo = mid + 1; else hi = mid; }
        ti = lo - 1 < 0 ? 0 : (lo - 1 >= T ? T - 1 : lo - 1);
      }
      if (o.table_ids) o.table_ids[ex] = ti;
      o.slots[ex] = s < a.S ? (int64_t)s - a.tbo[ti] * a.t.C : -1;
      ++ex;
    }
  }
  if (threadIdx.x == kScanThreads - 1) s_ex[kScanTile] = ex;
  __syncthreads();
  // unique offsets of the tables whose first key lies in this tile (or behind the batch: the last tile writes those)
  for (int t = threadIdx.x; t <= T; t += kScanThreads) {
    const int64_t p = s_seg[t];
    if (p >= tile0 && p < tile0 + kScanTile && p < n) o.table_offsets[t] = s_ex[p - tile0];
    else if (p >= n && blockIdx.x == gridDim.x - 1) o.table_offsets[t] = pre + tot;
  }
  if (!ptr) return;
  // ---- row pointers of the backward's CSR + hot rows (scan_down_kernel's scheme: ONE atomic triple per block) ----
  int h_ex = 0, t_ex = 0, w_ex = 0;
  if (build_hot) {
    int nh_local = 0, nt_local = 0, nw_local = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
      if (!f[k]) continue;
      if (oc[k] > hot.khot && oc[k] <= hot.kwave) ++nw_local;
      else if (oc[k] > hot.khot) { ++nh_local; nt_local += (oc[k] + hot.kchunk - 1) / hot.kchunk; }
    }
    if (__syncthreads_or(nh_local | nw_local)) {   // (most tiles hold no hot row: the Zipf head is numbered by the first ones)
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
  }
  const int p2 = lookback_prefix(a.tstat, (int)blockIdx.x, tot2);
  ex2 += p2;
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { ptr[pre + tot] = p2 + tot2; *o.total = p2 + tot2; }
  ex = ex_first;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (!f[k]) continue;
    ptr[ex] = ex2;
    if (build_hot && oc[k] > hot.khot && oc[k] <= hot.kwave) {
      const int w = w_ex++;
      if (w < hot.max_hot) { hot.wave_u[w] = ex; hot.wave_lo[w] = ex2; hot.wave_cnt[w] = oc[k]; }
    } else if (build_hot && oc[k] > hot.khot) {
      const int nch = (oc[k] + hot.kchunk - 1) / hot.kchunk;
      const int h = h_ex++, t0 = t_ex;
      t_ex += nch;
      if (h < hot.max_hot && t0 + nch <= hot.max_tasks) {   // tasks are written out by csr_scatter_kernel
        hot.hot_done[h] = 0;
        hot.hot_nchunks[h] = nch;
        hot.hot_u[h] = ex;
        hot.hot_lo[h] = ex2;
        hot.hot_cnt[h] = oc[k];
        hot.hot_t0[h] = t0;
      }
    }
    ex2 += oc[k];
    ++ex;
  }
}

// ---- round 3, path (c): the partition kernel ALSO writes the backward's CSR -- no scatter kernel ---------------------------------
// Measured in round 2 / 3 (phase stamps, tools/index_phase_stamps.py): every kernel of the chain costs 4-8 us beyond the life of
// its blocks (dispatch ramp, tail, end-of-kernel write-back), and the scatter kernel existed only because the CSR positions come
// out of the partition kernel.  Here the probe kernel leaves the bag id in every record (key seen once in its tile) or a pointer
// to the key's bag list of the tile (tile_bags), so the partition block writes the CSR entries itself: the bag id, or REFERENCES
// ~(list index) that the backward follows (backward.hip: entry_src) -- nobody copies the lists, the hot partition stores 20 K
// words and is done.  The per-occurrence outputs nothing on the training path reads (reverse indices, full ranks) come out of
// the records on demand (mi355_demb_fused_materialize).
//  * 1024 threads per partition (two records, two hash entries per thread): the block is a chain of short phases, and its
//    life is what the launch costs;
//  * a 2048-entry LDS hash whose entry ORDER is the unique order: ONE block scan over the entries yields the local unique ids,
//    the CSR prefix and the hot-list positions (no per-unique counters, no second id array);
//  * keys whose bucket was full are evicted for right here (every record of a bucket is in this block); their occurrences
//    carry the address word 1 and the gather finds the row through the record (gather_dev.h: LateRefs).
//  (Measured and rejected in round 3: the partition blocks riding in the gather's launch.  The gather's throughput is
//   proportional to its resident waves -- 4.6 us per bag and lane group whatever the occupancy -- and the partition code's 79
//   registers / 24 KB of LDS take a quarter of them: 54-60 us for the fused launch against 21 + 30 us apart.)
// The partition kernel of path (c) is a template over the record capacity of a partition (round 5): 2 048 for batches up to
// 1 M keys (the probe kernel's reservation lists).  A partition
// holds at most CAP records, hence at most as many distinct slots: the LDS hash has CAP entries.
constexpr int kP3Threads = 1024;
constexpr int kRecLate = 1 << 30;      // count word of a record whose key took the eviction path
constexpr int kPartMaxBig = 4096;      // partition slots of the aux header / the partition table (layout constant; path (c) uses up to kPartMax)

template <int HASH> __device__ __forceinline__ int p2_hash(int slot) { return (int)((uint32_t)slot * 2654435761u >> (32 - __builtin_ctz(HASH))) & (HASH - 1); }
template <int HASH>
__device__ __forceinline__ int p2_find(const int* h_slot, int slot) {   // -1: not present
  int h = p2_hash<HASH>(slot);
  for (int n = 0; n < HASH; ++n) {
    const int cur = h_slot[h];
    if (cur == slot) return h;
    if (cur == -1) return -1;
    h = (h + 1) & (HASH - 1);
  }
  return -1;
}
template <int HASH>
__device__ __forceinline__ int p2_insert(int* h_slot, int slot, bool* claimed) {
  int h = p2_hash<HASH>(slot);
  *claimed = false;
  while (true) {
    const int cur = atomicCAS(&h_slot[h], -1, slot);
    if (cur == -1) { *claimed = true; return h; }
    if (cur == slot) return h;
    h = (h + 1) & (HASH - 1);
  }
}

// exclusive scan of five ints per thread across a THREADS-thread block; totals in `tot`.  Two levels: every wave scans its own
// values, ONE wave scans the wave totals (a thread reading all 16 totals of five values itself was 80 LDS reads per thread).
template <int THREADS, typename Publish>
__device__ __forceinline__ void block_scan5(int (&v)[5], int (&tot)[5], Publish publish) {
  constexpr int NW = THREADS / 64;
  __shared__ int s_w5[5][NW + 1];      // [i][w]: exclusive base of wave w; [i][NW]: total
  const int w = threadIdx.x >> 6;
  int incl[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) incl[i] = wave_incl_scan(v[i]);
  if (lane_id() == 63) {
#pragma unroll
    for (int i = 0; i < 5; ++i) s_w5[i][w] = incl[i];
  }
  __syncthreads();
  if (w == 0) {
    int t5[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int x = lane_id() < NW ? s_w5[i][lane_id()] : 0;
      const int in2 = wave_incl_scan(x);
      if (lane_id() < NW) s_w5[i][lane_id()] = in2 - x;
      if (lane_id() == NW - 1) s_w5[i][NW] = in2;
      t5[i] = in2;
    }
    // round 5: the lane that holds the block totals hands them on at once (the partition kernel publishes its look-back words
    // from here, one barrier and an LDS read-back earlier than from behind the scan)
    if (lane_id() == NW - 1) publish(t5);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    tot[i] = s_w5[i][NW];
    v[i] = s_w5[i][w] + incl[i] - v[i];
  }
  // (no trailing barrier: the one caller uses the staging array once per block)
}

// sums of two packed words over ALL predecessors of partition t: one word pair per thread and round of 1 024 (one round for the
// batches up to 1 M keys), all of a round polled together
template <int THREADS = kP3Threads>
__device__ __forceinline__ void lookback_sum2_1024(const unsigned long long* ta, const unsigned long long* tb, int t,
                                                   unsigned long long& pre_a, unsigned long long& pre_b) {
  __shared__ unsigned long long s_sa, s_sb;
  if (threadIdx.x == 0) { s_sa = 0; s_sb = 0; }
  __syncthreads();
  unsigned long long xa = 0, xb = 0;
  for (int r0 = 0; r0 < t; r0 += THREADS) {
    const int idx = t - 1 - r0 - (int)threadIdx.x;
    unsigned long long va = idx >= 0 ? stat_load(ta + idx) : kStatAgg, vb = idx >= 0 ? stat_load(tb + idx) : kStatAgg;
    while ((va & kStatMask) == 0) { __builtin_amdgcn_s_sleep(1); va = stat_load(ta + idx); }
    while ((vb & kStatMask) == 0) { __builtin_amdgcn_s_sleep(1); vb = stat_load(tb + idx); }
    xa += va & ~kStatMask;
    xb += vb & ~kStatMask;
  }
  // the five packed fields are summed over the wave on the DPP path (each stays below 2^31 over all partitions)
  const int f0 = wave_sum((int)(xa & 0x7fffffffull)), f1 = wave_sum((int)(xa >> 31));
  const int f2 = wave_sum((int)(xb >> 40)), f3 = wave_sum((int)((xb >> 20) & 0xfffff)), f4 = wave_sum((int)(xb & 0xfffff));
  xa = ((unsigned long long)(unsigned)f1 << 31) + (unsigned long long)(unsigned)f0;
  xb = ((unsigned long long)(unsigned)f2 << 40) + ((unsigned long long)(unsigned)f3 << 20) + (unsigned long long)(unsigned)f4;
  if (lane_id() == 0) { if (xa) atomicAdd(&s_sa, xa); if (xb) atomicAdd(&s_sb, xb); }
  __syncthreads();
  pre_a = s_sa; pre_b = s_sb;
}

// ---- deferred keys of a partition: the bucket had no free slot.  Evict the minimum score among the slots this batch does not use
//      (the partition's LDS hash knows them all: every record of the bucket is in this block) and that nobody pinned
//      (kernels.cuh:226-287, types.cuh:398-512); 8 lanes per key, a hashed LDS lock per bucket.  The slot goes back into the
//      record: the gather finds the rows of the key's occurrences there.  Shared by the two partition kernels.
#ifdef P3_EVICT_NOINLINE
#define P3_EVICT_ATTR __attribute__((noinline))
#else
#define P3_EVICT_ATTR __forceinline__
#endif
template <int HASH, typename DRec, int THREADS = kP3Threads>
__device__ P3_EVICT_ATTR void part_evict(FusedArgs& a, int nd, int64_t rec_base, const DRec* d_rec, int* d_ent, int* d_base, int* s_lock,
                                           unsigned* s_late, int* h_slot, int* h_cnt, int tbl, int64_t tp0, int64_t rowb, int64_t s0,
                                           const uint64_t* d_key = nullptr, const int2* d_zw = nullptr, unsigned* s_fresh = nullptr) {
      // s_fresh (optional, LDS, one bit per deferred record, zeroed by the caller): rows of freshly taken slots are NOT
      // initialised by the 8 lanes that evicted for them (16 Philox draws each in a row: 12.5 K of the 27.8 K cycles one eviction
      // took, profiles/r05_eviction_stamps.txt) but by the whole block behind the pass, one element per thread
      // d_key / d_zw (optional, LDS): key and (slot code, count) of every deferred record, left by the thread that held the
      // record in registers -- two dependent round trips (record, key) off the front of the chain
      EST(0);
      if (!a.timer) a.timer = device_clock();
      const int g = lane_id() & (G - 1);
      const int gpb = THREADS / G;
      const int C = (int)a.t.C;
      for (int e0 = 0; e0 < nd; e0 += gpb) {
        const int e = e0 + (int)threadIdx.x / G;
        const bool act = e < nd;
        const int64_t r = rec_base + (act ? (int)d_rec[e] : 0);
        uint64_t key;
        int cnt, zc;
        if (d_key) {
          key = d_key[act ? e : 0];
          const int2 zw = d_zw[act ? e : 0];
          zc = zw.x; cnt = zw.y;
        } else {
          const uint4 rd = a.rec[r];
          int64_t kp = (int64_t)rd.x; kp = kp < a.n ? kp : a.n - 1;
          key = a.keys[kp];
          cnt = (int)rd.w; zc = (int)rd.z;
        }
        const int64_t bucket = act ? -(int64_t)zc - 2 : 0;
        const int64_t hash = (int64_t)(fmix64(key) & 0x7FFFFFFFFFFFFFFFull);
        bool done = !act;
        int guard = 0;
        while (__ballot(!done)) {
          if (!done) {
            int got = 0;
            if (g == 0) got = atomicCAS(&s_lock[bucket & 255], 0, 1) == 0 ? 1 : 0;
            got = group_bcast(got, 0);
            if (got) {
              EST(1);
              uint64_t* ks = a.t.keys(bucket);
              int found_slot, empty_slot;
              // (the re-probe stays even for the first eviction a bucket sees in this block: skipping it -- "the bucket is as the
              //  probe kernel left it" -- measured nothing (36.4 us either way) and is wrong for a key the probe kernel gave up on
              //  behind a stuck Locked word, which may be in the bucket by now)
              group_probe(a.t, bucket, key, hash, true, true, found_slot, empty_slot);
              EST(2);
              int slot = -1;
              bool fresh_row = false;
              if (found_slot >= 0) {             // another record of the same key got here first
                slot = found_slot;
                if (g == 0) score_found(a, a.t.scores(bucket) + (int64_t)slot * a.t.ns, cnt);
              } else if (empty_slot >= 0) {      // a slot was freed meanwhile
                slot = empty_slot;
                fresh_row = true;
                if (g == 0) {
                  store_digest(a.t.dig(bucket) + slot, digest_of(hash));
                  score_new(a, a.t.scores(bucket) + (int64_t)slot * a.t.ns, cnt);
                  atomicAdd(&a.bucket_sizes[bucket], 1);
                }
              } else {
                // The lane's minimum FIRST, its eligibility second (round 5).  The scan used to test every slot that beat the
                // running minimum as it went -- score, then key, then pin counter, three dependent round trips, up to 16 times per
                // lane: ~25 us for one key, and a partition kernel that evicts for a single key holds up every partition behind
                // it in the look-back (table at 75 % load: 19.8 -> 48 us, profiles/r05_eviction_regime.txt).  Now the lane's 16
                // scores are fetched six at a time, the smallest (score, slot) is tested, and only a refused one (locked, pinned, in
                // use by this batch: rare) sends the lane round again for the next smallest.
                uint64_t best = ~0ull, bkey = 0;
                int bslot = -1;
                const uint64_t* sc = a.t.scores(bucket);
                const int32_t* pin = a.counter ? a.counter + bucket * a.t.C : nullptr;
                uint64_t lo_s = 0;
                int lo_slot = -1;                  // candidates are (score, slot) > (lo_s, lo_slot), lexicographically
                for (int tries = 0; tries < C; ++tries) {
                  uint64_t cs = ~0ull;
                  int cslot = -1;
                  for (int s1 = 2 * g; s1 < C; s1 += 6 * G) {    // (three passes at the usual 128 slots per bucket, 6 loads in flight: 8 leave 20 bytes
                    uint64_t v[6];                                 //  of scratch in the kernel's steady path (+1.2 us), 16 -- or 8 in front of the probe -- more)
  #pragma unroll
                    for (int u = 0; u < 6; ++u) {
                      const int s2 = s1 + (u >> 1) * 2 * G + (u & 1);
                      v[u] = s2 < C ? ald64(sc + (int64_t)s2 * a.t.ns + (a.t.ns - 1)) : ~0ull;
                    }
  #pragma unroll
                    for (int u = 0; u < 6; ++u) {
                      const int s2 = s1 + (u >> 1) * 2 * G + (u & 1);
                      const bool above = v[u] > lo_s || (v[u] == lo_s && s2 > lo_slot);
                      if (s2 < C && above && (v[u] < cs || cslot < 0)) { cs = v[u]; cslot = s2; }   // (ascending slots: ties keep the lower one)
                    }
                  }
                  if (cslot < 0) break;
                  if (cs >= a.protect) { cslot = -1; break; }     // everything from here up belongs to a step in flight (ascending order)
                  const uint64_t k2 = ald64(ks + cslot);
                  const int pv = pin ? __hip_atomic_load(pin + cslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                  if (k2 != kLockedKey && k2 != kEmptyKey && pv <= 0 && p2_find<HASH>(h_slot, (int)(bucket * a.t.C + cslot)) < 0) {
                    best = cs; bslot = cslot; bkey = k2;
                    break;
                  }
                  lo_s = cs; lo_slot = cslot;
                }
                EST(3);
                group_argmin(best, bslot, bkey);
                if (bslot >= 0) {
                  slot = bslot;
                  fresh_row = true;
                  if (g == 0) {
                    ast64(ks + slot, kLockedKey);
                    store_digest(a.t.dig(bucket) + slot, digest_of(hash));
                    for (int64_t w = 0; w < a.t.ns; ++w) ast64((uint64_t*)sc + (int64_t)slot * a.t.ns + w, 0);
                    score_new(a, a.t.scores(bucket) + (int64_t)slot * a.t.ns, cnt);
                    if (bkey == kReclaimKey) atomicAdd(&a.bucket_sizes[bucket], 1);
                  }
                }
              }
              EST(4);
              int gslot = (int)a.S;             // no slot could be had: the key is served without a row this step
              if (slot >= 0) {
                gslot = (int)(bucket * a.t.C + slot);
                if (fresh_row && s_fresh) {
                  if (g == 0) { atomicOr(&s_fresh[e >> 5], 1u << (e & 31)); ast64(ks + slot, key); }
                } else if (fresh_row) {
                  void* rp = reinterpret_cast<void*>((uintptr_t)(tp0 + ((int64_t)gslot - s0) * rowb));
                  const int ed = (int)a.table_emb_dims[tbl], vd = (int)a.table_value_dims[tbl];
                  for (int el = g; el < vd; el += G) {
                    const float v = el < ed ? init_value(a.init, key, (uint32_t)el) : a.init.state_init;
                    if (a.value_dtype == kF32) st1<kF32>(rp, el, v);
                    else if (a.value_dtype == kBF16) st1<kBF16>(rp, el, v);
                    else st1<kF16>(rp, el, v);
                  }
                  EST(5);
                  if (g == 0) {
                    // the row is complete before its key can be seen -- needed only where somebody reads rows while this kernel
                    // runs: the gather blocks of the fused launch (part3_lean.h) behind the partition's ready flag
                    if (a.part_ready) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    ast64(ks + slot, key);
                  }
                }
              }
              EST(6);
              if (g == 0) {
                // the slot enters the hash BEFORE the lock is released: that is what protects it from the next eviction
                bool cl;
                const int ent = p2_insert<HASH>(h_slot, gslot, &cl);
                if (cl) atomicOr(&s_late[ent >> 5], 1u << (ent & 31));
                d_ent[e] = ent; d_base[e] = atomicAdd(&h_cnt[ent], cnt);
                a.rec[r].z = (uint32_t)gslot;
                a.rec[r].w = (uint32_t)(cnt | kRecLate);
                __threadfence_block();
                atomicExch(&s_lock[bucket & 255], 0);
              }
              EST(7);
              done = true;
            } else if (++guard > (1 << 22)) {
              if (g == 0) {
                bool cl;
                const int ent = p2_insert<HASH>(h_slot, (int)a.S, &cl);
                if (cl) atomicOr(&s_late[ent >> 5], 1u << (ent & 31));
                d_ent[e] = ent; d_base[e] = atomicAdd(&h_cnt[ent], cnt);
                a.rec[r].z = (uint32_t)a.S;
                a.rec[r].w = (uint32_t)(cnt | kRecLate);
              }
              done = true;
            }
          }
        }
      }
      EST(8);
      if (s_fresh) {     // rows of the freshly taken slots: every thread of the block draws its elements
        __syncthreads();
        const int ed = (int)a.table_emb_dims[tbl], vd = (int)a.table_value_dims[tbl];
        for (int item = (int)threadIdx.x; item < nd * vd; item += THREADS) {
          const int e = item / vd, el = item - e * vd;
          if (!((s_fresh[e >> 5] >> (e & 31)) & 1u)) continue;
          const int gslot = h_slot[d_ent[e]];
          void* rp = reinterpret_cast<void*>((uintptr_t)(tp0 + ((int64_t)gslot - s0) * rowb));
          uint64_t key;
          if (d_key) key = d_key[e];
          else {                       // (callers without the LDS copy: through the record)
            int64_t kp = (int64_t)a.rec[rec_base + (int)d_rec[e]].x;
            kp = kp < a.n ? kp : a.n - 1;
            key = a.keys[kp];
          }
          const float v = el < ed ? init_value(a.init, key, (uint32_t)el) : a.init.state_init;
          if (a.value_dtype == kF32) st1<kF32>(rp, el, v);
          else if (a.value_dtype == kBF16) st1<kBF16>(rp, el, v);
          else st1<kF16>(rp, el, v);
        }
        // (rows read while this kernel runs -- the gather blocks of the fused launch behind the partition's ready flag -- must be
        //  complete before the caller's barrier lets thread 0 publish the flag)
        if (a.part_ready) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      EST(9);
}

// every block of the probe kernel has retired when a partition block starts: the overflow word of this step is final
__device__ __forceinline__ void publish_notice(const FusedArgs& a) {
  const unsigned long long f = __hip_atomic_load(&a.hdr[a.ovf_word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.ovf_val ? 1ull : 0ull;
  __hip_atomic_store(a.notice, (unsigned long long)(unsigned)a.ovf_val | (f << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int CAP>
__global__ void __launch_bounds__(kP3Threads)
fused_part3_kernel(FusedArgs a, EmitOut o, int* __restrict__ ptr, int* __restrict__ csr_src, HotList hot) {
  constexpr int kPartCap = CAP, kSubCap = CAP / kPartSub, kP2Hash = CAP, kP3Items = CAP / kP3Threads, kP3Ent = CAP / kP3Threads;
  __shared__ int h_slot[kP2Hash];         // slot of the entry (-1: free)
  __shared__ int h_cnt[kP2Hash];          // occurrences of the slot
  __shared__ int h_pl[kP2Hash];           // occurrences in front of the entry's row (partition-local)
  __shared__ unsigned short h_lid[kP2Hash];   // local unique id of the entry
  __shared__ short d_rec[kPartCap];       // deferred records (bucket full)
  __shared__ int d_ent[kPartCap / 4], d_base[kPartCap / 4];   // their hash entry / rank base once resolved (first CAP / 4 per step)
  __shared__ uint64_t d_key[kPartCap / 4];                    // ... their keys and (slot code, count): see part_evict
  __shared__ int2 d_zw[kPartCap / 4];
  __shared__ unsigned s_fresh[kPartCap / 4 / 32];            // ... and which of them took a fresh slot (row initialised by the block)
  __shared__ int s_lock[256];             // bucket locks of the eviction (hashed)
  __shared__ unsigned s_late[kP2Hash / 32];
  __shared__ int s_nd, s_nbig;
  constexpr int kBigMax = 512;            // records with more than 8 occurrences (reference lists expanded wave by wave)
  __shared__ int b_pos[kBigMax], b_ref[kBigMax], b_cnt[kBigMax];
  QST(0);
  constexpr int kDefMax = kPartCap / 4;
  const int p = blockIdx.x;
  if (a.notice && p == 0 && threadIdx.x == 0) publish_notice(a);
  // the sub-list counts come in through the VECTOR memory path (one lane per sub-list, broadcast behind the barrier): as uniform
  // scalar loads they share the LDS counter of the wave, and the barrier below -- which waits for the LDS initialisation --
  // waited for their round trip as well (profiles/r05_index_phase_stamps_1024x2.txt: 2 us in front of the first barrier)
  const int mv = a.pcount[p * kPartSub + ((int)threadIdx.x & (kPartSub - 1))];
  uint4 rc[kP3Items];
#pragma unroll
  for (int k = 0; k < kP3Items; ++k) rc[k] = a.rec[(int64_t)p * kPartCap + threadIdx.x + k * kP3Threads];
  for (int i = threadIdx.x; i < kP2Hash; i += kP3Threads) { h_slot[i] = -1; h_cnt[i] = 0; }
  if (threadIdx.x < 256) s_lock[threadIdx.x] = 0;
  if (threadIdx.x < kP2Hash / 32) s_late[threadIdx.x] = 0;
  if (threadIdx.x < kPartCap / 4 / 32) s_fresh[threadIdx.x] = 0;
  if (threadIdx.x == 0) { s_nd = 0; s_nbig = 0; }
  QST(1);
  __syncthreads();
  QST(2);
  int msub[kPartSub];
#pragma unroll
  for (int r = 0; r < kPartSub; ++r) msub[r] = __builtin_amdgcn_readlane(mv, r);
  // (rows of the partition's table: three scalars every output needs -- fetched behind the first barrier, used much later;
  //  several tables: the probe kernel's block 0 left the table of every partition in ptab)
  int tbl = 0;
  bool first_of_table = p == 0;
  if (a.mt) { tbl = a.ptab[p]; first_of_table = p == 0 || a.ptab[p - 1] != tbl; }
  const int64_t tp0 = a.table_ptrs[tbl], rowb = a.table_value_dims[tbl] * a.elem_bytes, s0 = a.tbo[tbl] * a.t.C;
  if (threadIdx.x < kPartSub) a.pcount[p * kPartSub + threadIdx.x] = 0;       // clean for the next step
  // ---- merge the records of a slot; the record that creates the entry owns the unique row's key
  int en[kP3Items], bs[kP3Items], dj[kP3Items];
  bool live[kP3Items], mine[kP3Items];
#pragma unroll
  for (int k = 0; k < kP3Items; ++k) {
    const int idx = threadIdx.x + k * kP3Threads;
    const int ms = msub[idx / kSubCap];
    live[k] = a.mt ? idx < (msub[0] < kPartCap ? msub[0] : kPartCap) : (idx % kSubCap) < (ms < kSubCap ? ms : kSubCap);
    en[k] = -1; bs[k] = 0; dj[k] = -1; mine[k] = false;
    if (!live[k]) continue;
    const int sl = (int)rc[k].z, cn = (int)rc[k].w;
    if (sl >= 0) {
      bool cl;
      en[k] = p2_insert<kP2Hash>(h_slot, sl, &cl);
      mine[k] = cl;
      bs[k] = atomicAdd(&h_cnt[en[k]], cn);
    } else {
      dj[k] = atomicAdd(&s_nd, 1);
      if (dj[k] < kDefMax) {
        d_rec[dj[k]] = (short)idx;
      } else {                           // beyond what one step evicts for: no slot this step (like an insert that returns Busy)
        dj[k] = -2;
        bool cl;
        en[k] = p2_insert<kP2Hash>(h_slot, (int)a.S, &cl);
        mine[k] = cl;
        bs[k] = atomicAdd(&h_cnt[en[k]], cn);
      }
    }
  }
  // the key of a record that created an entry is the unique key of its row (fetched now, stored with the outputs)
  uint64_t ky[kP3Items];
#pragma unroll
  for (int k = 0; k < kP3Items; ++k) {
    int64_t pc = live[k] ? (int64_t)rc[k].x : 0;
    pc = pc < a.n ? pc : a.n - 1;
    ky[k] = a.keys[pc];
  }
  QST(3);
  __syncthreads();
  QST(4);
  // ---- deferred keys: the bucket had no free slot.  Evict the minimum score among the slots this batch does not use (the
  //      hash above knows them all: every record of the bucket is in this block) and that nobody pinned (kernels.cuh:226-287,
  //      types.cuh:398-512); 8 lanes per key, a hashed LDS lock per bucket.  The slot goes back into the record: the gather
  //      finds the rows of the key's occurrences there.
#ifdef P3_NO_EVICT
  const int nd = 0;
#else
  const int nd = s_nd < kDefMax ? s_nd : kDefMax;
#endif
  if (nd > 0) {
    // (key and (slot code, count) of the deferred records go to LDS here, behind the barrier and only in a block that evicts: in
    //  front of it the stores made every block wait for its key loads -- +1.2 us on the steady-state kernel)
#pragma unroll
    for (int k = 0; k < kP3Items; ++k)
      if (dj[k] >= 0) { d_key[dj[k]] = ky[k]; d_zw[dj[k]] = make_int2((int)rc[k].z, (int)rc[k].w); }
    __syncthreads();
    part_evict<kP2Hash>(a, nd, (int64_t)p * kPartCap, d_rec, d_ent, d_base, s_lock, s_late, h_slot, h_cnt, tbl, tp0, rowb, s0, d_key, d_zw,
#ifdef P3_NO_FRESH
                        nullptr
#else
                        s_fresh
#endif
                        );
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kP3Items; ++k)
      if (dj[k] >= 0) {
        en[k] = d_ent[dj[k]]; bs[k] = d_base[dj[k]];
        // the first record (rank base 0) of an entry created by the eviction owns the unique row's key
        if (((s_late[en[k] >> 5] >> (en[k] & 31)) & 1u) && bs[k] == 0) mine[k] = true;
      }
  }
#pragma unroll
  for (int k = 0; k < kP3Items; ++k)
    if (dj[k] == -2) {   // beyond the eviction budget: the record says "no slot"
      const int64_t r = (int64_t)p * kPartCap + threadIdx.x + k * kP3Threads;
      a.rec[r].z = (uint32_t)a.S;
      a.rec[r].w = (uint32_t)((int)rc[k].w | kRecLate);
    }
  if (nd > 0) __syncthreads();      // (block uniform; without deferred keys the hash has not changed since the barrier above)
  // ---- one scan over the hash ENTRIES: local unique id, occurrence prefix, hot-list positions (entry order = unique order)
  const bool hots = hot.n_tasks != nullptr;
  int es[kP3Ent], ec[kP3Ent];
  int v5[5] = {0, 0, 0, 0, 0}, tot5[5];
#pragma unroll
  for (int k = 0; k < kP3Ent; ++k) {
    const int e = threadIdx.x * kP3Ent + k;
    es[k] = h_slot[e];
    ec[k] = es[k] != -1 ? h_cnt[e] : 0;
    if (es[k] == -1) continue;
    ++v5[0];
    v5[1] += ec[k];
    if (hots && ec[k] > hot.khot && ec[k] <= hot.kwave) ++v5[4];
    else if (hots && ec[k] > hot.khot) { ++v5[2]; v5[3] += (ec[k] + hot.kchunk - 1) / hot.kchunk; }
  }
  unsigned long long* tb = a.tstat + a.P;
  // the partition's sums go out as early as they are known -- from inside the scan, by the lane that computes the block totals:
  // the successors' look-back waits for them
  block_scan5<kP3Threads>(v5, tot5, [&](const int (&t5)[5]) {
    stat_store(a.tstat + p, kStatAgg | ((unsigned long long)t5[0] << 31) | (unsigned)t5[1]);
    stat_store(tb + p, kStatAgg | ((unsigned long long)t5[2] << 40) | ((unsigned long long)t5[3] << 20) | (unsigned long long)t5[4]);
  });
  const int nu = tot5[0], tot2 = tot5[1], th = tot5[2], tt = tot5[3], tw = tot5[4];
  {
    int lid = v5[0], pre = v5[1];
#pragma unroll
    for (int k = 0; k < kP3Ent; ++k) {
      if (es[k] == -1) continue;
      h_pl[threadIdx.x * kP3Ent + k] = pre;
      h_lid[threadIdx.x * kP3Ent + k] = (unsigned short)lid;
      ++lid; pre += ec[k];
    }
  }
  QST(5);
  unsigned long long pre_a = 0, pre_b = 0;
  lookback_sum2_1024(a.tstat, tb, p, pre_a, pre_b);     // (its barriers also publish h_pl to the block)
  QST(6);
  const int upre = (int)(pre_a >> 31), spre = (int)(pre_a & 0x7fffffffull);
  // ---- outputs per unique row: by the thread that owns the entry (consecutive entries -> consecutive unique ids)
  {
    int h_ex = v5[2] + (int)(pre_b >> 40), t_ex = v5[3] + (int)((pre_b >> 20) & 0xfffff), w_ex = v5[4] + (int)(pre_b & 0xfffff);
    int uid = upre + v5[0], pv = spre + v5[1];
#pragma unroll
    for (int k = 0; k < kP3Ent; ++k) {
      const int gs = es[k];
      if (gs == -1) continue;
      const int c = ec[k];
      o.csr_cnt[uid] = c;
      if (o.freq) o.freq[uid] = c;
      o.row_addr[uid] = gs < a.S ? tp0 + ((int64_t)gs - s0) * rowb : 0;
      if (o.table_ids) o.table_ids[uid] = tbl;
      o.slots[uid] = gs < a.S ? (int64_t)gs - s0 : -1;
      ptr[uid] = pv;
      if (hots && c > hot.khot && c <= hot.kwave) {
        const int w = w_ex++;
        if (w < hot.max_hot) { hot.wave_u[w] = uid; hot.wave_lo[w] = pv; hot.wave_cnt[w] = c; }
      } else if (hots && c > hot.khot) {
        const int nch = (c + hot.kchunk - 1) / hot.kchunk;
        const int h = h_ex++, t0 = t_ex;
        t_ex += nch;
        if (h < hot.max_hot && t0 + nch <= hot.max_tasks) {
          hot.hot_done[h] = 0;
          hot.hot_nchunks[h] = nch;
          hot.hot_u[h] = uid;
          hot.hot_lo[h] = pv;
          hot.hot_cnt[h] = c;
          hot.hot_t0[h] = t0;
          for (int cc = 0; cc < nch; ++cc) {
            hot.task_u[t0 + cc] = uid;
            hot.task_h[t0 + cc] = h;
            hot.task_lo[t0 + cc] = pv + cc * hot.kchunk;
            const int hi2 = pv + (cc + 1) * hot.kchunk;
            hot.task_hi[t0 + cc] = hi2 < pv + c ? hi2 : pv + c;
          }
          if (nch > 1)
            for (int e2 = 0; e2 < hot.dim; ++e2) hot.hot_acc[(int64_t)h * hot.dim + e2] = 0.f;
        }
      }
      ++uid; pv += c;
    }
  }
  QST(7);
  // ---- outputs per record: unique id / rank base / CSR position (lazy reverse indices), the unique row's key, the CSR entries
#pragma unroll
  for (int k = 0; k < kP3Items; ++k) {
    if (!live[k]) continue;
    const int idx = threadIdx.x + k * kP3Threads;
    const int uid = upre + (int)h_lid[en[k]];
    const int pos = spre + h_pl[en[k]] + bs[k];
    const int cn = (int)rc[k].w, br = (int)rc[k].y;
    a.rec_out4[(int64_t)p * kPartCap + idx] = make_int4((int)rc[k].z < 0 ? ~uid : uid, bs[k], pos, 0);
    if (mine[k]) o.unique_keys[uid] = ky[k];
    if (cn == 1) csr_src[pos] = br;
    else if (cn <= 8) {
      for (int j = 0; j < cn; ++j) csr_src[pos + j] = ~(br + j);
    } else {   // a long list (a hot key's occurrences in one tile): expanded by a whole wave below, not by this thread alone
      const int q = atomicAdd(&s_nbig, 1);
      if (q < kBigMax) { b_pos[q] = pos; b_ref[q] = br; b_cnt[q] = cn; }
      else for (int j = 0; j < cn; ++j) csr_src[pos + j] = ~(br + j);
    }
  }
  __syncthreads();
  {
    const int nbig = s_nbig < kBigMax ? s_nbig : kBigMax;
    for (int q = threadIdx.x >> 6; q < nbig; q += kP3Threads >> 6) {
      const int pos = b_pos[q], br = b_ref[q], cn = b_cnt[q];
      for (int j = lane_id(); j < cn; j += 64) csr_src[pos + j] = ~(br + j);
    }
  }
  // unique rows in front of the partition's table (the first partition of every table; partitions are table-major)
  if (first_of_table && threadIdx.x == 0)
    o.table_offsets[tbl] = __hip_atomic_load(&a.hdr[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 0 : upre;
  if (p == (int)a.P - 1 && threadIdx.x == 0) {
    int U = upre + nu;
    const int O = spre + tot2;
    int nh = (int)(pre_b >> 40) + th, ntk = (int)((pre_b >> 20) & 0xfffff) + tt, nwv = (int)(pre_b & 0xfffff) + tw;
    // A record list overflowed in the probe kernel (sticky flag, a key stream that defeats the hash): the CSR of this step lacks
    // occurrences, so NO row may be updated from it -- the step reports zero unique rows, its backward does nothing, and the
    // module raises at its next check (the pooled output of the forward is complete: it does not depend on the records).
    if (__hip_atomic_load(&a.hdr[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { U = 0; nh = 0; ntk = 0; nwv = 0; }
    if (hots) { *hot.n_hot = nh; *hot.n_tasks = ntk; *hot.n_wave = nwv; }
    o.table_offsets[a.T] = U;
    *o.total = O;
    if (U) ptr[U] = O;
  }
  QST(8);
  QST(9);
}


// pooled gather of path (c): per-occurrence addresses with late rows (address word 1 -> the slot in the key's record)
#ifndef LATE_UNR
#define LATE_UNR 2      // rows of a bag loaded per round.  Round 6 (profiles/r06_gather_variants.txt): 2 beats 4 by 1.6-1.7 us at C2 (bags of 1..10
                        // rows: a round of 4 points up to 3 loads at the zero row, and those requests are what the kernel is short of); 1 and 3
                        // lose, 8 loses; exec-masked loads instead of zero-row loads lose at every width
#endif
#ifndef LATE_KIT
#define LATE_KIT 4      // consecutive bags per lane group
#endif
template <int SDT, int DDT>
__global__ void __launch_bounds__(256) gather_pooled_late_kernel(PoolArgs g, LateRefs late, int lpr_log2) {
#if MI355_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 16384) { g_st_fgather[blockIdx.x * 4] = __builtin_amdgcn_s_memtime(); g_st_fgather[blockIdx.x * 4 + 2] = wall_clock64(); }
#endif
  const int64_t sg = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 >> lpr_log2) + (lane_id() >> lpr_log2);
  gather_pooled_pipe<SDT, DDT, 3, LATE_UNR, LATE_KIT>(g, late, lpr_log2, sg);
#if MI355_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 16384) { g_st_fgather[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime(); g_st_fgather[blockIdx.x * 4 + 3] = wall_clock64(); }
#endif
}

// A wave copies the 64 rows whose addresses its lanes hold (lane l: row i0 + l; 0 = no row -> zeros): 64 / LPR rows per step,
// the addresses handed round with shuffles, UN steps' loads issued before their stores -- a wave keeps UN x 64 / LPR rows in flight
// (one row per lane group and step left the copy latency bound: 2 K waves x 2 rows x 512 B = 2 MB in flight against the ~12 MB
// that 8 TB/s x 1.5 us asks for).  Rows of one column group (D <= 4 LPR), 16-byte aligned; every load is unconditional (lanes
// without a row read the zero row, see gather_dev.h).
template <int SDT, int DDT>
__device__ __forceinline__ void wave_copy_rows(uintptr_t rp, int64_t i0, int64_t n, int D, void* dst, int64_t dst_stride, int lpr_log2) {
  constexpr int UN = 8;
  constexpr int EB = SDT == kF32 ? 4 : 2;
  const gptr_t zero = (gptr_t)(uintptr_t)g_zero_row;
  const int lane = lane_id();
  const int LPR = 1 << lpr_log2, R = 64 >> lpr_log2;
  const int sub = lane >> lpr_log2, c = lane & (LPR - 1);
  const int rlo = (int)(rp & 0xffffffffu), rhi = (int)(rp >> 32);
  const bool col = 4 * c < D;
  for (int q0 = 0; q0 < 64; q0 += R * UN) {
    float4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int src = (q0 + u * R + sub) & 63;
      const uintptr_t ad = (uintptr_t)(unsigned)__shfl(rlo, src, 64) | ((uintptr_t)(unsigned)__shfl(rhi, src, 64) << 32);
      const gptr_t p = (ad != 0 && col) ? (gptr_t)(ad + (uintptr_t)(4 * c * EB)) : zero;
      v[u] = ld4g<SDT>(p);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int src = q0 + u * R + sub;
      const int64_t i = i0 + src;
      if (src < 64 && i < n && col) st4<DDT>(dst, i * dst_stride + 4 * c, v[u]);
    }
  }
}

// sequence gather of path (c): out[j, :D] = the row of occurrence j, late rows through the key's record.  A wave owns 64
// consecutive occurrences: one coalesced load of their address words, then wave_copy_rows.
template <int SDT, int DDT>
__global__ void __launch_bounds__(256)
gather_rows_late_kernel(const int64_t* __restrict__ occ_addr, LateRefs late, int64_t n, int D, void* dst, int64_t dst_stride, int lpr_log2) {
  const int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
  if (i0 >= n) return;
  const int64_t j = i0 + lane_id();
  uintptr_t rp = j < n ? (uintptr_t)occ_addr[j] : 0;
  if (__ballot(rp == 1)) { if (rp == 1) rp = late_row<false>(late, j); }
  wave_copy_rows<SDT, DDT>(rp, i0, n, D, dst, dst_stride, lpr_log2);
}

// eval / inference forward in one launch (gather_dev.h: gather_pooled_eval)
template <int SDT, int DDT, bool kMT>
__global__ void __launch_bounds__(256) gather_pooled_eval_kernel(PoolArgs g, ProbeRefs pr, int lpr_log2) {
  __shared__ EvalTabs tabs;
  if constexpr (kMT) eval_tabs_load(tabs, pr, g.offsets, g.B, g.n);
  const int64_t sg = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 >> lpr_log2) + (lane_id() >> lpr_log2);
  gather_pooled_eval<SDT, DDT, 4, 4, kMT>(g, pr, lpr_log2, sg, &tabs);
}

// eval / inference forward of SEQUENCE lookups in one launch (round 4): lane l of a wave probes key i0 + l (64 independent
// three-hop chains per wave), then the wave copies the 64 rows, 64 / LPR at a time, the addresses handed round with shuffles;
// unknown keys give zero rows (batched_dynamicemb_tables.py:1140-1218).  16-byte aligned rows; the conditions of the pooled form.
template <int SDT, int DDT, bool kMT>
__global__ void __launch_bounds__(256)
gather_rows_eval_kernel(ProbeRefs pr, const int64_t* __restrict__ offsets, int B, int64_t n, int D, void* dst, int64_t dst_stride, int lpr_log2) {
  __shared__ EvalTabs tabs;
  if constexpr (kMT) eval_tabs_load(tabs, pr, offsets, B, n);
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
  const int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
  if (i0 >= n) return;
  const int64_t j = i0 + lane_id();
  const int64_t jc = j < n ? j : n - 1;
  const uint64_t key = pr.keys[jc];
  uintptr_t rp;
  if constexpr (kMT) rp = eval_probe_key(pr, eval_tab_of(tabs, pr.T, jc), key, j < n, C, cshift);
  else rp = eval_probe_key(pr, one, key, j < n, C, cshift);
  wave_copy_rows<SDT, DDT>(rp, i0, n, D, dst, dst_stride, lpr_log2);
}

}  // namespace mi355
#include "part3_lean.h"
namespace mi355 {

// lazy per-occurrence outputs of path (c): reverse index and full rank of every occurrence from its record, late row addresses
__global__ void __launch_bounds__(256)
occ_from_records_kernel(const int32_t* __restrict__ occ_slot, const int32_t* __restrict__ occ_trank, const int4* __restrict__ rec_out4,
                        const int64_t* __restrict__ row_addr, int64_t n, int64_t* __restrict__ rev, int32_t* __restrict__ rank,
                        int64_t* __restrict__ occ_addr, const int* __restrict__ rerun_mark) {
  if (*rerun_mark == 1) return;   // the step overflowed and was re-run on the per-slot-counter path: its outputs are already eager
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const int ref = occ_slot[j];
    const int4 ro = rec_out4[ref >= 0 ? ref : 0];
    const bool late = ro.x < 0;
    const int uid = late ? ~ro.x : ro.x;
    // an occurrence the probe could not record (its slot range's list overflowed: sticky flag, the step reports no uniques)
    // belongs to no unique row: -1, not the id of an unrelated one
    rev[j] = ref >= 0 ? uid : -1;
    if (ref >= 0) {
      rank[j] = occ_trank[j] + ro.y;
      if (late) occ_addr[j] = row_addr[uid];
    } else {
      rank[j] = 0;
    }
  }
}

// exclusive scan of the per-tile representative counts when there are too many tiles for every block to sum its
// predecessors itself (batches beyond 4 M keys)
__global__ void __launch_bounds__(kScanThreads) fused_scan_partials_kernel(int* partial, int64_t nb) {
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int64_t b0 = 0; b0 < nb; b0 += kScanThreads) {
    const int64_t b = b0 + threadIdx.x;
    const int v = b < nb ? partial[b] : 0;
    int tot;
    const int ex = block_excl_scan(v, tot);
    const int carry = s_carry;
    if (b < nb) partial[b] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + tot;
    __syncthreads();
  }
}

}  // namespace mi355

using namespace mi355;
STAMP_EXPORT(mi355_debug_stamps_probe, g_st_probe)
STAMP_EXPORT(mi355_debug_stamps_part, g_st_part)
STAMP_EXPORT(mi355_debug_stamps_evict, g_st_evict)
STAMP_EXPORT(mi355_debug_stamps_fgather, g_st_fgather)

// Side stream on which the forward numbers the uniques and builds the backward's CSR while its own gather runs.
// One per process; the join events form a small ring, a forward hands its token to the backward of the same batch
// (the backward of a CUDA autograd node runs on another host thread than the forward: no thread-local state here).
namespace {
struct SideCsr {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr;
  hipEvent_t join[8] = {};
  unsigned next = 0;
  int last = -1;        // token of the most recent fork
  bool joined = true;   // the most recent side work has been waited for on the main stream
  bool ok = false;
  pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
};
SideCsr g_side;
bool side_init() {
  if (g_side.ok) return true;
  if (hipStreamCreateWithFlags(&g_side.side, hipStreamNonBlocking) != hipSuccess) return false;
  if (hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming) != hipSuccess) return false;
  for (auto& e : g_side.join)
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
  g_side.ok = true;
  return true;
}
}  // namespace

extern "C" {

struct GatherTimer {   // bench.py's live timing of the gather launch (err.hip: mi355_profile_kernels)
  hipStream_t s;
  explicit GatherTimer(hipStream_t st) : s(st) { mi355i_prof_mark(0, 0, s); }
  ~GatherTimer() { mi355i_prof_mark(0, 1, s); }
};

// (exported for callers with a stream-aware allocator, and used by pipeline.hip)
void* mi355_early_csr_stream(void) {
  pthread_mutex_lock(&g_side.mu);
  const bool ok = side_init();
  pthread_mutex_unlock(&g_side.mu);
  return ok ? (void*)g_side.side : nullptr;
}

// fork: side waits for everything issued on `stream`; returns the side stream (nullptr on failure)
hipStream_t mi355i_side_fork(hipStream_t stream) {
  pthread_mutex_lock(&g_side.mu);
  hipStream_t r = nullptr;
  if (side_init() && hipEventRecord(g_side.fork, stream) == hipSuccess &&
      hipStreamWaitEvent(g_side.side, g_side.fork, 0) == hipSuccess)
    r = g_side.side;
  pthread_mutex_unlock(&g_side.mu);
  return r;
}
// record the join point of the work issued on the side stream since the fork; returns its token (< 0: failure)
int mi355i_side_mark(void) {
  pthread_mutex_lock(&g_side.mu);
  int tok = (int)(g_side.next++ % 8);
  if (hipEventRecord(g_side.join[tok], g_side.side) != hipSuccess) tok = -1;
  else { g_side.last = tok; g_side.joined = false; }
  pthread_mutex_unlock(&g_side.mu);
  return tok;
}
// `stream` waits for the join point `token` (the side stream is in order, so a re-used event still covers the work)
int mi355i_side_join(int token, hipStream_t stream) {
  pthread_mutex_lock(&g_side.mu);
  int rc = MI355_OK;
  if (!g_side.ok || token < 0 || token >= 8 || hipStreamWaitEvent(stream, g_side.join[token], 0) != hipSuccess) rc = MI355_ELAUNCH;
  else if (token == g_side.last) g_side.joined = true;
  pthread_mutex_unlock(&g_side.mu);
  return rc;
}
// a new forward must not start before the previous side work (which clears the slot counters) is done
int mi355i_side_join_pending(hipStream_t stream) {
  pthread_mutex_lock(&g_side.mu);
  int rc = MI355_OK;
  if (g_side.ok && !g_side.joined && g_side.last >= 0) {
    if (hipStreamWaitEvent(stream, g_side.join[g_side.last], 0) != hipSuccess) rc = MI355_ELAUNCH;
    else g_side.joined = true;
  }
  pthread_mutex_unlock(&g_side.mu);
  return rc;
}

// public form of the join (a caller that reads the unique numbering of a fused forward before its backward)
int mi355_side_join(int token, hipStream_t stream) {
  const int rc = mi355i_side_join(token, stream);
  if (rc != MI355_OK) mi355_set_error("side-stream join failed");
  return rc;
}

// ---- round 6: the overflow notice of path (c).  64 pinned, host-coherent words {epoch (low 32), flooded (bit 32)}: slot epoch % 64
// is written by the partition kernel of the step with that epoch (publish_notice) and read by the host when the step's
// backward is issued.  Epochs are process-wide and never zero.
static constexpr int kNoticeRing = 64;
static unsigned long long* notice_ring() {
  static unsigned long long* ring = [] {
    unsigned long long* r = nullptr;
    if (hipHostMalloc((void**)&r, kNoticeRing * sizeof(unsigned long long), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess)
      return (unsigned long long*)nullptr;
    for (int i = 0; i < kNoticeRing; ++i) r[i] = 0;
    return r;
  }();
  return ring;
}
static int g_epoch = 0;
// mi355_demb_forward_fused_rerun: the epoch of the step being regrouped (0: an ordinary forward)
static thread_local int t_rerun_epoch = 0;
// stages of a path-(c) forward (mi355_demb_plan_stage): 0 the whole forward, 1 the index stage only (probe + partition kernel:
// everything the backward needs and the per-occurrence row addresses), 2 the gather only (of a step whose stage 1 ran earlier,
// typically on another stream).  t_protect: FusedArgs::protect of the call.
static thread_local int t_stage = 0;
static thread_local uint64_t t_protect = ~0ull;
void mi355i_fused_stage(int stage, uint64_t protect) { t_stage = stage; t_protect = protect; }

// Has the forward with this epoch flooded a partition's record list?  0: no (its CSR is complete), 1: yes (re-run its index stage
// with mi355_demb_forward_fused_rerun / mi355_demb_plan_rerun before its backward), -1: not known within wait_ms milliseconds
// (the forward has not reached its partition kernel: a stuck GPU), 2: the notice was overwritten by a step 64 epochs later --
// treat as flooded (a re-run of a clean step is harmless), 3: a gather block of the step abandoned its bounded wait for a partition
// block of the same launch (sequence lookups; gather_dev.h: late_row) -- the step's OUTPUT lacks rows: an error.  The wait spins on pinned memory; with a dense model between forward
// and backward the word has long been written.
int mi355_demb_fused_step_flooded(int epoch, int wait_ms) {
  if (epoch <= 0) return 0;
  volatile unsigned long long* slot = notice_ring();
  if (!slot) return 2;
  slot += epoch % kNoticeRing;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; ++spin) {
    const unsigned long long v = *slot;
    const int e = (int)(unsigned)(v & 0xffffffffull);
    if (e == epoch) return ((v >> 33) & 1ull) ? 3 : (int)((v >> 32) & 1ull);   // 3: a gather block gave up waiting for its partition block
    if (e > epoch) return 2;   // a later epoch owns the slot (64 forwards were issued before this step's backward)
    if ((spin & 1023) == 1023 &&
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > wait_ms)
      return -1;
    __builtin_ia32_pause();
  }
}

int64_t mi355_demb_aux_numel(int64_t total_slots, int64_t num_buckets) {
  return kAuxHdr + 2 * (total_slots + 1) + num_buckets;
}

static inline int64_t al256(int64_t x) { return (x + 255) / 256 * 256; }

// partitions of the partitioned index stage for a batch of n keys (0: the batch takes the per-slot-counter path)
static inline int part_count(int64_t n, int64_t num_tables) {
  static const int env = getenv("MI355_FUSED_PART") ? atoi(getenv("MI355_FUSED_PART")) : 2;
  // several tables (round 4): path (c) only, with table-aligned partitions -- every table owns at least one, so the batch needs
  // a few per table (MI355_FUSED_MT=0: multi-table batches keep the per-slot-counter path)
  constexpr int mt_env = 1;
  if (!env || n < (64 << 10)) return 0;
  if (n > (int64_t)kPartMax * 1024) return 0;
  if (num_tables != 1 && (!mt_env || env < 2 || num_tables < 1 || num_tables > kFusedMaxT)) return 0;
  // keys per partition (MI355_FUSED_KPP, default 1024): the partition kernel is one block per partition and a chain of dependent
  // phases -- below 256 partitions it leaves CUs idle, so small batches may as well get thinner partitions
  constexpr int kpp_env = 1024;
  const int kpp = kpp_env >= 256 && kpp_env <= 1024 ? kpp_env : 1024;
  int P = (int)((n + 1023) / 1024);
  if (kpp < 1024 && P < 256) { P = (int)((n + kpp - 1) / kpp); if (P > 256) P = 256; }
  if (num_tables > 1 && P < 4 * num_tables) return 0;
  // the partition kernel of path (c) is one 1024-thread block per partition and per CU: up to 1.5 K keys per partition (about
  // 1.1 K records of the 2 K a partition can hold) the batch gets exactly one block per CU instead of a second, thin generation
  if (env >= 2 && P > 256 && n <= 256 * 1536) P = 256;
  return P;
}

// slot-range partitions the fused forward would use for this batch / table (0: the per-slot-counter path)
int mi355_demb_forward_fused_partitions(int64_t n, int64_t num_tables, int64_t num_buckets) {
  const int P = part_count(n, num_tables);
  return (P > 0 && num_buckets >= 8 * (int64_t)P) ? P : 0;
}

int64_t mi355_demb_forward_fused_workspace_bytes(int64_t n, int64_t num_tables) {
  const int64_t nt = (n + 1023) / 1024 + 2;
  return al256(8 * (num_tables + 1)) + al256(8 * n) /*unique keys*/ + al256(8 * n) /*occ_addr*/ + al256(4 * n) /*occ_slot*/ +
         al256(4 * nt) + 256 /*total*/ + al256(8 * n) + 4 * al256(4 * n) /*deferred*/ + al256(24 * (nt > 258 ? nt : 258)) /*look-back: two words per 1024 keys, and per partition*/ + 256 +
         (part_count(n, num_tables) ? 32 * (int64_t)part_count(n, num_tables) * kPartCap + 5 * 256 + 4 * kPartMaxBig : 0) /*partition records, table of every partition*/;
}

// The fused forward (see the header of this file).  Persisted outputs as mi355_demb_forward; in eval mode (train == 0)
// only `out` is produced (no unique numbering).  `join_token` (train): >= 0 when the CSR was built on the side stream --
// hand it to mi355_demb_backward(prepared = 2 + token); -1: built on `stream`.
int mi355_demb_forward_fused(
    /* table */ void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity, int64_t num_scores,
    int32_t* bucket_sizes, int32_t* counter, int64_t counter_numel, int32_t* aux, int64_t aux_numel, int64_t num_buckets,
    /* values */ const int64_t* table_ptrs, const int64_t* table_value_dims, const int64_t* table_emb_dims,
    int value_dtype, int64_t emb_dim, int64_t value_dim,
    /* batch */ const void* keys, int64_t num_keys, const int64_t* offsets, int64_t num_bags, int64_t batch_size,
    const int64_t* feature_offsets, int64_t num_tables,
    /* policies */ int train, int find_policy, int insert_policy, uint64_t score_value, int use_count,
    uint64_t timer_override, int pin,
    /* initializer */ int init_mode, float p0, float p1, float p2, float p3, uint64_t seed, float state_init,
    /* output */ int combiner, const int32_t* D_offsets, int64_t total_D, void* out, int out_dtype, int aligned16,
    /* persisted */ int64_t* reverse_indices, int64_t* unique_offsets, int64_t* table_ids, int64_t* slots,
    int64_t* row_addr, int64_t* freq, int32_t* csr_cnt, int32_t* csr_rank,
    /* CSR of the backward */ void* backward_workspace, int64_t backward_workspace_bytes, int use_side_stream,
    int* join_token,
    /* scratch */ void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(workspace && workspace_bytes >= mi355_demb_forward_fused_workspace_bytes(num_keys, num_tables), "workspace too small");
  MI355_CHECK_ARG(num_tables >= 1 && num_tables <= kFusedMaxT, "fused forward: too many tables");
  MI355_CHECK_ARG(bucket_capacity > 0 && bucket_capacity % 16 == 0, "bucket capacity must be a positive multiple of 16");
  const int64_t S = num_buckets * bucket_capacity;
  MI355_CHECK_ARG(S < 0x7fffff00LL && num_keys < 0x7fffff00LL, "fused forward: table / batch too large for 32-bit slots");
  MI355_CHECK_ARG(aux && aux_numel >= mi355_demb_aux_numel(S, num_buckets), "aux buffer too small");
  MI355_CHECK_ARG(find_policy >= kConst && find_policy <= kLruLfu && insert_policy >= kConst && insert_policy <= kLruLfu, "bad score policy");
  MI355_CHECK_ARG((find_policy != kLruLfu && insert_policy != kLruLfu) || num_scores == 2, "LRU_LFU needs num_scores == 2");
  if (join_token) *join_token = -1;
  if (num_keys == 0 && combiner < 0) return MI355_OK;
  int rc;
#define STEP(call) do { rc = (call); if (rc != MI355_OK) return rc; } while (0)
  STEP(mi355i_side_join_pending(stream));
  uint8_t* w = (uint8_t*)workspace;
  const int64_t n = num_keys;
  const int64_t nt = (n + 1023) / 1024 + 2;
  FusedArgs a;
  a.t = make_table(storage, bucket_capacity, num_scores);
  a.tbo = table_bucket_offsets; a.bucket_sizes = bucket_sizes; a.counter = counter;
  a.hdr = aux; a.occ = aux + kAuxHdr; a.locks = a.occ + 2 * (S + 1); a.S = S;
  a.table_ptrs = table_ptrs; a.table_value_dims = table_value_dims; a.table_emb_dims = table_emb_dims;
  a.elem_bytes = value_dtype == 0 ? 4 : 2; a.value_dtype = value_dtype;
  a.keys = (const uint64_t*)keys; a.n = n; a.offsets = offsets; a.feature_offsets = feature_offsets; a.num_bags = num_bags; a.batch = batch_size;
  a.T = (int)num_tables; a.find_policy = find_policy; a.insert_policy = insert_policy; a.use_count = use_count;
  a.score_value = score_value; a.timer = timer_override;
  constexpr int dbg_env = 0;
  a.dbg = dbg_env;
  a.init = InitArgs{init_mode, p0, p1, p2, p3, seed, state_init};
  a.seg_out = (int64_t*)w; w += al256(8 * (num_tables + 1));
  uint64_t* unique_keys = (uint64_t*)w; w += al256(8 * n);
  a.occ_addr = (int64_t*)w; w += al256(8 * n);
  a.occ_slot = (int32_t*)w; w += al256(4 * n);
  a.partial = (int32_t*)w; w += al256(4 * nt);
  int32_t* total = (int32_t*)w; w += 256;
  a.d_key = (uint64_t*)w; w += al256(8 * n);
  a.d_tid = (int32_t*)w; w += al256(4 * n);
  a.d_cnt = (int32_t*)w; w += al256(4 * n);
  a.d_slot = (int32_t*)w; w += al256(4 * n);
  a.d_base = (int32_t*)w; w += al256(4 * n);
  a.tstat = (unsigned long long*)w; w += al256(24 * (nt > 258 ? nt : 258));
  // partitioned index stage: one table, a batch of 64 K .. 1 M keys, at least 8 buckets per partition, training
  a.P = 0; a.spp = 1; a.pcount = aux + 64;
  a.rec = nullptr; a.rec_out = nullptr; a.rec_out4 = nullptr;
  a.tile_bags = nullptr; a.occ_trank = nullptr;
  a.mt = 0; a.ptab = nullptr;
  a.gate = nullptr; a.gate_val = 0; a.ovf_word = 5; a.ovf_val = 1; a.rerun_mark = nullptr; a.tl = 0;
  a.protect = t_protect;
  a.cap = kPartCap; a.part_ready = nullptr;
  a.magic0 = num_buckets > 0 ? ~0ull / (uint64_t)num_buckets : 0ull;   // (one table: its buckets are all the buckets)
  {
    const int P = train ? part_count(n, num_tables) : 0;
    if (P > 0 && num_buckets >= 8 * (int64_t)P) {
      const int64_t per = ((S + 1 + P - 1) / P + bucket_capacity - 1) / bucket_capacity * bucket_capacity;
      if (per < 0x7fffffffLL) {
        a.P = P; a.spp = (int)per;
        a.cap = kPartCap;
        const int64_t nr = (int64_t)P * a.cap;
        a.rec = (uint4*)w; w += al256(16 * nr);
        a.rec_out = (int2*)w; a.rec_out4 = (int4*)w; w += al256(16 * nr);   // (path (a): 8-byte entries, path (c): 16-byte ones)
        a.ptab = (int32_t*)w; w += 4 * kPartMaxBig;
        a.mt = num_tables > 1;
      }
    }
  }
  bool part = a.P > 0;
  a.csr_rank = csr_rank;
  // backward workspace (row pointers, CSR, hot lists) -- carved before the probe launch, which clears the hot-list header
  int32_t* bptr = nullptr; int32_t* bcsr = nullptr; void* hot_ws = nullptr; int64_t hot_bytes_ = 0;
  a.hot_counters = nullptr;
  if (train && backward_workspace) {
    MI355_CHECK_ARG(backward_workspace_bytes >= mi355_demb_backward_workspace_bytes(n, emb_dim), "backward workspace too small");
    uint8_t* bw = (uint8_t*)backward_workspace;
    bptr = (int32_t*)bw; bw += al256(4 * (n + 1));
    bcsr = (int32_t*)bw; bw += al256(4 * n);
    bw += mi355_group_by_unique_workspace_bytes(n, n);
    hot_ws = bw; hot_bytes_ = mi355_backward_workspace_bytes(n, emb_dim);
    a.hot_counters = (int*)hot_ws;
  }
  if (train) MI355_CHECK_ARG(reverse_indices && unique_offsets && slots && row_addr && csr_cnt && csr_rank, "persisted outputs required in train mode");
  // ---- path (c): partition blocks write the CSR and ride in the gather's launch (pooled training forward of one-column-group
  //      rows with short bags; MI355_FUSED_PART=1 keeps round 2's probe / partition / scatter / gather chain)
  static const int part_env = getenv("MI355_FUSED_PART") ? atoi(getenv("MI355_FUSED_PART")) : 2;
  int lg = 3;
  while ((4 << lg) < emb_dim && lg < 6) ++lg;
  // sequence lookups (combiner -1, round 4; MI355_FUSED_SEQ=0 keeps them on the probe / partition / scatter chain): occurrence j
  // is its own bag
  constexpr int seq_env = 1;
  const bool seq = combiner == -1;
  bool pathc = part && part_env >= 2 && train && (combiner >= 0 || (seq && seq_env)) && hot_ws && bcsr && aligned16 &&
                     emb_dim <= (4 << lg) && (seq || n <= 8 * num_bags) && value_dtype <= 1 && out_dtype <= 1 &&
                     num_bags < (1ll << 31) - 4096;
  // round 6: the partitioned stage is path (c) or nothing.  Round 2's form (a) -- the same record lists, merged by fused_part_kernel
  // and scattered by a third kernel -- reported a flooded list through a sticky flag and skipped the step's update; what is not
  // eligible for path (c) (long bags, fp16 rows, unaligned rows, MI355_FUSED_PART=1) takes the per-slot counters, which cannot flood
  if (part && !pathc) { part = false; a.P = 0; a.mt = 0; }
  // MI355_PROBE_C (a TEST hook: the tile shape of the probe kernel -- 1 the run-time rule, 2 two keys per thread, 3 full
  // 1 024-key tiles); MI355_ENV_LIVE=1 (the test suite) re-reads it on every call, otherwise once: getenv walks the environment
  static const bool env_live = getenv("MI355_ENV_LIVE") != nullptr;
  static int pc_c = 1;
  static bool env_have = false;
  if (env_live || !env_have) {
    const char* e2 = getenv("MI355_PROBE_C");
    pc_c = e2 ? atoi(e2) : 1;
    if (pc_c < 1 || pc_c > 3) pc_c = 1;
    env_have = true;
  }
  // division-free bucket arithmetic (multiply-high by a per-table magic): power-of-two bucket capacities
  const bool fast = part && pathc && (a.t.C & (a.t.C - 1)) == 0 && (a.S >> __builtin_ctzll((unsigned long long)a.t.C)) < (1ll << 31);
  const int pcv = pc_c;
  if (!part) pathc = false;
  // round 5: the partition blocks ride in the gather's launch (part3_lean.h) for SEQUENCE lookups (8 x 16 K tokens 0.076 -> 0.068 ms);
  // pooled batches keep the partition kernel of its own (the same role in front of the pooled gather's blocks measured a loss at
  // C2: 0.1227 -> 0.1288 ms, profiles/r05_part_fused.txt)
  const bool part_fused = pathc && t_stage == 0 && seq;   // (staged forwards: two launches)
  if (part_fused) a.part_ready = (int32_t*)(a.tstat + 2 * a.P);
  // opt-in: an overflowed step is re-run on the per-slot-counter path inside this call (the reference never skips an update,
  // unique_op.cu:484-714): three more launches behind the gather that return at once in the steady state
  // (round 5: the chain on the SIDE stream, forked in front of the gather -- its head kernel holding it back in a flagged step until
  //  every wave of the gather's launch had counted itself done -- was built, passed the flood tests, and cost 0.252 ms per C2 step
  //  against 0.169 in line and 0.162 without: the fork / join event pair across two queues is far dearer than three empty launches)
  // round 6: no update is ever lost, and the steady state pays nothing for it.  Default = the NOTICE: the partition kernel's
  // first block stores {epoch, flooded} into pinned memory, the host reads it when the step's backward is issued and re-runs the
  // index stage on the per-slot-counter path first if it has to (mi355_demb_forward_fused_rerun).  Where the host cannot wait
  // for a word -- the forward is being CAPTURED into a graph -- or where the step's pin counters are taken from the numbering
  // (pin != 0), the chain runs IN LINE instead: three launches behind the gather, gated on the epoch, that return at once in the
  // steady state (MI355_FUSED_OVERFLOW_RERUN=1 forces this form: the round-5 behaviour, kept for A/B).
  static const int rerun_env = getenv("MI355_FUSED_OVERFLOW_RERUN") ? atoi(getenv("MI355_FUSED_OVERFLOW_RERUN")) : 0;
  const int rerun_only = t_rerun_epoch;
  const int stage = rerun_only ? 0 : t_stage;
  if (stage && !pathc) return 3;     // (nothing launched) not a path-(c) batch: the caller keeps its one-call forward
  bool capturing = false;
  if (pathc && !rerun_only) {
    hipStreamCaptureStatus cst = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cst) == hipSuccess) capturing = cst != hipStreamCaptureStatusNone;
  }
  unsigned long long* ring = pathc ? notice_ring() : nullptr;
  const bool notice_mode = pathc && !rerun_only && !pin && !capturing && rerun_env != 1 && ring != nullptr;
  const bool rerun = pathc && stage != 2 && (rerun_only || !notice_mode);
  int epoch = 0;
  a.notice = nullptr;
  if (pathc) {
    a.tile_bags = (int32_t*)((uint8_t*)backward_workspace + al256(4 * (n + 1)) + al256(4 * n));   // head of the grouping workspace
    a.occ_trank = a.d_tid;   // (the deferred-key arrays belong to path (b))
    a.rerun_mark = total + 16;   // cleared by the probe kernel on every step: the lazy materialisation reads it
    epoch = rerun_only ? rerun_only : (stage == 2 ? 0 : (int)(__sync_add_and_fetch(&g_epoch, 1) & 0x3fffffff) + 1);
    a.ovf_word = 6; a.ovf_val = epoch;
    if (notice_mode && stage != 2) a.notice = ring + epoch % kNoticeRing;
  } else if (rerun_only) {
    mi355_set_error("mi355_demb_forward_fused_rerun: not a path-(c) step (nothing to re-run)");
    return MI355_EINVAL;
  }
  // ---- eval / inference forward of one table with pooled output: ONE kernel (every lane probes its own keys; no dedup, no
  //      unique numbering, no address array).  MI355_EVAL_FUSED=0 keeps the probe + gather pair.
  constexpr int eval_env = 1;
  if (!train && eval_env && n > 0 && num_tables >= 1 && num_tables <= kEvalMaxT && combiner >= -1 && aligned16 && !use_count &&
      (find_policy == kConst || find_policy == kAssign || find_policy == kGlobalTimer) && (bucket_capacity & (bucket_capacity - 1)) == 0 &&
      value_dtype <= 1 && out_dtype <= 1 && num_buckets < (1ll << 31) && (combiner == -1 || n <= 8 * num_bags)) {
    int le = 3;
    while ((4 << le) < emb_dim && le < 6) ++le;
    constexpr int eval_mt_env = 1;
    const bool mt = num_tables > 1;
    if (emb_dim <= (4 << le) && (eval_mt_env || (!mt && combiner >= 0))) {
      RoctxRange rr("op:eval_lookup+gather_embedding");
      ProbeRefs pr;
      pr.keys = (const uint64_t*)keys; pr.t = a.t; pr.tbo = table_bucket_offsets; pr.table_ptrs = table_ptrs;
      pr.table_value_dims = table_value_dims; pr.elem_bytes = a.elem_bytes; pr.find_policy = find_policy;
      pr.score_value = score_value; pr.timer = timer_override;
      pr.T = (int)num_tables; pr.feature_offsets = feature_offsets;
      GatherTimer gt(stream);
      if (combiner == -1) {
        const unsigned grid = (unsigned)ceil_div(n, 256);
#define LAUNCH_ER(S, D)                                                                                                                      \
  do {                                                                                                                                       \
    if (mt) hipLaunchKernelGGL((gather_rows_eval_kernel<S, D, true>), dim3(grid), dim3(256), 0, stream, pr, offsets, (int)batch_size, n,    \
                               (int)emb_dim, out, emb_dim, le);                                                                              \
    else hipLaunchKernelGGL((gather_rows_eval_kernel<S, D, false>), dim3(grid), dim3(256), 0, stream, pr, offsets, (int)batch_size, n,      \
                            (int)emb_dim, out, emb_dim, le);                                                                                 \
  } while (0)
        if (value_dtype == 0 && out_dtype == 0) LAUNCH_ER(kF32, kF32);
        else if (value_dtype == 0) LAUNCH_ER(kF32, kBF16);
        else if (out_dtype == 0) LAUNCH_ER(kBF16, kF32);
        else LAUNCH_ER(kBF16, kBF16);
#undef LAUNCH_ER
      } else {
        PoolArgs g;
        g.src = nullptr; g.src_stride = 0; g.row_addr = nullptr; g.rev = nullptr; g.offsets = offsets; g.D_offsets = D_offsets;
        g.dst = out; g.FB = num_bags; g.n = n; g.B = (int)batch_size; g.D = (int)emb_dim; g.total_D = (int)total_D; g.combiner = combiner;
        const unsigned grid = (unsigned)grid_for(num_bags, 4 * (64 >> le) * 4, 1 << 20);
#define LAUNCH_EV(S, D)                                                                                                        \
  do {                                                                                                                         \
    if (mt) hipLaunchKernelGGL((gather_pooled_eval_kernel<S, D, true>), dim3(grid), dim3(256), 0, stream, g, pr, le);        \
    else hipLaunchKernelGGL((gather_pooled_eval_kernel<S, D, false>), dim3(grid), dim3(256), 0, stream, g, pr, le);          \
  } while (0)
        if (value_dtype == 0 && out_dtype == 0) LAUNCH_EV(kF32, kF32);
        else if (value_dtype == 0) LAUNCH_EV(kF32, kBF16);
        else if (out_dtype == 0) LAUNCH_EV(kBF16, kF32);
        else LAUNCH_EV(kBF16, kBF16);
#undef LAUNCH_EV
      }
      MI355_LAUNCH_CHECK();
      return MI355_OK;
    }
  }
  if (n > 0 && !rerun_only && stage != 2) {
    RoctxRange rr("op:fused_index(segmented_unique+storage_find+storage_insert+initializer)");
    // keys per tile / threads per block: one key per thread keeps every probe chain (digest vector -> key -> slot counter)
    // in flight at once; larger tiles cost fewer (tile, key) pairs = fewer device-scope atomics
    constexpr int cfg_env = -1;
    // measured at C2 (360 K keys): 2048-key tiles / 1024 threads 33.5 us, 1024 / 1024 37.8 us, 1024 / 512 35.2 us
    const int cfg = cfg_env >= 0 ? cfg_env : (n >= (64 << 10) ? 3 : 0);
#define LAUNCH_PROBE(TILE, THREADS)                                                                                        \
  do {                                                                                                                     \
    const unsigned grid = (unsigned)ceil_div(n, TILE);                                                                     \
    if (train) hipLaunchKernelGGL((fused_probe_kernel<TILE, THREADS, true>), dim3(grid), dim3(THREADS), 0, stream, a);    \
    else hipLaunchKernelGGL((fused_probe_kernel<TILE, THREADS, false>), dim3(grid), dim3(THREADS), 0, stream, a);         \
  } while (0)
    if (part) {
      // (part implies path (c) since round 6)  round 5: the rebuilt probe kernel (probe_c.h) wherever the bucket arithmetic is
      // division-free; other bucket capacities keep the round-3 probe kernel
      if (fast) {
        // ONE block per CU in one generation while the batch allows it: tile length = ceil(n / #CUs), rounded to 64, in the kernel
        // with one key per thread (<= 1 024 keys per tile) or two (<= 2 048); larger batches run full 2 048-key tiles in
        // generations.  MI355_PROBE_C: 1 this rule, 2 always the two-keys-per-thread kernel, 3 full 1 024-key tiles (two blocks
        // per CU: the form measured first, profiles/r05_index_phase_stamps_1024x2.txt), 0 the round-3 kernel.
        static int ncu_p = 0;
        if (!ncu_p) {
          int dev = 0;
          hipDeviceProp_t prop;
          if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu_p = prop.multiProcessorCount;
          if (ncu_p <= 0) ncu_p = 256;
        }
        const int cap = pcv == 3 ? 1024 : ((pcv == 2 || n > (int64_t)ncu_p * 1024) ? 2048 : 1024);
        int64_t tlen = pcv == 3 ? cap : (ceil_div(n, ncu_p) + 63) / 64 * 64;
        if (tlen > cap) tlen = cap;
        if (tlen < 256) tlen = 256;
        a.tl = (int)tlen;
#define LAUNCH_PC(TILE, THREADS, WPS)                                                                                                   \
  do {                                                                                                                                   \
    const dim3 grid((unsigned)ceil_div(n, tlen)), blk(THREADS);                                                                          \
    if (a.mt && seq) hipLaunchKernelGGL((probe_c_kernel<TILE, THREADS, WPS, true, true>), grid, blk, 0, stream, a);                     \
    else if (a.mt) hipLaunchKernelGGL((probe_c_kernel<TILE, THREADS, WPS, true, false>), grid, blk, 0, stream, a);                      \
    else if (seq) hipLaunchKernelGGL((probe_c_kernel<TILE, THREADS, WPS, false, true>), grid, blk, 0, stream, a);                       \
    else hipLaunchKernelGGL((probe_c_kernel<TILE, THREADS, WPS, false, false>), grid, blk, 0, stream, a);                               \
  } while (0)
        if (cap == 2048) LAUNCH_PC(2048, 1024, 4);
        else LAUNCH_PC(1024, 1024, 4);      // (one block per CU by construction: the register budget of four waves per SIMD)
        (void)0;
#undef LAUNCH_PC
      } else {
#define LAUNCH_C(MT, SEQ) hipLaunchKernelGGL((fused_probe_kernel<2048, 1024, true, true, false, true, MT, SEQ>), dim3((unsigned)ceil_div(n, 2048)), dim3(1024), 0, stream, a)
        // (bucket capacities that are not a power of two: the round-3 probe kernel with the generic bucket arithmetic)
        const int v = (a.mt ? 2 : 0) | (seq ? 1 : 0);
        switch (v) {
          case 0: LAUNCH_C(false, false); break;
          case 1: LAUNCH_C(false, true); break;
          case 2: LAUNCH_C(true, false); break;
          default: LAUNCH_C(true, true); break;
        }
#undef LAUNCH_C
      }
    } else if (cfg == 3) LAUNCH_PROBE(2048, 1024);
    else LAUNCH_PROBE(1024, 1024);
#undef LAUNCH_PROBE
    MI355_LAUNCH_CHECK();
  }
  // ---- train: unique numbering + CSR of the backward on the side stream (forked BEFORE the gather is queued, so both
  //      start as soon as the index stage is done); eval: only the gather
  if (pathc) {
    RoctxRange rr("op:unique_numbering+backward_csr");
    EmitOut o;
    o.unique_keys = unique_keys; o.table_offsets = unique_offsets; o.table_ids = table_ids; o.slots = slots;
    o.row_addr = row_addr; o.freq = freq; o.csr_cnt = csr_cnt; o.total = total + 8;
    HotList hot = hot_carve(hot_ws, n, emb_dim);
    PoolArgs g;
    g.src = nullptr; g.src_stride = 0; g.row_addr = a.occ_addr; g.rev = nullptr; g.offsets = offsets; g.D_offsets = D_offsets;
    g.dst = out; g.FB = num_bags; g.n = n; g.B = (int)batch_size; g.D = (int)emb_dim; g.total_D = (int)total_D; g.combiner = combiner;
    LateRefs late; late.occ_slot = a.occ_slot; late.rec = a.rec; late.S = (int)S;
    late.table_ptrs = table_ptrs; late.table_value_dims = table_value_dims; late.tbo = table_bucket_offsets;
    late.C = bucket_capacity; late.elem_bytes = a.elem_bytes; late.T = (int)num_tables;
    const int nsub = 64 >> lg;
    if (rerun_only) goto rerun_chain;      // (a flooded step: its forward ran, only the index stage is redone)
    if (part_fused) { late.ready = a.part_ready; late.cap = kPartCap; late.notice = a.notice; }
    else if (stage == 2) { }               // (the partition kernel ran with the step's index stage)
    else hipLaunchKernelGGL(fused_part3_kernel<kPartCap>, dim3((unsigned)a.P), dim3(kP3Threads), 0, stream, a, o, bptr, bcsr, hot);
    MI355_LAUNCH_CHECK();
    if (stage == 1) goto rerun_chain;      // index stage only: the gather follows in a stage-2 call
    if (seq) {
      RoctxRange rg("op:gather_embedding");
      GatherTimer gt(stream);
      const unsigned grid = (unsigned)ceil_div(n, 256);
#define LAUNCH_RG(S, D)                                                                                                                \
  do {                                                                                                                                 \
    if (part_fused) hipLaunchKernelGGL((gather_rows_part_kernel<S, D>), dim3((grid + 1) / 2 + (unsigned)a.P), dim3(kP3lThreads), 0, stream, a, o, bptr,  \
                                       bcsr, hot, a.occ_addr, late, n, (int)emb_dim, out, emb_dim, lg);                                \
    else hipLaunchKernelGGL((gather_rows_late_kernel<S, D>), dim3(grid), dim3(256), 0, stream, a.occ_addr, late, n, (int)emb_dim, out, \
                            emb_dim, lg);                                                                                              \
  } while (0)
      if (value_dtype == 0 && out_dtype == 0) LAUNCH_RG(kF32, kF32);
      else if (value_dtype == 0) LAUNCH_RG(kF32, kBF16);
      else if (out_dtype == 0) LAUNCH_RG(kBF16, kF32);
      else LAUNCH_RG(kBF16, kBF16);
#undef LAUNCH_RG
    } else {
      RoctxRange rg("op:gather_embedding");
      GatherTimer gt(stream);
      const unsigned grid = (unsigned)grid_for(num_bags, 4 * nsub * LATE_KIT, 1 << 20);
#define LAUNCH_PG(S, D)                                                                                                                \
  do {                                                                                                                                 \
    hipLaunchKernelGGL((gather_pooled_late_kernel<S, D>), dim3(grid), dim3(256), 0, stream, g, late, lg);                              \
  } while (0)
      if (value_dtype == 0 && out_dtype == 0) LAUNCH_PG(kF32, kF32);
      else if (value_dtype == 0) LAUNCH_PG(kF32, kBF16);
      else if (out_dtype == 0) LAUNCH_PG(kBF16, kF32);
      else LAUNCH_PG(kBF16, kBF16);
#undef LAUNCH_PG
    }
    MI355_LAUNCH_CHECK();
  rerun_chain:
    if (rerun) {
      // the chain of the per-slot-counter path over the same buffers, gated on this call's epoch: probe (keys already inserted are
      // found; Assign / timer scores are idempotent, counting ones see the step twice), numbering, CSR scatter (eager reverse
      // indices); the pooled output above is complete either way
      FusedArgs b = a;
      b.P = 0; b.spp = 1; b.rec = nullptr; b.rec_out = nullptr; b.rec_out4 = nullptr; b.tile_bags = nullptr; b.occ_trank = nullptr;
      b.mt = 0; b.ptab = nullptr; b.rerun_mark = nullptr;
      b.notice = nullptr;
      if (!rerun_only) { b.gate = a.hdr + 6; b.gate_val = epoch; }      // in line: gated on this call's epoch; re-run: the host knows
      hipLaunchKernelGGL((fused_probe_kernel<2048, 1024, true>), dim3((unsigned)ceil_div(n, 2048)), dim3(1024), 0, stream, b);
      static int ncu_r = 0;
      if (!ncu_r) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu_r = prop.multiProcessorCount;
        if (ncu_r <= 0) ncu_r = 64;
      }
      hipLaunchKernelGGL(fused_mid_kernel<true>, dim3((unsigned)ceil_div(n, kScanTile)), dim3(kScanThreads), 0, stream, b, o, bptr, hot, true, ncu_r);
      MI355_LAUNCH_CHECK();
      STEP(mi355i_csr_from_slots(csr_rank, b.occ_slot, b.occ + 1, reverse_indices, n, combiner >= 0 ? offsets : nullptr, num_bags,
                                 bptr, bcsr, hot_ws, hot_bytes_, emb_dim, a.hdr, nullptr, stream, b.gate, epoch, a.rerun_mark));
    }
    const int64_t* nu_dev = unique_offsets + num_tables;
    if (pin && !rerun_only && stage != 2) STEP(mi355_table_update_counter(counter, counter_numel, slots, n, nu_dev, 1, table_ids, table_bucket_offsets,
                                                            bucket_capacity, stream));
    // reverse indices / full ranks: on demand (mi355_demb_fused_materialize).  -2: the CSR is final (a flooded step was re-run in
    // line); -(2 + epoch): ask mi355_demb_fused_step_flooded(epoch) before the CSR or the unique numbering is used
    if (join_token) *join_token = (notice_mode && stage != 2) ? -(2 + epoch) : -2;
    return MI355_OK;
  }
  hipStream_t cs = stream;
  bool forked = false;
  // The side stream (numbering / CSR under the gather) is NOT used any more: the kernels that give evicted-into keys their row
  // address run behind the gather there, which then pools zero rows for them (round-2 advisor finding); measured slower back to
  // back anyway, and path (c) has nothing left to fork.  The argument is kept for ABI stability.
  (void)use_side_stream;
  if (false) {
    hipStream_t s2 = mi355i_side_fork(stream);
    if (!s2) { mi355_set_error("side stream fork failed"); return MI355_ELAUNCH; }
    cs = s2;
    forked = true;
  }
  auto gather = [&]() -> int {
    RoctxRange rr("op:gather_embedding");
    GatherTimer gt(stream);
    if (combiner >= 0)
      return mi355_gather_pooled(nullptr, 0, a.occ_addr, value_dtype, nullptr, n, offsets, num_bags, batch_size, combiner,
                                 emb_dim, D_offsets, total_D, out, out_dtype, aligned16, stream);
    if (combiner == -1)
      return mi355_gather_rows(nullptr, 0, a.occ_addr, value_dtype, nullptr, n, nullptr, emb_dim, out, emb_dim, out_dtype,
                               aligned16, stream);
    return MI355_OK;
  };
  if (forked) STEP(gather());
  if (train && n > 0) {
    RoctxRange rr("op:unique_numbering+backward_csr");
    EmitOut o;
    o.unique_keys = unique_keys; o.table_offsets = unique_offsets; o.table_ids = table_ids; o.slots = slots;
    o.row_addr = row_addr; o.freq = freq; o.csr_cnt = csr_cnt; o.total = total + 8;
    const int64_t* nu_dev = unique_offsets + num_tables;
    const int64_t ntile = ceil_div(n, kScanTile);
    static int ncu = 0;
    if (!ncu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
      if (ncu <= 0) ncu = 64;
    }
    HotList hot{};
    if (hot_ws) hot = hot_carve(hot_ws, n, emb_dim);
    if (ntile <= kSelfPrefixMaxTiles) {
      hipLaunchKernelGGL(fused_mid_kernel<true>, dim3((unsigned)ntile), dim3(kScanThreads), 0, cs, a, o, bptr, hot, hot_ws != nullptr, ncu);
    } else {
      hipLaunchKernelGGL(fused_scan_partials_kernel, dim3(1), dim3(kScanThreads), 0, cs, a.partial, ntile);
      hipLaunchKernelGGL(fused_mid_kernel<false>, dim3((unsigned)ntile), dim3(kScanThreads), 0, cs, a, o, bptr, hot, hot_ws != nullptr, ncu);
    }
    MI355_LAUNCH_CHECK();
    STEP(mi355i_csr_from_slots(csr_rank, a.occ_slot, a.occ + 1, reverse_indices, n, combiner >= 0 ? offsets : nullptr, num_bags,
                               bptr, bcsr, hot_ws, hot_bytes_, emb_dim, a.hdr, nullptr, cs));
    if (pin) STEP(mi355_table_update_counter(counter, counter_numel, slots, n, nu_dev, 1, table_ids, table_bucket_offsets,
                                             bucket_capacity, cs));
    if (forked) {
      const int tok = mi355i_side_mark();
      if (tok < 0) { mi355_set_error("side stream join record failed"); return MI355_ELAUNCH; }
      if (join_token) *join_token = tok;
      else STEP(mi355i_side_join(tok, stream));
    }
  }
  if (!forked) STEP(gather());
#undef STEP
  return MI355_OK;
}


// The index stage of a path-(c) step whose partition lists flooded (mi355_demb_fused_step_flooded(epoch) == 1), redone on the
// per-slot-counter path over the SAME buffers: probe (the step's keys are in the table by now: found; Assign / timer scores are
// idempotent, counting ones see the step twice), unique numbering, CSR of the backward, eager reverse indices.  Arguments: exactly
// those of the step's mi355_demb_forward_fused call (`out` is not written: the forward's output stands), plus its epoch.  After it
// the step's backward runs with prepared = 1 as usual.  Reference: the reference's unique op serves any key stream in one pass
// (unique_op.cu:484-714) -- this is what keeps that promise for a stream that defeats the slot-range partition.
int mi355_demb_forward_fused_rerun(
    void* storage, const int64_t* table_bucket_offsets, int64_t bucket_capacity, int64_t num_scores,
    int32_t* bucket_sizes, int32_t* counter, int64_t counter_numel, int32_t* aux, int64_t aux_numel, int64_t num_buckets,
    const int64_t* table_ptrs, const int64_t* table_value_dims, const int64_t* table_emb_dims,
    int value_dtype, int64_t emb_dim, int64_t value_dim,
    const void* keys, int64_t num_keys, const int64_t* offsets, int64_t num_bags, int64_t batch_size,
    const int64_t* feature_offsets, int64_t num_tables,
    int train, int find_policy, int insert_policy, uint64_t score_value, int use_count,
    uint64_t timer_override, int pin,
    int init_mode, float p0, float p1, float p2, float p3, uint64_t seed, float state_init,
    int combiner, const int32_t* D_offsets, int64_t total_D, void* out, int out_dtype, int aligned16,
    int64_t* reverse_indices, int64_t* unique_offsets, int64_t* table_ids, int64_t* slots,
    int64_t* row_addr, int64_t* freq, int32_t* csr_cnt, int32_t* csr_rank,
    void* backward_workspace, int64_t backward_workspace_bytes, int use_side_stream,
    int epoch,
    void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(epoch > 0 && train && num_keys > 0, "rerun: the epoch of a training step required");
  t_rerun_epoch = epoch;
  int tok = 0;
  const int rc = mi355_demb_forward_fused(storage, table_bucket_offsets, bucket_capacity, num_scores, bucket_sizes, counter, counter_numel, aux,
                                          aux_numel, num_buckets, table_ptrs, table_value_dims, table_emb_dims, value_dtype, emb_dim, value_dim,
                                          keys, num_keys, offsets, num_bags, batch_size, feature_offsets, num_tables, train, find_policy,
                                          insert_policy, score_value, use_count, timer_override, pin, init_mode, p0, p1, p2, p3, seed,
                                          state_init, combiner, D_offsets, total_D, out, out_dtype, aligned16, reverse_indices, unique_offsets,
                                          table_ids, slots, row_addr, freq, csr_cnt, csr_rank, backward_workspace, backward_workspace_bytes,
                                          use_side_stream, &tok, workspace, workspace_bytes, stream);
  t_rerun_epoch = 0;
  return rc;
}

// Per-occurrence outputs of a path-(c) forward that nothing on the training path reads (reference: the `inverse` of
// segmented_unique_cuda, unique_op.cu:484-714): reverse_indices [num_keys] and the rank of every occurrence inside its unique
// row's list (csr_rank), produced on demand from the records of the step's workspace; late (evicted-into) rows are patched into
// the per-occurrence address array as well.  `workspace` is the forward's, untouched since.
int mi355_demb_fused_materialize(void* workspace, int64_t workspace_bytes, int64_t num_keys, int64_t num_tables,
                                 const int64_t* row_addr, int64_t* reverse_indices, int32_t* csr_rank, hipStream_t stream) {
  MI355_CHECK_ARG(workspace && workspace_bytes >= mi355_demb_forward_fused_workspace_bytes(num_keys, num_tables), "workspace too small");
  MI355_CHECK_ARG(row_addr && reverse_indices && csr_rank, "outputs required");
  const int P = part_count(num_keys, num_tables);
  MI355_CHECK_ARG(P > 0, "not a partitioned step");
  if (num_keys == 0) return MI355_OK;
  uint8_t* w = (uint8_t*)workspace;
  const int64_t n = num_keys, nt = (n + 1023) / 1024 + 2;
  w += al256(8 * (num_tables + 1)) + al256(8 * n);
  int64_t* occ_addr = (int64_t*)w; w += al256(8 * n);
  const int32_t* occ_slot = (const int32_t*)w; w += al256(4 * n);
  w += al256(4 * nt);
  const int* rerun_mark = (const int*)w + 16;     // (the forward's `total` block: [8] occurrences, [16] the re-run mark)
  w += 256 + al256(8 * n);
  const int32_t* occ_trank = (const int32_t*)w; w += 4 * al256(4 * n);
  w += al256(24 * (nt > 258 ? nt : 258));
  const int64_t nr = (int64_t)P * kPartCap;
  w += al256(16 * nr);
  const int4* rec_out4 = (const int4*)w;
  hipLaunchKernelGGL(occ_from_records_kernel, dim3((unsigned)grid_for(n, 256, 4096)), dim3(256), 0, stream, occ_slot, occ_trank, rec_out4,
                     row_addr, n, reverse_indices, csr_rank, occ_addr, rerun_mark);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

}  // extern "C"
