// Scored hash table ("LinearBucketTable") kernels for gfx950.
//
// Replaces (reference, corelib/dynamicemb/src/table_operation/): table_lookup_kernel
// (kernels.cuh:81-187), table_insert_kernel / table_insert_and_evict_kernel /
// table_unlock_kernel (kernels.cuh:189-585), table_erase_kernel (:587-652),
// update_counter_with_layout_kernel (insert_and_evict.cu:27-58), the probe/reduce of
// types.cuh:308-512 and the score policies of score.cuh:30-99.
//
// MI355X design (not a translation of the one-thread-per-key CUDA kernels):
//  * a bucket's digest array is C bytes = ONE 128-B line at the default C = 128, so an
//    8-lane group owns one key: every lane pulls 16 digests with one dwordx4 load and the
//    whole bucket is matched in a single step (no serial 16-slot probe chain), a wave64
//    resolves 8 keys at a time; candidate key words are then fetched only by the lanes
//    whose digests matched.
//  * the min-score eviction scan is the same 8 lanes striding the bucket's score words as
//    coalesced 16-B pairs, followed by a 3-step xor-shuffle arg-min; no LDS staging or
//    cp.async-style double buffer is needed because the group reads a full line per step.
//  * slot lock = 64-bit CAS on the key word at agent scope (sc1), exactly one lane per
//    group issues it; evicted-stream compaction is a wave-wide ballot + mbcnt prefix and
//    ONE atomic per wave.
//  * all counts can stay on the device (`n_dev`), so the pipeline never syncs the host.
#include "common.h"
#include "../../include/recsys_amd.h"
#include "internal.h"

#include "table_dev.h"

namespace mi355 {


__global__ void __launch_bounds__(256) table_init_kernel(Table t, int64_t num_buckets) {
  const uint8_t ed = digest_of((int64_t)(fmix64(kEmptyKey) & 0x7FFFFFFFFFFFFFFFull));
  const int64_t total = num_buckets * t.C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t b = i / t.C, s = i % t.C;
    t.keys(b)[s] = kEmptyKey;
    t.dig(b)[s] = ed;
    for (int64_t k = 0; k < t.ns; ++k) t.scores(b)[s * t.ns + k] = 0;
  }
}

template <bool kLockFree>
__global__ void __launch_bounds__(256)
table_lookup_kernel(Table t, const int64_t* __restrict__ tbo, int64_t n, const int64_t* __restrict__ n_dev,
                    const uint64_t* __restrict__ keys, const int64_t* __restrict__ table_ids,
                    const uint64_t* __restrict__ score_in, int policy, uint64_t timer_override,
                    int64_t* __restrict__ score_out, uint8_t* __restrict__ founds, int64_t* __restrict__ indices) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int g = lane_id() & (G - 1);
  const int64_t gpb = blockDim.x / G;
  const uint64_t timer = timer_override ? timer_override : device_clock();
  for (int64_t base = (int64_t)blockIdx.x * gpb; base < n; base += (int64_t)gridDim.x * gpb) {
    const int64_t i = base + threadIdx.x / G;
    const bool act = i < n;
    uint64_t key = act ? keys[i] : kEmptyKey;
    int64_t tid = act ? (table_ids ? table_ids[i] : 0) : 0;
    uint64_t score = act ? policy_get(policy, score_in, i, timer) : 0;
    Located L = locate(key, tid, tbo, t.C);
    int found_slot = -1, empty_slot = -1;
    if (L.ok) group_probe(t, L.bucket, key, L.hash, false, false, found_slot, empty_slot);
    if (act && g == 0) {
      bool found = found_slot >= 0;
      int64_t index = -1;
      if (found) {
        uint64_t* sc = t.scores(L.bucket) + (int64_t)found_slot * t.ns;
        if (policy == kConst) {
          score = sc[t.ns - 1];
        } else if (kLockFree && (policy == kAssign || policy == kGlobalTimer)) {
          // Callers that order every table-mutating kernel on ONE stream (the fused forward) need no slot lock to
          // overwrite a score: the store is idempotent and nothing can evict the slot meanwhile.  Saves one
          // device-scope CAS + drain + unlock store per found key (~10 G/s of atomics is the whole kernel otherwise).
          ast64(sc, score);
        } else {
          uint64_t* kp = t.keys(L.bucket) + found_slot;
          uint64_t exp = key;
          if (cas64(kp, exp, kLockedKey)) {
            score = policy_update(policy, sc, score, timer);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ast64(kp, key);
          } else {
            found = false;  // concurrent writer owns the slot (kernels.cuh:142-145)
            score = 0;
          }
        }
        if (found) index = (L.bucket - L.bkt_begin) * t.C + found_slot;
      }
      if (score_out) score_out[i] = (int64_t)score;
      founds[i] = found ? 1 : 0;
      indices[i] = index;
    }
  }
}

// insert / insert_and_evict (kernels.cuh:189-567).  `skip` (optional): keys with skip[i] != 0 are
// left alone (their `indices[i]` stays as the caller set it) -- lets "lookup then insert the
// missing ones" run without a host-side compaction.
__global__ void __launch_bounds__(256)
table_insert_kernel(Table t, const int64_t* __restrict__ tbo, int32_t* __restrict__ bucket_sizes,
                    int32_t* __restrict__ counter, int64_t n, const int64_t* __restrict__ n_dev,
                    const uint64_t* __restrict__ keys, const int64_t* __restrict__ table_ids,
                    const uint64_t* __restrict__ score_in, int policy, uint64_t timer_override,
                    const uint8_t* __restrict__ skip, int64_t* __restrict__ indices,
                    uint8_t* __restrict__ results, int64_t* __restrict__ score_out,
                    unsigned long long* __restrict__ num_evicted, uint64_t* __restrict__ ev_keys,
                    int64_t* __restrict__ ev_indices, int64_t* __restrict__ ev_scores,
                    int64_t* __restrict__ ev_table_ids, uint64_t* __restrict__ dense_ev_key = nullptr,
                    int64_t* __restrict__ dense_ev_score = nullptr) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const int g = lane_id() & (G - 1);
  const int64_t gpb = blockDim.x / G;
  const uint64_t timer = timer_override ? timer_override : device_clock();
  const int C = (int)t.C;
  for (int64_t base = (int64_t)blockIdx.x * gpb; base < n; base += (int64_t)gridDim.x * gpb) {
    const int64_t i = base + threadIdx.x / G;
    const bool act = i < n && !(skip && skip[i]);
    uint64_t key = act ? keys[i] : kEmptyKey;
    int64_t tid = act ? (table_ids ? table_ids[i] : 0) : 0;
    uint64_t score = act ? policy_get(policy, score_in, i, timer) : 0;
    Located L = locate(key, tid, tbo, t.C);
    int res = act ? (L.ok ? kInit : kIllegal) : kIllegal;
    int slot = -1;
    uint64_t ev_key = 0, ev_score = 0;

    if (L.ok) {
      uint64_t* ks = t.keys(L.bucket);
      // ---- existing key / first Empty slot (insert_probe, kernels.cuh:189-224) ----
      bool fresh = false;
      while (true) {
        int found_slot, empty_slot;
        group_probe(t, L.bucket, key, L.hash, fresh, true, found_slot, empty_slot);
        int ok = 0;  // 0 exhausted, 1 assigned, 2 inserted, 3 retry, 4 lost the lock on an existing key
        if (found_slot >= 0) {
          if (g == 0) { uint64_t e = key; ok = cas64(ks + found_slot, e, kLockedKey) ? 1 : 4; }
          ok = group_bcast(ok, 0);
          slot = found_slot;
        } else if (empty_slot >= 0) {
          if (g == 0) {
            uint64_t e = kEmptyKey;
            if (cas64(ks + empty_slot, e, kLockedKey)) {
              t.dig(L.bucket)[empty_slot] = digest_of(L.hash);
              atomicAdd(&bucket_sizes[L.bucket], 1);
              ok = 2;
            } else ok = 3;
          }
          ok = group_bcast(ok, 0);
          slot = empty_slot;
        }
        if (ok == 1) { res = kAssigned; break; }
        if (ok == 2) { res = kInsert; break; }
        if (ok == 3) { fresh = true; continue; }
        break;  // exhausted or lock lost -> eviction path with res == kInit
      }
      // ---- bucket full: evict the minimum score (insert(), kernels.cuh:226-287) ----
      while (res == kInit) {
        uint64_t best = ~0ull, bkey = 0;
        int bslot = -1;
        const uint64_t* sc = t.scores(L.bucket);
        const int32_t* cnt = counter ? counter + L.bucket * t.C : nullptr;
        for (int s0 = 2 * g; s0 < C; s0 += 2 * G) {
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int s = s0 + u;
            uint64_t v = ald64(sc + (int64_t)s * t.ns + (t.ns - 1));
            if (v < best) {
              uint64_t k = ald64(ks + s);
              if (k == kLockedKey || k == kEmptyKey) continue;
              if (cnt && __hip_atomic_load(cnt + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0) continue;
              best = v; bslot = s; bkey = k;
            }
          }
        }
        group_argmin(best, bslot, bkey);
        if (bslot < 0) { res = kBusy; ev_key = key; ev_score = score; break; }
        int ok = 0;  // 0 retry, 1 reclaim, 2 evict
        if (g == 0) {
          uint64_t e = bkey;
          if (cas64(ks + bslot, e, kLockedKey)) {
            bool good = true;
            if (bkey != kReclaimKey && cnt &&
                __hip_atomic_load(cnt + bslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0)
              good = false;
            if (good && ald64(sc + (int64_t)bslot * t.ns + (t.ns - 1)) != best) good = false;
            if (!good) {
              ast64(ks + bslot, bkey);
            } else {
              t.dig(L.bucket)[bslot] = digest_of(L.hash);
              if (bkey == kReclaimKey) { atomicAdd(&bucket_sizes[L.bucket], 1); ok = 1; }
              else {
                for (int64_t w = 0; w < t.ns; ++w) ast64((uint64_t*)sc + (int64_t)bslot * t.ns + w, 0);
                ok = 2;
              }
            }
          }
        }
        ok = group_bcast(ok, 0);
        if (ok == 1) { res = kReclaim; slot = bslot; }
        else if (ok == 2) { res = kEvict; slot = bslot; ev_key = bkey; ev_score = best; }
      }
    }

    int64_t index = -1;
    if (act && g == 0 && res <= kEvict) {
      score = policy_update(policy, t.scores(L.bucket) + (int64_t)slot * t.ns, score, timer);
      index = (L.bucket - L.bkt_begin) * t.C + slot;
    }
    // ---- evicted-stream compaction: one ballot + one atomic per wave (kernels.cuh:522-556) ----
    if (num_evicted) {
      const bool ev = act && g == 0 && (res == kEvict || res == kBusy);
      const uint64_t vote = __ballot(ev);
      if (vote) {
        const int lane = lane_id();
        const int leader = __ffsll((unsigned long long)vote) - 1;
        unsigned long long off = 0;
        if (lane == leader) off = atomicAdd(num_evicted, (unsigned long long)__popcll(vote));
        uint32_t lo = __shfl((int)(uint32_t)off, leader, 64), hi = __shfl((int)(uint32_t)(off >> 32), leader, 64);
        off = ((unsigned long long)hi << 32) | lo;
        if (ev) {
          int64_t o = (int64_t)off + __popcll(vote & ((1ull << lane) - 1));
          ev_keys[o] = ev_key;
          ev_scores[o] = (int64_t)ev_score;
          ev_indices[o] = res == kEvict ? index : -(i + 1);
          ev_table_ids[o] = tid;
        }
      }
    }
    if (act && g == 0) {
      indices[i] = index;
      if (results) results[i] = (uint8_t)res;
      if (score_out) score_out[i] = (int64_t)score;
      // overflow build: the evicted record stays per key; the overflow pass decides what is reported and compacts
      if (dense_ev_key) { dense_ev_key[i] = ev_key; dense_ev_score[i] = (int64_t)ev_score; }
    }
  }
}

// ---- overflow region (scored_hashtable.py:426-474): ONE bucket of `ovf.C` (= 3 x bucket capacity) slots per logical
// table, linear probing from hash % ovf.C, victims chosen by ref-counter == 0 instead of by score.  Indices of overflow
// entries are table-relative: out_off[t] (= the table's main capacity) + position.  The region is only touched by keys
// the main table refused (every slot of their bucket pinned), so one thread per key is enough.

// overflow_find + the overflow branch of table_lookup_kernel (kernels.cuh:153-183, 711-736)
__global__ void __launch_bounds__(256)
table_lookup_ovf_kernel(Table ovf, const int64_t* __restrict__ out_off, int64_t n, const int64_t* __restrict__ n_dev,
                        const uint64_t* __restrict__ keys, const int64_t* __restrict__ table_ids,
                        const uint64_t* __restrict__ score_in, int policy, uint64_t timer_override,
                        int64_t* __restrict__ score_out, uint8_t* __restrict__ founds, int64_t* __restrict__ indices) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const uint64_t timer = timer_override ? timer_override : device_clock();
  const int64_t cap = ovf.C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (founds[i]) continue;
    const uint64_t key = keys[i];
    if (!is_valid(key)) continue;
    const int64_t tid = table_ids ? table_ids[i] : 0;
    const int64_t hash = (int64_t)(fmix64(key) & 0x7FFFFFFFFFFFFFFFull);
    uint64_t* ks = ovf.keys(tid);
    int64_t pos = hash % cap, hit = -1;
    for (int64_t scan = 0; scan < cap; ++scan) {
      const uint64_t k = ald64(ks + pos);
      if (k == key) { hit = pos; break; }
      if (k == kEmptyKey) break;
      if (++pos == cap) pos = 0;
    }
    if (hit < 0) continue;
    uint64_t* sc = ovf.scores(tid) + hit * ovf.ns;
    uint64_t score = policy_get(policy, score_in, i, timer);
    if (policy == kConst) {
      score = sc[ovf.ns - 1];
    } else {
      uint64_t e = key;
      if (cas64(ks + hit, e, kLockedKey)) {   // (an entry that moved out meanwhile keeps the input score, :170-176)
        score = policy_update(policy, sc, score, timer);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ast64(ks + hit, key);
      }
    }
    if (score_out) score_out[i] = (int64_t)score;
    founds[i] = 1;
    indices[i] = hit + out_off[tid];
  }
}

// overflow_insert_and_evict + the overflow branch of table_insert_and_evict_kernel (kernels.cuh:468-566, 738-800):
// second pass over the keys the main insert left Busy, then the evicted-stream compaction for ALL keys (the main kernel
// ran with per-key evicted records: dense_ev_*).
__global__ void __launch_bounds__(256)
table_insert_ovf_kernel(Table ovf, const int64_t* __restrict__ out_off, int32_t* __restrict__ ovf_sizes,
                        int32_t* __restrict__ ovf_counter, int64_t n, const int64_t* __restrict__ n_dev,
                        const uint64_t* __restrict__ keys, const int64_t* __restrict__ table_ids,
                        const uint64_t* __restrict__ score_in, int policy, uint64_t timer_override,
                        const uint8_t* __restrict__ skip, int64_t* __restrict__ indices, uint8_t* __restrict__ results,
                        int64_t* __restrict__ score_out, const uint64_t* __restrict__ dense_ev_key,
                        const int64_t* __restrict__ dense_ev_score, unsigned long long* __restrict__ num_evicted,
                        uint64_t* __restrict__ ev_keys, int64_t* __restrict__ ev_indices, int64_t* __restrict__ ev_scores,
                        int64_t* __restrict__ ev_table_ids) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  const uint64_t timer = timer_override ? timer_override : device_clock();
  const int64_t cap = ovf.C;
  const int64_t n_up = (n + 63) / 64 * 64;   // whole waves: every lane takes part in the ballot
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_up; i += (int64_t)gridDim.x * blockDim.x) {
    const bool act = i < n && !(skip && skip[i]);
    int res = act ? (int)results[i] : (int)kIllegal;
    const uint64_t key = act ? keys[i] : kEmptyKey;
    const int64_t tid = act ? (table_ids ? table_ids[i] : 0) : 0;
    uint64_t f_key = act ? dense_ev_key[i] : 0;
    const int64_t f_score = act ? dense_ev_score[i] : 0;
    int64_t f_index = res == kEvict ? indices[i] : -(i + 1);
    if (res == kBusy && is_valid(key)) {
      const int64_t hash = (int64_t)(fmix64(key) & 0x7FFFFFFFFFFFFFFFull);
      uint64_t* ks = ovf.keys(tid);
      int32_t* cnt = ovf_counter + tid * cap;
      int64_t pos = hash % cap, at = -1;
      int ores = kBusy;
      uint64_t okey = 0;
      for (int64_t scan = 0; scan < cap; ++scan, pos = pos + 1 == cap ? 0 : pos + 1) {
        uint64_t k = ald64(ks + pos);
        if (k == key) { at = pos; ores = kAssigned; break; }
        if (k == kEmptyKey) {
          uint64_t e = kEmptyKey;
          if (cas64(ks + pos, e, kLockedKey)) {
            ovf.dig(tid)[pos] = digest_of(hash);
            atomicAdd(&ovf_sizes[tid], 1);
            at = pos; ores = kInsert; break;
          }
          k = ald64(ks + pos);
          if (k == key) { at = pos; ores = kAssigned; break; }
          continue;
        }
        if (k == kLockedKey || k == kReclaimKey) continue;
        if (__hip_atomic_load(cnt + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
          uint64_t e = k;
          if (cas64(ks + pos, e, kLockedKey)) {
            if (atomicAdd(cnt + pos, 0) > 0) { ast64(ks + pos, k); continue; }
            ovf.dig(tid)[pos] = digest_of(hash);
            okey = k; at = pos; ores = kEvict; break;
          }
        }
      }
      if (at >= 0) {
        uint64_t* sc = ovf.scores(tid) + at * ovf.ns;
        uint64_t score = policy_get(policy, score_in, i, timer);
        if (ores == kAssigned) {   // not locked by the find: guard the score write (:505-513)
          uint64_t e = key;
          if (cas64(ks + at, e, kLockedKey)) {
            score = policy_update(policy, sc, score, timer);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ast64(ks + at, key);
          }
        } else {                   // newly occupied, stays locked until table_unlock_ovf_kernel
          for (int64_t w = 0; w < ovf.ns; ++w) sc[w] = 0;
          score = policy_update(policy, sc, score, timer);
          if (ores == kEvict) { f_key = okey; f_index = at + out_off[tid]; }
        }
        res = ores;
        indices[i] = at + out_off[tid];
        results[i] = (uint8_t)res;
        if (score_out) score_out[i] = (int64_t)score;
      }
    }
    if (num_evicted) {
      const bool ev = act && (res == kEvict || res == kBusy);
      const uint64_t vote = __ballot(ev);
      if (vote) {
        const int lane = lane_id();
        const int leader = __ffsll((unsigned long long)vote) - 1;
        unsigned long long off = 0;
        if (lane == leader) off = atomicAdd(num_evicted, (unsigned long long)__popcll(vote));
        uint32_t lo = __shfl((int)(uint32_t)off, leader, 64), hi = __shfl((int)(uint32_t)(off >> 32), leader, 64);
        off = ((unsigned long long)hi << 32) | lo;
        if (ev) {
          int64_t o = (int64_t)off + __popcll(vote & ((1ull << lane) - 1));
          ev_keys[o] = f_key;
          ev_scores[o] = f_score;
          ev_indices[o] = f_index;
          ev_table_ids[o] = tid;
        }
      }
    }
  }
}

// table_unlock_kernel for the overflow slots taken above (Insert / Evict results whose index lies in the overflow range)
__global__ void __launch_bounds__(256)
table_unlock_ovf_kernel(Table ovf, const int64_t* __restrict__ out_off, int64_t n, const int64_t* __restrict__ n_dev,
                        const uint64_t* __restrict__ keys, const int64_t* __restrict__ table_ids,
                        const uint8_t* __restrict__ skip, const int64_t* __restrict__ indices,
                        const uint8_t* __restrict__ results) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (skip && skip[i]) continue;
    const int r = results[i];
    if (r != kInsert && r != kEvict) continue;
    const int64_t tid = table_ids ? table_ids[i] : 0;
    const int64_t local = indices[i] - out_off[tid];
    if (local < 0) continue;
    ast64(ovf.keys(tid) + local, keys[i]);
  }
}

// table_unlock_kernel (kernels.cuh:569-585): publish the real key into every slot taken above.
__global__ void __launch_bounds__(256)
table_unlock_kernel(Table t, const int64_t* __restrict__ tbo, int64_t n, const int64_t* __restrict__ n_dev,
                    const uint64_t* __restrict__ keys, const int64_t* __restrict__ table_ids,
                    const uint8_t* __restrict__ skip, const int64_t* __restrict__ indices,
                    const int64_t* __restrict__ table_ptrs, const int64_t* __restrict__ table_value_dims, int elem_bytes,
                    int64_t* __restrict__ row_addr) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = indices[i];
    const int64_t tid = table_ids ? table_ids[i] : 0;
    // fused mi355_row_addresses: every key (found earlier or placed now) gets the address of its row
    if (row_addr) row_addr[i] = idx < 0 ? 0 : table_ptrs[tid] + idx * table_value_dims[tid] * elem_bytes;
    if (skip && skip[i]) continue;
    if (idx < 0) continue;
    int64_t b = tbo[tid] + idx / t.C;
    ast64(t.keys(b) + idx % t.C, keys[i]);
  }
}

// table_erase_kernel (kernels.cuh:587-652)
__global__ void __launch_bounds__(256)
table_erase_kernel(Table t, const int64_t* __restrict__ tbo, int32_t* __restrict__ bucket_sizes, int64_t n,
                   const uint64_t* __restrict__ keys, const int64_t* __restrict__ table_ids,
                   int64_t* __restrict__ indices) {
  const int g = lane_id() & (G - 1);
  const int64_t gpb = blockDim.x / G;
  const uint8_t ed = digest_of((int64_t)(fmix64(kEmptyKey) & 0x7FFFFFFFFFFFFFFFull));
  for (int64_t base = (int64_t)blockIdx.x * gpb; base < n; base += (int64_t)gridDim.x * gpb) {
    const int64_t i = base + threadIdx.x / G;
    const bool act = i < n;
    uint64_t key = act ? keys[i] : kEmptyKey;
    int64_t tid = act ? (table_ids ? table_ids[i] : 0) : 0;
    Located L = locate(key, tid, tbo, t.C);
    int found_slot = -1, empty_slot = -1;
    if (L.ok) group_probe(t, L.bucket, key, L.hash, false, false, found_slot, empty_slot);
    if (act && g == 0) {
      int64_t index = -1;
      if (found_slot >= 0) {
        uint64_t* kp = t.keys(L.bucket) + found_slot;
        uint64_t e = key;
        if (cas64(kp, e, kLockedKey)) {
          ast64(t.scores(L.bucket) + (int64_t)found_slot * t.ns, 0);
          t.dig(L.bucket)[found_slot] = ed;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          ast64(kp, kReclaimKey);
          atomicSub(&bucket_sizes[L.bucket], 1);
          index = (L.bucket - L.bkt_begin) * t.C + found_slot;
        }
      }
      if (indices) indices[i] = index;
    }
  }
}

// update_counter_with_layout_kernel (insert_and_evict.cu:27-58), main table part: non-atomic +=
// (the caller guarantees unique (table, slot) pairs, scored_hashtable.py:680-684).
__global__ void __launch_bounds__(256)
update_counter_kernel(int32_t* __restrict__ counter, int64_t total, const int64_t* __restrict__ slots, int64_t n,
                      const int64_t* __restrict__ n_dev, int32_t delta, const int64_t* __restrict__ table_ids,
                      const int64_t* __restrict__ tbo, int64_t C, const uint8_t* __restrict__ flags = nullptr, int want = 0,
                      int64_t main_capacity = 0, const int64_t* __restrict__ ovf_out_off = nullptr, int64_t ovf_cap = 0) {
  if (n_dev) { int64_t m = *n_dev; n = m < n ? m : n; }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t s = slots[i];
    if (s < 0) continue;
    if (flags && (flags[i] != 0) != (want != 0)) continue;
    int64_t flat = (table_ids && tbo) ? tbo[table_ids[i]] * C + s : s;
    if (ovf_out_off) {   // insert_and_evict.cu:42-50: overflow slots live behind the main arena, table by table
      const int64_t tid = table_ids ? table_ids[i] : 0;
      const int64_t per = ovf_out_off[tid];
      flat = s < per ? tbo[tid] * C + s : main_capacity + tid * ovf_cap + (s - per);
    }
    if (flat >= 0 && flat < total) counter[flat] += delta;
  }
}

__global__ void device_timestamp_kernel(int64_t* out) { *out = (int64_t)device_clock(); }

// ---- table scan: export / count (table_export_batch_kernel kernels.cuh:655-705, EvalAndCount :59-79) ----
// One thread per slot of [begin, end): flag = valid key && (no threshold || score[score_index] >= threshold).
// The dense (flag, key, score, index) arrays are then compacted in slot order (mi355_flagged_compact), so the
// export order is deterministic; the reference's is atomicAdd order.
__global__ void __launch_bounds__(256)
table_scan_kernel(Table t, int64_t begin, int64_t end, int64_t table_begin, int has_thr, uint64_t thr, int64_t score_index,
                  uint8_t* __restrict__ flags, uint64_t* __restrict__ keys, uint64_t* __restrict__ scores,
                  int64_t* __restrict__ indices) {
  for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / t.C, it = i % t.C;
    const uint64_t k = t.keys(b)[it];
    const uint64_t sc = t.scores(b)[it * t.ns + score_index];
    const bool m = is_valid(k) && (!has_thr || sc >= thr);
    const int64_t o = i - begin;
    flags[o] = m ? 1 : 0;
    keys[o] = k;
    scores[o] = sc;
    indices[o] = i - table_begin;
  }
}

__global__ void __launch_bounds__(256)
table_count_kernel(Table t, int64_t begin, int64_t end, uint64_t thr, int64_t score_index, unsigned long long* counter) {
  int cnt = 0;
  for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = i / t.C, it = i % t.C;
    cnt += (is_valid(t.keys(b)[it]) && t.scores(b)[it * t.ns + score_index] >= thr) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane_id() == 0 && cnt) atomicAdd(counter, (unsigned long long)cnt);
}

// copy / gather / scatter of whole score blocks (kernels.cuh:839-910); slots are table-relative flat indices
__global__ void __launch_bounds__(256)
score_blocks_kernel(int mode, Table src, int64_t src_bkt_begin, Table dst, int64_t dst_bkt_begin, int64_t n,
                    const int64_t* __restrict__ src_slots, const int64_t* __restrict__ dst_slots, uint64_t* dense) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t ns = dst.ns;
  if (mode == 0) {  // copy src[src_slots[i]] -> dst[dst_slots[i]]
    const int64_t ss = src_slots[i], ds = dst_slots[i];
    if (ss < 0 || ds < 0) return;
    const uint64_t* sp = src.scores(src_bkt_begin + ss / src.C) + (ss % src.C) * src.ns;
    uint64_t* dp = dst.scores(dst_bkt_begin + ds / dst.C) + (ds % dst.C) * dst.ns;
    for (int64_t k = 0; k < ns; ++k) dp[k] = sp[k];
  } else if (mode == 1) {  // gather -> dense [n, ns]
    const int64_t sl = src_slots[i];
    if (sl < 0) { for (int64_t k = 0; k < ns; ++k) dense[i * ns + k] = 0; return; }
    const uint64_t* sp = src.scores(src_bkt_begin + sl / src.C) + (sl % src.C) * src.ns;
    for (int64_t k = 0; k < ns; ++k) dense[i * ns + k] = sp[k];
  } else {  // scatter dense [n, ns] -> dst
    const int64_t sl = dst_slots[i];
    if (sl < 0) return;
    uint64_t* dp = dst.scores(dst_bkt_begin + sl / dst.C) + (sl % dst.C) * dst.ns;
    for (int64_t k = 0; k < ns; ++k) dp[k] = dense[i * ns + k];
  }
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_table_init(void* storage, int64_t num_buckets, int64_t C, int64_t num_scores, hipStream_t stream) {
  MI355_CHECK_ARG(C > 0 && C % 16 == 0, "bucket capacity must be a positive multiple of 16");
  MI355_CHECK_ARG(num_scores >= 1 && num_scores <= 2, "num_scores must be 1 or 2");
  if (num_buckets == 0) return MI355_OK;
  Table t = make_table(storage, C, num_scores);
  hipLaunchKernelGGL(table_init_kernel, dim3(grid_for(num_buckets * C, 256)), dim3(256), 0, stream, t, num_buckets);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

static int table_lookup_impl(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores, int64_t n,
                             const int64_t* n_dev, const void* keys, const int64_t* table_ids, const void* score_in,
                             int policy, uint64_t timer_override, int64_t* score_out, uint8_t* founds, int64_t* indices,
                             bool lock_free, hipStream_t stream) {
  MI355_CHECK_ARG(C > 0 && C % 16 == 0, "bucket capacity must be a positive multiple of 16");
  MI355_CHECK_ARG(policy >= kConst && policy <= kLruLfu, "bad score policy");
  MI355_CHECK_ARG(policy == kConst || policy == kGlobalTimer || score_in, "score_in required by this policy");
  MI355_CHECK_ARG(policy != kLruLfu || num_scores == 2, "LRU_LFU needs num_scores == 2");
  if (n == 0) return MI355_OK;
  Table t = make_table(storage, C, num_scores);
  if (lock_free)
    hipLaunchKernelGGL(table_lookup_kernel<true>, dim3(grid_for(n, 256 / G)), dim3(256), 0, stream, t, table_bucket_offsets, n,
                       n_dev, (const uint64_t*)keys, table_ids, (const uint64_t*)score_in, policy, timer_override,
                       score_out, founds, indices);
  else
    hipLaunchKernelGGL(table_lookup_kernel<false>, dim3(grid_for(n, 256 / G)), dim3(256), 0, stream, t, table_bucket_offsets, n,
                       n_dev, (const uint64_t*)keys, table_ids, (const uint64_t*)score_in, policy, timer_override,
                       score_out, founds, indices);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// table_lookup with the reference's slot-lock protocol (safe against concurrent inserts on other streams)
int mi355_table_lookup(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores, int64_t n,
                       const int64_t* n_dev, const void* keys, const int64_t* table_ids, const void* score_in,
                       int policy, uint64_t timer_override, int64_t* score_out, uint8_t* founds, int64_t* indices,
                       hipStream_t stream) {
  return table_lookup_impl(storage, table_bucket_offsets, C, num_scores, n, n_dev, keys, table_ids, score_in, policy,
                           timer_override, score_out, founds, indices, false, stream);
}

int mi355_table_insert(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores,
                       int32_t* bucket_sizes, int32_t* counter, int64_t n, const int64_t* n_dev, const void* keys,
                       const int64_t* table_ids, const void* score_in, int policy, uint64_t timer_override,
                       const uint8_t* skip, int64_t* indices, uint8_t* results, int64_t* score_out,
                       int64_t* num_evicted, void* evicted_keys, int64_t* evicted_indices, int64_t* evicted_scores,
                       int64_t* evicted_table_ids, hipStream_t stream) {
  MI355_CHECK_ARG(C > 0 && C % 16 == 0, "bucket capacity must be a positive multiple of 16");
  MI355_CHECK_ARG(policy >= kConst && policy <= kLruLfu, "bad score policy");
  MI355_CHECK_ARG(policy == kConst || policy == kGlobalTimer || score_in, "score_in required by this policy");
  MI355_CHECK_ARG(policy != kLruLfu || num_scores == 2, "LRU_LFU needs num_scores == 2");
  MI355_CHECK_ARG(!num_evicted || (evicted_keys && evicted_indices && evicted_scores && evicted_table_ids),
                  "evicted output buffers required with num_evicted");
  if (num_evicted) {
    if (hipMemsetAsync(num_evicted, 0, sizeof(int64_t), stream) != hipSuccess) { mi355_set_error("memset failed"); return MI355_ELAUNCH; }
  }
  if (n == 0) return MI355_OK;
  Table t = make_table(storage, C, num_scores);
  hipLaunchKernelGGL(table_insert_kernel, dim3(grid_for(n, 256 / G)), dim3(256), 0, stream, t, table_bucket_offsets,
                     bucket_sizes, counter, n, n_dev, (const uint64_t*)keys, table_ids, (const uint64_t*)score_in,
                     policy, timer_override, skip, indices, results, score_out, (unsigned long long*)num_evicted,
                     (uint64_t*)evicted_keys, evicted_indices, evicted_scores, evicted_table_ids);
  MI355_LAUNCH_CHECK();
  hipLaunchKernelGGL(table_unlock_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, t, table_bucket_offsets, n, n_dev,
                     (const uint64_t*)keys, table_ids, skip, indices, (const int64_t*)nullptr, (const int64_t*)nullptr, 0,
                     (int64_t*)nullptr);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// table_lookup for callers that keep every table-mutating kernel on one stream: ASSIGN / GLOBAL_TIMER scores are written
// without the slot lock (see table_lookup_kernel)
int mi355i_table_lookup(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores, int64_t n,
                        const int64_t* n_dev, const void* keys, const int64_t* table_ids, const void* score_in,
                        int policy, uint64_t timer_override, int64_t* score_out, uint8_t* founds, int64_t* indices,
                        hipStream_t stream) {
  return table_lookup_impl(storage, table_bucket_offsets, C, num_scores, n, n_dev, keys, table_ids, score_in, policy,
                           timer_override, score_out, founds, indices, true, stream);
}

int mi355i_table_insert(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores,
                        int32_t* bucket_sizes, int32_t* counter, int64_t n, const int64_t* n_dev, const void* keys,
                        const int64_t* table_ids, const void* score_in, int policy, uint64_t timer_override,
                        const uint8_t* skip, int64_t* indices, uint8_t* results, const int64_t* table_ptrs,
                        const int64_t* table_value_dims, int elem_bytes, int64_t* row_addr_out, hipStream_t stream) {
  MI355_CHECK_ARG(C > 0 && C % 16 == 0, "bucket capacity must be a positive multiple of 16");
  MI355_CHECK_ARG(policy >= kConst && policy <= kLruLfu, "bad score policy");
  MI355_CHECK_ARG(policy == kConst || policy == kGlobalTimer || score_in, "score_in required by this policy");
  MI355_CHECK_ARG(policy != kLruLfu || num_scores == 2, "LRU_LFU needs num_scores == 2");
  if (n == 0) return MI355_OK;
  Table t = make_table(storage, C, num_scores);
  hipLaunchKernelGGL(table_insert_kernel, dim3(grid_for(n, 256 / G)), dim3(256), 0, stream, t, table_bucket_offsets,
                     bucket_sizes, counter, n, n_dev, (const uint64_t*)keys, table_ids, (const uint64_t*)score_in,
                     policy, timer_override, skip, indices, results, (int64_t*)nullptr, (unsigned long long*)nullptr,
                     (uint64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr);
  MI355_LAUNCH_CHECK();
  if (!row_addr_out) return MI355_OK;   // the caller runs the unlock pass itself (mi355i_unlock_init_rows)
  hipLaunchKernelGGL(table_unlock_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, t, table_bucket_offsets, n, n_dev,
                     (const uint64_t*)keys, table_ids, skip, indices, table_ptrs, table_value_dims, elem_bytes, row_addr_out);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_table_erase(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores,
                      int32_t* bucket_sizes, int64_t n, const void* keys, const int64_t* table_ids, int64_t* indices,
                      hipStream_t stream) {
  MI355_CHECK_ARG(C > 0 && C % 16 == 0, "bucket capacity must be a positive multiple of 16");
  if (n == 0) return MI355_OK;
  Table t = make_table(storage, C, num_scores);
  hipLaunchKernelGGL(table_erase_kernel, dim3(grid_for(n, 256 / G)), dim3(256), 0, stream, t, table_bucket_offsets,
                     bucket_sizes, n, (const uint64_t*)keys, table_ids, indices);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_table_update_counter(int32_t* counter, int64_t counter_numel, const int64_t* slot_indices, int64_t n,
                               const int64_t* n_dev, int32_t delta, const int64_t* table_ids,
                               const int64_t* table_bucket_offsets, int64_t C, hipStream_t stream) {
  MI355_CHECK_ARG(counter && slot_indices, "counter and slot_indices required");
  if (n == 0) return MI355_OK;
  hipLaunchKernelGGL(update_counter_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, counter, counter_numel,
                     slot_indices, n, n_dev, delta, table_ids, table_bucket_offsets, C);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_table_update_counter_overflow(int32_t* counter, int64_t counter_numel, const int64_t* slot_indices, int64_t n,
                                        const int64_t* n_dev, int32_t delta, const int64_t* table_ids,
                                        const int64_t* table_bucket_offsets, int64_t C, int64_t main_capacity,
                                        const int64_t* ovf_output_offsets, int64_t ovf_capacity, hipStream_t stream) {
  MI355_CHECK_ARG(table_bucket_offsets && ovf_output_offsets, "table_bucket_offsets and ovf_output_offsets required");
  if (n == 0) return MI355_OK;
  hipLaunchKernelGGL(update_counter_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, counter, counter_numel,
                     slot_indices, n, n_dev, delta, table_ids, table_bucket_offsets, C, (const uint8_t*)nullptr, 0,
                     main_capacity, ovf_output_offsets, ovf_capacity);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_table_lookup_overflow(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores, int64_t n,
                                const int64_t* n_dev, const void* keys, const int64_t* table_ids, const void* score_in,
                                int policy, uint64_t timer_override, void* ovf_storage, int64_t ovf_capacity,
                                const int64_t* ovf_output_offsets, int64_t* score_out, uint8_t* founds, int64_t* indices,
                                hipStream_t stream) {
  MI355_CHECK_ARG(ovf_storage && ovf_output_offsets && ovf_capacity > 0 && ovf_capacity % 16 == 0,
                  "overflow storage, output offsets and a positive capacity (multiple of 16) required");
  int rc = table_lookup_impl(storage, table_bucket_offsets, C, num_scores, n, n_dev, keys, table_ids, score_in, policy,
                             timer_override, score_out, founds, indices, false, stream);
  if (rc != MI355_OK || n == 0) return rc;
  Table ovf = make_table(ovf_storage, ovf_capacity, num_scores);
  hipLaunchKernelGGL(table_lookup_ovf_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, ovf, ovf_output_offsets, n,
                     n_dev, (const uint64_t*)keys, table_ids, (const uint64_t*)score_in, policy, timer_override, score_out,
                     founds, indices);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int64_t mi355_table_insert_overflow_workspace_bytes(int64_t n) { return 16 * (n > 0 ? n : 1); }

int mi355_table_insert_overflow(void* storage, const int64_t* table_bucket_offsets, int64_t C, int64_t num_scores,
                                int32_t* bucket_sizes, int32_t* counter, int64_t n, const int64_t* n_dev, const void* keys,
                                const int64_t* table_ids, const void* score_in, int policy, uint64_t timer_override,
                                const uint8_t* skip, void* ovf_storage, int64_t ovf_capacity, int32_t* ovf_bucket_sizes,
                                int32_t* ovf_counter, const int64_t* ovf_output_offsets, int64_t* indices,
                                uint8_t* results, int64_t* score_out, int64_t* num_evicted, void* evicted_keys,
                                int64_t* evicted_indices, int64_t* evicted_scores, int64_t* evicted_table_ids,
                                void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(C > 0 && C % 16 == 0, "bucket capacity must be a positive multiple of 16");
  MI355_CHECK_ARG(policy >= kConst && policy <= kLruLfu, "bad score policy");
  MI355_CHECK_ARG(policy == kConst || policy == kGlobalTimer || score_in, "score_in required by this policy");
  MI355_CHECK_ARG(policy != kLruLfu || num_scores == 2, "LRU_LFU needs num_scores == 2");
  MI355_CHECK_ARG(ovf_storage && ovf_bucket_sizes && ovf_counter && ovf_output_offsets && ovf_capacity > 0 &&
                      ovf_capacity % 16 == 0, "overflow storage / sizes / counter / output offsets required");
  MI355_CHECK_ARG(counter && results, "the overflow insert needs the ref-counter and a results buffer");
  MI355_CHECK_ARG(!num_evicted || (evicted_keys && evicted_indices && evicted_scores && evicted_table_ids),
                  "evicted output buffers required with num_evicted");
  MI355_CHECK_ARG(workspace && workspace_bytes >= mi355_table_insert_overflow_workspace_bytes(n), "workspace too small");
  if (num_evicted) {
    if (hipMemsetAsync(num_evicted, 0, sizeof(int64_t), stream) != hipSuccess) { mi355_set_error("memset failed"); return MI355_ELAUNCH; }
  }
  if (n == 0) return MI355_OK;
  Table t = make_table(storage, C, num_scores);
  Table ovf = make_table(ovf_storage, ovf_capacity, num_scores);
  uint64_t* d_key = (uint64_t*)workspace;
  int64_t* d_score = (int64_t*)workspace + n;
  // main table, evicted records kept per key; its own slots are published before the overflow pass runs
  hipLaunchKernelGGL(table_insert_kernel, dim3(grid_for(n, 256 / G)), dim3(256), 0, stream, t, table_bucket_offsets,
                     bucket_sizes, counter, n, n_dev, (const uint64_t*)keys, table_ids, (const uint64_t*)score_in,
                     policy, timer_override, skip, indices, results, score_out, (unsigned long long*)nullptr,
                     (uint64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t*)nullptr, d_key, d_score);
  MI355_LAUNCH_CHECK();
  hipLaunchKernelGGL(table_unlock_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, t, table_bucket_offsets, n, n_dev,
                     (const uint64_t*)keys, table_ids, skip, indices, (const int64_t*)nullptr, (const int64_t*)nullptr, 0,
                     (int64_t*)nullptr);
  MI355_LAUNCH_CHECK();
  hipLaunchKernelGGL(table_insert_ovf_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, ovf, ovf_output_offsets,
                     ovf_bucket_sizes, ovf_counter, n, n_dev, (const uint64_t*)keys, table_ids,
                     (const uint64_t*)score_in, policy, timer_override, skip, indices, results, score_out, d_key, d_score,
                     (unsigned long long*)num_evicted, (uint64_t*)evicted_keys, evicted_indices, evicted_scores,
                     evicted_table_ids);
  MI355_LAUNCH_CHECK();
  hipLaunchKernelGGL(table_unlock_ovf_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, ovf, ovf_output_offsets, n,
                     n_dev, (const uint64_t*)keys, table_ids, skip, indices, results);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

// mi355_table_update_counter restricted to the keys whose flag equals `want` (fused forward: pin the slots the lookup
// found before the insert, release them / pin the new ones after it)
int mi355i_table_update_counter_where(int32_t* counter, int64_t counter_numel, const int64_t* slot_indices, int64_t n,
                                      const int64_t* n_dev, int32_t delta, const int64_t* table_ids,
                                      const int64_t* table_bucket_offsets, int64_t C, const uint8_t* flags, int want,
                                      hipStream_t stream) {
  if (n == 0) return MI355_OK;
  hipLaunchKernelGGL(update_counter_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, counter, counter_numel,
                     slot_indices, n, n_dev, delta, table_ids, table_bucket_offsets, C, flags, want);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int64_t mi355_table_export_batch_workspace_bytes(int64_t batch) {
  const int64_t a = (batch + 255) / 256 * 256;
  return a + 3 * 8 * a + 8 * a + mi355_flagged_compact_workspace_bytes(batch) + 256;
}

int mi355_table_export_batch(const void* storage, int64_t num_buckets, int64_t C, int64_t num_scores, int64_t batch,
                             int64_t offset, int has_threshold, uint64_t threshold, int64_t table_begin,
                             int64_t score_index, int64_t* counter, void* keys, int64_t* scores, int64_t* indices,
                             void* workspace, int64_t workspace_bytes, hipStream_t stream) {
  MI355_CHECK_ARG(C > 0 && C % 16 == 0, "bucket capacity must be a positive multiple of 16");
  MI355_CHECK_ARG(score_index >= 0 && score_index < num_scores, "score_index out of range");
  MI355_CHECK_ARG(offset >= 0 && offset + batch <= num_buckets * C, "Offset and batch size overflow.");
  MI355_CHECK_ARG(workspace_bytes >= mi355_table_export_batch_workspace_bytes(batch), "workspace too small");
  if (batch == 0) return MI355_OK;
  const int64_t a = (batch + 255) / 256 * 256;
  uint8_t* w = (uint8_t*)workspace;
  uint8_t* flags = w; w += a;
  uint64_t* k_tmp = (uint64_t*)w; w += 8 * a;
  uint64_t* s_tmp = (uint64_t*)w; w += 8 * a;
  int64_t* i_tmp = (int64_t*)w; w += 8 * a;
  int64_t* pos = (int64_t*)w; w += 8 * a;
  Table t = make_table(const_cast<void*>(storage), C, num_scores);
  hipLaunchKernelGGL(table_scan_kernel, dim3(grid_for(batch, 256)), dim3(256), 0, stream, t, offset, offset + batch,
                     table_begin, has_threshold, threshold, score_index, flags, k_tmp, s_tmp, i_tmp);
  MI355_LAUNCH_CHECK();
  const void* ins[3] = {k_tmp, s_tmp, i_tmp};
  void* outs[3] = {keys, scores, indices};
  return mi355_flagged_compact(flags, batch, nullptr, counter, pos, 3, ins, outs, w,
                               mi355_flagged_compact_workspace_bytes(batch), stream);
}

int mi355_table_count_matched(const void* storage, int64_t num_buckets, int64_t C, int64_t num_scores, uint64_t threshold,
                              int64_t begin, int64_t end, int64_t score_index, int64_t* num_matched, hipStream_t stream) {
  MI355_CHECK_ARG(C > 0 && C % 16 == 0, "bucket capacity must be a positive multiple of 16");
  MI355_CHECK_ARG(score_index >= 0 && score_index < num_scores, "score_index out of range");
  if (begin < 0) begin = 0;
  if (end < 0) end = num_buckets * C;
  if (hipMemsetAsync(num_matched, 0, 8, stream) != hipSuccess) { mi355_set_error("hipMemsetAsync failed"); return MI355_ELAUNCH; }
  if (end - begin <= 0) return MI355_OK;
  MI355_CHECK_ARG(end <= num_buckets * C, "range exceeds the table");
  Table t = make_table(const_cast<void*>(storage), C, num_scores);
  hipLaunchKernelGGL(table_count_kernel, dim3(grid_for(end - begin, 256, 2048)), dim3(256), 0, stream, t, begin, end,
                     threshold, score_index, (unsigned long long*)num_matched);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_table_score_blocks(int mode, const void* src_storage, int64_t src_C, int64_t src_bkt_begin, void* dst_storage,
                             int64_t dst_C, int64_t dst_bkt_begin, int64_t num_scores, int64_t n, const int64_t* src_slots,
                             const int64_t* dst_slots, int64_t* dense, hipStream_t stream) {
  MI355_CHECK_ARG(mode >= 0 && mode <= 2, "mode must be 0 copy / 1 gather / 2 scatter");
  if (n == 0) return MI355_OK;
  Table s = make_table(const_cast<void*>(src_storage ? src_storage : dst_storage), src_C > 0 ? src_C : dst_C, num_scores);
  Table d = make_table(dst_storage ? dst_storage : const_cast<void*>(src_storage), dst_C > 0 ? dst_C : src_C, num_scores);
  hipLaunchKernelGGL(score_blocks_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, mode, s, src_bkt_begin, d,
                     dst_bkt_begin, n, src_slots, dst_slots, (uint64_t*)dense);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

int mi355_device_timestamp(int64_t* out, hipStream_t stream) {
  hipLaunchKernelGGL(device_timestamp_kernel, dim3(1), dim3(1), 0, stream, out);
  MI355_LAUNCH_CHECK();
  return MI355_OK;
}

}  // extern "C"
