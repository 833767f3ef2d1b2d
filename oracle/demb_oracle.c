/*
 * demb_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Sequential CPU restatement of the integer/byte algorithms of the DynamicEmb
 * lookup path of NVIDIA/recsys-examples (the "scored hash table", segmented
 * unique, key routing).  It is the parity checker for the HIP kernels in
 * recsys-examples_amd/csrc; nothing in the product path may call it.
 *
 * Pinning status: the reference's native code for this path is CUDA-only
 * (nvcc + libcu++), cannot be compiled here, and ships no golden files, so
 * parity against the CUDA binaries themselves is UNPINNED.  What pins this
 * restatement (tests/test_oracle_*.py): the reference's own Python copy of the
 * hash (scored_hashtable.py:279-291) and empty-digest rule (:476-495), the
 * invariants of test_unique_op.py / test_table_operation.py, the 11-key fixture
 * of test_batched_dynamic_embedding_tables_v2.py:1517-1522 and the DEBUG
 * initializer closed forms (test/unit_tests/debug.py:157-224).
 *
 * All file:line citations are relative to /root/reference/corelib/dynamicemb/.
 *
 * Concurrency model restated here: the reference kernels run one thread per
 * key with a CAS lock on the key word.  Sequentially that becomes "process
 * keys in input order; a slot newly taken during this call is LOCKED (not
 * evictable, not matchable) until the call ends" -- exactly what
 * table_insert_kernel + table_unlock_kernel do (src/table_operation/
 * kernels.cuh:289-379,569-585).  With at most one key per bucket per call
 * (DEMB_DETERMINISM_MODE waves, scored_hashtable.py:1451-1558) the result is
 * schedule independent, which is the mode used for bit-exact parity.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EMPTY_KEY   UINT64_C(0xFFFFFFFFFFFFFFFF) /* types.cuh:117 */
#define LOCKED_KEY  UINT64_C(0xFFFFFFFFFFFFFFFD) /* types.cuh:118 */
#define RECLAIM_KEY UINT64_C(0xFFFFFFFFFFFFFFFE) /* types.cuh:119 */
#define RESERVE_MASK UINT64_C(0xFFFFFFFFFFFFFFFC) /* types.cuh:121 */

/* score.cuh:30-43 */
enum { POLICY_CONST = 0, POLICY_ASSIGN = 1, POLICY_ACCUMULATE = 2,
       POLICY_GLOBAL_TIMER = 3, POLICY_LRU_LFU = 4 };
/* types.cuh:52-61 */
enum { RES_INSERT = 0, RES_RECLAIM = 1, RES_ASSIGN = 2, RES_EVICT = 3,
       RES_DUPLICATED = 4, RES_BUSY = 5, RES_ILLEGAL = 6, RES_INIT = 7 };

/* types.cuh:123-131 (fmix64 of MurmurHash3), without the sign mask */
uint64_t orc_fmix64(uint64_t k) {
  k ^= k >> 33; k *= UINT64_C(0xff51afd7ed558ccd);
  k ^= k >> 33; k *= UINT64_C(0xc4ceb9fe1a85ec53);
  k ^= k >> 33;
  return k;
}
/* types.cuh:130  hash = fmix64 & INT64_MAX */
int64_t orc_hash(uint64_t key) { return (int64_t)(orc_fmix64(key) & INT64_MAX); }
/* types.cuh:207-210  digest = (uint8)(hash >> 32) */
uint8_t orc_digest(uint64_t key) { return (uint8_t)(orc_hash(key) >> 32); }
uint8_t orc_empty_digest(void) { return orc_digest(EMPTY_KEY); }
/* types.cuh:144-146 */
int orc_is_valid(uint64_t key) { return (key & RESERVE_MASK) != RESERVE_MASK; }

/* bucket view: SoA [keys u64 x C][digests u8 x C][scores u64 x C x ns]
 * (types.cuh:242-284); bucket stride (8+1+8*ns)*C (types.cuh:255-258). */
typedef struct { uint64_t *keys; uint8_t *dig; uint64_t *scores; } bucket_t;
static bucket_t bucket_at(uint8_t *storage, int64_t C, int64_t ns, int64_t b) {
  uint8_t *base = storage + (uint64_t)(9 + 8 * ns) * C * b;
  bucket_t r;
  r.keys = (uint64_t *)base;
  r.dig = base + 8 * C;
  r.scores = (uint64_t *)(base + 9 * C);
  return r;
}

/* scored_hashtable.py:476-495 _init_table */
void orc_table_init(uint8_t *storage, int64_t num_buckets, int64_t C, int64_t ns) {
  uint8_t ed = orc_empty_digest();
  for (int64_t b = 0; b < num_buckets; ++b) {
    bucket_t bk = bucket_at(storage, C, ns, b);
    for (int64_t i = 0; i < C; ++i) { bk.keys[i] = EMPTY_KEY; bk.dig[i] = ed; }
    for (int64_t i = 0; i < C * ns; ++i) bk.scores[i] = 0;
  }
}

/* bucket choice: kernels.cuh:107-125.  Returns 0 when the key has no table
 * (invalid key or zero-capacity table). */
static int locate(uint64_t key, int64_t tid, const int64_t *tbo, int64_t C,
                  int64_t *hash, int64_t *bkt_begin, int64_t *bucket_id) {
  if (!orc_is_valid(key)) return 0;
  *hash = orc_hash(key);
  *bkt_begin = tbo[tid];
  int64_t cap = (tbo[tid + 1] - tbo[tid]) * C;
  if (cap <= 0) return 0;
  *bucket_id = *bkt_begin + (*hash % cap) / C;
  return 1;
}

/* probe: types.cuh:308-396.  16-digest groups from align16(hash % C), inside a
 * group 4-byte vectors in order; per vector first the digest matches (key
 * compare), then the empty-digest slots (key == Empty).
 * returns 1 = Existed, 2 = Empty, 3 = Exhausted; *iter = slot. */
static int probe(bucket_t bk, int64_t C, uint64_t key, int64_t hash, int64_t *iter) {
  uint8_t d = (uint8_t)(hash >> 32), ed = orc_empty_digest();
  int64_t it = (hash % C) & ~INT64_C(15);
  for (int64_t step = 0; step < C; step += 16) {
    for (int v = 0; v < 4; ++v) {
      for (int o = 0; o < 4; ++o) {
        int64_t p = it + v * 4 + o;
        if (bk.dig[p] == d && bk.keys[p] == key) { *iter = p; return 1; }
      }
      for (int o = 0; o < 4; ++o) {
        int64_t p = it + v * 4 + o;
        if (bk.dig[p] == ed && bk.keys[p] == EMPTY_KEY) { *iter = p; return 2; }
      }
    }
    it = (it + 16) % C;
  }
  return 3;
}

/* score.cuh:51-63 ScorePolicy::get */
static uint64_t policy_get(int policy, const uint64_t *score_in, int64_t i, uint64_t timer) {
  if (policy == POLICY_CONST) return 0;
  if (policy == POLICY_GLOBAL_TIMER) return timer;
  return score_in[i];
}
/* score.cuh:72-96 ScorePolicy::update; `s` = first score word of the slot */
static uint64_t policy_update(int policy, uint64_t *s, uint64_t score, uint64_t timer) {
  switch (policy) {
  case POLICY_CONST: return *s;
  case POLICY_ACCUMULATE: score += *s; *s = score; return score;
  case POLICY_LRU_LFU: s[0] = timer; score += s[1]; s[1] = score; return score;
  default: *s = score; return score;
  }
}

/* table_lookup_kernel: kernels.cuh:81-187 (main table; overflow buckets are
 * restated in orc_table_lookup_ovf below). */
void orc_table_lookup(uint8_t *storage, const int64_t *tbo, int64_t C, int64_t ns,
                      int64_t n, const uint64_t *keys, const int64_t *table_ids,
                      const uint64_t *score_in, int policy, uint64_t timer,
                      int64_t *score_out, uint8_t *founds, int64_t *indices) {
  for (int64_t i = 0; i < n; ++i) {
    uint64_t key = keys[i];
    uint64_t score = policy_get(policy, score_in, i, timer);
    int64_t hash = 0, bb = 0, b = 0;
    if (!locate(key, table_ids[i], tbo, C, &hash, &bb, &b)) {
      score_out[i] = (int64_t)score; founds[i] = 0; indices[i] = -1; continue;
    }
    bucket_t bk = bucket_at(storage, C, ns, b);
    int64_t it = 0;
    int found = probe(bk, C, key, hash, &it) == 1;
    int64_t index = -1;
    if (found) {
      if (policy == POLICY_CONST) score = bk.scores[it * ns + (ns - 1)];
      else score = policy_update(policy, bk.scores + it * ns, score, timer);
      index = (b - bb) * C + it;
    }
    score_out[i] = (int64_t)score; founds[i] = (uint8_t)found; indices[i] = index;
  }
}

/* reduce: types.cuh:398-512.  Ascending scan from slot 0, strict '<' against
 * the running best (starts at UINT64_MAX = score_for_compare, score.cuh:65-68),
 * candidates are non-locked, non-empty keys whose ref-counter is 0.  `locked`
 * marks slots taken earlier in this call. */
static int reduce_min(bucket_t bk, int64_t C, int64_t ns, const uint8_t *locked,
                      const int32_t *counter, int64_t *slot, uint64_t *ekey, uint64_t *escore) {
  uint64_t best = UINT64_MAX; int ok = 0;
  for (int64_t p = 0; p < C; ++p) {
    uint64_t s = bk.scores[p * ns + (ns - 1)];
    if (s < best) {
      uint64_t k = bk.keys[p];
      if (locked[p] || k == EMPTY_KEY) continue;
      if (counter && counter[p] > 0) continue;
      *slot = p; *ekey = k; best = s; ok = 1;
    }
  }
  *escore = best;
  return ok;
}

/* One key of table_insert_kernel / table_insert_and_evict_kernel (kernels.cuh:189-369, 389-466): probe, take an
 * Empty slot, or evict the minimum score among the unpinned slots.  Returns the InsertResult; *index = table-relative
 * slot on success; (*ekey, *escore) = the evicted record (Evict) or the key itself and its input score (Busy). */
static int insert_one(uint8_t *storage, const int64_t *tbo, int64_t C, int64_t ns, int32_t *bucket_sizes,
                      int32_t *counter, uint8_t *lock_scratch, uint64_t key, int64_t tid, uint64_t *score,
                      int policy, uint64_t timer, int64_t *index, uint64_t *ekey, uint64_t *escore) {
  int64_t hash = 0, bb = 0, b = 0;
  int res = RES_ILLEGAL;
  *index = -1; *ekey = 0; *escore = 0;
  if (!locate(key, tid, tbo, C, &hash, &bb, &b)) return res;
  bucket_t bk = bucket_at(storage, C, ns, b);
  uint8_t *lk = lock_scratch + b * C;
  int64_t it = 0;
  int pr = probe(bk, C, key, hash, &it);
  res = RES_INIT;
  if (pr == 1) { res = RES_ASSIGN; lk[it] = 1; }           /* kernels.cuh:201-207 */
  else if (pr == 2) {                                        /* kernels.cuh:208-216 */
    lk[it] = 1; bk.keys[it] = LOCKED_KEY; bk.dig[it] = (uint8_t)(hash >> 32);
    bucket_sizes[b] += 1; res = RES_INSERT;
  } else {                                                   /* kernels.cuh:226-287 */
    /* ref-counter: one int32 per slot of the whole arena, table t's region
     * starting at tbo[t]*C (update_counter_with_layout_kernel,
     * insert_and_evict.cu:27-58).  NOTE: the reference's insert kernels read
     * counter[(bucket_id - bkt_begin)*C + iter] (kernels.cuh:355,451), which
     * for table_id > 0 does not match the layout its own update kernel
     * writes; both agree for a single table.  This build uses the update
     * kernel's layout everywhere (counter[bucket_id*C + iter]). */
    const int32_t *c0 = counter ? counter + b * C : NULL;
    int64_t slot = 0;
    if (reduce_min(bk, C, ns, lk, c0, &slot, ekey, escore)) {
      it = slot; lk[it] = 1; bk.keys[it] = LOCKED_KEY;
      bk.dig[it] = (uint8_t)(hash >> 32);
      if (*ekey == RECLAIM_KEY) { bucket_sizes[b] += 1; res = RES_RECLAIM; }
      else {
        for (int64_t s = 0; s < ns; ++s) bk.scores[it * ns + s] = 0;
        res = RES_EVICT;
      }
    } else {
      res = RES_BUSY; *ekey = key; *escore = *score;          /* kernels.cuh:277-282 */
    }
  }
  if (res <= RES_EVICT) {                                    /* kernels.cuh:363-369 */
    *score = policy_update(policy, bk.scores + it * ns, *score, timer);
    *index = (b - bb) * C + it;
  }
  return res;
}

/* table_unlock_kernel: kernels.cuh:569-585 (main-table slots) */
static void unlock_main(uint8_t *storage, const int64_t *tbo, int64_t C, int64_t ns, uint8_t *lock_scratch,
                        int64_t n, const uint64_t *keys, const int64_t *table_ids, const int64_t *indices,
                        const int64_t *main_cap) {
  for (int64_t i = 0; i < n; ++i) {
    if (indices[i] < 0) continue;
    if (main_cap && indices[i] >= main_cap[table_ids[i]]) continue;   /* overflow slot */
    int64_t bb = tbo[table_ids[i]];
    int64_t b = bb + indices[i] / C, it = indices[i] % C;
    bucket_at(storage, C, ns, b).keys[it] = keys[i];
    lock_scratch[b * C + it] = 0;
  }
}

/* table_insert_kernel / table_insert_and_evict_kernel + table_unlock_kernel:
 * kernels.cuh:189-585.  Evicted streams may be NULL (plain insert).
 * `lock_scratch` : num_buckets_total * C bytes, zero on entry, zero on exit. */
void orc_table_insert(uint8_t *storage, const int64_t *tbo, int64_t C, int64_t ns,
                      int32_t *bucket_sizes, int32_t *counter, uint8_t *lock_scratch,
                      int64_t n, const uint64_t *keys, const int64_t *table_ids,
                      const uint64_t *score_in, int policy, uint64_t timer,
                      int64_t *indices, uint8_t *results, int64_t *score_out,
                      int64_t *num_evicted, uint64_t *ev_keys, int64_t *ev_indices,
                      int64_t *ev_scores, int64_t *ev_table_ids) {
  int64_t nev = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t score = policy_get(policy, score_in, i, timer), ekey = 0, escore = 0;
    int64_t index = -1;
    int res = insert_one(storage, tbo, C, ns, bucket_sizes, counter, lock_scratch, keys[i], table_ids[i], &score,
                         policy, timer, &index, &ekey, &escore);
    if (ev_keys && (res == RES_EVICT || res == RES_BUSY)) {   /* kernels.cuh:522-556 */
      ev_keys[nev] = ekey; ev_scores[nev] = (int64_t)escore;
      ev_indices[nev] = res == RES_EVICT ? index : -(i + 1); ev_table_ids[nev] = table_ids[i]; ++nev;
    }
    indices[i] = index;
    if (results) results[i] = (uint8_t)res;
    if (score_out) score_out[i] = (int64_t)score;
  }
  unlock_main(storage, tbo, C, ns, lock_scratch, n, keys, table_ids, indices, NULL);
  if (num_evicted) *num_evicted = nev;
}

/* ---- overflow region (scored_hashtable.py:426-474): one bucket of `ocap` slots per logical table in a second arena;
 * table-relative index of an overflow entry = out_off[t] + position. */

/* table_lookup_kernel with EnableOverflow (kernels.cuh:153-183) + overflow_find (:711-736) */
void orc_table_lookup_ovf(uint8_t *storage, const int64_t *tbo, int64_t C, int64_t ns,
                          uint8_t *ovf_storage, int64_t ocap, const int64_t *out_off,
                          int64_t n, const uint64_t *keys, const int64_t *table_ids,
                          const uint64_t *score_in, int policy, uint64_t timer,
                          int64_t *score_out, uint8_t *founds, int64_t *indices) {
  orc_table_lookup(storage, tbo, C, ns, n, keys, table_ids, score_in, policy, timer, score_out, founds, indices);
  for (int64_t i = 0; i < n; ++i) {
    if (founds[i] || !orc_is_valid(keys[i])) continue;
    int64_t t = table_ids[i], hash = orc_hash(keys[i]);
    bucket_t ob = bucket_at(ovf_storage, ocap, ns, t);
    for (int64_t scan = 0; scan < ocap; ++scan) {
      int64_t pos = (hash % ocap + scan) % ocap;
      uint64_t k = ob.keys[pos];
      if (k == keys[i]) {
        uint64_t score = policy_get(policy, score_in, i, timer);
        if (policy == POLICY_CONST) score = ob.scores[pos * ns + (ns - 1)];
        else score = policy_update(policy, ob.scores + pos * ns, score, timer);
        score_out[i] = (int64_t)score; founds[i] = 1; indices[i] = pos + out_off[t];
        break;
      }
      if (k == EMPTY_KEY) break;
    }
  }
}

/* table_insert_and_evict_kernel with UseOverflow (kernels.cuh:468-566) + overflow_insert_and_evict (:738-800).
 * `ovf_lock` : T * ocap bytes, zero on entry and exit (slots taken in this call stay Locked until the unlock pass). */
void orc_table_insert_ovf(uint8_t *storage, const int64_t *tbo, int64_t C, int64_t ns,
                          int32_t *bucket_sizes, int32_t *counter, uint8_t *lock_scratch,
                          uint8_t *ovf_storage, int64_t ocap, int32_t *ovf_sizes, int32_t *ovf_counter,
                          const int64_t *out_off, uint8_t *ovf_lock,
                          int64_t n, const uint64_t *keys, const int64_t *table_ids,
                          const uint64_t *score_in, int policy, uint64_t timer,
                          int64_t *indices, uint8_t *results, int64_t *score_out,
                          int64_t *num_evicted, uint64_t *ev_keys, int64_t *ev_indices,
                          int64_t *ev_scores, int64_t *ev_table_ids) {
  int64_t nev = 0;
  for (int64_t i = 0; i < n; ++i) {
    uint64_t key = keys[i];
    int64_t t = table_ids[i];
    uint64_t score = policy_get(policy, score_in, i, timer), ekey = 0, escore = 0;
    int64_t index = -1;
    int res = insert_one(storage, tbo, C, ns, bucket_sizes, counter, lock_scratch, key, t, &score,
                         policy, timer, &index, &ekey, &escore);
    uint64_t f_key = ekey; int64_t f_index = res == RES_EVICT ? index : -(i + 1);
    if (res == RES_BUSY && orc_is_valid(key)) {
      int64_t hash = orc_hash(key);
      bucket_t ob = bucket_at(ovf_storage, ocap, ns, t);
      uint8_t *lk = ovf_lock + t * ocap;
      int32_t *cnt = ovf_counter + t * ocap;
      int ores = RES_BUSY; int64_t at = -1; uint64_t okey = 0;
      for (int64_t scan = 0; scan < ocap; ++scan) {
        int64_t pos = (hash % ocap + scan) % ocap;
        if (lk[pos]) continue;                                   /* LockedKey */
        uint64_t k = ob.keys[pos];
        if (k == key) { at = pos; ores = RES_ASSIGN; break; }
        if (k == EMPTY_KEY) {
          lk[pos] = 1; ob.dig[pos] = (uint8_t)(hash >> 32); ovf_sizes[t] += 1;
          at = pos; ores = RES_INSERT; break;
        }
        if (k == RECLAIM_KEY) continue;
        if (cnt[pos] == 0) {
          lk[pos] = 1; ob.dig[pos] = (uint8_t)(hash >> 32);
          okey = k; at = pos; ores = RES_EVICT; break;
        }
      }
      if (at >= 0) {
        index = at + out_off[t];
        res = ores;
        score = policy_get(policy, score_in, i, timer);
        if (ores == RES_ASSIGN) score = policy_update(policy, ob.scores + at * ns, score, timer);
        else {
          for (int64_t s = 0; s < ns; ++s) ob.scores[at * ns + s] = 0;
          score = policy_update(policy, ob.scores + at * ns, score, timer);
          ob.keys[at] = LOCKED_KEY;
          if (ores == RES_EVICT) { f_key = okey; f_index = index; }
        }
      }
    }
    if (ev_keys && (res == RES_EVICT || res == RES_BUSY)) {
      ev_keys[nev] = f_key; ev_scores[nev] = (int64_t)escore;
      ev_indices[nev] = f_index; ev_table_ids[nev] = t; ++nev;
    }
    indices[i] = index;
    if (results) results[i] = (uint8_t)res;
    if (score_out) score_out[i] = (int64_t)score;
  }
  unlock_main(storage, tbo, C, ns, lock_scratch, n, keys, table_ids, indices, out_off);
  for (int64_t i = 0; i < n; ++i) {                              /* overflow slots taken above */
    int64_t t = table_ids[i];
    if (indices[i] < out_off[t]) continue;
    int64_t pos = indices[i] - out_off[t];
    if (ovf_lock[t * ocap + pos]) {
      bucket_at(ovf_storage, ocap, ns, t).keys[pos] = keys[i];
      ovf_lock[t * ocap + pos] = 0;
    }
  }
  if (num_evicted) *num_evicted = nev;
}

/* table_erase_kernel: kernels.cuh:587-652 */
void orc_table_erase(uint8_t *storage, const int64_t *tbo, int64_t C, int64_t ns,
                     int32_t *bucket_sizes, int64_t n, const uint64_t *keys,
                     const int64_t *table_ids, int64_t *indices) {
  for (int64_t i = 0; i < n; ++i) {
    int64_t hash = 0, bb = 0, b = 0, index = -1;
    if (locate(keys[i], table_ids[i], tbo, C, &hash, &bb, &b)) {
      bucket_t bk = bucket_at(storage, C, ns, b);
      int64_t it = 0;
      if (probe(bk, C, keys[i], hash, &it) == 1) {
        bk.scores[it * ns] = 0; bk.dig[it] = orc_empty_digest();
        bk.keys[it] = RECLAIM_KEY; bucket_sizes[b] -= 1;
        index = (b - bb) * C + it;
      }
    }
    if (indices) indices[i] = index;
  }
}

/* bucketize_keys: src/table_operation/bucketize.cu:40-116,121-260.  Stable sort
 * by (bucket_id, key) with the key compared as signed int64 (CUB radix sort on a
 * (segment_id, key) tuple of int64), `inverse` = original position. Returns the
 * number of non-empty buckets; offsets has that many + 1 entries. */
typedef struct { int64_t seg; int64_t key; int64_t pos; } bk_item_t;
static int bk_cmp(const void *a, const void *b) {
  const bk_item_t *x = a, *y = b;
  if (x->seg != y->seg) return x->seg < y->seg ? -1 : 1;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos);
}
int64_t orc_bucketize_keys(const int64_t *tbo, int64_t C, int64_t n, const uint64_t *keys,
                           const int64_t *table_ids, uint64_t *keys_out,
                           int64_t *offsets, int64_t *inverse) {
  bk_item_t *it = malloc(sizeof(bk_item_t) * (n ? n : 1));
  for (int64_t i = 0; i < n; ++i) {
    int64_t h = orc_hash(keys[i]);
    int64_t bb = tbo[table_ids[i]];
    int64_t cap = (tbo[table_ids[i] + 1] - bb) * C;
    it[i].seg = cap == 0 ? bb : bb + (int64_t)((uint64_t)h % (uint64_t)cap) / C;
    it[i].key = (int64_t)keys[i]; it[i].pos = i;
  }
  qsort(it, n, sizeof(bk_item_t), bk_cmp);
  int64_t nb = 0; offsets[0] = 0;
  for (int64_t i = 0; i < n; ++i) {
    keys_out[i] = (uint64_t)it[i].key; inverse[i] = it[i].pos;
    if (i + 1 == n || it[i + 1].seg != it[i].seg) offsets[++nb] = i + 1;
  }
  free(it);
  return nb;
}

/* segmented_unique contract: src/unique_op.cu:250-462, test/test_unique_op.py:81-150.
 * The reference's order of unique keys inside a table is schedule dependent; the
 * contract is (a) unique_keys[output_indices[i]] == keys[i], (b) uniques grouped
 * by table with table_offsets, (c) freq sums.  This build fixes the free choice
 * to FIRST-OCCURRENCE order (deterministic), which satisfies (a)-(c).
 * in_freq == NULL && count_freq -> each key counts 1 (unique_op.cu:510-513). */
typedef struct { uint64_t key; int64_t pos; } uq_item_t;
static int uq_cmp(const void *a, const void *b) {
  const uq_item_t *x = a, *y = b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->pos < y->pos ? -1 : (x->pos > y->pos);
}
void orc_segmented_unique(int64_t n, const uint64_t *keys, const int64_t *seg, int64_t T,
                          const int64_t *in_freq, int count_freq, uint64_t *unique_keys,
                          int64_t *output_indices, int64_t *table_offsets, int64_t *freq) {
  int64_t nu = 0;
  uq_item_t *tmp = malloc(sizeof(uq_item_t) * (n ? n : 1));
  int64_t *first = malloc(sizeof(int64_t) * (n ? n : 1));
  for (int64_t t = 0; t < T; ++t) {
    table_offsets[t] = nu;
    int64_t lo = seg[t], hi = seg[t + 1], m = hi - lo;
    for (int64_t i = 0; i < m; ++i) { tmp[i].key = keys[lo + i]; tmp[i].pos = lo + i; }
    qsort(tmp, m, sizeof(uq_item_t), uq_cmp);
    /* first[i] = first occurrence position of keys[i] */
    for (int64_t i = 0; i < m;) {
      int64_t j = i; while (j < m && tmp[j].key == tmp[i].key) { first[tmp[j].pos] = tmp[i].pos; ++j; }
      i = j;
    }
    for (int64_t i = lo; i < hi; ++i) {
      if (first[i] == i) {
        unique_keys[nu] = keys[i]; output_indices[i] = nu;
        if (count_freq) freq[nu] = 0;
        ++nu;
      } else output_indices[i] = output_indices[first[i]];
      if (count_freq) freq[output_indices[i]] += in_freq ? in_freq[i] : 1;
    }
  }
  table_offsets[T] = nu;
  free(tmp); free(first);
}

/* expand_table_ids_kernel: unique_op.cu:471-480 (upper_bound(offsets, i) - 1) */
void orc_expand_table_ids(const int64_t *offsets, int64_t T, int64_t n, int64_t *table_ids) {
  int64_t t = 0;
  for (int64_t i = 0; i < n; ++i) { while (t + 1 <= T && offsets[t + 1] <= i) ++t; table_ids[i] = t; }
}

/* get_table_range_kernel: index_calculation.cu:78-91 range[t] = offsets[feature_offsets[t]*B] */
void orc_get_table_range(const int64_t *offsets, const int64_t *feature_offsets, int64_t T,
                         int64_t B, int64_t *range) {
  for (int64_t t = 0; t <= T; ++t) range[t] = offsets[feature_offsets[t] * B];
}

/* block_bucketize_sparse_features, row-wise routing before the all-to-all:
 * src/sparse_block_bucketize_features.cu:31-38 (hash_key = fmix64),:296-350.
 * dist_type 0 continuous (p = idx / blk, new = idx % blk), 1 roundrobin
 * (p = idx % W, new = idx), 2 hash_roundrobin (p = fmix64(idx) % W, new = idx).
 * lengths is [F*B] feature-major; new_lengths [W*F*B]; new_indices grouped by
 * rank then bag, order inside a bag preserved (sequence mode);
 * unbucketize_permute[j] = position of original key j in the bucketized stream. */
void orc_block_bucketize(int64_t W, int64_t FB, int64_t B, const int64_t *offsets,
                         const uint64_t *indices, const int64_t *block_sizes, int dist_type,
                         int64_t *new_lengths, int64_t *new_offsets, uint64_t *new_indices,
                         int64_t *unbucketize_permute) {
  for (int64_t i = 0; i < W * FB; ++i) new_lengths[i] = 0;
  for (int64_t bag = 0; bag < FB; ++bag) {
    int64_t f = bag / B;
    for (int64_t j = offsets[bag]; j < offsets[bag + 1]; ++j) {
      uint64_t idx = indices[j]; int64_t p;
      if (dist_type == 0) { uint64_t blk = (uint64_t)block_sizes[f]; p = idx < blk * (uint64_t)W ? (int64_t)(idx / blk) : (int64_t)(idx % (uint64_t)W); }
      else if (dist_type == 1) p = (int64_t)(idx % (uint64_t)W);
      else p = (int64_t)(orc_fmix64(idx) % (uint64_t)W);
      new_lengths[p * FB + bag] += 1;
    }
  }
  new_offsets[0] = 0;
  for (int64_t i = 0; i < W * FB; ++i) new_offsets[i + 1] = new_offsets[i] + new_lengths[i];
  int64_t *cur = malloc(sizeof(int64_t) * (W * FB ? W * FB : 1));
  memcpy(cur, new_offsets, sizeof(int64_t) * W * FB);
  for (int64_t bag = 0; bag < FB; ++bag) {
    int64_t f = bag / B;
    for (int64_t j = offsets[bag]; j < offsets[bag + 1]; ++j) {
      uint64_t idx = indices[j]; int64_t p; uint64_t nw = idx;
      if (dist_type == 0) { uint64_t blk = (uint64_t)block_sizes[f];
        if (idx < blk * (uint64_t)W) { p = (int64_t)(idx / blk); nw = idx % blk; }
        else { p = (int64_t)(idx % (uint64_t)W); nw = idx / (uint64_t)W; } }
      else if (dist_type == 1) p = (int64_t)(idx % (uint64_t)W);
      else p = (int64_t)(orc_fmix64(idx) % (uint64_t)W);
      int64_t dst = cur[p * FB + bag]++;
      new_indices[dst] = nw;
      if (unbucketize_permute) unbucketize_permute[j] = dst;
    }
  }
  free(cur);
}

/* The same op with FBGEMM's two generalisations (sparse_block_bucketize_features.cu:194-211, 228-292, 320-347):
 * bag_feature [FB] (nullable) = the feature of every bag when the features have DIFFERENT batch sizes (the reference's
 * length_to_feature_idx); pos_concat / pos_offsets [F + 1] (nullable) = uneven shard boundaries per feature: lb = (index of the
 * first boundary > idx) - 1, rank = lb < W ? lb : idx % W, new index = lb < W ? idx - boundary[lb] : idx / W (the dist types do
 * not apply then).  dist_type per feature [F] (nullable: 0). */
void orc_block_bucketize_ex(int64_t W, int64_t FB, int64_t B, const int64_t *offsets, const uint64_t *indices,
                            const int64_t *block_sizes, const int32_t *dist_type, const int64_t *bag_feature,
                            const int64_t *pos_concat, const int64_t *pos_offsets,
                            int64_t *new_lengths, int64_t *new_offsets, uint64_t *new_indices, int64_t *unbucketize_permute) {
  for (int64_t i = 0; i < W * FB; ++i) new_lengths[i] = 0;
  int64_t *cur = malloc(sizeof(int64_t) * (W * FB ? W * FB : 1));
  for (int pass = 0; pass < 2; ++pass) {
    for (int64_t bag = 0; bag < FB; ++bag) {
      int64_t f = bag_feature ? bag_feature[bag] : bag / B;
      int dist = dist_type ? dist_type[f] : 0;
      for (int64_t j = offsets[bag]; j < offsets[bag + 1]; ++j) {
        uint64_t idx = indices[j], nw = idx; int64_t p;
        if (pos_concat) {
          int64_t first = pos_offsets[f], last = pos_offsets[f + 1];
          while (first < last) { int64_t mid = first + (last - first) / 2; if ((uint64_t)pos_concat[mid] <= idx) first = mid + 1; else last = mid; }
          uint64_t lb = (uint64_t)(first - pos_offsets[f] - 1);
          if (lb < (uint64_t)W) { p = (int64_t)lb; nw = idx - (uint64_t)pos_concat[pos_offsets[f] + (int64_t)lb]; }
          else { p = (int64_t)(idx % (uint64_t)W); nw = idx / (uint64_t)W; }
        } else if (dist == 0) { uint64_t blk = (uint64_t)block_sizes[f];
          if (idx < blk * (uint64_t)W) { p = (int64_t)(idx / blk); nw = idx % blk; }
          else { p = (int64_t)(idx % (uint64_t)W); nw = idx / (uint64_t)W; } }
        else if (dist == 1) p = (int64_t)(idx % (uint64_t)W);
        else p = (int64_t)(orc_fmix64(idx) % (uint64_t)W);
        if (pass == 0) new_lengths[p * FB + bag] += 1;
        else { int64_t dst = cur[p * FB + bag]++; new_indices[dst] = nw; if (unbucketize_permute) unbucketize_permute[j] = dst; }
      }
    }
    if (pass == 0) {
      new_offsets[0] = 0;
      for (int64_t i = 0; i < W * FB; ++i) new_offsets[i + 1] = new_offsets[i] + new_lengths[i];
      memcpy(cur, new_offsets, sizeof(int64_t) * W * FB);
    }
  }
  free(cur);
}

/* initializer DEBUG mode: src/initializer.cuh:158-176, dynamicemb_config.py:45:
 * every element of the row = float(key % 100000). */
void orc_debug_init(int64_t n, int64_t dim, int64_t stride, const uint64_t *keys, float *rows) {
  for (int64_t i = 0; i < n; ++i)
    for (int64_t d = 0; d < dim; ++d) rows[i * stride + d] = (float)(keys[i] % UINT64_C(100000));
}
