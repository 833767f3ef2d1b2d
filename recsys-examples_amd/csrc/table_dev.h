// Device-side pieces of the scored hash table shared by table.hip and fused_fwd.hip: layout, hash / digest,
// 16-byte digest-vector match, 8-lane group probe, score policies (types.cuh:88-396, score.cuh:30-99 of the reference).
#pragma once
#include "common.h"

namespace mi355 {

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr uint64_t kLockedKey = 0xFFFFFFFFFFFFFFFDull;
constexpr uint64_t kReclaimKey = 0xFFFFFFFFFFFFFFFEull;
constexpr uint64_t kReserveMask = 0xFFFFFFFFFFFFFFFCull;

enum Policy : int { kConst = 0, kAssign = 1, kAccumulate = 2, kGlobalTimer = 3, kLruLfu = 4 };
enum Result : uint8_t { kInsert = 0, kReclaim = 1, kAssigned = 2, kEvict = 3, kDuplicated = 4,
                        kBusy = 5, kIllegal = 6, kInit = 7 };

constexpr int G = 8;  // lanes per key

struct Table {
  uint8_t* storage;
  int64_t C;       // slots per bucket, multiple of 16
  int64_t ns;      // score words per slot
  int64_t stride;  // bytes per bucket = (9 + 8 ns) C
  __device__ __forceinline__ uint64_t* keys(int64_t b) const { return (uint64_t*)(storage + b * stride); }
  __device__ __forceinline__ uint8_t* dig(int64_t b) const { return storage + b * stride + 8 * C; }
  __device__ __forceinline__ uint64_t* scores(int64_t b) const { return (uint64_t*)(storage + b * stride + 9 * C); }
};

__device__ __forceinline__ bool is_valid(uint64_t k) { return (k & kReserveMask) != kReserveMask; }
__device__ __forceinline__ uint8_t digest_of(int64_t h) { return (uint8_t)(h >> 32); }

// 16-bit mask of the bytes of v equal to d (exact zero-byte detection, no false positives)
__device__ __forceinline__ uint32_t eq_mask16(uint4 v, uint32_t d) {
  const uint32_t s = d * 0x01010101u;
  uint32_t w[4] = {v.x ^ s, v.y ^ s, v.z ^ s, v.w ^ s};
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t x = w[i];
    uint32_t t = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);  // 0x80 where byte == 0
    uint32_t nib = (((t >> 7) * 0x00204081u) >> 21) & 0xFu;
    m |= nib << (4 * i);
  }
  return m;
}

__device__ __forceinline__ uint4 load_dig16(const uint8_t* p, bool fresh) {
  if (!fresh) return *reinterpret_cast<const uint4*>(p);
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
  uint4 r;
  r.x = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.z = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  r.w = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return r;
}

// value of `v` held by lane `src` of this lane's 8-lane group
template <typename T>
__device__ __forceinline__ T group_bcast(T v, int src_in_group) {
  int src = (lane_id() & ~(G - 1)) | src_in_group;
  if constexpr (sizeof(T) == 8) {
    uint64_t u = (uint64_t)v;
    uint32_t lo = __shfl((int)(uint32_t)u, src, 64), hi = __shfl((int)(uint32_t)(u >> 32), src, 64);
    return (T)(((uint64_t)hi << 32) | lo);
  } else {
    return (T)__shfl((int)v, src, 64);
  }
}
// 8-bit mask of the group's lanes with pred set (control flow must be group-uniform)
__device__ __forceinline__ uint32_t group_ballot(bool pred) {
  uint64_t b = __ballot(pred);
  return (uint32_t)(b >> (lane_id() & ~(G - 1))) & 0xFFu;
}

struct Located {
  int64_t hash, bkt_begin, bucket;
  bool ok;
};
// bucket choice (kernels.cuh:107-125)
__device__ __forceinline__ Located locate(uint64_t key, int64_t tid, const int64_t* __restrict__ tbo, int64_t C) {
  Located r{0, 0, 0, false};
  if (!is_valid(key)) return r;
  r.hash = (int64_t)(fmix64(key) & 0x7FFFFFFFFFFFFFFFull);
  r.bkt_begin = tbo[tid];
  int64_t cap = (tbo[tid + 1] - r.bkt_begin) * C;
  if (cap <= 0) return r;
  uint64_t local = (uint64_t)r.hash % (uint64_t)cap;
  r.bucket = r.bkt_begin + (int64_t)(local / (uint64_t)C);
  r.ok = true;
  return r;
}

// Group-cooperative probe of one bucket (types.cuh:308-396 semantics: first slot in probe
// order -- 16-aligned start, wrap around -- that holds `key`; if none, the first Empty slot).
// Returns via reference: found_slot (>=0 or -1), empty_slot (>=0 or -1).  All 8 lanes of the
// group call with the same arguments; results are group-uniform.
__device__ __forceinline__ void group_probe(const Table& t, int64_t b, uint64_t key, int64_t hash, bool fresh,
                                            bool want_empty, int& found_slot, int& empty_slot) {
  const int g = lane_id() & (G - 1);
  const int C = (int)t.C;
  const uint32_t d = digest_of(hash);
  const uint32_t ed = digest_of((int64_t)(fmix64(kEmptyKey) & 0x7FFFFFFFFFFFFFFFull));
  const int start = (int)(((C & (C - 1)) == 0 ? ((uint64_t)hash & (uint64_t)(C - 1)) : ((uint64_t)hash % (uint64_t)C))) & ~15;
  const uint8_t* dg = t.dig(b);
  const uint64_t* ks = t.keys(b);
  found_slot = -1;
  empty_slot = -1;
  for (int chunk = 0; chunk < C; chunk += 16 * G) {
    const int rank0 = chunk + g * 16;
    const bool in = rank0 < C;
    int p0 = start + rank0;
    if (p0 >= C) p0 -= C;
    int my_found = -1, my_empty = -1;
    if (in) {
      uint4 dv = load_dig16(dg + p0, fresh);
      uint32_t m = eq_mask16(dv, d);
      while (m) {
        int bit = __ffs(m) - 1;
        m &= m - 1;
        uint64_t k = fresh ? ald64(ks + p0 + bit) : ks[p0 + bit];
        if (k == key) { my_found = p0 + bit; break; }
      }
      if (want_empty && my_found < 0) {
        uint32_t me = eq_mask16(dv, ed);
        while (me) {
          int bit = __ffs(me) - 1;
          me &= me - 1;
          uint64_t k = fresh ? ald64(ks + p0 + bit) : ks[p0 + bit];
          if (k == kEmptyKey) { my_empty = p0 + bit; break; }
        }
      }
    }
    uint32_t fm = group_ballot(my_found >= 0);
    if (fm) { found_slot = group_bcast(my_found, __ffs(fm) - 1); return; }
    if (want_empty && empty_slot < 0) {
      uint32_t em = group_ballot(my_empty >= 0);
      if (em) empty_slot = group_bcast(my_empty, __ffs(em) - 1);  // lowest lane == lowest probe rank
    }
    // a key can never sit behind an Empty slot in probe order (slots never return to Empty),
    // so once an Empty slot is known the key is absent
    if (want_empty && empty_slot >= 0) return;
  }
}

__device__ __forceinline__ uint64_t policy_get(int policy, const uint64_t* score_in, int64_t i, uint64_t timer) {
  if (policy == kConst) return 0;
  if (policy == kGlobalTimer) return timer;
  return score_in[i];
}
// score.cuh:72-96; s = first score word of the slot
__device__ __forceinline__ uint64_t policy_update(int policy, uint64_t* s, uint64_t score, uint64_t timer) {
  switch (policy) {
    case kConst: return ald64(s);
    case kAccumulate: score += ald64(s); ast64(s, score); return score;
    case kLruLfu: ast64(s, timer); score += ald64(s + 1); ast64(s + 1, score); return score;
    default: ast64(s, score); return score;
  }
}

// group arg-min over (score, slot): smaller score wins, ties -> lower slot
__device__ __forceinline__ void group_argmin(uint64_t& s, int& slot, uint64_t& k) {
#pragma unroll
  for (int off = 1; off < G; off <<= 1) {
    int src = lane_id() ^ off;
    uint32_t lo = __shfl((int)(uint32_t)s, src, 64), hi = __shfl((int)(uint32_t)(s >> 32), src, 64);
    uint64_t os = ((uint64_t)hi << 32) | lo;
    int oslot = __shfl(slot, src, 64);
    uint32_t klo = __shfl((int)(uint32_t)k, src, 64), khi = __shfl((int)(uint32_t)(k >> 32), src, 64);
    uint64_t ok = ((uint64_t)khi << 32) | klo;
    bool take = (oslot >= 0) && (slot < 0 || os < s || (os == s && oslot < slot));
    if (take) { s = os; slot = oslot; k = ok; }
  }
}

static inline Table make_table(void* storage, int64_t C, int64_t ns) {
  Table t;
  t.storage = (uint8_t*)storage;
  t.C = C;
  t.ns = ns;
  t.stride = (9 + 8 * ns) * C;
  return t;
}

}  // namespace mi355
