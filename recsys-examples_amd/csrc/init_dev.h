// Counter-based first-touch initialisers shared by value_ops.hip and fused_fwd.hip (initializer.cuh:158-176 of the reference
// for the DEBUG mode; the random modes are Philox4x32-10 keyed by (seed, row key, element)).
#pragma once
#include "common.h"

namespace mi355 {

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

}  // namespace mi355
