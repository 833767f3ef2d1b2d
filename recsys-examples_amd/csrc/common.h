// Shared device/host helpers for the gfx950 kernels of the DynamicEmb / HSTU hot path.
// gfx950 only: wave = 64 lanes, no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define MI355_OK 0
#define MI355_EINVAL (-1)
#define MI355_ELAUNCH (-2)

extern "C" void mi355_set_error(const char* msg);

#define MI355_CHECK_ARG(cond, msg)                                   \
  do {                                                               \
    if (!(cond)) { mi355_set_error(msg); return MI355_EINVAL; }      \
  } while (0)

#define MI355_LAUNCH_CHECK()                                         \
  do {                                                               \
    hipError_t e__ = hipGetLastError();                              \
    if (e__ != hipSuccess) { mi355_set_error(hipGetErrorString(e__)); return MI355_ELAUNCH; } \
  } while (0)

namespace mi355 {

constexpr int kWave = 64;

enum DType : int { kF32 = 0, kBF16 = 1, kF16 = 2 };

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid for a grid-stride kernel over `work` items, `per_block` items per block per iteration:
// enough blocks to fill 256 CUs x 8 blocks, never more than the work.
static inline int grid_for(int64_t work, int64_t per_block, int64_t cap = 256 * 16) {
  int64_t g = ceil_div(work, per_block);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- 64-bit hash of the scored hash table (fmix64 of MurmurHash3) ----
__host__ __device__ __forceinline__ uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
  k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
  k ^= k >> 33;
  return k;
}

// ---- bf16 / f16 <-> f32 (round to nearest even) ----
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float f16_to_f32(uint16_t h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ uint16_t f32_to_f16(float f) { return __half_as_ushort(__float2half_rn(f)); }

template <int DT> struct Elem;
template <> struct Elem<kF32> {
  using T = float;
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Elem<kBF16> {
  using T = uint16_t;
  static __device__ __forceinline__ float ld(const uint16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_bf16(v); }
  static __device__ __forceinline__ float rnd(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};
template <> struct Elem<kF16> {
  using T = uint16_t;
  static __device__ __forceinline__ float ld(const uint16_t* p) { return f16_to_f32(*p); }
  static __device__ __forceinline__ void st(uint16_t* p, float v) { *p = f32_to_f16(v); }
  static __device__ __forceinline__ float rnd(float v) { return f16_to_f32(f32_to_f16(v)); }
};

static inline size_t dtype_bytes(int dt) { return dt == kF32 ? 4 : 2; }

// load 4 consecutive elements as fp32 (16 B for f32, 8 B for 16-bit types)
template <int DT>
__device__ __forceinline__ float4 ld4(const void* base, int64_t elem_off) {
  if constexpr (DT == kF32) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem_off);
  } else {
    uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + elem_off);
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
template <int DT>
__device__ __forceinline__ void st4(void* base, int64_t elem_off, float4 v) {
  if constexpr (DT == kF32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem_off) = v;
  } else {
    uint2 r;
    if constexpr (DT == kBF16) {
      r.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
      r.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    } else {
      r.x = (uint32_t)f32_to_f16(v.x) | ((uint32_t)f32_to_f16(v.y) << 16);
      r.y = (uint32_t)f32_to_f16(v.z) | ((uint32_t)f32_to_f16(v.w) << 16);
    }
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + elem_off) = r;
  }
}
template <int DT>
__device__ __forceinline__ float ld1(const void* base, int64_t elem_off) {
  return Elem<DT>::ld(reinterpret_cast<const typename Elem<DT>::T*>(base) + elem_off);
}
template <int DT>
__device__ __forceinline__ void st1(void* base, int64_t elem_off, float v) {
  Elem<DT>::st(reinterpret_cast<typename Elem<DT>::T*>(base) + elem_off, v);
}

// agent-scope relaxed atomics on 64-bit words (the slot lock word of the table)
__device__ __forceinline__ uint64_t ald64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ast64(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool cas64(uint64_t* p, uint64_t& expected, uint64_t desired) {
  return __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// device wall clock in ns-like ticks (monotonic; used for the GLOBAL_TIMER / LRU score)
__device__ __forceinline__ uint64_t device_clock() { return (uint64_t)wall_clock64(); }

// dispatch helpers
#define MI355_DISPATCH_DTYPE(dt, NAME, ...)                                  \
  [&] {                                                                      \
    switch (dt) {                                                            \
      case mi355::kF32: { constexpr int NAME = mi355::kF32; return __VA_ARGS__(); }   \
      case mi355::kBF16: { constexpr int NAME = mi355::kBF16; return __VA_ARGS__(); } \
      case mi355::kF16: { constexpr int NAME = mi355::kF16; return __VA_ARGS__(); }   \
      default: mi355_set_error("unsupported dtype"); return MI355_EINVAL;    \
    }                                                                        \
  }()

}  // namespace mi355
