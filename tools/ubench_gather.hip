// What the pooled gather of C2 can reach: the real data shape (65,536 bags of 1..10 keys, Zipf-0.99 rows out of 10 M x 128 fp32,
// per-occurrence 64-bit row addresses, bf16 [B, 128] output) through deliberately simple kernels -- one 32-lane group per bag,
// every lane loads the (group-uniform) address word itself, U rows in flight -- to find which feature of the library kernel
// (value_ops.hip: one key per lane + shuffles, KIT bags per group, 3-hop software pipeline) costs what.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather.bin tools/ubench_gather.hip && tools/ubench_gather.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <dlfcn.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f4* gp4;
__device__ __attribute__((aligned(16))) float g_zero[128];
static inline uint64_t fmix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }
__device__ __forceinline__ uint32_t bf16pack(float a, float b) {
  uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
  x += 0x7fffu + ((x >> 16) & 1u); y += 0x7fffu + ((y >> 16) & 1u);
  return (x >> 16) | (y & 0xffff0000u);
}
// KIT consecutive bags per lane group, U rows in flight per round; kPre: the address words of the NEXT bag are loaded before the
// rows of the current one (their latency hides under the rows)
template <int U, int KIT, bool kPre>
__global__ void __launch_bounds__(256) bag_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ addr, int B, uint2* out) {
  const int lane = threadIdx.x & 63, sub = lane >> 5, c = lane & 31;
  const int g = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + sub;
  const uintptr_t zero = (uintptr_t)g_zero;
  for (int k = 0; k < KIT; ++k) {
    const int bag = g * KIT + k;
    if (bag >= B) return;
    const int64_t lo = offsets[bag], hi = offsets[bag + 1];
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r = lo; r < hi; r += U) {
      uintptr_t p[U];
#pragma unroll
      for (int q = 0; q < U; ++q) { const int64_t j = r + q < hi ? r + q : hi - 1; p[q] = (uintptr_t)addr[j]; }
      f4 v[U];
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = *(gp4)((r + q < hi ? p[q] : zero) + 16 * c);
#pragma unroll
      for (int q = 0; q < U; ++q) acc += v[q];
    }
    out[(int64_t)bag * 32 + c] = make_uint2(bf16pack(acc.x, acc.y), bf16pack(acc.z, acc.w));
  }
}
// persistent: lane groups take bags g, g + G, g + 2G, ...; the offsets / addresses of the next bag are fetched under the current
// bag's rows (software pipeline over the grid-stride loop)
template <int U>
__global__ void __launch_bounds__(256) bag_persist_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ addr, int B, uint2* out) {
  const int lane = threadIdx.x & 63, sub = lane >> 5, c = lane & 31;
  const int G = gridDim.x * (blockDim.x >> 6) * 2;
  int bag = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + sub;
  const uintptr_t zero = (uintptr_t)g_zero;
  if (bag >= B) return;
  int64_t lo = offsets[bag], hi = offsets[bag + 1];
  uintptr_t p[U];
#pragma unroll
  for (int q = 0; q < U; ++q) { const int64_t j = lo + q < hi ? lo + q : hi - 1; p[q] = (uintptr_t)addr[j]; }
  while (true) {
    const int nb = bag + G;
    const int nbc = nb < B ? nb : B - 1;
    const int64_t nlo = offsets[nbc], nhi = offsets[nbc + 1];      // next bag's offsets: issued first (oldest)
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    f4 v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) v[q] = *(gp4)((lo + q < hi ? p[q] : zero) + 16 * c);
    uintptr_t np[U];
#pragma unroll
    for (int q = 0; q < U; ++q) { const int64_t j = nlo + q < nhi ? nlo + q : nhi - 1; np[q] = (uintptr_t)addr[j]; }   // waits for nlo only
#pragma unroll
    for (int q = 0; q < U; ++q) acc += v[q];
    for (int64_t r = lo + U; r < hi; r += U) {   // bags longer than U
      uintptr_t pp[U];
#pragma unroll
      for (int q = 0; q < U; ++q) { const int64_t j = r + q < hi ? r + q : hi - 1; pp[q] = (uintptr_t)addr[j]; }
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = *(gp4)((r + q < hi ? pp[q] : zero) + 16 * c);
#pragma unroll
      for (int q = 0; q < U; ++q) acc += v[q];
    }
    out[(int64_t)bag * 32 + c] = make_uint2(bf16pack(acc.x, acc.y), bf16pack(acc.z, acc.w));
    if (nb >= B) break;
    bag = nb; lo = nlo; hi = nhi;
#pragma unroll
    for (int q = 0; q < U; ++q) p[q] = np[q];
  }
}

// flat row stream: a lane group owns KB consecutive bags = one contiguous run of rows, walked in chunks of U rows whatever the
// bag boundaries are; the rows of chunk k+1 are issued BEFORE chunk k is accumulated (double buffer), the address words of
// chunk k+2 before that; a bag's sum is stored when its last row has been added.  LPR lanes per row (32: 16 B per lane,
// 16: 2 x 16 B per lane).
template <int U, int KB, int LPR>
__global__ void __launch_bounds__(256) flat_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ addr, int B, uint2* out) {
  constexpr int NL = 32 / LPR;            // 16-byte loads per lane and row
  const int lane = threadIdx.x & 63, c = lane & (LPR - 1);
  const int g = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / LPR) + (lane / LPR);
  const uintptr_t zero = (uintptr_t)g_zero;
  const int b0 = g * KB;
  if (b0 >= B) return;
  const int bn = b0 + KB < B ? KB : B - b0;
  int64_t myoff = offsets[b0 + (c <= bn ? c : bn)];
  const int olo = (int)myoff, ohi = (int)(myoff >> 32);
  auto off = [&](int i) -> int64_t { return (int64_t)(((uint64_t)(unsigned)__shfl(ohi, i, LPR) << 32) | (unsigned)__shfl(olo, i, LPR)); };
  const int64_t lo = off(0), hi = off(bn);
  auto fetch_addr = [&](int64_t r) -> uintptr_t { int64_t j = r + c; j = j < hi ? j : hi - 1; j = j < 0 ? 0 : j; return (uintptr_t)addr[j]; };
  auto issue = [&](uintptr_t a, int64_t r, f4 (&v)[U][NL]) {
    const int alo = (int)(a & 0xffffffffu), ahi = (int)(a >> 32);
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const uintptr_t base = (uintptr_t)(unsigned)__shfl(alo, q, LPR) | ((uintptr_t)(unsigned)__shfl(ahi, q, LPR) << 32);
      const uintptr_t p = r + q < hi ? base : zero;
#pragma unroll
      for (int h = 0; h < NL; ++h) v[q][h] = *(gp4)(p + (r + q < hi ? 16 * (c + h * LPR) : 0));
    }
  };
  int b = 0;                 // current bag (relative)
  int64_t bend = off(1);
  f4 acc[NL];
#pragma unroll
  for (int h = 0; h < NL; ++h) acc[h] = (f4){0.f, 0.f, 0.f, 0.f};
  auto flush = [&]() {
#pragma unroll
    for (int h = 0; h < NL; ++h) {
      out[(int64_t)(b0 + b) * 32 + c + h * LPR] = make_uint2(bf16pack(acc[h].x, acc[h].y), bf16pack(acc[h].z, acc[h].w));
      acc[h] = (f4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto consume = [&](int64_t r, f4 (&v)[U][NL]) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t j = r + q;
      if (j < hi) {
#pragma unroll
        for (int h = 0; h < NL; ++h) acc[h] += v[q][h];
        while (b < bn && j + 1 == bend) { flush(); ++b; bend = off(b + 1 <= bn ? b + 1 : bn); if (b + 1 > bn) break; }
      }
    }
  };
  while (b < bn && bend == lo) { flush(); ++b; bend = off(b + 1 <= bn ? b + 1 : bn); }   // leading empty bags
  uintptr_t a0 = fetch_addr(lo), a1 = fetch_addr(lo + U);
  f4 va[U][NL], vb[U][NL];
  issue(a0, lo, va);
  for (int64_t r = lo; r < hi; r += 2 * U) {
    const uintptr_t a2 = fetch_addr(r + 2 * U);
    issue(a1, r + U, vb);
    consume(r, va);
    if (r + U >= hi) break;
    const uintptr_t a3 = fetch_addr(r + 3 * U);
    issue(a2, r + 2 * U, va);
    consume(r + U, vb);
    a1 = a3;
  }
}
// key-balanced flat stream (round 4): the bag -> lane group deal is by KEYS, not by bags.  Group g owns the bags whose first
// key lies in [g Q, (g + 1) Q) of the batch's key stream (the offsets are the prefix sum): every group then moves Q +- one bag
// of rows instead of KB bags of 1..10 keys each (4 bags: 4..40 rows -- all groups are resident at once, the kernel ends with the
// unluckiest one).  The two slice ends are found by an LPR-ary search of the group's lanes over the offsets (4 rounds of one
// load per lane at 64 K bags, both searches in the same rounds); the bags are then walked as flat_kernel does, in runs of at
// most LPR - 1 bags.
template <int LPR>
__device__ __forceinline__ void lower_bound2(const int64_t* __restrict__ off, int nOff, int64_t t0, int64_t t1, int c, int& r0, int& r1) {
  int lo0 = 0, len0 = nOff, lo1 = 0, len1 = nOff;     // answer_i in [lo_i, lo_i + len_i]: first index with off[index] >= t_i (nOff: none)
  while (__any(len0 > 0 || len1 > 0)) {
    const int st0 = (len0 + LPR - 1) / LPR, st1 = (len1 + LPR - 1) / LPR;
    const int i0 = lo0 + c * st0, i1 = lo1 + c * st1;
    const bool v0 = len0 > 0 && i0 < lo0 + len0, v1 = len1 > 0 && i1 < lo1 + len1;
    const int64_t x0 = off[v0 ? i0 : 0], x1 = off[v1 ? i1 : 0];
    const unsigned long long m0 = __ballot(v0 && x0 < t0), m1 = __ballot(v1 && x1 < t1);
    const int sh = (threadIdx.x & 63) & ~(LPR - 1);
    const unsigned long long gm = LPR == 64 ? ~0ull : ((1ull << LPR) - 1);
    const int c0 = __popcll((m0 >> sh) & gm), c1 = __popcll((m1 >> sh) & gm);
    if (len0 > 0) { if (c0 == 0) len0 = 0; else { const int nl = lo0 + (c0 - 1) * st0 + 1, end = lo0 + len0; lo0 = nl; len0 = end - nl < st0 - 1 ? end - nl : st0 - 1; } }
    if (len1 > 0) { if (c1 == 0) len1 = 0; else { const int nl = lo1 + (c1 - 1) * st1 + 1, end = lo1 + len1; lo1 = nl; len1 = end - nl < st1 - 1 ? end - nl : st1 - 1; } }
  }
  r0 = lo0; r1 = lo1;
}
template <int U, int Q, int LPR>
__global__ void __launch_bounds__(256) flat_bal_kernel(const int64_t* __restrict__ offsets, const int64_t* __restrict__ addr, int B, uint2* out) {
  constexpr int NL = 32 / LPR, KB = LPR - 1;
  const int lane = threadIdx.x & 63, c = lane & (LPR - 1);
  const int g = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (64 / LPR) + (lane / LPR);
  const uintptr_t zero = (uintptr_t)g_zero;
  const int64_t n = offsets[B];
  const int64_t t0 = (int64_t)g * Q, t1 = t0 + Q;
  int blo, bhi;
  lower_bound2<LPR>(offsets, B + 1, t0 < n ? t0 : n + 1, t1, c, blo, bhi);   // (whole waves search: the shuffles / ballots need every lane)
  if (t0 >= n && !(n == 0 && g == 0)) return;
  if (t1 >= n) bhi = B;                     // the last slice also owns the trailing empty bags
  if (n == 0) blo = 0;
  if (bhi > B) bhi = B;
  for (int b0 = blo; b0 < bhi; b0 += KB) {
    const int bn = bhi - b0 < KB ? bhi - b0 : KB;
    int64_t myoff = offsets[b0 + (c <= bn ? c : bn)];
    const int olo = (int)myoff, ohi = (int)(myoff >> 32);
    auto off = [&](int i) -> int64_t { return (int64_t)(((uint64_t)(unsigned)__shfl(ohi, i, LPR) << 32) | (unsigned)__shfl(olo, i, LPR)); };
    const int64_t lo = off(0), hi = off(bn);
    auto fetch_addr = [&](int64_t r) -> uintptr_t { int64_t j = r + c; j = j < hi ? j : hi - 1; j = j < 0 ? 0 : j; return (uintptr_t)addr[j]; };
    auto issue = [&](uintptr_t a, int64_t r, f4 (&v)[U][NL]) {
      const int alo = (int)(a & 0xffffffffu), ahi = (int)(a >> 32);
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const uintptr_t base = (uintptr_t)(unsigned)__shfl(alo, q, LPR) | ((uintptr_t)(unsigned)__shfl(ahi, q, LPR) << 32);
        const uintptr_t p = r + q < hi ? base : zero;
#pragma unroll
        for (int h = 0; h < NL; ++h) v[q][h] = *(gp4)(p + (r + q < hi ? 16 * (c + h * LPR) : 0));
      }
    };
    int b = 0;
    int64_t bend = off(1);
    f4 acc[NL];
#pragma unroll
    for (int h = 0; h < NL; ++h) acc[h] = (f4){0.f, 0.f, 0.f, 0.f};
    auto flush = [&]() {
#pragma unroll
      for (int h = 0; h < NL; ++h) {
        out[(int64_t)(b0 + b) * 32 + c + h * LPR] = make_uint2(bf16pack(acc[h].x, acc[h].y), bf16pack(acc[h].z, acc[h].w));
        acc[h] = (f4){0.f, 0.f, 0.f, 0.f};
      }
    };
    auto consume = [&](int64_t r, f4 (&v)[U][NL]) {
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int64_t j = r + q;
        if (j < hi) {
#pragma unroll
          for (int h = 0; h < NL; ++h) acc[h] += v[q][h];
          while (b < bn && j + 1 == bend) { flush(); ++b; bend = off(b + 1 <= bn ? b + 1 : bn); if (b + 1 > bn) break; }
        }
      }
    };
    while (b < bn && bend == lo) { flush(); ++b; bend = off(b + 1 <= bn ? b + 1 : bn); }   // leading empty bags
    if (lo < hi) {
      uintptr_t a0 = fetch_addr(lo), a1 = fetch_addr(lo + U);
      f4 va[U][NL], vb[U][NL];
      issue(a0, lo, va);
      for (int64_t r = lo; r < hi; r += 2 * U) {
        const uintptr_t a2 = fetch_addr(r + 2 * U);
        issue(a1, r + U, vb);
        consume(r, va);
        if (r + U >= hi) break;
        const uintptr_t a3 = fetch_addr(r + 3 * U);
        issue(a2, r + 2 * U, va);
        consume(r + U, vb);
        a1 = a3;
      }
    }
    while (b < bn) { flush(); ++b; }
  }
}
__global__ void fill_kernel(float* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023) * 1e-3f;
}
int main() {
  const int64_t rows = 10000000; const int B = 65536, NSET = 6;
  float* table; CK(hipMalloc(&table, rows * 512));
  fill_kernel<<<4096, 256>>>(table, rows * 128);
  std::vector<double> cdf(rows);
  { double s = 0; for (int64_t r = 0; r < rows; ++r) { s += pow((double)(r + 1), -0.99); cdf[r] = s; } for (auto& x : cdf) x /= s; }
  std::vector<int64_t*> d_off, d_addr; std::vector<int64_t> nts;
  for (int s = 0; s < NSET; ++s) {
    uint64_t st = 99 + 13 * s;
    auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
    std::vector<int64_t> off(B + 1); off[0] = 0;
    for (int b = 0; b < B; ++b) off[b + 1] = off[b] + 1 + (int64_t)(rnd() * 10);
    const int64_t nt = off[B];
    std::vector<int64_t> ad(nt);
    for (int64_t i = 0; i < nt; ++i) {
      const int64_t r = std::lower_bound(cdf.begin(), cdf.end(), rnd()) - cdf.begin();
      ad[i] = (int64_t)(uintptr_t)table + (int64_t)(fmix64((uint64_t)(r < rows ? r : rows - 1) + 77) % (uint64_t)rows) * 512;
    }
    int64_t *o, *a; CK(hipMalloc(&o, (B + 1) * 8)); CK(hipMalloc(&a, nt * 8));
    CK(hipMemcpy(o, off.data(), (B + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(a, ad.data(), nt * 8, hipMemcpyHostToDevice));
    d_off.push_back(o); d_addr.push_back(a); nts.push_back(nt);
  }
  uint2* out; CK(hipMalloc(&out, (int64_t)B * 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](auto kern, const char* name, int blocks) {
    for (int s = 0; s < 3; ++s) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_off[s], d_addr[s], B, out);
    CK(hipDeviceSynchronize());
    float tot = 0.f, best = 1e9f; const int reps = 18;
    for (int i = 0; i < reps; ++i) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_off[i % NSET], d_addr[i % NSET], B, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; best = ms < best ? ms : best;
    }
    printf("  %-34s blocks %6d: avg %6.1f us  best %6.1f us\n", name, blocks, tot / reps * 1e3, best * 1e3);
  };
  printf("C2 gather shape: B %d, keys per batch ~%lld\n", B, (long long)nts[0]);
  {  // the flat-stream kernels against the simplest one, bit for bit
    std::vector<uint2> ref((size_t)B * 32), got((size_t)B * 32);
    hipLaunchKernelGGL((bag_kernel<4, 1, false>), dim3(B / 8), dim3(256), 0, 0, d_off[0], d_addr[0], B, out);
    CK(hipMemcpy(ref.data(), out, ref.size() * 8, hipMemcpyDeviceToHost));
    auto chk = [&](auto kern, const char* nm, int blocks) {
      CK(hipMemset(out, 0xff, (size_t)B * 256));
      hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_off[0], d_addr[0], B, out);
      CK(hipMemcpy(got.data(), out, got.size() * 8, hipMemcpyDeviceToHost));
      size_t bad = 0; for (size_t i = 0; i < ref.size(); ++i) bad += (ref[i].x != got[i].x) || (ref[i].y != got[i].y);
      printf("  check %-28s mismatching 8-byte words: %zu\n", nm, bad);
    };
    chk(flat_kernel<4, 8, 32>, "flat U=4 KB=8 LPR=32", (B / 8 + 7) / 8);
    chk(flat_kernel<2, 4, 16>, "flat U=2 KB=4 LPR=16", (B / 4 + 15) / 16);
    chk(flat_kernel<4, 16, 32>, "flat U=4 KB=16 LPR=32", (B / 16 + 7) / 8);
    chk(flat_bal_kernel<4, 22, 32>, "balanced U=4 Q=22 LPR=32", (int)((nts[0] / 22 + 1 + 7) / 8));
    chk(flat_bal_kernel<4, 32, 32>, "balanced U=4 Q=32 LPR=32", (int)((nts[0] / 32 + 1 + 7) / 8));
    chk(flat_bal_kernel<4, 16, 16>, "balanced U=4 Q=16 LPR=16", (int)((nts[0] / 16 + 1 + 15) / 16));
  }
  {  // the library's own kernels on exactly this data (MI355_POOL_VARIANT picks: 0 flat, 30 the round-2 pipelined one)
    void* h = dlopen(getenv("MI355_LIB") ? getenv("MI355_LIB") : "recsys-examples_amd/lib/librecsys_amd.so", RTLD_NOW);
    if (h) {
      typedef int (*gp_t)(const void*, int64_t, const int64_t*, int, const int64_t*, int64_t, const int64_t*, int64_t, int64_t, int, int64_t,
                          const int32_t*, int64_t, void*, int, int, hipStream_t);
      gp_t gp = (gp_t)dlsym(h, "mi355_gather_pooled");
      for (int s = 0; s < 3; ++s) gp(nullptr, 0, d_addr[s], 0, nullptr, nts[s], d_off[s], B, B, 0, 128, nullptr, 128, out, 1, 1, 0);
      CK(hipDeviceSynchronize());
      float tot = 0.f, best = 1e9f; const int reps = 18;
      for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(e0));
        gp(nullptr, 0, d_addr[i % NSET], 0, nullptr, nts[i % NSET], d_off[i % NSET], B, B, 0, 128, nullptr, 128, out, 1, 1, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; best = ms < best ? ms : best;
      }
      printf("  library mi355_gather_pooled (MI355_POOL_VARIANT=%s): avg %6.1f us  best %6.1f us\n", getenv("MI355_POOL_VARIANT") ? getenv("MI355_POOL_VARIANT") : "0", tot / reps * 1e3, best * 1e3);
    } else printf("  (library not found: %s)\n", dlerror());
  }
#define ONE(U, KIT) run(bag_kernel<U, KIT, false>, "one-shot U=" #U " KIT=" #KIT, (B / KIT + 7) / 8)
  ONE(4, 1); ONE(8, 1); ONE(10, 1); ONE(4, 2); ONE(8, 2); ONE(10, 2); ONE(4, 4); ONE(8, 4); ONE(10, 4);
#define FLAT(U, KB, LPR) run(flat_kernel<U, KB, LPR>, "flat double-buffered U=" #U " KB=" #KB " LPR=" #LPR, (B / KB + (256 / LPR) - 1) / (256 / LPR))
  FLAT(4, 4, 32); FLAT(4, 8, 32); FLAT(4, 16, 32); FLAT(6, 8, 32); FLAT(8, 8, 32); FLAT(2, 4, 32); FLAT(2, 8, 32);
  FLAT(2, 4, 16); FLAT(2, 8, 16); FLAT(4, 4, 16); FLAT(4, 8, 16); FLAT(3, 8, 16);
#define BAL(U, Q, LPR) run(flat_bal_kernel<U, Q, LPR>, "key-balanced flat U=" #U " Q=" #Q " LPR=" #LPR, (int)((*std::max_element(nts.begin(), nts.end()) / Q + 1 + (256 / LPR) - 1) / (256 / LPR)))
  BAL(4, 16, 32); BAL(4, 22, 32); BAL(4, 28, 32); BAL(4, 44, 32); BAL(2, 22, 32); BAL(6, 22, 32); BAL(8, 44, 32); BAL(4, 22, 16); BAL(4, 44, 16); BAL(4, 88, 32);
  for (int bpc : {4}) { run(bag_persist_kernel<4>, "persistent pipelined U=4", 256 * bpc); run(bag_persist_kernel<8>, "persistent pipelined U=8", 256 * bpc);
                                 run(bag_persist_kernel<10>, "persistent pipelined U=10", 256 * bpc); }
  return 0;
}
