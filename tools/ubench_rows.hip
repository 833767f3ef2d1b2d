// Speed of light of random 512-byte row traffic on one MI355X (no torch, no library): what a gather / read-modify-write of
// 128-D fp32 embedding rows can reach as a function of rows in flight per lane group, waves per CU and key distribution.
// The C2 kernels (value_ops.hip, backward.hip) are judged against these figures, not against a streaming copy.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_rows tools/ubench_rows.hip && /tmp/ubench_rows
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) f4* gp4;

static inline uint64_t fmix64(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k;
}

// one 32-lane group per chunk of `per` consecutive indices, U rows in flight; every `bag` rows one 256-B bf16-sized store
// (8 B per lane) -- the shape of the pooled gather.  kMode 0 gather-sum, 1 read-modify-write of the row (the SGD update).
template <int U, int kMode>
__global__ void __launch_bounds__(256) rows_kernel(float* table, const int* __restrict__ idx, int n, int per, int bag, float* out) {
  const int lane = threadIdx.x & 63, sub = lane >> 5, c = lane & 31;
  const int64_t ngroups = (int64_t)gridDim.x * (blockDim.x >> 6) * 2;
  for (int64_t g = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + sub; g * per < n; g += ngroups) {
    const int lo = (int)(g * per);
    int hi = lo + per; hi = hi < n ? hi : n;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    int since = 0;
    for (int j0 = lo; j0 < hi; j0 += U) {
      int r[U];
#pragma unroll
      for (int q = 0; q < U; ++q) { int j = j0 + q; j = j < hi ? j : hi - 1; r[q] = idx[j]; }
      f4 v[U];
#pragma unroll
      for (int q = 0; q < U; ++q) v[q] = *(gp4)(uintptr_t)(table + (int64_t)r[q] * 128 + 4 * c);
#pragma unroll
      for (int q = 0; q < U; ++q) {
        if (j0 + q < hi) {
          if (kMode == 1) {
            f4 w = v[q]; w.x -= 0.1f; w.y -= 0.1f; w.z -= 0.1f; w.w -= 0.1f;
            *(f4*)(table + (int64_t)r[q] * 128 + 4 * c) = w;
          } else { acc += v[q]; }
        }
      }
      since += U;
      if (kMode == 0 && since >= bag) {
        float2 o = {acc.x + acc.y, acc.z + acc.w};
        *(float2*)(out + ((int64_t)((j0 / bag) & 0x3ffff) * 64 + 2 * c)) = o;
        acc = (f4){0.f, 0.f, 0.f, 0.f}; since = 0;
      }
    }
  }
}

__global__ void fill_kernel(float* p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023) * 1e-3f;
}
__global__ void copy_kernel(const f4* a, f4* b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
  const int64_t rows = 10000000;
  float* table; CK(hipMalloc(&table, rows * 512));
  fill_kernel<<<4096, 256>>>(table, rows * 128);
  float* out; CK(hipMalloc(&out, 80 << 20));
  const int NSET = 6;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  {  // streaming copy reference (read + write bytes)
    const int64_t n = rows * 32 / 2;
    copy_kernel<<<8192, 256>>>((const f4*)table, (f4*)table + n, n);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) copy_kernel<<<8192, 256>>>((const f4*)table, (f4*)table + n, n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream copy: %.0f GB/s (read + write)\n", 5.0 * n * 32 / (ms * 1e6));
  }
  for (int dist = 0; dist < 2; ++dist) {
    for (int64_t n : {360000LL, 5760000LL}) {
      std::vector<int*> sets;
      for (int s = 0; s < NSET; ++s) {
        std::vector<int> h(n);
        uint64_t st = 1234567 + 977 * s + 31 * dist;
        for (int64_t i = 0; i < n; ++i) {
          st = st * 6364136223846793005ULL + 1442695040888963407ULL;
          const double u = (double)(st >> 11) * (1.0 / 9007199254740992.0);
          int64_t r = dist == 0 ? (int64_t)(u * rows) : (int64_t)exp(u * log((double)rows));   // uniform | p(r) ~ 1/r
          if (r >= rows) r = rows - 1;
          h[i] = (int)(fmix64((uint64_t)r + 0x9E37) % (uint64_t)rows);
        }
        int* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
        sets.push_back(d);
      }
      auto run = [&](auto kern, const char* name, int blocks, int per, int bag, double bytes_per_row) {
        for (int s = 0; s < 2; ++s) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, table, sets[s], (int)n, per, bag, out);
        CK(hipDeviceSynchronize());
        float best = 1e9f, tot = 0.f;
        const int reps = 12;
        for (int i = 0; i < reps; ++i) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, table, sets[i % NSET], (int)n, per, bag, out);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          best = ms < best ? ms : best; tot += ms;
        }
        printf("  %-28s blocks %6d per %5d: avg %7.1f us  best %7.1f us  -> %6.0f GB/s (avg)\n", name, blocks, per, tot / reps * 1e3,
               best * 1e3, n * bytes_per_row / (tot / reps * 1e6));
      };
      printf("%s keys, n = %lld\n", dist == 0 ? "uniform" : "zipf(1.0)", (long long)n);
      const int bag = 6;
      for (int per : {6, 24, 96}) {
        const int groups = (int)((n + per - 1) / per);
        const int blocks = (groups + 7) / 8;
        run(rows_kernel<2, 0>, "gather U=2 one-shot", blocks, per, bag, 512 + 256.0 / bag);
        run(rows_kernel<4, 0>, "gather U=4 one-shot", blocks, per, bag, 512 + 256.0 / bag);
        run(rows_kernel<8, 0>, "gather U=8 one-shot", blocks, per, bag, 512 + 256.0 / bag);
      }
      for (int bpc : {2, 4, 8}) {
        run(rows_kernel<4, 0>, "gather U=4 persistent", 256 * bpc, 6, bag, 512 + 256.0 / bag);
        run(rows_kernel<8, 0>, "gather U=8 persistent", 256 * bpc, 8, bag, 512 + 256.0 / bag);
      }
      for (int per : {4, 16}) {
        const int groups = (int)((n + per - 1) / per);
        run(rows_kernel<2, 1>, "rmw U=2 one-shot", (groups + 7) / 8, per, bag, 1024);
        run(rows_kernel<4, 1>, "rmw U=4 one-shot", (groups + 7) / 8, per, bag, 1024);
      }
      for (int bpc : {4, 8}) run(rows_kernel<4, 1>, "rmw U=4 persistent", 256 * bpc, 4, bag, 1024);
      for (int* d : sets) CK(hipFree(d));
    }
  }
  return 0;
}
