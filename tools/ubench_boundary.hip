// What a dependent kernel boundary costs on MI355X as a function of the producer's shape: from the last instruction of kernel A's
// last block to the first instruction of kernel B's first block (constant 100 MHz wall clock inside the kernels), for A with
// 256- / 1024-thread blocks, with / without a large LDS allocation, leaving 0 .. 32 MB of freshly written data behind.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int LDSB>
__global__ void producer(unsigned long long* t, float4* out, long long n16, int spin_us) {
  __shared__ char lds[LDSB > 0 ? LDSB : 4];
  if (LDSB > 0 && threadIdx.x == 0) lds[0] = 1;
  const unsigned long long t0 = wall_clock64();
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
  while (wall_clock64() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  __syncthreads();
  if (threadIdx.x == 0) t[blockIdx.x] = wall_clock64();
}
__global__ void consumer(unsigned long long* t) {
  if (threadIdx.x == 0) t[4096 + blockIdx.x] = wall_clock64();
}
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 8192 * 8));
  std::vector<unsigned long long> h(8192);
  float4* out; CK(hipMalloc(&out, 64 << 20));
  for (int threads : {256, 1024}) for (int lds : {0, 1}) for (int mb : {0, 2, 8, 32}) for (int cblocks : {256, 2048}) {
    std::vector<double> gaps;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipMemset(d, 0, 8192 * 8));
      const int blocks = threads == 256 ? 1024 : 256;
      const long long n16 = (long long)mb << 16;
      if (lds) hipLaunchKernelGGL(producer<34000>, dim3(blocks), dim3(threads), 0, 0, d, out, n16, 15);
      else hipLaunchKernelGGL(producer<0>, dim3(blocks), dim3(threads), 0, 0, d, out, n16, 15);
      hipLaunchKernelGGL(consumer, dim3(cblocks), dim3(256), 0, 0, d);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), d, 8192 * 8, hipMemcpyDeviceToHost));
      unsigned long long e = 0, b = ~0ull;
      for (int i = 0; i < blocks; ++i) e = std::max(e, h[i]);
      for (int i = 0; i < cblocks; ++i) b = std::min(b, h[4096 + i]);
      if (rep >= 2) gaps.push_back(((double)b - (double)e) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("producer %4d threads x %4d blocks, LDS %5d B, %2d MB written -> consumer of %4d blocks: gap median %5.2f us (min %5.2f)\n", threads,
           threads == 256 ? 1024 : 256, lds ? 34000 : 0, mb, cblocks, gaps[gaps.size() / 2], gaps[0]);
  }
  return 0;
}
