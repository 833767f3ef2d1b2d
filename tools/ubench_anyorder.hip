// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(unsigned long long* t, int us) {
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
  while (wall_clock64() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(16);
  if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = wall_clock64();
}
int main() {
  unsigned long long *d, h[4];
  CK(hipMalloc(&d, 32));
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int flags = 0; flags < 2; ++flags) {
    for (int rep = 0; rep < 3; ++rep) {
      hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, 0, d, 100);
      hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s, nullptr, nullptr, flags, d + 2, 100);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
      printf("flags %d: A %.1f..%.1f us, B %.1f..%.1f us -> %s\n", flags, 0.0, (h[1] - h[0]) / 100.0, ((long long)h[2] - (long long)h[0]) / 100.0,
             ((long long)h[3] - (long long)h[0]) / 100.0, h[2] < h[1] ? "OVERLAP" : "serial");
    }
  }
  return 0;
}
