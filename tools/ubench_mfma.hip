// What the matrix pipe of an MI355X CU sustains for v_mfma_f32_32x32x16_bf16, in s_memtime ticks and in wall time, as a function of
// how much of the chip runs it -- the yardstick for the "45-52 cycles per MFMA" the attention kernels' phase stamps show.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mfma.hip -o /tmp/ubench_mfma && /tmp/ubench_mfma
// Variants: independent accumulators (4) or one dependent chain; 1 or 2 waves per SIMD; the second wave of a SIMD running MFMAs
// too, or SiLU-like VALU work (exp2 + rcp + 6 plain ops per element), or LDS reads; 1, 32 (one XCD) or 256 CUs busy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

// mode of the second wave of every SIMD (waves 4-7 of a 512-thread block): 0 MFMA like the first, 1 VALU (SiLU-like), 2 LDS reads, 3 idle exit
template <int NACC>
__global__ void __launch_bounds__(512) k(unsigned long long* out, float* sink, int iters, int mode2) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i * 1e-6f;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  float res = 0.f;
  if (wv < 4 || mode2 == 0) {
    f32x16_t acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8_t x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(0.001f * (lane + e)); y[e] = (__bf16)(0.002f * (lane - e)); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    for (int a = 0; a < NACC; ++a) res += acc[a][lane & 15];
  } else if (mode2 == 1) {
    float v[16];
    for (int e = 0; e < 16; ++e) v[e] = 0.01f * (lane + e);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[e] * -1.44f));
        v[e] = v[e] * 0.5f * sg + (1.f + v[e] * (1.f - sg)) * sg * 0.25f;
      }
    }
    for (int e = 0; e < 16; ++e) res += v[e];
  } else if (mode2 == 2) {
    float4 s = {0, 0, 0, 0};
    for (int it = 0; it < 5 * iters; ++it) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float4 q = *reinterpret_cast<const float4*>(&lds[((lane + 64 * e + it) * 4) & 8188]);
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
      }
    }
    res = s.x + s.y + s.z + s.w;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (res == 12345.678f) sink[0] = res;
  if (lane == 0) { out[(blockIdx.x * 8 + wv) * 2] = t1 - t0; out[(blockIdx.x * 8 + wv) * 2 + 1] = r1 - r0; }
}

template <int NACC>
static int run(const char* what, int blocks, int threads, int mode2, unsigned long long* d, float* sink) {
  const int iters = 4000;   // x 16 MFMAs
  std::vector<unsigned long long> h(256 * 8 * 2);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best_ms = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemset(d, 0, h.size() * 8));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, sink, iters, mode2);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best_ms = std::min(best_ms, (double)ms);
  }
  CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
  double ticks = 0, real = 0; int n = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) { ticks += h[(b * 8 + w) * 2]; real += h[(b * 8 + w) * 2 + 1]; ++n; }
  ticks /= n; real /= n;
  const double nm = 16.0 * iters;
  const int mfma_waves = blocks * (threads == 512 && mode2 == 0 ? 8 : 4);
  printf("%-58s CUs %3d  ticks/MFMA %6.1f  ns/MFMA %6.2f  ticks per us %7.1f  chip %7.1f TFLOP/s (kernel %.3f ms)\n", what, blocks, ticks / nm, real * 10.0 / nm,
         ticks / (real / 100.0), mfma_waves * nm * 32768.0 / (best_ms * 1e-3) / 1e12, best_ms);
  return 0;
}

int main() {
  unsigned long long* d; CK(hipMalloc(&d, 256 * 8 * 2 * 8));
  float* sink; CK(hipMalloc(&sink, 64));
  for (int blocks : {1, 32, 256}) {
    run<4>("1 wave / SIMD, 4 independent accumulators", blocks, 256, 3, d, sink);
    run<1>("1 wave / SIMD, one dependent chain", blocks, 256, 3, d, sink);
    run<4>("2 waves / SIMD, both MFMA (4 accumulators each)", blocks, 512, 0, d, sink);
    run<1>("2 waves / SIMD, both MFMA (one chain each)", blocks, 512, 0, d, sink);
    run<4>("2 waves / SIMD, partner runs SiLU-like VALU", blocks, 512, 1, d, sink);
    run<1>("2 waves / SIMD, one chain + partner SiLU-like VALU", blocks, 512, 1, d, sink);
    run<4>("2 waves / SIMD, partner reads LDS (b128)", blocks, 512, 2, d, sink);
  }
  return 0;
}
