// Issue cost of the VALU instructions the attention kernels' elementwise phases are made of, in shader clocks per wave-instruction
// (one wave per SIMD, 8 groups of 4 independent instructions per loop trip, no memory traffic):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ inline f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
#define REP8(X) X X X X X X X X
#define BODY32(INS) REP8(INS) REP8(INS) REP8(INS) REP8(INS)
#define KERNEL(NAME, DECL, INS, ...)                                                              \
  __global__ void __launch_bounds__(256) NAME(unsigned long long* out, float* sink, int iters) {   \
    DECL;                                                                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                    \
    for (int it = 0; it < iters; ++it) { BODY32(asm volatile(INS : __VA_ARGS__);) }                        \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                    \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                               \
    if (sink == (float*)1) sink[0] = (float)t1;                                                    \
  }
#define D4 float a = threadIdx.x * 0.001f + 1.f, a1 = a + 1.f, a2 = a + 2.f, a3 = a + 3.f; float b = 1.0001f
#define Q4(OP) OP " %0, %0\n\t" OP " %1, %1\n\t" OP " %2, %2\n\t" OP " %3, %3"
#define Q4B(OP) OP " %0, %0, %4\n\t" OP " %1, %1, %4\n\t" OP " %2, %2, %4\n\t" OP " %3, %3, %4"
#define O4 "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b)
// four independent chains per kind: throughput, not latency
KERNEL(k_mul, D4, Q4B("v_mul_f32"), O4)
KERNEL(k_exp, D4, Q4("v_exp_f32"), O4)
KERNEL(k_rcp, D4, Q4("v_rcp_f32"), O4)
KERNEL(k_exp16, D4, Q4("v_exp_f16"), O4)
KERNEL(k_rcp16, D4, Q4("v_rcp_f16"), O4)
KERNEL(k_cvt, D4, Q4B("v_cvt_pk_bf16_f32"), O4)
KERNEL(k_pkmul, f2 a = mk2(threadIdx.x * 0.001f, 1.f); f2 a1 = a + 1.f; f2 a2 = a + 2.f; f2 a3 = a + 3.f; f2 b = mk2(1.0001f, 0.9999f), Q4B("v_pk_mul_f32"), O4)
KERNEL(k_acc, D4, "v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3", "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "a0", "a1", "a2", "a3")
// a transcendental next to independent plain ops: does it occupy the VALU for its whole duration?
KERNEL(k_exp_mul3, D4, "v_exp_f32 %0, %0\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4", O4)
KERNEL(k_exp2_mul2, D4, "v_exp_f32 %0, %0\n\tv_mul_f32 %1, %1, %4\n\tv_rcp_f32 %2, %2\n\tv_mul_f32 %3, %3, %4", O4)
#define RUN(NAME, WHAT, PER)                                                                         \
  do {                                                                                               \
    hipLaunchKernelGGL(NAME, dim3(256), dim3(256), 0, 0, d, (float*)nullptr, iters);                 \
    CK(hipDeviceSynchronize());                                                                      \
    CK(hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost));                                             \
    double s = 0; for (int i = 0; i < 256; ++i) s += (double)h[i];                                   \
    printf("%-44s %6.2f clocks per wave-instruction group (%d instr)\n", WHAT, s / 256 / (32.0 * iters), PER); \
  } while (0)
int main() {
  unsigned long long* d; CK(hipMalloc(&d, 256 * 8));
  unsigned long long h[256];
  const int iters = 2000;
  RUN(k_mul, "4 x v_mul_f32", 4); RUN(k_exp, "4 x v_exp_f32", 4); RUN(k_rcp, "4 x v_rcp_f32", 4);
  RUN(k_exp16, "4 x v_exp_f16", 4); RUN(k_rcp16, "4 x v_rcp_f16", 4); RUN(k_cvt, "4 x v_cvt_pk_bf16_f32", 4);
  RUN(k_pkmul, "4 x v_pk_mul_f32 (8 elements)", 4); RUN(k_acc, "2 x v_accvgpr_write + 2 x v_accvgpr_read", 4);
  RUN(k_exp_mul3, "v_exp_f32 + 3 v_mul_f32", 4); RUN(k_exp2_mul2, "v_exp_f32, v_mul_f32, v_rcp_f32, v_mul_f32", 4);
  return 0;
}
