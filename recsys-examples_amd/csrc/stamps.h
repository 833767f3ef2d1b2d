// Phase stamps for profiling builds (-DMI355_STAMPS=1; never in the shipped library): thread 0 of a block writes the shader
// clock (s_memtime) at phase boundaries and the constant 100 MHz wall clock at its first and last stamp, into a device array
// a debug export copies out (tools/index_phase_stamps.py).  Compiles to nothing otherwise.
#pragma once
#ifndef MI355_STAMPS
#define MI355_STAMPS 0
#endif
#if MI355_STAMPS
#define STAMP_ARRAY(NAME, NBLK, NPH) __device__ unsigned long long NAME[(NBLK) * ((NPH) + 2)];
#define STAMP(NAME, NBLK, NPH, ph)                                                                       \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < (NBLK)) {                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
      NAME[blockIdx.x * ((NPH) + 2) + (ph)] = __builtin_amdgcn_s_memtime();                              \
      if ((ph) == 0) NAME[blockIdx.x * ((NPH) + 2) + (NPH)] = wall_clock64();                            \
      if ((ph) == (NPH) - 1) NAME[blockIdx.x * ((NPH) + 2) + (NPH) + 1] = wall_clock64();                \
      __builtin_amdgcn_sched_barrier(0);                                                                 \
    }                                                                                                    \
  } while (0)
#define STAMP_EXPORT(FN, NAME)                                                                           \
  extern "C" int FN(void* host, long long bytes) {                                                       \
    if (bytes > (long long)sizeof(NAME)) bytes = sizeof(NAME);                                           \
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(NAME), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; \
  }
#else
#define STAMP_ARRAY(NAME, NBLK, NPH)
#define STAMP(NAME, NBLK, NPH, ph) do { } while (0)
#define STAMP_EXPORT(FN, NAME)
#endif
