// thread-local last-error string of the C-ABI
#include "common.h"
#include <string.h>
static thread_local char g_err[256] = "";
extern "C" void mi355_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* mi355_last_error(void) { return g_err; }
extern "C" int mi355_abi_version(void) { return 1; }

// ---- live kernel timing for bench.py: while enabled, the library brackets the launches of its two bandwidth kernels
// (slot 0: the gather of the fused forward, slot 1: the fused reduce + optimizer kernel of the backward) with HIP events
// on the launch stream; mi355_profile_ms(slot) synchronises the pair of the last launch and returns its duration.
static bool g_prof_on = false;
static hipEvent_t g_prof_ev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
extern "C" int mi355_profile_kernels(int enable) {
  if (enable && !g_prof_ev[0][0]) {
    for (auto& pair : g_prof_ev)
      for (auto& e : pair)
        if (hipEventCreate(&e) != hipSuccess) { mi355_set_error("hipEventCreate failed"); return MI355_ELAUNCH; }
  }
  g_prof_on = enable != 0;
  return MI355_OK;
}
extern "C" float mi355_profile_ms(int slot) {
  float ms = -1.f;
  if (slot < 0 || slot > 1 || !g_prof_ev[slot][0] || hipEventSynchronize(g_prof_ev[slot][1]) != hipSuccess) return -1.f;
  if (hipEventElapsedTime(&ms, g_prof_ev[slot][0], g_prof_ev[slot][1]) != hipSuccess) return -1.f;
  return ms;
}
extern "C" void mi355i_prof_mark(int slot, int end, hipStream_t stream) {
  if (g_prof_on) hipEventRecord(g_prof_ev[slot][end], stream);
}
