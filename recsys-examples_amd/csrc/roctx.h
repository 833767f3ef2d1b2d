// roctx ranges with the reference's op names (its NVTX ranges `op:<name>`, corelib/dynamicemb/dynamicemb/
// batched_dynamicemb_function.py:100-1233), so that a rocprofv3 --marker-trace of a step reads like the reference's nsys
// per-op tables.  libroctx64 is opened at run time (no link dependency); without it the ranges are no-ops.
#pragma once
#include <dlfcn.h>

namespace mi355 {

struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    void* h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) h = dlopen("libroctx64.so.4", RTLD_LAZY | RTLD_GLOBAL);
    if (h) {
      push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
    }
  }
};
inline Roctx& roctx() { static Roctx r; return r; }

struct RoctxRange {
  bool on;
  explicit RoctxRange(const char* name) {
    Roctx& r = roctx();
    on = r.push && r.pop;
    if (on) r.push(name);
  }
  ~RoctxRange() { if (on) roctx().pop(); }
};

}  // namespace mi355
