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
