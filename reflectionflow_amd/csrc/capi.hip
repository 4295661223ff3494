// Error plumbing + identification entry points of librf_flux.so.
#include <stdarg.h>

#include "common.hpp"

namespace rf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return RF_ERR_HIP;
}

}  // namespace rf

extern "C" const char* rf_last_error(void) { return rf::g_err; }
extern "C" int rf_abi_version(void) { return 5; }
extern "C" int rf_target_arch(void) { return 950; }
