// Error plumbing + identification entry points of librf_flux.so.
#include <stdarg.h>
#include <stdlib.h>

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

// ---- kernel-class timing hook (bench.py's roofline) --------------------------------------------------
// While a profile is open every launch site constructs a ProfScope, which records ONE hipEvent on the launch stream
// in front of its kernel(s), tagged with the kernel class and its algorithmic work (FLOPs for MFMA kernels, bytes for
// row kernels).  rf_profile_end records a closing event; launch i's duration is event[i+1] - event[i]: kernel time
// plus the gap to the next launch, so the per-class sums add up EXACTLY to the wall time of the profiled region
// (one event per launch also halves the perturbation of an event pair).
// Bench-only: one profile at a time, one host thread, one stream; never open during hipGraph capture.
static struct {
  bool on = false;
  int cap = 0, n = 0, dropped = 0;
  hipEvent_t* ev = nullptr;   // [cap + 1]
  int* cls = nullptr;
  double* work = nullptr;
  hipStream_t last = nullptr;
} g_prof;

ProfScope::ProfScope(int cls, double work, hipStream_t s) : idx_(-1), s_(s) {
  if (!g_prof.on) return;
  if (g_prof.n >= g_prof.cap) { ++g_prof.dropped; return; }
  idx_ = g_prof.n++;
  g_prof.cls[idx_] = cls;
  g_prof.work[idx_] = work;
  g_prof.last = s;
  (void)hipEventRecord(g_prof.ev[idx_], s_);
}
ProfScope::~ProfScope() {}
void ProfScope::reclass(int cls) {
  if (idx_ >= 0) g_prof.cls[idx_] = cls;
}
bool prof_open() { return g_prof.on; }

static void prof_free() {
  for (int i = 0; i <= g_prof.cap; ++i)
    if (g_prof.ev && g_prof.ev[i]) (void)hipEventDestroy(g_prof.ev[i]);
  free(g_prof.ev); free(g_prof.cls); free(g_prof.work);
  g_prof.ev = nullptr; g_prof.cls = nullptr; g_prof.work = nullptr;
  g_prof.cap = g_prof.n = g_prof.dropped = 0;
  g_prof.on = false;
}

}  // namespace rf

extern "C" int rf_profile_begin(int32_t max_launches) {
  using namespace rf;
  RF_REQUIRE(!g_prof.on, RF_ERR_UNSUPPORTED, "rf_profile_begin: a profile is already open");
  RF_REQUIRE(max_launches > 0 && max_launches <= (1 << 20), RF_ERR_SHAPE, "rf_profile_begin: max_launches=%d", max_launches);
  g_prof.ev = (hipEvent_t*)calloc((size_t)max_launches + 1, sizeof(hipEvent_t));
  g_prof.cls = (int*)calloc(max_launches, sizeof(int));
  g_prof.work = (double*)calloc(max_launches, sizeof(double));
  RF_REQUIRE(g_prof.ev && g_prof.cls && g_prof.work, RF_ERR_HIP, "rf_profile_begin: out of host memory");
  g_prof.cap = max_launches;
  for (int i = 0; i <= max_launches; ++i) {
    hipError_t e = hipEventCreate(&g_prof.ev[i]);
    if (e != hipSuccess) { prof_free(); return hip_fail(e, "hipEventCreate"); }
  }
  g_prof.n = g_prof.dropped = 0;
  g_prof.last = nullptr;
  g_prof.on = true;
  return RF_OK;
}

extern "C" int rf_profile_end(double* us_sum, int64_t* launches, double* work_sum, int32_t* dropped) {
  using namespace rf;
  RF_REQUIRE(g_prof.on, RF_ERR_UNSUPPORTED, "rf_profile_end: no profile is open");
  RF_REQUIRE(us_sum && launches && work_sum, RF_ERR_NULL, "rf_profile_end: NULL output");
  g_prof.on = false;
  for (int c = 0; c < RF_KC_COUNT; ++c) { us_sum[c] = 0.0; launches[c] = 0; work_sum[c] = 0.0; }
  int rc = RF_OK;
  if (g_prof.n > 0) {
    hipError_t e = hipEventRecord(g_prof.ev[g_prof.n], g_prof.last);      // closing event behind the last launch
    if (e == hipSuccess) e = hipEventSynchronize(g_prof.ev[g_prof.n]);
    if (e != hipSuccess) rc = hip_fail(e, "rf_profile_end");
    for (int i = 0; i < g_prof.n && rc == RF_OK; ++i) {
      float ms = 0.f;
      e = hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]);
      if (e != hipSuccess) { rc = hip_fail(e, "rf_profile_end"); break; }
      const int c = g_prof.cls[i];
      us_sum[c] += (double)ms * 1000.0;
      launches[c] += 1;
      work_sum[c] += g_prof.work[i];
    }
  }
  if (dropped) *dropped = g_prof.dropped;
  prof_free();
  return rc;
}

extern "C" const char* rf_last_error(void) { return rf::g_err; }
extern "C" int rf_abi_version(void) { return 15; }
extern "C" int rf_target_arch(void) { return 950; }
