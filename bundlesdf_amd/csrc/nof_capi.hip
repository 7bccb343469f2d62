// Error plumbing of the C ABI (include/nof_hip.h).
#include "nof_common.h"

static thread_local char g_nof_err[512] = "";

int nof_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_nof_err, sizeof(g_nof_err), fmt, ap);
  va_end(ap);
  return code;
}

int nof_cu_count(void) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n = prop.multiProcessorCount;
    (void)hipGetLastError();
    cus = n;
  }
  return cus;
}

extern "C" const char* nof_last_error(void) { return g_nof_err; }
extern "C" int nof_version(void) { return NOF_ABI_VERSION; }
