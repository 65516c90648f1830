// C-ABI plumbing shared by all entry points of libsdamd.so: error reporting and build info.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

int sdamd_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* sdamd_last_error() { return g_err; }

extern "C" int sdamd_abi_version() { return 7; }

// Number of visible HIP devices, or a negative HIP error code (no GPU / no driver).
extern "C" int sdamd_device_count() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { (void)hipGetLastError(); return -(int)e; }
  return n;
}
