// Thread-local error string of the C ABI (bn_last_error).
#include <stdarg.h>
#include <stdio.h>
#include "brdfnerf_hip.h"

static thread_local char g_err[512] = "";

void bn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *bn_last_error(void) { return g_err; }
extern "C" int bn_abi_version(void) { return BN_ABI_VERSION; }
