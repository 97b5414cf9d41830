// Thread-local error string of the C ABI (bn_last_error).
#include <atomic>
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <utility>
#include "brdfnerf_hip.h"
#include "diag.h"

static thread_local char g_err[512] = "";

void bn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *bn_last_error(void) { return g_err; }
extern "C" int bn_abi_version(void) { return BN_ABI_VERSION; }

// sha256 prefix of the sources this library was compiled from (brdf_nerf_amd/build.py source_hash(), passed as -DBN_SOURCE_HASH):
// the loader refuses a library that is older than the tree beside it, build() recompiles on a mismatch whatever the mtimes say.
#ifndef BN_SOURCE_HASH
#define BN_SOURCE_HASH "unknown"
#endif
extern "C" const char *bn_source_hash(void) { return "BN_SOURCE_HASH=" BN_SOURCE_HASH + 15; }

// Every -D switch a source file of the library reacts to is declared in diag.h (variant builds pass the same defines to every file).
extern "C" const char *bn_build_flags(void) { return BN_BUILD_FLAGS_STRING; }

static std::atomic<int> g_deterministic{0};
int bn_deterministic() { return g_deterministic.load(std::memory_order_relaxed); }
extern "C" int bn_set_deterministic(int on) { return g_deterministic.exchange(on ? 1 : 0); }
extern "C" int bn_get_deterministic(void) { return g_deterministic.load(std::memory_order_relaxed); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of a kernel: remember what has been set for
// (device, kernel) so that a second device in the same process gets its own call and concurrent callers do not race.
int bn_configure_lds(const void *kernel, size_t lds, const char *what) {
  static std::mutex mu;
  static std::map<std::pair<int, const void *>, size_t> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { bn_set_error("%s: hipGetDevice failed", what); return BN_ELAUNCH; }
  std::lock_guard<std::mutex> lock(mu);
  size_t &have = done[std::make_pair(dev, kernel)];
  if (lds > have) {
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      bn_set_error("%s: cannot get %zu B of LDS: %s", what, lds, hipGetErrorString(e));
      return BN_ELAUNCH;
    }
    have = lds;
  }
  return 0;
}
