#pragma once
#include <hip/hip_runtime.h>
// kernel ids reported by bn_prof_collect
enum { BN_K_PACK = 0, BN_K_FWD_SIGMA, BN_K_FWD_FULL, BN_K_BWD_CHAIN, BN_K_WGRAD, BN_K_SKINNY, BN_K_COMPOSITE_FWD,
       BN_K_COMPOSITE_BWD, BN_K_GUIDED, BN_K_STRATIFIED, BN_K_ADAM, BN_K_BRDF, BN_K_ADJOINT, BN_K_ADJBWD, BN_K_WGRAD_REDUCE, BN_K_COUNT };
extern bool g_bn_prof_on;
void bn_prof_start(int id, hipStream_t st);
void bn_prof_stop(hipStream_t st);
struct BnProfScope {
  hipStream_t st; bool on;
  BnProfScope(int id, hipStream_t s) : st(s), on(g_bn_prof_on) { if (on) bn_prof_start(id, st); }
  ~BnProfScope() { if (on) bn_prof_stop(st); }
};
