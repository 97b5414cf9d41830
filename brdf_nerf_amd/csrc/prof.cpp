// Optional per-kernel timing with HIP events on the caller's stream (used by bench.py for the roofline line).
// Disabled by default: launch sites then cost one predictable branch.
#include <hip/hip_runtime.h>
#include <vector>
#include "brdfnerf_hip.h"
#include "prof.h"

struct ProfRec { int id; hipEvent_t a, b; };
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
bool g_bn_prof_on = false;

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  hipEventCreate(&e);
  return e;
}

void bn_prof_start(int id, hipStream_t st) {
  ProfRec r{id, get_event(), get_event()};
  hipEventRecord(r.a, st);
  g_recs.push_back(r);
}
void bn_prof_stop(hipStream_t st) { hipEventRecord(g_recs.back().b, st); }

extern "C" int bn_prof_enable(int on) {
  g_bn_prof_on = on != 0;
  return 0;
}
// Synchronises the recorded events, adds their durations per kernel id and clears the log.
extern "C" int bn_prof_collect(double *ms_sum, int *count, int n_ids) {
  for (int i = 0; i < n_ids; ++i) { ms_sum[i] = 0; count[i] = 0; }
  for (auto &r : g_recs) {
    hipEventSynchronize(r.b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, r.a, r.b);
    if (r.id >= 0 && r.id < n_ids) { ms_sum[r.id] += ms; count[r.id] += 1; }
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return 0;
}
