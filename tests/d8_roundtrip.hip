// Test program (built and run by tests/test_gpu_parity.py::test_d8_derivative_stash_round_trip; not part of the library):
// the 8-bit derivative stash of the 16-bit modes (csrc/field_kernels.h d8_pack4 / d8_unpack4) encoded and decoded on the
// device over a sweep of cosines, the two ReLU mask values and out-of-range / NaN inputs.
#include <cstdio>
#include <cmath>
#include <vector>
#include "field_kernels.h"

__global__ void roundtrip(const float *c, float *sine, float *relu, unsigned int *words, int n4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const unsigned int w = d8_pack4(c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]);
  words[i] = w;
  float k, b, d[4];
  d8_consts(BN_ACT_SIN, 1.f, k, b);
  d8_unpack4(w, k, b, d);
  for (int e = 0; e < 4; ++e) sine[4 * i + e] = d[e];
  d8_consts(BN_ACT_RELU, 1.f, k, b);
  d8_unpack4(w, k, b, d);
  for (int e = 0; e < 4; ++e) relu[4 * i + e] = d[e];
}

int main() {
  const int n = 1 << 16;
  std::vector<float> c(n);
  for (int i = 0; i < n; ++i) c[i] = -1.f + 2.f * (float)i / (float)(n - 1);
  // the last quads: ReLU masks, then a NaN / out-of-range value between two ordinary ones
  const float tail[12] = {0.f, 1.f, 1.f, 0.f, 0.25f, NAN, -0.5f, 0.75f, 0.25f, 3.f, -7.f, 0.75f};
  for (int i = 0; i < 12; ++i) c[n - 12 + i] = tail[i];
  float *dc, *ds, *dr;
  unsigned int *dw;
  hipMalloc(&dc, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dw, n);
  hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
  roundtrip<<<n / 4 / 256, 256>>>(dc, ds, dr, dw, n / 4);
  std::vector<float> s(n), r(n);
  if (hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost) != hipSuccess) {
    printf("FAIL: hip error\n");
    return 1;
  }
  double worst = 0., sum = 0., sq = 0.;
  for (int i = 0; i < n - 12; ++i) {
    const double e = (double)s[i] - (double)c[i];
    worst = fmax(worst, fabs(e)); sum += e; sq += e * e;
  }
  const int m = n - 12;
  printf("sine decode over %d values in [-1, 1]: max abs err %.6f (bound %.6f), mean err %+.2e, rms %.2e\n", m, worst, 128.5 / 32767., sum / m, sqrt(sq / m));
  bool ok = worst <= 128.5 / 32767. && fabs(sum / m) < 1e-4;
  printf("relu decode of (0, 1, 1, 0): %g %g %g %g\n", r[n - 12], r[n - 11], r[n - 10], r[n - 9]);
  ok = ok && r[n - 12] == 0.f && r[n - 11] == 1.f && r[n - 10] == 1.f && r[n - 9] == 0.f;
  printf("sine decode of (0.25, NaN, -0.5, 0.75): %g %g %g %g;  of (0.25, 3, -7, 0.75): %g %g %g %g\n", s[n - 8], s[n - 7], s[n - 6], s[n - 5],
         s[n - 4], s[n - 3], s[n - 2], s[n - 1]);
  const double tol = 128.5 / 32767.;
  ok = ok && fabs(s[n - 8] - 0.25) <= tol && fabs(s[n - 6] + 0.5) <= tol && fabs(s[n - 5] - 0.75) <= tol && std::isfinite(s[n - 7]) && fabs(s[n - 7]) <= tol;
  ok = ok && fabs(s[n - 4] - 0.25) <= tol && fabs(s[n - 1] - 0.75) <= tol && fabs(s[n - 3] - 1.) <= tol && fabs(s[n - 2] + 1.) <= tol;
  printf(ok ? "OK\n" : "FAIL\n");
  return ok ? 0 : 1;
}
