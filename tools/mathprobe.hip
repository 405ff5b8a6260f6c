// mathprobe.hip -- accuracy of the fp32 transcendental building blocks on gfx950 (design probe)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "../pymde_amd/csrc/mde_functions.h"
void mde_set_error(const char*, ...) {}
int mde_hip_fail(hipError_t, const char*, const char*, int) { return -5; }
__global__ void k(int n, const float* u, float* a, float* b, float* c) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = u[i];
  a[i] = mde_log1p(x);
  float t = 1.0f + x;
  float cc = x - (t - 1.0f);
  b[i] = fmaf(cc, mde_rcp(t), mde_log(t));       // compensated only
  c[i] = mde_log(t);                              // naive
}
int main() {
  const int n = 1 << 20;
  std::vector<float> hu(n);
  for (int i = 0; i < n; ++i) hu[i] = (float)exp(log(1e-7) + (log(1e4) - log(1e-7)) * i / (n - 1));
  float *u, *a, *b, *c;
  hipMalloc(&u, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
  hipMemcpy(u, hu.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, n, u, a, b, c);
  std::vector<float> ha(n), hb(n), hc(n);
  hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
  double ea = 0, eb = 0, ec = 0; float wa = 0, wb = 0;
  for (int i = 0; i < n; ++i) {
    double ref = log1p((double)hu[i]);
    double ra = fabs(ha[i] - ref) / ref, rb = fabs(hb[i] - ref) / ref, rc = fabs(hc[i] - ref) / ref;
    if (ra > ea) { ea = ra; wa = hu[i]; }
    if (rb > eb) { eb = rb; wb = hu[i]; }
    if (rc > ec) ec = rc;
  }
  printf("log1p max rel err: poly+compensated %.3g (u=%g) | compensated only %.3g (u=%g) | naive %.3g\n", ea, wa, eb, wb, ec);
  return 0;
}
