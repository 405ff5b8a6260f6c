// bankprobe.hip -- LDS bank-conflict rules of gfx950 for 8-byte and 4-byte per-lane reads (design
// probe, not part of the product).  Sixteen waves per CU issue ds_read_b64 / ds_read_b32 with a given
// lane -> address pattern; clocks per instruction are derived from the wall time.
//   ./bankprobe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int WIDE>
__global__ __launch_bounds__(1024) void k_probe(const uint32_t* __restrict__ addr, int iters, float* out) {
  __shared__ __attribute__((aligned(16))) char L[160 * 1024 - 64];
  for (int i = threadIdx.x; i < (int)sizeof(L) / 4; i += 1024) reinterpret_cast<float*>(L)[i] = 1.0f;
  __syncthreads();
  const uint32_t a = addr[threadIdx.x & 63];
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (WIDE == 2) {
      float2 v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile(
          "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8\n\tds_read_b64 %2, %8\n\tds_read_b64 %3, %8\n\t"
          "ds_read_b64 %4, %8\n\tds_read_b64 %5, %8\n\tds_read_b64 %6, %8\n\tds_read_b64 %7, %8\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
          : "v"(a) : "memory");
      acc += v0.x + v1.x + v2.x + v3.x + v4.x + v5.x + v6.x + v7.y;
    } else {
      float v0, v1, v2, v3, v4, v5, v6, v7;
      asm volatile(
          "ds_read_b32 %0, %8\n\tds_read_b32 %1, %8\n\tds_read_b32 %2, %8\n\tds_read_b32 %3, %8\n\t"
          "ds_read_b32 %4, %8\n\tds_read_b32 %5, %8\n\tds_read_b32 %6, %8\n\tds_read_b32 %7, %8\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
          : "v"(a) : "memory");
      acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

static uint32_t* d_addr;
static float* d_out;

template <int WIDE>
static double run(const uint32_t* h) {
  CK(hipMemcpy(d_addr, h, 64 * 4, hipMemcpyHostToDevice));
  const int iters = 4000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k_probe<WIDE>, dim3(256), dim3(1024), 0, 0, d_addr, 100, d_out);
  CK(hipDeviceSynchronize());
  // two lengths: the difference removes launch and fill time
  float ms1, ms2;
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k_probe<WIDE>, dim3(256), dim3(1024), 0, 0, d_addr, iters, d_out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms1, a, b));
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k_probe<WIDE>, dim3(256), dim3(1024), 0, 0, d_addr, 3 * iters, d_out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms2, a, b));
  CK(hipGetLastError());
  return (ms2 - ms1) * 1e-3 * 2.4e9 / (2.0 * iters * 8 * 16);  // LDS clocks per wave instruction at 2.4 GHz, 16 waves per CU
}

int main() {
  CK(hipMalloc(&d_addr, 64 * 4)); CK(hipMalloc(&d_out, 64));
  uint32_t h[64];
  printf("== 8-byte reads (ds_read_b64); slot = 8 bytes\n");
  for (int l = 0; l < 64; ++l) h[l] = l * 8;
  printf("linear (lane -> slot lane):            %.2f clk\n", run<2>(h));
  for (int l = 0; l < 64; ++l) h[l] = 0;
  printf("broadcast (all lanes one slot):        %.2f clk\n", run<2>(h));
  for (int k = 2; k <= 128; k *= 2) {
    for (int l = 0; l < 64; ++l) h[l] = (uint32_t)l * 8u * k;
    printf("stride %3d slots:                      %.2f clk\n", k, run<2>(h));
  }
  // which lanes share a pass: lane l and lane l ^ g read the same banks at different rows
  for (int g = 1; g <= 32; g *= 2) {
    for (int l = 0; l < 64; ++l) h[l] = (l & g) ? (uint32_t)((l ^ g) * 8 + 32768) : (uint32_t)(l * 8);
    printf("partner l ^ %2d on the same banks:      %.2f clk\n", g, run<2>(h));
  }
  // n lanes of the first 16 / 32 on one bank pair (different rows), the rest linear
  for (int n = 2; n <= 8; n *= 2) {
    for (int l = 0; l < 64; ++l) h[l] = l * 8;
    for (int l = 0; l < n; ++l) h[l] = (uint32_t)(l * 4096);
    printf("%d lanes (0..%d) on one bank pair:       %.2f clk\n", n, n - 1, run<2>(h));
    for (int l = 0; l < 64; ++l) h[l] = l * 8;
    for (int l = 0; l < n; ++l) h[l * 8] = (uint32_t)(l * 4096);
    printf("%d lanes (0, 8, ..) on one bank pair:    %.2f clk\n", n, run<2>(h));
    for (int l = 0; l < 64; ++l) h[l] = l * 8;
    for (int l = 0; l < n; ++l) h[l * 16 % 64 + l * 16 / 64] = (uint32_t)(l * 4096);
    printf("%d lanes (0, 16, ..) on one bank pair:   %.2f clk\n", n, run<2>(h));
  }
  // random slots in 96 KB (the x_u pattern), several seeds
  for (int s = 0; s < 3; ++s) {
    srand(17 + s);
    for (int l = 0; l < 64; ++l) h[l] = (uint32_t)(rand() % 12288) * 8u;
    printf("random slots in 96 KB (seed %d):        %.2f clk\n", s, run<2>(h));
  }
  // offset of half a slot pair: does a b64 read at slot s use banks 2s, 2s+1 only?
  for (int l = 0; l < 64; ++l) h[l] = (uint32_t)(l * 8 + ((l & 1) ? 32768 - 8 : 0));
  printf("odd lanes one slot down, other row:    %.2f clk\n", run<2>(h));
  printf("== 4-byte reads (ds_read_b32); word = 4 bytes\n");
  for (int l = 0; l < 64; ++l) h[l] = l * 4;
  printf("linear:                                %.2f clk\n", run<1>(h));
  for (int k = 2; k <= 128; k *= 2) {
    for (int l = 0; l < 64; ++l) h[l] = (uint32_t)l * 4u * k;
    printf("stride %3d words:                      %.2f clk\n", k, run<1>(h));
  }
  for (int g = 1; g <= 32; g *= 2) {
    for (int l = 0; l < 64; ++l) h[l] = (l & g) ? (uint32_t)((l ^ g) * 4 + 32768) : (uint32_t)(l * 4);
    printf("partner l ^ %2d on the same bank:       %.2f clk\n", g, run<1>(h));
  }
  for (int s = 0; s < 3; ++s) {
    srand(17 + s);
    for (int l = 0; l < 64; ++l) h[l] = (uint32_t)(rand() % 24576) * 4u;
    printf("random words in 96 KB (seed %d):        %.2f clk\n", s, run<1>(h));
  }
  return 0;
}
