// gprobe.hip -- random-gather rate of the MI355X memory hierarchy (design probe, not product).
// 100M gathers of 8 B from a table of T bytes, (a) every workgroup over the whole table,
// (b) workgroup b restricted to slice b % 8 of the table (XCD-affine slices), with 1 or 4
// independent gathers in flight per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// idx[h]: uniform in the whole table (sliced=0) or in slice (block_of(h) % 8) (sliced=1)
__global__ void k_gen(int64_t H, int64_t nv, int sliced, int per_block, int* idx, float* w) {
  for (int64_t h = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; h < H; h += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t r = splitmix(h);
    int64_t v;
    if (sliced) {
      const int64_t blk = h / per_block;
      const int64_t sl = nv / 8;
      v = (blk % 8) * sl + (int64_t)(r % (uint64_t)sl);
    } else {
      v = (int64_t)(r % (uint64_t)nv);
    }
    idx[h] = (int)v;
    w[h] = 1.0f;
  }
}
// block b processes the contiguous chunk [b*per_block, (b+1)*per_block)
template <int U>
__global__ __launch_bounds__(256) void k_gather(int per_block, const int* __restrict__ idx,
                                                const float* __restrict__ w, const float2* __restrict__ X,
                                                float* out) {
  const int64_t base = (int64_t)blockIdx.x * per_block;
  float acc = 0;
  for (int i = threadIdx.x; i < per_block; i += 256 * U) {
    int u[U];
    float ww[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int j = i + q * 256;
      u[q] = j < per_block ? idx[base + j] : 0;
      ww[q] = j < per_block ? w[base + j] : 0.f;
    }
    float2 x[U];
#pragma unroll
    for (int q = 0; q < U; ++q) x[q] = X[u[q]];
#pragma unroll
    for (int q = 0; q < U; ++q) acc += ww[q] * (x[q].x + x[q].y);
  }
  if (acc == 123.456f) out[0] = acc;
}
template <class F>
static double time_ms(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}
int main() {
  const int64_t H = 100000000;
  const int per_block = 12208;  // 8192 blocks
  const int nblocks = (int)((H + per_block - 1) / per_block);
  const int64_t Hp = (int64_t)nblocks * per_block;
  int* idx; float* w; float2* X; float* out;
  CK(hipMalloc(&idx, Hp * 4)); CK(hipMalloc(&w, Hp * 4)); CK(hipMalloc(&X, (size_t)256 << 20)); CK(hipMalloc(&out, 64));
  CK(hipMemset(X, 0, (size_t)256 << 20));
  const double mbs[] = {0.5, 1, 2, 4, 8, 16, 64, 256};
  for (int sliced = 0; sliced < 2; ++sliced)
    for (double mb : mbs) {
      const int64_t nv = (int64_t)(mb * 1048576 / 8);
      hipLaunchKernelGGL(k_gen, dim3(4096), dim3(256), 0, 0, Hp, nv, sliced, per_block, idx, w);
      CK(hipDeviceSynchronize());
      double t1 = time_ms([&]() { hipLaunchKernelGGL(k_gather<1>, dim3(nblocks), dim3(256), 0, 0, per_block, idx, w, X, out); }, 10);
      double t4 = time_ms([&]() { hipLaunchKernelGGL(k_gather<4>, dim3(nblocks), dim3(256), 0, 0, per_block, idx, w, X, out); }, 10);
      printf("%s table %6.1f MB: U1 %.3f ms (%.0f G/s)  U4 %.3f ms (%.0f G/s)\n", sliced ? "sliced(b%8)" : "whole     ", mb,
             t1, Hp / t1 / 1e6, t4, Hp / t4 / 1e6);
    }
  return 0;
}
