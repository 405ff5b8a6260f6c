// rowprobe.hip -- how fast can the chip gather random 512-byte rows (d = 128 fp32) from a
// 256 MB table?  (design probe for the general-d kernel at BASELINE config 5; not part of the product)
//   ./rowprobe [rows_M] [reps]
// Each half-wave (32 lanes x 16 B) reads one row per load instruction; U loads in flight per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// idx: precomputed random row ids (4 B per gathered row, streamed like the nbr array)
template <int U, bool SORTED>
__global__ __launch_bounds__(256) void k_gather(const f4* __restrict__ X, const int* __restrict__ idx, int64_t ngather,
                                                int nrows, float* out) {
  const int lane = threadIdx.x & 63, half = lane >> 5, l32 = lane & 31;
  const int64_t w0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * 256) >> 6;
  f4 acc = {0, 0, 0, 0};
  // a wave consumes 2 * U rows per step
  for (int64_t g = w0 * (2 * U); g < ngather; g += nw * (2 * U)) {
    f4 v[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const int64_t k = g + 2 * q + half;
      const int r = idx[k < ngather ? k : 0];
      v[q] = X[(size_t)r * 32 + l32];
    }
#pragma unroll
    for (int q = 0; q < U; ++q) acc += v[q];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}
__global__ void k_idx(int64_t n, int nrows, int deg, int* idx) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    idx[i] = (int)(hash32((uint32_t)i * 2654435761u + 12345u) % (uint32_t)nrows);
}

template <class F>
static float timeit(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char** argv) {
  const int nrows = argc > 1 ? atoi(argv[1]) : 500000;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const int64_t ng = 40000000;
  f4* X; int* idx; float* out;
  CK(hipMalloc(&X, (size_t)nrows * 512)); CK(hipMemset(X, 0, (size_t)nrows * 512));
  CK(hipMalloc(&idx, ng * 4)); CK(hipMalloc(&out, 64));
  hipLaunchKernelGGL(k_idx, dim3(2048), dim3(256), 0, 0, ng, nrows, 80, idx);
  CK(hipDeviceSynchronize());
#define RUN(U, blocks) { float ms = timeit([&]() { hipLaunchKernelGGL((k_gather<U, false>), dim3(blocks), dim3(256), 0, 0, X, idx, ng, nrows, out); }, reps); \
    printf("rows=%d U=%-2d blocks=%-5d  %.3f ms  %.2f TB/s of row gathers (40M x 512 B)\n", nrows, U, blocks, ms, ng * 512.0 / (ms * 1e-3) / 1e12); }
  RUN(2, 2048); RUN(4, 2048); RUN(8, 2048); RUN(16, 2048); RUN(8, 4096); RUN(16, 1024); RUN(8, 1024);
  return 0;
}
