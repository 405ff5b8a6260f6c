// mde_common.h -- shared internals of libmde_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/mde_hip.h"

#define MDE_WAVE 64          // CDNA wavefront
#define MDE_BLOCK 256        // default workgroup: 4 waves, one per SIMD
#define MDE_MAX_PARTIALS 4096  // upper bound on workgroups that write reduction partials

void mde_set_error(const char* fmt, ...);
int mde_hip_fail(hipError_t e, const char* what, const char* file, int line);

#define MDE_HIP(call)                                                   \
  do {                                                                  \
    hipError_t e__ = (call);                                            \
    if (e__ != hipSuccess) return mde_hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

#define MDE_LAUNCH_CHECK()                                              \
  do {                                                                  \
    hipError_t e__ = hipGetLastError();                                 \
    if (e__ != hipSuccess) return mde_hip_fail(e__, "kernel launch", __FILE__, __LINE__); \
  } while (0)

static inline hipStream_t mde_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// number of workgroups for a memory-bound grid-stride kernel over `items` work items
static inline int mde_grid(int64_t items, int per_block, int max_blocks = 2048) {
  int64_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return (int)b;
}

// ---------------------------------------------------------------- device helpers
#if defined(__HIPCC__)

// sum across the 64 lanes of a wave; result valid in every lane (xor butterflies lower to
// DPP row_shr / row_bcast / ds_swizzle forms on gfx950 -- no LDS traffic).
template <typename T>
__device__ __forceinline__ T mde_wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <typename T>
__device__ __forceinline__ T mde_wave_max(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    T w = __shfl_xor(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}
// sum across an aligned group of G lanes (G power of two <= 64)
template <int G, typename T>
__device__ __forceinline__ T mde_group_sum(T v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of doubles for a 256-thread block; result valid in thread 0.
__device__ __forceinline__ double mde_block_sum(double v, double* smem /* >= 4 doubles */) {
  v = mde_wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  return r;
}
__device__ __forceinline__ double mde_block_max(double v, double* smem) {
  v = mde_wave_max(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    r = smem[0];
    for (int i = 1; i < nw; ++i) r = smem[i] > r ? smem[i] : r;
  }
  return r;
}

#endif  // __HIPCC__
