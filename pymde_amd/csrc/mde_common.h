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
// mde_plan.partials: [0, MDE_MAX_PARTIALS) loss partials | [MAX] arrival ticket | [MAX + 1] scratch flag |
// [MAX + 2] the loss of the last evaluation in DOUBLE (round 6: every finaliser writes it beside the float -- a
// row-sharded solve sums the ranks' shares before the one rounding to float, mde_plan_loss_double)
#define MDE_PARTIALS_LOSS_D (MDE_MAX_PARTIALS + 2)
#define MDE_PARTIALS_DOUBLES (MDE_MAX_PARTIALS + 4)

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
// Grid-wide "am I the last workgroup?" (the final reduction then needs no second launch).  The L2s of
// the eight XCDs are not coherent with each other, and a device-scope fence (__threadfence) writes a
// whole L2 back -- measured at ~8 us per kernel here.  So the partials are exchanged with device-scope
// RELAXED atomic stores / loads instead (write-through / L2-bypassing accesses of just those words):
// every block writes its partials with mde_st_partial, the barrier below waits for their completion,
// thread 0 takes a ticket, and the one block that arrives last reads all partials with mde_ld_partial.
// Returns true in every thread of that block.  `ticket` is zero between launches.
__device__ __forceinline__ void mde_st_partial(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double mde_ld_partial(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool mde_last_block(unsigned int* ticket) {
  __shared__ int mde_last_flag;
  // every thread's partial stores must have completed before thread 0 takes the ticket (a
  // workgroup-scope barrier alone does not wait for global stores when the workgroup sits on one CU)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int v = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    mde_last_flag = (v == gridDim.x * gridDim.y - 1u) ? 1 : 0;
    if (mde_last_flag) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  return mde_last_flag != 0;
}

// Final reduction by the last workgroup (256 threads): out[q] = sum (or max, bit q of max_mask) over
// b of partial[q * nb + b], q < nq.  A wave takes four rows at a time and issues all their loads
// before the first use (the partials come from memory: one latency per batch, not per row); the
// order of the additions is fixed.  Meant for a handful of rows and nb <= 256.
__device__ __forceinline__ void mde_final_rows(int nq, int nb, const double* partial, double* out,
                                               unsigned long long max_mask, double scale = 1.0) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int q0 = wave; q0 < nq; q0 += 4 * nw) {
    double acc[4];
    bool is_max[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + r * nw;
      is_max[r] = q < 64 && ((max_mask >> q) & 1ull);
      acc[r] = is_max[r] ? -1.0e308 : 0.0;
    }
    for (int k = lane; k < nb; k += 64) {
      double v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = q0 + r * nw;
        v[r] = (q < nq) ? mde_ld_partial(partial + (int64_t)q * nb + k) : 0.0;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = is_max[r] ? (v[r] > acc[r] ? v[r] : acc[r]) : acc[r] + v[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = q0 + r * nw;
      const double t = is_max[r] ? mde_wave_max(acc[r]) : mde_wave_sum(acc[r]);
      if (lane == 0 && q < nq) out[q] = t * scale;
    }
  }
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
