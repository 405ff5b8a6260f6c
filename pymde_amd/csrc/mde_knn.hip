// mde_knn.hip -- exact k-nearest-neighbour search on a data matrix.
//   [ref: pymde/preprocess/data_matrix.py:91-178 k_nearest_neighbors -- sklearn brute force for
//    n < 10 000, pynndescent (approximate, un-vendored) above; here: exact at every size]
// Squared distances are formed as |x|^2 + |y|^2 - 2 x.y with the Gram tile x.y on the f32 matrix
// cores (v_mfma_f32_32x32x2_f32, exact f32): a 256-thread workgroup owns 64 query rows and walks
// the candidates 64 at a time; each wave accumulates one 32x32 quadrant of the 64x64 tile over the
// features, staged through LDS in 32-wide chunks (rows padded to 33 floats: conflict-free operand
// reads; the next chunk's global loads overlap the current chunk's MFMAs).  The tile of squared distances is parked in LDS and one thread per query row merges its
// 64 candidates into the row's sorted top-k list (insertion only when a candidate beats the current
// k-th best, which becomes rare quickly).  Self matches are excluded by index.
#include "mde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KNN_BM 64
#define KNN_BN 64
#define KNN_KB 32
#define KNN_KBP 33
#define KNN_MAXK 64

__global__ __launch_bounds__(MDE_BLOCK) void k_row_sqnorm(int64_t n, int nf, const float* __restrict__ X,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * MDE_BLOCK) >> 6;
  for (int64_t r = w0; r < n; r += nw) {
    float s = 0.0f;
    for (int c = lane; c < nf; c += 64) {
      const float v = X[r * nf + c];
      s = fmaf(v, v, s);
    }
    s = mde_wave_sum(s);
    if (lane == 0) out[r] = s;
  }
}

__global__ __launch_bounds__(MDE_BLOCK) void k_knn(int n, int nf, int k, const float* __restrict__ X,
                                                   const float* __restrict__ sqn,
                                                   int32_t* __restrict__ idx_out,
                                                   float* __restrict__ d2_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sA = lds;                           // [KNN_BM][KNN_KBP]
  float* sB = sA + KNN_BM * KNN_KBP;         // [KNN_BN][KNN_KBP]
  float* sD = sB + KNN_BN * KNN_KBP;         // [KNN_BM][KNN_BN + 1] squared distances of the tile
  float* bestd = sD + KNN_BM * (KNN_BN + 1); // [KNN_BM][k]
  int* besti = reinterpret_cast<int*>(bestd + KNN_BM * k);  // [KNN_BM][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;   // quadrant of the 64x64 tile
  const int row0 = blockIdx.x * KNN_BM;
  for (int i = tid; i < KNN_BM * k; i += MDE_BLOCK) {
    bestd[i] = 3.402823466e+38f;
    besti[i] = -1;
  }
  float worst = 3.402823466e+38f;            // thread t < 64: current k-th best of row t
  const int li = lane & 31, lk = lane >> 5;
  for (int col0 = 0; col0 < n; col0 += KNN_BN) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    // feature chunks of KNN_KB: the next chunk's global loads are issued before the MFMAs of the
    // current one and committed to LDS after them (register double buffering), so the matrix
    // cores do not wait for the staging latency
    constexpr int STG = (KNN_BM * KNN_KB) / MDE_BLOCK;
    float ra[STG], rb[STG];
    auto fetch = [&](int k0) {
#pragma unroll
      for (int q = 0; q < STG; ++q) {
        const int e = tid + q * MDE_BLOCK;
        const int r = e >> 5, c = e & 31;
        const int gr = row0 + r, gc = col0 + r, f = k0 + c;
        // plain loads from clamped addresses, zeroed afterwards: a predicated load is a branch around
        // it and the sixteen loads of a chunk would go out one memory latency after the other
        const int fc = f < nf ? f : nf - 1;
        const float va = X[(int64_t)(gr < n ? gr : n - 1) * nf + fc];
        const float vb = X[(int64_t)(gc < n ? gc : n - 1) * nf + fc];
        ra[q] = (gr < n && f < nf) ? va : 0.0f;
        rb[q] = (gc < n && f < nf) ? vb : 0.0f;
      }
    };
    fetch(0);
    for (int k0 = 0; k0 < nf; k0 += KNN_KB) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < STG; ++q) {
        const int e = tid + q * MDE_BLOCK;
        const int r = e >> 5, c = e & 31;
        sA[r * KNN_KBP + c] = ra[q];
        sB[r * KNN_KBP + c] = rb[q];
      }
      __syncthreads();
      if (k0 + KNN_KB < nf) fetch(k0 + KNN_KB);
      const float* pa = sA + (wi * 32 + li) * KNN_KBP + lk;
      const float* pb = sB + (wj * 32 + li) * KNN_KBP + lk;
#pragma unroll
      for (int kk = 0; kk < KNN_KB; kk += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[kk], pb[kk], acc, 0, 0, 0);
    }
    // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int r = wi * 32 + (q & 3) + 8 * (q >> 2) + 4 * lk;
      const int c = wj * 32 + li;
      const int gr = row0 + r, gc = col0 + c;
      float d2 = 3.402823466e+38f;
      if (gr < n && gc < n && gr != gc) d2 = fmaxf(sqn[gr] + sqn[gc] - 2.0f * acc[q], 0.0f);
      sD[r * (KNN_BN + 1) + c] = d2;
    }
    __syncthreads();
    if (tid < KNN_BM) {
      float* bd = bestd + tid * k;
      int* bi = besti + tid * k;
      for (int c = 0; c < KNN_BN; ++c) {
        const float d2 = sD[tid * (KNN_BN + 1) + c];
        if (d2 < worst) {
          int pos = k - 1;
          while (pos > 0 && bd[pos - 1] > d2) {
            bd[pos] = bd[pos - 1];
            bi[pos] = bi[pos - 1];
            --pos;
          }
          bd[pos] = d2;
          bi[pos] = col0 + c;
          worst = bd[k - 1];
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < KNN_BM * k; i += MDE_BLOCK) {
    const int r = i / k, gr = row0 + r;
    if (gr < n) {
      idx_out[(int64_t)gr * k + (i % k)] = besti[i];
      d2_out[(int64_t)gr * k + (i % k)] = bestd[i];
    }
  }
}

// idx_out [n, k] int32 (-1 where fewer than k other items exist), d2_out [n, k] squared Euclidean
// distances, ascending per row.  sqn_work: n floats of scratch.
extern "C" int mde_knn(int64_t n, int32_t nf, const float* data, int32_t k, int32_t* idx_out,
                       float* d2_out, float* sqn_work, void* stream) {
  if (n <= 0 || nf <= 0 || k <= 0 || k > KNN_MAXK || !data || !idx_out || !d2_out || !sqn_work) {
    mde_set_error("mde_knn: invalid arguments (1 <= k <= %d)", KNN_MAXK);
    return MDE_E_INVALID;
  }
  if (n >= ((int64_t)1 << 31)) return MDE_E_TOO_LARGE;
  hipStream_t st = mde_stream(stream);
  hipLaunchKernelGGL(k_row_sqnorm, dim3(mde_grid(n * 64, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, n, nf,
                     data, sqn_work);
  MDE_LAUNCH_CHECK();
  const size_t lds = sizeof(float) * (size_t)(KNN_BM * KNN_KBP + KNN_BN * KNN_KBP + KNN_BM * (KNN_BN + 1)) +
                     (size_t)KNN_BM * k * (sizeof(float) + sizeof(int));
  static bool attr = false;
  if (!attr) {
    MDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(k_knn, dim3((unsigned)((n + KNN_BM - 1) / KNN_BM)), dim3(MDE_BLOCK), lds, st, (int)n,
                     nf, k, data, sqn_work, idx_out, d2_out);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// Directed neighbour lists -> edge list for mde_edges_count_unique: pairs_out[r * k + c] = (r, idx[r][c]).
// Empty slots (idx < 0) and, when `val` is given, entries with val > max_value become the self pair
// (r, r), which the edge counter drops [ref: data_matrix.py:147-175 -- neighbours beyond max_distance
// get weight 0 and vanish from the graph].
__global__ __launch_bounds__(MDE_BLOCK) void k_knn_pairs(int64_t total, int k, const int32_t* __restrict__ idx,
                                                         const float* __restrict__ val, float max_value,
                                                         int64_t* __restrict__ pairs) {
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    const int64_t r = i / k;
    int64_t j = idx[i];
    if (j < 0 || (val && !(val[i] <= max_value))) j = r;
    reinterpret_cast<longlong2*>(pairs)[i] = make_longlong2((long long)r, (long long)j);
  }
}
extern "C" int mde_knn_pairs(int64_t n, int32_t k, const int32_t* idx, const float* val, float max_value,
                             int64_t* pairs_out, void* stream) {
  if (n <= 0 || k <= 0 || !idx || !pairs_out) return MDE_E_INVALID;
  hipLaunchKernelGGL(k_knn_pairs, dim3(mde_grid(n * k, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, mde_stream(stream),
                     n * (int64_t)k, k, idx, val, max_value, pairs_out);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
