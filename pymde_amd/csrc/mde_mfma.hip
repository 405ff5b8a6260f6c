// mde_mfma.hip -- the Standardized projections at embedding widths 5 .. 128 (register tiles of 32, 64, 128 columns): exact-f32 matrix
// core kernels (v_mfma_f32_32x32x2_f32 runs at the f32 vector rate, 64 FLOP/clk/SIMD; at d = 128 the
// d x d Gram matrix and the n x d by d x d product are ~16 GFLOP each for n = 500k, i.e. as long
// on the matrix pipe as their operands take to stream from HBM -- the kernels below keep both busy).
//   [ref: pymde/constraints.py:167-200 (Standardized), pymde/util.py:129-171 (proj_standardized)]
//
// k_gram_rows     G = (A - mean)^T (B - mean) over a chunk of rows.  One wave owns ALL W x W output
//                 tiles (W = d / 32): lane (k, c) = (l >> 5, l & 31) loads W consecutive floats of
//                 row r + k -- one fully coalesced 16-byte load per operand at d = 128 -- and holds
//                 columns W c .. W c + W - 1.  The k index of the MFMA is the row, so register q of
//                 the A-side and register q' of the B-side feed MFMA (q, q'): 16 MFMAs per two
//                 16-byte loads, no LDS, no transposes.  Output tile (q, q'), register v, lane l is
//                 G[W i + q][W (l & 31) + q'] with i = (v & 3) + 8 (v >> 2) + 4 (l >> 5).
// k_rmul_rows     out = base + alpha (A - mean) M, 32 rows per wave.  The product is formed transposed
//                 (M^T as the 32 x 2 operand, the rows as the 2 x 32 operand): lane l keeps HALF of
//                 row l & 31 (columns (l >> 5) d/2 ...) in registers -- 16-byte loads along the row --
//                 and step s multiplies inner indices s and d/2 + s.  M sits in LDS as fp32; a lane
//                 ends up with 4-column groups of its row and stores them as float4.
// invsqrt         coupled Newton-Schulz in double, two launches per step, convergence read back per
//                 batch of steps (see mde_invsqrt_grid).
#include "mde_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Round 6: the kernels take ANY width d <= 32 W (W = 1, 2, 4: d <= 32, 64, 128) -- rows are d floats apart, the columns
// d .. 32 W - 1 are zero in registers and never stored; the d x d matrices (Gram, M) and the column means are compact.
// Rows then start at any float: the vector accesses below are 4-byte aligned only (gfx950 runs global accesses in
// unaligned mode; hipcc keeps the wide instructions for these types).  Until then every other width ran the round-1
// generic kernels: Standardized at n = 500k took 8.0 / 6.6 ms per tangent projection / retraction at d = 100 against
// 0.52 / 0.64 at d = 128 (tools/r6_proj_sweep.py).
typedef float mf4 __attribute__((ext_vector_type(4)));
typedef float mf2 __attribute__((ext_vector_type(2)));
typedef mf4 mf4u __attribute__((aligned(4)));
typedef mf2 mf2u __attribute__((aligned(4)));

// W consecutive floats at p = columns col0 .. col0 + W - 1 of a row with `valid` columns; pad[] beyond the row's end
template <int W, bool EXACT>
__device__ __forceinline__ void ldw(const float* p, int col0, int valid, const float (&pad)[W], float (&v)[W]) {
  if (EXACT || col0 + W <= valid) {
    if constexpr (W == 4) {
      const mf4 t = *reinterpret_cast<const mf4u*>(p);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else if constexpr (W == 2) {
      const mf2 t = *reinterpret_cast<const mf2u*>(p);
      v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
      for (int q = 0; q < W; ++q) v[q] = p[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < W; ++q) v[q] = col0 + q < valid ? p[q] : pad[q];
  }
}

// ---------------------------------------------------------------- column sums (means)
template <int W, bool EXACT>
__global__ __launch_bounds__(MDE_BLOCK) void k_colsum_rows(int64_t n, int ld_, int d_, const float* __restrict__ Z, int64_t rows_per_wg,
                                                           double* __restrict__ partial /* [gridDim.x][d] */) {
  constexpr int D = 32 * W;
  const int d = EXACT ? D : d_;    // (EXACT: the width is the register tile's -- rounds 3-5's kernels, no guards)
  const int ld = EXACT ? D : ld_;  // floats between rows (> d when Z is a block of columns of a wider matrix)
  float zero[W];
#pragma unroll
  for (int q = 0; q < W; ++q) zero[q] = 0.0f;
  __shared__ double red[MDE_BLOCK / 64][64][W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kk = lane >> 5, c = lane & 31;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < n ? r0 + rows_per_wg : n;
  double acc[W];
#pragma unroll
  for (int q = 0; q < W; ++q) acc[q] = 0.0;
  constexpr int U = 8;
  for (int64_t r = r0 + 2 * wave; r < r1; r += 8 * U) {
    float a[U][W];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + 8 * u + kk;
      if (rr < r1) {
        ldw<W, EXACT>(Z + rr * ld + W * c, W * c, d, zero, a[u]);
      } else {
#pragma unroll
        for (int q = 0; q < W; ++q) a[u][q] = 0.0f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int q = 0; q < W; ++q) acc[q] += (double)a[u][q];
  }
#pragma unroll
  for (int q = 0; q < W; ++q) red[wave][lane][q] = acc[q];
  __syncthreads();
  for (int i = threadIdx.x; i < D; i += MDE_BLOCK) {
    const int cc = i / W, q = i % W;
    double s = 0.0;
    for (int w = 0; w < MDE_BLOCK / 64; ++w) s += red[w][cc][q] + red[w][32 + cc][q];
    if (i < d) partial[(int64_t)blockIdx.x * d + i] = s;
  }
}

// ---------------------------------------------------------------- Gram matrix
// (The W x W tiles are split between the waves: wave w takes A-side register q = w mod W against all W
// B-side registers -- 4 tiles = 64 accumulator registers at d = 128, which leaves room for a second
// register set of operands in flight and for two or three waves per SIMD; sixteen tiles in one wave
// spill, eight still do under hipcc.  The B-side rows are then read by W waves: L1 hits.)
template <int W, bool SAME, bool EXACT>
__global__ __launch_bounds__(MDE_BLOCK, 2) void k_gram_rows(int64_t n, int ld_, int wa_, int wb_, const float* __restrict__ A,
                                                         const float* __restrict__ B,
                                                         const double* __restrict__ meanA, const double* __restrict__ meanB,
                                                         int64_t rows_per_wg,
                                                         double* __restrict__ partial /* [gridDim.x][wa * wb] */) {
  static_assert(W == 1 || W == 2 || W == 4, "A-side halves of equal size");
  constexpr int D = 32 * W;
  // (A and B may be blocks of <= D columns of wider matrices: wa / wb valid columns, rows ld floats apart)
  const int ld = EXACT ? D : ld_, wa = EXACT ? D : wa_, wb = EXACT ? D : wb_;
  constexpr int NH = W;                       // parts of the A-side registers: one register per wave
  constexpr int QH = W / NH;                  // A-side registers per part
  constexpr int NRG = (MDE_BLOCK / 64) / NH;  // wave groups interleaving the row pairs
  __shared__ float red[W * W * 16 * 64];
  const int lane = threadIdx.x & 63, kk = lane >> 5, c = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q0 = (wave % NH) * QH, rg = wave / NH;  // my A-side columns W c + q0 .. + QH - 1 (an address offset only)
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < n ? r0 + rows_per_wg : n;
  float mub[W], mua[QH];
#pragma unroll
  for (int p = 0; p < W; ++p) mub[p] = (meanB && W * c + p < wb) ? (float)meanB[W * c + p] : 0.0f;
#pragma unroll
  for (int q = 0; q < QH; ++q) mua[q] = (meanA && W * c + q0 + q < wa) ? (float)meanA[W * c + q0 + q] : 0.0f;
  f32x16 acc[QH][W];
#pragma unroll
  for (int q = 0; q < QH; ++q)
#pragma unroll
    for (int p = 0; p < W; ++p)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[q][p][v] = 0.0f;
  // Two register sets: the loads of the next batch of row pairs are in flight while the MFMAs of this
  // one run.  (A == B: the A-side values are loaded again -- an L1 hit -- rather than picked out of the
  // B-side registers with a wave-dependent register index.)  (Padded widths: the guarded loads need registers of
  // their own -- batches of four row pairs instead of eight keep the 128-column tile out of scratch memory.)
  constexpr int U = EXACT ? 8 : 4;
  const float* Aq = A + W * c + q0;
  const float* Bq = B + W * c;
  auto load = [&](int64_t r, float (&a)[U][QH], float (&b)[U][W]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t rr = r + 2 * NRG * u + kk;
      if (rr < r1) {
        ldw<W, EXACT>(Bq + rr * ld, W * c, wb, mub, b[u]);   // (padding = the mean: centred to zero below)
        ldw<QH, EXACT>(Aq + rr * ld, W * c + q0, wa, mua, a[u]);
      } else {
#pragma unroll
        for (int p = 0; p < W; ++p) b[u][p] = mub[p];  // (centred to zero below)
#pragma unroll
        for (int q = 0; q < QH; ++q) a[u][q] = mua[q];
      }
    }
  };
  auto mma = [&](float (&a)[U][QH], float (&b)[U][W]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int p = 0; p < W; ++p) b[u][p] -= mub[p];
#pragma unroll
      for (int q = 0; q < QH; ++q) a[u][q] -= mua[q];
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int p = 0; p < W; ++p)
          acc[q][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][q], b[u][p], acc[q][p], 0, 0, 0);
    }
  };
  float a0[U][QH], b0[U][W], a1[U][QH], b1[U][W];
  constexpr int STEP = 2 * NRG * U;  // rows per batch of a wave group
  int64_t r = r0 + 2 * rg;
  if (r < r1) load(r, a0, b0);
  while (r < r1) {
    load(r + STEP, a1, b1);
    mma(a0, b0);
    r += STEP;
    if (r >= r1) break;
    load(r + STEP, a0, b0);
    mma(a1, b1);
    r += STEP;
  }
  // add the wave groups' tiles in order (fixed order: reproducible), then write the chunk's Gram
  // matrix in double
  for (int g = 0; g < NRG; ++g) {
    if (rg == g) {
#pragma unroll
      for (int q = 0; q < QH; ++q)
#pragma unroll
        for (int p = 0; p < W; ++p)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int idx = ((((q0 + q) * W + p) * 16 + v) << 6) + lane;
            red[idx] = (g == 0 ? 0.0f : red[idx]) + acc[q][p][v];
          }
    }
    __syncthreads();
  }
  double* out = partial + (int64_t)blockIdx.x * wa * wb;
  for (int idx = threadIdx.x; idx < W * W * 16 * 64; idx += MDE_BLOCK) {
    const int l = idx & 63, v = (idx >> 6) & 15, t = idx >> 10;
    const int q = t / W, p = t % W;
    const int i = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5);
    const int row = W * i + q, col = W * (l & 31) + p;
    if (row < wa && col < wb) out[row * wb + col] = (double)red[idx];
  }
}

// out[q] = scale * sum_c partial[c * m + q]: a block takes 32 consecutive outputs, eight threads per
// output add every eighth chunk and the eight sums are added in a fixed order
__global__ __launch_bounds__(MDE_BLOCK) void k_sum_chunks(int64_t m, int nc, const double* __restrict__ partial,
                                                          double scale, double* __restrict__ out) {
  __shared__ double sh[8][32];
  const int t = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t q = (int64_t)blockIdx.x * 32 + t;
  double s = 0.0;
  if (q < m)
    for (int c = sl; c < nc; c += 8) s += partial[(int64_t)c * m + q];
  sh[sl][t] = s;
  __syncthreads();
  if (sl == 0 && q < m) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += sh[k][t];
    out[q] = tot * scale;
  }
}

// the same for a wa x wb block that lands inside a wider matrix: out[(q / wb) * ld + q % wb]
__global__ __launch_bounds__(MDE_BLOCK) void k_sum_chunks_blk(int wa, int wb, int nc, const double* __restrict__ partial,
                                                              double scale, double* __restrict__ out, int ld) {
  __shared__ double sh[8][32];
  const int t = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t m = (int64_t)wa * wb;
  const int64_t q = (int64_t)blockIdx.x * 32 + t;
  double s = 0.0;
  if (q < m)
    for (int c = sl; c < nc; c += 8) s += partial[(int64_t)c * m + q];
  sh[sl][t] = s;
  __syncthreads();
  if (sl == 0 && q < m) {
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += sh[k][t];
    out[(q / wb) * ld + q % wb] = tot * scale;
  }
}

// ---------------------------------------------------------------- right-multiplication
template <int W, bool EXACT>
__global__ __launch_bounds__(MDE_BLOCK) void k_rmul_rows(int64_t n, int ld_, int wk_, int wj_, const float* A,
                                                         const double* __restrict__ M, int ldm_,
                                                         const double* __restrict__ mean, float alpha,
                                                         const float* base, float* out) {
  constexpr int D = 32 * W, KH = D / 2;
  // (out[:, 0..wj) = base + alpha (A[:, 0..wk) - mean) M[0..wk, 0..wj): blocks of wider matrices when ld > wk, wj)
  const int ld = EXACT ? D : ld_, wk = EXACT ? D : wk_, wj = EXACT ? D : wj_, ldm = EXACT ? D : ldm_;
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [D][D] fp32 copy of M (zero beyond d), then [D] column means
  float* smean = sm + D * D;
  for (int i = threadIdx.x; i < D * D; i += MDE_BLOCK) {
    const int r = i / D, c = i % D;
    sm[i] = (r < wk && c < wj) ? (float)M[r * ldm + c] : 0.0f;
  }
  for (int i = threadIdx.x; i < D; i += MDE_BLOCK) smean[i] = (mean && i < wk) ? (float)mean[i] : 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, rn = lane & 31;
  const float* smh = sm + (size_t)h * KH * D + rn;
  const int64_t ntiles = (n + 31) >> 5;
  const int64_t tstep = (int64_t)gridDim.x * (MDE_BLOCK / 64);
  // my half of my row (inner indices h KH .. h KH + KH - 1), as float4 loads along the row; the
  // next tile's rows are requested before this tile's MFMAs start
  mf4 nx[KH / 4];
  auto fetch = [&](int64_t tile) __attribute__((always_inline)) {
    const int64_t row = tile * 32 + rn;
    const float* ap = A + (row < n ? row : 0) * ld + h * KH;
#pragma unroll
    for (int t = 0; t < KH / 4; ++t) {
      const int c0 = h * KH + 4 * t;
      if (EXACT || c0 + 4 <= wk) {
        nx[t] = *reinterpret_cast<const mf4u*>(ap + 4 * t);
      } else {
        nx[t] = mf4{c0 < wk ? ap[4 * t] : 0.0f, c0 + 1 < wk ? ap[4 * t + 1] : 0.0f, c0 + 2 < wk ? ap[4 * t + 2] : 0.0f, 0.0f};
      }
    }
  };
  int64_t tile = (int64_t)blockIdx.x * (MDE_BLOCK / 64) + wave;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += tstep) {
    const int64_t row = tile * 32 + rn;
    const bool ok = row < n;
    float a[KH];
#pragma unroll
    for (int t = 0; t < KH / 4; ++t) {
      a[4 * t] = nx[t].x; a[4 * t + 1] = nx[t].y; a[4 * t + 2] = nx[t].z; a[4 * t + 3] = nx[t].w;
    }
    if (tile + tstep < ntiles) fetch(tile + tstep);
    if (mean) {
#pragma unroll
      for (int s = 0; s < KH; ++s) a[s] -= smean[h * KH + s];
    }
    if (!ok) {
#pragma unroll
      for (int s = 0; s < KH; ++s) a[s] = 0.0f;
    }
#pragma unroll 1
    for (int jt = 0; jt < W; ++jt) {
      f32x16 acc;
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
      for (int s = 0; s < KH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(smh[(size_t)s * D + jt * 32], a[s], acc, 0, 0, 0);
      // lane: row rn, output columns jt 32 + 8 g + 4 h + (0..3), g = 0..3
      if (ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = jt * 32 + 8 * g + 4 * h;
          const int64_t o = row * ld + c0;
          mf4 r = {alpha * acc[4 * g], alpha * acc[4 * g + 1], alpha * acc[4 * g + 2], alpha * acc[4 * g + 3]};
          if (EXACT || c0 + 4 <= wj) {
            if (base) r += *reinterpret_cast<const mf4u*>(base + o);
            *reinterpret_cast<mf4u*>(out + o) = r;
          } else {
#pragma unroll
            for (int e = 0; e < 3; ++e)
              if (c0 + e < wj) out[o + e] = r[e] + (base ? base[o + e] : 0.0f);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- host side
static bool g_mfma_off() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("MDE_NO_MFMA");
    v = (e && atoi(e)) ? 1 : 0;
  }
  return v != 0;
}
// widths these kernels take (square d x d problems): 5 .. 128 in one register tile, up to 512 in column blocks of 128
bool mde_mfma_width_ok(int d) { return !g_mfma_off() && d >= 5 && d <= 512; }
static int width_class(int w) { return w <= 32 ? 1 : (w <= 64 ? 2 : 4); }
#define MFMA_BLK 128

// mean[c] = column mean of Z (partial: >= 1024 * d doubles)
int mde_mfma_colmean(int64_t n, int d, const float* Z, double* partial, double* mean, hipStream_t st) {
  int64_t nwg = (n + 2047) / 2048;
  if (nwg > 1024) nwg = 1024;
  if (nwg < 1) nwg = 1;
  const int64_t rpw = ((n + nwg - 1) / nwg + 7) & ~(int64_t)7;
  nwg = (n + rpw - 1) / rpw;
  for (int c0 = 0; c0 < d; c0 += MFMA_BLK) {
    const int w = d - c0 < MFMA_BLK ? d - c0 : MFMA_BLK;
    const int wc = width_class(w);
    const bool exact = (d == 32 * wc);  // (one block that fills its register tile)
#define CS(W_)                                                                                                              \
  do {                                                                                                                      \
    if (exact)                                                                                                              \
      hipLaunchKernelGGL((k_colsum_rows<W_, true>), dim3((unsigned)nwg), dim3(MDE_BLOCK), 0, st, n, d, w, Z + c0, rpw, partial);  \
    else                                                                                                                    \
      hipLaunchKernelGGL((k_colsum_rows<W_, false>), dim3((unsigned)nwg), dim3(MDE_BLOCK), 0, st, n, d, w, Z + c0, rpw, partial); \
  } while (0)
    if (wc == 1) CS(1); else if (wc == 2) CS(2); else CS(4);
#undef CS
    MDE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_chunks, dim3((w + 31) / 32), dim3(MDE_BLOCK), 0, st, (int64_t)w, (int)nwg, partial,
                       1.0 / (double)n, mean + c0);
    MDE_LAUNCH_CHECK();
  }
  return MDE_OK;
}

// out[d x d] = (A - mean)^T (B - mean) in double (f32 products and f32 sums over <= 2048 rows per
// chunk, the chunks added in double); partial: max_partial doubles of scratch.  d > 128: blocks of 128 x 128.
int mde_mfma_gram(int64_t n, int d, const float* A, const float* B, const double* mean, double* out, double* partial,
                  int64_t max_partial, hipStream_t st) {
  for (int i0 = 0; i0 < d; i0 += MFMA_BLK) {
    for (int j0 = 0; j0 < d; j0 += MFMA_BLK) {
      const int wa = d - i0 < MFMA_BLK ? d - i0 : MFMA_BLK, wb = d - j0 < MFMA_BLK ? d - j0 : MFMA_BLK;
      const int64_t m = (int64_t)wa * wb;
      int64_t nwg = (n + 2047) / 2048;
      if (nwg < 256 && n >= 256 * 64) nwg = 256;  // one workgroup per CU at least
      if (nwg * m > max_partial) nwg = max_partial / m;
      if (nwg < 1) return MDE_E_UNSUPPORTED;
      const int64_t rpw = ((n + nwg - 1) / nwg + 7) & ~(int64_t)7;
      nwg = (n + rpw - 1) / rpw;
      const bool same = (A == B) && i0 == j0;
      const int wc = width_class(wa > wb ? wa : wb);
      const bool exact = (d == 32 * wc);
      const float *Ab = A + i0, *Bb = B + j0;
      const double *mA = mean ? mean + i0 : nullptr, *mB = mean ? mean + j0 : nullptr;
#define GR(W_)                                                                                                   \
  do {                                                                                                           \
    if (same && exact)                                                                                           \
      hipLaunchKernelGGL((k_gram_rows<W_, true, true>), dim3((unsigned)nwg), dim3(MDE_BLOCK), 0, st, n, d, wa, wb, Ab, Bb, mA, mB, rpw, partial); \
    else if (same)                                                                                               \
      hipLaunchKernelGGL((k_gram_rows<W_, true, false>), dim3((unsigned)nwg), dim3(MDE_BLOCK), 0, st, n, d, wa, wb, Ab, Bb, mA, mB, rpw, partial); \
    else if (exact)                                                                                              \
      hipLaunchKernelGGL((k_gram_rows<W_, false, true>), dim3((unsigned)nwg), dim3(MDE_BLOCK), 0, st, n, d, wa, wb, Ab, Bb, mA, mB, rpw, partial); \
    else                                                                                                         \
      hipLaunchKernelGGL((k_gram_rows<W_, false, false>), dim3((unsigned)nwg), dim3(MDE_BLOCK), 0, st, n, d, wa, wb, Ab, Bb, mA, mB, rpw, partial); \
  } while (0)
      if (wc == 1) GR(1); else if (wc == 2) GR(2); else GR(4);
#undef GR
      MDE_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_sum_chunks_blk, dim3((unsigned)((m + 31) / 32)), dim3(MDE_BLOCK), 0, st, wa, wb, (int)nwg, partial, 1.0,
                         out + (int64_t)i0 * d + j0, d);
      MDE_LAUNCH_CHECK();
    }
  }
  return MDE_OK;
}

// out = base + alpha (A - mean) M  (M: d x d doubles; base may be null; out may alias A or base).  d > 128: column
// blocks of 128 -- out[:, J] = base[:, J] + alpha sum_K (A[:, K] - mean[K]) M[K, J], the K blocks accumulated in place;
// when out aliases A the rows are read from a stream-ordered scratch copy (the blocks of A are needed after their
// columns of out have been written).
int mde_mfma_rmul(int64_t n, int d, const float* A, const double* M, const double* mean, float alpha, const float* base,
                  float* out, hipStream_t st) {
  const float* Asrc = A;
  float* tmp = nullptr;
  if (d > MFMA_BLK && A == out) {
    MDE_HIP(hipMallocAsync(reinterpret_cast<void**>(&tmp), (size_t)n * d * sizeof(float), st));
    MDE_HIP(hipMemcpyAsync(tmp, A, (size_t)n * d * sizeof(float), hipMemcpyDeviceToDevice, st));
    Asrc = tmp;
  }
  static bool raised = false;
  if (!raised) {
    // 64.5 KB of dynamic LDS at the 128-column tile: above the default cap of a launch
    const int lds4 = (int)((128 * 128 + 128) * sizeof(float));
    MDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rmul_rows<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4));
    MDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rmul_rows<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds4));
    raised = true;
  }
  for (int j0 = 0; j0 < d; j0 += MFMA_BLK) {
    for (int k0 = 0; k0 < d; k0 += MFMA_BLK) {
      const int wk = d - k0 < MFMA_BLK ? d - k0 : MFMA_BLK, wj = d - j0 < MFMA_BLK ? d - j0 : MFMA_BLK;
      const int wc = width_class(wk > wj ? wk : wj), D = 32 * wc;
      const bool exact = (d == D);
      const size_t lds = ((size_t)D * D + D) * sizeof(float);
      int64_t nb = ((n + 31) / 32 + 3) / 4;
      const int64_t cap = (lds > 40960) ? 512 : 1024;  // two (three) workgroups per CU fit their copy of M
      if (nb > cap) nb = cap;
      const float* bs = (k0 == 0) ? (base ? base + j0 : nullptr) : out + j0;
      const float* Ab = Asrc + k0;
      const double* Mb = M + (int64_t)k0 * d + j0;
      const double* mb = mean ? mean + k0 : nullptr;
      float* ob = out + j0;
#define RM(W_)                                                                                                                     \
  do {                                                                                                                             \
    if (exact)                                                                                                                     \
      hipLaunchKernelGGL((k_rmul_rows<W_, true>), dim3((unsigned)nb), dim3(MDE_BLOCK), lds, st, n, d, wk, wj, Ab, Mb, d, mb, alpha, bs, ob);  \
    else                                                                                                                           \
      hipLaunchKernelGGL((k_rmul_rows<W_, false>), dim3((unsigned)nb), dim3(MDE_BLOCK), lds, st, n, d, wk, wj, Ab, Mb, d, mb, alpha, bs, ob); \
  } while (0)
      if (wc == 1) RM(1); else if (wc == 2) RM(2); else RM(4);
#undef RM
      MDE_LAUNCH_CHECK();
    }
  }
  if (tmp) MDE_HIP(hipFreeAsync(tmp, st));
  return MDE_OK;
}
