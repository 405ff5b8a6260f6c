// mde_distortion.hip -- the hot kernel: fused forward + backward of the average distortion
//   E(X) = (1/p) sum_k f_k(||x_i - x_j||),   dE/dX
// [ref: pymde/average_distortion.py:62-106 `_AverageDistortion.forward/backward`], plus the
// edge-order evaluators behind MDE.differences/distances/distortions
// [ref: pymde/problem.py:246-308, average_distortion.py:38-55] and the unfused fallback used
// for arbitrary Python callables.
//
// Execution model (CDNA4, wave64): the plan is a symmetrised incidence CSR.  A group of G
// lanes owns one vertex row v at a time; the lanes stride over the row's half-edges
// (coalesced int32 neighbour + fp32 parameter streams), gather x_u (the only random access,
// served by L2 / Infinity Cache: the table is n*d*4 bytes), evaluate f and f'/d in registers,
// accumulate g (x_v - x_u) and reduce across the G lanes with xor butterflies (DPP).  There
// are no atomics and no [p, d] temporaries; the reference materialises five of them.
// Each edge is visited from both endpoints, so its loss term is counted with weight 1/2.
#include "mde_common.h"
#include "mde_functions.h"
#include "mde_plan.h"
#define COMMA ,

double* mde_plan_partials(mde_plan* p);
float mde_plan_avg_degree(const mde_plan* p);
int mde_ring_try(mde_plan* plan, const float* X, int d, const mde_func* f, float grad_scale,
                  float* grad, float inv_p, hipStream_t st, int* nblocks, float* loss_out,
                  double loss_scale);

// ---------------------------------------------------------------- small-d fused kernel
// D in {1,2,3,4}: one lane per half-edge, G lanes per row.
template <int D>
struct VecD {
  float v[D];
};
template <int D>
MDE_DEV VecD<D> load_row(const float* __restrict__ X, int64_t r) {
  VecD<D> o;
  if constexpr (D == 2) {
    const float2 t = reinterpret_cast<const float2*>(X)[r];
    o.v[0] = t.x;
    o.v[1] = t.y;
  } else if constexpr (D == 4) {
    const float4 t = reinterpret_cast<const float4*>(X)[r];
    o.v[0] = t.x;
    o.v[1] = t.y;
    o.v[2] = t.z;
    o.v[3] = t.w;
  } else {
#pragma unroll
    for (int c = 0; c < D; ++c) o.v[c] = X[r * D + c];
  }
  return o;
}

template <int D, int G, bool INDIRECT, class Fn>
__global__ __launch_bounds__(MDE_BLOCK) void k_fused_small(
    int nrows, int row_lo, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ eid, const float* __restrict__ a0, const float* __restrict__ a1,
    int a0_scalar, int a1_scalar, const float* __restrict__ X, float* __restrict__ grad,
    double* __restrict__ loss_partials, Fn fn, float inv_p, float grad_scale) {
  __shared__ double smem[8];
  const int lig = threadIdx.x & (G - 1);
  const int group = (blockIdx.x * MDE_BLOCK + threadIdx.x) / G;
  const int ngroups = (gridDim.x * MDE_BLOCK) / G;
  float loss = 0.0f;
  const float a0s = a0_scalar ? a0[0] : 0.0f;
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;

  for (int r = group; r < nrows; r += ngroups) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    const int64_t v = (int64_t)row_lo + r;
    const VecD<D> xv = load_row<D>(X, v);
    float acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] = 0.0f;
#pragma unroll 2
    for (int h = beg + lig; h < end; h += G) {
      const int u = nbr[h];
      float p0, p1 = a1s;
      if constexpr (INDIRECT) {
        const int k = eid[h];
        p0 = a0[k];
        p1 = a1[k];
      } else {
        p0 = a0_scalar ? a0s : a0[h];
        if (a1 && !a1_scalar) p1 = a1[h];
      }
      const VecD<D> xu = load_row<D>(X, u);
      float diff[D], ss = 0.0f;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        diff[c] = xv.v[c] - xu.v[c];
        ss = fmaf(diff[c], diff[c], ss);
      }
      float f, gd;
      fn.eval(ss, p0, p1, f, gd);
      const float g = mde_fix_g(gd * inv_p);
      loss += f;
#pragma unroll
      for (int c = 0; c < D; ++c) acc[c] = fmaf(g, diff[c], acc[c]);
    }
    if (grad) {
#pragma unroll
      for (int c = 0; c < D; ++c) acc[c] = mde_group_sum<G>(acc[c]);
      if (lig == 0) {
        if constexpr (D == 2) {
          reinterpret_cast<float2*>(grad)[v] = make_float2(acc[0] * grad_scale, acc[1] * grad_scale);
        } else if constexpr (D == 4) {
          reinterpret_cast<float4*>(grad)[v] = make_float4(acc[0] * grad_scale, acc[1] * grad_scale,
                                                           acc[2] * grad_scale, acc[3] * grad_scale);
        } else {
#pragma unroll
          for (int c = 0; c < D; ++c) grad[v * D + c] = acc[c] * grad_scale;
        }
      }
    }
  }
  const double bs = mde_block_sum((double)loss, smem);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = bs;
}

// ---------------------------------------------------------------- small-d, edge-balanced
// The row-per-group kernel above waits on three dependent memory hops per row (row pointer ->
// neighbour -> x_u) and its waves run as long as their longest row: on graphs with hubs (k-NN
// graphs, scale-free graphs) it is latency-bound (70 us for 2.9M half-edges).  Here a wave takes a
// TILE of MDE_FLAT_T consecutive half-edge positions instead -- every load of the tile is issued
// before the first use, whatever the rows look like -- and sums the contributions of equal rows
// with a segmented scan across the lanes (rows are contiguous runs of positions).  A run that lies
// inside the tile is a finished gradient row.  A row that crosses tile boundaries leaves one
// record per tile (the tile's first run if the row began earlier, its last run if the row goes
// on); k_flat_fixup adds the records of such a row in tile order.  Tiles are aligned to global
// half-edge positions (phase = positions of the rows below row_lo, mod the tile), so a row is
// summed by the same lanes in the same order whichever rank owns it: shards stay bit-equal to
// the single-process result.  No atomics.
template <int D, bool INDIRECT, class Fn>
__global__ __launch_bounds__(MDE_BLOCK) void k_fused_flat(
    int64_t H, int phase, int64_t n_tiles, int row_lo, const int32_t* __restrict__ hrow,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ eid, const float* __restrict__ a0,
    const float* __restrict__ a1, int a0_scalar, int a1_scalar, const float* __restrict__ X,
    float* __restrict__ grad, float* __restrict__ rec, double* __restrict__ loss_partials, Fn fn,
    float inv_p, float grad_scale) {
  __shared__ double smem[8];
  constexpr int U = MDE_FLAT_U;
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = ((int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * MDE_BLOCK) >> 6;
  const float a0s = a0_scalar ? a0[0] : 0.0f;
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;
  float loss = 0.0f;
  for (int64_t t = wave0; t < n_tiles; t += nwaves) {
    const int64_t base = t * MDE_FLAT_T - phase;  // local position of the tile's first slot
    int row[U], un[U];
    float p0[U], p1[U];
    bool ok[U];
    // ---- every index / parameter load of the tile
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t h = base + u * 64 + lane;
      ok[u] = h >= 0 && h < H;
      const int64_t hc = ok[u] ? h : 0;
      const int rr = hrow[hc];  // (hc is in range either way: a plain load, the select afterwards)
      row[u] = ok[u] ? rr : -1;
      un[u] = nbr[hc];
      p1[u] = a1s;
      if constexpr (INDIRECT) {
        const int k = eid[hc];
        p0[u] = a0[k];
        p1[u] = a1[k];
      } else {
        p0[u] = a0_scalar ? a0s : a0[hc];
        if (a1 && !a1_scalar) p1[u] = a1[hc];
      }
    }
    // rows of the positions just before and just after the tile: a first run with the former
    // began earlier, a last run with the latter goes on
    const int prev_row = base > 0 ? hrow[base - 1] : -1;
    const int next_row = base + MDE_FLAT_T < H ? hrow[base + MDE_FLAT_T] : -1;
    // ---- every gather of the tile
    VecD<D> xv[U], xu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      xv[u] = load_row<D>(X, (int64_t)row_lo + (ok[u] ? row[u] : 0));
      xu[u] = load_row<D>(X, un[u]);
    }
    float c[U][D];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float ss = 0.0f;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        c[u][k] = xv[u].v[k] - xu[u].v[k];
        ss = fmaf(c[u][k], c[u][k], ss);
      }
      float f, gd;
      fn.eval(ss, p0[u], p1[u], f, gd);
      const float g = ok[u] ? mde_fix_g(gd * inv_p) : 0.0f;
      loss += ok[u] ? f : 0.0f;
#pragma unroll
      for (int k = 0; k < D; ++k) c[u][k] *= g;
    }
    if (!grad) continue;
    // ---- rows: segmented inclusive scan per wave iteration, carry from one iteration to the next
    int carry_row = -1, first_row = -1, first_ends = 0;
    float carry[D], first_tot[D];
#pragma unroll
    for (int k = 0; k < D; ++k) carry[k] = first_tot[k] = 0.0f;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = row[u];
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int k2 = __shfl_up(key, off, 64);
        const bool same = lane >= off && k2 == key;
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const float v2 = __shfl_up(c[u][k], off, 64);
          c[u][k] += same ? v2 : 0.0f;
        }
      }
      float tot[D];
#pragma unroll
      for (int k = 0; k < D; ++k) tot[k] = c[u][k] + (key == carry_row ? carry[k] : 0.0f);
      // row of the next position: the lane above, or lane 0 of the next iteration / tile
      int nxt = __shfl_down(key, 1, 64);
      const int over = (u + 1 < U) ? __builtin_amdgcn_readlane(row[(u + 1 < U) ? u + 1 : u], 0) : next_row;
      if (lane == 63) nxt = over;
      const bool row_ends = key >= 0 && nxt != key;  // the row's last position
      if (row_ends) {
        if (key != prev_row) {
          const int64_t v = (int64_t)row_lo + key;
          if constexpr (D == 2) {
            reinterpret_cast<float2*>(grad)[v] = make_float2(tot[0] * grad_scale, tot[1] * grad_scale);
          } else if constexpr (D == 4) {
            reinterpret_cast<float4*>(grad)[v] = make_float4(tot[0] * grad_scale, tot[1] * grad_scale,
                                                             tot[2] * grad_scale, tot[3] * grad_scale);
          } else {
#pragma unroll
            for (int k = 0; k < D; ++k) grad[v * D + k] = tot[k] * grad_scale;
          }
        }
      }
      // the run that began before the tile ends here: its record (at most one lane of the tile)
      const unsigned long long m = __ballot(row_ends && key == prev_row);
      if (m) {
        const int src = __builtin_ctzll(m);
        first_row = __builtin_amdgcn_readlane(key, src);
        first_ends = 1;
#pragma unroll
        for (int k = 0; k < D; ++k) first_tot[k] = __shfl(tot[k], src, 64);
      }
      // lane 63's row goes on: carry its running sum into the next iteration (or out of the tile)
      const int k63 = __builtin_amdgcn_readlane(key, 63);
      carry_row = (k63 >= 0 && over == k63) ? k63 : -1;
#pragma unroll
      for (int k = 0; k < D; ++k) carry[k] = __shfl(tot[k], 63, 64);
    }
    if (lane == 0) {
      // records: [first | last] x (row, ends, sum[4], -, -)
      float* r0 = rec + (size_t)t * 16;
      int last_row = -1;
      if (carry_row >= 0) {
        if (carry_row == prev_row) {  // the whole tile is one row that began earlier and goes on
          first_row = carry_row;
          first_ends = 0;
#pragma unroll
          for (int k = 0; k < D; ++k) first_tot[k] = carry[k];
        } else {
          last_row = carry_row;
        }
      }
      reinterpret_cast<int*>(r0)[0] = first_row;
      reinterpret_cast<int*>(r0)[1] = first_ends;
      reinterpret_cast<int*>(r0)[8] = last_row;
#pragma unroll
      for (int k = 0; k < D; ++k) {
        r0[2 + k] = first_tot[k];
        r0[10 + k] = carry[k];
      }
    }
  }
  const double bs = mde_block_sum((double)loss, smem);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = bs;
}

// rows that cross tile boundaries: the tile in which such a row ends adds its records in tile order
// (its extra last workgroup adds the loss partials of the fused kernel: no third launch)
template <int D>
__global__ __launch_bounds__(MDE_BLOCK) void k_flat_fixup(int64_t n_tiles, int phase, int row_lo,
                                                          const int32_t* __restrict__ rowptr,
                                                          const float* __restrict__ rec,
                                                          float* __restrict__ grad, float grad_scale,
                                                          double* __restrict__ loss_partials, int nparts,
                                                          double loss_scale, float* __restrict__ loss_out) {
  if (loss_out && blockIdx.x == gridDim.x - 1) {
    __shared__ double smem[8];
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += MDE_BLOCK) s += loss_partials[i];
    const double tot = mde_block_sum(s, smem);
    if (threadIdx.x == 0) {
      *loss_out = (float)(tot * loss_scale);
      loss_partials[MDE_PARTIALS_LOSS_D] = tot * loss_scale;
    }
    return;
  }
  const int64_t t = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (t >= n_tiles) return;
  const float* rt = rec + (size_t)t * 16;
  const int r = reinterpret_cast<const int*>(rt)[0];
  if (r < 0 || reinterpret_cast<const int*>(rt)[1] == 0) return;
  const int64_t ta = ((int64_t)rowptr[r] + phase) / MDE_FLAT_T;  // the tile the row begins in
  float s[D];
#pragma unroll
  for (int k = 0; k < D; ++k) s[k] = rec[(size_t)ta * 16 + 10 + k];
  for (int64_t q = ta + 1; q <= t; ++q) {
#pragma unroll
    for (int k = 0; k < D; ++k) s[k] += rec[(size_t)q * 16 + 2 + k];
  }
  const int64_t v = (int64_t)row_lo + r;
#pragma unroll
  for (int k = 0; k < D; ++k) grad[v * D + k] = s[k] * grad_scale;
}

// ---------------------------------------------------------------- general-d fused kernel
// One wave owns a row.  GL lanes cooperate on one half-edge (64/GL half-edges in flight per
// wave); lane `lig` holds components c = lig + j*GL, j < K (coalesced 4-byte lanes; rows of
// d floats are contiguous so a half-edge gather is one or more full 128/256-byte segments).
template <int GL, int K, bool INDIRECT, class Fn>
__global__ __launch_bounds__(MDE_BLOCK) void k_fused_wide(
    int nrows, int row_lo, int d, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ nbr, const int32_t* __restrict__ eid, const float* __restrict__ a0,
    const float* __restrict__ a1, int a0_scalar, int a1_scalar, const float* __restrict__ X,
    float* __restrict__ grad, double* __restrict__ loss_partials, Fn fn, float inv_p,
    float grad_scale) {
  __shared__ double smem[8];
  constexpr int E = 64 / GL;                       // half-edges per wave step
  constexpr int U = (K <= 2) ? 4 : (K <= 4 ? 2 : 1);  // independent steps in flight (row gathers
                                                      // are long-latency HBM reads at large d)
  const int lane = threadIdx.x & 63;
  const int lig = lane & (GL - 1);
  const int sub = lane / GL;  // which of the E half-edges this lane works on
  const int wave = (blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * MDE_BLOCK) >> 6;
  float loss = 0.0f;
  const float a0s = a0_scalar ? a0[0] : 0.0f;
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;

  for (int r = wave; r < nrows; r += nwaves) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    const int64_t v = (int64_t)row_lo + r;
    float xv[K], acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int c = lig + j * GL;
      xv[j] = (c < d) ? X[v * d + c] : 0.0f;
      acc[j] = 0.0f;
    }
    for (int h0 = beg; h0 < end; h0 += E * U) {
      bool live[U];
      int64_t u[U];
      float p0[U], p1[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int h = h0 + q * E + sub;
        live[q] = h < end;
        const int hh = live[q] ? h : beg;
        u[q] = nbr[hh];
        p1[q] = a1s;
        if constexpr (INDIRECT) {
          const int k = eid[hh];
          p0[q] = a0[k];
          p1[q] = a1[k];
        } else {
          p0[q] = a0_scalar ? a0s : a0[hh];
          if (a1 && !a1_scalar) p1[q] = a1[hh];
        }
      }
      float xu[U][K];
#pragma unroll
      for (int q = 0; q < U; ++q)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const int c = lig + j * GL;
          xu[q][j] = (c < d) ? X[u[q] * d + c] : 0.0f;
        }
#pragma unroll
      for (int q = 0; q < U; ++q) {
        float diff[K], ss = 0.0f;
#pragma unroll
        for (int j = 0; j < K; ++j) {
          diff[j] = xv[j] - xu[q][j];
          ss = fmaf(diff[j], diff[j], ss);
        }
        ss = mde_group_sum<GL>(ss);
        float f, gd;
        fn.eval(ss, p0[q], p1[q], f, gd);
        float g = mde_fix_g(gd * inv_p);
        if (!live[q]) {
          g = 0.0f;
          f = 0.0f;
        }
        if (lig == 0) loss += f;
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = fmaf(g, diff[j], acc[j]);
      }
    }
    if (grad) {
      // combine the E sub-groups (lanes with equal lig): xor over the sub index bits
#pragma unroll
      for (int j = 0; j < K; ++j) {
        float a = acc[j];
#pragma unroll
        for (int o = 32; o >= GL; o >>= 1) a += __shfl_xor(a, o, 64);
        acc[j] = a;
      }
      if (sub == 0) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const int c = lig + j * GL;
          if (c < d) grad[v * d + c] = acc[j] * grad_scale;
        }
      }
    }
  }
  const double bs = mde_block_sum((double)loss, smem);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = bs;
}


// ---------------------------------------------------------------- general-d fused kernel, 16-byte form
// d a multiple of 4 and 16-byte aligned rows: GL lanes cooperate on one half-edge and every lane
// moves K4 float4s of the row, so a row gather is 16 bytes per lane per instruction (a d = 128 row
// is ONE load instruction of a half wave).  The 4-byte form above issues four times as many load
// instructions and tops out at ~3 TB/s of row gathers at BASELINE config 5 (n = 500k, d = 128);
// tools/rowprobe measures the chip's ceiling for random 512-byte rows from a 256 MB table at
// 7.4 TB/s with this access shape.
typedef float wide_f4 __attribute__((ext_vector_type(4)));
template <int GL, int K4, bool INDIRECT, class Fn>
__global__ __launch_bounds__(MDE_BLOCK) void k_fused_wide4(
    int nrows, int row_lo, int d4, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ eid, const float* __restrict__ a0, const float* __restrict__ a1, int a0_scalar,
    int a1_scalar, const wide_f4* __restrict__ X4, wide_f4* __restrict__ grad4, double* __restrict__ loss_partials,
    Fn fn, float inv_p, float grad_scale) {
  __shared__ double smem[8];
  constexpr int E = 64 / GL;                 // half-edges per wave step
  constexpr int U = (K4 == 1) ? 4 : (K4 == 2 ? 2 : 1);  // independent steps in flight
  const int lane = threadIdx.x & 63;
  const int lig = lane & (GL - 1);
  const int sub = lane / GL;
  const int wave = (blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * MDE_BLOCK) >> 6;
  float loss = 0.0f;
  const float a0s = a0_scalar ? a0[0] : 0.0f;
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;
  const wide_f4 zero = {0.f, 0.f, 0.f, 0.f};
  const bool full = d4 == GL * K4;

  for (int r = wave; r < nrows; r += nwaves) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    const int64_t v = (int64_t)row_lo + r;
    wide_f4 xv[K4], acc[K4];
#pragma unroll
    for (int j = 0; j < K4; ++j) {
      const int c = lig + j * GL;
      xv[j] = (c < d4) ? X4[v * d4 + c] : zero;
      acc[j] = zero;
    }
    // neighbour ids and parameters are fetched one step ahead: their latency overlaps the row
    // gathers of the current step instead of preceding them
    bool live_n[U];
    int64_t u_n[U];
    float p0_n[U], p1_n[U];
    auto load_meta = [&](int h0) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < U; ++q) {
        const int h = h0 + q * E + sub;
        live_n[q] = h < end;
        const int hh = live_n[q] ? h : beg;
        u_n[q] = nbr[hh];
        p1_n[q] = a1s;
        if constexpr (INDIRECT) {
          const int k = eid[hh];
          p0_n[q] = a0[k];
          p1_n[q] = a1[k];
        } else {
          p0_n[q] = a0_scalar ? a0s : a0[hh];
          if (a1 && !a1_scalar) p1_n[q] = a1[hh];
        }
      }
    };
    if (beg < end) load_meta(beg);
    for (int h0 = beg; h0 < end; h0 += E * U) {
      bool live[U];
      int64_t u[U];
      float p0[U], p1[U];
#pragma unroll
      for (int q = 0; q < U; ++q) {
        live[q] = live_n[q];
        u[q] = u_n[q];
        p0[q] = p0_n[q];
        p1[q] = p1_n[q];
      }
      wide_f4 xu[U][K4];
      if (full) {
        // (every lane has a column: plain loads -- a predicated load is a branch around it, and the
        // gathers of a step then go out one after the other)
#pragma unroll
        for (int q = 0; q < U; ++q)
#pragma unroll
          for (int j = 0; j < K4; ++j) xu[q][j] = X4[u[q] * d4 + lig + j * GL];
      } else {
#pragma unroll
        for (int q = 0; q < U; ++q)
#pragma unroll
          for (int j = 0; j < K4; ++j) {
            const int c = lig + j * GL;
            xu[q][j] = (c < d4) ? X4[u[q] * d4 + c] : zero;
          }
      }
      if (h0 + E * U < end) load_meta(h0 + E * U);
#pragma unroll
      for (int q = 0; q < U; ++q) {
        wide_f4 diff[K4];
        float ss = 0.0f;
#pragma unroll
        for (int j = 0; j < K4; ++j) {
          diff[j] = xv[j] - xu[q][j];
          ss = fmaf(diff[j].x, diff[j].x, ss);
          ss = fmaf(diff[j].y, diff[j].y, ss);
          ss = fmaf(diff[j].z, diff[j].z, ss);
          ss = fmaf(diff[j].w, diff[j].w, ss);
        }
        ss = mde_group_sum<GL>(ss);
        float f, gd;
        fn.eval(ss, p0[q], p1[q], f, gd);
        float g = mde_fix_g(gd * inv_p);
        if (!live[q]) {
          g = 0.0f;
          f = 0.0f;
        }
        if (lig == 0) loss += f;
#pragma unroll
        for (int j = 0; j < K4; ++j) acc[j] += g * diff[j];
      }
    }
    if (grad4) {
      // combine the E sub-groups (lanes with equal lig): xor over the sub index bits
#pragma unroll
      for (int j = 0; j < K4; ++j) {
#pragma unroll
        for (int o = 32; o >= GL; o >>= 1) {
          acc[j].x += __shfl_xor(acc[j].x, o, 64);
          acc[j].y += __shfl_xor(acc[j].y, o, 64);
          acc[j].z += __shfl_xor(acc[j].z, o, 64);
          acc[j].w += __shfl_xor(acc[j].w, o, 64);
        }
      }
      if (sub == 0) {
#pragma unroll
        for (int j = 0; j < K4; ++j) {
          const int c = lig + j * GL;
          if (c < d4) grad4[v * d4 + c] = acc[j] * grad_scale;
        }
      }
    }
  }
  const double bs = mde_block_sum((double)loss, smem);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = bs;
}

// ---------------------------------------------------------------- general-d fused kernel, pipelined form (round 6)
// Rows of ncol = ceil(d / 4) float4 columns, GL lanes x K4 float4s per half-edge with GL * (K4 - 1) < ncol <= GL * K4 (GL = 1 .. 32,
// K4 = 2 .. 4: every d = 5 .. 512; d = 128 is <8, 4>).  Same lane layout as k_fused_wide4<GL, K4>
// (GL lanes share a half-edge, 64 / GL half-edges per wave step); what changes is everything around the arithmetic:
//  * the row gathers of step s + 1 are in flight while step s is evaluated (two register buffers, the loop unrolled
//    by two; the meta words -- neighbour id, parameters -- run two steps ahead, the next row's first meta words and
//    row pointers a whole row ahead): a wave never sits behind its own loads with nothing issued;
//  * |x_v - x_u|^2 with packed fp32 math (v_pk_add_f32 / v_pk_fma_f32: 16 instructions instead of 32), the sum over
//    the lanes of a group with DPP adds instead of LDS permutes;
//  * the E partial gradient rows of a wave are added by a TRANSPOSING reduction (v_permlane32_swap / v_permlane16_swap:
//    one swap + one add folds two registers into one whose halves hold the two sums): 12 swaps + 16 adds for the 16
//    floats of a lane instead of 48 LDS permutes + 48 adds, and every lane ends up owning one float4 of the row;
//  * XCD-aware row order: the rows are cut into chunks of 2^chunk_log rows, chunk c belongs to XCD c % 8 (block b runs
//    on XCD b % 8), and the waves of an XCD walk ITS chunks in order -- the rows an XCD works on at any moment are a
//    band of ~1000 consecutive rows, so on a graph with locality (neighbours within a window of the vertex order)
//    the band's neighbours stay in that XCD's 4 MB L2 instead of being fetched by all eight.
// Summation order inside a row: fixed by the row's own half-edge positions (steps of E from the row's first entry),
// so a vertex-range shard produces the bits of the single process.
typedef float wide_f2 __attribute__((ext_vector_type(2)));
typedef unsigned wide_u2 __attribute__((ext_vector_type(2)));
typedef wide_f4 wide_f4u __attribute__((aligned(4)));  // a float4 at any float of a row (unaligned access mode)
template <int GL>
__device__ __forceinline__ float wide_group_sum(float v) {
  static_assert(GL == 1 || GL == 2 || GL == 4 || GL == 8 || GL == 16 || GL == 32, "group widths of the pipelined kernel");
  if constexpr (GL >= 2) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  if constexpr (GL >= 4) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  if constexpr (GL >= 8) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  if constexpr (GL >= 16) v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  if constexpr (GL >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}
// a <- the xor-32 sums of a in lanes 0..31 and of b in lanes 32..63
__device__ __forceinline__ float wide_fold32(float a, float b) {
  const wide_u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
// a <- the xor-16 sums of a in rows 0, 2 and of b in rows 1, 3 (rows of 16 lanes)
__device__ __forceinline__ float wide_fold16(float a, float b) {
  const wide_u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}
template <int GL, int K4, bool INDIRECT, class Fn>
__global__ __launch_bounds__(MDE_BLOCK) void k_fused_wide4p(
    int nrows, int row_lo, int d, int chunk_log, const int32_t* __restrict__ order, const int32_t* __restrict__ rowptr,
    const int32_t* __restrict__ nbr,
    const int32_t* __restrict__ eid, const float* __restrict__ a0, const float* __restrict__ a1, int a0_scalar,
    int a1_scalar, const float* __restrict__ X, float* __restrict__ grad, double* __restrict__ loss_partials,
    Fn fn, float inv_p, float grad_scale) {
  __shared__ double smem[8];
  constexpr int E = 64 / GL;   // half-edges per wave step
  constexpr int WPB = MDE_BLOCK / 64;
  const int lane = threadIdx.x & 63;
  const int lig = lane & (GL - 1);
  const int sub = lane / GL;
  const int xcd = blockIdx.x & 7;
  const int wpx = (gridDim.x >> 3) * WPB;  // waves per XCD (the grid is a multiple of 8 blocks)
  const int cmask = (1 << chunk_log) - 1;
  float loss = 0.0f;
  const float a0s = a0_scalar ? a0[0] : 0.0f;
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;
  // Columns: float4 number c covers the floats [4 c, 4 c + 4) of a row, except the row's LAST column when d is not a
  // multiple of 4: that one covers [d - 4, d) -- shifted back so that it stays inside the row -- and only its last
  // d % 4 components count.  Rows start at multiples of d floats, so the 16-byte accesses are 4-byte aligned only
  // (the hardware's unaligned access mode; wide_f4u below).  A lane's columns are lig + j GL, j < K4, with
  // GL (K4 - 1) < ncol <= GL K4: only its last column can be the row's last, or lie behind it (then it loads the
  // row's last column again and every component is masked).
  const int ncol = (d + 3) >> 2, rem = d & 3;
  const int c_mine = lig + (K4 - 1) * GL;
  const bool tail_dead = c_mine >= ncol;
  const bool tail_part = rem != 0 && c_mine == ncol - 1;
  const int off_tail = (tail_dead || tail_part) ? (rem ? d - 4 : 4 * (ncol - 1)) : 4 * c_mine;
  const int first_kept = tail_dead ? 4 : (tail_part ? 4 - rem : 0);  // components below it are masked
  const wide_f2 keep_lo = {first_kept <= 0 ? 1.0f : 0.0f, first_kept <= 1 ? 1.0f : 0.0f};
  const wide_f2 keep_hi = {first_kept <= 2 ? 1.0f : 0.0f, first_kept <= 3 ? 1.0f : 0.0f};

  // position of the i-th row of this XCD's sequence in the processing order, or -1 behind the end; the row at a
  // position is order[position] when the plan carries a processing order (mde_plan_row_order), the position itself
  // otherwise
  auto pos_of = [&](int i) __attribute__((always_inline)) -> int {
    const int q = ((((i >> chunk_log) << 3) + xcd) << chunk_log) + (i & cmask);
    return q < nrows ? q : -1;
  };
  auto row_at = [&](int q) __attribute__((always_inline)) -> int {
    return (q >= 0 && order) ? order[q] : q;
  };
  auto row_of = [&](int i) __attribute__((always_inline)) -> int { return row_at(pos_of(i)); };
  struct Meta {
    int u;
    float p0, p1;
  };
  auto load_meta = [&](int h, int end) __attribute__((always_inline)) -> Meta {
    Meta m;
    int hh = h < end ? h : end - 1;
    hh = hh < 0 ? 0 : hh;
    m.u = nbr[hh];
    m.p1 = a1s;
    if constexpr (INDIRECT) {
      const int k = eid[hh];
      m.p0 = a0[k];
      m.p1 = a1[k];
    } else {
      m.p0 = a0_scalar ? a0s : a0[hh];
      if (a1 && !a1_scalar) m.p1 = a1[hh];
    }
    return m;
  };
  auto load_rows = [&](wide_f4 (&x)[K4], int64_t u) __attribute__((always_inline)) {
    const float* p = X + u * d;
#pragma unroll
    for (int j = 0; j < K4 - 1; ++j) x[j] = *reinterpret_cast<const wide_f4u*>(p + 4 * (lig + j * GL));
    x[K4 - 1] = *reinterpret_cast<const wide_f4u*>(p + off_tail);
  };
  // the lane's column j of gradient row v (j >= K4, or a column behind the row's end: nothing; the row's last column
  // when d is not a multiple of 4: its last d % 4 components, one by one -- the float4 would overlap the column before)
  auto store_col = [&](int64_t v, int j, wide_f4 val) __attribute__((always_inline)) {
    float* g = grad + v * d;
    if (j < K4 - 1) {
      *reinterpret_cast<wide_f4u*>(g + 4 * (lig + j * GL)) = val;
    } else if (j == K4 - 1 && !tail_dead) {
      if (!tail_part) {
        *reinterpret_cast<wide_f4u*>(g + off_tail) = val;
      } else {
        if (rem >= 3) g[d - 3] = val.y;
        if (rem >= 2) g[d - 2] = val.z;
        g[d - 1] = val.w;
      }
    }
  };

  int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x >> 3) * WPB + (int)(threadIdx.x >> 6));
  int r = row_of(i);
  int beg = 0, end = 0, r_n = -1, beg_n = 0, end_n = 0, r_nn = -1, beg_nn = 0, end_nn = 0, r_n3 = -1;
  Meta mA = {0, 0.f, 0.f}, mB = mA, m0n = mA, m1n = mA;
  wide_f4 xv[K4], bufA[K4], bufB[K4];
  if (r >= 0) {
    beg = rowptr[r];
    end = rowptr[r + 1];
    mA = load_meta(beg + sub, end);
    mB = load_meta(beg + E + sub, end);
    r_n = row_of(i + wpx);
    if (r_n >= 0) {
      beg_n = rowptr[r_n];
      end_n = rowptr[r_n + 1];
      m0n = load_meta(beg_n + sub, end_n);
      m1n = load_meta(beg_n + E + sub, end_n);
      r_nn = row_of(i + 2 * wpx);
      if (r_nn >= 0) {
        beg_nn = rowptr[r_nn];
        end_nn = rowptr[r_nn + 1];
        r_n3 = row_of(i + 3 * wpx);
      }
    }
    load_rows(xv, (int64_t)row_lo + r);
    load_rows(bufA, mA.u);
#pragma unroll
    for (int j = 0; j < K4; ++j) {
      xv[j] = -xv[j];
      asm volatile("" : "+v"(xv[j]));
    }
  }
  // The wave's rows form ONE software pipeline: at the top of a row bufA already holds (or is about to receive) the
  // gathers of the row's first step -- issued by the previous row's last step --, its first two meta words are in
  // registers, and so are the row pointers and the first two meta words of the row after it and the row pointers of
  // the row after that.
  while (r >= 0) {
    const int64_t v = (int64_t)row_lo + r;
    // (an empty row runs one step with every lane dead -- clamped positions, nothing added: no special case for the
    // compiler's load bookkeeping to be conservative about)
    const int nsteps = end > beg ? (end - beg + E - 1) / E : 1;
    // three rows ahead: the row pointers; four rows ahead: which row that is (scalar loads; they are back by the
    // time the row's end rotates them in)
    i += wpx;
    int beg_n3 = 0, end_n3 = 0;
    if (r_n3 >= 0) {
      beg_n3 = rowptr[r_n3];
      end_n3 = rowptr[r_n3 + 1];
    }
    const int r_n4 = r_n3 >= 0 ? row_of(i + 3 * wpx) : -1;
    // the next row's x_v, a row ahead like everything else of it; the first two meta words of the row after next
    // (loaded here, not at the row's end: the copy into the registers the next trip reads would otherwise wait for
    // loads that have only just gone out; behind the last row beg_nn = end_nn = 0: position 0)
    wide_f4 xvn[K4];
    load_rows(xvn, (int64_t)row_lo + (r_n >= 0 ? r_n : r));
    const Meta m0nn = load_meta(beg_nn + sub, end_nn);
    const Meta m1nn = load_meta(beg_nn + E + sub, end_nn);
    wide_f2 acc[2 * K4];
#pragma unroll
    for (int j = 0; j < 2 * K4; ++j) acc[j] = wide_f2{0.f, 0.f};

    // one step: x holds the gathered rows of E half-edges (one per group of GL lanes).
    // (xv holds -x_v: dd = x_u - x_v is then a packed ADD -- hipcc has no packed form for a subtraction of two
    // register pairs --, the accumulators collect -g (x_v - x_u) and the sign goes into the scale of the final
    // store: negation is exact, every intermediate is the negative of the plain form's, bit for bit)
    auto step = [&](wide_f4 (&x)[K4], const Meta& m, bool live) __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      wide_f2 dd[2 * K4], s2 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
      for (int j = 0; j < K4; ++j) {
        dd[2 * j] = wide_f2{xv[j].x, xv[j].y} + wide_f2{x[j].x, x[j].y};
        dd[2 * j + 1] = wide_f2{xv[j].z, xv[j].w} + wide_f2{x[j].z, x[j].w};
        if (j == K4 - 1) {
          // (x 1 where the row is exactly GL x K4 float4s wide: exact)
          dd[2 * j] *= keep_lo;
          dd[2 * j + 1] *= keep_hi;
        }
        s2 = dd[2 * j] * dd[2 * j] + s2;
        t2 = dd[2 * j + 1] * dd[2 * j + 1] + t2;
      }
      s2 += t2;
      const float ss = wide_group_sum<GL>(s2.x + s2.y);
      float f, gd;
      fn.eval(ss, m.p0, m.p1, f, gd);
      float g = mde_fix_g(gd * inv_p);
      if (!live) {
        g = 0.0f;
        f = 0.0f;
      }
      if (lig == 0) loss += f;
      const wide_f2 g2 = {g, g};
#pragma unroll
      for (int j = 0; j < 2 * K4; ++j) acc[j] = g2 * dd[j] + acc[j];
      __builtin_amdgcn_sched_barrier(0);
    };

    int h = beg;
    int k = 0;
    do {
      // (every load below is issued on every trip -- clamped positions behind the row's end --: a branch around a
      // gather makes the compiler's vmcnt bookkeeping wait for EVERYTHING at the next use.  Meta words go out before
      // the gathers that follow them: vmcnt counts in order -- both of a trip's at its top, so that the rotation at
      // its bottom finds them behind gathers that step B has already waited for.)
      const Meta mC = load_meta(h + 2 * E + sub, end);
      const Meta mD = load_meta(h + 3 * E + sub, end);
      load_rows(bufB, mB.u);
      step(bufA, mA, h + sub < end);
      const bool last = k + 2 >= nsteps;
      load_rows(bufA, last ? m0n.u : mC.u);   // behind the row's last step: the NEXT row's first gathers
      if (k + 1 < nsteps) step(bufB, mB, h + E + sub < end);
      mA = mC;
      mB = mD;
      h += 2 * E;
      k += 2;
    } while (k < nsteps);
    mA = m0n;
    mB = m1n;
    m0n = m0nn;
    m1n = m1nn;
    if (grad) {
      // transposing reduction over the E groups of the wave (lanes with equal lig)
      const float gsc = -grad_scale;
      float e[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e[2 * j] = j < 2 * K4 ? acc[j < 2 * K4 ? j : 0].x : 0.0f;
        e[2 * j + 1] = j < 2 * K4 ? acc[j < 2 * K4 ? j : 0].y : 0.0f;
      }
      // xor 32: element k with k + 8 -> lanes < 32 keep k, lanes >= 32 keep k + 8
      float e8[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e8[k] = wide_fold32(e[k], e[k + 8]);
      if constexpr (GL == 32) {
        // lane (b5, lig): float4s 2 b5 and 2 b5 + 1 of the row
        const int j0 = (lane >> 5) * 2;
        store_col(v, j0, wide_f4{e8[0], e8[1], e8[2], e8[3]} * gsc);
        store_col(v, j0 + 1, wide_f4{e8[4], e8[5], e8[6], e8[7]} * gsc);
      } else {
        // xor 16: element k with k + 4 -> rows 0, 2 keep k, rows 1, 3 keep k + 4
        float e4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) e4[k] = wide_fold16(e8[k], e8[k + 4]);
        if constexpr (GL <= 8) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            e4[k] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e4[k]), 0x128, 0xF, 0xF, true));  // row_ror:8
        }
        if constexpr (GL <= 4) {
          // (period 8 now: one more rotation by 4 adds the other residue)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            e4[k] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e4[k]), 0x124, 0xF, 0xF, true));  // row_ror:4
        }
        if constexpr (GL <= 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            e4[k] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e4[k]), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        }
        if constexpr (GL == 1) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            e4[k] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(e4[k]), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        }
        // lane (b5, b4, .): float4 number b4 + 2 b5 of the row; below GL = 16 the lanes with (lane & 15) >= GL hold
        // copies
        const int j = (lane >> 4) & 3;
        if ((lane & 15) < GL) store_col(v, j, wide_f4{e4[0], e4[1], e4[2], e4[3]} * gsc);
      }
    }
    // xv <- -x_v of the next row (see step()); the asm keeps hipcc from folding the sign back into a subtraction
#pragma unroll
    for (int j = 0; j < K4; ++j) {
      xv[j] = -xvn[j];
      asm volatile("" : "+v"(xv[j]));
    }
    r = r_n;
    beg = beg_n;
    end = end_n;
    r_n = r_nn;
    beg_n = beg_nn;
    end_n = end_nn;
    r_nn = r_n3;
    beg_nn = beg_n3;
    end_nn = end_n3;
    r_n3 = r_n4;
  }
  const double bs = mde_block_sum((double)loss, smem);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = bs;
}

// loss = scale * sum(partials[0..nb)) in a fixed order (one block)
__global__ void k_finalize_loss(double* __restrict__ partials, int nb, double scale,
                                float* __restrict__ loss_out) {
  __shared__ double smem[8];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += partials[i];
  const double t = mde_block_sum(s, smem);
  if (threadIdx.x == 0) {
    *loss_out = (float)(t * scale);
    partials[MDE_PARTIALS_LOSS_D] = t * scale;
  }
}

// ---------------------------------------------------------------- functors of the unfused paths
struct FnScatter {  // a0 = d(mean f)/d(dist) from autograd, a1 = dist  (average_distortion.py:81)
  MDE_DEV void eval(float ss, float a0, float a1, float& f, float& gd) const {
    f = 0.0f;
    gd = a0 / a1;
  }
};
struct FnDistBackward {  // backward of the 2-norm, average_distortion.py:46-52
  MDE_DEV void eval(float ss, float a0, float a1, float& f, float& gd) const {
    f = 0.0f;
    gd = a0 * mde_rcp(mde_sqrt(ss));
  }
};

// ---------------------------------------------------------------- launch helpers
struct FusedArgs {
  const mde_plan* plan;
  const float* X;
  int d;
  const float *a0, *a1;
  int a0_scalar, a1_scalar;
  float* grad;
  double* partials;
  float inv_p, grad_scale;
  hipStream_t st;
  int nblocks;  // out
  // in: where the loss goes (a kernel that has a follow-up launch anyway reduces it there and sets
  // loss_done; otherwise mde_average_distortion launches k_finalize_loss)
  float* loss_out = nullptr;
  double loss_scale = 0.0;
  int loss_done = 0;
};

static int g_group_override = 0;  // MDE_GROUP env (tuning experiments)
static int pick_group(float avg_degree) {
  if (g_group_override == 0) {
    const char* e = getenv("MDE_GROUP");
    g_group_override = e ? atoi(e) : -1;
  }
  if (g_group_override > 0) return g_group_override;
  // few, very long rows (dense problems such as all-pairs distance graphs): widen the group so
  // that the grid still fills the chip
  if (avg_degree >= 1024.f) return 64;
  if (avg_degree >= 256.f) return 32;
  if (avg_degree >= 48.f) return 16;
  if (avg_degree >= 20.f) return 8;
  return 4;
}

template <int D, int G, bool IND, class Fn>
static int launch_small_g(FusedArgs& A, const Fn& fn) {
  const mde_plan* P = A.plan;
  const int nrows = (int)(mde_plan_row_hi(P) - mde_plan_row_lo(P));
  const int64_t threads = (int64_t)nrows * G;
  const int nb = mde_grid(threads, MDE_BLOCK, 2048);
  A.nblocks = nb;
  hipLaunchKernelGGL((k_fused_small<D, G, IND, Fn>), dim3(nb), dim3(MDE_BLOCK), 0, A.st, nrows,
                     (int)mde_plan_row_lo(P), mde_plan_rowptr(P), mde_plan_nbr(P), mde_plan_eid(P), A.a0,
                     A.a1, A.a0_scalar, A.a1_scalar, A.X, A.grad, A.partials, fn, A.inv_p,
                     A.grad_scale);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
// MDE_FLAT env: unset / 1 the edge-balanced kernel on sparse graphs, 2 always, 0 never (the row-per-group kernel)
static int flat_mode() {
  const char* e = getenv("MDE_FLAT");  // (read per call: a test can switch it inside one process)
  return e ? atoi(e) : 1;
}

template <int D, bool IND, class Fn>
static int launch_flat(FusedArgs& A, const Fn& fn) {
  mde_plan* P = const_cast<mde_plan*>(A.plan);
  int rc = mde_plan_flat(P, A.st);
  if (rc != MDE_OK) return rc;
  const int64_t nloc = P->row_hi - P->row_lo;
  const int phase = (int)(P->h_offset % MDE_FLAT_T);
  float* gloc = A.grad ? A.grad + (size_t)P->row_lo * D : nullptr;
  if (A.grad && P->has_empty) MDE_HIP(hipMemsetAsync(gloc, 0, (size_t)nloc * D * sizeof(float), A.st));
  const int nb = mde_grid(P->n_tiles * 64, MDE_BLOCK, 2048);
  A.nblocks = nb;
  hipLaunchKernelGGL((k_fused_flat<D, IND, Fn>), dim3(nb), dim3(MDE_BLOCK), 0, A.st, P->H, phase, P->n_tiles,
                     (int)P->row_lo, P->hrow, P->nbr, P->eid, A.a0, A.a1, A.a0_scalar, A.a1_scalar, A.X, A.grad,
                     P->flat_rec, A.partials, fn, A.inv_p, A.grad_scale);
  MDE_LAUNCH_CHECK();
  if (A.grad) {
    const unsigned fb = (unsigned)((P->n_tiles + MDE_BLOCK - 1) / MDE_BLOCK);
    hipLaunchKernelGGL(k_flat_fixup<D>, dim3(fb + (A.loss_out ? 1u : 0u)), dim3(MDE_BLOCK), 0, A.st, P->n_tiles, phase,
                       (int)P->row_lo, P->rowptr, P->flat_rec, A.grad, A.grad_scale, A.partials, nb, A.loss_scale,
                       A.loss_out);
    MDE_LAUNCH_CHECK();
    if (A.loss_out) A.loss_done = 1;
  }
  return MDE_OK;
}

template <int D, bool IND, class Fn>
static int launch_small(FusedArgs& A, const Fn& fn) {
  // sparse graphs (the usual case): edge-balanced tiles.  Dense problems (hundreds of half-edges
  // per row on average, e.g. all-pairs distance graphs) keep a row per 32 / 64 lanes: enough loads
  // in flight per row, and no segmented scan (config 3: 0.67 vs 0.77 ms per evaluation)
  const int fm = flat_mode();
  if (fm != 0 && A.plan->H > 0 && (fm == 2 || A.plan->avg_degree < 256.f))
    return launch_flat<D, IND, Fn>(A, fn);
  switch (pick_group(mde_plan_avg_degree(A.plan))) {
    case 4: return launch_small_g<D, 4, IND, Fn>(A, fn);
    case 8: return launch_small_g<D, 8, IND, Fn>(A, fn);
    case 32: return launch_small_g<D, 32, IND, Fn>(A, fn);
    case 64: return launch_small_g<D, 64, IND, Fn>(A, fn);
    default: return launch_small_g<D, 16, IND, Fn>(A, fn);
  }
}
template <int GL, int K, bool IND, class Fn>
static int launch_wide_gk(FusedArgs& A, const Fn& fn) {
  const mde_plan* P = A.plan;
  const int nrows = (int)(mde_plan_row_hi(P) - mde_plan_row_lo(P));
  const int nb = mde_grid((int64_t)nrows * 64, MDE_BLOCK, 2048);
  A.nblocks = nb;
  hipLaunchKernelGGL((k_fused_wide<GL, K, IND, Fn>), dim3(nb), dim3(MDE_BLOCK), 0, A.st, nrows,
                     (int)mde_plan_row_lo(P), A.d, mde_plan_rowptr(P), mde_plan_nbr(P), mde_plan_eid(P),
                     A.a0, A.a1, A.a0_scalar, A.a1_scalar, A.X, A.grad, A.partials, fn, A.inv_p,
                     A.grad_scale);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
template <int GL, int K4, bool IND, class Fn>
static int launch_wide4_gk(FusedArgs& A, const Fn& fn) {
  const mde_plan* P = A.plan;
  const int nrows = (int)(mde_plan_row_hi(P) - mde_plan_row_lo(P));
  const int nb = mde_grid((int64_t)nrows * 64, MDE_BLOCK, 2048);
  A.nblocks = nb;
  hipLaunchKernelGGL((k_fused_wide4<GL, K4, IND, Fn>), dim3(nb), dim3(MDE_BLOCK), 0, A.st, nrows,
                     (int)mde_plan_row_lo(P), A.d / 4, mde_plan_rowptr(P), mde_plan_nbr(P), mde_plan_eid(P),
                     A.a0, A.a1, A.a0_scalar, A.a1_scalar, reinterpret_cast<const wide_f4*>(A.X),
                     reinterpret_cast<wide_f4*>(A.grad), A.partials, fn, A.inv_p, A.grad_scale);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
template <int GL, int K4, bool IND, class Fn>
static int launch_wide4p(FusedArgs& A, const Fn& fn) {
  const mde_plan* P = A.plan;
  const int nrows = (int)(mde_plan_row_hi(P) - mde_plan_row_lo(P));
  int nb = mde_grid((int64_t)nrows * 64, MDE_BLOCK, 2048);
  nb = (nb + 7) & ~7;  // (the row order is per XCD: block b runs on XCD b % 8)
  A.nblocks = nb;
  // chunks of up to 2048 rows, at least 64 chunks (8 per XCD) so that short row ranges still fill every XCD
  const int chunk_env = getenv("MDE_WIDE_CHUNK") ? atoi(getenv("MDE_WIDE_CHUNK")) : 11;  // (read per call: tests switch it)
  int chunk_log = chunk_env;
  while (chunk_log > 0 && ((int64_t)64 << chunk_log) > nrows) --chunk_log;
  // graphs whose locality is hidden by the vertex numbering: a breadth-first processing order, built once per plan
  // and kept only when it brings the endpoints of an edge closer together (mde_plan.hip: mde_plan_row_order)
  const int order_env = getenv("MDE_ROW_ORDER") ? atoi(getenv("MDE_ROW_ORDER")) : 1;
  if (order_env && P->order_state == 0) {
    const int rc = mde_plan_row_order(const_cast<mde_plan*>(P), order_env, A.st, nullptr);
    if (rc != MDE_OK) return rc;
  }
  hipLaunchKernelGGL((k_fused_wide4p<GL, K4, IND, Fn>), dim3(nb), dim3(MDE_BLOCK), 0, A.st, nrows,
                     (int)mde_plan_row_lo(P), A.d, chunk_log, P->order, mde_plan_rowptr(P), mde_plan_nbr(P), mde_plan_eid(P),
                     A.a0, A.a1, A.a0_scalar, A.a1_scalar, A.X, A.grad, A.partials, fn, A.inv_p, A.grad_scale);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
template <bool IND, class Fn>
static int launch_wide(FusedArgs& A, const Fn& fn) {
  const int d = A.d;
  {
    // Round 6: every d = 5 .. 512 takes the pipelined kernel (MDE_WIDE_P=0: off).  Its 16-byte accesses start at any
    // float of a row (4-byte aligned; the hardware's unaligned access mode), so d need not be a multiple of 4.
    const int d4 = (d + 3) >> 2;  // float4 columns of a row
    const int wide_p = getenv("MDE_WIDE_P") ? atoi(getenv("MDE_WIDE_P")) : 1;
    if (wide_p && d >= 5 && d4 <= 128) {
      // GL lanes x K4 float4s cover the row: GL * (K4 - 1) < d4 <= GL * K4 (the lanes of the last float4 that lie
      // behind the row's end are masked)
      if (d4 == 2) return launch_wide4p<1, 2, IND, Fn>(A, fn);
      if (d4 == 3) return launch_wide4p<1, 3, IND, Fn>(A, fn);
      if (d4 == 4) return launch_wide4p<1, 4, IND, Fn>(A, fn);
      if (d4 <= 6) return launch_wide4p<2, 3, IND, Fn>(A, fn);
      if (d4 <= 8) return launch_wide4p<2, 4, IND, Fn>(A, fn);
      if (d4 <= 12) return launch_wide4p<4, 3, IND, Fn>(A, fn);
      if (d4 <= 16) return launch_wide4p<4, 4, IND, Fn>(A, fn);
      if (d4 <= 24) return launch_wide4p<8, 3, IND, Fn>(A, fn);
      if (d4 <= 32) return launch_wide4p<8, 4, IND, Fn>(A, fn);
      if (d4 <= 48) return launch_wide4p<16, 3, IND, Fn>(A, fn);
      if (d4 <= 64) return launch_wide4p<16, 4, IND, Fn>(A, fn);
      if (d4 <= 96) return launch_wide4p<32, 3, IND, Fn>(A, fn);
      return launch_wide4p<32, 4, IND, Fn>(A, fn);
    }
  }
  if ((d & 3) == 0 && ((reinterpret_cast<uintptr_t>(A.X) | reinterpret_cast<uintptr_t>(A.grad)) & 15) == 0) {
    const int d4 = d >> 2;
    if (d4 <= 2) return launch_wide4_gk<2, 1, IND, Fn>(A, fn);
    if (d4 <= 4) return launch_wide4_gk<4, 1, IND, Fn>(A, fn);
    if (d4 <= 8) return launch_wide4_gk<8, 1, IND, Fn>(A, fn);
    if (d4 <= 16) return launch_wide4_gk<16, 1, IND, Fn>(A, fn);
    // Round 6: FOUR float4s per lane where the row is wide enough -- 8 lanes cooperate on a 512-byte row instead of
    // 32, a wave step takes 8 half-edges instead of 2 and the cross-lane sum of |x_v - x_u|^2 is 3 steps instead of 5:
    // config 5 (d = 128, uniform graph) 3.13 -> 2.93 ms (6.5 -> 7.0 of the 7.4 TB/s random-row ceiling), neighbours within
    // 1000 / 100 rows 2.36 -> 1.66 / 1.84 -> 1.20 ms (`profiles/r06_d128_locality.txt`).  MDE_WIDE_K4=1: one float4 per lane.
    static const int k4 = getenv("MDE_WIDE_K4") ? atoi(getenv("MDE_WIDE_K4")) : 4;
    if (k4 >= 4 && d4 > 8) {
      if (d4 <= 16) return launch_wide4_gk<4, 4, IND, Fn>(A, fn);
      if (d4 <= 32) return launch_wide4_gk<8, 4, IND, Fn>(A, fn);
      if (d4 <= 64) return launch_wide4_gk<16, 4, IND, Fn>(A, fn);
      if (d4 <= 128) return launch_wide4_gk<32, 4, IND, Fn>(A, fn);
    }
    if (d4 <= 32) return launch_wide4_gk<32, 1, IND, Fn>(A, fn);
    if (d4 <= 64) return launch_wide4_gk<64, 1, IND, Fn>(A, fn);
    if (d4 <= 128) return launch_wide4_gk<64, 2, IND, Fn>(A, fn);
    if (d4 <= 256) return launch_wide4_gk<64, 4, IND, Fn>(A, fn);
    if (d4 <= 512) return launch_wide4_gk<64, 8, IND, Fn>(A, fn);
  }
  if (d <= 8) return launch_wide_gk<8, 1, IND, Fn>(A, fn);
  if (d <= 16) return launch_wide_gk<16, 1, IND, Fn>(A, fn);
  if (d <= 32) return launch_wide_gk<32, 1, IND, Fn>(A, fn);
  if (d <= 64) return launch_wide_gk<64, 1, IND, Fn>(A, fn);
  if (d <= 128) return launch_wide_gk<64, 2, IND, Fn>(A, fn);
  if (d <= 256) return launch_wide_gk<64, 4, IND, Fn>(A, fn);
  if (d <= 512) return launch_wide_gk<64, 8, IND, Fn>(A, fn);
  if (d <= 1024) return launch_wide_gk<64, 16, IND, Fn>(A, fn);
  if (d <= 2048) return launch_wide_gk<64, 32, IND, Fn>(A, fn);
  mde_set_error("embedding dimension %d > 2048 is not supported by the fused kernel", d);
  return MDE_E_UNSUPPORTED;
}
template <bool IND, class Fn>
static int launch_any_d(FusedArgs& A, const Fn& fn) {
  switch (A.d) {
    case 1: return launch_small<1, IND, Fn>(A, fn);
    case 2: return launch_small<2, IND, Fn>(A, fn);
    case 3: return launch_small<3, IND, Fn>(A, fn);
    case 4: return launch_small<4, IND, Fn>(A, fn);
    default: return launch_wide<IND, Fn>(A, fn);
  }
}

static MdeFuncArgs func_args(const mde_func* f) {
  MdeFuncArgs a;
  a.kind = f->kind;
  a.kind_neg = f->kind_neg;
  a.S = {f->s0, f->s1, f->s2};
  a.N = {f->n0, f->n1, f->n2};
  return a;
}

// dedicated instantiations for the functions the recipes use by default; everything else
// goes through the universal run-time functor.
static int dispatch_fused(FusedArgs& A, const mde_func* f) {
  const MdeFuncArgs a = func_args(f);
  if (A.d == 2 || A.d == 3) {
    const int ea = mde_exp_class(f->s0), en = mde_exp_class(f->n0);
#define SMALL(FN)                                       \
  do {                                                  \
    FN fn{a};                                           \
    if (A.d == 2) return launch_small<2, false, FN>(A, fn); \
    return launch_small<3, false, FN>(A, fn);           \
  } while (0)
    if (f->kind_neg == MDE_F_NONE) {
      if (f->kind == MDE_F_QUADRATIC) SMALL(FnSingle<MDE_F_QUADRATIC COMMA 0>);
      if (f->kind == MDE_F_LOG1P && ea == 2) SMALL(FnSingle<MDE_F_LOG1P COMMA 2>);
      if (f->kind == MDE_F_L_QUADRATIC) SMALL(FnSingle<MDE_F_L_QUADRATIC COMMA 0>);
      if (f->kind == MDE_F_L_ABSOLUTE) SMALL(FnSingle<MDE_F_L_ABSOLUTE COMMA 0>);
      if (f->kind == MDE_F_L_HUBER) SMALL(FnSingle<MDE_F_L_HUBER COMMA 0>);
    } else if (f->kind == MDE_F_LOG1P && ea == 2) {
      if (f->kind_neg == MDE_F_LOG && en == 1)
        SMALL(FnPushPull<MDE_F_LOG1P COMMA 2 COMMA MDE_F_LOG COMMA 1>);
      if (f->kind_neg == MDE_F_LOGRATIO && en == 3)
        SMALL(FnPushPull<MDE_F_LOG1P COMMA 2 COMMA MDE_F_LOGRATIO COMMA 3>);
    }
#undef SMALL
  }
  if (A.d > 4 && f->kind_neg == MDE_F_NONE && f->kind == MDE_F_LOG1P && mde_exp_class(f->s0) == 2) {
    // wide embeddings (config 5: d = 128): the run-time functor's switch and general pow cost 0.5 ms of
    // the 3.7 ms launch; Log1p(1.5) is the recipes' default attractive penalty
    FnSingle<MDE_F_LOG1P, 2> fn{a};
    return launch_wide<false, FnSingle<MDE_F_LOG1P, 2> >(A, fn);
  }
  FnRuntime fn{a};
  return launch_any_d<false, FnRuntime>(A, fn);
}

static int check_func(const mde_func* f) {
  if (!f || !mde_kind_valid(f->kind) || (f->kind_neg != MDE_F_NONE && !mde_kind_valid(f->kind_neg))) {
    mde_set_error("unknown distortion function kind");
    return MDE_E_UNSUPPORTED;
  }
  if (!f->a0) {
    mde_set_error("distortion function has no per-edge parameter array");
    return MDE_E_INVALID;
  }
  const bool needs_a1 = f->kind == MDE_F_L_WEIGHTED_QUADRATIC || f->kind == MDE_F_L_WEIGHTED_POWER ||
                        f->kind_neg == MDE_F_L_WEIGHTED_QUADRATIC ||
                        f->kind_neg == MDE_F_L_WEIGHTED_POWER;
  if (needs_a1 && !f->a1) {
    mde_set_error("weighted loss needs the second per-edge array (a1)");
    return MDE_E_INVALID;
  }
  return MDE_OK;
}

extern "C" int mde_average_distortion(mde_plan* plan, const float* X, int32_t d, const mde_func* f,
                                      float grad_scale, float* grad, float* loss_out, void* stream) {
  if (!plan || !X || d <= 0 || !loss_out) return MDE_E_INVALID;
  int rc = check_func(f);
  if (rc != MDE_OK) return rc;
  const int64_t p = mde_plan_p(plan);
  FusedArgs A;
  A.plan = plan;
  A.X = X;
  A.d = d;
  A.a0 = f->a0;
  A.a1 = f->a1;
  A.a0_scalar = f->a0_scalar;
  A.a1_scalar = f->a1_scalar;
  A.grad = grad;
  A.partials = mde_plan_partials(plan);
  A.inv_p = p > 0 ? (float)(1.0 / (double)p) : 0.0f;
  A.grad_scale = grad_scale;
  A.st = mde_stream(stream);
  A.nblocks = 0;
  // every edge is seen from both endpoints: weight 1/2; mean over p edges
  const double scale = p > 0 ? 0.5 / (double)p : 0.0;
  if (f->layout == 1) {
    // parameters are in LDS-ring order: the LDS-resident kernel (mde_ring.hip), which also
    // reduces the loss (last workgroup to arrive)
    rc = mde_ring_try(plan, X, d, f, grad_scale, grad, A.inv_p, A.st, &A.nblocks, loss_out, scale);
    if (rc == 0) {
      mde_set_error("mde_func.layout = 1 but the plan has no LDS-ring layout for d = %d", d);
      return MDE_E_INVALID;
    }
    return rc < 0 ? rc : MDE_OK;
  }
  A.loss_out = loss_out;
  A.loss_scale = scale;
  rc = dispatch_fused(A, f);
  if (rc != MDE_OK) return rc;
  if (!A.loss_done) {
    hipLaunchKernelGGL(k_finalize_loss, dim3(1), dim3(MDE_BLOCK), 0, A.st, A.partials, A.nblocks, scale,
                       loss_out);
    MDE_LAUNCH_CHECK();
  }
  return MDE_OK;
}

extern "C" int mde_scatter(const mde_plan* plan, const float* X, int32_t d, const float* gnorm,
                           const float* dist, float scale, float* grad, void* stream) {
  if (!plan || !X || d <= 0 || !gnorm || !dist || !grad) return MDE_E_INVALID;
  FusedArgs A;
  A.plan = plan;
  A.X = X;
  A.d = d;
  A.a0 = gnorm;
  A.a1 = dist;
  A.a0_scalar = 0;
  A.a1_scalar = 0;
  A.grad = grad;
  A.partials = mde_plan_partials(const_cast<mde_plan*>(plan));
  A.inv_p = 1.0f;
  A.grad_scale = scale;
  A.st = mde_stream(stream);
  FnScatter fn;
  return launch_any_d<true, FnScatter>(A, fn);
}

extern "C" int mde_distances_backward(const mde_plan* plan, const float* X, int32_t d,
                                      const float* gout, float* grad, void* stream) {
  if (!plan || !X || d <= 0 || !gout || !grad) return MDE_E_INVALID;
  FusedArgs A;
  A.plan = plan;
  A.X = X;
  A.d = d;
  A.a0 = gout;
  A.a1 = gout;
  A.a0_scalar = 0;
  A.a1_scalar = 0;
  A.grad = grad;
  A.partials = mde_plan_partials(const_cast<mde_plan*>(plan));
  A.inv_p = 1.0f;
  A.grad_scale = 1.0f;
  A.st = mde_stream(stream);
  FnDistBackward fn;
  return launch_any_d<true, FnDistBackward>(A, fn);
}

// ---------------------------------------------------------------- edge-order evaluators
// GL lanes per edge, components strided over the lanes.
template <int GL, bool WRITE_DIFF>
__global__ __launch_bounds__(MDE_BLOCK) void k_edge_order(int64_t p, int d,
                                                          const int64_t* __restrict__ edges,
                                                          const float* __restrict__ X,
                                                          float* __restrict__ out) {
  const int lig = threadIdx.x & (GL - 1);
  const int64_t g0 = ((int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x) / GL;
  const int64_t ng = ((int64_t)gridDim.x * MDE_BLOCK) / GL;
  for (int64_t k = g0; k < p; k += ng) {
    const longlong2 e = reinterpret_cast<const longlong2*>(edges)[k];
    float ss = 0.0f;
    for (int c = lig; c < d; c += GL) {
      const float df = X[e.x * d + c] - X[e.y * d + c];
      if constexpr (WRITE_DIFF) out[k * d + c] = df;
      ss = fmaf(df, df, ss);
    }
    if constexpr (!WRITE_DIFF) {
      ss = mde_group_sum<GL>(ss);
      // sqrt(sum of squares): correctly rounded sqrt, as torch's pow(2).sum().sqrt()
      if (lig == 0) out[k] = sqrtf(ss);
    }
  }
}

template <bool WRITE_DIFF>
static int launch_edge_order(int64_t n, int64_t p, const int64_t* edges, const float* X, int d,
                             float* out, hipStream_t st) {
  if (n <= 0 || p < 0 || d <= 0 || (p > 0 && (!edges || !X || !out))) return MDE_E_INVALID;
  if (p == 0) return MDE_OK;
#define EO(GLV)                                                                                   \
  hipLaunchKernelGGL((k_edge_order<GLV, WRITE_DIFF>), dim3(mde_grid(p * GLV, MDE_BLOCK, 4096)),   \
                     dim3(MDE_BLOCK), 0, st, p, d, edges, X, out)
  if (d <= 4)
    EO(1);
  else if (d <= 16)
    EO(4);
  else if (d <= 64)
    EO(16);
  else
    EO(64);
#undef EO
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

extern "C" int mde_differences(int64_t n, int64_t p, const int64_t* edges, const float* X, int32_t d,
                               float* diff_out, void* stream) {
  return launch_edge_order<true>(n, p, edges, X, d, diff_out, mde_stream(stream));
}
extern "C" int mde_distances(int64_t n, int64_t p, const int64_t* edges, const float* X, int32_t d,
                             float* dist_out, void* stream) {
  return launch_edge_order<false>(n, p, edges, X, d, dist_out, mde_stream(stream));
}

__global__ __launch_bounds__(MDE_BLOCK) void k_distortions(int64_t p, const float* __restrict__ dist,
                                                           const float* __restrict__ a0,
                                                           const float* __restrict__ a1, int a0_scalar,
                                                           int a1_scalar, FnRuntime fn,
                                                           float* __restrict__ out,
                                                           float* __restrict__ dout) {
  const float a0s = a0_scalar ? a0[0] : 0.0f;
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;
  for (int64_t k = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; k < p;
       k += (int64_t)gridDim.x * MDE_BLOCK) {
    const float dk = dist[k];
    const float p0 = a0_scalar ? a0s : a0[k];
    const float p1 = (a1 && !a1_scalar) ? a1[k] : a1s;
    float f, gd;
    fn.eval(dk * dk, p0, p1, f, gd);
    out[k] = f;
    if (dout) dout[k] = gd * dk;  // f'(d) = (f'(d)/d) d
  }
}

extern "C" int mde_distortions(int64_t p, const float* dist, const mde_func* f, float* out,
                               float* dout, void* stream) {
  if (p < 0 || (p > 0 && (!dist || !out))) return MDE_E_INVALID;
  int rc = check_func(f);
  if (rc != MDE_OK) return rc;
  if (p == 0) return MDE_OK;
  FnRuntime fn{func_args(f)};
  hipLaunchKernelGGL(k_distortions, dim3(mde_grid(p, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0,
                     mde_stream(stream), p, dist, f->a0, f->a1, f->a0_scalar, f->a1_scalar, fn, out, dout);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
