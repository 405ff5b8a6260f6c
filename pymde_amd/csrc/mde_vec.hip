// mde_vec.hip -- the solver-side kernels: constraint projections/retractions, small Gram
// matrices (f32 MFMA when the widths allow), fused vector statistics, and the device-resident
// L-BFGS memory.
//   constraints  [ref: pymde/constraints.py:94-200, pymde/util.py:129-171]
//   vector ops   [ref: pymde/lbfgs.py:350-376, 461-530; pymde/optim.py:94-147]
// All reductions are two-stage (block partials in double, one fixed-order final pass): no
// atomics, bitwise reproducible.
#include "mde_common.h"

#include <algorithm>
#include <cmath>
#include <atomic>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// mde_mfma.hip: the projections' kernels at widths 32 / 64 / 128
bool mde_mfma_width_ok(int d);
int mde_mfma_colmean(int64_t n, int d, const float* Z, double* partial, double* mean, hipStream_t st);
int mde_mfma_gram(int64_t n, int d, const float* A, const float* B, const double* mean, double* out, double* partial,
                  int64_t max_partial, hipStream_t st);
int mde_mfma_rmul(int64_t n, int d, const float* A, const double* M, const double* mean, float alpha, const float* base,
                  float* out, hipStream_t st);

// ---------------------------------------------------------------- work buffer layout
// [0, 4096)                    small scalars / staging
// [4096, 4096 + 8 d^2)         d x d matrices (Gram result, Newton-Schulz iterates, M)
// [4096 + 8 d^2, ...)          reduction partials (<= MDE_PARTIAL_DOUBLES)
#define MDE_SMALL_DOUBLES 4096
#define MDE_PARTIAL_DOUBLES (4 << 20)
#define MDE_RED_BLOCKS 256

extern "C" int64_t mde_work_doubles(int32_t d) {
  const int64_t dd = (int64_t)(d > 0 ? d : 1);
  return MDE_SMALL_DOUBLES + 8 * dd * dd + MDE_PARTIAL_DOUBLES;
}
// the small area, in doubles (one place: the users are spread over this file):
//   [0, 2048)      column means, Gram staging and other per-launch scalars
//   [2304, 3072)   k_lb_fused: three rows of MDE_LB_FUSED_MAXBLOCKS arrival flags (32-bit words)
//   [3072, 3076)   k_lb_fused / k_lb_rescue: gave-up, done and rescue-count words; [3076, 3088) probe stamps (-DMDE_LB_PROBE)
//   [3200]         the gate word of a pre-enqueued L-BFGS step (mde_turn_*)
//   [4064, 4096)   arrival tickets of the reductions that finish in their last workgroup
#define MDE_WS_MEAN_END 2048
#define MDE_WS_LB_FLAGS 2304
#define MDE_WS_LB_VERDICT 3072
#define MDE_WS_LB_VERDICT_END 3088
#define MDE_WS_TURN_GATE 3200
#define MDE_WS_TICKETS (MDE_SMALL_DOUBLES - 32)
static_assert(MDE_WS_MEAN_END <= MDE_WS_LB_FLAGS && MDE_WS_LB_VERDICT_END <= MDE_WS_TURN_GATE &&
                  MDE_WS_TURN_GATE + 1 <= MDE_WS_TICKETS && MDE_WS_TICKETS + 32 == MDE_SMALL_DOUBLES,
              "small area of the work buffer: regions overlap");
static inline double* work_mats(double* work) { return work + MDE_SMALL_DOUBLES; }
// arrival counters of the kernels that finish their reduction in the last workgroup: the last 32
// doubles of the small area (zero between launches; the caller zeroes the buffer once)
enum { TK_STATS = 0, TK_GRAM = 1, TK_CENTER = 2, TK_LB_STAGE = 3, TK_LB_COMBINE = 4, TK_RETRACT = 5 };
static inline unsigned int* work_ticket(double* work, int which) {
  return reinterpret_cast<unsigned int*>(work + MDE_WS_TICKETS) + which;
}
static inline double* work_partials(double* work, int d) {
  return work + MDE_SMALL_DOUBLES + 8 * (int64_t)d * d;
}

// device -> pinned host copy on the stream (the solver's one read-back per evaluation)
extern "C" int mde_copy_to_host(void* dst_host, const void* src_dev, int64_t bytes, void* stream) {
  if (!dst_host || !src_dev || bytes < 0) return MDE_E_INVALID;
  if (bytes == 0) return MDE_OK;
  MDE_HIP(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, mde_stream(stream)));
  return MDE_OK;
}

// ---------------------------------------------------------------- generic row reduction
// out[q] = reduce_b partial[q * nb + b]; mode_mask bit q set -> max, else sum
__global__ void k_reduce_rows(int nq, int nb, const double* __restrict__ partial,
                              unsigned long long max_mask, double* __restrict__ out) {
  __shared__ double smem[8];
  const int q = blockIdx.x;
  if (q >= nq) return;
  const bool is_max = (q < 64) && ((max_mask >> q) & 1ull);
  double s = is_max ? -1.0e308 : 0.0;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) {
    const double v = partial[(int64_t)q * nb + b];
    s = is_max ? (v > s ? v : s) : s + v;
  }
  const double r = is_max ? mde_block_max(s, smem) : mde_block_sum(s, smem);
  if (threadIdx.x == 0) out[q] = r;
}

// ---------------------------------------------------------------- axpy
__global__ __launch_bounds__(MDE_BLOCK) void k_axpy(int64_t N, float alpha,
                                                    const float* __restrict__ x,
                                                    const float* __restrict__ y,
                                                    float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * MDE_BLOCK;
  const int64_t N4 = N >> 2;
  const bool al = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                    reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  if (al) {
    for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N4; i += stride) {
      const float4 a = reinterpret_cast<const float4*>(x)[i];
      const float4 b = reinterpret_cast<const float4*>(y)[i];
      reinterpret_cast<float4*>(out)[i] = make_float4(fmaf(alpha, a.x, b.x), fmaf(alpha, a.y, b.y),
                                                      fmaf(alpha, a.z, b.z), fmaf(alpha, a.w, b.w));
    }
    for (int64_t i = (N4 << 2) + (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N; i += stride)
      out[i] = fmaf(alpha, x[i], y[i]);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N; i += stride)
      out[i] = fmaf(alpha, x[i], y[i]);
  }
}

static int axpy_impl(int64_t N, float alpha, const float* x, const float* y, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_axpy, dim3(mde_grid((N + 3) / 4, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, N, alpha, x, y, out);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
extern "C" int mde_axpy(int64_t N, float alpha, const float* x, const float* y, float* out,
                        void* stream) {
  if (N < 0 || (N > 0 && (!x || !y || !out))) return MDE_E_INVALID;
  if (N == 0) return MDE_OK;
  return axpy_impl(N, alpha, x, y, out, mde_stream(stream));
}

// eight block-wide reductions with two barriers: wave results to LDS, thread q < 8 combines the waves
// and publishes partial[q * nb + b] (rows in max_mask: maxima)
__device__ __forceinline__ void mde_publish8(const double (&v)[8], unsigned max_mask, double* partial, int nb, int b) {
  __shared__ double sm8[MDE_BLOCK / 64][8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const double r = ((max_mask >> q) & 1u) ? mde_wave_max(v[q]) : mde_wave_sum(v[q]);
    if (lane == 0) sm8[wave][q] = r;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    const int q = threadIdx.x;
    double r = sm8[0][q];
#pragma unroll
    for (int w = 1; w < MDE_BLOCK / 64; ++w) r = ((max_mask >> q) & 1u) ? (sm8[w][q] > r ? sm8[w][q] : r) : r + sm8[w][q];
    mde_st_partial(partial + q * nb + b, r);
  }
}

// ---------------------------------------------------------------- host mirror of the solver's read-back
// The last kernel of an evaluation can write [loss | status | pad | board[0,24)] straight into the pinned
// host mirror (device-visible at the same address) instead of a copy being enqueued behind it: one launch
// (~4 us of a ~0.1 ms iteration) less.  host == nullptr: nothing is written.
struct MdeMirror {
  const float* loss_dev;
  const int32_t* status;
  const double* board;   // the device board whose first 8 doubles are `stats` of the kernel
  char* host;
  int head_bytes;        // bytes of [loss | status | pad] in front of the board
  double seq;            // written behind the 24 board entries once they are out (polled by the host)
  // mde_turn_*: the first pass of the line search at t = 1 decided HERE (the host reads the verdict in
  // mirrored board slot 8 and does not test again), and a word that opens or closes the L-BFGS step of
  // the next iteration, which is queued right behind this kernel (see MdeGate).  gate == nullptr: none.
  unsigned int* gate;
  unsigned int gate_value;
  double f0, c1, c2;
};
// A launch that runs only when the iteration before it accepted its first trial point: the kernels of
// the next L-BFGS direction update are enqueued BEHIND an iteration's last kernel, before the host has
// seen its result (the host's turn-around -- mirror write, poll, test, launch -- was 15 us in which the
// GPU idled, `tools/iter_trace.sh`); that last kernel writes `value` to the word when the trial is
// accepted, anything else when it is not, and the gated kernels leave at once unless they find `value`.
// word == nullptr: an ordinary launch.
struct MdeGate {
  const unsigned int* word;
  unsigned int value;
};
__device__ __forceinline__ bool mde_gate_closed(const MdeGate& g) {
  return g.word != nullptr && __hip_atomic_load(g.word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != g.value;
}
// Where the TAGGED copy of the record lives: four 64-byte lines behind the plain mirror, each seven payload doubles
// and the sequence number -- [loss, status, board[0..24)] -- written by ONE store instruction of one wave, so that a
// line and its tag arrive together whatever the order the lines reach host memory in.  (Round 6: the plain form --
// data, system-scope fences, then the sequence word in a line of its own -- let the host read a record ONE
// ITERATION OLD about once in thirty cold-started solves: the sequence word can pass the data lines on the way to
// host memory (the solves' iterates were unaffected -- the device decides the trial from device memory -- but the
// recorded loss and step size of that iteration were the previous one's, and the bit-for-bit test of the turn calls
// against the call-by-call loop failed now and then).)
#define MDE_MIRROR_TAG_LINES 4
__host__ __device__ __forceinline__ double* mde_mirror_tagged(const double* host_board) {
  return reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(host_board + 32) + 63) & ~(uintptr_t)63);
}
__device__ __forceinline__ void mde_mirror_write(const MdeMirror& m) {
  if (!m.host) return;
  __shared__ double s_pay[7 * MDE_MIRROR_TAG_LINES];
  __threadfence_block();
  __syncthreads();  // the statistics rows written by this block are visible to it
  double* hb = reinterpret_cast<double*>(m.host + m.head_bytes);
  if (threadIdx.x < 24 && !(m.gate && threadIdx.x == 8)) {  // (slot 8: the verdict below, when there is one)
    const volatile double* b = m.board;
    s_pay[2 + threadIdx.x] = b[threadIdx.x];
  }
  if (threadIdx.x >= 24 && threadIdx.x < 26) s_pay[2 + threadIdx.x] = 0.0;  // (spare payload slots)
  if (threadIdx.x == 32) s_pay[0] = (double)*reinterpret_cast<const volatile float*>(m.loss_dev);
  if (threadIdx.x == 33) s_pay[1] = m.status ? (double)*reinterpret_cast<const volatile int32_t*>(m.status) : 0.0;
  if (m.gate && threadIdx.x == 34) {
    // [ref: lbfgs.py:88-110, first pass of the bracketing loop at t = 1: finite, Armijo, curvature]
    // (separately rounded product and sum: the value the host-side search computes)
    // (five independent loads in flight at once: relaxed device-scope loads, not `volatile` ones, which
    // the compiler keeps apart -- a memory latency each, in the last kernel of every iteration)
    const double gtd_new = __hip_atomic_load(m.board + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double gtd0 = __hip_atomic_load(m.board + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double nonfinite = __hip_atomic_load(m.board + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double f_new = (double)__hip_atomic_load(m.loss_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int st = m.status ? __hip_atomic_load(m.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const bool bad = !(fabs(f_new) <= 1.7976931348623157e308) || nonfinite != 0.0;
    const bool accept = !bad && !(f_new > __dadd_rn(m.f0, __dmul_rn(m.c1, gtd0))) && (fabs(gtd_new) <= -__dmul_rn(m.c2, gtd0));
    s_pay[2 + 8] = accept ? 1.0 : 0.0;
    *m.gate = (accept && st == 0) ? m.gate_value : 0u;
  }
  __syncthreads();
  // the plain mirror (what rounds 3-5 read; kept for whoever looks at the pinned buffer directly)
  if (threadIdx.x < 24) hb[threadIdx.x] = s_pay[2 + threadIdx.x];
  if (threadIdx.x == 32) *reinterpret_cast<float*>(m.host) = (float)s_pay[0];
  if (threadIdx.x == 33) *reinterpret_cast<int32_t*>(m.host + 4) = (int32_t)s_pay[1];
  // the tagged lines: the second wave's lanes 0..31 write 32 consecutive doubles, slot 7 of every line the tag
  if (threadIdx.x >= 64 && threadIdx.x < 64 + 8 * MDE_MIRROR_TAG_LINES) {
    const int t = (int)threadIdx.x - 64;
    mde_mirror_tagged(hb)[t] = (t & 7) == 7 ? m.seq : s_pay[(t >> 3) * 7 + (t & 7)];
  }
  // the sequence word goes out behind the data: every writer fences to system scope, then one thread
  // publishes (the host polls it instead of waiting for the stream's completion signal)
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    hb[24] = m.seq;
    __threadfence_system();
  }
}

// ---------------------------------------------------------------- vector statistics
// partial[q * nb + b], q: 0 g.d 1 g.g 2 sum|g| 3 max|g| 4 #nonfinite 5 d.d 6 max|d| 7 x.x
__global__ __launch_bounds__(MDE_BLOCK) void k_vec_stats(int64_t N, const float* __restrict__ g,
                                                         const float* __restrict__ d,
                                                         const float* __restrict__ x,
                                                         double* __restrict__ partial,
                                                         double* stats,
                                                         unsigned int* __restrict__ ticket, MdeMirror mirror) {
  __shared__ double smem[8];
  double gd = 0, gg = 0, g1 = 0, gm = 0, nf = 0, dd = 0, dm = 0, xx = 0;
  auto one = [&](float gv, float dvf, float xvf) __attribute__((always_inline)) {
    const double gvd = gv;
    gg += gvd * gvd;
    const double ag = fabs(gvd);
    g1 += ag;
    gm = ag > gm ? ag : gm;  // NaN never wins; counted below
    nf += (fabsf(gv) <= 3.402823466e+38f) ? 0.0 : 1.0;
    if (d) {
      const double dv = dvf;
      gd += gvd * dv;
      dd += dv * dv;
      const double ad = fabs(dv);
      dm = ad > dm ? ad : dm;
    }
    if (x) {
      const double xv = xvf;
      xx += xv * xv;
    }
  };
  // 16-byte loads (256 MB vectors at d = 128: the 4-byte form ran at 0.7 TB/s); each thread keeps its
  // own fixed subsequence, so the sums do not depend on timing
  const bool vec = ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(d ? d : g) |
                     reinterpret_cast<uintptr_t>(x ? x : g)) & 15) == 0;
  const int64_t N4 = vec ? (N >> 2) : 0;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* d4 = reinterpret_cast<const float4*>(d);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const int64_t stride = (int64_t)gridDim.x * MDE_BLOCK;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // four 16-byte loads per array in flight per thread, unpredicated in the main loop (a predicated
  // load is a branch: the loads of a trip then go out one memory latency after the other)
  int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (d && x) {
    for (; i + 3 * stride < N4; i += 4 * stride) {
      float4 ga[4], da[4], xa[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ga[u] = g4[i + u * stride];
        da[u] = d4[i + u * stride];
        xa[u] = x4[i + u * stride];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        one(ga[u].x, da[u].x, xa[u].x);
        one(ga[u].y, da[u].y, xa[u].y);
        one(ga[u].z, da[u].z, xa[u].z);
        one(ga[u].w, da[u].w, xa[u].w);
      }
    }
  }
  // the remaining trips (fewer than four when d and x are given), their loads issued TOGETHER from clamped
  // indices and only the arithmetic predicated (round 5: 2M-element vectors on 256 workgroups are ~8 float4 per
  // thread -- one unrolled trip and then three or four dependent ones, a memory latency each: 18 -> ~11 us; the
  // elements a thread adds, and their order, are what they were)
  while (i < N4) {
    float4 ga[4], da[4], xa[4];
    bool in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t iu = i + u * stride;
      in[u] = iu < N4;
      const int64_t ic = in[u] ? iu : N4 - 1;
      ga[u] = g4[ic];
      da[u] = d ? d4[ic] : z4;
      xa[u] = x ? x4[ic] : z4;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (in[u]) {
        one(ga[u].x, da[u].x, xa[u].x);
        one(ga[u].y, da[u].y, xa[u].y);
        one(ga[u].z, da[u].z, xa[u].z);
        one(ga[u].w, da[u].w, xa[u].w);
      }
    i += 4 * stride;
  }
  for (int64_t i = (N4 << 2) + (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N; i += stride)
    one(g[i], d ? d[i] : 0.0f, x ? x[i] : 0.0f);
  const int nb = gridDim.x, b = blockIdx.x;
  const double v[8] = {gd, gg, g1, gm, nf, dd, dm, xx};
  mde_publish8(v, (1u << 3) | (1u << 6), partial, nb, b);
  // the last block to arrive reduces the partials (fixed order: independent of arrival order)
  if (!mde_last_block(ticket)) return;
  mde_final_rows(8, nb, partial, stats, (1ull << 3) | (1ull << 6));
  mde_mirror_write(mirror);
}

static int vec_stats_impl(int64_t N, const float* g, const float* d, const float* x, double* stats,
                          double* work, hipStream_t st, MdeMirror mirror = MdeMirror{}) {
  const int nb = mde_grid(N, MDE_BLOCK * 8, MDE_RED_BLOCKS);
  hipLaunchKernelGGL(k_vec_stats, dim3(nb), dim3(MDE_BLOCK), 0, st, N, g, d, x, work + MDE_SMALL_DOUBLES, stats,
                     work_ticket(work, TK_STATS), mirror);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

extern "C" int mde_vec_stats(int64_t N, const float* g, const float* d, const float* x,
                             double* stats, double* work, void* stream) {
  if (N <= 0 || !g || !stats || !work) return MDE_E_INVALID;
  return vec_stats_impl(N, g, d, x, stats, work, mde_stream(stream));
}

// ---------------------------------------------------------------- column sums / centring
// threads are laid out (rows_per_pass x dp), dp = pow2 >= min(d,256); coalesced over columns
// STEP: Z <- X0 + t DIR first (the line-search trial point: one launch less than axpy + centring; the
// sums are those of the rounded floats, exactly what the two launches produce)
template <bool STEP>
__global__ __launch_bounds__(MDE_BLOCK) void k_colsum(int64_t n, int d, int dp, float* Z, const float* X0,
                                                      const float* DIR, float t,
                                                      double* __restrict__ partial /* [nb][d] */,
                                                      double* __restrict__ mean, unsigned int* ticket) {
  __shared__ double sm[MDE_BLOCK];
  const int tc = threadIdx.x & (dp - 1);
  const int tr = threadIdx.x / dp;
  const int rpp = MDE_BLOCK / dp;
  for (int c0 = 0; c0 < d; c0 += dp) {
    const int c = c0 + tc;
    double s = 0.0;
    if (c < d) {
      // eight rows of the thread's sequence in flight (clamped indices, the arithmetic predicated; round 5: one
      // row per trip was a memory latency per row -- 17 us for 24 MB at n = 1M, d = 2); the rows a thread adds,
      // and their order, are what they were
      const int64_t rstep = (int64_t)gridDim.x * rpp;
      for (int64_t r0 = (int64_t)blockIdx.x * rpp + tr; r0 < n; r0 += 8 * rstep) {
        float a[8], b[8];
        bool in[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t r = r0 + u * rstep;
          in[u] = r < n;
          const int64_t rc = in[u] ? r : n - 1;
          if (STEP) {
            a[u] = DIR[rc * d + c];
            b[u] = X0[rc * d + c];
          } else {
            a[u] = Z[rc * d + c];
            b[u] = 0.0f;
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (in[u]) {
            float z;
            if (STEP) {
              z = fmaf(t, a[u], b[u]);
              Z[(r0 + u * rstep) * d + c] = z;
            } else {
              z = a[u];
            }
            s += (double)z;
          }
      }
    }
    __syncthreads();
    sm[threadIdx.x] = s;
    __syncthreads();
    if (tr == 0 && c < d) {
      double t = 0.0;
      for (int k = 0; k < rpp; ++k) t += sm[k * dp + tc];
      mde_st_partial(partial + (int64_t)blockIdx.x * d + c, t);
    }
  }
  // narrow matrices (ticket given): the last workgroup to arrive turns the partial sums into the
  // column means; wide ones leave that to k_colsum_final (one wave per column, in parallel)
  if (!ticket || !mde_last_block(ticket)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nb = gridDim.x;
  for (int c = wave; c < d; c += MDE_BLOCK / 64) {
    double s = 0.0;
    for (int b = lane; b < nb; b += 64) s += mde_ld_partial(partial + (int64_t)b * d + c);
    s = mde_wave_sum(s);
    if (lane == 0) mean[c] = s / (double)n;
  }
}
// mean[c] = (sum_b partial[b][c]) / n
// (one wave per column: lane-strided partial sums + a fixed-order wave reduction)
__global__ __launch_bounds__(64) void k_colsum_final(int nb, int d, int64_t n, const double* __restrict__ partial,
                                                     double* __restrict__ mean) {
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = threadIdx.x; b < nb; b += 64) s += partial[(int64_t)b * d + c];
  s = mde_wave_sum(s);
  if (threadIdx.x == 0) mean[c] = s / (double)n;
}
__global__ __launch_bounds__(MDE_BLOCK) void k_sub_mean(int64_t N, int d, float* __restrict__ Z,
                                                        const double* __restrict__ mean, double scale = 1.0) {
  const int64_t stride = (int64_t)gridDim.x * MDE_BLOCK;
  for (int64_t i0 = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i0 < N; i0 += 4 * stride) {
    float z[4];
    bool in[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      in[u] = i0 + u * stride < N;
      z[u] = Z[in[u] ? i0 + u * stride : N - 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (in[u]) Z[i0 + u * stride] = (float)((double)z[u] - scale * mean[(i0 + u * stride) % d]);  // (scale = 1: exact)
  }
}

static int pow2_ge(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}

// halves: 1 = the column means only (left in work[0 .. d)), 2 = the subtraction only (of scale x work[0 .. d)), 3 = both
static int center_impl(int64_t n, int d, float* Z, double* work, hipStream_t st, const float* X0 = nullptr,
                       const float* DIR = nullptr, float t = 0.0f, int halves = 3, double scale = 1.0) {
  double* mean = work;  // d doubles (d <= 2048 fits the small area)
  double* partial = work_partials(work, d);
  int dp = pow2_ge(d);
  if (dp > MDE_BLOCK) dp = MDE_BLOCK;
  const int rpp = MDE_BLOCK / dp;
  int nb = mde_grid(n, rpp * 8, MDE_RED_BLOCKS);
  // (a last workgroup adding d columns of nb partials one after the other only pays for a few columns)
  const bool fused_final = d <= 16;
  if (!(halves & 1)) {
    hipLaunchKernelGGL(k_sub_mean, dim3(mde_grid(n * d, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, n * d, d, Z, mean, scale);
    MDE_LAUNCH_CHECK();
    return MDE_OK;
  }
  if (DIR)
    hipLaunchKernelGGL(k_colsum<true>, dim3(nb), dim3(MDE_BLOCK), 0, st, n, d, dp, Z, X0, DIR, t, partial, mean,
                       fused_final ? work_ticket(work, TK_CENTER) : (unsigned int*)nullptr);
  else
    hipLaunchKernelGGL(k_colsum<false>, dim3(nb), dim3(MDE_BLOCK), 0, st, n, d, dp, Z, X0, DIR, t, partial, mean,
                       fused_final ? work_ticket(work, TK_CENTER) : (unsigned int*)nullptr);
  MDE_LAUNCH_CHECK();
  if (!fused_final) {
    hipLaunchKernelGGL(k_colsum_final, dim3(d), dim3(64), 0, st, nb, d, n, partial, mean);
    MDE_LAUNCH_CHECK();
  }
  if (!(halves & 2)) return MDE_OK;
  hipLaunchKernelGGL(k_sub_mean, dim3(mde_grid(n * d, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, n * d, d, Z,
                     mean, scale);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// The two halves of mde_center_step for a row-sharded solve (round 6): _begin forms this rank's rows of Z = X + t dir
// (dir == NULL: Z as it is) and leaves the column MEANS over these n rows in work[0 .. d); the caller sums them across
// the ranks in place; _end subtracts scale x work[0 .. d) (scale = this rank's share of the rows, n / n_total).
extern "C" int mde_center_step_begin(int64_t n, int32_t d, const float* X, const float* dir, float t, float* Z, double* work,
                                     void* stream) {
  if (n <= 0 || d <= 0 || d > 2048 || !Z || !work || (dir && !X)) return MDE_E_INVALID;
  return center_impl(n, d, Z, work, mde_stream(stream), dir ? X : nullptr, dir, t, 1);
}
extern "C" int mde_center_step_end(int64_t n, int32_t d, float* Z, double* work, double scale, void* stream) {
  if (n <= 0 || d <= 0 || d > 2048 || !Z || !work) return MDE_E_INVALID;
  return center_impl(n, d, Z, work, mde_stream(stream), nullptr, nullptr, 0.0f, 2, scale);
}

extern "C" int mde_center(int64_t n, int32_t d, float* Z, double* work, void* stream) {
  if (n <= 0 || d <= 0 || d > 2048 || !Z || !work) return MDE_E_INVALID;
  return center_impl(n, d, Z, work, mde_stream(stream));
}

// Z <- centre(X + t DIR): the trial point of the line search under the Centered constraint
extern "C" int mde_center_step(int64_t n, int32_t d, const float* X, const float* dir, float t, float* Z,
                               double* work, void* stream) {
  if (n <= 0 || d <= 0 || d > 2048 || !X || !dir || !Z || !work) return MDE_E_INVALID;
  return center_impl(n, d, Z, work, mde_stream(stream), X, dir, t);
}

// ---------------------------------------------------------------- anchors
__global__ void k_anchor_rows(int64_t na, int d, const int64_t* __restrict__ anchors,
                              const float* __restrict__ values, float* __restrict__ Z) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < na * d;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t a = i / d;
    const int c = (int)(i % d);
    Z[anchors[a] * d + c] = values ? values[i] : 0.0f;
  }
}
extern "C" int mde_anchor_rows(int64_t n_anchors, int32_t d, const int64_t* anchors,
                               const float* values, float* Z, void* stream) {
  if (n_anchors < 0 || d <= 0 || (n_anchors > 0 && (!anchors || !Z))) return MDE_E_INVALID;
  if (n_anchors == 0) return MDE_OK;
  hipLaunchKernelGGL(k_anchor_rows, dim3(mde_grid(n_anchors * d, MDE_BLOCK)), dim3(MDE_BLOCK), 0,
                     mde_stream(stream), n_anchors, d, anchors, values, Z);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// ---------------------------------------------------------------- rows on a sphere
// [ref: constraints.py:203-231, `_Sphere`: private there, used by no recipe]  X == NULL: the retraction
// Z[r] <- (Z[r] / |Z[r]|) radius (divide, then multiply, as the reference does); else the tangent projection
// Z[r] -= (1 / radius) (Z[r] . X[r]) X[r] -- the reference's own scale, 1 / radius and not 1 / radius^2.
// Narrow rows: one thread per row; wide rows: one wave per row, lanes strided over the columns.
template <bool TANGENT>
__global__ __launch_bounds__(MDE_BLOCK) void k_sphere_narrow(int64_t n, int d, const float* __restrict__ X,
                                                             float* __restrict__ Z, float radius) {
  for (int64_t r = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * MDE_BLOCK) {
    float* z = Z + r * d;
    float s = 0.0f;
    if (TANGENT) {
      const float* x = X + r * d;
      for (int c = 0; c < d; ++c) s = fmaf(z[c], x[c], s);
      const float k = (1.0f / radius) * s;
      for (int c = 0; c < d; ++c) z[c] -= k * x[c];
    } else {
      for (int c = 0; c < d; ++c) s = fmaf(z[c], z[c], s);
      const float nrm = sqrtf(s);
      for (int c = 0; c < d; ++c) z[c] = (z[c] / nrm) * radius;
    }
  }
}
template <bool TANGENT>
__global__ __launch_bounds__(MDE_BLOCK) void k_sphere_wide(int64_t n, int d, const float* __restrict__ X,
                                                           float* __restrict__ Z, float radius) {
  const int lane = threadIdx.x & 63;
  const int64_t wpg = MDE_BLOCK / 64;
  for (int64_t r = (int64_t)blockIdx.x * wpg + (threadIdx.x >> 6); r < n; r += (int64_t)gridDim.x * wpg) {
    float* z = Z + r * d;
    const float* x = TANGENT ? X + r * d : z;
    double s = 0.0;
    for (int c = lane; c < d; c += 64) s += (double)z[c] * (double)x[c];
    s = mde_wave_sum(s);  // (a butterfly: every lane holds the sum)
    if (TANGENT) {
      const float k = (1.0f / radius) * (float)s;
      for (int c = lane; c < d; c += 64) z[c] -= k * x[c];
    } else {
      const float nrm = sqrtf((float)s);
      for (int c = lane; c < d; c += 64) z[c] = (z[c] / nrm) * radius;
    }
  }
}
extern "C" int mde_sphere_rows(int64_t n, int32_t d, const float* X, float* Z, float radius, void* stream) {
  if (n <= 0 || d <= 0 || !Z || !(radius > 0.0f)) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  if (d <= 32) {
    const int nb = mde_grid(n, MDE_BLOCK);
    if (X)
      hipLaunchKernelGGL(k_sphere_narrow<true>, dim3(nb), dim3(MDE_BLOCK), 0, st, n, d, X, Z, radius);
    else
      hipLaunchKernelGGL(k_sphere_narrow<false>, dim3(nb), dim3(MDE_BLOCK), 0, st, n, d, X, Z, radius);
  } else {
    const int nb = mde_grid(n, MDE_BLOCK / 64);
    if (X)
      hipLaunchKernelGGL(k_sphere_wide<true>, dim3(nb), dim3(MDE_BLOCK), 0, st, n, d, X, Z, radius);
    else
      hipLaunchKernelGGL(k_sphere_wide<false>, dim3(nb), dim3(MDE_BLOCK), 0, st, n, d, X, Z, radius);
  }
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// ---------------------------------------------------------------- Gram matrices  out = A^T B
// (a) tiny widths: one thread per row, da*db register accumulators in double
template <int DA, int DB>
__global__ __launch_bounds__(MDE_BLOCK) void k_gram_tiny(int64_t n, const float* __restrict__ A,
                                                         const float* __restrict__ B,
                                                         double* __restrict__ partial /*[m][nb]*/,
                                                         double* __restrict__ out, unsigned int* ticket) {
  __shared__ double smem[8];
  double acc[DA * DB];
#pragma unroll
  for (int i = 0; i < DA * DB; ++i) acc[i] = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * MDE_BLOCK) {
    double a[DA], b[DB];
#pragma unroll
    for (int i = 0; i < DA; ++i) a[i] = A[r * DA + i];
#pragma unroll
    for (int j = 0; j < DB; ++j) b[j] = B[r * DB + j];
#pragma unroll
    for (int i = 0; i < DA; ++i)
#pragma unroll
      for (int j = 0; j < DB; ++j) acc[i * DB + j] = fma(a[i], b[j], acc[i * DB + j]);
  }
#pragma unroll
  for (int q = 0; q < DA * DB; ++q) {
    const double r = mde_block_sum(acc[q], smem);
    if (threadIdx.x == 0) mde_st_partial(partial + (int64_t)q * gridDim.x + blockIdx.x, r);
  }
  // the last workgroup to arrive adds the partials (one wave per entry, fixed order)
  if (!mde_last_block(ticket)) return;
  mde_final_rows(DA * DB, gridDim.x, partial, out, 0ull);
}

// (b) any widths: 16x16 output tile per block, rows split into chunks (grid.y)
__global__ __launch_bounds__(MDE_BLOCK) void k_gram_generic(int64_t n, int da, int db,
                                                            const float* __restrict__ A,
                                                            const float* __restrict__ B,
                                                            int64_t rows_per_chunk, int tiles_j,
                                                            double* __restrict__ partial /*[m][nc]*/) {
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x % tiles_j;
  const int i = ti * 16 + (threadIdx.x >> 4), j = tj * 16 + (threadIdx.x & 15);
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > n) r1 = n;
  double acc = 0.0;
  if (i < da && j < db)
    for (int64_t r = r0; r < r1; ++r) acc = fma((double)A[r * da + i], (double)B[r * db + j], acc);
  if (i < da && j < db) partial[((int64_t)i * db + j) * gridDim.y + blockIdx.y] = acc;
}

// (c) widths that are multiples of 32: f32 MFMA (v_mfma_f32_32x32x2_f32), one 32x32 output tile
// per wave, k = rows of the chunk.  A-operand lane l: A[r + (l>>5)][ti*32 + (l&31)], B alike:
// both are fully coalesced 128-byte row segments.  C/D map: col = l&31,
// row = (reg&3) + 8*(reg>>2) + 4*(l>>5).  Accumulates in f32 over <= 4096 rows per chunk, the
// chunk partials are then summed in double.
__global__ __launch_bounds__(MDE_BLOCK) void k_gram_mfma(int64_t n, int da, int db,
                                                         const float* __restrict__ A,
                                                         const float* __restrict__ B,
                                                         int64_t rows_per_chunk, int tiles_j,
                                                         int ntiles,
                                                         double* __restrict__ partial /*[m][nc]*/) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= ntiles) return;
  const int ti = tile / tiles_j, tj = tile % tiles_j;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk;
  int64_t r1 = r0 + rows_per_chunk;
  if (r1 > n) r1 = n;
  const int kk = lane >> 5, cc = lane & 31;
  const float* pa = A + (int64_t)ti * 32 + cc;
  const float* pb = B + (int64_t)tj * 32 + cc;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  for (int64_t r = r0; r < r1; r += 2) {
    const int64_t rr = r + kk;
    const bool ok = rr < r1;
    const float a = ok ? pa[rr * da] : 0.0f;
    const float b = ok ? pb[rr * db] : 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int row = (q & 3) + 8 * (q >> 2) + 4 * kk;
    const int i = ti * 32 + row, j = tj * 32 + cc;
    partial[((int64_t)i * db + j) * gridDim.y + blockIdx.y] = (double)acc[q];
  }
}

// out[q] = sum_c partial[q * nc + c]: one wave per output entry (small outputs; a serial loop per
// entry is latency-bound) or one thread per entry (large outputs)
__global__ __launch_bounds__(64) void k_gram_final_wave(int nc, const double* __restrict__ partial,
                                                        double* __restrict__ out) {
  const int64_t q = blockIdx.x;
  double s = 0.0;
  for (int c = threadIdx.x; c < nc; c += 64) s += partial[q * nc + c];
  s = mde_wave_sum(s);
  if (threadIdx.x == 0) out[q] = s;
}
__global__ void k_gram_final(int64_t m, int nc, const double* __restrict__ partial,
                             double* __restrict__ out) {
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < m;
       q += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int c = 0; c < nc; ++c) s += partial[q * nc + c];
    out[q] = s;
  }
}

static int g_no_mfma = -1;
static int gram_impl(int64_t n, int da, int db, const float* A, const float* B, double* out,
                     double* partial, unsigned int* ticket, hipStream_t st) {
  const int64_t m = (int64_t)da * db;
  if (g_no_mfma < 0) {
    const char* e = getenv("MDE_NO_MFMA");
    g_no_mfma = (e && atoi(e)) ? 1 : 0;
  }
#define TINY(DA_, DB_)                                                                              \
  if (da == DA_ && db == DB_) {                                                                     \
    const int nb = mde_grid(n, MDE_BLOCK * 2, MDE_RED_BLOCKS);                                      \
    hipLaunchKernelGGL((k_gram_tiny<DA_, DB_>), dim3(nb), dim3(MDE_BLOCK), 0, st, n, A, B, partial,  \
                       out, ticket);                                                                \
    MDE_LAUNCH_CHECK();                                                                             \
    return MDE_OK;                                                                                  \
  }
  TINY(1, 1) TINY(2, 2) TINY(3, 3) TINY(4, 4)
#undef TINY
  // chunking: as many row chunks as the partial area holds, at most 256, >= 2 rows each
  int64_t nc = MDE_PARTIAL_DOUBLES / m;
  if (nc > 256) nc = 256;
  if (nc < 1) {
    mde_set_error("gram: %d x %d output does not fit the work buffer", da, db);
    return MDE_E_UNSUPPORTED;
  }
  int64_t rpc = (n + nc - 1) / nc;
  if (rpc < 64) rpc = 64;
  const bool mfma = !g_no_mfma && (da % 32 == 0) && (db % 32 == 0);
  if (mfma) {
    if (rpc > 4096) rpc = 4096;  // bound the f32 accumulation length
    rpc = (rpc + 1) & ~(int64_t)1;
  }
  nc = (n + rpc - 1) / rpc;
  if (mfma && nc * m > MDE_PARTIAL_DOUBLES) {
    // too many chunks for the partial area: lengthen chunks
    nc = MDE_PARTIAL_DOUBLES / m;
    rpc = ((n + nc - 1) / nc + 1) & ~(int64_t)1;
    nc = (n + rpc - 1) / rpc;
  }
  if (mfma) {
    const int tiles_j = db / 32, ntiles = (da / 32) * tiles_j;
    hipLaunchKernelGGL(k_gram_mfma, dim3((ntiles + 3) / 4, (unsigned)nc), dim3(MDE_BLOCK), 0, st, n, da,
                       db, A, B, rpc, tiles_j, ntiles, partial);
  } else {
    const int tiles_i = (da + 15) / 16, tiles_j = (db + 15) / 16;
    hipLaunchKernelGGL(k_gram_generic, dim3(tiles_i * tiles_j, (unsigned)nc), dim3(MDE_BLOCK), 0, st, n,
                       da, db, A, B, rpc, tiles_j, partial);
  }
  MDE_LAUNCH_CHECK();
  if (m <= 65536)
    hipLaunchKernelGGL(k_gram_final_wave, dim3((unsigned)m), dim3(64), 0, st, (int)nc, partial, out);
  else
    hipLaunchKernelGGL(k_gram_final, dim3(mde_grid(m, 256, 256)), dim3(256), 0, st, m, (int)nc, partial,
                       out);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

extern "C" int mde_gram(int64_t n, int32_t da, int32_t db, const float* A, const float* B, double* out,
                        double* work, void* stream) {
  if (n <= 0 || da <= 0 || db <= 0 || !A || !B || !out || !work) return MDE_E_INVALID;
  const int dm = da > db ? da : db;
  return gram_impl(n, da, db, A, B, out, work_partials(work, dm), work_ticket(work, TK_GRAM), mde_stream(stream));
}

// ---------------------------------------------------------------- Z (+)= alpha * A M
// out[r][j] = (base ? base[r][j] : 0) + alpha * sum_c A[r][c] M[c][j].   out may alias A or base.
template <int D>
__global__ __launch_bounds__(MDE_BLOCK) void k_rmul_tiny(int64_t n, const float* __restrict__ A,
                                                         const double* __restrict__ M, float alpha,
                                                         const float* base, float* out) {
  float m[D * D];
#pragma unroll
  for (int i = 0; i < D * D; ++i) m[i] = (float)(M[i] * (double)alpha);
  for (int64_t r = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; r < n;
       r += (int64_t)gridDim.x * MDE_BLOCK) {
    float a[D], o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) a[c] = A[r * D + c];
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = base ? base[r * D + j] : 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int j = 0; j < D; ++j) o[j] = fmaf(a[c], m[c * D + j], o[j]);
#pragma unroll
    for (int j = 0; j < D; ++j) out[r * D + j] = o[j];
  }
}

__global__ __launch_bounds__(MDE_BLOCK) void k_rmul_generic(int64_t n, int d, int d2, int rows_tile,
                                                            const float* __restrict__ A,
                                                            const double* __restrict__ M, float alpha,
                                                            const float* base, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sa = reinterpret_cast<float*>(smem_raw);  // [rows_tile][d]
  for (int64_t t0 = (int64_t)blockIdx.x * rows_tile; t0 < n; t0 += (int64_t)gridDim.x * rows_tile) {
    int64_t rows = n - t0;
    if (rows > rows_tile) rows = rows_tile;
    __syncthreads();
    for (int64_t i = threadIdx.x; i < rows * d; i += MDE_BLOCK) sa[i] = A[t0 * d + i];
    __syncthreads();
    for (int64_t i = threadIdx.x; i < rows * d2; i += MDE_BLOCK) {
      const int64_t r = i / d2;
      const int j = (int)(i % d2);
      float acc = 0.0f;
      for (int c = 0; c < d; ++c) acc = fmaf(sa[r * d + c], (float)M[(int64_t)c * d2 + j], acc);
      const float b = base ? base[(t0 + r) * d2 + j] : 0.0f;
      out[(t0 + r) * d2 + j] = fmaf(alpha, acc, b);
    }
  }
}

// Widths that are multiples of 32 with d * d2 floats <= 64 KB (d = d2 <= 128): f32 MFMA
// (v_mfma_f32_32x32x2_f32).  A 256-thread workgroup takes 32 rows: their d columns are staged in
// LDS with coalesced loads (row stride d + 1: the A-operand reads then walk distinct banks), M is
// staged once as fp32, and wave w produces the 32 x 32 output tiles j = w, w + 4, ...  Operand
// map as in k_gram_mfma: A-operand lane l = (row l & 31, k = l >> 5), B-operand lane l =
// (k = l >> 5, column l & 31); C/D: column l & 31, row (reg & 3) + 8 (reg >> 2) + 4 (l >> 5).
__global__ __launch_bounds__(MDE_BLOCK) void k_rmul_mfma(int64_t n, int d, int d2, const float* __restrict__ A,
                                                         const double* __restrict__ M, float alpha,
                                                         const float* base, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* sm = reinterpret_cast<float*>(smem_raw);  // [d][d2] fp32 copy of M
  float* sa = sm + (size_t)d * d2;                 // [32][d + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kk = lane >> 5, cc = lane & 31;
  const int lda = d + 1;
  for (int i = tid; i < d * d2; i += MDE_BLOCK) sm[i] = (float)M[i];
  for (int64_t t0 = (int64_t)blockIdx.x * 32; t0 < n; t0 += (int64_t)gridDim.x * 32) {
    __syncthreads();  // (previous tile's reads of sa are done; sm is ready on the first trip)
    for (int i = tid; i < 32 * d; i += MDE_BLOCK) {
      const int r = i / d, c = i - r * d;
      sa[r * lda + c] = (t0 + r < n) ? A[(t0 + r) * d + c] : 0.0f;
    }
    __syncthreads();
    for (int tj = wave; tj * 32 < d2; tj += MDE_BLOCK / 64) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
      const float* pa = sa + cc * lda + kk;
      const float* pb = sm + (size_t)kk * d2 + tj * 32 + cc;
#pragma unroll 4
      for (int c = 0; c < d; c += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[c], pb[(size_t)c * d2], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (q & 3) + 8 * (q >> 2) + 4 * kk;
        if (t0 + row < n) {
          const int64_t o = (t0 + row) * d2 + tj * 32 + cc;
          const float b = base ? base[o] : 0.0f;
          out[o] = fmaf(alpha, acc[q], b);
        }
      }
    }
  }
}

static int rmul_impl(int64_t n, int d, int d2, const float* A, const double* M, float alpha,
                     const float* base, float* out, hipStream_t st) {
#define RT(D_)                                                                                     \
  if (d == D_ && d2 == D_) {                                                                       \
    hipLaunchKernelGGL((k_rmul_tiny<D_>), dim3(mde_grid(n, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, n, A, \
                       M, alpha, base, out);                                                       \
    MDE_LAUNCH_CHECK();                                                                            \
    return MDE_OK;                                                                                 \
  }
  RT(1) RT(2) RT(3) RT(4)
#undef RT
  if (g_no_mfma < 0) {
    const char* e = getenv("MDE_NO_MFMA");
    g_no_mfma = (e && atoi(e)) ? 1 : 0;
  }
  if (!g_no_mfma && d % 32 == 0 && d2 % 32 == 0 && (size_t)d * d2 * sizeof(float) <= 65536 && (A != out || d == d2)) {
    const size_t lds = ((size_t)d * d2 + 32 * (size_t)(d + 1)) * sizeof(float);
    hipLaunchKernelGGL(k_rmul_mfma, dim3(mde_grid(n, 32, 2048)), dim3(MDE_BLOCK), lds, st, n, d, d2, A, M, alpha,
                       base, out);
    MDE_LAUNCH_CHECK();
    return MDE_OK;
  }
  int rows_tile = 8192 / d;
  if (rows_tile < 1) rows_tile = 1;
  if (rows_tile > 64) rows_tile = 64;
  const size_t lds = (size_t)rows_tile * d * sizeof(float);
  hipLaunchKernelGGL(k_rmul_generic, dim3(mde_grid(n, rows_tile, 2048)), dim3(MDE_BLOCK), lds, st, n, d,
                     d2, rows_tile, A, M, alpha, base, out);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

extern "C" int mde_right_multiply(int64_t n, int32_t d, int32_t d2, const float* A, const double* M,
                                  float* out, void* stream) {
  if (n <= 0 || d <= 0 || d2 <= 0 || d > 8192 || !A || !M || !out) return MDE_E_INVALID;
  if (A == out && d != d2) return MDE_E_INVALID;
  return rmul_impl(n, d, d2, A, M, 1.0f, nullptr, out, mde_stream(stream));
}

extern "C" int mde_right_multiply_add(int64_t n, int32_t d, int32_t d2, const float* A, const double* M,
                                      float alpha, const float* base, float* out, void* stream) {
  if (n <= 0 || d <= 0 || d2 <= 0 || d > 8192 || !A || !M || !out) return MDE_E_INVALID;
  if (A == out && d != d2) return MDE_E_INVALID;
  return rmul_impl(n, d, d2, A, M, alpha, base, out, mde_stream(stream));
}

__global__ __launch_bounds__(MDE_BLOCK) void k_row_scale(int64_t N, int d, const float* __restrict__ scale,
                                                         float* __restrict__ Z) {
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * MDE_BLOCK)
    Z[i] *= scale[i / d];
}
// Z[r, :] += shift  (shift: d device doubles)
__global__ __launch_bounds__(MDE_BLOCK) void k_shift_rows(int64_t N, int d, const double* __restrict__ shift,
                                                          float* __restrict__ Z) {
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * MDE_BLOCK)
    Z[i] = (float)((double)Z[i] + shift[i % d]);
}
extern "C" int mde_shift_rows(int64_t n, int32_t d, const double* shift, float* Z, void* stream) {
  if (n <= 0 || d <= 0 || !shift || !Z) return MDE_E_INVALID;
  hipLaunchKernelGGL(k_shift_rows, dim3(mde_grid(n * d, MDE_BLOCK)), dim3(MDE_BLOCK), 0, mde_stream(stream),
                     n * d, d, shift, Z);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
extern "C" int mde_row_scale(int64_t n, int32_t d, const float* scale, float* Z, void* stream) {
  if (n <= 0 || d <= 0 || !scale || !Z) return MDE_E_INVALID;
  hipLaunchKernelGGL(k_row_scale, dim3(mde_grid(n * d, MDE_BLOCK)), dim3(MDE_BLOCK), 0, mde_stream(stream),
                     n * d, d, scale, Z);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// ---------------------------------------------------------------- Standardized: tangent space
// Z -= (1/n) X (Z^T X)                                    [ref: constraints.py:186-192]
extern "C" int mde_std_tangent(int64_t n, int32_t d, const float* X, float* Z, double* work,
                               void* stream) {
  if (n <= 0 || d <= 0 || d > 2048 || !X || !Z || !work) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  double* G = work_mats(work);  // d x d : G[i][j] = sum_r Z[r][i] X[r][j]
  if (mde_mfma_width_ok(d)) {
    int rc = mde_mfma_gram(n, d, Z, X, nullptr, G, work_partials(work, d), MDE_PARTIAL_DOUBLES, st);
    if (rc != MDE_OK) return rc;
    return mde_mfma_rmul(n, d, X, G, nullptr, (float)(-1.0 / (double)n), Z, Z, st);
  }
  int rc = gram_impl(n, d, d, Z, X, G, work_partials(work, d), work_ticket(work, TK_GRAM), st);
  if (rc != MDE_OK) return rc;
  return rmul_impl(n, d, d, X, G, (float)(-1.0 / (double)n), Z, Z, st);
}

// The same with the statistics of the result folded in (d <= 4: one launch less per evaluation): what
// mde_std_tangent(X, Z) followed by mde_vec_stats(Z, dir, X) leaves in `stats`.
template <int D>
__global__ __launch_bounds__(MDE_BLOCK) void k_rmul_tiny_stats(int64_t n, const float* __restrict__ X,
                                                               const double* __restrict__ G, float alpha, float* Z,
                                                               const float* __restrict__ dir, double* __restrict__ partial,
                                                               double* stats, unsigned int* __restrict__ ticket,
                                                               MdeMirror mirror) {
  float m[D * D];
#pragma unroll
  for (int i = 0; i < D * D; ++i) m[i] = (float)(G[i] * (double)alpha);
  double gd = 0, gg = 0, g1 = 0, gm = 0, nf = 0, dd = 0, dm = 0, xx = 0;
  for (int64_t r = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * MDE_BLOCK) {
    float a[D], o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) a[c] = X[r * D + c];
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = Z[r * D + j];
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int j = 0; j < D; ++j) o[j] = fmaf(a[c], m[c * D + j], o[j]);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      Z[r * D + j] = o[j];
      const double gv = o[j], xv = a[j];
      gg += gv * gv;
      const double ag = fabs(gv);
      g1 += ag;
      gm = ag > gm ? ag : gm;
      nf += (fabsf(o[j]) <= 3.402823466e+38f) ? 0.0 : 1.0;
      xx += xv * xv;
      if (dir) {
        const double dv = dir[r * D + j];
        gd += gv * dv;
        dd += dv * dv;
        const double ad = fabs(dv);
        dm = ad > dm ? ad : dm;
      }
    }
  }
  const double v[8] = {gd, gg, g1, gm, nf, dd, dm, xx};
  mde_publish8(v, (1u << 3) | (1u << 6), partial, gridDim.x, blockIdx.x);
  if (!mde_last_block(ticket)) return;
  mde_final_rows(8, gridDim.x, partial, stats, (1ull << 3) | (1ull << 6));
  mde_mirror_write(mirror);
}

static int std_tangent_stats_impl(int64_t n, int32_t d, const float* X, float* Z, const float* dir, double* stats,
                                  double* work, void* stream, MdeMirror mirror) {
  hipStream_t st = mde_stream(stream);
  if (d > 4) {
    const int rc = mde_std_tangent(n, d, X, Z, work, stream);
    if (rc != MDE_OK) return rc;
    return vec_stats_impl(n * (int64_t)d, Z, dir, X, stats, work, st, mirror);
  }
  double* G = work_mats(work);
  const int rc = gram_impl(n, d, d, Z, X, G, work_partials(work, d), work_ticket(work, TK_GRAM), st);
  if (rc != MDE_OK) return rc;
  const int nb = mde_grid(n, MDE_BLOCK * 2, MDE_RED_BLOCKS);
  const float alpha = (float)(-1.0 / (double)n);
#define TINY(D_)                                                                                                  \
  if (d == D_)                                                                                                    \
    hipLaunchKernelGGL(k_rmul_tiny_stats<D_>, dim3(nb), dim3(MDE_BLOCK), 0, st, n, X, G, alpha, Z, dir,           \
                       work_partials(work, d), stats, work_ticket(work, TK_STATS), mirror);
  // (the partials go behind the matrices: G itself is still being read by workgroups that start late)
  TINY(1) TINY(2) TINY(3) TINY(4)
#undef TINY
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

extern "C" int mde_std_tangent_stats(int64_t n, int32_t d, const float* X, float* Z, const float* dir, double* stats,
                                     double* work, void* stream) {
  if (n <= 0 || d <= 0 || d > 2048 || !X || !Z || !stats || !work) return MDE_E_INVALID;
  return std_tangent_stats_impl(n, d, X, Z, dir, stats, work, stream, MdeMirror{});
}

// ---------------------------------------------------------------- C^{-1/2} of a small SPD matrix
// One workgroup.  d = 1, 2: closed form.  d >= 3: coupled Newton-Schulz iteration in double
//   Y0 = C/s, Z0 = I;  T = (3I - Z Y)/2;  Y <- Y T;  Z <- T Z;   Z -> (C/s)^{-1/2}
// with s = ||C||_F.  M = out_scale * C^{-1/2}.  status != 0: C not numerically SPD.
// (a device function of one 256-thread workgroup: also called by the last workgroup of the fused
// small-d retraction kernel, right after it has written C -- hence no __restrict__ / read-only loads)
__device__ void invsqrt_block(int d, const double* C, double out_scale, double* M,
                              double* scratch /* 5 d^2 */, int32_t* status) {
  __shared__ double smem[8];
  __shared__ double sh_val;
  __shared__ int sh_done;
  const int tid = threadIdx.x;
  const int m = d * d;
  if (d == 1) {
    if (tid == 0) {
      const double c = C[0];
      const bool ok = c > 0.0 && c < 1e300;
      M[0] = ok ? out_scale / sqrt(c) : 0.0;
      if (status && !ok) *status = 1;
    }
    return;
  }
  if (d == 2) {
    if (tid == 0) {
      const double a = C[0], b = 0.5 * (C[1] + C[2]), c = C[3];
      const double det = a * c - b * b, tr = a + c;
      const bool ok = det > 0.0 && tr > 0.0 && det < 1e300;
      if (ok) {
        const double s = sqrt(det), t = sqrt(tr + 2.0 * s);
        const double k = out_scale / (s * t);
        M[0] = (c + s) * k;
        M[1] = -b * k;
        M[2] = -b * k;
        M[3] = (a + s) * k;
      } else {
        M[0] = M[1] = M[2] = M[3] = 0.0;
        if (status) *status = 1;
      }
    }
    return;
  }
  double* Y = scratch;
  double* Z = scratch + m;
  double* T = scratch + 2 * (int64_t)m;
  double* Yn = scratch + 3 * (int64_t)m;
  double* Zn = scratch + 4 * (int64_t)m;
  // s = ||C||_F
  double ss = 0.0;
  for (int i = tid; i < m; i += MDE_BLOCK) ss += C[i] * C[i];
  const double tot = mde_block_sum(ss, smem);
  if (tid == 0) sh_val = sqrt(tot);
  __syncthreads();
  const double s = sh_val;
  if (!(s > 0.0) || !(s < 1e300)) {
    for (int i = tid; i < m; i += MDE_BLOCK) M[i] = 0.0;
    if (tid == 0 && status) *status = 1;
    return;
  }
  for (int i = tid; i < m; i += MDE_BLOCK) {
    const int r = i / d, c = i % d;
    Y[i] = 0.5 * (C[i] + C[c * d + r]) / s;  // symmetrise
    Z[i] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  bool converged = false;
  for (int it = 0; it < 200; ++it) {
    // T = (3I - Z Y)/2, residual = max |I - Z Y|
    double res = 0.0;
    for (int i = tid; i < m; i += MDE_BLOCK) {
      const int r = i / d, c = i % d;
      double acc = 0.0;
      for (int k = 0; k < d; ++k) acc = fma(Z[r * d + k], Y[k * d + c], acc);
      const double e = ((r == c) ? 1.0 : 0.0) - acc;
      res = fabs(e) > res ? fabs(e) : res;
      T[i] = ((r == c) ? 1.0 : 0.0) + 0.5 * e;
    }
    const double rmax = mde_block_max(res, smem);
    if (tid == 0) sh_done = (rmax < 1e-13) ? 1 : ((rmax == rmax && rmax < 1e300) ? 0 : 2);
    __syncthreads();
    if (sh_done == 1) {
      converged = true;
      break;
    }
    if (sh_done == 2) break;
    for (int i = tid; i < m; i += MDE_BLOCK) {
      const int r = i / d, c = i % d;
      double ay = 0.0, az = 0.0;
      for (int k = 0; k < d; ++k) {
        ay = fma(Y[r * d + k], T[k * d + c], ay);
        az = fma(T[r * d + k], Z[k * d + c], az);
      }
      Yn[i] = ay;
      Zn[i] = az;
    }
    __syncthreads();
    for (int i = tid; i < m; i += MDE_BLOCK) {
      Y[i] = Yn[i];
      Z[i] = Zn[i];
    }
    __syncthreads();
  }
  const double k = out_scale / sqrt(s);
  for (int i = tid; i < m; i += MDE_BLOCK) M[i] = converged ? Z[i] * k : 0.0;
  if (tid == 0 && status && !converged) *status = 1;
}
__global__ __launch_bounds__(MDE_BLOCK) void k_invsqrt(int d, const double* C, double out_scale, double* M,
                                                       double* scratch /* 5 d^2 */, int32_t* status) {
  invsqrt_block(d, C, out_scale, M, scratch, status);
}

// ---------------------------------------------------------------- small-d retraction in two launches
// Z <- sqrt(n) (Z - mean) C^{-1/2} for d <= 4.  One pass accumulates the column sums and the RAW
// Gram matrix Z^T Z in double; the last workgroup to arrive forms mean, C = Z^T Z - n mean mean^T
// (the Gram matrix of the centred rows) and out_scale C^{-1/2}; k_center_rmul_tiny then applies
// both.  (Seven launches -- column sums, their reduction, the subtraction, Gram, its reduction,
// the inverse square root, the multiply -- when composed from the general-d pieces.)
// X0 / DIR given: Z <- X0 + t DIR first (the trial point of the line search: one launch less).
template <int D>
__global__ __launch_bounds__(MDE_BLOCK) void k_retract_stats_tiny(int64_t n, float* Z, const float* X0,
                                                                  const float* DIR, float t,
                                                                  int demean, double out_scale,
                                                                  double* partial /* [D*D + D][nb] */,
                                                                  double* mean /* D */, double* C, double* M,
                                                                  double* scratch, int32_t* status,
                                                                  unsigned int* ticket) {
  __shared__ double smem[8];
  constexpr int NQ = D * D + D;
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = 0.0;
  for (int64_t r = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * MDE_BLOCK) {
    double a[D];
    if (DIR) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const float z = fmaf(t, DIR[r * D + i], X0[r * D + i]);
        Z[r * D + i] = z;
        a[i] = z;
      }
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i) a[i] = Z[r * D + i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      acc[D * D + i] += a[i];
#pragma unroll
      for (int j = 0; j < D; ++j) acc[i * D + j] = fma(a[i], a[j], acc[i * D + j]);
    }
  }
  const int nb = gridDim.x;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const double r = mde_block_sum(acc[q], smem);
    if (threadIdx.x == 0) mde_st_partial(partial + (int64_t)q * nb + blockIdx.x, r);
  }
  if (!mde_last_block(ticket)) return;
  __shared__ double tot[NQ];
  mde_final_rows(NQ, nb, partial, tot, 0ull);
  __syncthreads();
  if (threadIdx.x < D * D) {
    const int i = threadIdx.x / D, j = threadIdx.x % D;
    const double mi = demean ? tot[D * D + i] / (double)n : 0.0;
    const double mj = demean ? tot[D * D + j] / (double)n : 0.0;
    C[threadIdx.x] = tot[threadIdx.x] - (double)n * mi * mj;
    if (j == 0) mean[i] = mi;
  }
  __threadfence_block();
  __syncthreads();
  invsqrt_block(D, C, out_scale, M, scratch, status);
}

template <int D>
__global__ __launch_bounds__(MDE_BLOCK) void k_center_rmul_tiny(int64_t n, const double* __restrict__ mean,
                                                                const double* __restrict__ M,
                                                                float* __restrict__ Z) {
  double mu[D];
  float m[D * D];
#pragma unroll
  for (int i = 0; i < D; ++i) mu[i] = mean[i];
#pragma unroll
  for (int i = 0; i < D * D; ++i) m[i] = (float)M[i];
  for (int64_t r = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; r < n; r += (int64_t)gridDim.x * MDE_BLOCK) {
    float a[D], o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) a[c] = (float)((double)Z[r * D + c] - mu[c]);
#pragma unroll
    for (int j = 0; j < D; ++j) o[j] = 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c)
#pragma unroll
      for (int j = 0; j < D; ++j) o[j] = fmaf(a[c], m[c * D + j], o[j]);
#pragma unroll
    for (int j = 0; j < D; ++j) Z[r * D + j] = o[j];
  }
}


// ---------------------------------------------------------------- C^{-1/2}, d >= 32: two launches per step
// The single-workgroup iteration above needs three d^3 double products per step out of global
// memory with 256 threads (22 ms at d = 128).  Here every product is spread over d^2 / 256
// workgroups, two launches per step (T = (3 I - Z Y) / 2 with the residual; then Y T and T Z).
// Every step has its own residual word, so a kernel can tell from the words of the earlier steps
// whether the iteration is over -- no flag kernel, no state to reset -- and the host reads the
// words back after each batch of MDE_NS_BATCH steps instead of enqueuing the worst-case count
// (round 2 enqueued 180 launches + a memset per retraction; near-orthonormal iterates need 6 - 9
// steps).
//   ctl[0] = 2 when C is not usable, ctl[2] = the scale s, ctl[8 + s] = max |I - Z Y| of step s
#define MDE_NS_MAX_STEPS 64
#define MDE_NS_BATCH 6
#define MDE_NS_TOL 1e-13
// 0: still iterating after steps [0, upto); 1: converged (at *where); 2: failed
__device__ __forceinline__ int ns_state(const double* __restrict__ ctl, int upto, int* where) {
  if (ctl[0] == 2.0) return 2;
  for (int s = 0; s < upto; ++s) {
    const double r = ctl[8 + s];
    if (r < MDE_NS_TOL) {
      if (where) *where = s;
      return 1;
    }
    if (!(r < 1e300)) return 2;
  }
  return 0;
}
// Y0 = sym(C) / s, Z0 = I with s = max_i sum_j |sym(C)_ij| >= lambda_max (Gershgorin): for the solver's
// near-orthonormal iterates C ~ n I the eigenvalues of Y0 start close to 1 and the iteration is over
// in 3 - 5 steps (the Frobenius norm used before starts them at 1 / sqrt(d): 9 - 10 steps).
// Every workgroup forms s itself (d^2 doubles out of L2) and initialises its own slice.
__global__ __launch_bounds__(MDE_BLOCK) void k_ns_init(int d, const double* __restrict__ C, double* __restrict__ Y,
                                                       double* __restrict__ Z, double* __restrict__ ctl) {
  __shared__ double smem[8];
  __shared__ double sh;
  const int m = d * d;
  // (column sums of |C|: coalesced; C is symmetric up to rounding, the symmetrised row sums differ in
  // the last bits, which the iteration does not mind)
  double mx = 0.0;
  for (int c = threadIdx.x; c < d; c += MDE_BLOCK) {
    double t = 0.0;
#pragma unroll 8
    for (int r = 0; r < d; ++r) t += fabs(C[r * d + c]);
    t = (t == t) ? t : 1e308;
    mx = t > mx ? t : mx;
  }
  const double tot = mde_block_max(mx, smem);
  if (threadIdx.x == 0) sh = tot;
  __syncthreads();
  const double s = sh;
  const bool ok = (s > 0.0) && (s < 1e300);
  for (int i = blockIdx.x * MDE_BLOCK + threadIdx.x; i < m; i += gridDim.x * MDE_BLOCK) {
    const int r = i / d, c = i % d;
    Y[i] = ok ? 0.5 * (C[i] + C[c * d + r]) / s : 0.0;  // symmetrise
    Z[i] = (r == c) ? 1.0 : 0.0;
  }
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < MDE_NS_MAX_STEPS; i += MDE_BLOCK) ctl[8 + i] = 0.0;
    if (threadIdx.x == 0) {
      ctl[0] = ok ? 0.0 : 2.0;
      ctl[2] = s;
    }
  }
}
// One 32 x 32 tile of P = A B (d x d doubles, row-major) per 256-thread workgroup: thread (ty, tx) owns
// the 2 x 2 outputs at rows 2 ty, columns 2 tx of the tile; the operands pass through LDS in 32-wide
// slabs of the inner index (row stride 33: no bank conflicts on the column reads).
__device__ __forceinline__ void ns_tile_product(int d, const double* __restrict__ A, const double* __restrict__ B,
                                                int ti, int tj, double (&o)[2][2]) {
  __shared__ double As[32][33], Bs[32][33];
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  o[0][0] = o[0][1] = o[1][0] = o[1][1] = 0.0;
  for (int k0 = 0; k0 < d; k0 += 32) {
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += MDE_BLOCK) {
      const int rr = e >> 5, cc = e & 31;
      const int ar = ti * 32 + rr, ac = k0 + cc, br = k0 + rr, bc = tj * 32 + cc;
      As[rr][cc] = (ar < d && ac < d) ? A[ar * d + ac] : 0.0;
      Bs[rr][cc] = (br < d && bc < d) ? B[br * d + bc] : 0.0;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      const double a0 = As[2 * ty][k], a1 = As[2 * ty + 1][k], b0 = Bs[k][2 * tx], b1 = Bs[k][2 * tx + 1];
      o[0][0] = fma(a0, b0, o[0][0]);
      o[0][1] = fma(a0, b1, o[0][1]);
      o[1][0] = fma(a1, b0, o[1][0]);
      o[1][1] = fma(a1, b1, o[1][1]);
    }
  }
}
// step: T = (3 I - Z Y) / 2 and the residual max |I - Z Y| (atomic max on the bit pattern of a
// non-negative double; the word starts at +0.0).  grid = (tiles, tiles)
__global__ __launch_bounds__(MDE_BLOCK) void k_ns_t(int d, int step, const double* __restrict__ Y,
                                                    const double* __restrict__ Z, double* __restrict__ T,
                                                    double* __restrict__ ctl) {
  if (ns_state(ctl, step, nullptr) != 0) return;
  __shared__ double smem[8];
  double o[2][2];
  ns_tile_product(d, Z, Y, blockIdx.y, blockIdx.x, o);
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  double emax = 0.0;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r = blockIdx.y * 32 + 2 * ty + a, c = blockIdx.x * 32 + 2 * tx + b;
      if (r < d && c < d) {
        const double e = ((r == c) ? 1.0 : 0.0) - o[a][b];
        T[r * d + c] = ((r == c) ? 1.0 : 0.0) + 0.5 * e;
        const double ae = (e == e) ? fabs(e) : 1e308;
        emax = ae > emax ? ae : emax;
      }
    }
  const double mx = mde_block_max(emax, smem);
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned long long*>(ctl + 8 + step), (unsigned long long)__double_as_longlong(mx));
}
// step: (Yn, Zn) = (Y T, T Z) unless the iteration ended with this step's residual.
// grid = (tiles, tiles, 2): z = 0 forms Y T, z = 1 forms T Z
__global__ __launch_bounds__(MDE_BLOCK) void k_ns_yz(int d, int step, const double* __restrict__ Y,
                                                     const double* __restrict__ Z, const double* __restrict__ T,
                                                     double* __restrict__ Yn, double* __restrict__ Zn,
                                                     const double* __restrict__ ctl) {
  if (ns_state(ctl, step + 1, nullptr) != 0) return;
  double o[2][2];
  const bool first = blockIdx.z == 0;
  ns_tile_product(d, first ? Y : T, first ? T : Z, blockIdx.y, blockIdx.x, o);
  double* out = first ? Yn : Zn;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int r = blockIdx.y * 32 + 2 * ty + a, c = blockIdx.x * 32 + 2 * tx + b;
      if (r < d && c < d) out[r * d + c] = o[a][b];
    }
}
// M = out_scale / sqrt(s) * Z of the step that converged (step s left its pair in buffer s & 1)
__global__ __launch_bounds__(MDE_BLOCK) void k_ns_finish(int d, int nsteps, const double* __restrict__ Z0,
                                                         const double* __restrict__ Z1, const double* __restrict__ ctl,
                                                         double out_scale, double* __restrict__ M,
                                                         int32_t* __restrict__ status) {
  int where = 0;
  const bool ok = ns_state(ctl, nsteps, &where) == 1;
  const double* Z = (where & 1) ? Z1 : Z0;
  const double k = ok ? out_scale / sqrt(ctl[2]) : 0.0;
  for (int i = blockIdx.x * MDE_BLOCK + threadIdx.x; i < d * d; i += gridDim.x * MDE_BLOCK) M[i] = ok ? Z[i] * k : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && status && !ok) *status = 1;
}

// M = out_scale * C^{-1/2}: the single-workgroup kernel for small d, the launch sequence above
// otherwise (SYNC for d >= 32: the residual words are read back after every batch of steps)
static int invsqrt_impl(int d, const double* C, double out_scale, double* M, double* scratch /* 5 d^2 + 8 + 64 */,
                        int32_t* status_dev, hipStream_t st) {
  if (d < 32) {
    hipLaunchKernelGGL(k_invsqrt, dim3(1), dim3(MDE_BLOCK), 0, st, d, C, out_scale, M, scratch, status_dev);
    MDE_LAUNCH_CHECK();
    return MDE_OK;
  }
  const int64_t m = (int64_t)d * d;
  double* Y[2] = {scratch, scratch + 3 * m};
  double* Z[2] = {scratch + m, scratch + 4 * m};
  double* T = scratch + 2 * m;
  double* ctl = scratch + 5 * m;         // 8 + MDE_NS_MAX_STEPS doubles
  hipLaunchKernelGGL(k_ns_init, dim3(mde_grid(m, MDE_BLOCK, 64)), dim3(MDE_BLOCK), 0, st, d, C, Y[0], Z[0], ctl);
  MDE_LAUNCH_CHECK();
  const unsigned nt = (unsigned)((d + 31) / 32);
  int steps = 0;
  double host[8 + MDE_NS_MAX_STEPS];
  bool done = false;
  while (!done && steps < MDE_NS_MAX_STEPS) {
    const int upto = std::min(MDE_NS_MAX_STEPS, steps + MDE_NS_BATCH);
    for (; steps < upto; ++steps) {
      const int h = steps & 1;
      hipLaunchKernelGGL(k_ns_t, dim3(nt, nt), dim3(MDE_BLOCK), 0, st, d, steps, Y[h], Z[h], T, ctl);
      hipLaunchKernelGGL(k_ns_yz, dim3(nt, nt, 2), dim3(MDE_BLOCK), 0, st, d, steps, Y[h], Z[h], T, Y[h ^ 1], Z[h ^ 1], ctl);
    }
    MDE_LAUNCH_CHECK();
    MDE_HIP(hipMemcpyAsync(host, ctl, sizeof(double) * (8 + (size_t)steps), hipMemcpyDeviceToHost, st));
    MDE_HIP(hipStreamSynchronize(st));
    if (host[0] == 2.0) done = true;
    for (int s = 0; s < steps && !done; ++s) done = (host[8 + s] < MDE_NS_TOL) || !(host[8 + s] < 1e300);
  }
  hipLaunchKernelGGL(k_ns_finish, dim3(mde_grid(m, MDE_BLOCK, 64)), dim3(MDE_BLOCK), 0, st, d, steps, Z[0], Z[1], ctl,
                     out_scale, M, status_dev);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// Z <- sqrt(n) (Z - mean) C^{-1/2},  C = (Z-mean)^T (Z-mean)          [ref: util.py:129-161]
// (= sqrt(n) U V^T of the thin SVD Z - mean = U S V^T, the reference's formula.)
// (X0, DIR, t given: Z <- X0 + t DIR first -- folded into the statistics pass at d <= 4)
static int std_retract_impl(int64_t n, int32_t d, float* Z, int32_t demean, double* work, int32_t* status_dev,
                            hipStream_t st, const float* X0 = nullptr, const float* DIR = nullptr, float t = 0.0f) {
  int rc = MDE_OK;
  if (DIR && d > 4) {
    rc = axpy_impl(n * (int64_t)d, t, DIR, X0, Z, st);
    if (rc != MDE_OK) return rc;
  }
  double* mats = work_mats(work);
  const int64_t m = (int64_t)d * d;
  double* C = mats;
  double* M = mats + m;
  double* scratch = mats + 2 * m;  // 5 m
  if (d <= 4) {
    const int nb = mde_grid(n, MDE_BLOCK * 2, MDE_RED_BLOCKS);
    const int nb2 = mde_grid(n, MDE_BLOCK, 2048);
    double* mean = work;  // small area
#define TINY(D_)                                                                                             \
  if (d == D_) {                                                                                             \
    hipLaunchKernelGGL(k_retract_stats_tiny<D_>, dim3(nb), dim3(MDE_BLOCK), 0, st, n, Z, X0, DIR, t,         \
                       (int)demean, sqrt((double)n), work_partials(work, d), mean, C, M, scratch, status_dev, \
                       work_ticket(work, TK_RETRACT));                                                       \
    MDE_LAUNCH_CHECK();                                                                                      \
    hipLaunchKernelGGL(k_center_rmul_tiny<D_>, dim3(nb2), dim3(MDE_BLOCK), 0, st, n, mean, M, Z);             \
    MDE_LAUNCH_CHECK();                                                                                      \
    return MDE_OK;                                                                                           \
  }
    TINY(1) TINY(2) TINY(3) TINY(4)
#undef TINY
  }
  if (mde_mfma_width_ok(d)) {
    // three reads and one write of Z: column means, the Gram matrix of the centred rows (the mean is
    // subtracted in registers), C^{-1/2}, and (Z - mean) M written in place
    double* mean = work;  // small area
    if (demean) {
      rc = mde_mfma_colmean(n, d, Z, work_partials(work, d), mean, st);
      if (rc != MDE_OK) return rc;
    }
    rc = mde_mfma_gram(n, d, Z, Z, demean ? mean : nullptr, C, work_partials(work, d), MDE_PARTIAL_DOUBLES, st);
    if (rc != MDE_OK) return rc;
    rc = invsqrt_impl(d, C, sqrt((double)n), M, scratch, status_dev, st);
    if (rc != MDE_OK) return rc;
    return mde_mfma_rmul(n, d, Z, M, demean ? mean : nullptr, 1.0f, nullptr, Z, st);
  }
  if (demean) {
    rc = center_impl(n, d, Z, work, st);
    if (rc != MDE_OK) return rc;
  }
  rc = gram_impl(n, d, d, Z, Z, C, work_partials(work, d), work_ticket(work, TK_GRAM), st);
  if (rc != MDE_OK) return rc;
  rc = invsqrt_impl(d, C, sqrt((double)n), M, scratch, status_dev, st);
  if (rc != MDE_OK) return rc;
  return rmul_impl(n, d, d, Z, M, 1.0f, nullptr, Z, st);
}

extern "C" int mde_std_retract(int64_t n, int32_t d, float* Z, int32_t demean, double* work,
                               int32_t* status_dev, void* stream) {
  if (n <= 0 || d <= 0 || d > 2048 || !Z || !work) return MDE_E_INVALID;
  return std_retract_impl(n, d, Z, demean, work, status_dev, mde_stream(stream));
}

// Z <- retract(X + t dir): the trial point of the line search under the Standardized constraint
extern "C" int mde_std_retract_step(int64_t n, int32_t d, const float* X, const float* dir, float t, float* Z,
                                    int32_t demean, double* work, int32_t* status_dev, void* stream) {
  if (n <= 0 || d <= 0 || d > 2048 || !X || !dir || !Z || !work) return MDE_E_INVALID;
  return std_retract_impl(n, d, Z, demean, work, status_dev, mde_stream(stream), X, dir, t);
}

// ---------------------------------------------------------------- L-BFGS memory
#define MDE_LB_GROUP 8
#define MDE_LB_NVAL (4 + 5 * MDE_LB_GROUP)
struct LbPtrs {
  const float* s[MDE_LB_GROUP];
  const float* y[MDE_LB_GROUP];
  float cs[MDE_LB_GROUP];
  float cy[MDE_LB_GROUP];
  int count;
};

// Device-resident bookkeeping of mde_lbfgs_dev_step: the Gram matrices of the stored pairs, the
// slot permutation and the direction coefficients never visit the host.
#define MDE_LB_MAX 63
#define MDE_LB_LD 64
struct LbDev {
  int count;                  // stored pairs
  int accepted;               // 1: the last staged pair was stored (y.s > 1e-10, lbfgs.py:472)
  int order[MDE_LB_LD];       // permutation of the history+1 slots: [0, count) oldest first, [count] = spare
  double H;                   // y.s / y.y of the newest pair (lbfgs.py:486)
  double SY[MDE_LB_LD * MDE_LB_LD];  // SY[i][j] = s_i . y_j
  double YY[MDE_LB_LD * MDE_LB_LD];  // YY[i][j] = y_i . y_j
  float c_g;                  // d = c_g g + sum_j cs[j] s_j + cy[j] y_j
  float cs[MDE_LB_LD];
  float cy[MDE_LB_LD];
};

struct mde_lbfgs {
  int64_t N = 0;
  int history = 0;
  float* buf = nullptr;  // 2 (history+1) N floats
  int order[64];         // slot ids, oldest first
  int count = 0;
  int spare = 0;
  bool staged = false;
  struct LbDev* dev = nullptr;  // device-resident bookkeeping of the device-driven step
  float* S(int slot) const { return buf + (int64_t)(2 * slot) * N; }
  float* Y(int slot) const { return buf + (int64_t)(2 * slot + 1) * N; }
};

extern "C" int mde_lbfgs_create(int64_t N, int32_t history, mde_lbfgs** out) {
  if (!out || N <= 0 || history <= 0 || history > 63) return MDE_E_INVALID;
  mde_lbfgs* o = new mde_lbfgs();
  o->N = N;
  o->history = history;
  hipError_t e = hipMalloc(&o->buf, sizeof(float) * 2 * (size_t)(history + 1) * (size_t)N);
  if (e != hipSuccess) {
    delete o;
    return mde_hip_fail(e, "hipMalloc(lbfgs history)", __FILE__, __LINE__);
  }
  o->count = 0;
  o->spare = 0;
  e = hipMalloc(&o->dev, sizeof(LbDev));
  if (e != hipSuccess) {
    (void)hipFree(o->buf);
    delete o;
    return mde_hip_fail(e, "hipMalloc(lbfgs device state)", __FILE__, __LINE__);
  }
  *out = o;
  return MDE_OK;
}
extern "C" int mde_lbfgs_destroy(mde_lbfgs* o) {
  if (!o) return MDE_OK;
  if (o->buf) (void)hipFree(o->buf);
  if (o->dev) (void)hipFree(o->dev);
  delete o;
  return MDE_OK;
}
extern "C" int mde_lbfgs_reset(mde_lbfgs* o) {
  if (!o) return MDE_E_INVALID;
  o->count = 0;
  o->spare = 0;
  o->staged = false;
  return MDE_OK;
}
extern "C" int32_t mde_lbfgs_count(const mde_lbfgs* o) { return o ? o->count : 0; }

// first launch (FIRST): forms y* = g - g_prev, s* = t d, writes them to the spare slot, sets
// g_prev <- g, and accumulates the 4 base dots; every launch accumulates, for its <= 8 stored
// pairs, the 5 dots against (y*, s*, g).  partial[q * nb + b].
template <bool FIRST>
__global__ __launch_bounds__(MDE_BLOCK) void k_lbfgs_stage(int64_t N, const float* __restrict__ g,
                                                           float* __restrict__ g_prev,
                                                           const float* __restrict__ d, float t,
                                                           float* __restrict__ s_new,
                                                           float* __restrict__ y_new, LbPtrs P,
                                                           int qbase, double* __restrict__ partial) {
  __shared__ double smem[8];
  double base[4] = {0, 0, 0, 0};
  double acc[MDE_LB_GROUP][5];
#pragma unroll
  for (int j = 0; j < MDE_LB_GROUP; ++j)
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[j][q] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    const float gv = g[i];
    float yv, sv;
    if constexpr (FIRST) {
      yv = gv - g_prev[i];
      sv = t * d[i];
      y_new[i] = yv;
      s_new[i] = sv;
      g_prev[i] = gv;
      base[0] = fma((double)yv, (double)sv, base[0]);
      base[1] = fma((double)yv, (double)yv, base[1]);
      base[2] = fma((double)sv, (double)gv, base[2]);
      base[3] = fma((double)yv, (double)gv, base[3]);
    } else {
      yv = y_new[i];
      sv = s_new[i];
    }
#pragma unroll
    for (int j = 0; j < MDE_LB_GROUP; ++j) {
      if (j < P.count) {
        const double sj = P.s[j][i], yj = P.y[j][i];
        acc[j][0] = fma(sj, (double)yv, acc[j][0]);  // s_j . y*
        acc[j][1] = fma(yj, (double)yv, acc[j][1]);  // y_j . y*
        acc[j][2] = fma((double)sv, yj, acc[j][2]);  // s* . y_j
        acc[j][3] = fma(sj, (double)gv, acc[j][3]);  // s_j . g
        acc[j][4] = fma(yj, (double)gv, acc[j][4]);  // y_j . g
      }
    }
  }
  const int nb = gridDim.x, b = blockIdx.x;
  if constexpr (FIRST) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double r = mde_block_sum(base[q], smem);
      if (threadIdx.x == 0) partial[(int64_t)q * nb + b] = r;
    }
  }
#pragma unroll
  for (int j = 0; j < MDE_LB_GROUP; ++j) {
    if (j < P.count) {
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const double r = mde_block_sum(acc[j][q], smem);
        if (threadIdx.x == 0) partial[(int64_t)(qbase + 5 * j + q) * nb + b] = r;
      }
    }
  }
}

extern "C" int mde_lbfgs_stage(mde_lbfgs* o, const float* g, float* g_prev, const float* d, float t,
                               double* dots, double* work, void* stream) {
  if (!o || !g || !g_prev || !d || !dots || !work) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  const int64_t N = o->N;
  const int nb = mde_grid(N, MDE_BLOCK * 2, 2048);  // >= 8 waves per SIMD-quad in flight: ~20 streams to hide
  double* partial = work + MDE_SMALL_DOUBLES;
  float* s_new = o->S(o->spare);
  float* y_new = o->Y(o->spare);
  int done = 0;
  bool first = true;
  do {
    LbPtrs P;
    P.count = o->count - done;
    if (P.count > MDE_LB_GROUP) P.count = MDE_LB_GROUP;
    for (int j = 0; j < MDE_LB_GROUP; ++j) {
      const int slot = (j < P.count) ? o->order[done + j] : 0;
      P.s[j] = o->S(slot);
      P.y[j] = o->Y(slot);
      P.cs[j] = P.cy[j] = 0.f;
    }
    const int qbase = 4 + 5 * done;
    if (first)
      hipLaunchKernelGGL((k_lbfgs_stage<true>), dim3(nb), dim3(MDE_BLOCK), 0, st, N, g, g_prev, d, t,
                         s_new, y_new, P, qbase, partial);
    else
      hipLaunchKernelGGL((k_lbfgs_stage<false>), dim3(nb), dim3(MDE_BLOCK), 0, st, N, g, g_prev, d, t,
                         s_new, y_new, P, qbase, partial);
    MDE_LAUNCH_CHECK();
    first = false;
    done += P.count;
  } while (done < o->count);
  const int nq = 4 + 5 * o->count;
  hipLaunchKernelGGL(k_reduce_rows, dim3(nq), dim3(MDE_BLOCK), 0, st, nq, nb, partial, 0ull, dots);
  MDE_LAUNCH_CHECK();
  o->staged = true;
  return MDE_OK;
}

extern "C" int mde_lbfgs_commit(mde_lbfgs* o, int32_t accept) {
  if (!o) return MDE_E_INVALID;
  if (!o->staged) {
    mde_set_error("mde_lbfgs_commit without a staged pair");
    return MDE_E_INVALID;
  }
  o->staged = false;
  if (!accept) return MDE_OK;
  const int new_slot = o->spare;
  if (o->count == o->history) {
    // drop the oldest: its slot becomes the spare   (lbfgs.py:474-478)
    o->spare = o->order[0];
    for (int i = 1; i < o->count; ++i) o->order[i - 1] = o->order[i];
    o->order[o->count - 1] = new_slot;
  } else {
    o->order[o->count++] = new_slot;
    // next unused slot id
    bool used[64] = {false};
    for (int i = 0; i < o->count; ++i) used[o->order[i]] = true;
    int sp = 0;
    while (used[sp]) ++sp;
    o->spare = sp;
  }
  return MDE_OK;
}

// ---------------------------------------------------------------- device-driven step
// The same history update and two-loop recursion with every decision taken on the device: the
// stage kernel finds its slots through LbDev, k_lbfgs_direction folds the staged dots into the Gram
// matrices (accept / drop oldest, lbfgs.py:472-486) and runs the recursion in coefficient form
// (lbfgs.py:490-507), the combine kernel reads the coefficients from LbDev.  No host read-back.
__global__ void k_lbfgs_dev_reset(LbDev* __restrict__ dv, int history) {
  const int t = threadIdx.x;
  if (t <= history) dv->order[t] = t;
  if (t == 0) {
    dv->count = 0;
    dv->accepted = 0;
    dv->H = 1.0;
    dv->c_g = -1.0f;
  }
}

// Sums over the 64 lanes of NV values at once: lane q < NV returns the total of val[q].  A butterfly
// in which every lane keeps half of its values per step (63 shuffles for up to 64 values; NV
// separate wave sums are 6 NV of them, and behind a run-time `q < nval` each is a serial chain of six
// cross-lane latencies -- 15 us of the 28 the staging kernel took at N = 140k).
template <int NV>
__device__ __forceinline__ double lb_transpose_sum(const double (&val)[NV]) {
  static_assert(NV <= 64, "one value per lane at most");
  double a[64];
#pragma unroll
  for (int k = 0; k < 64; ++k) a[k] = k < NV ? val[k] : 0.0;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const bool hi = (lane & o) != 0;
#pragma unroll
    for (int k = 0; k < o; ++k) {
      if (k >= NV) continue;  // both halves are zero in every lane
      const double keep = hi ? a[k + o] : a[k];
      const double send = hi ? a[k] : a[k + o];
      a[k] = keep + __shfl_xor(send, o, 64);
    }
  }
  return a[0];
}

// The staging pass of the device-driven step over the elements i = b * MDE_BLOCK + thread (+ k nb
// MDE_BLOCK): the new pair y = g - g_prev, s = t d is written to the spare slot, g_prev <- g, and the
// workgroup's share of every dot product the history update needs goes to partial[row * nb + b]
// (rows 0..3: y*.s*, y*.y*, s*.g, y*.g; 4 + 5 j + k: pair j: s_j.y*, y_j.y*, s*.y_j, s_j.g, y_j.g),
// G stored pairs per pass (8; 12 when the history is 9..12 pairs: one pass instead of two, whose second
// would read the vectors again for the last few pairs).  PUBLISH: relaxed device-scope stores (read again
// inside the launch).
template <bool PUBLISH, int G = MDE_LB_GROUP>
__device__ __forceinline__ void lb_stage_phase(int64_t N, const float* __restrict__ g, float* __restrict__ g_prev,
                                               const float* __restrict__ d, float t, float* __restrict__ buf,
                                               const LbDev* __restrict__ dv, double* __restrict__ partial,
                                               double (*sm)[4 + 5 * G]) {
  constexpr int NVAL = 4 + 5 * G;
  const int count = dv->count;
  const int spare = dv->order[count];
  float* s_new = buf + (int64_t)(2 * spare) * N;
  float* y_new = buf + (int64_t)(2 * spare + 1) * N;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nb = gridDim.x, b = blockIdx.x;
  for (int done = 0; done == 0 || done < count; done += G) {
    const bool first = done == 0;
    int pc = count - done;
    if (pc > G) pc = G;
    const float* ps[G];
    const float* py[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int slot = (j < pc) ? dv->order[done + j] : spare;
      ps[j] = buf + (int64_t)(2 * slot) * N;
      py[j] = buf + (int64_t)(2 * slot + 1) * N;
    }
    double val[NVAL];
#pragma unroll
    for (int q = 0; q < NVAL; ++q) val[q] = 0.0;
    for (int64_t i = (int64_t)b * MDE_BLOCK + threadIdx.x; i < N; i += (int64_t)nb * MDE_BLOCK) {
      const float gv = g[i];
      float yv, sv;
      if (first) {
        yv = gv - g_prev[i];
        sv = t * d[i];
        y_new[i] = yv;
        s_new[i] = sv;
        g_prev[i] = gv;
        val[0] = fma((double)yv, (double)sv, val[0]);
        val[1] = fma((double)yv, (double)yv, val[1]);
        val[2] = fma((double)sv, (double)gv, val[2]);
        val[3] = fma((double)yv, (double)gv, val[3]);
      } else {
        yv = y_new[i];
        sv = s_new[i];
      }
      // all sixteen loads before the first use (slots beyond pc alias the spare pair: valid memory,
      // their sums are never written) -- a branch per pair would serialise sixteen memory latencies
      float sj[G], yj[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        sj[j] = ps[j][i];
        yj[j] = py[j][i];
      }
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const double sd = sj[j], yd = yj[j];
        val[4 + 5 * j + 0] = fma(sd, (double)yv, val[4 + 5 * j + 0]);  // s_j . y*
        val[4 + 5 * j + 1] = fma(yd, (double)yv, val[4 + 5 * j + 1]);  // y_j . y*
        val[4 + 5 * j + 2] = fma((double)sv, yd, val[4 + 5 * j + 2]);  // s* . y_j
        val[4 + 5 * j + 3] = fma(sd, (double)gv, val[4 + 5 * j + 3]);  // s_j . g
        val[4 + 5 * j + 4] = fma(yd, (double)gv, val[4 + 5 * j + 4]);  // y_j . g
      }
    }
    // lane q holds the wave's total of value q -> LDS -> thread q adds the four waves
    const int nval = 4 + 5 * pc;
    const double tot = lb_transpose_sum<NVAL>(val);
    if (lane < NVAL) sm[wave][lane] = tot;
    __syncthreads();
    if ((int)threadIdx.x < nval && (first || threadIdx.x >= 4)) {
      const int q = threadIdx.x;
      double r = 0.0;
#pragma unroll
      for (int w = 0; w < MDE_BLOCK / 64; ++w) r += sm[w][q];
      const int row = q < 4 ? q : 4 + 5 * done + (q - 4);
      if (PUBLISH)
        mde_st_partial(partial + (int64_t)row * nb + b, r);
      else
        partial[(int64_t)row * nb + b] = r;
    }
    __syncthreads();
  }
}

// Results of the direction step, kept in LDS until they are written back to LbDev.
struct LbOut {
  int m, accepted;
  double H;
  float c_g;
  float cs[MDE_LB_LD], cy[MDE_LB_LD];
  int order[MDE_LB_LD];
  double Sg[MDE_LB_LD], Yg[MDE_LB_LD], cq[MDE_LB_LD], al[MDE_LB_LD], cr[MDE_LB_LD];  // scratch of the recursion
};

// LDS written by some lanes of ONE wave and read by others: the hardware executes a wave's LDS
// accesses in order, the compiler must not move them across this point (no workgroup barrier: in the
// fused kernel the other waves of the block do not take part).
__device__ __forceinline__ void lb_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// One wave.  Lane j owns column j of everything; the Gram matrices (SY, YY: LD x LD doubles of LDS,
// LD > history) are edited in place, the coefficients of the new direction go to *o.  `dots` as
// k_lb_reduce leaves them.  Nothing of LbDev is written.
template <int LD>
__device__ void lb_direction_core(const LbDev* __restrict__ dv, const double* dots, int history, double* SY,
                                  double* YY, LbOut* o) {
  const int j = threadIdx.x & 63;
  const int c = dv->count;
  {
    // the stored Gram matrices -> LDS, eight row groups per batch with all their loads in flight
    // together (a loop of one row per trip pays one memory latency per row: 10 us at ten pairs)
    constexpr int RP = 64 / (LD < 64 ? LD : 64);  // rows per pass of the 64 lanes
    const int col = j % LD, rsub = j / LD;
    for (int i0 = 0; i0 < c; i0 += 8 * RP) {
      double vs[8], vy[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * RP + rsub;
        const bool in = i < c && col < c;
        vs[u] = in ? dv->SY[i * MDE_LB_LD + col] : 0.0;
        vy[u] = in ? dv->YY[i * MDE_LB_LD + col] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * RP + rsub;
        if (i < c) {
          SY[i * LD + col] = vs[u];
          YY[i * LD + col] = vy[u];
        }
      }
    }
  }
  const double ys = dots[0], yy = dots[1], sg_new = dots[2], yg_new = dots[3];
  // per stored pair j: s_j.y*, y_j.y*, s*.y_j, s_j.g, y_j.g
  double s_old_ynew = 0, y_old_ynew = 0, snew_y_old = 0, Sg = 0, Yg = 0;
  if (j < c) {
    s_old_ynew = dots[4 + 5 * j + 0];
    y_old_ynew = dots[4 + 5 * j + 1];
    snew_y_old = dots[4 + 5 * j + 2];
    Sg = dots[4 + 5 * j + 3];
    Yg = dots[4 + 5 * j + 4];
  }
  const bool accepted = ys > 1e-10;  // lbfgs.py:472
  int m = c;
  double H = dv->H;
  int ord = (j <= history) ? dv->order[j] : 0;
  lb_wave_sync();
  if (accepted) {
    // append row / column c
    if (j < c) {
      SY[j * LD + c] = s_old_ynew;
      SY[c * LD + j] = snew_y_old;
      YY[j * LD + c] = y_old_ynew;
      YY[c * LD + j] = y_old_ynew;
    }
    if (j == c) {
      SY[c * LD + c] = ys;
      YY[c * LD + c] = yy;
      Sg = sg_new;
      Yg = yg_new;
    }
    lb_wave_sync();
    m = c + 1;
    H = ys / yy;  // lbfgs.py:486
    if (c == history) {
      // drop the oldest pair (lbfgs.py:474-478): shift everything up / left by one
      // (in place, rows ascending: row i is read before row i - 1 is written, and a wave's LDS
      // accesses execute in program order; per-lane row buffers would live in scratch memory --
      // twenty dependent memory round trips, most of this kernel's 12 us)
      for (int i = 1; i < m; ++i) {
        const double a = (j + 1 < m && j + 1 < LD) ? SY[i * LD + j + 1] : 0.0;
        const double y = (j + 1 < m && j + 1 < LD) ? YY[i * LD + j + 1] : 0.0;
        if (j < LD) {
          SY[(i - 1) * LD + j] = a;
          YY[(i - 1) * LD + j] = y;
        }
      }
      Sg = __shfl_down(Sg, 1, 64);
      Yg = __shfl_down(Yg, 1, 64);
      m = history;
      lb_wave_sync();
    }
    // slot permutation: the spare becomes the newest pair; when full, the oldest slot is the new spare
    if (c == history) {
      const int first = __shfl(ord, 0, 64);
      const int next = __shfl_down(ord, 1, 64);
      ord = (j < history) ? next : first;
    }
  }
  double my_cq = 0.0, my_cr = 0.0;
  if constexpr (LD <= 16) {
    // Short histories (the default memory is 10): lane i keeps row i and column i of SY and row i of
    // YY in registers; both loops are substitutions in which lane k finishes its coefficient, hands it
    // to the others with v_readlane and every later lane folds it into its own running sum -- LD
    // fully unrolled steps of a readlane and one FMA, no LDS and no cross-lane sums (the two loops
    // take 1 us; one wave sum per step took 9, the same recursion serially from LDS 9 as well).
    auto bcast = [](double v, int k) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), k), hi = __builtin_amdgcn_readlane(__double2hiint(v), k);
      return __hiloint2double(hi, lo);
    };
    const int li = j < LD ? j : 0;
    double rS[LD], cS[LD], rY[LD];
#pragma unroll
    for (int k = 0; k < LD; ++k) {
      rS[k] = SY[li * LD + k];
      cS[k] = SY[k * LD + li];
      rY[k] = YY[li * LD + k];
    }
    const double rho = (j < m) ? 1.0 / SY[li * LD + li] : 0.0;
    double acc = -Sg, al = 0.0, cq = 0.0;
#pragma unroll
    for (int k = LD - 1; k >= 0; --k) {
      const double a_k = bcast(rho * acc, k);
      if (k < m) {
        if (j == k) {
          al = a_k;
          cq = -a_k;
        }
        if (j < k) acc = fma(-a_k, rS[k], acc);
      }
    }
    double yq = -Yg;
#pragma unroll
    for (int k = 0; k < LD; ++k) {
      const double cqk = bcast(cq, k);
      if (k < m) yq = fma(cqk, rY[k], yq);
    }
    double acc2 = H * yq, cr = 0.0;
#pragma unroll
    for (int k = 0; k < LD; ++k) {
      const double c_k = bcast(al - rho * acc2, k);
      if (k < m) {
        if (j == k) cr = c_k;
        if (j > k) acc2 = fma(c_k, cS[k], acc2);
      }
    }
    my_cq = cq;
    my_cr = cr;
  } else {
    // two-loop recursion in coefficient form: q = -g + sum cq_j y_j ; r = H q + sum cr_j s_j.  m is a
    // dozen: every lane runs the SAME serial recursion on the LDS copies (uniform addresses: broadcast
    // reads) -- ~m^2 dependent fp64 FMAs, 2 us; one cross-lane sum per step was 20 serial round trips
    // of six shuffles each, 9 of the kernel's 12 us.
    if (j < LD) {
      o->Sg[j] = Sg;
      o->Yg[j] = Yg;
    }
    lb_wave_sync();
    for (int i = m - 1; i >= 0; --i) {
      double a0 = -o->Sg[i], a1 = 0.0;
      int k = i + 1;
      for (; k + 1 < m; k += 2) {
        a0 = fma(o->cq[k], SY[i * LD + k], a0);
        a1 = fma(o->cq[k + 1], SY[i * LD + k + 1], a1);
      }
      if (k < m) a0 = fma(o->cq[k], SY[i * LD + k], a0);
      const double a = (a0 + a1) / SY[i * LD + i];
      lb_wave_sync();
      if (j == 0) {
        o->al[i] = a;
        o->cq[i] = -a;
      }
      lb_wave_sync();
    }
    for (int i = 0; i < m; ++i) {
      double q0 = -o->Yg[i], q1 = 0.0;
      int k = 0;
      for (; k + 1 < m; k += 2) {
        q0 = fma(o->cq[k], YY[i * LD + k], q0);
        q1 = fma(o->cq[k + 1], YY[i * LD + k + 1], q1);
      }
      if (k < m) q0 = fma(o->cq[k], YY[i * LD + k], q0);
      double r0 = H * (q0 + q1), r1 = 0.0;
      k = 0;
      for (; k + 1 < i; k += 2) {
        r0 = fma(o->cr[k], SY[k * LD + i], r0);
        r1 = fma(o->cr[k + 1], SY[(k + 1) * LD + i], r1);
      }
      if (k < i) r0 = fma(o->cr[k], SY[k * LD + i], r0);
      const double c = o->al[i] - (r0 + r1) / SY[i * LD + i];
      lb_wave_sync();
      if (j == 0) o->cr[i] = c;
      lb_wave_sync();
      if (j == i) {
        my_cr = c;
        my_cq = o->cq[i];
      }
    }
  }
  o->cs[j] = (j < m) ? (float)my_cr : 0.0f;
  o->cy[j] = (j < m) ? (float)(H * my_cq) : 0.0f;
  o->order[j] = ord;
  if (j == 0) {
    o->m = m;
    o->accepted = accepted ? 1 : 0;
    o->H = H;
    o->c_g = (float)(-H);
  }
  lb_wave_sync();
}

// One wave: *o and the edited Gram matrices -> LbDev.
template <int LD>
__device__ void lb_write_back(LbDev* __restrict__ dv, int history, const double* SY, const double* YY,
                              const LbOut* o) {
  const int j = threadIdx.x & 63;
  const int m = o->m;
  if (j < m) {
    dv->cs[j] = o->cs[j];
    dv->cy[j] = o->cy[j];
  }
  if (o->accepted) {
    for (int i = 0; i < m; ++i) {
      if (j < m) {
        dv->SY[i * MDE_LB_LD + j] = SY[i * LD + j];
        dv->YY[i * MDE_LB_LD + j] = YY[i * LD + j];
      }
    }
    if (j <= history) dv->order[j] = o->order[j];
  }
  if (j == 0) {
    dv->count = m;
    dv->accepted = o->accepted;
    dv->H = o->H;
    dv->c_g = o->c_g;
  }
}

#define MDE_LB_DIR_LDS(LD) (2 * (LD) * (LD) * sizeof(double) + sizeof(LbOut))
template <int LD>
__global__ __launch_bounds__(64) void k_lbfgs_direction(LbDev* __restrict__ dv, const double* __restrict__ dots,
                                                        int history, MdeGate gate) {
  if (mde_gate_closed(gate)) return;
  extern __shared__ __attribute__((aligned(16))) char lb_lds[];
  double* SY = reinterpret_cast<double*>(lb_lds);
  double* YY = SY + LD * LD;
  LbOut* out = reinterpret_cast<LbOut*>(YY + LD * LD);
  lb_direction_core<LD>(dv, dots, history, SY, YY, out);
  lb_write_back<LD>(dv, history, SY, YY, out);
}

// ---- the same step in ONE launch, for vectors small enough that launches, not bytes, are what an
// iteration costs (configs 2 and 3: N = 140k / 80k floats; every kernel boundary is ~5 us on this
// stack and the four launches below took 60 us of a 150 us iteration).  At most 128 workgroups, all
// resident at once, in phases separated by a grid-wide arrival counter:
//   1. stage the new pair and form the partial dot products of the workgroup's own elements;
//   -- every workgroup has published its partials (relaxed device-scope stores, as mde_last_block) --
//   2. workgroup q adds row q of the partials and publishes the dot product; -- arrival counter --
//      EVERY workgroup then runs the direction step on the same numbers in its own LDS: identical
//      coefficients everywhere, nothing to broadcast;
//   3. combine the new direction over the workgroup's own elements (it re-reads only what it wrote
//      itself in phase 1), publish the statistics partials; the last workgroup to arrive reduces
//      them and writes the bookkeeping back to LbDev (everyone has finished reading it by then).
#define MDE_LB_FUSED_LD 16       // history <= 15
#ifndef MDE_LB_FUSED_MAXN
#define MDE_LB_FUSED_MAXN (1 << 18)
#endif
#define MDE_LB_FUSED_MAXBLOCKS 512
// the flag words of the fused kernel: doubles [MDE_WS_LB_FLAGS, MDE_WS_LB_VERDICT) of the work buffer's small area
// (three rows of MDE_LB_FUSED_MAXBLOCKS 32-bit words; zero or an older epoch between launches), the verdict
// words right behind them
static_assert(MDE_WS_LB_FLAGS + 3 * MDE_LB_FUSED_MAXBLOCKS / 2 == MDE_WS_LB_VERDICT, "k_lb_fused: flag rows and verdict words");
static inline unsigned int* work_lb_flags(double* work) { return reinterpret_cast<unsigned int*>(work + MDE_WS_LB_FLAGS); }
// (design probe, -DMDE_LB_PROBE: workgroup 0 leaves wall-clock stamps -- 100 MHz -- of its phases behind the
// verdict words; tools/lbfused_probe.py prints them)
#ifdef MDE_LB_PROBE
#define LB_STAMP(k)                                                                                    \
  do {                                                                                                 \
    if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(verdict + 8)[k] = wall_clock64(); \
  } while (0)
#else
#define LB_STAMP(k) ((void)0)
#endif
// G: history pairs per pass over the vectors (8, or 12 when the history has 9 .. 12 pairs: ONE pass in both
// streaming phases instead of two -- the phases are bound by the latency of their dependent trips)
template <int G>
__global__ __launch_bounds__(MDE_BLOCK) void k_lb_fused(int64_t N, const float* __restrict__ g, float* __restrict__ g_prev,
                                                        const float* d, float t, float* __restrict__ buf,
                                                        LbDev* __restrict__ dv, int history, float* out,  // (out may be d)
                                                        double* __restrict__ partial, double* __restrict__ stats,
                                                        unsigned int* __restrict__ flags, unsigned int epoch,
                                                        unsigned int spin_limit, MdeGate gate) {
  if (mde_gate_closed(gate)) return;  // (every workgroup reads the same word: all of them leave, or none)
  constexpr int LD = MDE_LB_FUSED_LD;
  extern __shared__ char lb_debug_lds[];  // (only a test asks for dynamic LDS: it limits the workgroups per CU)
  (void)lb_debug_lds;
  // words behind the three flag rows: [0] = epoch once some workgroup gave up waiting, [1] = epoch once
  // the step is complete (k_lb_rescue redoes the step from the staged sums when it is not)
  unsigned int* verdict = flags + 3 * MDE_LB_FUSED_MAXBLOCKS;
  __shared__ double sm[MDE_BLOCK / 64][4 + 5 * G];
  __shared__ double s_dots[4 + 5 * LD];
  __shared__ double s_SY[LD * LD], s_YY[LD * LD];
  __shared__ LbOut s_out;
  const int count = dv->count;
  const int wave = threadIdx.x >> 6;
  const int nb = gridDim.x, b = blockIdx.x;
  LB_STAMP(0);
  // ---- phase 1
  lb_stage_phase<true, G>(N, g, g_prev, d, t, buf, dv, partial, sm);
  LB_STAMP(1);
  // Arrival point `which`: every workgroup raises its own flag word to this launch's epoch (plain
  // device-scope stores to distinct addresses; an arrival COUNTER costs ~20 ns per workgroup because
  // same-address atomics are executed one after the other at the memory side -- 5.6 us for 274
  // workgroups, three times per launch), the waiting workgroups read all flags, one per thread.
  auto grid_arrive = [&](int which, bool wait) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this thread's published values have completed
    __syncthreads();
    unsigned int* f = flags + which * MDE_LB_FUSED_MAXBLOCKS;
    if (threadIdx.x == 0) __hip_atomic_store(f + b, epoch + (unsigned int)which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!wait) return true;
    // The arrival points need every workgroup of the launch resident at once (an ordinary launch of at
    // most 128 workgroups: normally true).  A workgroup that does not see the others arrive within the
    // spin limit -- another process holds the CUs -- says so in verdict[0] and LEAVES: nobody gets past
    // a later arrival point then, nothing of the step's outputs is final, and k_lb_rescue (queued
    // behind this kernel) redoes the step from the staged sums.  Never a wrong direction, never a hang.
    for (unsigned int spins = 0;; ++spins) {
      bool ok = true;
      for (int k = threadIdx.x; k < nb; k += MDE_BLOCK)
        ok = ok && __hip_atomic_load(f + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch + (unsigned int)which;
      if (__syncthreads_and(ok ? 1 : 0)) break;
      const bool failed = __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
      if (__syncthreads_or((failed || spins >= spin_limit) ? 1 : 0)) {
        if (threadIdx.x == 0) __hip_atomic_store(verdict, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    return __hip_atomic_load(verdict, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch;
  };
  if (!grid_arrive(0, true)) return;
  LB_STAMP(2);
  // ---- phase 2a: workgroup q adds row q of the partials (k_lb_reduce: same order of additions)
  const int nrows = 4 + 5 * count;
  double* dots_g = partial + (int64_t)(4 + 5 * LD + 8) * nb;  // behind the dot-product and statistics rows
  for (int q = b; q < nrows; q += nb) {
    double r = 0.0;
    for (int k = threadIdx.x; k < nb; k += MDE_BLOCK) r += mde_ld_partial(partial + (int64_t)q * nb + k);
    const double tot = mde_block_sum(r, &sm[0][0]);
    if (threadIdx.x == 0) mde_st_partial(dots_g + q, tot);
    __syncthreads();
  }
  LB_STAMP(3);
  if (!grid_arrive(1, true)) return;
  LB_STAMP(4);
  // ---- phase 2b: every workgroup runs the direction step on the same numbers
  for (int q = threadIdx.x; q < nrows; q += MDE_BLOCK) s_dots[q] = mde_ld_partial(dots_g + q);
  __syncthreads();
  if (wave == 0) lb_direction_core<LD>(dv, s_dots, history, s_SY, s_YY, &s_out);
  __syncthreads();
  LB_STAMP(5);
  // ---- phase 3 (k_lb_combine_all), with the coefficients and the slot order of s_out
  const int m = s_out.m;
  const float c_g = s_out.c_g;
  double gd = 0, gg = 0, g1 = 0, gm = 0, nf = 0, dd = 0, dm = 0;
  for (int64_t i = (int64_t)b * MDE_BLOCK + threadIdx.x; i < N; i += (int64_t)nb * MDE_BLOCK) {
    const float gv = g[i];
    float v = c_g * gv;
    for (int j0 = 0; j0 < m; j0 += G) {
      float sv[G], yv[G];
#pragma unroll
      for (int j = 0; j < G; ++j) {
        const int jj = (j0 + j < m) ? j0 + j : 0;
        const float* sp = buf + (int64_t)(2 * s_out.order[jj]) * N;
        sv[j] = sp[i];
        yv[j] = sp[N + i];
      }
#pragma unroll
      for (int j = 0; j < G; ++j)
        if (j0 + j < m) v = fmaf(s_out.cy[j0 + j], yv[j], fmaf(s_out.cs[j0 + j], sv[j], v));
    }
    out[i] = v;
    const double gvd = gv, dv2 = v;
    gg += gvd * gvd;
    const double ag = fabs(gvd);
    g1 += ag;
    gm = ag > gm ? ag : gm;
    nf += (fabsf(gv) <= 3.402823466e+38f) ? 0.0 : 1.0;
    gd += gvd * dv2;
    dd += dv2 * dv2;
    const double ad = fabs(dv2);
    dm = ad > dm ? ad : dm;
  }
  const double v8[8] = {gd, gg, g1, gm, nf, dd, dm, 0.0};
  double* spart = partial + (int64_t)(4 + 5 * LD) * nb;  // behind the dot-product rows
  mde_publish8(v8, (1u << 3) | (1u << 6), spart, nb, b);
  LB_STAMP(6);
  // workgroup 0 finishes: the statistics rows, and the bookkeeping back to LbDev (everyone has
  // finished reading it when the third flag is up)
  if (!grid_arrive(2, b == 0)) return;
  if (b != 0) return;
  LB_STAMP(7);
  mde_final_rows(8, nb, spart, stats, (1ull << 3) | (1ull << 6));
  if (wave == 0) lb_write_back<LD>(dv, history, s_SY, s_YY, &s_out);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(verdict + 1, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  LB_STAMP(8);
}

// Queued behind every k_lb_fused launch: nothing to do when that launch completed (verdict[1] ==
// epoch; one workgroup, a few microseconds).  Otherwise -- some workgroup gave up at an arrival point --
// ONE workgroup redoes phases 2 and 3 from the sums every workgroup staged in phase 1 (phase 1 has no
// waits: all of them ran it; its side effects -- g_prev <- g, the new pair in its slot -- are exactly
// what the four-launch path leaves behind its first launch), slowly but correctly: row sums, the
// direction step, d_out and its statistics, the bookkeeping.
__global__ __launch_bounds__(MDE_BLOCK) void k_lb_rescue(int64_t N, const float* __restrict__ g, const float* __restrict__ buf,
                                                         LbDev* __restrict__ dv, int history, float* out, int nb,
                                                         double* __restrict__ partial, double* __restrict__ stats,
                                                         unsigned int* __restrict__ flags, unsigned int epoch, MdeGate gate) {
  constexpr int LD = MDE_LB_FUSED_LD;
  if (mde_gate_closed(gate)) return;
  unsigned int* verdict = flags + 3 * MDE_LB_FUSED_MAXBLOCKS;
  if (__hip_atomic_load(verdict + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return;
  if (threadIdx.x == 0) verdict[2] += 1u;  // (how often the rescue had to run: tests read it from the work buffer)
  __shared__ double sm[MDE_BLOCK / 64][MDE_LB_NVAL];
  __shared__ double s_dots[4 + 5 * LD];
  __shared__ double s_SY[LD * LD], s_YY[LD * LD];
  __shared__ LbOut s_out;
  const int wave = threadIdx.x >> 6;
  const int nrows = 4 + 5 * dv->count;
  for (int q = 0; q < nrows; ++q) {
    double r = 0.0;
    for (int k = threadIdx.x; k < nb; k += MDE_BLOCK) r += mde_ld_partial(partial + (int64_t)q * nb + k);
    const double tot = mde_block_sum(r, &sm[0][0]);
    if (threadIdx.x == 0) s_dots[q] = tot;
    __syncthreads();
  }
  if (wave == 0) lb_direction_core<LD>(dv, s_dots, history, s_SY, s_YY, &s_out);
  __syncthreads();
  const int m = s_out.m;
  const float c_g = s_out.c_g;
  double gd = 0, gg = 0, g1 = 0, gm = 0, nf = 0, dd = 0, dm = 0;
  for (int64_t i = threadIdx.x; i < N; i += MDE_BLOCK) {
    const float gv = g[i];
    float v = c_g * gv;
    for (int j = 0; j < m; ++j) {
      const float* sp = buf + (int64_t)(2 * s_out.order[j]) * N;
      v = fmaf(s_out.cy[j], sp[N + i], fmaf(s_out.cs[j], sp[i], v));
    }
    out[i] = v;
    const double gvd = gv, dv2 = v;
    gg += gvd * gvd;
    const double ag = fabs(gvd);
    g1 += ag;
    gm = ag > gm ? ag : gm;
    nf += (fabsf(gv) <= 3.402823466e+38f) ? 0.0 : 1.0;
    gd += gvd * dv2;
    dd += dv2 * dv2;
    const double ad = fabs(dv2);
    dm = ad > dm ? ad : dm;
  }
  const double v8[8] = {gd, gg, g1, gm, nf, dd, dm, 0.0};
  for (int q = 0; q < 8; ++q) {
    const bool is_max = q == 3 || q == 6;
    const double r = is_max ? mde_block_max(v8[q], &sm[0][0]) : mde_block_sum(v8[q], &sm[0][0]);
    if (threadIdx.x == 0) stats[q] = r;
    __syncthreads();
  }
  if (wave == 0) lb_write_back<LD>(dv, history, s_SY, s_YY, &s_out);
}

// ---- the device-driven step in four launches
// (1) k_lb_stage_all: stage the new pair and form every dot product against the stored pairs, eight
//     pairs per pass over the vectors; (2) k_lb_reduce adds the workgroups' partials to `dots`.
// (3) k_lbfgs_direction (above).
// (4) k_lb_combine_all: d_out from all pairs, its statistics against g in the same pass, reduced by
//     the last workgroup (what mde_vec_stats(g, d_out, NULL) writes).
template <int G>
__global__ __launch_bounds__(MDE_BLOCK) void k_lb_stage_all(int64_t N, const float* __restrict__ g,
                                                            float* __restrict__ g_prev,
                                                            const float* __restrict__ d, float t,
                                                            float* __restrict__ buf,
                                                            const LbDev* __restrict__ dv,
                                                            double* __restrict__ partial, MdeGate gate) {
  if (mde_gate_closed(gate)) return;
  __shared__ double sm[MDE_BLOCK / 64][4 + 5 * G];
  lb_stage_phase<false, G>(N, g, g_prev, d, t, buf, dv, partial, sm);
}

// dots[q] = sum_b partial[q * nb + b] for the 4 + 5 count rows that were written (one workgroup per
// row: 54 rows of up to 1024 partials are too much for one last workgroup)
__global__ void k_lb_reduce(int nb, const double* __restrict__ partial, const LbDev* __restrict__ dv,
                            double* __restrict__ dots, MdeGate gate) {
  if (mde_gate_closed(gate)) return;
  __shared__ double smem[8];
  const int q = blockIdx.x;
  if (q >= 4 + 5 * dv->count) return;
  double s = 0.0;
  for (int b = threadIdx.x; b < nb; b += blockDim.x) s += partial[(int64_t)q * nb + b];
  const double r = mde_block_sum(s, smem);
  if (threadIdx.x == 0) dots[q] = r;
}

// statistics rows as in k_vec_stats (x absent): 0 g.d 1 g.g 2 sum|g| 3 max|g| 4 #nonfinite 5 d.d 6 max|d| 7 0
__global__ __launch_bounds__(MDE_BLOCK) void k_lb_combine_all(int64_t N, const float* __restrict__ g,
                                                              const float* __restrict__ buf,
                                                              const LbDev* __restrict__ dv,
                                                              float* __restrict__ out,
                                                              double* __restrict__ partial,
                                                              double* __restrict__ stats,
                                                              unsigned int* __restrict__ ticket, MdeGate gate) {
  if (mde_gate_closed(gate)) return;
  __shared__ double smem[8];
  __shared__ float s_cs[MDE_LB_LD + MDE_LB_GROUP], s_cy[MDE_LB_LD + MDE_LB_GROUP];
  __shared__ int s_slot[MDE_LB_LD + MDE_LB_GROUP];
  const int count = dv->count;
  const float c_g = dv->c_g;
  if (threadIdx.x < MDE_LB_LD + MDE_LB_GROUP) {
    const bool in = (int)threadIdx.x < count;
    s_cs[threadIdx.x] = in ? dv->cs[threadIdx.x] : 0.0f;
    s_cy[threadIdx.x] = in ? dv->cy[threadIdx.x] : 0.0f;
    s_slot[threadIdx.x] = in ? dv->order[threadIdx.x] : 0;
  }
  __syncthreads();
  double gd = 0, gg = 0, g1 = 0, gm = 0, nf = 0, dd = 0, dm = 0;
  auto tally = [&](float gv, float v) __attribute__((always_inline)) {
    const double gvd = gv, dv2 = v;
    gg += gvd * gvd;
    const double ag = fabs(gvd);
    g1 += ag;
    gm = ag > gm ? ag : gm;
    nf += (fabsf(gv) <= 3.402823466e+38f) ? 0.0 : 1.0;
    gd += gvd * dv2;
    dd += dv2 * dv2;
    const double ad = fabs(dv2);
    dm = ad > dm ? ad : dm;
  };
  // 16-byte form (N a multiple of 4, 16-byte aligned vectors -- every history slot then is too): four
  // times the bytes per load instruction; 256 MB vectors: 3.8 -> TB/s measured in profiles/
  const bool vec = (N & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(buf) |
                                     reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const int64_t N4 = vec ? (N >> 2) : 0;
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N4; i += (int64_t)gridDim.x * MDE_BLOCK) {
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 v = make_float4(c_g * gv.x, c_g * gv.y, c_g * gv.z, c_g * gv.w);
    for (int j0 = 0; j0 < count; j0 += MDE_LB_GROUP) {
      float4 sv[MDE_LB_GROUP], yv[MDE_LB_GROUP];
#pragma unroll
      for (int j = 0; j < MDE_LB_GROUP; ++j) {
        const float4* sp = reinterpret_cast<const float4*>(buf + (int64_t)(2 * s_slot[j0 + j]) * N);
        sv[j] = sp[i];
        yv[j] = sp[N4 + i];
      }
#pragma unroll
      for (int j = 0; j < MDE_LB_GROUP; ++j)
        if (j0 + j < count) {
          const float cs = s_cs[j0 + j], cy = s_cy[j0 + j];
          v.x = fmaf(cy, yv[j].x, fmaf(cs, sv[j].x, v.x));
          v.y = fmaf(cy, yv[j].y, fmaf(cs, sv[j].y, v.y));
          v.z = fmaf(cy, yv[j].z, fmaf(cs, sv[j].z, v.z));
          v.w = fmaf(cy, yv[j].w, fmaf(cs, sv[j].w, v.w));
        }
    }
    reinterpret_cast<float4*>(out)[i] = v;
    tally(gv.x, v.x);
    tally(gv.y, v.y);
    tally(gv.z, v.z);
    tally(gv.w, v.w);
  }
  for (int64_t i = (N4 << 2) + (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    const float gv = g[i];
    float v = c_g * gv;
    for (int j0 = 0; j0 < count; j0 += MDE_LB_GROUP) {
      // the sixteen loads of eight pairs before the first use (entries beyond count: slot 0, weight 0)
      float sv[MDE_LB_GROUP], yv[MDE_LB_GROUP];
#pragma unroll
      for (int j = 0; j < MDE_LB_GROUP; ++j) {
        const float* sp = buf + (int64_t)(2 * s_slot[j0 + j]) * N;
        sv[j] = sp[i];
        yv[j] = sp[N + i];
      }
#pragma unroll
      for (int j = 0; j < MDE_LB_GROUP; ++j)
        if (j0 + j < count) v = fmaf(s_cy[j0 + j], yv[j], fmaf(s_cs[j0 + j], sv[j], v));
    }
    out[i] = v;
    tally(gv, v);
  }
  const int nb = gridDim.x, b = blockIdx.x;
  const double v8[8] = {gd, gg, g1, gm, nf, dd, dm, 0.0};
  mde_publish8(v8, (1u << 3) | (1u << 6), partial, nb, b);
  if (!mde_last_block(ticket)) return;
  mde_final_rows(8, nb, partial, stats, (1ull << 3) | (1ull << 6));
}

extern "C" int mde_lbfgs_dev_reset(mde_lbfgs* o, void* stream) {
  if (!o || !o->dev) return MDE_E_INVALID;
  hipLaunchKernelGGL(k_lbfgs_dev_reset, dim3(1), dim3(64), 0, mde_stream(stream), o->dev, o->history);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// (environment knobs are read once, not per step: MDE_LB_UNFUSED forces the four-launch path;
// MDE_LB_DEBUG = "blocks,spins,lds_bytes" lets a test launch more workgroups than can be resident and
// watch the rescue kernel take over; mde_lbfgs_debug_knobs changes them inside a process)
struct LbKnobs {
  bool unfused;
  int blocks, spins, lds;
  int max_blocks;  // workgroups of the one-launch step (MDE_LB_MAXBLOCKS: design probe)
  LbKnobs() : unfused(getenv("MDE_LB_UNFUSED") != nullptr), blocks(0), spins(0), lds(0), max_blocks(256) {
    if (const char* e = getenv("MDE_LB_DEBUG")) sscanf(e, "%d,%d,%d", &blocks, &spins, &lds);
    if (const char* e = getenv("MDE_LB_MAXBLOCKS")) max_blocks = std::min(std::max(atoi(e), 1), MDE_LB_FUSED_MAXBLOCKS);
  }
};
static LbKnobs& lb_knobs() {
  static LbKnobs k;
  return k;
}
// (test hook: one process-wide record, no synchronisation -- see include/mde_hip.h)
extern "C" int mde_lbfgs_debug_knobs(int32_t unfused, int32_t blocks, int32_t spins, int32_t lds_bytes) {
  LbKnobs& k = lb_knobs();
  if (unfused >= 0) k.unfused = unfused != 0;
  if (blocks >= 0) k.blocks = blocks;
  if (spins >= 0) k.spins = spins;
  if (lds_bytes >= 0) {
    if (lds_bytes == 0 && k.lds > 0) {
      // back to no dynamic LDS: the attribute a test raised on the kernels goes back too
      MDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lb_fused<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 0));
      MDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lb_fused<MDE_LB_GROUP>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 0));
    }
    k.lds = lds_bytes;
  }
  return MDE_OK;
}

// One whole L-BFGS direction update without a host round trip: stage (y = g - g_prev, s = t d,
// g_prev <- g), accept / reject, two-loop recursion, d_out = the new direction, stats as
// mde_vec_stats(g, d_out, NULL).  ASYNC.  (The host-driven mde_lbfgs_stage / commit / combine act on
// the host-side bookkeeping and must not be mixed with this on one object.)
static int lbfgs_dev_step_impl(mde_lbfgs* o, const float* g, float* g_prev, const float* d, float t,
                               float* d_out, double* stats, double* work, void* stream, MdeGate gate) {
  if (!o || !o->dev || !g || !g_prev || !d || !d_out || !stats || !work) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  const int64_t N = o->N;
  double* partial = work + MDE_SMALL_DOUBLES;
  // (environment knobs are read once: MDE_LB_UNFUSED forces the four-launch path; MDE_LB_DEBUG =
  // "blocks,spins,lds_bytes" lets a test launch more workgroups than can be resident and watch the
  // rescue kernel take over)
  const LbKnobs& knobs = lb_knobs();
  if (N <= MDE_LB_FUSED_MAXN && o->history < MDE_LB_FUSED_LD && !knobs.unfused) {
    // At most 256 workgroups: the arrival points need every workgroup of the launch resident at once.
    // The kernel fits two workgroups per CU (198 VGPRs), i.e. 512 on the chip -- 256 leaves half of that
    // to whatever else is resident, e.g. the same kernel of another process sharing the GPU.  (Round 4:
    // 128 until the phase stamps of tools/lbfused_probe.py showed the two streaming phases at 11.6 us
    // each at N = 140k -- five dependent trips per thread, nothing to hide their latency behind.)
    // Residency is still not GUARANTEED by an ordinary launch: a workgroup that waits longer than the
    // spin limit gives up, and k_lb_rescue (always queued behind) redoes the step -- see k_lb_fused.
    int nbf = mde_grid(N, MDE_BLOCK, knobs.max_blocks);
    if (knobs.blocks > 0) nbf = std::min(knobs.blocks, MDE_LB_FUSED_MAXBLOCKS);
    const unsigned int spin_limit = knobs.spins > 0 ? (unsigned int)knobs.spins : (1u << 20);
    static std::atomic<unsigned int> launches{0};  // one epoch per launch, shared by every solver object
    const unsigned int epoch = 4u * (launches.fetch_add(1u) + 1u);
    auto kern = (o->history > MDE_LB_GROUP && o->history <= 12) ? k_lb_fused<12> : k_lb_fused<MDE_LB_GROUP>;
    if (knobs.lds > 0)
      MDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, knobs.lds));
    hipLaunchKernelGGL(kern, dim3(nbf), dim3(MDE_BLOCK), (size_t)std::max(knobs.lds, 0), st, N, g, g_prev, d, t, o->buf,
                       o->dev, o->history, d_out, partial, stats, work_lb_flags(work), epoch, spin_limit, gate);
    MDE_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_lb_rescue, dim3(1), dim3(MDE_BLOCK), 0, st, N, g, o->buf, o->dev, o->history, d_out, nbf, partial, stats,
                       work_lb_flags(work), epoch, gate);
    MDE_LAUNCH_CHECK();
    return MDE_OK;
  }
  const int nb = mde_grid(N, MDE_BLOCK * 2, 1024);
  double* dots = work;  // the small area: 4 + 5 * 63 doubles at most
  if (o->history > MDE_LB_GROUP && o->history <= 12)
    hipLaunchKernelGGL(k_lb_stage_all<12>, dim3(nb), dim3(MDE_BLOCK), 0, st, N, g, g_prev, d, t, o->buf, o->dev, partial, gate);
  else
    hipLaunchKernelGGL(k_lb_stage_all<MDE_LB_GROUP>, dim3(nb), dim3(MDE_BLOCK), 0, st, N, g, g_prev, d, t, o->buf,
                       o->dev, partial, gate);
  MDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_lb_reduce, dim3(4 + 5 * o->history), dim3(MDE_BLOCK), 0, st, nb, partial, o->dev, dots, gate);
  MDE_LAUNCH_CHECK();
  if (o->history < 16) {
    hipLaunchKernelGGL(k_lbfgs_direction<16>, dim3(1), dim3(64), MDE_LB_DIR_LDS(16), st, o->dev, dots, o->history, gate);
  } else {
    static bool dir_attr = false;
    if (!dir_attr) {
      MDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lbfgs_direction<MDE_LB_LD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)MDE_LB_DIR_LDS(MDE_LB_LD)));
      dir_attr = true;
    }
    hipLaunchKernelGGL(k_lbfgs_direction<MDE_LB_LD>, dim3(1), dim3(64), MDE_LB_DIR_LDS(MDE_LB_LD), st, o->dev, dots,
                       o->history, gate);
  }
  MDE_LAUNCH_CHECK();
  const int nbc = mde_grid(N, MDE_BLOCK * 2, MDE_RED_BLOCKS);  // (its last workgroup adds nbc partials per row)
  hipLaunchKernelGGL(k_lb_combine_all, dim3(nbc), dim3(MDE_BLOCK), 0, st, N, g, o->buf, o->dev, d_out, partial,
                     stats, work_ticket(work, TK_LB_COMBINE), gate);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
extern "C" int mde_lbfgs_dev_step(mde_lbfgs* o, const float* g, float* g_prev, const float* d, float t,
                                  float* d_out, double* stats, double* work, void* stream) {
  return lbfgs_dev_step_impl(o, g, g_prev, d, t, d_out, stats, work, stream, MdeGate{nullptr, 0u});
}

// ---- round 6: the same step in two halves, for a solve whose vectors are sharded by rows across ranks
// (pymde_amd/optim.py, _ShardedEngine).  Every rank keeps the history of ITS rows only (o->N = its elements):
// mde_lbfgs_dev_stage stages the pair and leaves the 4 + 5 * history inner products over the rank's elements in
// work[0 ..); the caller sums them across the ranks in place (one small all-reduce of doubles); mde_lbfgs_dev_finish
// takes the accept / drop-oldest decision and runs the two-loop recursion on those sums -- every rank the same
// arithmetic on the same numbers -- and writes the rank's rows of the new direction, with the statistics of (g, d_out)
// over its elements (partial: the caller reduces them, mde_rank_reduce).  Always the four-launch form's kernels.
extern "C" int mde_lbfgs_dev_stage(mde_lbfgs* o, const float* g, float* g_prev, const float* d, float t, double* work,
                                   void* stream) {
  if (!o || !o->dev || !g || !g_prev || !d || !work) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  const int64_t N = o->N;
  double* partial = work + MDE_SMALL_DOUBLES;
  const MdeGate gate{nullptr, 0u};
  const int nb = mde_grid(N, MDE_BLOCK * 2, 1024);
  if (o->history > MDE_LB_GROUP && o->history <= 12)
    hipLaunchKernelGGL(k_lb_stage_all<12>, dim3(nb), dim3(MDE_BLOCK), 0, st, N, g, g_prev, d, t, o->buf, o->dev, partial, gate);
  else
    hipLaunchKernelGGL(k_lb_stage_all<MDE_LB_GROUP>, dim3(nb), dim3(MDE_BLOCK), 0, st, N, g, g_prev, d, t, o->buf,
                       o->dev, partial, gate);
  MDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_lb_reduce, dim3(4 + 5 * o->history), dim3(MDE_BLOCK), 0, st, nb, partial, o->dev, work, gate);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
extern "C" int mde_lbfgs_dev_finish(mde_lbfgs* o, const float* g, float* d_out, double* stats, double* work, void* stream) {
  if (!o || !o->dev || !g || !d_out || !stats || !work) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  const int64_t N = o->N;
  double* partial = work + MDE_SMALL_DOUBLES;
  double* dots = work;
  const MdeGate gate{nullptr, 0u};
  if (o->history < 16) {
    hipLaunchKernelGGL(k_lbfgs_direction<16>, dim3(1), dim3(64), MDE_LB_DIR_LDS(16), st, o->dev, dots, o->history, gate);
  } else {
    MDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lbfgs_direction<MDE_LB_LD>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)MDE_LB_DIR_LDS(MDE_LB_LD)));
    hipLaunchKernelGGL(k_lbfgs_direction<MDE_LB_LD>, dim3(1), dim3(64), MDE_LB_DIR_LDS(MDE_LB_LD), st, o->dev, dots,
                       o->history, gate);
  }
  MDE_LAUNCH_CHECK();
  const int nbc = mde_grid(N, MDE_BLOCK * 2, MDE_RED_BLOCKS);
  hipLaunchKernelGGL(k_lb_combine_all, dim3(nbc), dim3(MDE_BLOCK), 0, st, N, g, o->buf, o->dev, d_out, partial,
                     stats, work_ticket(work, TK_LB_COMBINE), gate);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
extern "C" int32_t mde_lbfgs_dev_dots(const mde_lbfgs* o) { return o ? 4 + 5 * o->history : 0; }

// out[q] = sum (or max, bit q of max_mask) over the ranks r, in rank order, of in[r * count + q]  (q < count <= 64);
// loss_slot >= 0: that slot's result also goes to *loss_out as a float.  The per-rank records of a sharded solve --
// statistics boards and loss shares, gathered with one all-gather -- reduced with mixed operations in one launch.
__global__ void k_rank_reduce(int world, int count, unsigned long long max_mask, const double* __restrict__ in,
                              double* __restrict__ out, int loss_slot, float* __restrict__ loss_out) {
  const int q = threadIdx.x;
  if (q >= count) return;
  const bool is_max = (max_mask >> q) & 1ull;
  double a = in[q];
  for (int r = 1; r < world; ++r) {
    const double v = in[(size_t)r * count + q];
    a = is_max ? (v > a ? v : a) : a + v;
  }
  out[q] = a;
  if (q == loss_slot && loss_out) *loss_out = (float)a;
}
extern "C" int mde_rank_reduce(int32_t world, int32_t count, uint64_t max_mask, const double* in, double* out,
                               int32_t loss_slot, float* loss_out, void* stream) {
  if (world < 1 || count < 1 || count > 64 || !in || !out) return MDE_E_INVALID;
  hipLaunchKernelGGL(k_rank_reduce, dim3(1), dim3(64), 0, mde_stream(stream), (int)world, (int)count,
                     (unsigned long long)max_mask, in, out, (int)loss_slot, loss_out);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// copies of the device bookkeeping for tests: count, accepted
extern "C" int mde_lbfgs_dev_info(const mde_lbfgs* o, int32_t* count_host, int32_t* accepted_host,
                                  void* stream) {
  if (!o || !o->dev || !count_host || !accepted_host) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  int h[2] = {0, 0};
  MDE_HIP(hipMemcpyAsync(h, o->dev, sizeof(h), hipMemcpyDeviceToHost, st));
  MDE_HIP(hipStreamSynchronize(st));
  *count_host = h[0];
  *accepted_host = h[1];
  return MDE_OK;
}

template <bool FIRST>
__global__ __launch_bounds__(MDE_BLOCK) void k_lbfgs_combine(int64_t N, const float* __restrict__ g,
                                                             float c_g, LbPtrs P,
                                                             float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    float v = FIRST ? c_g * g[i] : out[i];
#pragma unroll
    for (int j = 0; j < MDE_LB_GROUP; ++j)
      if (j < P.count) v = fmaf(P.cy[j], P.y[j][i], fmaf(P.cs[j], P.s[j][i], v));
    out[i] = v;
  }
}

extern "C" int mde_lbfgs_combine(mde_lbfgs* o, const float* g, float c_g, const float* cs,
                                 const float* cy, float* d_out, double* stats, double* work,
                                 void* stream) {
  if (!o || !g || !d_out || !stats || !work || (o->count > 0 && (!cs || !cy))) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  const int64_t N = o->N;
  const int nb = mde_grid(N, MDE_BLOCK * 2, 2048);
  int done = 0;
  bool first = true;
  do {
    LbPtrs P;
    P.count = o->count - done;
    if (P.count > MDE_LB_GROUP) P.count = MDE_LB_GROUP;
    for (int j = 0; j < MDE_LB_GROUP; ++j) {
      const int slot = (j < P.count) ? o->order[done + j] : 0;
      P.s[j] = o->S(slot);
      P.y[j] = o->Y(slot);
      P.cs[j] = (j < P.count) ? cs[done + j] : 0.f;
      P.cy[j] = (j < P.count) ? cy[done + j] : 0.f;
    }
    if (first)
      hipLaunchKernelGGL((k_lbfgs_combine<true>), dim3(nb), dim3(MDE_BLOCK), 0, st, N, g, c_g, P, d_out);
    else
      hipLaunchKernelGGL((k_lbfgs_combine<false>), dim3(nb), dim3(MDE_BLOCK), 0, st, N, g, c_g, P,
                         d_out);
    MDE_LAUNCH_CHECK();
    first = false;
    done += P.count;
  } while (done < o->count);
  return vec_stats_impl(N, g, d_out, nullptr, stats, work, st);
}

// ---------------------------------------------------------------- one solver iteration as two calls
// the gate word of the pre-enqueued L-BFGS step (MdeGate): a double of the work buffer's small area nobody
// else uses
static inline unsigned int* work_turn_gate(double* work) { return reinterpret_cast<unsigned int*>(work + MDE_WS_TURN_GATE); }
static bool turn_pre_enabled() {
  static const bool on = getenv("MDE_TURN_NOPRE") == nullptr;  // (design probe: no look-ahead of the L-BFGS step)
  return on;
}

// One iteration: [L-BFGS step unless step_done] retraction of X + dir, evaluation, statistics -- whose
// kernel also DECIDES the first pass of the line search (f0, c1, c2: MdeMirror) -- and, with allow_pre, the
// L-BFGS step of the iteration after this one behind a gate that opens iff this trial is accepted.
static int turn_enqueue_impl(mde_turn_desc* T, int32_t cur, float t_prev, double f0, double c1, double c2,
                             bool allow_pre, bool step_done, void* stream) {
  if (!T || (cur != 0 && cur != 1) || (T->kind != 0 && T->kind != 1)) return MDE_E_INVALID;
  const int64_t N = T->n * (int64_t)T->d;
  float* Xc = T->X[cur];
  float* Xt = T->X[1 - cur];
  int rc = MDE_OK;
  if (!step_done) {
    rc = lbfgs_dev_step_impl(T->lbfgs, T->g, T->g_prev, T->dir, t_prev, T->dir, T->board + 16, T->work, stream,
                             MdeGate{nullptr, 0u});
    if (rc != MDE_OK) return rc;
  }
  if (T->kind == 0)
    rc = mde_center_step(T->n, T->d, Xc, T->dir, 1.0f, Xt, T->work, stream);
  else
    rc = mde_std_retract_step(T->n, T->d, Xc, T->dir, 1.0f, Xt, 1, T->work, T->status, stream);
  if (rc != MDE_OK) return rc;
  rc = mde_average_distortion(T->plan, Xt, T->d, T->func, 1.0f, T->g, T->loss_dev, stream);
  if (rc != MDE_OK) return rc;
  // the last kernel writes [loss | status | board] into the pinned mirror itself (no copy behind it)
  static std::atomic<unsigned long long> turns{0};
  const unsigned long long id = turns.fetch_add(1ull) + 1ull;
  T->seq = (double)id;  // (exact in a double for 2^53 iterations)
  unsigned int gate_value = (unsigned int)(id & 0xffffffffull);
  if (gate_value == 0u) gate_value = 1u;
  MdeMirror mirror{T->loss_dev, T->status, T->board, reinterpret_cast<char*>(T->host_dst),
                   (int)(T->read_bytes - 8 * 24), T->seq, work_turn_gate(T->work), gate_value, f0, c1, c2};
  if (T->kind == 0)
    rc = vec_stats_impl(N, T->g, T->dir, Xt, T->board, T->work, mde_stream(stream), mirror);
  else
    rc = std_tangent_stats_impl(T->n, T->d, Xt, T->g, T->dir, T->board, T->work, stream, mirror);
  if (rc != MDE_OK) return rc;
  T->pre_id = 0.0;
  if (allow_pre && turn_pre_enabled()) {
    // the direction update of the NEXT iteration (an accepted first trial is the step t = 1), gated
    rc = lbfgs_dev_step_impl(T->lbfgs, T->g, T->g_prev, T->dir, 1.0f, T->dir, T->board + 16, T->work, stream,
                             MdeGate{work_turn_gate(T->work), gate_value});
    if (rc != MDE_OK) return rc;
    T->pre_id = (double)gate_value;
  }
  return MDE_OK;
}

extern "C" int mde_turn_enqueue(mde_turn_desc* T, int32_t cur, float t_prev, double f0, double c1, double c2,
                                int32_t allow_pre, void* stream) {
  // a gated L-BFGS step that has run but was not followed by its iteration (mde_turn_wait with allow_next = 0
  // on an accepted trial; out[20] said so) IS this iteration's step: running it again would stage s = t d with
  // the direction the first run has already overwritten and y = g - g_prev = 0
  const bool pending = T && T->pre_id == -1.0;
  if (pending) T->pre_id = 0.0;
  return turn_enqueue_impl(T, cur, t_prev, f0, c1, c2, allow_pre != 0, pending, stream);
}

extern "C" int mde_turn_wait(mde_turn_desc* T, int32_t cur, double f0, int32_t allow_next, double c1, double c2,
                             double eps_pre, double* out, void* stream) {
  if (!T || !out || (cur != 0 && cur != 1)) return MDE_E_INVALID;
  // Poll the sequence word the iteration's last kernel writes behind its data (a completion signal
  // takes microseconds longer to reach a waiting thread); the stream is queried now and then, so the
  // wait also ends -- with everything visible -- should the word never show up.
  {
    const volatile double* flag = T->host_board + 24;
    hipStream_t st = mde_stream(stream);
    for (unsigned spins = 0;; ++spins) {
      if (*flag == T->seq) break;
      if ((spins & 255u) == 255u) {
        const hipError_t e = hipStreamQuery(st);
        if (e == hipSuccess) break;
        if (e != hipErrorNotReady) MDE_HIP(e);
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  // The record is read from the TAGGED lines (mde_mirror_write): every 64-byte line carries the sequence number in
  // its last slot and arrives whole, so a line whose tag is this iteration's holds this iteration's values.  A line
  // that is still on its way is waited for; should one never arrive (it cannot once the stream is idle) that is an error.
  double pay[7 * MDE_MIRROR_TAG_LINES];
  {
    const volatile double* tg = mde_mirror_tagged(T->host_board);
    hipStream_t st = mde_stream(stream);
    bool synced = false;
    for (unsigned spins = 0;; ++spins) {
      bool ok = true;
      for (int l = 0; l < MDE_MIRROR_TAG_LINES; ++l) ok = ok && tg[8 * l + 7] == T->seq;
      if (ok) {
        std::atomic_thread_fence(std::memory_order_acquire);
        for (int l = 0; l < MDE_MIRROR_TAG_LINES; ++l)
          for (int q = 0; q < 7; ++q) pay[7 * l + q] = tg[8 * l + q];
        // (a line rewritten between the tag check and the copy is impossible: the next iteration is not enqueued yet)
        break;
      }
      if ((spins & 4095u) == 4095u) {
        if (synced) {
          mde_set_error("mde_turn_wait: the iteration's record never reached the host mirror");
          return MDE_E_HIP;
        }
        MDE_HIP(hipStreamSynchronize(st));
        synced = true;
      }
    }
  }
  const double f_new = pay[0];
  const double* hb = pay + 2;  // board[0..24)
  const int32_t status_word = (int32_t)pay[1];
  for (int q = 0; q < 8; ++q) {
    out[4 + q] = hb[q];
    out[12 + q] = hb[16 + q];
  }
  out[0] = f_new;
  out[3] = (double)status_word;
  // the first pass of the bracketing loop at t = 1 (lbfgs.py:88-110): not bad, Armijo, curvature -- decided
  // by the iteration's last kernel (mde_mirror_write) from f0 as passed when it was enqueued; the gate of a
  // pre-enqueued L-BFGS step followed THAT verdict, so it is the one to go by
  (void)f0;
  const bool accept = hb[8] != 0.0;
  out[1] = accept ? 1.0 : 0.0;
  out[2] = 0.0;
  const bool step_done = accept && status_word == 0 && T->pre_id > 0.0;  // (its gate opened)
  T->pre_id = 0.0;
  out[20] = 0.0;
  if (step_done && !allow_next) {
    // the speculative step of the next iteration has run (g_prev <- g, a new history pair, dir and the
    // direction statistics overwritten) and the caller does not want that iteration launched now: the step
    // stays PENDING -- the next mde_turn_enqueue on this descriptor takes it as done -- and the caller is told
    T->pre_id = -1.0;
    out[20] = 1.0;
  }
  if (accept && allow_next && status_word == 0) {
    // (hb[1]: |g|^2 at the accepted point -- the next iteration goes on to another one iff it is above eps)
    const bool allow_pre = eps_pre >= 0.0 && std::sqrt(hb[1]) > eps_pre;
    const int rc = turn_enqueue_impl(T, 1 - cur, 1.0f, f_new, c1, c2, allow_pre, step_done, stream);
    if (rc != MDE_OK) return rc;
    out[2] = 1.0;
  }
  return MDE_OK;
}
