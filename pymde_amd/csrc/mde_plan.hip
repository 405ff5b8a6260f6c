// mde_plan.hip -- one-time device preprocessing of an edge list into the symmetrised
// incidence CSR the hot kernel streams.  Replaces the reference's `edges/_lhs/_rhs`
// buffers and their validation [ref: pymde/problem.py:129-170,
// pymde/average_distortion.py:58-59].
//
// Layout produced (all int32, device):
//   rowptr[nloc+1]   offsets of the local rows v = row_lo + r
//   nbr[H]           neighbour vertex of each half-edge
//   eid[H]           original edge id of each half-edge (param expansion, fallback path)
// Half-edges of a row are ordered by original edge id (stable radix sort), so every
// per-row sum has a fixed order: results are bitwise reproducible.
#include <hipcub/hipcub.hpp>

#include <stdarg.h>

#include "mde_common.h"
#include "mde_plan.h"

// ---------------------------------------------------------------- error plumbing
static thread_local std::string g_last_error;

void mde_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
int mde_hip_fail(hipError_t e, const char* what, const char* file, int line) {
  mde_set_error("HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return MDE_E_HIP;
}
extern "C" const char* mde_last_error(void) { return g_last_error.c_str(); }
extern "C" int mde_abi_version(void) { return MDE_ABI_VERSION; }

// ---------------------------------------------------------------- plan object
extern "C" int64_t mde_plan_n(const mde_plan* p) { return p ? p->n : 0; }
extern "C" int64_t mde_plan_p(const mde_plan* p) { return p ? p->p : 0; }
extern "C" int64_t mde_plan_half_edges(const mde_plan* p) { return p ? p->H : 0; }
extern "C" int64_t mde_plan_row_lo(const mde_plan* p) { return p ? p->row_lo : 0; }
extern "C" int64_t mde_plan_row_hi(const mde_plan* p) { return p ? p->row_hi : 0; }
extern "C" const int32_t* mde_plan_rowptr(const mde_plan* p) { return p ? p->rowptr : nullptr; }
extern "C" const int32_t* mde_plan_nbr(const mde_plan* p) { return p ? p->nbr : nullptr; }
extern "C" const int32_t* mde_plan_eid(const mde_plan* p) { return p ? p->eid : nullptr; }

// internal accessor used by the kernel translation unit
double* mde_plan_partials(mde_plan* p) { return p->partials; }
void mde_ring_release(mde_plan* plan);  // mde_ring.hip

// The device-wide sort and scan of this unit, shared with the layout builder of mde_ring.hip: a plan is
// created before its layout is built, so the builder finds this unit's code object loaded (its own stays a
// few hundred KB: HIP loads a unit's code object on the first launch from it, ~1.5 ms per MB).
hipError_t mde_sort_pairs_u32(void* tmp, size_t& tmp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                              const uint32_t* vals_in, uint32_t* vals_out, int n, int begin_bit, int end_bit,
                              hipStream_t st) {
  return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, st);
}
hipError_t mde_exclusive_sum_i32(void* tmp, size_t& tmp_bytes, const int32_t* in, int32_t* out, int n, hipStream_t st) {
  return hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, in, out, n, st);
}
float mde_plan_avg_degree(const mde_plan* p) { return p->avg_degree; }

extern "C" int mde_plan_destroy(mde_plan* plan) {
  if (!plan) return MDE_OK;
  if (plan->rowptr) (void)hipFree(plan->rowptr);
  if (plan->nbr) (void)hipFree(plan->nbr);
  if (plan->eid) (void)hipFree(plan->eid);
  if (plan->partials) (void)hipFree(plan->partials);
  if (plan->hrow) (void)hipFree(plan->hrow);
  if (plan->flat_rec) (void)hipFree(plan->flat_rec);
  if (plan->order) (void)hipFree(plan->order);
  mde_ring_release(plan);
  delete plan;
  return MDE_OK;
}

// ---------------------------------------------------------------- kernels
// flags[0] = #self edges, flags[1] = #out-of-range endpoints     [ref: problem.py:134-140]
__global__ void k_validate(int64_t n, int64_t p, const int64_t* __restrict__ edges, int* flags) {
  int self = 0, oor = 0;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < p;
       k += (int64_t)gridDim.x * blockDim.x) {
    const longlong2 e = reinterpret_cast<const longlong2*>(edges)[k];
    self += (e.x == e.y);
    oor += (e.x < 0 || e.x >= n || e.y < 0 || e.y >= n);
  }
  self = mde_wave_sum(self);
  oor = mde_wave_sum(oor);
  if ((threadIdx.x & 63) == 0) {
    if (self) atomicAdd(&flags[0], self);
    if (oor) atomicAdd(&flags[1], oor);
  }
}

// half-edge h = 2k + side: side 0 lives in row i (neighbour j), side 1 in row j (neighbour i).
// key = local row index, or nloc for rows another rank owns.
__global__ void k_make_keys(int64_t p, const int64_t* __restrict__ edges, int64_t row_lo,
                            int64_t row_hi, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t nloc = (uint32_t)(row_hi - row_lo);
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < p;
       k += (int64_t)gridDim.x * blockDim.x) {
    const longlong2 e = reinterpret_cast<const longlong2*>(edges)[k];
    const uint32_t k0 = (e.x >= row_lo && e.x < row_hi) ? (uint32_t)(e.x - row_lo) : nloc;
    const uint32_t k1 = (e.y >= row_lo && e.y < row_hi) ? (uint32_t)(e.y - row_lo) : nloc;
    reinterpret_cast<uint2*>(keys)[k] = make_uint2(k0, k1);
    reinterpret_cast<uint2*>(vals)[k] = make_uint2((uint32_t)(2 * k), (uint32_t)(2 * k + 1));
  }
}

// rowptr[r] = first sorted position whose key >= r, for r = 0..nloc
__global__ void k_rowptr(int64_t H2, uint32_t nloc, const uint32_t* __restrict__ keys,
                         int32_t* __restrict__ rowptr) {
  for (int64_t h = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; h <= H2;
       h += (int64_t)gridDim.x * blockDim.x) {
    const int64_t prev = (h == 0) ? -1 : (int64_t)keys[h - 1];
    int64_t cur = (h == H2) ? (int64_t)nloc : (int64_t)keys[h];
    if (cur > (int64_t)nloc) cur = nloc;
    for (int64_t r = prev + 1; r <= cur; ++r) rowptr[r] = (int32_t)h;
  }
}

__global__ void k_fill_half_edges(int64_t H, const uint32_t* __restrict__ vals,
                                  const int64_t* __restrict__ edges, int32_t* __restrict__ nbr,
                                  int32_t* __restrict__ eid) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < H;
       q += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = vals[q];
    const uint32_t k = h >> 1;
    // side 0 -> neighbour is edges[k][1]; side 1 -> neighbour is edges[k][0]
    const int64_t u = edges[2 * (int64_t)k + (1 - (h & 1))];
    nbr[q] = (int32_t)u;
    eid[q] = (int32_t)k;
  }
}

// number of edge endpoints below row_lo = half-edges of the rows another rank owns before ours
__global__ void k_count_below(int64_t p, const int64_t* __restrict__ edges, int64_t row_lo,
                              unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < p;
       k += (int64_t)gridDim.x * blockDim.x) {
    const longlong2 e = reinterpret_cast<const longlong2*>(edges)[k];
    c += (e.x < row_lo) + (e.y < row_lo);
  }
  c = mde_wave_sum(c);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// hrow[h] = row whose [rowptr[r], rowptr[r + 1]) holds position h (binary search; one-time)
__global__ void k_fill_hrow(int64_t H, int32_t nloc, const int32_t* __restrict__ rowptr,
                            int32_t* __restrict__ hrow) {
  for (int64_t h = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; h < H;
       h += (int64_t)gridDim.x * blockDim.x) {
    int32_t lo = 0, hi = nloc;  // invariant: rowptr[lo] <= h < rowptr[hi]
    while (hi - lo > 1) {
      const int32_t mid = lo + ((hi - lo) >> 1);
      if (rowptr[mid] <= (int32_t)h) lo = mid; else hi = mid;
    }
    hrow[h] = lo;
  }
}
__global__ void k_any_empty_row(int32_t nloc, const int32_t* __restrict__ rowptr, int* __restrict__ flag) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nloc;
       r += (int64_t)gridDim.x * blockDim.x)
    if (rowptr[r] == rowptr[r + 1]) *flag = 1;
}

// Build the flat schedule of the CSR kernel (idempotent).
int mde_plan_flat(mde_plan* plan, hipStream_t st) {
  if (plan->hrow || plan->H <= 0) return MDE_OK;
  const int64_t nloc = plan->row_hi - plan->row_lo;
  const int64_t phase = plan->h_offset % MDE_FLAT_T;
  const int64_t nt = (plan->H + phase + MDE_FLAT_T - 1) / MDE_FLAT_T;
  int32_t* hrow = nullptr;
  float* rec = nullptr;
  int* flag = reinterpret_cast<int*>(plan->partials + MDE_MAX_PARTIALS + 1);  // scratch word
  int hflag = 0;
  hipError_t e = hipMalloc(&hrow, (size_t)plan->H * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&rec, (size_t)nt * 16 * sizeof(float));
  if (e == hipSuccess) e = hipMemsetAsync(flag, 0, sizeof(int), st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_fill_hrow, dim3(mde_grid(plan->H, MDE_BLOCK, 8192)), dim3(MDE_BLOCK), 0, st, plan->H,
                       (int32_t)nloc, plan->rowptr, hrow);
    hipLaunchKernelGGL(k_any_empty_row, dim3(mde_grid(nloc, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, (int32_t)nloc,
                       plan->rowptr, flag);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemsetAsync(flag, 0, sizeof(int), st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) {
    if (hrow) (void)hipFree(hrow);
    if (rec) (void)hipFree(rec);
    return mde_hip_fail(e, "mde_plan_flat", __FILE__, __LINE__);
  }
  plan->hrow = hrow;
  plan->flat_rec = rec;
  plan->n_tiles = nt;
  plan->has_empty = hflag;
  return MDE_OK;
}

__global__ void k_degree(int64_t p, const int64_t* __restrict__ edges, int32_t* __restrict__ deg) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < p;
       k += (int64_t)gridDim.x * blockDim.x) {
    const longlong2 e = reinterpret_cast<const longlong2*>(edges)[k];
    atomicAdd(&deg[e.x], 1);
    atomicAdd(&deg[e.y], 1);
  }
}

// bounds[r] = smallest v with cum_incl[v-1] >= r * total / world (cum_incl inclusive scan of deg)
__global__ void k_bounds(int64_t n, int32_t world, const int32_t* __restrict__ cum_incl,
                         int64_t* __restrict__ bounds) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > world) return;
  if (r == 0) {
    bounds[0] = 0;
    return;
  }
  if (r == world) {
    bounds[world] = n;
    return;
  }
  const int64_t total = cum_incl[n - 1];
  const int64_t target = (total * r) / world;
  int64_t lo = 0, hi = n;  // first v with cum_incl[v] >= target
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)cum_incl[mid] >= target)
      hi = mid;
    else
      lo = mid + 1;
  }
  bounds[r] = lo + 1 > n ? n : lo + 1;
}

__global__ void k_expand(int64_t H, const int32_t* __restrict__ eid, const float* __restrict__ in,
                         float* __restrict__ out) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < H;
       q += (int64_t)gridDim.x * blockDim.x)
    out[q] = in[eid[q]];
}

// ---------------------------------------------------------------- host side
static int validate_edges(int64_t n, int64_t p, const int64_t* edges, hipStream_t st) {
  int* flags = nullptr;
  MDE_HIP(hipMalloc(&flags, 2 * sizeof(int)));
  hipError_t e = hipMemsetAsync(flags, 0, 2 * sizeof(int), st);
  if (e == hipSuccess && p > 0) {
    hipLaunchKernelGGL(k_validate, dim3(mde_grid(p, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, n, p, edges,
                       flags);
    e = hipGetLastError();
  }
  int h[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(h, flags, sizeof(h), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(flags);
  if (e != hipSuccess) return mde_hip_fail(e, "validate_edges", __FILE__, __LINE__);
  if (h[1]) {
    mde_set_error("%d edge endpoints are outside [0, %lld)", h[1], (long long)n);
    return MDE_E_RANGE;
  }
  if (h[0]) {
    mde_set_error("The edge list must not contain self edges (%d found)", h[0]);
    return MDE_E_SELF_EDGE;
  }
  return MDE_OK;
}

static int bits_for(uint64_t maxval) {
  int b = 1;
  while (b < 32 && (maxval >> b)) ++b;
  return b;
}

extern "C" int mde_plan_create(int64_t n, int64_t p, const int64_t* edges, int64_t row_lo,
                               int64_t row_hi, void* stream, mde_plan** out) {
  if (!out) return MDE_E_INVALID;
  *out = nullptr;
  if (n <= 0 || p < 0 || (p > 0 && !edges) || row_lo < 0 || row_hi > n || row_lo > row_hi) {
    mde_set_error("mde_plan_create: invalid arguments (n=%lld p=%lld rows=[%lld,%lld))", (long long)n,
                  (long long)p, (long long)row_lo, (long long)row_hi);
    return MDE_E_INVALID;
  }
  if (n >= (int64_t)1 << 31 || 2 * p >= ((int64_t)1 << 31) - 1) {
    mde_set_error("mde_plan_create: n=%lld / p=%lld exceed the int32 plan (n < 2^31, 2p < 2^31)",
                  (long long)n, (long long)p);
    return MDE_E_TOO_LARGE;
  }
  hipStream_t st = mde_stream(stream);
  int rc = validate_edges(n, p, edges, st);
  if (rc != MDE_OK) return rc;

  mde_plan* plan = new mde_plan();
  plan->n = n;
  plan->p = p;
  plan->row_lo = row_lo;
  plan->row_hi = row_hi;
  const int64_t nloc = row_hi - row_lo;
  const int64_t H2 = 2 * p;

  uint32_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;
  void* tmp = nullptr;
  auto cleanup = [&]() {
    if (keys) (void)hipFree(keys);
    if (vals) (void)hipFree(vals);
    if (keys2) (void)hipFree(keys2);
    if (vals2) (void)hipFree(vals2);
    if (tmp) (void)hipFree(tmp);
  };
#define PLAN_HIP(call)                                            \
  do {                                                            \
    hipError_t e__ = (call);                                      \
    if (e__ != hipSuccess) {                                      \
      cleanup();                                                  \
      mde_plan_destroy(plan);                                     \
      return mde_hip_fail(e__, #call, __FILE__, __LINE__);        \
    }                                                             \
  } while (0)

  PLAN_HIP(hipMalloc(&plan->rowptr, (nloc + 1) * sizeof(int32_t)));
  // (+2: the slot after the partials is the arrival counter of the in-kernel loss reduction, the
  // one after it a scratch flag of mde_plan_expand_codebook)
  PLAN_HIP(hipMalloc(&plan->partials, MDE_PARTIALS_DOUBLES * sizeof(double)));
  PLAN_HIP(hipMemsetAsync(plan->partials + MDE_MAX_PARTIALS, 0, 4 * sizeof(double), st));
  int32_t Hlocal = 0;
  if (H2 > 0) {
    const size_t bytes = (size_t)H2 * sizeof(uint32_t);
    PLAN_HIP(hipMalloc(&keys, bytes));
    PLAN_HIP(hipMalloc(&vals, bytes));
    PLAN_HIP(hipMalloc(&keys2, bytes));
    PLAN_HIP(hipMalloc(&vals2, bytes));
    hipLaunchKernelGGL(k_make_keys, dim3(mde_grid(p, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, p, edges,
                       row_lo, row_hi, keys, vals);
    PLAN_HIP(hipGetLastError());
    size_t tmp_bytes = 0;
    const int end_bit = bits_for((uint64_t)nloc);
    PLAN_HIP(mde_sort_pairs_u32(nullptr, tmp_bytes, keys, keys2, vals, vals2, (int)H2, 0, end_bit, st));
    PLAN_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    PLAN_HIP(mde_sort_pairs_u32(tmp, tmp_bytes, keys, keys2, vals, vals2, (int)H2, 0, end_bit, st));
    hipLaunchKernelGGL(k_rowptr, dim3(mde_grid(H2 + 1, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, H2,
                       (uint32_t)nloc, keys2, plan->rowptr);
    PLAN_HIP(hipGetLastError());
    PLAN_HIP(hipMemcpyAsync(&Hlocal, plan->rowptr + nloc, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    PLAN_HIP(hipStreamSynchronize(st));
  } else {
    PLAN_HIP(hipMemsetAsync(plan->rowptr, 0, (nloc + 1) * sizeof(int32_t), st));
    PLAN_HIP(hipStreamSynchronize(st));
  }
  plan->H = Hlocal;
  plan->avg_degree = nloc > 0 ? (float)((double)Hlocal / (double)nloc) : 0.f;
  if (row_lo > 0 && p > 0) {
    // global position of the first local half-edge (the flat schedule aligns its tiles to it)
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(plan->partials);
    unsigned long long below = 0;
    PLAN_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(k_count_below, dim3(mde_grid(p, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, p, edges, row_lo, cnt);
    PLAN_HIP(hipGetLastError());
    PLAN_HIP(hipMemcpyAsync(&below, cnt, sizeof(below), hipMemcpyDeviceToHost, st));
    PLAN_HIP(hipStreamSynchronize(st));
    plan->h_offset = (int64_t)below;
  }
  const size_t hb = (size_t)(Hlocal > 0 ? Hlocal : 1) * sizeof(int32_t);
  PLAN_HIP(hipMalloc(&plan->nbr, hb));
  PLAN_HIP(hipMalloc(&plan->eid, hb));
  if (Hlocal > 0) {
    hipLaunchKernelGGL(k_fill_half_edges, dim3(mde_grid(Hlocal, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st,
                       (int64_t)Hlocal, vals2, edges, plan->nbr, plan->eid);
    PLAN_HIP(hipGetLastError());
  }
  PLAN_HIP(hipStreamSynchronize(st));
  cleanup();
#undef PLAN_HIP
  *out = plan;
  return MDE_OK;
}

// ---------------------------------------------------------------- processing order of the rows (round 6)
// The general-d kernel (mde_distortion.hip: k_fused_wide4p) gathers a 512-byte row of X per half-edge at d = 128; what
// it costs depends on where those rows come from.  When the caller's vertex numbering hides the graph's locality (a
// k-NN graph whose items arrive in arbitrary order), the rows a chip works on at the same moment reference neighbours
// all over the table and every gather goes to HBM.  mde_plan_row_order builds a PROCESSING order instead of moving any
// data: a breadth-first search over the local rows (components one after the other, from the lowest unvisited row),
// rows sorted by (level, row id) -- the neighbours of a row then sit in its own level or the two next to it, so the
// rows evaluated together share their neighbours through L2 / Infinity Cache.  The CSR arrays, X and the gradient
// keep the caller's numbering; the kernel only asks "which row is q-th".  Kept when the mean distance between the
// positions of an edge's two ends at least halves (mode 1), always (mode 2).
//
// One launch + one 4-byte read-back per level (a band graph of 500k rows and window 100 has ~5000 levels: ~0.1 s, once
// per plan); abandoned when a level holds more than an eighth of the rows (the graph expands like a random one: no
// order helps), after 256 components or 20000 levels (~0.5 s).
#define MDE_ORDER_UNSEEN (-1)
#define MDE_ORDER_ISOLATED 0x7ffffff0
__global__ void k_order_init(int32_t nloc, const int32_t* __restrict__ rowptr, int32_t* __restrict__ level,
                             uint32_t* __restrict__ iota) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nloc; r += (int64_t)gridDim.x * blockDim.x) {
    level[r] = rowptr[r] == rowptr[r + 1] ? MDE_ORDER_ISOLATED : MDE_ORDER_UNSEEN;
    iota[r] = (uint32_t)r;
  }
}
// state[0] = cursor (every row below it has been seen), state[1] = the row found or -1; that row gets `lvl`
__global__ void k_order_next(int32_t nloc, int32_t* __restrict__ level, int32_t lvl, int32_t* __restrict__ state,
                             int32_t* __restrict__ frontier) {
  __shared__ int found;
  int32_t c = state[0];
  for (;;) {
    if (threadIdx.x == 0) found = 0x7fffffff;
    __syncthreads();
    const int32_t r = c + (int32_t)threadIdx.x;
    if (r < nloc && level[r] == MDE_ORDER_UNSEEN) atomicMin(&found, r);
    __syncthreads();
    const int f = found;
    __syncthreads();
    if (f != 0x7fffffff) {
      if (threadIdx.x == 0) {
        level[f] = lvl;
        frontier[0] = f;
        state[0] = f + 1;
        state[1] = f;
      }
      return;
    }
    c += (int32_t)blockDim.x;
    if (c >= nloc) {
      if (threadIdx.x == 0) {
        state[0] = nloc;
        state[1] = -1;
      }
      return;
    }
  }
}
// one wave per frontier row: unseen local neighbours get `next_level` and join the next frontier
__global__ __launch_bounds__(MDE_BLOCK) void k_order_expand(int32_t m, const int32_t* __restrict__ frontier,
                                                            int32_t next_level, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ nbr, int64_t row_lo, int32_t nloc,
                                                            int32_t* __restrict__ level, int32_t* __restrict__ out,
                                                            int32_t* __restrict__ count) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * MDE_BLOCK) >> 6;
  for (int64_t w = w0; w < m; w += nw) {
    const int32_t v = frontier[w];
    const int32_t beg = rowptr[v], end = rowptr[v + 1];
    for (int32_t h0 = beg; h0 < end; h0 += 64) {
      const int32_t h = h0 + lane;
      bool won = false;
      int32_t u = 0;
      if (h < end) {
        const int64_t ug = (int64_t)nbr[h] - row_lo;
        if (ug >= 0 && ug < nloc) {
          u = (int32_t)ug;
          if (level[u] == MDE_ORDER_UNSEEN) won = atomicCAS(&level[u], MDE_ORDER_UNSEEN, next_level) == MDE_ORDER_UNSEEN;
        }
      }
      const unsigned long long mask = __ballot(won);
      if (mask) {
        int32_t base = 0;
        if (lane == 0) base = atomicAdd(count, (int32_t)__popcll(mask));
        base = __shfl(base, 0, 64);
        if (won) out[base + (int32_t)__popcll(mask & ((1ull << lane) - 1ull))] = u;
      }
    }
  }
}
__global__ void k_order_rank(int32_t nloc, const uint32_t* __restrict__ order, int32_t* __restrict__ rank) {
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nloc; q += (int64_t)gridDim.x * blockDim.x)
    rank[order[q]] = (int32_t)q;
}
// sums[0] += sum |v - u|, sums[1] += sum |rank[v] - rank[u]| over the half-edges with both ends local (exact: integers)
__global__ __launch_bounds__(MDE_BLOCK) void k_order_metric(int32_t nloc, int64_t row_lo, const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ nbr, const int32_t* __restrict__ rank,
                                                            unsigned long long* __restrict__ sums) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * MDE_BLOCK) >> 6;
  unsigned long long a = 0, b = 0, c = 0;
  for (int64_t v = w0; v < nloc; v += nw) {
    const int32_t beg = rowptr[v], end = rowptr[v + 1];
    const int32_t rv = rank ? rank[v] : 0;
    for (int32_t h = beg + lane; h < end; h += 64) {
      const int64_t u = (int64_t)nbr[h] - row_lo;
      if (u >= 0 && u < nloc) {
        a += (unsigned long long)(u > v ? u - v : v - u);
        if (rank) {
          const int32_t ru = rank[u];
          b += (unsigned long long)(ru > rv ? ru - rv : rv - ru);
        }
        c += 1;
      }
    }
  }
  a = mde_wave_sum(a);
  b = mde_wave_sum(b);
  c = mde_wave_sum(c);
  if (lane == 0 && c) {
    atomicAdd(&sums[0], a);
    atomicAdd(&sums[1], b);
    atomicAdd(&sums[2], c);
  }
}

extern "C" int mde_plan_row_order(mde_plan* plan, int32_t mode, void* stream, double* info_host) {
  if (!plan) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  auto report = [&]() {
    if (info_host) {
      info_host[0] = plan->order_state == 1 ? 1.0 : 0.0;
      info_host[1] = plan->order_before;
      info_host[2] = plan->order_after;
      info_host[3] = (double)plan->order_levels;
    }
  };
  if (mode == 0) {
    if (plan->order) {
      MDE_HIP(hipStreamSynchronize(st));
      (void)hipFree(plan->order);
      plan->order = nullptr;
    }
    plan->order_state = -1;
    report();
    return MDE_OK;
  }
  // (built before, or tried and rejected: nothing to do -- except that mode 2 overrides an earlier rejection)
  if (plan->order_state == 1 || (plan->order_state == -1 && mode < 2)) {
    report();
    return MDE_OK;
  }
  plan->order_state = -1;
  const int64_t nloc64 = plan->row_hi - plan->row_lo;
  // (a table of a few thousand rows sits in every L2 whatever the order)
  if (nloc64 < 8192 || plan->H <= 0) {
    report();
    return MDE_OK;
  }
  const int32_t nloc = (int32_t)nloc64;
  int32_t *level = nullptr, *fa = nullptr, *fb = nullptr, *state = nullptr, *rank = nullptr;
  uint32_t *iota = nullptr, *keys2 = nullptr, *ord = nullptr;
  unsigned long long* sums = nullptr;
  void* tmp = nullptr;
  auto cleanup = [&]() {
    for (void* q : {(void*)level, (void*)fa, (void*)fb, (void*)state, (void*)rank, (void*)iota, (void*)keys2, (void*)sums, tmp})
      if (q) (void)hipFree(q);
  };
#define ORD_HIP(call)                                             \
  do {                                                            \
    hipError_t e__ = (call);                                      \
    if (e__ != hipSuccess) {                                      \
      cleanup();                                                  \
      if (ord) (void)hipFree(ord);                                \
      return mde_hip_fail(e__, #call, __FILE__, __LINE__);        \
    }                                                             \
  } while (0)
  const size_t nb4 = (size_t)nloc * sizeof(int32_t);
  ORD_HIP(hipMalloc(&level, nb4));
  ORD_HIP(hipMalloc(&fa, nb4));
  ORD_HIP(hipMalloc(&fb, nb4));
  ORD_HIP(hipMalloc(&iota, nb4));
  ORD_HIP(hipMalloc(&state, 4 * sizeof(int32_t)));
  ORD_HIP(hipMalloc(&sums, 4 * sizeof(unsigned long long)));
  ORD_HIP(hipMemsetAsync(state, 0, 4 * sizeof(int32_t), st));
  ORD_HIP(hipMemsetAsync(sums, 0, 4 * sizeof(unsigned long long), st));
  hipLaunchKernelGGL(k_order_init, dim3(mde_grid(nloc, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, nloc, plan->rowptr, level, iota);
  ORD_HIP(hipGetLastError());
  int32_t lvl = 0, components = 0, hstate[4] = {0, 0, 0, 0};
  bool abandoned = false;
  int64_t launches = 0;
  for (;;) {
    // the lowest row nobody has seen starts the next component
    hipLaunchKernelGGL(k_order_next, dim3(1), dim3(1024), 0, st, nloc, level, lvl, state, fa);
    ORD_HIP(hipGetLastError());
    ORD_HIP(hipMemcpyAsync(hstate, state, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    ORD_HIP(hipStreamSynchronize(st));
    if (hstate[1] < 0) break;
    if (++components > 256) {
      abandoned = true;
      break;
    }
    int32_t m = 1;
    int32_t *cur = fa, *nxt = fb;
    while (m > 0) {
      ORD_HIP(hipMemsetAsync(state + 2, 0, sizeof(int32_t), st));
      hipLaunchKernelGGL(k_order_expand, dim3(mde_grid((int64_t)m * 64, MDE_BLOCK, 2048)), dim3(MDE_BLOCK), 0, st, m, cur,
                         lvl + 1, plan->rowptr, plan->nbr, plan->row_lo, nloc, level, nxt, state + 2);
      ORD_HIP(hipGetLastError());
      ORD_HIP(hipMemcpyAsync(&m, state + 2, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      ORD_HIP(hipStreamSynchronize(st));
      ++lvl;
      if (m > nloc / 8 || ++launches > 20000) {
        abandoned = true;
        break;
      }
      int32_t* t = cur;
      cur = nxt;
      nxt = t;
    }
    if (abandoned) break;
    // (lvl is now one past the component's last level: the next component starts there, in fa[0] again)
  }
  plan->order_levels = lvl;
  if (!abandoned) {
    // order = rows sorted by (level, row id): a stable sort of the levels with the row ids as values
    size_t tmp_bytes = 0;
    ORD_HIP(hipMalloc(&keys2, nb4));
    ORD_HIP(hipMalloc(&ord, nb4));
    ORD_HIP(hipMalloc(&rank, nb4));
    ORD_HIP(mde_sort_pairs_u32(nullptr, tmp_bytes, reinterpret_cast<uint32_t*>(level), keys2, iota, ord, nloc, 0, 31, st));
    ORD_HIP(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
    ORD_HIP(mde_sort_pairs_u32(tmp, tmp_bytes, reinterpret_cast<uint32_t*>(level), keys2, iota, ord, nloc, 0, 31, st));
    hipLaunchKernelGGL(k_order_rank, dim3(mde_grid(nloc, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, nloc, ord, rank);
    ORD_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_order_metric, dim3(mde_grid((int64_t)nloc * 64, MDE_BLOCK, 2048)), dim3(MDE_BLOCK), 0, st, nloc,
                     plan->row_lo, plan->rowptr, plan->nbr, rank, sums);
  ORD_HIP(hipGetLastError());
  unsigned long long hs[4] = {0, 0, 0, 0};
  ORD_HIP(hipMemcpyAsync(hs, sums, sizeof(hs), hipMemcpyDeviceToHost, st));
  ORD_HIP(hipStreamSynchronize(st));
  const double cnt = hs[2] ? (double)hs[2] : 1.0;
  plan->order_before = (double)hs[0] / cnt;
  plan->order_after = abandoned ? 0.0 : (double)hs[1] / cnt;
  if (!abandoned && (mode >= 2 || plan->order_after <= 0.5 * plan->order_before)) {
    plan->order = reinterpret_cast<int32_t*>(ord);
    plan->order_state = 1;
    ord = nullptr;
  }
  cleanup();
  if (ord) (void)hipFree(ord);
#undef ORD_HIP
  report();
  return MDE_OK;
}

extern "C" int mde_shard_bounds(int64_t n, int64_t p, const int64_t* edges, int32_t world,
                                int64_t* bounds_host, void* stream) {
  if (n <= 0 || p < 0 || world <= 0 || !bounds_host || (p > 0 && !edges)) return MDE_E_INVALID;
  if (n >= (int64_t)1 << 31 || 2 * p >= ((int64_t)1 << 31) - 1) return MDE_E_TOO_LARGE;
  hipStream_t st = mde_stream(stream);
  int rc = validate_edges(n, p, edges, st);
  if (rc != MDE_OK) return rc;
  int32_t *deg = nullptr, *cum = nullptr;
  int64_t* bounds = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  hipError_t e = hipMalloc(&deg, n * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&cum, n * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&bounds, (world + 1) * sizeof(int64_t));
  if (e == hipSuccess) e = hipMemsetAsync(deg, 0, n * sizeof(int32_t), st);
  if (e == hipSuccess && p > 0) {
    hipLaunchKernelGGL(k_degree, dim3(mde_grid(p, MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, p, edges, deg);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, deg, cum, (int)n, st);
  if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
  if (e == hipSuccess) e = hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, deg, cum, (int)n, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_bounds, dim3((world + 1 + 63) / 64), dim3(64), 0, st, n, world, cum, bounds);
    e = hipGetLastError();
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(bounds_host, bounds, (world + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (deg) (void)hipFree(deg);
  if (cum) (void)hipFree(cum);
  if (bounds) (void)hipFree(bounds);
  if (tmp) (void)hipFree(tmp);
  if (e != hipSuccess) return mde_hip_fail(e, "mde_shard_bounds", __FILE__, __LINE__);
  // monotone clean-up (degenerate graphs)
  for (int r = 1; r <= world; ++r)
    if (bounds_host[r] < bounds_host[r - 1]) bounds_host[r] = bounds_host[r - 1];
  bounds_host[world] = n;
  return MDE_OK;
}

extern "C" int mde_plan_expand(const mde_plan* plan, const float* in_edge, float* out_half,
                               void* stream) {
  if (!plan || !in_edge || !out_half) return MDE_E_INVALID;
  if (plan->H == 0) return MDE_OK;
  hipLaunchKernelGGL(k_expand, dim3(mde_grid(plan->H, MDE_BLOCK)), dim3(MDE_BLOCK), 0,
                     mde_stream(stream), plan->H, plan->eid, in_edge, out_half);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

extern "C" int mde_plan_export(const mde_plan* plan, int32_t* rowptr_out, int32_t* nbr_out,
                               int32_t* eid_out, void* stream) {
  if (!plan) return MDE_E_INVALID;
  hipStream_t st = mde_stream(stream);
  const int64_t nloc = plan->row_hi - plan->row_lo;
  if (rowptr_out)
    MDE_HIP(hipMemcpyAsync(rowptr_out, plan->rowptr, (nloc + 1) * sizeof(int32_t),
                           hipMemcpyDeviceToDevice, st));
  if (nbr_out && plan->H > 0)
    MDE_HIP(hipMemcpyAsync(nbr_out, plan->nbr, plan->H * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  if (eid_out && plan->H > 0)
    MDE_HIP(hipMemcpyAsync(eid_out, plan->eid, plan->H * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  return MDE_OK;
}

// one 16-lane group per row: out[v] = sum of w over the row's half-edges (fixed order)
__global__ __launch_bounds__(MDE_BLOCK) void k_weighted_degree(int nrows, int row_lo,
                                                               const int32_t* __restrict__ rowptr,
                                                               const float* __restrict__ w,
                                                               float* __restrict__ out) {
  constexpr int G = 16;
  const int lig = threadIdx.x & (G - 1);
  const int group = (blockIdx.x * MDE_BLOCK + threadIdx.x) / G;
  const int ngroups = (gridDim.x * MDE_BLOCK) / G;
  for (int r = group; r < nrows; r += ngroups) {
    float s = 0.0f;
    for (int h = rowptr[r] + lig; h < rowptr[r + 1]; h += G) s += w[h];
    s = mde_group_sum<G>(s);
    if (lig == 0) out[row_lo + r] = s;
  }
}
extern "C" int mde_weighted_degree(const mde_plan* plan, const float* w_plan_order, float* out,
                                   void* stream) {
  if (!plan || !w_plan_order || !out) return MDE_E_INVALID;
  const int64_t nloc = plan->row_hi - plan->row_lo;
  if (nloc <= 0) return MDE_OK;
  hipLaunchKernelGGL(k_weighted_degree, dim3(mde_grid(nloc * 16, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0,
                     mde_stream(stream), (int)nloc, (int)plan->row_lo, plan->rowptr, w_plan_order, out);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}
