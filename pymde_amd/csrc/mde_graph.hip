// mde_graph.hip -- shortest-path distances on a graph.
//   [ref: pymde/preprocess/graph.py:286-474 shortest_paths / _shortest_paths and the Cython BFS
//    pymde/preprocess/_graph.pyx:10-52 -- the reference's only native code]
// The reference runs one BFS (unit weights) or Dijkstra per node in a multiprocessing pool and
// keeps, for node i, the targets j > i at finite positive distance (<= max_length), each with
// probability retain_fraction.  Here a batch of B sources is solved at once on the device: the
// distance rows dist[b][.] live in HBM/L2 and are relaxed with a pull-style Bellman-Ford sweep
//      dist[b][u] = min(dist[b][u], min_{v ~ u} dist[b][v] + w_uv)
// over the symmetrised CSR of the edge plan (no atomics; in-place updates are safe because every
// value is always the length of a real path and only decreases).  With unit weights the sweep count
// is the hop eccentricity of the batch (a level-synchronous BFS); the fixed point is detected with
// a device flag checked every few sweeps.  Retained pairs are selected by a per-pair hash (so the
// outcome does not depend on the batching), compacted with wave-aggregated atomics and finally
// sorted by (i, j).
#include <hipcub/hipcub.hpp>

#include "mde_common.h"
#include "mde_plan.h"

#define MDE_INF_F 3.402823466e+38f

__device__ __forceinline__ uint64_t gsplitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ __launch_bounds__(MDE_BLOCK) void k_sp_init(int64_t B, int64_t n, int64_t src0, float* __restrict__ dist) {
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < B * n;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    const int64_t b = i / n, v = i % n;
    dist[i] = (v == src0 + b) ? 0.0f : MDE_INF_F;
  }
}

// one relaxation sweep over every (source b, vertex u)
__global__ __launch_bounds__(MDE_BLOCK) void k_sp_relax(int64_t B, int n, const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ nbr,
                                                        const float* __restrict__ w, float max_length,
                                                        float* __restrict__ dist, int* __restrict__ changed) {
  bool any = false;
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < B * (int64_t)n;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    const int64_t b = i / n;
    const int u = (int)(i % n);
    float* row = dist + b * (int64_t)n;
    const float cur = row[u];
    float best = cur;
    for (int h = rowptr[u]; h < rowptr[u + 1]; ++h) {
      const float dv = row[nbr[h]];
      if (dv < MDE_INF_F) {
        const float cand = dv + (w ? w[h] : 1.0f);
        best = cand < best ? cand : best;
      }
    }
    if (best < cur && best <= max_length) {
      row[u] = best;
      any = true;
    }
  }
  if (__any(any) && (threadIdx.x & 63) == 0) *changed = 1;
}

// keep (i = src0 + b, j) with j > i, 0 < dist < inf, hash(i, j) < threshold
__global__ __launch_bounds__(MDE_BLOCK) void k_sp_emit(int64_t B, int64_t n, int64_t src0,
                                                       const float* __restrict__ dist, uint64_t seed,
                                                       uint64_t threshold, int keep_all, int64_t capacity,
                                                       unsigned long long* __restrict__ counter,
                                                       uint64_t* __restrict__ keys, float* __restrict__ vals) {
  for (int64_t i0 = (int64_t)blockIdx.x * MDE_BLOCK; i0 < B * n; i0 += (int64_t)gridDim.x * MDE_BLOCK) {
    const int64_t i = i0 + threadIdx.x;
    bool keep = false;
    uint64_t key = 0;
    float d = 0.0f;
    if (i < B * n) {
      const int64_t src = src0 + i / n, j = i % n;
      d = dist[i];
      if (j > src && d > 0.0f && d < MDE_INF_F) {
        key = (uint64_t)src * (uint64_t)n + (uint64_t)j;
        keep = keep_all || gsplitmix64(seed ^ gsplitmix64(key)) < threshold;
      }
    }
    // wave-aggregated slot reservation
    const unsigned long long mask = __ballot(keep);
    const int lane = threadIdx.x & 63;
    unsigned long long base = 0;
    if (mask) {
      const int leader = __ffsll((long long)mask) - 1;
      if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(mask));
      base = __shfl(base, leader, 64);
      if (keep) {
        const unsigned long long pos = base + __popcll(mask & ((1ull << lane) - 1ull));
        if ((int64_t)pos < capacity) {
          keys[pos] = key;
          vals[pos] = d;
        }
      }
    }
  }
}

__global__ __launch_bounds__(MDE_BLOCK) void k_sp_unpack(int64_t n, int64_t m, const uint64_t* __restrict__ keys,
                                                         int64_t* __restrict__ edges) {
  for (int64_t k = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; k < m;
       k += (int64_t)gridDim.x * MDE_BLOCK) {
    const uint64_t key = keys[k];
    reinterpret_cast<longlong2*>(edges)[k] = make_longlong2((long long)(key / (uint64_t)n),
                                                             (long long)(key % (uint64_t)n));
  }
}

// ---- unit lengths: bit-parallel multi-source BFS [ref: pymde/preprocess/_graph.pyx:10-52, one BFS per
// source there].  64 sources share a machine word: per vertex v and word wi, frontier / visited hold
// which of the sources 64 wi .. 64 wi + 63 have v on their current level / have reached v at all.  One
// level for ALL sources of a batch is
//      next[v] = (OR over neighbours u of frontier[u]) & ~visited[v];  visited[v] |= next[v]
// i.e. O(half-edges x words) word operations instead of the O(half-edges x sources) relaxations of
// the Bellman-Ford sweep above (config 3's 40k-node graph: 72 sweeps x 38 ms -> see DESIGN.md section 6).
// A source's newly reached vertices get their hop count written into the same dist[b][v] matrix the
// emit / top-k kernels read.
__global__ __launch_bounds__(MDE_BLOCK) void k_bfs_seed(int64_t Bc, int W, int64_t src0, uint64_t* __restrict__ visited,
                                                        uint64_t* __restrict__ frontier) {
  const int64_t b = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (b >= Bc) return;
  // (source b IS vertex src0 + b: no two sources share a word of one vertex)
  const int64_t at = (src0 + b) * W + (b >> 6);
  visited[at] = 1ull << (b & 63);
  frontier[at] = 1ull << (b & 63);
}

__global__ __launch_bounds__(MDE_BLOCK) void k_bfs_level(int64_t n, int W, int64_t Bc, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ nbr,
                                                         const uint64_t* __restrict__ frontier,
                                                         uint64_t* __restrict__ visited, uint64_t* __restrict__ next,
                                                         float* __restrict__ dist, float level, int* __restrict__ changed) {
  bool any = false;
  for (int64_t t = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; t < n * W; t += (int64_t)gridDim.x * MDE_BLOCK) {
    const int64_t v = t / W;
    const int wi = (int)(t - v * W);
    uint64_t acc = 0;
    for (int h = rowptr[v]; h < rowptr[v + 1]; ++h) acc |= frontier[(int64_t)nbr[h] * W + wi];
    uint64_t nx = acc & ~visited[t];
    next[t] = nx;
    if (nx) {
      visited[t] |= nx;
      any = true;
      while (nx) {
        const int bit = __ffsll((long long)nx) - 1;
        nx &= nx - 1;
        const int64_t b = (int64_t)wi * 64 + bit;
        if (b < Bc) dist[b * n + v] = level;
      }
    }
  }
  if (__any(any) && (threadIdx.x & 63) == 0) *changed = 1;
}

struct GBuf {
  void* p = nullptr;
  ~GBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T>
  T* as() {
    return reinterpret_cast<T*>(p);
  }
};

// one-hop lengths: dist[b][v] = min over parallel edges (src0 + b) ~ v of w  (one wave per source)
__global__ __launch_bounds__(MDE_BLOCK) void k_sp_direct(int64_t B, int64_t n, int64_t src0,
                                                         const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ nbr,
                                                         const float* __restrict__ w, float max_length,
                                                         float* __restrict__ dist) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * MDE_BLOCK) >> 6;
  for (int64_t b = w0; b < B; b += nw) {
    const int64_t u = src0 + b;
    float* row = dist + b * n;
    // parallel edges of one row are rare; a wave walks the row serially per lane stride and
    // resolves collisions with an integer atomic min on the (non-negative) float bits
    for (int h = rowptr[u] + lane; h < rowptr[u + 1]; h += 64) {
      const float len = w ? w[h] : 1.0f;
      if (len <= max_length && nbr[h] != u)
        atomicMin(reinterpret_cast<unsigned int*>(row + nbr[h]), __float_as_uint(len));
    }
  }
}

// k nearest targets of every source of the batch by (distance, index): one workgroup per source,
// k rounds of a block-wide lexicographic arg-min over the distance row (fixed tie-break: index).
__global__ __launch_bounds__(MDE_BLOCK) void k_sp_topk(int64_t n, int64_t src0, const float* __restrict__ dist,
                                                       int k, int32_t* __restrict__ idx_out,
                                                       float* __restrict__ dist_out) {
  __shared__ float sd[MDE_BLOCK / 64];
  __shared__ int sj[MDE_BLOCK / 64];
  __shared__ float bd;
  __shared__ int bj;
  const int64_t b = blockIdx.x, src = src0 + b;
  const float* row = dist + b * n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float last_d = -1.0f;
  int last_j = -1;
  for (int r = 0; r < k; ++r) {
    float best = MDE_INF_F;
    int bestj = 0x7fffffff;
    for (int64_t j = tid; j < n; j += MDE_BLOCK) {
      const float d = row[j];
      const bool ok = j != src && d > 0.0f && d < MDE_INF_F &&
                      (d > last_d || (d == last_d && (int)j > last_j));
      if (ok && (d < best || (d == best && (int)j < bestj))) {
        best = d;
        bestj = (int)j;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float od = __shfl_xor(best, off, 64);
      const int oj = __shfl_xor(bestj, off, 64);
      if (od < best || (od == best && oj < bestj)) {
        best = od;
        bestj = oj;
      }
    }
    if (lane == 0) {
      sd[wave] = best;
      sj[wave] = bestj;
    }
    __syncthreads();
    if (tid == 0) {
      float d0 = sd[0];
      int j0 = sj[0];
      for (int q = 1; q < MDE_BLOCK / 64; ++q)
        if (sd[q] < d0 || (sd[q] == d0 && sj[q] < j0)) {
          d0 = sd[q];
          j0 = sj[q];
        }
      bd = d0;
      bj = j0;
      const bool found = d0 < MDE_INF_F;
      idx_out[src * k + r] = found ? j0 : -1;
      dist_out[src * k + r] = found ? d0 : MDE_INF_F;
    }
    __syncthreads();
    last_d = bd;
    last_j = bj;
    if (!(last_d < MDE_INF_F)) {
      // fewer than k reachable targets: the remaining slots are empty
      for (int q = r + 1 + tid; q < k; q += MDE_BLOCK) {
        idx_out[src * k + q] = -1;
        dist_out[src * k + q] = MDE_INF_F;
      }
      break;
    }
    __syncthreads();
  }
}

// dist[b][.] = shortest-path lengths from source src0 + b, b < Bc (relaxed to the fixed point).
// direct != 0: one-hop lengths only (the graph's own edge lengths; everything else stays INF).
// masks: 3 x n x ceil(B / 64) words for the unit-length BFS (NULL: always the Bellman-Ford sweeps).
static int sp_solve_batch(const mde_plan* plan, const float* w, float max_length, int direct, int64_t src0,
                          int64_t Bc, float* dist, int* changed, int nblk, hipStream_t st, uint64_t* masks = nullptr,
                          int W = 0) {
  const int64_t n = plan->n;
  hipLaunchKernelGGL(k_sp_init, dim3(nblk), dim3(MDE_BLOCK), 0, st, Bc, n, src0, dist);
  MDE_LAUNCH_CHECK();
  if (!direct && !w && masks && W > 0) {
    const size_t words = (size_t)n * W;
    uint64_t *visited = masks, *fr = masks + words, *nx = masks + 2 * words;
    MDE_HIP(hipMemsetAsync(visited, 0, 2 * words * sizeof(uint64_t), st));
    hipLaunchKernelGGL(k_bfs_seed, dim3((unsigned)((Bc + MDE_BLOCK - 1) / MDE_BLOCK)), dim3(MDE_BLOCK), 0, st, Bc, W, src0, visited,
                       fr);
    MDE_LAUNCH_CHECK();
    const int nb = mde_grid((int64_t)words, MDE_BLOCK, 16384);
    float level = 1.0f;
    for (int64_t it = 0; it < n + 8 && level <= max_length; it += 4) {
      MDE_HIP(hipMemsetAsync(changed, 0, sizeof(int), st));
      for (int s4 = 0; s4 < 4 && level <= max_length; ++s4) {
        hipLaunchKernelGGL(k_bfs_level, dim3(nb), dim3(MDE_BLOCK), 0, st, n, W, Bc, plan->rowptr, plan->nbr, fr, visited, nx, dist,
                           level, changed);
        MDE_LAUNCH_CHECK();
        uint64_t* t = fr;
        fr = nx;
        nx = t;
        level += 1.0f;
      }
      int h = 0;
      MDE_HIP(hipMemcpyAsync(&h, changed, sizeof(int), hipMemcpyDeviceToHost, st));
      MDE_HIP(hipStreamSynchronize(st));
      if (!h) break;
    }
    return MDE_OK;
  }
  if (direct) {
    hipLaunchKernelGGL(k_sp_direct, dim3(mde_grid(Bc * 64, MDE_BLOCK, 8192)), dim3(MDE_BLOCK), 0, st, Bc, n,
                       src0, plan->rowptr, plan->nbr, w, max_length, dist);
    MDE_LAUNCH_CHECK();
    return MDE_OK;
  }
  for (int64_t sweep = 0; sweep < 4 * n + 8; sweep += 4) {
    MDE_HIP(hipMemsetAsync(changed, 0, sizeof(int), st));
    for (int s4 = 0; s4 < 4; ++s4) {
      hipLaunchKernelGGL(k_sp_relax, dim3(nblk), dim3(MDE_BLOCK), 0, st, Bc, (int)n, plan->rowptr, plan->nbr, w,
                         max_length, dist, changed);
      MDE_LAUNCH_CHECK();
    }
    int h = 0;
    MDE_HIP(hipMemcpyAsync(&h, changed, sizeof(int), hipMemcpyDeviceToHost, st));
    MDE_HIP(hipStreamSynchronize(st));
    if (!h) break;
  }
  return MDE_OK;
}

// plan: a FULL plan of the graph's edges (its symmetrised CSR is the adjacency); w: per-half-edge
// edge lengths in plan (CSR) order, or NULL for unit lengths.  Writes at most `capacity` pairs
// (i < j, sorted by (i, j)) with their shortest-path distances; *count_host = number of pairs (if it
// exceeds `capacity` the call fails with MDE_E_INVALID and reports the needed size).  SYNC.
extern "C" int mde_graph_shortest_paths(const mde_plan* plan, const float* w, float max_length,
                                        double retain_fraction, uint64_t seed, int64_t capacity,
                                        int64_t* edges_out, float* dist_out, int64_t* count_host,
                                        void* stream) {
  if (!plan || !edges_out || !dist_out || !count_host || capacity < 0) return MDE_E_INVALID;
  if (plan->row_lo != 0 || plan->row_hi != plan->n) {
    mde_set_error("mde_graph_shortest_paths needs a full (unsharded) plan");
    return MDE_E_INVALID;
  }
  hipStream_t st = mde_stream(stream);
  const int64_t n = plan->n;
  *count_host = 0;
  if (capacity >= ((int64_t)1 << 31) - 1) return MDE_E_TOO_LARGE;
  if (!(max_length > 0.0f)) max_length = MDE_INF_F;
  const int keep_all = retain_fraction >= 1.0;
  uint64_t threshold = ~0ull;
  if (!keep_all) {
    const double t = retain_fraction <= 0.0 ? 0.0 : retain_fraction * 18446744073709551616.0;
    threshold = t >= 18446744073709549568.0 ? ~0ull : (uint64_t)t;
  }
  // batch of sources: about 1 GiB of distance rows
  int64_t B = ((int64_t)1 << 28) / (n > 0 ? n : 1);
  if (B < 1) B = 1;
  if (B > n) B = n;
  GBuf dist, keys, vals, counter, changed, masks;
  const int W = (int)((B + 63) / 64);
  if (!w) MDE_HIP(masks.alloc(3 * (size_t)n * W * sizeof(uint64_t)));
  MDE_HIP(dist.alloc((size_t)B * n * sizeof(float)));
  MDE_HIP(keys.alloc((size_t)(capacity + 1) * sizeof(uint64_t)));
  MDE_HIP(vals.alloc((size_t)(capacity + 1) * sizeof(float)));
  MDE_HIP(counter.alloc(sizeof(unsigned long long)));
  MDE_HIP(changed.alloc(sizeof(int)));
  MDE_HIP(hipMemsetAsync(counter.p, 0, sizeof(unsigned long long), st));
  const int nblk = mde_grid(B * n, MDE_BLOCK, 8192);
  for (int64_t src0 = 0; src0 < n; src0 += B) {
    const int64_t Bc = (src0 + B <= n) ? B : n - src0;
    const int rc = sp_solve_batch(plan, w, max_length, 0, src0, Bc, dist.as<float>(), changed.as<int>(), nblk, st,
                                  w ? nullptr : masks.as<uint64_t>(), W);
    if (rc != MDE_OK) return rc;
    hipLaunchKernelGGL(k_sp_emit, dim3(nblk), dim3(MDE_BLOCK), 0, st, Bc, n, src0, dist.as<float>(), seed, threshold,
                       keep_all, capacity, counter.as<unsigned long long>(), keys.as<uint64_t>(), vals.as<float>());
    MDE_LAUNCH_CHECK();
  }
  unsigned long long total = 0;
  MDE_HIP(hipMemcpyAsync(&total, counter.p, sizeof(total), hipMemcpyDeviceToHost, st));
  MDE_HIP(hipStreamSynchronize(st));
  *count_host = (int64_t)total;
  if ((int64_t)total > capacity) {
    mde_set_error("shortest paths: %lld pairs retained but the output holds %lld", (long long)total,
                  (long long)capacity);
    return MDE_E_INVALID;
  }
  if (total == 0) return MDE_OK;
  // sort by (i, j): the emission order depends on atomic arbitration, the result must not
  GBuf keys2, tmp;
  MDE_HIP(keys2.alloc((size_t)total * sizeof(uint64_t)));
  size_t tb = 0;
  int end_bit = 1;
  while (end_bit < 64 && (((uint64_t)n * (uint64_t)n) >> end_bit)) ++end_bit;
  MDE_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys.as<uint64_t>(), keys2.as<uint64_t>(), vals.as<float>(),
                                             dist_out, (int)total, 0, end_bit, st));
  MDE_HIP(tmp.alloc(tb));
  MDE_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, keys.as<uint64_t>(), keys2.as<uint64_t>(), vals.as<float>(),
                                             dist_out, (int)total, 0, end_bit, st));
  hipLaunchKernelGGL(k_sp_unpack, dim3(mde_grid((int64_t)total, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, n,
                     (int64_t)total, keys2.as<uint64_t>(), edges_out);
  MDE_LAUNCH_CHECK();
  MDE_HIP(hipStreamSynchronize(st));
  return MDE_OK;
}

// k nearest neighbours of every vertex under the shortest-path metric of the graph (direct == 0) or
// among its graph neighbours by edge length (direct != 0)  [ref: preprocess/graph.py:502-587].
// idx_out [n, k] int32 (-1 where fewer than k targets lie within max_length), dist_out [n, k]
// ascending per row; ties are broken by the smaller index.  SYNC.
extern "C" int mde_graph_knn(const mde_plan* plan, const float* w, float max_length, int32_t direct, int32_t k,
                             int32_t* idx_out, float* dist_out, void* stream) {
  if (!plan || !idx_out || !dist_out || k <= 0) return MDE_E_INVALID;
  if (plan->row_lo != 0 || plan->row_hi != plan->n) {
    mde_set_error("mde_graph_knn needs a full (unsharded) plan");
    return MDE_E_INVALID;
  }
  hipStream_t st = mde_stream(stream);
  const int64_t n = plan->n;
  if (n >= ((int64_t)1 << 31)) return MDE_E_TOO_LARGE;
  if (!(max_length > 0.0f)) max_length = MDE_INF_F;
  int64_t B = ((int64_t)1 << 28) / (n > 0 ? n : 1);
  if (B < 1) B = 1;
  if (B > n) B = n;
  GBuf dist, changed, masks;
  const int W = (int)((B + 63) / 64);
  if (!w && !direct) MDE_HIP(masks.alloc(3 * (size_t)n * W * sizeof(uint64_t)));
  MDE_HIP(dist.alloc((size_t)B * n * sizeof(float)));
  MDE_HIP(changed.alloc(sizeof(int)));
  const int nblk = mde_grid(B * n, MDE_BLOCK, 8192);
  for (int64_t src0 = 0; src0 < n; src0 += B) {
    const int64_t Bc = (src0 + B <= n) ? B : n - src0;
    const int rc = sp_solve_batch(plan, w, max_length, direct, src0, Bc, dist.as<float>(), changed.as<int>(), nblk, st,
                                  (!w && !direct) ? masks.as<uint64_t>() : nullptr, W);
    if (rc != MDE_OK) return rc;
    hipLaunchKernelGGL(k_sp_topk, dim3((unsigned)Bc), dim3(MDE_BLOCK), 0, st, n, src0, dist.as<float>(), k, idx_out,
                       dist_out);
    MDE_LAUNCH_CHECK();
  }
  MDE_HIP(hipStreamSynchronize(st));
  return MDE_OK;
}
