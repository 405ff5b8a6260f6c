// mde_edges.hip -- edge-list preprocessing on the device:
// de-duplication and uniform negative-edge sampling with exclusion.
//   [ref: pymde/preprocess/preprocess.py:11-80 sample_edges, :83-113 dissimilar_edges,
//         :116-129 deduplicate_edges]
// The reference does this on the host with np.unique(axis=0) over the concatenated edge lists and
// Generator.choice(C(n,2), replace=False); at 5e7 edges that takes longer than the whole solve.
// Here an edge (i < j) is one 64-bit key i * n + j: de-duplication is a radix sort + unique
// (rocPRIM), exclusion a binary search against the sorted excluded keys, sampling a counter-based
// generator (one SplitMix64 stream per draw) mapped through the triangular-number bijection the
// reference uses (preprocess.py:64-69).  Output is sorted by (i, j) -- np.unique's order.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "mde_common.h"

__device__ __forceinline__ uint64_t mde_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// canonical key of an edge: (min, max) -> min * n + max
__global__ __launch_bounds__(MDE_BLOCK) void k_edge_keys(int64_t n, int64_t p, const int64_t* __restrict__ edges,
                                                         uint64_t* __restrict__ keys) {
  for (int64_t k = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; k < p;
       k += (int64_t)gridDim.x * MDE_BLOCK) {
    const longlong2 e = reinterpret_cast<const longlong2*>(edges)[k];
    const int64_t a = e.x < e.y ? e.x : e.y, b = e.x < e.y ? e.y : e.x;
    keys[k] = (uint64_t)a * (uint64_t)n + (uint64_t)b;
  }
}

__global__ __launch_bounds__(MDE_BLOCK) void k_keys_to_edges(int64_t n, int64_t m, const uint64_t* __restrict__ keys,
                                                             int64_t* __restrict__ edges) {
  for (int64_t k = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; k < m;
       k += (int64_t)gridDim.x * MDE_BLOCK) {
    const uint64_t key = keys[k];
    reinterpret_cast<longlong2*>(edges)[k] = make_longlong2((long long)(key / (uint64_t)n),
                                                             (long long)(key % (uint64_t)n));
  }
}

// draw t -> uniform index in [0, C(n,2)) -> edge (u < v) by the triangular bijection
//   u = n - 2 - floor(sqrt(-8 idx + 4 n (n-1) - 7) / 2 - 1/2),
//   v = idx + u + 1 - n(n-1)/2 + (n-u)(n-u-1)/2                     (preprocess.py:64-69)
// evaluated in integers with a +-1 correction of the double-precision square root.
__global__ __launch_bounds__(MDE_BLOCK) void k_sample_keys(int64_t n, int64_t draws, uint64_t seed,
                                                           uint64_t* __restrict__ keys) {
  const uint64_t total = (uint64_t)n * (uint64_t)(n - 1) / 2;
  for (int64_t t = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; t < draws;
       t += (int64_t)gridDim.x * MDE_BLOCK) {
    // 64 random bits -> index by multiply-shift (bias < total / 2^64)
    const uint64_t r = mde_splitmix64(seed ^ mde_splitmix64((uint64_t)t));
    const uint64_t idx = __umul64hi(r, total);  // floor(r * total / 2^64)
    // row u: largest u with off(u) = u n - u (u+1)/2 <= idx
    const double disc = 4.0 * (double)n * (double)(n - 1) - 8.0 * (double)idx - 7.0;
    int64_t u = (int64_t)n - 2 - (int64_t)floor(sqrt(disc) / 2.0 - 0.5);
    if (u < 0) u = 0;
    if (u > n - 2) u = n - 2;
    auto off = [n](int64_t uu) { return (uint64_t)uu * (uint64_t)n - (uint64_t)uu * (uint64_t)(uu + 1) / 2; };
    while (u > 0 && off(u) > idx) --u;
    while (u < n - 2 && off(u + 1) <= idx) ++u;
    const uint64_t v = idx - off(u) + (uint64_t)u + 1;
    keys[t] = (uint64_t)u * (uint64_t)n + v;
  }
}

// flags[k] = 1 when keys[k] is NOT in the sorted array excl[0..m)
__global__ __launch_bounds__(MDE_BLOCK) void k_not_excluded(int64_t cnt, const uint64_t* __restrict__ keys,
                                                            int64_t m, const uint64_t* __restrict__ excl,
                                                            uint8_t* __restrict__ flags) {
  for (int64_t k = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; k < cnt;
       k += (int64_t)gridDim.x * MDE_BLOCK) {
    const uint64_t key = keys[k];
    int64_t lo = 0, hi = m;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (excl[mid] < key)
        lo = mid + 1;
      else
        hi = mid;
    }
    flags[k] = (lo < m && excl[lo] == key) ? 0 : 1;
  }
}

static int bits_for_key(uint64_t maxval) {
  int b = 1;
  while (b < 64 && (maxval >> b)) ++b;
  return b;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T>
  T* as() {
    return reinterpret_cast<T*>(p);
  }
};

// sort + unique of `cnt` keys in `keys` (clobbered); result in `out`, count returned via *m_out. SYNC.
static int sort_unique(uint64_t* keys, int64_t cnt, uint64_t max_key, uint64_t* out, int64_t* m_out,
                       hipStream_t st) {
  if (cnt == 0) {
    *m_out = 0;
    return MDE_OK;
  }
  DevBuf sorted, tmp, num;
  MDE_HIP(sorted.alloc(cnt * sizeof(uint64_t)));
  MDE_HIP(num.alloc(sizeof(int64_t)));
  size_t tb = 0;
  const int end_bit = bits_for_key(max_key);
  MDE_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, keys, sorted.as<uint64_t>(), (int)cnt, 0, end_bit, st));
  MDE_HIP(tmp.alloc(tb));
  MDE_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tb, keys, sorted.as<uint64_t>(), (int)cnt, 0, end_bit, st));
  size_t tb2 = 0;
  MDE_HIP(hipcub::DeviceSelect::Unique(nullptr, tb2, sorted.as<uint64_t>(), out, num.as<int64_t>(), (int)cnt, st));
  DevBuf tmp2;
  MDE_HIP(tmp2.alloc(tb2));
  MDE_HIP(hipcub::DeviceSelect::Unique(tmp2.p, tb2, sorted.as<uint64_t>(), out, num.as<int64_t>(), (int)cnt, st));
  MDE_HIP(hipMemcpyAsync(m_out, num.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  MDE_HIP(hipStreamSynchronize(st));
  return MDE_OK;
}

static int check_sizes(int64_t n, int64_t p) {
  if (n < 2 || p < 0) return MDE_E_INVALID;
  if (p >= ((int64_t)1 << 31) - 1 || n >= ((int64_t)1 << 31)) {
    mde_set_error("edge preprocessing supports up to 2^31 - 1 edges and n < 2^31");
    return MDE_E_TOO_LARGE;
  }
  return MDE_OK;
}

// edges_out [>= p, 2] receives the unique edges with i < j, sorted by (i, j); *count_host their
// number.  Self edges are kept as (i, i) like np.unique would.  SYNC.
extern "C" int mde_edges_deduplicate(int64_t n, int64_t p, const int64_t* edges, int64_t* edges_out,
                                     int64_t* count_host, void* stream) {
  if (!edges_out || !count_host || (p > 0 && !edges)) return MDE_E_INVALID;
  int rc = check_sizes(n, p);
  if (rc != MDE_OK) return rc;
  hipStream_t st = mde_stream(stream);
  *count_host = 0;
  if (p == 0) return MDE_OK;
  DevBuf keys, uniq;
  MDE_HIP(keys.alloc(p * sizeof(uint64_t)));
  MDE_HIP(uniq.alloc(p * sizeof(uint64_t)));
  hipLaunchKernelGGL(k_edge_keys, dim3(mde_grid(p, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, n, p, edges,
                     keys.as<uint64_t>());
  MDE_LAUNCH_CHECK();
  int64_t m = 0;
  rc = sort_unique(keys.as<uint64_t>(), p, (uint64_t)n * (uint64_t)n, uniq.as<uint64_t>(), &m, st);
  if (rc != MDE_OK) return rc;
  hipLaunchKernelGGL(k_keys_to_edges, dim3(mde_grid(m, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, n, m,
                     uniq.as<uint64_t>(), edges_out);
  MDE_LAUNCH_CHECK();
  MDE_HIP(hipStreamSynchronize(st));
  *count_host = m;
  return MDE_OK;
}

// Unique edges (i < j, sorted by (i, j)) with their multiplicity as a float weight: duplicated
// edges have their unit weights summed, as Graph.from_edges does [ref: preprocess/graph.py:51-72,
// data_matrix.py:170-178] (a mutual k-NN pair gets weight 2).  Rows with i == j are dropped.  SYNC.
__global__ __launch_bounds__(MDE_BLOCK) void k_runs_to_edges(int64_t n, int64_t m, const uint64_t* __restrict__ keys,
                                                             const int32_t* __restrict__ counts,
                                                             int64_t* __restrict__ edges, float* __restrict__ w) {
  for (int64_t k = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; k < m;
       k += (int64_t)gridDim.x * MDE_BLOCK) {
    const uint64_t key = keys[k];
    reinterpret_cast<longlong2*>(edges)[k] = make_longlong2((long long)(key / (uint64_t)n),
                                                             (long long)(key % (uint64_t)n));
    w[k] = (float)counts[k];
  }
}
__global__ __launch_bounds__(MDE_BLOCK) void k_flag_valid_pairs(int64_t p, const int64_t* __restrict__ edges,
                                                                uint8_t* __restrict__ flags) {
  for (int64_t k = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; k < p;
       k += (int64_t)gridDim.x * MDE_BLOCK) {
    const longlong2 e = reinterpret_cast<const longlong2*>(edges)[k];
    flags[k] = (e.x != e.y && e.x >= 0 && e.y >= 0) ? 1 : 0;
  }
}

extern "C" int mde_edges_count_unique(int64_t n, int64_t p, const int64_t* edges, int64_t* edges_out,
                                      float* weights_out, int64_t* count_host, void* stream) {
  if (!edges_out || !weights_out || !count_host || (p > 0 && !edges)) return MDE_E_INVALID;
  int rc = check_sizes(n, p);
  if (rc != MDE_OK) return rc;
  hipStream_t st = mde_stream(stream);
  *count_host = 0;
  if (p == 0) return MDE_OK;
  DevBuf keys, flags, kept, sorted, uniq, counts, num, tmp;
  MDE_HIP(keys.alloc(p * sizeof(uint64_t)));
  MDE_HIP(kept.alloc(p * sizeof(uint64_t)));
  MDE_HIP(flags.alloc(p));
  MDE_HIP(num.alloc(sizeof(int64_t)));
  hipLaunchKernelGGL(k_edge_keys, dim3(mde_grid(p, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, n, p, edges,
                     keys.as<uint64_t>());
  MDE_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_flag_valid_pairs, dim3(mde_grid(p, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, p, edges,
                     flags.as<uint8_t>());
  MDE_LAUNCH_CHECK();
  size_t tb = 0;
  MDE_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb, keys.as<uint64_t>(), flags.as<uint8_t>(), kept.as<uint64_t>(),
                                        num.as<int64_t>(), (int)p, st));
  MDE_HIP(tmp.alloc(tb));
  MDE_HIP(hipcub::DeviceSelect::Flagged(tmp.p, tb, keys.as<uint64_t>(), flags.as<uint8_t>(), kept.as<uint64_t>(),
                                        num.as<int64_t>(), (int)p, st));
  int64_t pv = 0;
  MDE_HIP(hipMemcpyAsync(&pv, num.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  MDE_HIP(hipStreamSynchronize(st));
  if (pv == 0) return MDE_OK;
  MDE_HIP(sorted.alloc(pv * sizeof(uint64_t)));
  MDE_HIP(uniq.alloc(pv * sizeof(uint64_t)));
  MDE_HIP(counts.alloc(pv * sizeof(int32_t)));
  size_t tb2 = 0;
  const int end_bit = bits_for_key((uint64_t)n * (uint64_t)n);
  DevBuf tmp2, tmp3;
  MDE_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb2, kept.as<uint64_t>(), sorted.as<uint64_t>(), (int)pv, 0,
                                            end_bit, st));
  MDE_HIP(tmp2.alloc(tb2));
  MDE_HIP(hipcub::DeviceRadixSort::SortKeys(tmp2.p, tb2, kept.as<uint64_t>(), sorted.as<uint64_t>(), (int)pv, 0,
                                            end_bit, st));
  size_t tb3 = 0;
  MDE_HIP(hipcub::DeviceRunLengthEncode::Encode(nullptr, tb3, sorted.as<uint64_t>(), uniq.as<uint64_t>(),
                                                counts.as<int32_t>(), num.as<int64_t>(), (int)pv, st));
  MDE_HIP(tmp3.alloc(tb3));
  MDE_HIP(hipcub::DeviceRunLengthEncode::Encode(tmp3.p, tb3, sorted.as<uint64_t>(), uniq.as<uint64_t>(),
                                                counts.as<int32_t>(), num.as<int64_t>(), (int)pv, st));
  int64_t m = 0;
  MDE_HIP(hipMemcpyAsync(&m, num.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  MDE_HIP(hipStreamSynchronize(st));
  hipLaunchKernelGGL(k_runs_to_edges, dim3(mde_grid(m, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, n, m,
                     uniq.as<uint64_t>(), counts.as<int32_t>(), edges_out, weights_out);
  MDE_LAUNCH_CHECK();
  MDE_HIP(hipStreamSynchronize(st));
  *count_host = m;
  return MDE_OK;
}

// Sample (at most) `num_edges` distinct edges i < j uniformly from the complement of `exclude`
// [n_exclude, 2] (may be NULL).  edges_out must hold num_edges rows.  SYNC.
extern "C" int mde_sample_edges(int64_t n, int64_t num_edges, uint64_t seed, const int64_t* exclude,
                                int64_t n_exclude, int64_t* edges_out, int64_t* count_host, void* stream) {
  if (!edges_out || !count_host || num_edges < 0 || n_exclude < 0 || (n_exclude > 0 && !exclude))
    return MDE_E_INVALID;
  int rc = check_sizes(n, num_edges + num_edges / 16 + 1024);
  if (rc == MDE_OK) rc = check_sizes(n, n_exclude);
  if (rc != MDE_OK) return rc;
  const double all_edges = (double)n * (double)(n - 1) / 2.0;
  if ((double)num_edges > all_edges - (double)n_exclude) {
    mde_set_error("Cannot sample more than (%lld choose 2) - %lld edges (requested: %lld edges)",
                  (long long)n, (long long)n_exclude, (long long)num_edges);
    return MDE_E_INVALID;
  }
  hipStream_t st = mde_stream(stream);
  *count_host = 0;
  if (num_edges == 0) return MDE_OK;
  // sorted unique excluded keys
  DevBuf ekeys, euniq;
  int64_t me = 0;
  if (n_exclude > 0) {
    MDE_HIP(ekeys.alloc(n_exclude * sizeof(uint64_t)));
    MDE_HIP(euniq.alloc(n_exclude * sizeof(uint64_t)));
    hipLaunchKernelGGL(k_edge_keys, dim3(mde_grid(n_exclude, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, n,
                       n_exclude, exclude, ekeys.as<uint64_t>());
    MDE_LAUNCH_CHECK();
    rc = sort_unique(ekeys.as<uint64_t>(), n_exclude, (uint64_t)n * (uint64_t)n, euniq.as<uint64_t>(), &me, st);
    if (rc != MDE_OK) return rc;
  }
  // draw with a small surplus, de-duplicate, drop excluded, repeat (with fresh streams) until
  // num_edges survive or the rounds are exhausted (dense exclusion sets)
  DevBuf acc;  // accumulated sorted unique accepted keys
  int64_t have = 0;
  MDE_HIP(acc.alloc((size_t)(num_edges + 1) * sizeof(uint64_t)));
  for (int round = 0; round < 8 && have < num_edges; ++round) {
    const int64_t need = num_edges - have;
    const double keep_frac = 1.0 - ((double)n_exclude + (double)have) / all_edges;
    int64_t draws = (int64_t)((double)need / (keep_frac > 0.05 ? keep_frac : 0.05) * 1.02) + 1024;
    if (draws > ((int64_t)1 << 30)) draws = (int64_t)1 << 30;
    DevBuf keys, uniq, flags, kept, num;
    MDE_HIP(keys.alloc((size_t)(draws + have) * sizeof(uint64_t)));
    MDE_HIP(uniq.alloc((size_t)(draws + have) * sizeof(uint64_t)));
    hipLaunchKernelGGL(k_sample_keys, dim3(mde_grid(draws, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, n,
                       draws, seed + 0x632BE59BD9B4E019ull * (uint64_t)(round + 1), keys.as<uint64_t>());
    MDE_LAUNCH_CHECK();
    int64_t mu = 0;
    rc = sort_unique(keys.as<uint64_t>(), draws, (uint64_t)n * (uint64_t)n, uniq.as<uint64_t>(), &mu, st);
    if (rc != MDE_OK) return rc;
    // drop excluded keys and keys accepted in earlier rounds
    MDE_HIP(flags.alloc(mu));
    MDE_HIP(kept.alloc((size_t)(mu + have) * sizeof(uint64_t)));
    MDE_HIP(num.alloc(sizeof(int64_t)));
    hipLaunchKernelGGL(k_not_excluded, dim3(mde_grid(mu, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, mu,
                       uniq.as<uint64_t>(), me, euniq.as<uint64_t>(), flags.as<uint8_t>());
    MDE_LAUNCH_CHECK();
    size_t tb = 0;
    MDE_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb, uniq.as<uint64_t>(), flags.as<uint8_t>(),
                                          kept.as<uint64_t>(), num.as<int64_t>(), (int)mu, st));
    DevBuf tmp;
    MDE_HIP(tmp.alloc(tb));
    MDE_HIP(hipcub::DeviceSelect::Flagged(tmp.p, tb, uniq.as<uint64_t>(), flags.as<uint8_t>(),
                                          kept.as<uint64_t>(), num.as<int64_t>(), (int)mu, st));
    int64_t mk = 0;
    MDE_HIP(hipMemcpyAsync(&mk, num.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    MDE_HIP(hipStreamSynchronize(st));
    // merge with the earlier rounds: concatenate + sort-unique
    if (have > 0)
      MDE_HIP(hipMemcpyAsync(kept.as<uint64_t>() + mk, acc.p, have * sizeof(uint64_t),
                             hipMemcpyDeviceToDevice, st));
    int64_t total = 0;
    rc = sort_unique(kept.as<uint64_t>(), mk + have, (uint64_t)n * (uint64_t)n, uniq.as<uint64_t>(), &total, st);
    if (rc != MDE_OK) return rc;
    // keep at most num_edges: a uniformly random subset would need a shuffle; a prefix of the
    // sorted keys would bias towards small i.  Surplus keys are thinned by stride instead.
    if (total > num_edges) {
      // total is only slightly above num_edges: drop every ceil(total/(total-num_edges))-th key
      // by recompacting with a flag kernel (host computes the stride)
      DevBuf f2, k2, n2;
      MDE_HIP(f2.alloc(total));
      MDE_HIP(k2.alloc((size_t)total * sizeof(uint64_t)));
      MDE_HIP(n2.alloc(sizeof(int64_t)));
      const int64_t drop = total - num_edges;
      // flag = 0 for the indices floor(j * total / drop), j = 0..drop-1
      std::vector<uint8_t> hf((size_t)total, 1);
      for (int64_t j = 0; j < drop; ++j) hf[(size_t)(((double)j + 0.5) * (double)total / (double)drop)] = 0;
      MDE_HIP(hipMemcpyAsync(f2.p, hf.data(), (size_t)total, hipMemcpyHostToDevice, st));
      MDE_HIP(hipStreamSynchronize(st));
      size_t tb3 = 0;
      MDE_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb3, uniq.as<uint64_t>(), f2.as<uint8_t>(), k2.as<uint64_t>(),
                                            n2.as<int64_t>(), (int)total, st));
      DevBuf t3;
      MDE_HIP(t3.alloc(tb3));
      MDE_HIP(hipcub::DeviceSelect::Flagged(t3.p, tb3, uniq.as<uint64_t>(), f2.as<uint8_t>(), k2.as<uint64_t>(),
                                            n2.as<int64_t>(), (int)total, st));
      MDE_HIP(hipMemcpyAsync(acc.p, k2.p, (size_t)num_edges * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
      MDE_HIP(hipStreamSynchronize(st));
      have = num_edges;
    } else {
      MDE_HIP(hipMemcpyAsync(acc.p, uniq.p, (size_t)total * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
      MDE_HIP(hipStreamSynchronize(st));
      have = total;
    }
  }
  hipLaunchKernelGGL(k_keys_to_edges, dim3(mde_grid(have, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, n, have,
                     acc.as<uint64_t>(), edges_out);
  MDE_LAUNCH_CHECK();
  MDE_HIP(hipStreamSynchronize(st));
  *count_host = have;
  return MDE_OK;
}
