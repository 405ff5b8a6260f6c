// kbench.hip -- standalone micro-benchmark of the hot kernel and of design alternatives
// (not part of the product; used on the GPU box to take design decisions with data).
//   ./kbench [n] [deg] [reps]
// Builds the BASELINE config-4-shaped graph on device (uniform random neighbours,
// out-degree `deg`), times:  plan build, the fused kernel (MDE_GROUP sweep), and three
// probes: stream-only, gather-only, one-sided + global fp32 atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#include <algorithm>

#include "../include/mde_hip.h"

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)
#define MK(x)                                                     \
  do {                                                            \
    int rc = (x);                                                 \
    if (rc != 0) {                                                \
      printf("mde error %d: %s (%s)\n", rc, mde_last_error(), #x); \
      exit(1);                                                    \
    }                                                             \
  } while (0)

__device__ inline uint64_t splitmix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ void k_gen(int64_t n, int64_t p, int deg, int64_t* edges, float* w) {
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < p;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t src = k / deg;
    const uint64_t h = splitmix(k);
    int64_t dst = (int64_t)(h % (uint64_t)(n - 1));
    dst += dst >= src;
    edges[2 * k] = src < dst ? src : dst;
    edges[2 * k + 1] = src < dst ? dst : src;
    w[k] = ((h >> 40) % 10) < 3 ? 2.0f : 1.0f;
  }
}
// KB_GRAPH=strat: every vertex has exactly one out-edge and (nearly) one in-edge per column stratum of n / deg
// vertices -- the rows' half-edges are spread EVENLY over the column sweep, so the consumer waves of the ring
// kernel advance in step whatever the row -> wave map is (the bound of what a balanced map can buy)
__global__ void k_gen_strat(int64_t n, int64_t p, int deg, int64_t* edges, float* w) {
  const int64_t m = n / deg;  // stratum width
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < p;
       k += (int64_t)gridDim.x * blockDim.x) {
    const int64_t src = k / deg;
    const int i = (int)(k % deg);
    const uint64_t hi = splitmix(1000003ull * (uint64_t)i + 17ull);
    // multiplier coprime with m (m = 20000 = 2^5 5^4 at config 4: odd, not a multiple of 5)
    uint64_t A = (hi % (uint64_t)m) | 1ull;
    while (A % 5ull == 0ull) A += 2ull;
    const uint64_t B = (hi >> 32) % (uint64_t)m;
    int64_t dst = (int64_t)i * m + (int64_t)(((uint64_t)src * A + B) % (uint64_t)m);
    if (dst == src) dst = (int64_t)i * m + (dst - (int64_t)i * m + 1) % m;
    const uint64_t h = splitmix(k);
    edges[2 * k] = src < dst ? src : dst;
    edges[2 * k + 1] = src < dst ? dst : src;
    w[k] = ((h >> 40) % 10) < 3 ? 2.0f : 1.0f;
  }
}
__global__ void k_genx(int64_t N, float* X) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t h = splitmix(i * 7919 + 13);
    // roughly N(0,1): sum of 4 uniforms, centred
    float s = 0;
    for (int q = 0; q < 4; ++q) s += ((h >> (16 * q)) & 0xffff) / 65536.0f;
    X[i] = (s - 2.0f) * 1.7320508f;
  }
}

// probe 1: stream the two half-edge arrays only
__global__ __launch_bounds__(256) void k_stream(int64_t H, const int* nbr, const float* w, float* out) {
  float acc = 0;
  for (int64_t h = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; h < H;
       h += (int64_t)gridDim.x * blockDim.x)
    acc += w[h] * (float)(nbr[h] & 1);
  if (acc == 123.456f) out[0] = acc;
}
// probe 2: stream + random gather of x_u (8 B), trivial math
__global__ __launch_bounds__(256) void k_gather(int64_t H, const int* nbr, const float* w,
                                                const float2* X, float* out) {
  float acc = 0;
  for (int64_t h = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; h < H;
       h += (int64_t)gridDim.x * blockDim.x) {
    const float2 x = X[nbr[h]];
    acc += w[h] * (x.x + x.y);
  }
  if (acc == 123.456f) out[0] = acc;
}
// probe 3: one-sided (rows only from edges where nbr > row) + global fp32 atomics on the
// destination side -- the design the plan deliberately avoids
__global__ __launch_bounds__(256) void k_atomic(int nrows, const int* rowptr, const int* nbr,
                                                const float* w, const float2* X, float* grad) {
  const int G = 16;
  const int lig = threadIdx.x & (G - 1);
  const int group = (blockIdx.x * 256 + threadIdx.x) / G;
  const int ngroups = gridDim.x * 256 / G;
  for (int r = group; r < nrows; r += ngroups) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    const float2 xv = X[r];
    float ax = 0, ay = 0;
    for (int h = beg + lig; h < end; h += G) {
      const int u = nbr[h];
      if (u > r) {
        const float2 xu = X[u];
        const float dx = xv.x - xu.x, dy = xv.y - xu.y;
        const float g = w[h] / (1.0f + dx * dx + dy * dy);
        ax += g * dx;
        ay += g * dy;
        atomicAdd(&grad[2 * u], -g * dx);
        atomicAdd(&grad[2 * u + 1], -g * dy);
      }
    }
    for (int o = G / 2; o > 0; o >>= 1) {
      ax += __shfl_xor(ax, o, 64);
      ay += __shfl_xor(ay, o, 64);
    }
    if (lig == 0) {
      atomicAdd(&grad[2 * r], ax);
      atomicAdd(&grad[2 * r + 1], ay);
    }
  }
}

template <class F>
static double time_ms(F f, int reps, hipStream_t st) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  f();
  f();
  CK(hipStreamSynchronize(st));
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(a, st));
    f();
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
  const int deg = argc > 2 ? atoi(argv[2]) : 50;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  const int64_t p = n * deg;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  int64_t* edges;
  float *w, *wh, *X, *grad, *loss;
  CK(hipMalloc(&edges, p * 16));
  CK(hipMalloc(&w, p * 4));
  CK(hipMalloc(&X, n * 2 * 4));
  CK(hipMalloc(&grad, n * 2 * 4));
  CK(hipMalloc(&loss, 64));
  if (getenv("KB_GRAPH") && !strcmp(getenv("KB_GRAPH"), "strat")) {
    printf("graph: stratified (one out-edge per column stratum of %lld vertices)\n", (long long)(n / deg));
    hipLaunchKernelGGL(k_gen_strat, dim3(2048), dim3(256), 0, st, n, p, deg, edges, w);
  } else
    hipLaunchKernelGGL(k_gen, dim3(2048), dim3(256), 0, st, n, p, deg, edges, w);
  hipLaunchKernelGGL(k_genx, dim3(2048), dim3(256), 0, st, n * 2, X);
  CK(hipStreamSynchronize(st));

  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  mde_plan* plan = nullptr;
  if (getenv("KB_TWICE")) {
    // a throw-away plan + layout first: what the first build in a process pays on top (code object load,
    // first use of fresh device memory)
    mde_plan* warm = nullptr;
    MK(mde_plan_create(n, p, edges, 0, n, st, &warm));
    CK(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    (void)mde_plan_layout(warm, 2, st);
    CK(hipStreamSynchronize(st));
    printf("first layout build in the process: %.2f ms\n",
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    mde_plan_destroy(warm);
  }
  CK(hipEventRecord(a, st));
  MK(mde_plan_create(n, p, edges, 0, n, st, &plan));
  CK(hipEventRecord(b, st));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const int64_t H = mde_plan_half_edges(plan);
  printf("n=%lld p=%lld H=%lld plan_build_ms=%.2f\n", (long long)n, (long long)p, (long long)H, ms);
  hipEvent_t a2, b2;
  CK(hipEventCreate(&a2));
  CK(hipEventCreate(&b2));
  CK(hipEventRecord(a2, st));
  const int layout = mde_plan_layout(plan, 2, st);
  CK(hipEventRecord(b2, st));
  CK(hipEventSynchronize(b2));
  CK(hipEventElapsedTime(&ms, a2, b2));
  printf("layout=%d (0 CSR, 1 LDS ring) layout_build_ms=%.2f\n", layout, ms);
  if (layout < 0) { printf("layout error %s\n", mde_last_error()); return 1; }
  const int64_t Hl = mde_plan_layout_half_edges(plan, layout);
  printf("layout entries %lld (%.1f%% padding)\n", (long long)Hl, 100.0 * (Hl - H) / (double)H);
  CK(hipMalloc(&wh, (size_t)Hl * 4));
  MK(mde_plan_expand_layout(plan, layout, w, wh, st));

  const double alg_bytes = 12.0 * p + 2.0 * n * 2 * 4;  // 8 + 4 B/edge + X read + grad write
  mde_func f = {};
  f.kind = MDE_F_LOG1P;
  f.kind_neg = MDE_F_NONE;
  f.a0 = wh;
  f.s0 = 1.5f;
  f.layout = layout;
  {
    double t = time_ms([&]() { MK(mde_average_distortion(plan, X, 2, &f, 1.0f, grad, loss, st)); }, reps, st);
    float hl;
    CK(hipMemcpy(&hl, loss, 4, hipMemcpyDeviceToHost));
    const char* ge = getenv("MDE_GROUP");
    printf("fused Log1p d=2 G=%s: %.3f ms  %.3e edges/s  alg %.2f TB/s (%.1f%% of 8 TB/s)  loss=%.6f\n",
           ge ? ge : "auto", t, p / (t * 1e-3), alg_bytes / (t * 1e-3) / 1e12,
           100.0 * alg_bytes / (t * 1e-3) / 8e12, hl);
    t = time_ms([&]() { MK(mde_average_distortion(plan, X, 2, &f, 1.0f, nullptr, loss, st)); }, reps, st);
    printf("fused forward-only: %.3f ms\n", t);
    const bool only = getenv("KB_LOG1P_ONLY") != nullptr;  // (libraries built with -DMDE_RING_MINIMAL)
    mde_func fq = f;
    fq.kind = MDE_F_QUADRATIC;
    mde_func fp = f;
    fp.kind_neg = MDE_F_LOG;
    fp.n0 = 1.0f;
    if (!only) {
      t = time_ms([&]() { MK(mde_average_distortion(plan, X, 2, &fq, 1.0f, grad, loss, st)); }, reps, st);
      printf("fused Quadratic d=2: %.3f ms\n", t);
      t = time_ms([&]() { MK(mde_average_distortion(plan, X, 2, &fp, 1.0f, grad, loss, st)); }, reps, st);
      printf("fused PushPull(Log1p,Log) d=2 (all w>0): %.3f ms\n", t);
    }
    if (layout == 1) {
      float* wcb;
      CK(hipMalloc(&wcb, (size_t)Hl * 4));
      int nv = 0;
      MK(mde_plan_expand_codebook(plan, w, wcb, &nv, st));
      printf("parameter codebook: %d distinct values\n", nv);
      if (nv > 0) {
        mde_func fc = f;
        fc.a0 = wcb;
        fc.a0_scalar = 2;
        t = time_ms([&]() { MK(mde_average_distortion(plan, X, 2, &fc, 1.0f, grad, loss, st)); }, reps, st);
        CK(hipMemcpy(&hl, loss, 4, hipMemcpyDeviceToHost));
        printf("fused Log1p d=2, codebook stream (4 B/half-edge): %.3f ms  %.3e edges/s  alg %.2f TB/s (%.1f%% of 8 TB/s)  loss=%.6f\n",
               t, p / (t * 1e-3), alg_bytes / (t * 1e-3) / 1e12, 100.0 * alg_bytes / (t * 1e-3) / 8e12, hl);
        mde_func fcp = fc;
        fcp.kind_neg = MDE_F_LOG;
        fcp.n0 = 1.0f;
        if (!only) t = time_ms([&]() { MK(mde_average_distortion(plan, X, 2, &fcp, 1.0f, grad, loss, st)); }, reps, st);
        printf("fused PushPull(Log1p,Log) d=2, codebook stream: %.3f ms\n", t);
      }
    }
  }
  if (layout == 1) {
    // correctness: the LDS-ring result against the CSR kernel on the same plan
    float *wh0, *grad0, *loss0;
    CK(hipMalloc(&wh0, (size_t)H * 4));
    CK(hipMalloc(&grad0, n * 2 * 4));
    CK(hipMalloc(&loss0, 64));
    MK(mde_plan_expand_layout(plan, 0, w, wh0, st));
    mde_func f0 = f;
    f0.a0 = wh0;
    f0.layout = 0;
    MK(mde_average_distortion(plan, X, 2, &f0, 1.0f, grad0, loss0, st));
    MK(mde_average_distortion(plan, X, 2, &f, 1.0f, grad, loss, st));
    CK(hipStreamSynchronize(st));
    std::vector<float> g1(n * 2), g0(n * 2);
    float l1, l0;
    CK(hipMemcpy(g1.data(), grad, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(g0.data(), grad0, n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&l1, loss, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&l0, loss0, 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxg = 0;
    int64_t bad = 0;
    for (int64_t i = 0; i < n * 2; ++i) {
      const double dd = fabs((double)g1[i] - (double)g0[i]);
      if (!(dd <= 1e30)) ++bad;
      maxd = std::max(maxd, dd);
      maxg = std::max(maxg, fabs((double)g0[i]));
    }
    printf("check vs CSR kernel: loss %.8f vs %.8f, max |dgrad| %.3e (max |grad| %.3e, rel %.2e), non-finite %lld  %s\n",
           l1, l0, maxd, maxg, maxd / maxg, (long long)bad,
           (maxd <= 1e-4 * maxg && fabs(l1 - l0) <= 1e-5 * fabs(l0) && bad == 0) ? "OK" : "MISMATCH");
  }
  if (!getenv("MDE_GROUP") && layout == 0) {
    const int* nbr = mde_plan_nbr(plan);
    const int* rowptr = mde_plan_rowptr(plan);
    double t = time_ms([&]() { hipLaunchKernelGGL(k_stream, dim3(4096), dim3(256), 0, st, H, nbr, wh, loss); }, reps, st);
    printf("probe stream-only (8 B/half-edge): %.3f ms  %.2f TB/s\n", t, 8.0 * H / (t * 1e-3) / 1e12);
    t = time_ms([&]() { hipLaunchKernelGGL(k_gather, dim3(4096), dim3(256), 0, st, H, nbr, wh, (const float2*)X, loss); }, reps, st);
    printf("probe stream+gather: %.3f ms  %.1f G gathers/s\n", t, H / (t * 1e-3) / 1e9);
    t = time_ms([&]() {
      CK(hipMemsetAsync(grad, 0, n * 8, st));
      hipLaunchKernelGGL(k_atomic, dim3(2048), dim3(256), 0, st, (int)n, rowptr, nbr, wh, (const float2*)X, grad);
    }, reps, st);
    printf("probe one-sided + global atomics: %.3f ms\n", t);
  }
  mde_plan_destroy(plan);
  return 0;
}
