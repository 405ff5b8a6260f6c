// mde_panel.hip -- LDS-tiled ("column panel") variant of the fused average-distortion kernel
// for small embedding dimensions (d <= 4) on graphs whose embedding table does not fit L2.
//
// Why: the CSR kernel's only random access is the gather of x_u.  Measured on MI355X
// (tools/gprobe.hip): random 8-byte gathers reach ~100 G/s from an 8 MB table (41 % L2 hit
// rate, 4.5 GB of fabric traffic per evaluation against 0.6 GB of algorithmic bytes) and only
// ~160 G/s even when the table is L2-resident -- the L2 request rate, not HBM, bounds the
// kernel at ~1.2 ms for 10^8 half-edges.  Here every random access is served by LDS instead:
//
//   * vertices are cut into row blocks (R rows: x_v and the gradient accumulators of the
//     block live in LDS for the whole kernel) and column panels (C vertices: x_u staged in
//     LDS, one panel at a time, with coalesced 16-byte loads);
//   * half-edges are grouped into tiles (row block, panel), sorted by row inside a tile, and
//     stored as one packed uint32 (row_local << 16 | col_local) + one fp32 parameter: the
//     same 8 streamed bytes per half-edge as the CSR layout;
//   * one workgroup owns a row block: it walks the panels, and for each tile streams the
//     packed half-edges, reads x_v / x_u from LDS, evaluates f and f'/d, and accumulates
//     g (x_v - x_u) into the block's LDS accumulators with ds_add_f32.  Rows are owned by
//     exactly one workgroup, so the gradient rows it writes at the end are final: no global
//     atomics, no partial-gradient pass.
//
// The summation order inside a row now depends on LDS atomic arbitration, so results are
// reproducible to fp32 rounding, not bitwise (the CSR kernel stays bitwise reproducible and is
// used for everything this layout does not cover).
#include <hipcub/hipcub.hpp>

#include "mde_common.h"
#include "mde_functions.h"
#include "mde_plan.h"
#define COMMA ,

#define MDE_LDS_BYTES 163840
#define MDE_PANEL_RESERVE 2048  // reduction scratch + slack

// ---------------------------------------------------------------- layout construction
__global__ __launch_bounds__(MDE_BLOCK) void k_panel_keys(int nrows, const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ nbr, int P_R,
                                                          int P_C, int NP, int KR_shift,
                                                          uint32_t* __restrict__ keys,
                                                          uint32_t* __restrict__ vals) {
  constexpr int G = 16;
  const int lig = threadIdx.x & (G - 1);
  const int group = (blockIdx.x * MDE_BLOCK + threadIdx.x) / G;
  const int ngroups = (gridDim.x * MDE_BLOCK) / G;
  for (int r = group; r < nrows; r += ngroups) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    const uint32_t rb = (uint32_t)(r / P_R), rl = (uint32_t)(r % P_R);
    for (int q = beg + lig; q < end; q += G) {
      const uint32_t cp = (uint32_t)(nbr[q] / P_C);
      keys[q] = ((rb * (uint32_t)NP + cp) << KR_shift) | rl;
      vals[q] = (uint32_t)q;
    }
  }
}

__global__ __launch_bounds__(MDE_BLOCK) void k_panel_fill(int64_t H, const uint32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ vals,
                                                          const int32_t* __restrict__ nbr,
                                                          const int32_t* __restrict__ eid, int P_C,
                                                          int NP, int KR_shift,
                                                          uint32_t* __restrict__ packed,
                                                          int32_t* __restrict__ peid) {
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < H;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    const uint32_t key = keys[i], q = vals[i];
    const uint32_t rl = key & ((1u << KR_shift) - 1u);
    const uint32_t cp = (key >> KR_shift) % (uint32_t)NP;
    const uint32_t cl = (uint32_t)nbr[q] - cp * (uint32_t)P_C;
    packed[i] = (rl << 16) | cl;
    peid[i] = eid[q];
  }
}

// tile_ptr[t] = first sorted position whose tile id (key >> shift) >= t, t = 0..ntiles
__global__ __launch_bounds__(MDE_BLOCK) void k_tile_ptr(int64_t H, uint32_t ntiles, int shift,
                                                        const uint32_t* __restrict__ keys,
                                                        int32_t* __restrict__ tile_ptr) {
  for (int64_t h = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; h <= H;
       h += (int64_t)gridDim.x * MDE_BLOCK) {
    const int64_t prev = (h == 0) ? -1 : (int64_t)(keys[h - 1] >> shift);
    int64_t cur = (h == H) ? (int64_t)ntiles : (int64_t)(keys[h] >> shift);
    if (cur > (int64_t)ntiles) cur = ntiles;
    for (int64_t t = prev + 1; t <= cur; ++t) tile_ptr[t] = (int32_t)h;
  }
}

// sub_ptr[t * NW + w] = first sorted position with key >= (t << shift | w * RW): the rows of a
// tile are split into NW contiguous ranges, one per wave of the workgroup, so that every
// accumulator row has exactly one writer (no LDS atomics, fixed summation order).
__global__ __launch_bounds__(MDE_BLOCK) void k_sub_ptr(int64_t H, uint32_t ntiles, int NW, int RW,
                                                       int shift, const uint32_t* __restrict__ keys,
                                                       int32_t* __restrict__ sub_ptr) {
  const int64_t total = (int64_t)ntiles * NW;
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i <= total;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    if (i == total) {
      sub_ptr[i] = (int32_t)H;
      continue;
    }
    const uint32_t t = (uint32_t)(i / NW), w = (uint32_t)(i % NW);
    const uint64_t target = ((uint64_t)t << shift) | (uint64_t)(w * (uint32_t)RW);
    int64_t lo = 0, hi = H;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if ((uint64_t)keys[mid] >= target)
        hi = mid;
      else
        lo = mid + 1;
    }
    sub_ptr[i] = (int32_t)lo;
  }
}

// Within a (tile, wave) sub-range of m row-sorted half-edges processed in K = ceil(m / 64) wave
// iterations, store element s at iteration s % K, lane s / K: two half-edges of the same row
// then share an iteration only when the row has more than K entries in the tile, so the common
// case needs no run folding at all.  Iteration k owns positions [off(k), off(k) + cnt(k)),
// cnt(k) = m / K + (k < m % K), off(k) = k * (m / K) + min(k, m % K).
__global__ __launch_bounds__(MDE_BLOCK) void k_interleave(int64_t nsub, const int32_t* __restrict__ sub_ptr,
                                                          const uint32_t* __restrict__ keys_in,
                                                          const uint32_t* __restrict__ vals_in,
                                                          uint32_t* __restrict__ keys_out,
                                                          uint32_t* __restrict__ vals_out,
                                                          int32_t* __restrict__ sub_qr) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * MDE_BLOCK) >> 6;
  for (int64_t i = w0; i < nsub; i += nw) {
    const int beg = sub_ptr[i], m = sub_ptr[i + 1] - beg;
    if (m <= 0) {
      if (lane == 0) sub_qr[i] = 0;
      continue;
    }
    const int K = (m + 63) >> 6, q = m / K, rem = m % K;
    if (lane == 0) sub_qr[i] = (rem << 8) | q;  // the kernel never divides
    for (int sidx = lane; sidx < m; sidx += 64) {
      const int k = sidx % K, l = sidx / K;
      const int pos = beg + k * q + (k < rem ? k : rem) + l;
      keys_out[pos] = keys_in[beg + sidx];
      vals_out[pos] = vals_in[beg + sidx];
    }
  }
}

static int g_panel_mode = -2;  // MDE_PANEL env: -1 auto, 0 never, 1 whenever the layout is feasible
static int panel_mode() {
  if (g_panel_mode == -2) {
    const char* e = getenv("MDE_PANEL");
    g_panel_mode = e ? atoi(e) : -1;
  }
  return g_panel_mode;
}

static int bits_for_u64(uint64_t maxval) {
  int b = 1;
  while (b < 64 && (maxval >> b)) ++b;
  return b;
}

// Decide tile sizes for dimension d; returns false when the layout is not worthwhile.
static bool choose_sizes(const mde_plan* plan, int d, int* P_R, int* P_C) {
  const int64_t nloc = plan->row_hi - plan->row_lo;
  if (d < 1 || d > 4 || nloc <= 0 || plan->H <= 0) return false;
  const int mode = panel_mode();
  if (mode == 0) return false;
  // rows: about one block per CU (256), multiple of 64, LDS: 2*d*4 bytes per row, at most
  // ~40 % of the LDS so the panel keeps the rest
  // (a rank that owns only n/N rows keeps the same block height as a full plan -- tiles must
  // not get thinner -- and fills the CUs with Q column groups per row block instead)
  int64_t pr = (plan->n + 255) / 256;
  if (pr > nloc) pr = nloc;
  pr = ((pr + 63) / 64) * 64;
  // (rows are padded by one slot per 32 against bank conflicts: 33/32 of the space)
  const int64_t pr_max = ((int64_t)(0.4 * MDE_LDS_BYTES) / (8 * d)) / 64 * 64;
  if (pr > pr_max) pr = pr_max;
  if (pr < 64) pr = 64;
  int64_t pc = (MDE_LDS_BYTES - MDE_PANEL_RESERVE - (pr + pr / 32) * 8 * d) / (4 * d);
  if (pc > 65535) pc = 65535;
  if (pc * d > 6 * 1024 * 4) pc = (6 * 1024 * 4) / d;  // MDE_PANEL_STG float4 per thread
  pc = (pc / 256) * 256;
  if (pc < 1024) return false;
  if (pc > plan->n) pc = ((plan->n + 255) / 256) * 256;
  const int64_t nrb = (nloc + pr - 1) / pr, np = (plan->n + pc - 1) / pc;
  if (nrb * 16 > MDE_MAX_PARTIALS) return false;
  int kr_shift = bits_for_u64((uint64_t)pr - 1);
  if (bits_for_u64((uint64_t)(nrb * np)) + kr_shift > 32) return false;
  if (mode != 1) {
    // auto: only when the table overflows L2 and tiles are big enough to amortise the staging
    if ((int64_t)plan->n * d * 4 < (6 << 20)) return false;
    if ((double)plan->H / (double)(nrb * np) < 1536.0) return false;
  }
  *P_R = (int)pr;
  *P_C = (int)pc;
  return true;
}

static int build_panels(mde_plan* plan, int d, hipStream_t st) {
  int P_R = 0, P_C = 0;
  if (!choose_sizes(plan, d, &P_R, &P_C)) return 0;
  mde_panel_layout& L = plan->panel;
  const int64_t nloc = plan->row_hi - plan->row_lo;
  const int64_t H = plan->H;
  const int NRB = (int)((nloc + P_R - 1) / P_R), NP = (int)((plan->n + P_C - 1) / P_C);
  const int KR_shift = bits_for_u64((uint64_t)P_R - 1);
  const uint32_t ntiles = (uint32_t)NRB * (uint32_t)NP;
  uint32_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr;
  void* tmp = nullptr;
  uint32_t* packed = nullptr;
  int32_t *peid = nullptr, *tile_ptr = nullptr, *sub_ptr = nullptr, *sub_qr = nullptr;
  hipError_t e = hipSuccess;
  auto fail = [&](hipError_t err, const char* what) {
    if (keys) (void)hipFree(keys);
    if (vals) (void)hipFree(vals);
    if (keys2) (void)hipFree(keys2);
    if (vals2) (void)hipFree(vals2);
    if (tmp) (void)hipFree(tmp);
    if (packed) (void)hipFree(packed);
    if (peid) (void)hipFree(peid);
    if (tile_ptr) (void)hipFree(tile_ptr);
    if (sub_ptr) (void)hipFree(sub_ptr);
    if (sub_qr) (void)hipFree(sub_qr);
    return mde_hip_fail(err, what, __FILE__, __LINE__);
  };
  const size_t hb = (size_t)H * sizeof(uint32_t);
#define PB(call)                                   \
  do {                                             \
    e = (call);                                    \
    if (e != hipSuccess) return fail(e, #call);    \
  } while (0)
  PB(hipMalloc(&keys, hb));
  PB(hipMalloc(&vals, hb));
  PB(hipMalloc(&keys2, hb));
  PB(hipMalloc(&vals2, hb));
  PB(hipMalloc(&packed, hb));
  PB(hipMalloc(&peid, hb));
  PB(hipMalloc(&tile_ptr, ((size_t)ntiles + 1) * sizeof(int32_t)));
  PB(hipMalloc(&sub_ptr, ((size_t)ntiles * MDE_PANEL_WAVES + 1) * sizeof(int32_t)));
  PB(hipMalloc(&sub_qr, ((size_t)ntiles * MDE_PANEL_WAVES + 1) * sizeof(int32_t)));
  hipLaunchKernelGGL(k_panel_keys, dim3(mde_grid(nloc * 16, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st,
                     (int)nloc, plan->rowptr, plan->nbr, P_R, P_C, NP, KR_shift, keys, vals);
  PB(hipGetLastError());
  size_t tmp_bytes = 0;
  const int end_bit = bits_for_u64((uint64_t)ntiles) + KR_shift;
  PB(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (int)H, 0,
                                        end_bit > 32 ? 32 : end_bit, st));
  PB(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  PB(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (int)H, 0,
                                        end_bit > 32 ? 32 : end_bit, st));
  hipLaunchKernelGGL(k_tile_ptr, dim3(mde_grid(H + 1, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, H,
                     ntiles, KR_shift, keys2, tile_ptr);
  PB(hipGetLastError());
  const int64_t nsub = (int64_t)ntiles * MDE_PANEL_WAVES;
  hipLaunchKernelGGL(k_sub_ptr, dim3(mde_grid(nsub + 1, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, H,
                     ntiles, MDE_PANEL_WAVES, P_R / MDE_PANEL_WAVES, KR_shift, keys2, sub_ptr);
  PB(hipGetLastError());
  hipLaunchKernelGGL(k_interleave, dim3(mde_grid(nsub * 64, MDE_BLOCK, 8192)), dim3(MDE_BLOCK), 0, st,
                     nsub, sub_ptr, keys2, vals2, keys, vals, sub_qr);
  PB(hipGetLastError());
  hipLaunchKernelGGL(k_panel_fill, dim3(mde_grid(H, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, H, keys,
                     vals, plan->nbr, plan->eid, P_C, NP, KR_shift, packed, peid);
  PB(hipGetLastError());
  PB(hipStreamSynchronize(st));
#undef PB
  (void)hipFree(keys);
  (void)hipFree(vals);
  (void)hipFree(keys2);
  (void)hipFree(vals2);
  (void)hipFree(tmp);
  if (L.packed) (void)hipFree(L.packed);
  if (L.eid) (void)hipFree(L.eid);
  if (L.tile_ptr) (void)hipFree(L.tile_ptr);
  if (L.sub_ptr) (void)hipFree(L.sub_ptr);
  if (L.sub_qr) (void)hipFree(L.sub_qr);
  L.sub_ptr = sub_ptr;
  L.sub_qr = sub_qr;
  if (L.partial) (void)hipFree(L.partial);
  L.partial = nullptr;
  // column groups: enough workgroups to fill 256 CUs, at least 4 panels per group
  int Q = 1;
  if (NRB <= 128) {
    Q = 256 / NRB;  // one resident round of workgroups: NRB * Q <= 256 CUs
    if (Q > NP / 4) Q = NP / 4;
    if (Q > 16) Q = 16;
    if (Q < 1) Q = 1;
  }
  L.col_groups = Q;
  if (Q > 1) {
    hipError_t pe = hipMalloc(&L.partial, sizeof(float) * (size_t)Q * (size_t)nloc * (size_t)d);
    if (pe != hipSuccess) return mde_hip_fail(pe, "hipMalloc(panel partials)", __FILE__, __LINE__);
  }
  L.rows_per_wave = P_R / MDE_PANEL_WAVES;
  L.d = d;
  L.rows_per_block = P_R;
  L.cols_per_panel = P_C;
  L.n_row_blocks = NRB;
  L.n_panels = NP;
  L.H = H;
  L.packed = packed;
  L.eid = peid;
  L.tile_ptr = tile_ptr;
  return 1;
}

// layout the fused kernel will use for dimension d: 0 = CSR, 1 = column panels (built on first
// request).  Negative: error.
extern "C" int mde_plan_layout(mde_plan* plan, int32_t d, void* stream) {
  if (!plan || d <= 0) return MDE_E_INVALID;
  if (plan->panel.packed && plan->panel.d == d) return 1;
  int P_R, P_C;
  if (!choose_sizes(plan, d, &P_R, &P_C)) return 0;
  return build_panels(plan, d, mde_stream(stream));
}

__global__ __launch_bounds__(MDE_BLOCK) void k_expand_panel(int64_t H, const int32_t* __restrict__ eid,
                                                            const float* __restrict__ in,
                                                            float* __restrict__ out) {
  for (int64_t q = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; q < H;
       q += (int64_t)gridDim.x * MDE_BLOCK)
    out[q] = in[eid[q]];
}

extern "C" int mde_plan_expand_layout(const mde_plan* plan, int32_t layout, const float* in_edge,
                                      float* out_half, void* stream) {
  if (!plan || !in_edge || !out_half) return MDE_E_INVALID;
  if (layout == 0) return mde_plan_expand(plan, in_edge, out_half, stream);
  if (layout != 1 || !plan->panel.eid) {
    mde_set_error("mde_plan_expand_layout: the panel layout has not been built");
    return MDE_E_INVALID;
  }
  if (plan->H == 0) return MDE_OK;
  hipLaunchKernelGGL(k_expand_panel, dim3(mde_grid(plan->H, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0,
                     mde_stream(stream), plan->H, plan->panel.eid, in_edge, out_half);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// ---------------------------------------------------------------- the kernel
// 1024 threads = 16 waves per workgroup; wave w owns rows [w * RW, (w+1) * RW) of the block:
// only it reads-modifies-writes their LDS accumulators, so no atomics are needed and the
// order of additions is fixed by the (sorted) tile order.  Inside one wave iteration lanes
// that hit the same row are adjacent (tiles are sorted by row): a segmented Hillis-Steele
// scan over the lanes (shuffles, early exit when no run is longer than the current stride)
// folds each run into its last lane, which performs the single read-add-write of that row.
//
// Software pipeline across tiles: while tile k is processed out of LDS, the x_u panel of tile
// k+1 (STG float4 per thread) and the wave's packed half-edges + parameters of tile k+1
// (MAXI registers each) are already in flight from L2 / HBM; they are committed to LDS /
// consumed after the two barriers that separate the tiles.
#define MDE_PANEL_BS (64 * MDE_PANEL_WAVES)
#define MDE_PANEL_STG (6144 / MDE_PANEL_BS)  // float4 staging registers per thread (panel <= 6144 float4)
#define MDE_PANEL_MAXI (128 / MDE_PANEL_WAVES)  // prefetched wave-iterations per tile

template <int D, class Fn, bool HAS_GRAD>
__global__ __launch_bounds__(MDE_PANEL_BS) void k_fused_panel(
    int nloc, int row_lo, int n, int P_R, int P_C, int NP, int Q, const int32_t* __restrict__ tile_ptr,
    const int32_t* __restrict__ sub_ptr, const int32_t* __restrict__ sub_qr,
    const uint32_t* __restrict__ packed, const float* __restrict__ a0, const float* __restrict__ a1,
    int a0_scalar, int a1_scalar, const float* __restrict__ X, float* __restrict__ grad,
    float* __restrict__ partial, double* __restrict__ loss_partials, Fn fn, float inv_p,
    float grad_scale) {
  constexpr int BS = MDE_PANEL_BS, NW = MDE_PANEL_WAVES, STG = MDE_PANEL_STG, MAXI = MDE_PANEL_MAXI;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // row arrays are padded by one slot every 32 rows: interleaved tiles make the lanes of an
  // iteration touch rows at a near-constant stride, which would otherwise pile onto few banks
  const int PRP = P_R + (P_R >> 5);
  float* XR = lds;                 // [PRP * D]  x_v of the block's rows
  float* GR = XR + PRP * D;        // [PRP * D]  gradient accumulators
  float* XC = GR + PRP * D;        // [P_C * D]  x_u of the current panel
  double* red = reinterpret_cast<double*>(XC + (size_t)P_C * D);  // [NW]
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned ulane = (unsigned)lane;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keep it scalar
  // block b = rb * Q + qg: row block rb, column group qg walks panels [cp_lo, cp_hi)
  const int rb = blockIdx.x / Q, qg = blockIdx.x % Q;
  const int cp_lo = (int)(((int64_t)NP * qg) / Q), cp_hi = (int)(((int64_t)NP * (qg + 1)) / Q);
  const int r0 = rb * P_R;
  const int nr = min(P_R, nloc - r0);
  const float a0s = a0_scalar ? a0[0] : 1.0f;
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;
  const bool a1_arr = a1 && !a1_scalar;
  const float* Xrow = X + (size_t)(row_lo + r0) * D;
  for (int i = tid; i < PRP * D; i += BS) {
    XR[i] = 0.0f;
    GR[i] = 0.0f;
  }
  __syncthreads();
  for (int r = tid; r < nr; r += BS) {
    const int slot = (r + (r >> 5)) * D;
#pragma unroll
    for (int c = 0; c < D; ++c) XR[slot + c] = Xrow[r * D + c];
  }
  float loss = 0.0f;
  const int32_t* tp = tile_ptr + (size_t)rb * NP;
  const int32_t* sp_base = sub_ptr + (size_t)rb * NP * NW + wave;
  const int32_t* qr_base = sub_qr + (size_t)rb * NP * NW + wave;

  // one half-edge per lane, split in two stages so that two iterations can be in flight per wave
  // (their LDS reads / transcendental chains are independent; only the accumulator updates are
  // ordered): compute() evaluates, commit() folds runs of equal rows and updates the LDS row.
  struct Item {
    int rslot, key;
    float v[D];
    bool active;
  };
  auto compute = [&](bool active, uint32_t pk, float p0, float p1, Item& it) {
    const int rl = (int)(pk >> 16), cl = (int)(pk & 0xffffu);
    it.rslot = (rl + (rl >> 5)) * D;
    float diff[D], ss = 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      diff[c] = XR[it.rslot + c] - XC[cl * D + c];
      ss = fmaf(diff[c], diff[c], ss);
    }
    float f, gd;
    fn.eval(ss, p0, p1, f, gd);
    const float g = active ? mde_fix_g(gd * inv_p) : 0.0f;
    loss += active ? f : 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c) it.v[c] = g * diff[c];
    // Lanes of one iteration hold ascending rows (row-sorted tile, interleaved storage), so
    // equal rows are adjacent lanes.  Inactive lanes carry unique negative keys.
    it.key = active ? rl : (-2 - lane);
    it.active = active;
  };
  auto commit = [&](Item& it) {
    if (!HAS_GRAD) return;
    const int key = it.key;
    bool tail = it.active;
    const int kprev = __builtin_amdgcn_update_dpp(-1, key, 0x138, 0xf, 0xf, false);  // wave_shr:1
    if (__any(kprev == key)) {
      // rare: a row has more entries in this tile than the wave has iterations.  Round r adds
      // the ORIGINAL contribution of lane i-r when it has the same row (keys / values shifted
      // one lane per round with DPP wave_shr); the last lane of each run writes.
      int kc = key;
      float sv[D];
#pragma unroll
      for (int c = 0; c < D; ++c) sv[c] = it.v[c];
#pragma nounroll
      for (int r = 1; r < 64; ++r) {
        kc = __builtin_amdgcn_update_dpp(-1, kc, 0x138, 0xf, 0xf, false);
        const bool m = (kc == key);
        if (!__any(m)) break;
#pragma unroll
        for (int c = 0; c < D; ++c) {
          sv[c] = __int_as_float(
              __builtin_amdgcn_update_dpp(0, __float_as_int(sv[c]), 0x138, 0xf, 0xf, false));
          it.v[c] += m ? sv[c] : 0.0f;
        }
      }
      const int knext = __builtin_amdgcn_update_dpp(-1, key, 0x130, 0xf, 0xf, false);  // wave_shl:1
      tail = it.active && (knext != key);
    }
    if (tail) {
#pragma unroll
      for (int c = 0; c < D; ++c) GR[it.rslot + c] += it.v[c];
    }
  };

  float4 stg[STG];
  uint32_t pkn[MAXI];
  float wn[MAXI];
  int nbeg = 0, nm = 0, nq = 0, nrem = 0;  // next tile's sub-range: start, size, m / K, m % K
  // issue every global load of tile `cp` (panel -> staging registers, stream -> pkn / wn)
  auto prefetch = [&](int cp) {
    const int c0 = cp * P_C;
    const int nc = min(P_C, n - c0);
    const int t4 = (nc * D) >> 2;  // c0 * D * 4 bytes is 16-byte aligned (P_C multiple of 256)
    const float4* s4 = reinterpret_cast<const float4*>(X + (size_t)c0 * D);
#pragma unroll
    for (int k = 0; k < STG; ++k) {
      const int i = tid + k * BS;
      stg[k] = (i < t4) ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int32_t* sp = sp_base + (size_t)cp * NW;
    nbeg = __builtin_amdgcn_readfirstlane(sp[0]);
    nm = __builtin_amdgcn_readfirstlane(sp[1]) - nbeg;
    const int qr = __builtin_amdgcn_readfirstlane(qr_base[(size_t)cp * NW]);
    nq = qr & 255;
    nrem = qr >> 8;
    // interleaved order (k_interleave): iteration k holds cnt(k) = q + (k < rem) entries
    const int K = (nm + 63) >> 6;
    const uint32_t* pb = packed + nbeg;  // wave-uniform bases: scalar address arithmetic
    const float* ab = a0 + nbeg;
    int off = 0;
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
      const int cnt = (k < K) ? nq + (k < nrem ? 1 : 0) : 0;
      const bool ok = lane < cnt;
      const uint32_t* pbo = pb + off;  // uniform: scalar base, the vector offset is just the lane
      const float* abo = ab + off;
      pkn[k] = ok ? pbo[ulane] : 0u;
      wn[k] = (ok && !a0_scalar) ? abo[ulane] : a0s;
      off += cnt;
    }
  };
  auto next_nonempty = [&](int cp) {
    while (cp < cp_hi && tp[cp] == tp[cp + 1]) ++cp;
    return cp;
  };

  int cp = next_nonempty(cp_lo);
  if (cp < cp_hi) prefetch(cp);
  while (cp < cp_hi) {
    const int c0 = cp * P_C;
    const int nc = min(P_C, n - c0);
    __syncthreads();  // everyone is done with the previous panel (and XR/GR are initialised)
    {
      float4* d4 = reinterpret_cast<float4*>(XC);
      const int t4 = (nc * D) >> 2;
#pragma unroll
      for (int k = 0; k < STG; ++k) {
        const int i = tid + k * BS;
        if (i < t4) d4[i] = stg[k];
      }
      const float* src = X + (size_t)c0 * D;
      for (int i = (t4 << 2) + tid; i < nc * D; i += BS) XC[i] = src[i];  // < 4 tail floats
    }
    // this tile's stream registers
    uint32_t pkc[MAXI];
    float wc[MAXI];
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
      pkc[k] = pkn[k];
      wc[k] = wn[k];
    }
    const int cbeg = nbeg, cm = nm, cq = nq, crem = nrem;
    const int cK = (cm + 63) >> 6;
    __syncthreads();
    const int cpn = next_nonempty(cp + 1);
    if (cpn < cp_hi) prefetch(cpn);  // in flight while this tile is processed
    int off = 0;
#pragma unroll
    for (int k = 0; k < MAXI; k += 2) {
      if (k >= cK) break;
      const int cnt0 = cq + (k < crem ? 1 : 0);
      const int cnt1 = (k + 1 < cK) ? cq + (k + 1 < crem ? 1 : 0) : 0;
      const bool act0 = lane < cnt0, act1 = lane < cnt1;
      const float p10 = a1_arr ? (act0 ? a1[cbeg + off + lane] : 1.0f) : a1s;
      const float p11 = a1_arr ? (act1 ? a1[cbeg + off + cnt0 + lane] : 1.0f) : a1s;
      Item ia, ib;
      compute(act0, pkc[k], wc[k], p10, ia);
      compute(act1, pkc[k + 1], wc[k + 1], p11, ib);
      commit(ia);
      commit(ib);
      off += cnt0 + cnt1;
    }
    for (int k = MAXI; k < cK; ++k) {  // oversized tiles (skewed degrees): not prefetched
      const int cnt = cq + (k < crem ? 1 : 0);
      const int h = cbeg + off + lane;
      const bool active = lane < cnt;
      const uint32_t pk = active ? packed[h] : 0u;
      const float p0 = (active && !a0_scalar) ? a0[h] : a0s;
      const float p1 = a1_arr ? (active ? a1[h] : 1.0f) : a1s;
      Item it;
      compute(active, pk, p0, p1, it);
      commit(it);
      off += cnt;
    }
    cp = cpn;
  }
  __syncthreads();
  if (HAS_GRAD) {
    // Q == 1: the rows are final.  Q > 1: unscaled per-group partials, summed by k_panel_combine
    float* grow = (Q == 1) ? grad + (size_t)(row_lo + r0) * D
                           : partial + ((size_t)qg * nloc + r0) * D;
    const float sc = (Q == 1) ? grad_scale : 1.0f;
    for (int r = tid; r < nr; r += BS) {
      const int slot = (r + (r >> 5)) * D;
#pragma unroll
      for (int c = 0; c < D; ++c) grow[r * D + c] = GR[slot + c] * sc;
    }
  }
  // block-wide loss partial
  double v = mde_wave_sum((double)loss);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < NW; ++i) s += red[i];
    loss_partials[blockIdx.x] = s;
  }
}

// grad[row] = scale * sum_q partial[q][row]  (fixed order)
__global__ __launch_bounds__(MDE_BLOCK) void k_panel_combine(int64_t nlocD, int Q, const float* __restrict__ partial,
                                                             float scale, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < nlocD;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    float s = 0.0f;
    for (int q = 0; q < Q; ++q) s += partial[(size_t)q * nlocD + i];
    out[i] = s * scale;
  }
}

struct PanelArgs {
  mde_plan* plan;
  const float* X;
  int d;
  const float *a0, *a1;
  int a0_scalar, a1_scalar;
  float* grad;
  float inv_p, grad_scale;
  hipStream_t st;
};

template <int D, class Fn>
static int launch_panel(const PanelArgs& A, const Fn& fn, int* nblocks) {
  const mde_panel_layout& L = A.plan->panel;
  const size_t prp = (size_t)L.rows_per_block + (size_t)(L.rows_per_block >> 5);
  const size_t lds = (prp * 2 * D + (size_t)L.cols_per_panel * D) * sizeof(float) +
                     MDE_PANEL_WAVES * sizeof(double) + 64;
  static bool attr_set = false;
  auto kern = A.grad ? k_fused_panel<D, Fn, true> : k_fused_panel<D, Fn, false>;
  static bool attr_set_fwd = false;
  bool& attr_done = A.grad ? attr_set : attr_set_fwd;
  if (!attr_done) {
    MDE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, MDE_LDS_BYTES));
    attr_done = true;
  }
  const int Q = L.col_groups;
  *nblocks = L.n_row_blocks * Q;
  hipLaunchKernelGGL(kern, dim3(L.n_row_blocks * Q), dim3(MDE_PANEL_BS), lds, A.st,
                     (int)(A.plan->row_hi - A.plan->row_lo), (int)A.plan->row_lo, (int)A.plan->n,
                     L.rows_per_block, L.cols_per_panel, L.n_panels, Q, L.tile_ptr, L.sub_ptr, L.sub_qr, L.packed,
                     A.a0, A.a1, A.a0_scalar, A.a1_scalar, A.X, A.grad, L.partial, A.plan->partials, fn,
                     A.inv_p, A.grad_scale);
  MDE_LAUNCH_CHECK();
  if (Q > 1 && A.grad) {
    const int64_t nlocD = (A.plan->row_hi - A.plan->row_lo) * (int64_t)D;
    hipLaunchKernelGGL(k_panel_combine, dim3(mde_grid(nlocD, MDE_BLOCK, 2048)), dim3(MDE_BLOCK), 0, A.st,
                       nlocD, Q, L.partial, A.grad_scale, A.grad + (size_t)A.plan->row_lo * D);
    MDE_LAUNCH_CHECK();
  }
  return MDE_OK;
}

static MdeFuncArgs panel_func_args(const mde_func* f) {
  MdeFuncArgs a;
  a.kind = f->kind;
  a.kind_neg = f->kind_neg;
  a.S = {f->s0, f->s1, f->s2};
  a.N = {f->n0, f->n1, f->n2};
  return a;
}

// Called by mde_average_distortion.  Returns 1 when the panel kernel was launched (nblocks =
// number of loss partials written), 0 when the caller should use the CSR kernel, < 0 on error.
int mde_panel_try(mde_plan* plan, const float* X, int d, const mde_func* f, float grad_scale,
                  float* grad, float inv_p, hipStream_t st, int* nblocks) {
  if (!plan->panel.packed || plan->panel.d != d) return 0;
  PanelArgs A{plan, X, d, f->a0, f->a1, f->a0_scalar, f->a1_scalar, grad, inv_p, grad_scale, st};
  const MdeFuncArgs a = panel_func_args(f);
  const int ea = mde_exp_class(f->s0), en = mde_exp_class(f->n0);
  int rc = MDE_OK;
#define PANEL(FN)                                               \
  do {                                                          \
    FN fn{a};                                                   \
    if (d == 2)                                                 \
      rc = launch_panel<2, FN>(A, fn, nblocks);                 \
    else if (d == 3)                                            \
      rc = launch_panel<3, FN>(A, fn, nblocks);                 \
    else if (d == 1)                                            \
      rc = launch_panel<1, FN>(A, fn, nblocks);                 \
    else                                                        \
      rc = launch_panel<4, FN>(A, fn, nblocks);                 \
    return rc == MDE_OK ? 1 : rc;                               \
  } while (0)
  if (d == 2 || d == 3) {
    if (f->kind_neg == MDE_F_NONE) {
      if (f->kind == MDE_F_LOG1P && ea == 2) PANEL(FnSingle<MDE_F_LOG1P COMMA 2>);
      if (f->kind == MDE_F_QUADRATIC) PANEL(FnSingle<MDE_F_QUADRATIC COMMA 0>);
    } else if (f->kind == MDE_F_LOG1P && ea == 2) {
      if (f->kind_neg == MDE_F_LOG && en == 1)
        PANEL(FnPushPull<MDE_F_LOG1P COMMA 2 COMMA MDE_F_LOG COMMA 1>);
      if (f->kind_neg == MDE_F_LOGRATIO && en == 3)
        PANEL(FnPushPull<MDE_F_LOG1P COMMA 2 COMMA MDE_F_LOGRATIO COMMA 3>);
    }
  }
  PANEL(FnRuntime);
#undef PANEL
}
