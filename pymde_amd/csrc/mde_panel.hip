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
//   * half-edges are grouped into tiles (row block, panel), sorted by row inside a tile and
//     split into 16 per-wave row sub-ranges; every (tile, wave) sub-range is padded to whole
//     wave iterations (64 entries, dummies at the end) and stored interleaved, so a wave
//     iteration is one coalesced 256-byte load of packed words + one of parameters;
//   * a packed word holds the two LDS byte addresses directly (row address << 17 | panel
//     offset): unpacking is one shift and one mask, the LDS region bases are instruction
//     immediates;
//   * one 16-wave workgroup owns a row block and each wave owns a fixed range of its rows: the
//     accumulator update is a plain LDS read-add-write by the only wave that ever touches that
//     row -- no atomics anywhere, one writer per gradient row, fixed summation order, so the
//     result is bitwise reproducible run to run.
#include <hipcub/hipcub.hpp>

#include "mde_common.h"
#include "mde_functions.h"
#include "mde_plan.h"
#define COMMA ,

#define MDE_LDS_BYTES 163840
// Static LDS map (bytes): [0, GR) x_v of the block's rows, [GR, XC) their gradient accumulators,
// [XC, XC + 96 KiB) the x_u panel.  The region bases are compile-time constants below 2^16 so the
// kernel's LDS instructions carry them as immediate offsets.
#define MDE_PANEL_GR_OFF 32752
#define MDE_PANEL_XC_OFF 65504
#define MDE_PANEL_XC_BYTES 98304
#define MDE_PANEL_DUMMY ((uint32_t)MDE_PANEL_GR_OFF << 17)  // padding entry: row address past every real row
#define MDE_PANEL_CB_VALUES 8   // parameter codebook entries (3 free low bits of the packed word at d = 2)
#define MDE_PANEL_CB_OFF (MDE_PANEL_XC_OFF + MDE_PANEL_XC_BYTES)  // LDS: the value table (32 bytes)

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

// sub_ptr[t * NW + w]: the row-sorted half-edges of tile t are cut into NW contiguous slices of
// (nearly) equal length, one per wave of the workgroup, each cut moved forward to the next row
// boundary so that a row's entries never straddle two waves: inside a tile every accumulator
// row has exactly one writer (no LDS atomics, fixed summation order), tiles are separated by
// workgroup barriers, and all waves of a tile run the same number of iterations.
__global__ __launch_bounds__(MDE_BLOCK) void k_sub_ptr(int64_t H, uint32_t ntiles, int NW,
                                                       const int32_t* __restrict__ tile_ptr,
                                                       const uint32_t* __restrict__ keys,
                                                       int32_t* __restrict__ sub_ptr) {
  const int64_t total = (int64_t)ntiles * NW;
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i <= total;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    if (i == total) {
      sub_ptr[i] = (int32_t)H;
      continue;
    }
    const uint32_t t = (uint32_t)(i / NW), w = (uint32_t)(i % NW);
    const int64_t beg = tile_ptr[t], end = tile_ptr[t + 1];
    int64_t pos = beg + ((end - beg) * (int64_t)w) / NW;
    // (keys inside a tile differ only in the row bits, so equal keys == equal rows)
    while (pos > beg && pos < end && keys[pos] == keys[pos - 1]) ++pos;
    sub_ptr[i] = (int32_t)pos;
  }
}

// next_tile[rb * (NP + 1) + cp] = first non-empty panel >= cp of row block rb (NP if none): the
// kernel skips empty tiles with one scalar load instead of a search loop
__global__ __launch_bounds__(MDE_BLOCK) void k_next_tile(int NRB, int NP, const int32_t* __restrict__ tile_ptr,
                                                         int32_t* __restrict__ next_tile) {
  const int rb = blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (rb >= NRB) return;
  int nxt = NP;
  next_tile[(size_t)rb * (NP + 1) + NP] = NP;
  for (int cp = NP - 1; cp >= 0; --cp) {
    const size_t t = (size_t)rb * NP + cp;
    if (tile_ptr[t] != tile_ptr[t + 1]) nxt = cp;
    next_tile[(size_t)rb * (NP + 1) + cp] = nxt;
  }
}

// K_i = wave iterations of sub-range i (0 for the extra terminal entry)
__global__ __launch_bounds__(MDE_BLOCK) void k_sub_iters(int64_t nsub, const int32_t* __restrict__ sub_ptr,
                                                         int32_t* __restrict__ iters) {
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i <= nsub;
       i += (int64_t)gridDim.x * MDE_BLOCK)
    iters[i] = (i < nsub) ? (sub_ptr[i + 1] - sub_ptr[i] + 63) >> 6 : 0;
}

// A (tile, wave) sub-range of m row-sorted half-edges is processed in K = ceil(m / 64) wave
// iterations and stored padded to 64 K entries starting at 64 * sub_off: element s sits at
// iteration s % K, lane s / K, so two half-edges of the same row share an iteration only when
// the row has more than K entries in the tile (the common case needs no run folding) and the
// dummies of an iteration are its highest lanes.  The packed word is ready-made LDS addressing:
// (byte address of the row's slot in the padded row arrays) << 17 | byte offset of x_u in the panel.
__global__ __launch_bounds__(MDE_BLOCK) void k_panel_pack(int64_t nsub, const int32_t* __restrict__ sub_ptr,
                                                          const int32_t* __restrict__ sub_off,
                                                          const uint32_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ vals,
                                                          const int32_t* __restrict__ nbr,
                                                          const int32_t* __restrict__ eid, int P_C, int NP,
                                                          int KR_shift, int d, uint32_t* __restrict__ packed,
                                                          int32_t* __restrict__ peid) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * MDE_BLOCK) >> 6;
  for (int64_t i = w0; i < nsub; i += nw) {
    const int beg = sub_ptr[i], m = sub_ptr[i + 1] - beg;
    if (m <= 0) continue;
    const int K = (m + 63) >> 6;
    const int64_t base = (int64_t)sub_off[i] * 64;
    for (int sidx = lane; sidx < m; sidx += 64) {
      const uint32_t key = keys[beg + sidx], q = vals[beg + sidx];
      const uint32_t rl = key & ((1u << KR_shift) - 1u);
      const uint32_t cp = (key >> KR_shift) % (uint32_t)NP;
      const uint32_t cl = (uint32_t)nbr[q] - cp * (uint32_t)P_C;
      const int64_t pos = base + (int64_t)(sidx % K) * 64 + sidx / K;
      packed[pos] = (((rl + (rl >> 5)) * 4u * (uint32_t)d) << 17) | (cl * 4u * (uint32_t)d);
      peid[pos] = eid[q];
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
  // rows: about one block per CU (256), multiple of 64; the padded row array (one spare slot
  // per 32 rows against bank conflicts) must fit its 32752-byte region
  // (a rank that owns only n/N rows keeps the same block height as a full plan -- tiles must
  // not get thinner -- and fills the CUs with Q column groups per row block instead)
  int64_t pr = (plan->n + 255) / 256;
  if (pr > nloc) pr = nloc;
  pr = ((pr + 63) / 64) * 64;
  const int64_t pr_max = ((int64_t)(MDE_PANEL_GR_OFF / (4 * d)) * 32 / 33) / 64 * 64;
  if (pr > pr_max) pr = pr_max;
  if (pr < 64) pr = 64;
  int64_t pc = MDE_PANEL_XC_BYTES / (4 * d);  // = MDE_PANEL_STG float4 per thread
  if (pc > 65535) pc = 65535;
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
  int32_t *peid = nullptr, *tile_ptr = nullptr, *sub_ptr = nullptr, *sub_off = nullptr, *iters = nullptr;
  int32_t* next_tile = nullptr;
  hipError_t e = hipSuccess;
  auto release = [&]() {
    if (keys) (void)hipFree(keys);
    if (vals) (void)hipFree(vals);
    if (keys2) (void)hipFree(keys2);
    if (vals2) (void)hipFree(vals2);
    if (tmp) (void)hipFree(tmp);
    if (packed) (void)hipFree(packed);
    if (peid) (void)hipFree(peid);
    if (tile_ptr) (void)hipFree(tile_ptr);
    if (sub_ptr) (void)hipFree(sub_ptr);
    if (sub_off) (void)hipFree(sub_off);
    if (iters) (void)hipFree(iters);
    if (next_tile) (void)hipFree(next_tile);
  };
  auto fail = [&](hipError_t err, const char* what) {
    release();
    return mde_hip_fail(err, what, __FILE__, __LINE__);
  };
  const size_t hb = (size_t)H * sizeof(uint32_t);
  const int64_t nsub = (int64_t)ntiles * MDE_PANEL_WAVES;
#define PB(call)                                   \
  do {                                             \
    e = (call);                                    \
    if (e != hipSuccess) return fail(e, #call);    \
  } while (0)
  PB(hipMalloc(&keys, hb));
  PB(hipMalloc(&vals, hb));
  PB(hipMalloc(&keys2, hb));
  PB(hipMalloc(&vals2, hb));
  PB(hipMalloc(&tile_ptr, ((size_t)ntiles + 1) * sizeof(int32_t)));
  PB(hipMalloc(&next_tile, (size_t)NRB * (NP + 1) * sizeof(int32_t)));
  PB(hipMalloc(&sub_ptr, ((size_t)nsub + 1) * sizeof(int32_t)));
  PB(hipMalloc(&sub_off, ((size_t)nsub + 1) * sizeof(int32_t)));
  PB(hipMalloc(&iters, ((size_t)nsub + 1) * sizeof(int32_t)));
  hipLaunchKernelGGL(k_panel_keys, dim3(mde_grid(nloc * 16, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st,
                     (int)nloc, plan->rowptr, plan->nbr, P_R, P_C, NP, KR_shift, keys, vals);
  PB(hipGetLastError());
  size_t tmp_bytes = 0, scan_bytes = 0;
  const int end_bit = bits_for_u64((uint64_t)ntiles) + KR_shift;
  PB(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (int)H, 0,
                                        end_bit > 32 ? 32 : end_bit, st));
  PB(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, iters, sub_off, (int)(nsub + 1), st));
  if (scan_bytes > tmp_bytes) tmp_bytes = scan_bytes;
  PB(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  PB(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (int)H, 0,
                                        end_bit > 32 ? 32 : end_bit, st));
  hipLaunchKernelGGL(k_tile_ptr, dim3(mde_grid(H + 1, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, H,
                     ntiles, KR_shift, keys2, tile_ptr);
  PB(hipGetLastError());
  hipLaunchKernelGGL(k_next_tile, dim3((NRB + MDE_BLOCK - 1) / MDE_BLOCK), dim3(MDE_BLOCK), 0, st, NRB, NP,
                     tile_ptr, next_tile);
  PB(hipGetLastError());
  hipLaunchKernelGGL(k_sub_ptr, dim3(mde_grid(nsub + 1, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, H,
                     ntiles, MDE_PANEL_WAVES, tile_ptr, keys2, sub_ptr);
  PB(hipGetLastError());
  hipLaunchKernelGGL(k_sub_iters, dim3(mde_grid(nsub + 1, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, nsub,
                     sub_ptr, iters);
  PB(hipGetLastError());
  PB(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, iters, sub_off, (int)(nsub + 1), st));
  int32_t total_iters = 0;
  PB(hipMemcpyAsync(&total_iters, sub_off + nsub, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  PB(hipStreamSynchronize(st));
  const int64_t Hp = (int64_t)total_iters * 64;  // padded half-edge count
  if (total_iters <= 0 || Hp >= ((int64_t)1 << 31) - 64) {
    release();
    return 0;  // too large for 32-bit positions: the caller keeps the CSR layout
  }
  PB(hipMalloc(&packed, (size_t)Hp * sizeof(uint32_t)));
  PB(hipMalloc(&peid, (size_t)Hp * sizeof(int32_t)));
  PB(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(packed), (int)MDE_PANEL_DUMMY, (size_t)Hp, st));
  PB(hipMemsetAsync(peid, 0xFF, (size_t)Hp * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_panel_pack, dim3(mde_grid(nsub * 64, MDE_BLOCK, 8192)), dim3(MDE_BLOCK), 0, st, nsub,
                     sub_ptr, sub_off, keys2, vals2, plan->nbr, plan->eid, P_C, NP, KR_shift, d, packed, peid);
  PB(hipGetLastError());
  PB(hipStreamSynchronize(st));
#undef PB
  (void)hipFree(keys);
  (void)hipFree(vals);
  (void)hipFree(keys2);
  (void)hipFree(vals2);
  (void)hipFree(tmp);
  (void)hipFree(sub_ptr);
  (void)hipFree(iters);
  (void)hipFree(tile_ptr);
  if (L.packed) (void)hipFree(L.packed);
  if (L.eid) (void)hipFree(L.eid);
  if (L.next_tile) (void)hipFree(L.next_tile);
  if (L.sub_off) (void)hipFree(L.sub_off);
  L.sub_off = sub_off;
  L.next_tile = next_tile;
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
  L.d = d;
  L.rows_per_block = P_R;
  L.cols_per_panel = P_C;
  L.n_row_blocks = NRB;
  L.n_panels = NP;
  L.H = Hp;
  L.packed = packed;
  L.eid = peid;
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
    out[q] = eid[q] >= 0 ? in[eid[q]] : 1.0f;  // padding entries carry a harmless parameter
}

// number of entries of a per-half-edge parameter array in the given layout (layout 1: the padded
// stream + MDE_PANEL_CB_VALUES spare entries, where a codebook stream keeps its value table)
extern "C" int64_t mde_plan_layout_half_edges(const mde_plan* plan, int32_t layout) {
  if (!plan) return 0;
  return (layout == 1 && plan->panel.packed) ? plan->panel.H + MDE_PANEL_CB_VALUES : plan->H;
}

// ---------------------------------------------------------------- parameter codebooks
// Neighbour-graph problems carry very few distinct per-edge parameters (k-NN weights 1 / 2, -1 for
// repulsive pairs).  At d = 2 the packed word's panel offset is a multiple of 8, so its 3 low bits
// can hold an index into a table of <= 8 values: the kernel then streams 4 bytes per half-edge
// instead of 8 (packed word + fp32 parameter) and looks the parameter up in LDS.
#define MDE_CB_EMPTY 0xFFFFFFFFu  // (a NaN pattern: NaN parameters simply disable the codebook)

// distinct bit patterns of in[0..p): inserted into table[0..8) with compare-and-swap; *overflow is
// set when a 9th value (or the EMPTY pattern) shows up
__global__ __launch_bounds__(MDE_BLOCK) void k_codebook_scan(int64_t p, const float* __restrict__ in,
                                                             unsigned int* __restrict__ table,
                                                             int* __restrict__ overflow) {
  __shared__ unsigned int stb[MDE_PANEL_CB_VALUES];
  if (threadIdx.x < MDE_PANEL_CB_VALUES) stb[threadIdx.x] = MDE_CB_EMPTY;
  __syncthreads();
  unsigned int last0 = MDE_CB_EMPTY, last1 = MDE_CB_EMPTY;
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < p;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    const unsigned int v = __float_as_uint(in[i]);
    if (v == last0 || v == last1) continue;
    last1 = last0;
    last0 = v;
    bool known = false;
#pragma unroll
    for (int s = 0; s < MDE_PANEL_CB_VALUES; ++s) known |= (stb[s] == v);
    if (known) continue;
    if (v == MDE_CB_EMPTY || *reinterpret_cast<volatile int*>(overflow)) {
      *overflow = 1;
      return;
    }
    bool placed = false;
    for (int s = 0; s < MDE_PANEL_CB_VALUES && !placed; ++s) {
      const unsigned int old = atomicCAS(&table[s], MDE_CB_EMPTY, v);
      placed = (old == MDE_CB_EMPTY || old == v);
    }
    if (!placed) {
      *overflow = 1;
      return;
    }
    for (int s = 0; s < MDE_PANEL_CB_VALUES; ++s) {  // remember it block-wide
      const unsigned int old = atomicCAS(&stb[s], MDE_CB_EMPTY, v);
      if (old == MDE_CB_EMPTY || old == v) break;
    }
  }
}

// out[q] = packed[q] | index of in[eid[q]] in table (padding entries stay MDE_PANEL_DUMMY)
__global__ __launch_bounds__(MDE_BLOCK) void k_codebook_pack(int64_t H, const uint32_t* __restrict__ packed,
                                                             const int32_t* __restrict__ eid,
                                                             const float* __restrict__ in,
                                                             const unsigned int* __restrict__ table,
                                                             uint32_t* __restrict__ out) {
  unsigned int tb[MDE_PANEL_CB_VALUES];
#pragma unroll
  for (int s = 0; s < MDE_PANEL_CB_VALUES; ++s) tb[s] = table[s];
  for (int64_t q = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; q < H;
       q += (int64_t)gridDim.x * MDE_BLOCK) {
    uint32_t w = packed[q];
    if (eid[q] >= 0) {
      const unsigned int v = __float_as_uint(in[eid[q]]);
      uint32_t idx = 0;
#pragma unroll
      for (int s = 1; s < MDE_PANEL_CB_VALUES; ++s) idx = (tb[s] == v) ? (uint32_t)s : idx;
      w |= idx;
    }
    out[q] = w;
  }
}

// Try to put a per-edge parameter array into codebook form for layout 1.  On success
// (*n_values_host in 1..8) out_half holds the H packed words with the value index in their 3 low
// bits, followed by the 8-entry value table; pass it as mde_func.a0 with a0_scalar = 2.
// *n_values_host = 0: not applicable (d != 2, more than 8 distinct values, NaNs) -- nothing is
// written and the caller uses mde_plan_expand_layout.  SYNC.
extern "C" int mde_plan_expand_codebook(const mde_plan* plan, const float* in_edge, float* out_half,
                                        int32_t* n_values_host, void* stream) {
  if (!plan || !in_edge || !out_half || !n_values_host) return MDE_E_INVALID;
  *n_values_host = 0;
  const mde_panel_layout& L = plan->panel;
  if (!L.packed || L.d != 2 || L.H == 0 || plan->p == 0) return MDE_OK;
  const char* e = getenv("MDE_CODEBOOK");
  if (e && atoi(e) == 0) return MDE_OK;
  hipStream_t st = mde_stream(stream);
  unsigned int* table = reinterpret_cast<unsigned int*>(out_half) + L.H;  // the spare entries
  int* overflow = nullptr;
  MDE_HIP(hipMalloc(&overflow, sizeof(int)));
  hipError_t err = hipMemsetAsync(overflow, 0, sizeof(int), st);
  if (err == hipSuccess) err = hipMemsetAsync(table, 0xFF, MDE_PANEL_CB_VALUES * sizeof(unsigned int), st);
  unsigned int host_tb[MDE_PANEL_CB_VALUES];
  int host_overflow = 0;
  if (err == hipSuccess) {
    hipLaunchKernelGGL(k_codebook_scan, dim3(mde_grid(plan->p, MDE_BLOCK, 2048)), dim3(MDE_BLOCK), 0, st, plan->p,
                       in_edge, table, overflow);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipMemcpyAsync(host_tb, table, sizeof(host_tb), hipMemcpyDeviceToHost, st);
  if (err == hipSuccess) err = hipMemcpyAsync(&host_overflow, overflow, sizeof(int), hipMemcpyDeviceToHost, st);
  if (err == hipSuccess) err = hipStreamSynchronize(st);
  (void)hipFree(overflow);
  if (err != hipSuccess) return mde_hip_fail(err, "parameter codebook scan", __FILE__, __LINE__);
  if (host_overflow) return MDE_OK;
  // canonical order (the insertion order above depends on scheduling): ascending bit patterns
  int nv = 0;
  unsigned int vals[MDE_PANEL_CB_VALUES];
  for (int s = 0; s < MDE_PANEL_CB_VALUES; ++s)
    if (host_tb[s] != MDE_CB_EMPTY) vals[nv++] = host_tb[s];
  if (nv == 0) return MDE_OK;
  for (int a = 1; a < nv; ++a)
    for (int b = a; b > 0 && vals[b - 1] > vals[b]; --b) {
      const unsigned int t = vals[b];
      vals[b] = vals[b - 1];
      vals[b - 1] = t;
    }
  for (int s = nv; s < MDE_PANEL_CB_VALUES; ++s) vals[s] = MDE_CB_EMPTY;
  MDE_HIP(hipMemcpyAsync(table, vals, sizeof(vals), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_codebook_pack, dim3(mde_grid(L.H, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, L.H, L.packed,
                     L.eid, in_edge, table, reinterpret_cast<uint32_t*>(out_half));
  MDE_LAUNCH_CHECK();
  MDE_HIP(hipStreamSynchronize(st));  // `vals` is a stack buffer
  *n_values_host = nv;
  return MDE_OK;
}

extern "C" int mde_plan_expand_layout(const mde_plan* plan, int32_t layout, const float* in_edge,
                                      float* out_half, void* stream) {
  if (!plan || !in_edge || !out_half) return MDE_E_INVALID;
  if (layout == 0) return mde_plan_expand(plan, in_edge, out_half, stream);
  if (layout != 1 || !plan->panel.eid) {
    mde_set_error("mde_plan_expand_layout: the panel layout has not been built");
    return MDE_E_INVALID;
  }
  if (plan->panel.H == 0) return MDE_OK;
  hipLaunchKernelGGL(k_expand_panel, dim3(mde_grid(plan->panel.H, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0,
                     mde_stream(stream), plan->panel.H, plan->panel.eid, in_edge, out_half);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// ---------------------------------------------------------------- the kernel
// 1024 threads = 16 waves per workgroup; inside a tile wave w owns the rows of its slice
// (k_sub_ptr): only it reads-modifies-writes their LDS accumulators between two workgroup
// barriers, so no atomics are needed and the order of additions is fixed by the (sorted) tile
// order.  Inside one wave iteration lanes
// that hit the same row are adjacent (tiles are sorted by row): when that happens (rare, see
// k_panel_pack) the contributions are folded into the last lane of each run with DPP wave
// shifts, and that lane performs the single read-add-write of the row.
//
// The software pipeline across tiles is described at the staging registers below.
#define MDE_PANEL_BS (64 * MDE_PANEL_WAVES)
#define MDE_PANEL_STG (MDE_PANEL_XC_BYTES / 16 / MDE_PANEL_BS)  // float4 staging registers per thread
#define MDE_PANEL_MAXI 6  // prefetched wave-iterations per (tile, wave); longer sub-ranges load in place

typedef float panel_f4 __attribute__((ext_vector_type(4)));  // plain vector: no struct copies

template <int D>
struct PanelVec;
template <>
struct PanelVec<1> {
  typedef float T;
};
template <>
struct PanelVec<2> {
  typedef float2 T;
};
template <>
struct PanelVec<4> {
  typedef float4 T;
};
// D floats at an LDS byte address (aligned to the vector size for D = 1, 2, 4)
template <int D>
__device__ __forceinline__ void panel_ld(const char* p, float (&v)[D]) {
  if constexpr (D == 3) {
    const float* q = reinterpret_cast<const float*>(p);
    v[0] = q[0];
    v[1] = q[1];
    v[2] = q[2];
  } else {
    const typename PanelVec<D>::T t = *reinterpret_cast<const typename PanelVec<D>::T*>(p);
    const float* q = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int c = 0; c < D; ++c) v[c] = q[c];
  }
}
template <int D>
__device__ __forceinline__ void panel_st(char* p, const float (&v)[D]) {
  if constexpr (D == 3) {
    float* q = reinterpret_cast<float*>(p);
    q[0] = v[0];
    q[1] = v[1];
    q[2] = v[2];
  } else {
    typename PanelVec<D>::T t;
    float* q = reinterpret_cast<float*>(&t);
#pragma unroll
    for (int c = 0; c < D; ++c) q[c] = v[c];
    *reinterpret_cast<typename PanelVec<D>::T*>(p) = t;
  }
}

// CB: the first parameter comes from a codebook -- `packed` is the stream with value indices in its
// 3 low bits (mde_plan_expand_codebook), a0 the 8-entry value table; no parameter stream is read.
template <int D, class Fn, bool HAS_GRAD, bool CB>
__global__ __launch_bounds__(MDE_PANEL_BS) void k_fused_panel(
    int nloc, int row_lo, int n, int P_R, int P_C, int NP, int Q, const int32_t* __restrict__ next_tile,
    const int32_t* __restrict__ sub_off, const uint32_t* __restrict__ packed,
    const float* __restrict__ a0, const float* __restrict__ a1, int a0_scalar, int a1_scalar,
    const float* __restrict__ X, float* __restrict__ grad, float* __restrict__ partial,
    double* __restrict__ loss_partials, Fn fn, float inv_p, float grad_scale,
    float* __restrict__ loss_out, double loss_scale) {
  constexpr int BS = MDE_PANEL_BS, NW = MDE_PANEL_WAVES, STG = MDE_PANEL_STG, MAXI = MDE_PANEL_MAXI;
  constexpr int GR_OFF = MDE_PANEL_GR_OFF, XC_OFF = MDE_PANEL_XC_OFF;
  constexpr uint32_t DUMMY = MDE_PANEL_DUMMY;
  // statically sized: the LDS addresses unpacked from the stream are absolute
  __shared__ __attribute__((aligned(16))) char L[MDE_PANEL_CB_OFF + 4 * MDE_PANEL_CB_VALUES];
  float* XR = reinterpret_cast<float*>(L);            // x_v of the block's rows (padded slots)
  float* GR = reinterpret_cast<float*>(L + GR_OFF);   // gradient accumulators (same slots)
  float* XC = reinterpret_cast<float*>(L + XC_OFF);   // x_u of the current panel
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned ulane = (unsigned)lane;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keep it scalar
  // block b = rb * Q + qg: row block rb, column group qg walks panels [cp_lo, cp_hi)
  const int rb = blockIdx.x / Q, qg = blockIdx.x % Q;
  const int cp_lo = (int)(((int64_t)NP * qg) / Q), cp_hi = (int)(((int64_t)NP * (qg + 1)) / Q);
  const int r0 = rb * P_R;
  const int nr = min(P_R, nloc - r0);
  const float a0s = (a0_scalar && !CB) ? a0[0] : 1.0f;
  if (CB && tid < MDE_PANEL_CB_VALUES) reinterpret_cast<float*>(L + MDE_PANEL_CB_OFF)[tid] = a0[tid];
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;
  const bool a1_arr = a1 && !a1_scalar;
  const float* Xrow = X + (size_t)(row_lo + r0) * D;
  {
    panel_f4* z = reinterpret_cast<panel_f4*>(L);
    for (int i = tid; i < XC_OFF / 16; i += BS) z[i] = panel_f4{0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();
  for (int r = tid; r < nr; r += BS) {
    const int slot = (r + (r >> 5)) * D;
#pragma unroll
    for (int c = 0; c < D; ++c) XR[slot + c] = Xrow[r * D + c];
  }
  float loss = 0.0f;
  const int32_t* nt = next_tile + (size_t)rb * (NP + 1);
  const int32_t* sp_base = sub_off + (size_t)rb * NP * NW + wave;

  // one wave iteration: 64 packed half-edges (padding entries compute on harmless addresses and
  // are masked out of the loss and of the accumulator update)
  auto process = [&](uint32_t pk, float p0, float p1) __attribute__((always_inline)) {
    const bool active = pk != DUMMY;
    const uint32_t rowaddr = pk >> 17, coladdr = pk & (CB ? 0x1fff8u : 0x1ffffu);
    if (CB) p0 = *reinterpret_cast<const float*>(L + MDE_PANEL_CB_OFF + ((pk & 7u) << 2));
    float xr[D], xc[D], v[D], ss = 0.0f;
    panel_ld<D>(L + rowaddr, xr);
    panel_ld<D>(L + XC_OFF + coladdr, xc);
#pragma unroll
    for (int c = 0; c < D; ++c) {
      v[c] = xr[c] - xc[c];
      ss = fmaf(v[c], v[c], ss);
    }
    float f, gd;
    fn.eval(ss, p0, p1, f, gd);
    const float g = mde_fix_g(gd * inv_p);
    loss += active ? f : 0.0f;
    if (!HAS_GRAD) return;
#pragma unroll
    for (int c = 0; c < D; ++c) v[c] *= g;
    // Lanes of one iteration hold ascending rows (row-sorted tile, interleaved storage), so equal
    // rows are adjacent lanes; padding lanes are the highest ones.
    const int key = (int)rowaddr;
    bool tail = active;
    const int kprev = __builtin_amdgcn_update_dpp(-1, key, 0x138, 0xf, 0xf, false);  // wave_shr:1
    if (__any(kprev == key && active)) {
      // rare: a row has more entries in this tile than the wave has iterations.  Round r adds
      // the ORIGINAL contribution of lane i-r when it has the same row (keys / values shifted
      // one lane per round with DPP wave_shr); the last lane of each run writes.
      int kc = key;
      float sv[D];
#pragma unroll
      for (int c = 0; c < D; ++c) sv[c] = v[c];
#pragma nounroll
      for (int r = 1; r < 64; ++r) {
        kc = __builtin_amdgcn_update_dpp(-1, kc, 0x138, 0xf, 0xf, false);
        const bool m = (kc == key);
        if (!__any(m && active)) break;
#pragma unroll
        for (int c = 0; c < D; ++c) {
          sv[c] = __int_as_float(
              __builtin_amdgcn_update_dpp(0, __float_as_int(sv[c]), 0x138, 0xf, 0xf, false));
          v[c] += m ? sv[c] : 0.0f;
        }
      }
      const int knext = __builtin_amdgcn_update_dpp(-1, key, 0x130, 0xf, 0xf, false);  // wave_shl:1
      tail = active && (knext != key);
    }
    if (tail) {
      float acc[D];
      panel_ld<D>(L + GR_OFF + rowaddr, acc);
#pragma unroll
      for (int c = 0; c < D; ++c) acc[c] += v[c];
      panel_st<D>(L + GR_OFF + rowaddr, acc);
    }
  };

  // Pipeline across tiles (all loads are unconditional and clamped, never predicated, so that the
  // compiler's vmcnt bookkeeping is exact on every path and no wait covers a load just issued):
  //   * x_u panels are prefetched TWO tiles ahead through two sets of staging registers: while tile
  //     t is processed out of LDS, panel t+1 already sits in one set and panel t+2 is being loaded
  //     into the set that tile t's commit has just freed -- a full tile of slack for the L2 / fabric
  //     latency;
  //   * the wave's stream registers are refilled in place: right after wave iteration k of tile t
  //     has consumed its packed word and parameter, the same registers receive iteration k of tile
  //     t+1;
  //   * loads are issued a few per wave iteration instead of in one burst after the barrier (the
  //     16 waves would otherwise queue on the CU's single address unit while nobody computes).
  panel_f4 stgA[STG], stgB[STG];
  uint32_t pk[MAXI];
  float wv[MAXI];
  const unsigned wlane = (a0_scalar || CB) ? 0u : ulane;
  const int wstride = (a0_scalar || CB) ? 0 : 64;
  struct PanelSrc {
    const panel_f4* s4;
    int last;
    bool full;
  };
  auto panel_src = [&](int cp) __attribute__((always_inline)) {
    const int c0 = cp * P_C;
    const int t4 = (min(P_C, n - c0) * D) >> 2;  // c0 * D * 4 bytes is 16-byte aligned (P_C multiple of 256)
    PanelSrc r;
    r.s4 = reinterpret_cast<const panel_f4*>(X + (size_t)c0 * D);
    r.last = t4 > 0 ? t4 - 1 : 0;
    r.full = t4 == STG * BS;
    return r;
  };
  auto load_panel_part = [&](const PanelSrc& src, panel_f4 (&stg)[STG], int j0, int j1)
                             __attribute__((always_inline)) {
#pragma unroll
    for (int j = j0; j < j1 && j < STG; ++j)
      stg[j] = src.full ? src.s4[tid + j * BS] : src.s4[min(tid + j * BS, src.last)];
  };
  auto store_panel = [&](int cp, const panel_f4 (&stg)[STG]) __attribute__((always_inline)) {
    const int c0 = cp * P_C;
    const int nc = min(P_C, n - c0);
    const int t4 = (nc * D) >> 2;
    panel_f4* d4 = reinterpret_cast<panel_f4*>(XC);
    if (t4 == STG * BS) {
#pragma unroll
      for (int k = 0; k < STG; ++k) d4[tid + k * BS] = stg[k];
    } else {
#pragma unroll
      for (int k = 0; k < STG; ++k) {
        const int i = tid + k * BS;
        if (i < t4) d4[i] = stg[k];
      }
      const float* src = X + (size_t)c0 * D;
      for (int i = (t4 << 2) + tid; i < nc * D; i += BS) XC[i] = src[i];  // < 4 tail floats
    }
  };
  struct StreamSrc {
    const uint32_t* pb;
    const float* ab;
    int K, b, kmax;
  };
  auto stream_src = [&](int cp) __attribute__((always_inline)) {
    const int32_t* sp = sp_base + (size_t)cp * NW;
    StreamSrc r;
    r.b = __builtin_amdgcn_readfirstlane(sp[0]);
    r.K = __builtin_amdgcn_readfirstlane(sp[1]) - r.b;
    const int bb = r.K > 0 ? r.b : 0;
    r.kmax = r.K > 0 ? r.K - 1 : 0;
    r.pb = packed + (size_t)bb * 64;
    r.ab = a0 + (size_t)bb * wstride;
    return r;
  };
  // slot k of the stream of a tile (index clamped into the slice: slots >= K are never consumed)
  auto load_stream_slot = [&](const StreamSrc& src, int k) __attribute__((always_inline)) {
    const int kk = min(k, src.kmax);  // wave-uniform: scalar base + the lane as the only vector offset
    pk[k] = (src.pb + (size_t)kk * 64)[ulane];
    if (!CB) wv[k] = (src.ab + (size_t)kk * wstride)[wlane];
  };
  auto next_nonempty = [&](int cp) __attribute__((always_inline)) {  // cp <= NP
    return __builtin_amdgcn_readfirstlane(nt[cp]);
  };
  // One tile.  cp: this tile (panel in `scur`), cpn: the next one (panel already in the other
  // set; may be >= cp_hi).  Loads panel cpn2 (the tile after cpn) into `scur` once it is free.
  auto tile_step = [&](int cp, int cpn, int cK, int cb, panel_f4 (&scur)[STG], int& nK, int& nb)
                       __attribute__((always_inline)) {
    __syncthreads();  // everyone is done with the previous panel (and XR/GR are initialised)
    store_panel(cp, scur);
    __syncthreads();
    const int cpn2 = cpn < cp_hi ? next_nonempty(cpn + 1) : cpn;
    // past the end: redundant reloads of this tile keep the number of loads path-independent
    const PanelSrc psrc = panel_src(cpn2 < cp_hi ? cpn2 : cp);
    const StreamSrc ssrc = stream_src(cpn < cp_hi ? cpn : cp);
    nK = ssrc.K;
    nb = ssrc.b;
    // panel loads per slot: all issued in the first 3 slots (spreading them over all 6 measured
    // 8 % slower: the last ones then land too close to the next commit)
    constexpr int PPS = (STG + 2) / 3;
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
      load_panel_part(psrc, scur, k * PPS, (k + 1) * PPS);
      if (k < cK) {
        const float p1 = a1_arr ? a1[(size_t)(cb + k) * 64 + ulane] : a1s;
        process(pk[k], CB ? 0.0f : wv[k], p1);
      }
      load_stream_slot(ssrc, k);
    }
    for (int k = MAXI; k < cK; ++k) {  // oversized slices (skewed degrees): not prefetched
      const size_t h = (size_t)(cb + k) * 64 + ulane;
      const float p0 = (a0_scalar || CB) ? a0s : a0[h];
      const float p1 = a1_arr ? a1[h] : a1s;
      process(packed[h], p0, p1);
    }
    return cpn2;
  };

  int cp = next_nonempty(cp_lo);
  if (cp < cp_hi) {
    int cpn = next_nonempty(cp + 1);
    int cK, cb, nK = 0, nb = 0;
    {
      load_panel_part(panel_src(cp), stgA, 0, STG);
      load_panel_part(panel_src(cpn < cp_hi ? cpn : cp), stgB, 0, STG);
      const StreamSrc s0 = stream_src(cp);
      cK = s0.K;
      cb = s0.b;
#pragma unroll
      for (int k = 0; k < MAXI; ++k) load_stream_slot(s0, k);
    }
    // the first tile is peeled so that the loop is entered in the steady state of the pipeline
    // (same loads in flight as after any other tile: exact wait counts inside the loop)
    int cpn2 = tile_step(cp, cpn, cK, cb, stgA, nK, nb);
    cp = cpn, cpn = cpn2, cK = nK, cb = nb;
    while (cp < cp_hi) {
      cpn2 = tile_step(cp, cpn, cK, cb, stgB, nK, nb);
      cp = cpn, cpn = cpn2, cK = nK, cb = nb;
      if (cp >= cp_hi) break;
      cpn2 = tile_step(cp, cpn, cK, cb, stgA, nK, nb);
      cp = cpn, cpn = cpn2, cK = nK, cb = nb;
    }
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
  // block-wide loss partial (the x_v region is free now), then the loss itself: the last
  // workgroup to arrive adds the partials of all of them in a fixed order (no second launch)
  double* red = reinterpret_cast<double*>(L);
  int* last_flag = reinterpret_cast<int*>(L + 256);
  unsigned int* ticket = reinterpret_cast<unsigned int*>(loss_partials + MDE_MAX_PARTIALS);
  const double v = mde_wave_sum((double)loss);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
    for (int i = 0; i < NW; ++i) s += red[i];
    loss_partials[blockIdx.x] = s;
    __threadfence();  // publish the partial (and this block's gradient rows) device-wide
    *last_flag = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1 : 0;
  }
  __syncthreads();
  if (*last_flag) {
    __threadfence();  // see the other workgroups' partials (L2 is not coherent across XCDs)
    double t = 0.0;
    for (int i = tid; i < (int)gridDim.x; i += BS) t += __builtin_nontemporal_load(loss_partials + i);
    t = mde_wave_sum(t);
    __syncthreads();
    if (lane == 0) red[wave] = t;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int i = 0; i < NW; ++i) s += red[i];
      *loss_out = (float)(s * loss_scale);
      *ticket = 0u;  // ready for the next launch (stream order)
    }
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
  float* loss_out;
  double loss_scale;
};

template <int D, class Fn>
static int launch_panel(const PanelArgs& A, const Fn& fn, int* nblocks) {
  const mde_panel_layout& L = A.plan->panel;
  const bool cb = A.a0_scalar == 2;
  if (cb && D != 2) {
    mde_set_error("codebook parameter streams exist for d = 2 only");
    return MDE_E_INVALID;
  }
  auto kern = A.grad ? k_fused_panel<D, Fn, true, false> : k_fused_panel<D, Fn, false, false>;
  if constexpr (D == 2) {
    if (cb) kern = A.grad ? k_fused_panel<D, Fn, true, true> : k_fused_panel<D, Fn, false, true>;
  }
  // codebook form: a0 = [H packed words | 8 values]
  const uint32_t* stream = cb ? reinterpret_cast<const uint32_t*>(A.a0) : L.packed;
  const float* a0 = cb ? A.a0 + L.H : A.a0;
  const int Q = L.col_groups;
  *nblocks = L.n_row_blocks * Q;
  hipLaunchKernelGGL(kern, dim3(L.n_row_blocks * Q), dim3(MDE_PANEL_BS), 0, A.st,
                     (int)(A.plan->row_hi - A.plan->row_lo), (int)A.plan->row_lo, (int)A.plan->n,
                     L.rows_per_block, L.cols_per_panel, L.n_panels, Q, L.next_tile, L.sub_off, stream,
                     a0, A.a1, A.a0_scalar, A.a1_scalar, A.X, A.grad, L.partial, A.plan->partials, fn,
                     A.inv_p, A.grad_scale, A.loss_out, A.loss_scale);
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

// Called by mde_average_distortion.  Returns 1 when the panel kernel was launched (it also writes
// *loss_out = loss_scale * sum of the workgroups' partials; nblocks = number of partials), 0 when the caller should use the CSR kernel, < 0 on error.
int mde_panel_try(mde_plan* plan, const float* X, int d, const mde_func* f, float grad_scale,
                  float* grad, float inv_p, hipStream_t st, int* nblocks, float* loss_out,
                  double loss_scale) {
  if (!plan->panel.packed || plan->panel.d != d) return 0;
  PanelArgs A{plan, X, d, f->a0, f->a1, f->a0_scalar, f->a1_scalar, grad, inv_p, grad_scale, st,
              loss_out, loss_scale};
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
