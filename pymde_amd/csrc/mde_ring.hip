// mde_ring.hip -- LDS-resident variant of the fused average-distortion kernel for small
// embedding dimensions (d <= 4) on graphs whose embedding table does not fit L2 (layout 1).
// [ref: pymde/average_distortion.py:62-106 -- the same loss and gradient]
//
// Why: the CSR kernel's only random access is the gather of x_u, and random 8-byte gathers are
// bound by the L2 request rate (~100 G/s from an 8 MB table: >= 1 ms for 10^8 half-edges whatever
// the HBM rate).  Here every random access is served by LDS:
//
//   * a 1024-thread workgroup owns a block of R rows: their x_v and their fp32 gradient
//     accumulators stay in LDS for the whole kernel (2 x 32 KB);
//   * the embedding table streams through a ring of S chunk slots in the remaining 96 KB, filled
//     by LDS-DMA (global_load_lds_dwordx4: L2 -> LDS, no VGPRs, no ds_write) by two producer
//     waves;
//   * the other 14 waves are consumers.  Each owns a fixed range of the block's rows and walks
//     its own contiguous stream of packed half-edges (row address << 17 | ring offset, 4 bytes,
//     chunk-major) -- because a row has exactly one wave that ever touches its accumulator, the
//     update is a plain LDS read-add-write, the summation order is fixed by the layout (bitwise
//     reproducible), and the waves need NO workgroup barrier between prologue and epilogue;
//   * producers and consumers synchronise through a handful of LDS words: a consumer publishes
//     the oldest chunk it still reads (prog[w]), a producer the next chunk it has not landed yet
//     (F[p]); a consumer polls F only when its next iteration needs a chunk it has not seen
//     landed, a producer polls prog only when the slot it wants to refill may still be in use.
//     The waves of a workgroup therefore drift freely within the ring's slack, and staging,
//     stream loads, LDS traffic and VALU work of different waves overlap instead of alternating
//     in barrier-separated phases.
//   * LDS fp32 atomics would remove the one-writer rule, but ds_add_f32 retires ~0.16 lanes per
//     clock per CU on gfx950 (tools/ldsprobe: 397 clocks for two of them per wave) -- 20x slower
//     than read-add-write.  Duplicate rows inside a wave iteration (the 64 entries are sorted by
//     row, so they are adjacent lanes) are folded with DPP wave shifts instead; the number of
//     fold rounds of each iteration is known at build time and rides in its header word.
#include <hipcub/hipcub.hpp>

#include "mde_common.h"
#include "mde_functions.h"
#include "mde_plan.h"
#define COMMA ,

// Static LDS map (bytes).  The region bases are compile-time constants below 2^16, so the
// kernel's LDS instructions carry them as immediate offsets and the packed words hold absolute
// row addresses.
#define MDE_RING_XR_OFF 0          // x_v of the block's rows (+ one dummy slot), control words at the tail
#define MDE_RING_GR_OFF 32768      // gradient accumulators (same slots)
#define MDE_RING_OFF 65536         // the chunk ring
#define MDE_RING_BYTES 98304
#define MDE_RING_CTRL_OFF (32768 - 256)
#define MDE_RING_CTRL_PROG (MDE_RING_CTRL_OFF)        // int prog[16]: oldest chunk consumer w still reads
#define MDE_RING_CTRL_F (MDE_RING_CTRL_OFF + 64)      // int F[2]: next chunk producer p has not landed
#define MDE_RING_CTRL_CB (MDE_RING_CTRL_OFF + 96)     // float[8]: parameter codebook
#define MDE_RING_NCW 14            // consumer waves
#define MDE_RING_NPROD 2           // producer waves
#define MDE_RING_BS 1024
#define MDE_RING_DEPTH 3           // chunks in flight per producer
#ifndef MDE_RING_PF
#define MDE_RING_PF 8              // stream slots prefetched per consumer wave
#endif
#define MDE_RING_CB_VALUES 8
#define MDE_RING_DONE 0x7fffffff

// chunk geometry per embedding dimension: CBYTES bytes (PIECES x 1 KiB DMA pieces) per chunk
__host__ __device__ constexpr int ring_chunk_cols(int d) { return d == 1 ? 2048 : (d == 2 ? 1024 : 512); }
__host__ __device__ constexpr int ring_chunk_bytes(int d) { return ring_chunk_cols(d) * 4 * d; }
__host__ __device__ constexpr int ring_slots(int d) { return MDE_RING_BYTES / ring_chunk_bytes(d); }
// an iteration may reference chunks m .. m + span, span <= S - NPROD * DEPTH: the producers keep
// their full depth in flight while the slowest consumer sits on its window
__host__ __device__ constexpr int ring_max_span(int d) { return ring_slots(d) - MDE_RING_NPROD * MDE_RING_DEPTH; }

// header word of a wave iteration: [15:0] m = lowest chunk referenced, [19:16] span (highest = m +
// span), [25:20] DPP fold rounds (longest run of equal rows - 1), [26] the iteration has padding
#define MDE_RING_HDR(m, span, rounds, pad) ((uint32_t)(m) | ((uint32_t)(span) << 16) | ((uint32_t)(rounds) << 20) | ((uint32_t)(pad) << 26))

// ---------------------------------------------------------------- layout construction
// bounds[rb * (NCW + 1) + w]: consumer wave w of row block rb owns local rows [bounds[w],
// bounds[w + 1]) -- cut so that every wave gets the same number of half-edges
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_bounds(int nloc, int R, int NRB, const int32_t* __restrict__ rowptr,
                                                           int32_t* __restrict__ bounds) {
  const int i = blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (i >= NRB * (MDE_RING_NCW + 1)) return;
  const int rb = i / (MDE_RING_NCW + 1), w = i % (MDE_RING_NCW + 1);
  const int r0 = rb * R, r1 = min(nloc, r0 + R);
  const int64_t lo = rowptr[r0], hi = rowptr[r1];
  const int64_t target = lo + ((hi - lo) * w) / MDE_RING_NCW;
  int a = r0, b = r1;  // smallest r in [r0, r1] with rowptr[r] >= target
  while (a < b) {
    const int mid = (a + b) >> 1;
    if (rowptr[mid] >= target) b = mid; else a = mid + 1;
  }
  bounds[i] = (w == MDE_RING_NCW) ? r1 : a;
}

// key = ((rb * Q + column group) * NCW + wave) << JB | chunk; CSR order (row, edge id) is kept
// inside equal keys by the stable sort
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_keys(int nrows, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ nbr,
                                                         const int32_t* __restrict__ bounds, int R, int Q,
                                                         int NC, int C, int JB, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals, int32_t* __restrict__ hrow) {
  constexpr int G = 16;
  const int lig = threadIdx.x & (G - 1);
  const int group = (blockIdx.x * MDE_BLOCK + threadIdx.x) / G;
  const int ngroups = (gridDim.x * MDE_BLOCK) / G;
  for (int r = group; r < nrows; r += ngroups) {
    const int beg = rowptr[r], end = rowptr[r + 1];
    const int rb = r / R;
    const int32_t* bd = bounds + (size_t)rb * (MDE_RING_NCW + 1);
    int w = 0;
    for (int t = 1; t < MDE_RING_NCW; ++t) w += (bd[t] <= r);
    for (int q = beg + lig; q < end; q += G) {
      const uint32_t j = (uint32_t)(nbr[q] / C);
      const uint32_t g = (uint32_t)(((uint64_t)j * (uint64_t)Q) / (uint64_t)NC);
      keys[q] = ((((uint32_t)rb * (uint32_t)Q + g) * MDE_RING_NCW + (uint32_t)w) << JB) | j;
      vals[q] = (uint32_t)q;
      hrow[q] = r;
    }
  }
}

// seg[t] = first sorted position whose stream id (key >> JB) >= t, t = 0..nseg
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_seg(int64_t H, uint32_t nseg, int JB,
                                                        const uint32_t* __restrict__ keys,
                                                        int32_t* __restrict__ seg) {
  for (int64_t h = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; h <= H;
       h += (int64_t)gridDim.x * MDE_BLOCK) {
    const int64_t prev = (h == 0) ? -1 : (int64_t)(keys[h - 1] >> JB);
    int64_t cur = (h == H) ? (int64_t)nseg : (int64_t)(keys[h] >> JB);
    if (cur > (int64_t)nseg) cur = nseg;
    for (int64_t t = prev + 1; t <= cur; ++t) seg[t] = (int32_t)h;
  }
}

// Cut each (block, group, wave) stream into wave iterations of <= 64 consecutive entries whose
// chunks span at most SPAN (the ring holds them all at once).  FILL = false counts, FILL = true
// writes the source range of every iteration.
template <bool FILL>
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_iters(int nseg, const int32_t* __restrict__ seg,
                                                          const uint32_t* __restrict__ keys, uint32_t JM, int SPAN,
                                                          int32_t* __restrict__ iters,
                                                          const int32_t* __restrict__ iter_base,
                                                          int32_t* __restrict__ it_src, int32_t* __restrict__ it_cnt) {
  const int i = blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (i >= nseg) return;
  int pos = seg[i];
  const int end = seg[i + 1];
  int out = FILL ? iter_base[i] : 0;
  while (pos < end) {
    int take = min(64, end - pos);
    const uint32_t lim = (keys[pos] & JM) + (uint32_t)SPAN;
    if ((keys[pos + take - 1] & JM) > lim) {
      int a = 1, b = take - 1;  // largest t in [1, take) with chunk(pos + t - 1) <= lim
      while (a < b) {
        const int mid = (a + b + 1) >> 1;
        if ((keys[pos + mid - 1] & JM) <= lim) a = mid; else b = mid - 1;
      }
      take = a;
    }
    if (FILL) {
      it_src[out] = pos;
      it_cnt[out] = take;
    }
    ++out;
    pos += take;
  }
  if (!FILL) iters[i] = out;
}

// One wave per iteration: sort its <= 64 entries by row (stable, so a row's entries keep their
// chunk-major order), pad to 64 with dummies in the highest lanes, write packed words, edge ids
// and the header.
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_pack(int64_t nit, const int32_t* __restrict__ it_src,
                                                         const int32_t* __restrict__ it_cnt,
                                                         const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals,
                                                         const int32_t* __restrict__ hrow,
                                                         const int32_t* __restrict__ nbr,
                                                         const int32_t* __restrict__ eid, int R, int Q, int C, int S,
                                                         int JB, int d, uint32_t* __restrict__ packed,
                                                         int32_t* __restrict__ peid, uint32_t* __restrict__ hdr) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = ((int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * MDE_BLOCK) >> 6;
  const uint32_t JM = (1u << JB) - 1u;
  for (int64_t it = w0; it < nit; it += nw) {
    const int src = __builtin_amdgcn_readfirstlane(it_src[it]);
    const int cnt = __builtin_amdgcn_readfirstlane(it_cnt[it]);
    const bool act = lane < cnt;
    uint32_t key = 0, q = 0;
    int rl = 0x7fffffff;
    if (act) {
      key = keys[src + lane];
      q = vals[src + lane];
      const int rb = (int)((key >> JB) / MDE_RING_NCW) / Q;
      rl = hrow[q] - rb * R;
    }
    const uint32_t j = key & JM;
    int rank = 0, same = 0;
    for (int t = 0; t < 64; ++t) {
      const int rt = __builtin_amdgcn_readlane(rl, t);
      rank += (rt < rl) || (rt == rl && t < lane);
      same += (rt == rl);
    }
    const int run = mde_wave_max(act ? same : 0);
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);          // entries are chunk-major
    const uint32_t need = (uint32_t)__builtin_amdgcn_readlane((int)j, cnt - 1);
    const size_t base = (size_t)it * 64;
    if (act) {
      const uint32_t col = (uint32_t)nbr[q];
      const uint32_t ring = ((j % (uint32_t)S) * (uint32_t)C + (col - j * (uint32_t)C)) * 4u * (uint32_t)d;
      packed[base + rank] = (((uint32_t)rl * 4u * (uint32_t)d) << 17) | ring;
      peid[base + rank] = eid[q];
    } else {
      // padding: the dummy row slot, a resident column (first of chunk m)
      packed[base + lane] = (((uint32_t)R * 4u * (uint32_t)d) << 17) | ((m % (uint32_t)S) * (uint32_t)C * 4u * (uint32_t)d);
      peid[base + lane] = -1;
    }
    if (lane == 0) hdr[it] = MDE_RING_HDR(m, need - m, min(run - 1, 63), cnt < 64);
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

struct RingSizes {
  int R, NRB, Q, C, NC, S, JB;
};

// Decide the block height and the column groups for dimension d; false when the layout is not
// worthwhile (the caller keeps the CSR kernel).
static bool choose_sizes(const mde_plan* plan, int d, RingSizes* z) {
  const int64_t nloc = plan->row_hi - plan->row_lo;
  if (d < 1 || d > 4 || nloc <= 0 || plan->H <= 0) return false;
  const int mode = panel_mode();
  if (mode == 0) return false;
  // rows: about one block per CU (256), multiple of 64, and x_v + the dummy slot must fit below
  // the control words.  A rank that owns n/N rows keeps the block height of the full plan (the
  // staging volume is NRB * table bytes) and fills the CUs with Q column groups per row block.
  int64_t pr = (plan->n + 255) / 256;
  if (pr > nloc) pr = nloc;
  pr = ((pr + 63) / 64) * 64;
  const int64_t pr_max = ((MDE_RING_CTRL_OFF - 4 * d) / (4 * d)) / 64 * 64;
  if (pr > pr_max) pr = pr_max;
  const int C = ring_chunk_cols(d);
  const int64_t nc = (plan->n + C - 1) / C;
  const int64_t nrb = (nloc + pr - 1) / pr;
  if (nc > 65535 || nc < 2) return false;
  int Q = 1;
  if (nrb <= 128) {
    Q = (int)(256 / nrb);  // one resident round of workgroups
    if (Q > nc / 16) Q = (int)(nc / 16);
    if (Q > 16) Q = 16;
    if (Q < 1) Q = 1;
  }
  if (nrb * Q > MDE_MAX_PARTIALS) return false;
  const int jb = bits_for_u64((uint64_t)nc - 1);
  if (bits_for_u64((uint64_t)(nrb * Q * MDE_RING_NCW)) + jb > 32) return false;
  if (mode != 1) {
    // auto: only when the table overflows L2 and a 64-entry iteration fits the ring window
    if ((int64_t)plan->n * d * 4 < (6 << 20)) return false;
    if ((double)plan->H / ((double)nrb * MDE_RING_NCW * (double)nc) < 64.0 / (0.6 * ring_max_span(d))) return false;
  }
  z->R = (int)pr;
  z->NRB = (int)nrb;
  z->Q = Q;
  z->C = C;
  z->NC = (int)nc;
  z->S = ring_slots(d);
  z->JB = jb;
  return true;
}

static void ring_free(mde_ring_layout& L) {
  if (L.packed) (void)hipFree(L.packed);
  if (L.eid) (void)hipFree(L.eid);
  if (L.hdr) (void)hipFree(L.hdr);
  if (L.wave_iter) (void)hipFree(L.wave_iter);
  if (L.partial) (void)hipFree(L.partial);
  L = mde_ring_layout();
}
void mde_ring_release(mde_plan* plan) { ring_free(plan->ring); }

static int build_ring(mde_plan* plan, int d, hipStream_t st) {
  RingSizes z;
  if (!choose_sizes(plan, d, &z)) return 0;
  const int64_t nloc = plan->row_hi - plan->row_lo;
  const int64_t H = plan->H;
  const int nseg = z.NRB * z.Q * MDE_RING_NCW;
  const uint32_t JM = (1u << z.JB) - 1u;
  uint32_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr, *packed = nullptr, *hdr = nullptr;
  int32_t *hrow = nullptr, *bounds = nullptr, *seg = nullptr, *iters = nullptr, *iter_base = nullptr;
  int32_t *it_src = nullptr, *it_cnt = nullptr, *peid = nullptr;
  float* partial = nullptr;
  void* tmp = nullptr;
  hipError_t e = hipSuccess;
  auto release = [&](bool all) {
    void* scratch[] = {keys, vals, keys2, vals2, hrow, bounds, seg, iters, it_src, it_cnt, tmp};
    for (void* p : scratch)
      if (p) (void)hipFree(p);
    if (all) {
      void* outs[] = {packed, hdr, peid, iter_base, partial};
      for (void* p : outs)
        if (p) (void)hipFree(p);
    }
  };
  auto fail = [&](hipError_t err, const char* what) {
    release(true);
    return mde_hip_fail(err, what, __FILE__, __LINE__);
  };
#define RB(call)                                 \
  do {                                           \
    e = (call);                                  \
    if (e != hipSuccess) return fail(e, #call);  \
  } while (0)
  const size_t hb = (size_t)H * sizeof(uint32_t);
  RB(hipMalloc(&keys, hb));
  RB(hipMalloc(&vals, hb));
  RB(hipMalloc(&keys2, hb));
  RB(hipMalloc(&vals2, hb));
  RB(hipMalloc(&hrow, hb));
  RB(hipMalloc(&bounds, (size_t)z.NRB * (MDE_RING_NCW + 1) * sizeof(int32_t)));
  RB(hipMalloc(&seg, ((size_t)nseg + 1) * sizeof(int32_t)));
  RB(hipMalloc(&iters, ((size_t)nseg + 1) * sizeof(int32_t)));
  RB(hipMalloc(&iter_base, ((size_t)nseg + 1) * sizeof(int32_t)));
  hipLaunchKernelGGL(k_ring_bounds, dim3((z.NRB * (MDE_RING_NCW + 1) + MDE_BLOCK - 1) / MDE_BLOCK), dim3(MDE_BLOCK), 0,
                     st, (int)nloc, z.R, z.NRB, plan->rowptr, bounds);
  RB(hipGetLastError());
  hipLaunchKernelGGL(k_ring_keys, dim3(mde_grid(nloc * 16, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, (int)nloc,
                     plan->rowptr, plan->nbr, bounds, z.R, z.Q, z.NC, z.C, z.JB, keys, vals, hrow);
  RB(hipGetLastError());
  size_t tmp_bytes = 0, scan_bytes = 0;
  const int end_bit = std::min(32, bits_for_u64((uint64_t)nseg) + z.JB);
  RB(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys, keys2, vals, vals2, (int)H, 0, end_bit, st));
  RB(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, iters, iter_base, nseg + 1, st));
  if (scan_bytes > tmp_bytes) tmp_bytes = scan_bytes;
  RB(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  RB(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys, keys2, vals, vals2, (int)H, 0, end_bit, st));
  hipLaunchKernelGGL(k_ring_seg, dim3(mde_grid(H + 1, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, H, (uint32_t)nseg,
                     z.JB, keys2, seg);
  RB(hipGetLastError());
  RB(hipMemsetAsync(iters, 0, ((size_t)nseg + 1) * sizeof(int32_t), st));
  hipLaunchKernelGGL(k_ring_iters<false>, dim3((nseg + MDE_BLOCK - 1) / MDE_BLOCK), dim3(MDE_BLOCK), 0, st, nseg, seg,
                     keys2, JM, ring_max_span(d), iters, nullptr, nullptr, nullptr);
  RB(hipGetLastError());
  RB(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, iters, iter_base, nseg + 1, st));
  int32_t total_iters = 0;
  RB(hipMemcpyAsync(&total_iters, iter_base + nseg, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  RB(hipStreamSynchronize(st));
  const int64_t Hp = (int64_t)total_iters * 64;  // padded half-edge count
  if (total_iters <= 0 || Hp >= ((int64_t)1 << 31) - 64) {
    release(true);
    return 0;  // too large for 32-bit positions: the caller keeps the CSR layout
  }
  RB(hipMalloc(&it_src, (size_t)total_iters * sizeof(int32_t)));
  RB(hipMalloc(&it_cnt, (size_t)total_iters * sizeof(int32_t)));
  RB(hipMalloc(&packed, (size_t)Hp * sizeof(uint32_t)));
  RB(hipMalloc(&peid, (size_t)Hp * sizeof(int32_t)));
  RB(hipMalloc(&hdr, (size_t)total_iters * sizeof(uint32_t)));
  if (z.Q > 1) RB(hipMalloc(&partial, sizeof(float) * (size_t)z.Q * (size_t)nloc * (size_t)d));
  hipLaunchKernelGGL(k_ring_iters<true>, dim3((nseg + MDE_BLOCK - 1) / MDE_BLOCK), dim3(MDE_BLOCK), 0, st, nseg, seg,
                     keys2, JM, ring_max_span(d), nullptr, iter_base, it_src, it_cnt);
  RB(hipGetLastError());
  hipLaunchKernelGGL(k_ring_pack, dim3(mde_grid((int64_t)total_iters * 64, MDE_BLOCK, 8192)), dim3(MDE_BLOCK), 0, st,
                     (int64_t)total_iters, it_src, it_cnt, keys2, vals2, hrow, plan->nbr, plan->eid, z.R, z.Q, z.C,
                     z.S, z.JB, d, packed, peid, hdr);
  RB(hipGetLastError());
  RB(hipStreamSynchronize(st));
#undef RB
  release(false);
  mde_ring_layout& L = plan->ring;
  ring_free(L);
  L.d = d;
  L.rows_per_block = z.R;
  L.n_row_blocks = z.NRB;
  L.col_groups = z.Q;
  L.chunk_cols = z.C;
  L.n_chunks = z.NC;
  L.n_iters = total_iters;
  L.H = Hp;
  L.packed = packed;
  L.eid = peid;
  L.hdr = hdr;
  L.wave_iter = iter_base;
  L.partial = partial;
  return 1;
}

// layout the fused kernel will use for dimension d: 0 = CSR, 1 = LDS ring (built on first
// request).  Negative: error.
extern "C" int mde_plan_layout(mde_plan* plan, int32_t d, void* stream) {
  if (!plan || d <= 0) return MDE_E_INVALID;
  if (plan->ring.packed && plan->ring.d == d) return 1;
  RingSizes z;
  if (!choose_sizes(plan, d, &z)) return 0;
  return build_ring(plan, d, mde_stream(stream));
}

__global__ __launch_bounds__(MDE_BLOCK) void k_expand_ring(int64_t H, const int32_t* __restrict__ eid,
                                                           const float* __restrict__ in,
                                                           float* __restrict__ out) {
  for (int64_t q = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; q < H;
       q += (int64_t)gridDim.x * MDE_BLOCK)
    out[q] = eid[q] >= 0 ? in[eid[q]] : 1.0f;  // padding entries carry a harmless parameter
}

// number of entries of a per-half-edge parameter array in the given layout (layout 1: the padded
// stream + MDE_RING_CB_VALUES spare entries, where a codebook stream keeps its value table)
extern "C" int64_t mde_plan_layout_half_edges(const mde_plan* plan, int32_t layout) {
  if (!plan) return 0;
  return (layout == 1 && plan->ring.packed) ? plan->ring.H + MDE_RING_CB_VALUES : plan->H;
}

// ---------------------------------------------------------------- parameter codebooks
// Neighbour-graph problems carry very few distinct per-edge parameters (k-NN weights 1 / 2, -1 for
// repulsive pairs).  At d = 2 the packed word's ring offset is a multiple of 8, so its 3 low bits
// can hold an index into a table of <= 8 values: the kernel then streams 4 bytes per half-edge
// instead of 8 (packed word + fp32 parameter) and looks the parameter up in LDS.
#define MDE_CB_EMPTY 0xFFFFFFFFu  // (a NaN pattern: NaN parameters simply disable the codebook)

// distinct bit patterns of in[0..p): inserted into table[0..8) with compare-and-swap; *overflow is
// set when a 9th value (or the EMPTY pattern) shows up
__global__ __launch_bounds__(MDE_BLOCK) void k_codebook_scan(int64_t p, const float* __restrict__ in,
                                                             unsigned int* __restrict__ table,
                                                             int* __restrict__ overflow) {
  __shared__ unsigned int stb[MDE_RING_CB_VALUES];
  if (threadIdx.x < MDE_RING_CB_VALUES) stb[threadIdx.x] = MDE_CB_EMPTY;
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
    for (int s = 0; s < MDE_RING_CB_VALUES; ++s) known |= (stb[s] == v);
    if (known) continue;
    if (v == MDE_CB_EMPTY || *reinterpret_cast<volatile int*>(overflow)) {
      *overflow = 1;
      return;
    }
    bool placed = false;
    for (int s = 0; s < MDE_RING_CB_VALUES && !placed; ++s) {
      const unsigned int old = atomicCAS(&table[s], MDE_CB_EMPTY, v);
      placed = (old == MDE_CB_EMPTY || old == v);
    }
    if (!placed) {
      *overflow = 1;
      return;
    }
    for (int s = 0; s < MDE_RING_CB_VALUES; ++s) {  // remember it block-wide
      const unsigned int old = atomicCAS(&stb[s], MDE_CB_EMPTY, v);
      if (old == MDE_CB_EMPTY || old == v) break;
    }
  }
}

// out[q] = packed[q] | index of in[eid[q]] in table (padding entries keep index 0)
__global__ __launch_bounds__(MDE_BLOCK) void k_codebook_pack(int64_t H, const uint32_t* __restrict__ packed,
                                                             const int32_t* __restrict__ eid,
                                                             const float* __restrict__ in,
                                                             const unsigned int* __restrict__ table,
                                                             uint32_t* __restrict__ out) {
  unsigned int tb[MDE_RING_CB_VALUES];
#pragma unroll
  for (int s = 0; s < MDE_RING_CB_VALUES; ++s) tb[s] = table[s];
  for (int64_t q = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; q < H;
       q += (int64_t)gridDim.x * MDE_BLOCK) {
    uint32_t w = packed[q];
    if (eid[q] >= 0) {
      const unsigned int v = __float_as_uint(in[eid[q]]);
      uint32_t idx = 0;
#pragma unroll
      for (int s = 1; s < MDE_RING_CB_VALUES; ++s) idx = (tb[s] == v) ? (uint32_t)s : idx;
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
  const mde_ring_layout& L = plan->ring;
  if (!L.packed || L.d != 2 || L.H == 0 || plan->p == 0) return MDE_OK;
  const char* e = getenv("MDE_CODEBOOK");
  if (e && atoi(e) == 0) return MDE_OK;
  hipStream_t st = mde_stream(stream);
  unsigned int* table = reinterpret_cast<unsigned int*>(out_half) + L.H;  // the spare entries
  // (the overflow flag lives in the plan's reduction scratch: no allocation per call)
  int* overflow = reinterpret_cast<int*>(plan->partials + MDE_MAX_PARTIALS + 1);
  hipError_t err = hipMemsetAsync(overflow, 0, sizeof(int), st);
  if (err == hipSuccess) err = hipMemsetAsync(table, 0xFF, MDE_RING_CB_VALUES * sizeof(unsigned int), st);
  unsigned int host_tb[MDE_RING_CB_VALUES];
  int host_overflow = 0;
  if (err == hipSuccess) {
    hipLaunchKernelGGL(k_codebook_scan, dim3(mde_grid(plan->p, MDE_BLOCK, 2048)), dim3(MDE_BLOCK), 0, st, plan->p,
                       in_edge, table, overflow);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipMemcpyAsync(host_tb, table, sizeof(host_tb), hipMemcpyDeviceToHost, st);
  if (err == hipSuccess) err = hipMemcpyAsync(&host_overflow, overflow, sizeof(int), hipMemcpyDeviceToHost, st);
  if (err == hipSuccess) err = hipStreamSynchronize(st);
  if (err != hipSuccess) return mde_hip_fail(err, "parameter codebook scan", __FILE__, __LINE__);
  if (host_overflow) return MDE_OK;
  // canonical order (the insertion order above depends on scheduling): ascending bit patterns
  int nv = 0;
  unsigned int vals[MDE_RING_CB_VALUES];
  for (int s = 0; s < MDE_RING_CB_VALUES; ++s)
    if (host_tb[s] != MDE_CB_EMPTY) vals[nv++] = host_tb[s];
  if (nv == 0) return MDE_OK;
  for (int a = 1; a < nv; ++a)
    for (int b = a; b > 0 && vals[b - 1] > vals[b]; --b) {
      const unsigned int t = vals[b];
      vals[b] = vals[b - 1];
      vals[b - 1] = t;
    }
  for (int s = nv; s < MDE_RING_CB_VALUES; ++s) vals[s] = MDE_CB_EMPTY;
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
  if (layout != 1 || !plan->ring.eid) {
    mde_set_error("mde_plan_expand_layout: the LDS-ring layout has not been built");
    return MDE_E_INVALID;
  }
  if (plan->ring.H == 0) return MDE_OK;
  hipLaunchKernelGGL(k_expand_ring, dim3(mde_grid(plan->ring.H, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0,
                     mde_stream(stream), plan->ring.H, plan->ring.eid, in_edge, out_half);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

// ---------------------------------------------------------------- the kernel
template <int D>
struct RingVec;
template <>
struct RingVec<1> {
  typedef float T;
};
template <>
struct RingVec<2> {
  typedef float2 T;
};
template <>
struct RingVec<4> {
  typedef float4 T;
};
// D floats at an LDS byte address (aligned to the vector size for D = 1, 2, 4)
template <int D>
__device__ __forceinline__ void ring_ld(const char* p, float (&v)[D]) {
  if constexpr (D == 3) {
    const float* q = reinterpret_cast<const float*>(p);
    v[0] = q[0];
    v[1] = q[1];
    v[2] = q[2];
  } else {
    const typename RingVec<D>::T t = *reinterpret_cast<const typename RingVec<D>::T*>(p);
    const float* q = reinterpret_cast<const float*>(&t);
#pragma unroll
    for (int c = 0; c < D; ++c) v[c] = q[c];
  }
}
template <int D>
__device__ __forceinline__ void ring_st(char* p, const float (&v)[D]) {
  if constexpr (D == 3) {
    float* q = reinterpret_cast<float*>(p);
    q[0] = v[0];
    q[1] = v[1];
    q[2] = v[2];
  } else {
    typename RingVec<D>::T t;
    float* q = reinterpret_cast<float*>(&t);
#pragma unroll
    for (int c = 0; c < D; ++c) q[c] = v[c];
    *reinterpret_cast<typename RingVec<D>::T*>(p) = t;
  }
}

// one 1 KiB LDS-DMA piece: lane l copies the 16 bytes at gsrc to LDS byte address lds_dst + 16 l
// (M0 carries the wave-uniform LDS base; hipcc neither counts the load nor preserves M0 around
// the statement, so M0 is saved and restored inside it and the waits are explicit)
__device__ __forceinline__ void ring_dma16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// control words: explicit LDS instructions on absolute addresses (a `volatile` generic pointer
// would turn into flat loads / stores and drag vmcnt(0) waits into the stream pipeline)
__device__ __forceinline__ void ring_ctrl_store(uint32_t addr, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ int ring_ctrl_load(uint32_t addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return v;
}

// CB: the first parameter comes from a codebook -- `packed` is the stream with value indices in its
// 3 low bits (mde_plan_expand_codebook), a0 the 8-entry value table; no parameter stream is read.
template <int D, class Fn, bool HAS_GRAD, bool CB>
__global__ __launch_bounds__(MDE_RING_BS) void k_fused_ring(
    int nloc, int row_lo, int n, int R, int Q, int NC, const int32_t* __restrict__ wave_iter,
    const uint32_t* __restrict__ hdr, const uint32_t* __restrict__ packed, const float* __restrict__ a0,
    const float* __restrict__ a1, int a0_scalar, int a1_scalar, const float* __restrict__ X,
    float* __restrict__ grad, float* __restrict__ partial, double* __restrict__ loss_partials, Fn fn,
    float inv_p, float grad_scale, float* __restrict__ loss_out, double loss_scale, int dbg) {
  constexpr int BS = MDE_RING_BS, NCW = MDE_RING_NCW, NPROD = MDE_RING_NPROD, PF = MDE_RING_PF;
  constexpr int GR_OFF = MDE_RING_GR_OFF, RING_OFF = MDE_RING_OFF;
  constexpr int C = ring_chunk_cols(D), CBYTES = ring_chunk_bytes(D), S = ring_slots(D), PIECES = CBYTES / 1024;
  // statically sized: the LDS addresses unpacked from the stream are absolute
  __shared__ __attribute__((aligned(16))) char L[MDE_RING_OFF + MDE_RING_BYTES];
  float* XR = reinterpret_cast<float*>(L);           // x_v of the block's rows
  float* GR = reinterpret_cast<float*>(L + GR_OFF);  // gradient accumulators (same slots)
  int* prog = reinterpret_cast<int*>(L + MDE_RING_CTRL_PROG);  // (prologue only; polled with ring_ctrl_*)
  int* F = reinterpret_cast<int*>(L + MDE_RING_CTRL_F);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: keep it scalar
  // block b = rb * Q + qg: row block rb, column group qg owns chunks [j_lo, j_hi)
  const int rb = blockIdx.x / Q, qg = blockIdx.x % Q;
  const int j_lo = (int)(((int64_t)qg * NC + Q - 1) / Q), j_hi = (int)(((int64_t)(qg + 1) * NC + Q - 1) / Q);
  const int r0 = rb * R;
  const int nr = min(R, nloc - r0);
  const uint32_t dummy_row = (uint32_t)R * 4u * (uint32_t)D;
  const float a0s = (a0_scalar && !CB) ? a0[0] : 1.0f;
  const float a1s = (a1 && a1_scalar) ? a1[0] : 0.0f;
  const bool a1_arr = a1 && !a1_scalar;
  // ---- prologue: accumulators, x_v, control words
  {
    float4* z = reinterpret_cast<float4*>(L + GR_OFF);
    for (int i = tid; i < (MDE_RING_OFF - GR_OFF) / 16; i += BS) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* Xrow = X + (size_t)(row_lo + r0) * D;
    for (int i = tid; i < nr * D; i += BS) XR[i] = Xrow[i];
    if (tid < D) XR[R * D + tid] = 0.0f;  // the dummy row
    if (tid < 16) prog[tid] = (tid < NCW) ? j_lo : MDE_RING_DONE;
    if (tid >= 16 && tid < 16 + NPROD) F[tid - 16] = j_lo + (tid - 16);
    if (CB && tid >= 32 && tid < 32 + MDE_RING_CB_VALUES)
      reinterpret_cast<float*>(L + MDE_RING_CTRL_CB)[tid - 32] = a0[tid - 32];
  }
  __syncthreads();
  // No compiler-counted load may be pending past this point: the producers' LDS-DMA pieces are
  // invisible to hipcc's vmcnt bookkeeping, and a wait it inserts for one of ITS loads (or for
  // re-using such a load's destination register) would drain the pieces in flight with it.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  float loss = 0.0f;

  if (wave >= NCW) {
    // ---------------- producer p: chunks j_lo + p, j_lo + p + NPROD, ...
    const int p = wave - NCW;
    const char* Xb = reinterpret_cast<const char*>(X);
    const size_t nbytes = (size_t)n * D * 4;
    const size_t last16 = nbytes - 16;
    int minprog = j_lo, infl = 0;
    for (int j = j_lo + p; j < j_hi; j += NPROD) {
      // slot j % S still holds chunk j - S: wait until every consumer is past it
      while (j - S >= minprog && !(dbg & 8)) {
        int v = ring_ctrl_load(MDE_RING_CTRL_PROG + 4u * (uint32_t)(lane < NCW ? lane : 0));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
        minprog = __builtin_amdgcn_readfirstlane(v);
        if (j - S >= minprog) __builtin_amdgcn_s_sleep(2);
      }
      const uint32_t dst = (uint32_t)RING_OFF + (uint32_t)(j % S) * (uint32_t)CBYTES;
      const size_t off0 = (size_t)j * CBYTES + (size_t)lane * 16;
      if (!(dbg & 2))
#pragma unroll
      for (int k = 0; k < PIECES; ++k) {
        const size_t off = off0 + (size_t)k * 1024;
        ring_dma16(Xb + (off < last16 ? off : last16), dst + (uint32_t)k * 1024u);
      }
      if (++infl == MDE_RING_DEPTH) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES * (MDE_RING_DEPTH - 1)) : "memory");
        // everything of mine before this chunk has landed
        ring_ctrl_store(MDE_RING_CTRL_F + 4u * (uint32_t)p, j - (MDE_RING_DEPTH - 1) * NPROD + NPROD);
        --infl;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the table's last chunk: lanes whose 16 bytes cross the end loaded a clamped address; when
    // the table is not a multiple of 16 bytes the final dwords are put in place by hand
    if ((nbytes & 15) && (NC - 1) >= j_lo && (NC - 1) < j_hi && ((NC - 1 - j_lo) % NPROD) == p) {
      const size_t tail0 = nbytes & ~(size_t)15;
      const int nt = (int)((nbytes - tail0) >> 2);
      if (lane < nt) {
        const size_t off = tail0 + (size_t)lane * 4 - (size_t)(NC - 1) * CBYTES;
        *reinterpret_cast<float*>(L + RING_OFF + ((NC - 1) % S) * CBYTES + off) =
            *reinterpret_cast<const float*>(Xb + tail0 + (size_t)lane * 4);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    ring_ctrl_store(MDE_RING_CTRL_F + 4u * (uint32_t)p, MDE_RING_DONE);
  } else {
    // ---------------- consumer: my contiguous stream of wave iterations
    const int ib = __builtin_amdgcn_readfirstlane(wave_iter[blockIdx.x * NCW + wave]);
    const int niter = __builtin_amdgcn_readfirstlane(wave_iter[blockIdx.x * NCW + wave + 1]) - ib;
    if (niter > 0 && !(dbg & 64)) {
      const uint32_t* sp = packed + (size_t)ib * 64 + lane;
      // (the header is loaded like the packed words, as a vector load every lane issues for the
      // same address: a scalar load would tie each iteration to an SMEM round trip through the
      // lgkmcnt(0) its out-of-order return forces)
      int zl;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zl));
      const uint32_t* hp = hdr + ib + zl;
      const bool a0_arr = !a0_scalar && !CB;
      const float* ap = a0_arr ? a0 + (size_t)ib * 64 + lane : a0;
      const int astride = a0_arr ? 64 : 0;
      const int last = niter - 1;
      uint32_t pk[PF] = {}, hd[PF] = {};
      float wv[PF] = {};
      // All stream loads are issued from ONE place (the refill after a slot is consumed; the first
      // trip of the loop below only fills), unconditional and clamped, never predicated: on every
      // path exactly PF - 1 younger loads are in flight when a slot is consumed, so the
      // compiler's vmcnt counts are exact and nothing waits for a load just issued.
      auto load_slot = [&](int k, int it) __attribute__((always_inline)) {
        int itc = min(it, last);
        if (dbg & 16) itc &= 31;  // (probe: an L1/L2-resident stream)
        pk[k] = (dbg & 32) ? __builtin_nontemporal_load(sp + (size_t)itc * 64) : sp[(size_t)itc * 64];
        hd[k] = hp[itc];
        if (!CB) wv[k] = ap[(size_t)itc * astride];
      };
      int ready = j_lo, published = j_lo;

      auto process = [&](uint32_t w, float p0, float p1, int rounds, bool padded) __attribute__((always_inline)) {
        const uint32_t rowaddr = w >> 17, coladdr = w & (CB ? 0x1fff8u : 0x1ffffu);
        if (CB) p0 = *reinterpret_cast<const float*>(L + MDE_RING_CTRL_CB + ((w & 7u) << 2));
        float xr[D], xc[D], v[D], ss = 0.0f;
        ring_ld<D>(L + rowaddr, xr);
        ring_ld<D>(L + RING_OFF + coladdr, xc);
#pragma unroll
        for (int c = 0; c < D; ++c) {
          v[c] = xr[c] - xc[c];
          ss = fmaf(v[c], v[c], ss);
        }
        float f, gd;
        fn.eval(ss, p0, p1, f, gd);
        const float g = mde_fix_g(gd * inv_p);
        if (padded)
          loss += (rowaddr != dummy_row) ? f : 0.0f;
        else
          loss += f;
        if (!HAS_GRAD) return;
#pragma unroll
        for (int c = 0; c < D; ++c) v[c] *= g;
        // The 64 entries are sorted by row: equal rows are adjacent lanes.  Round r adds the
        // ORIGINAL contribution of lane i - r when it has the same row (keys / values shifted one
        // lane per round with DPP wave_shr:1); the last lane of each run then holds the run's sum
        // and performs the single read-add-write of the row.
        const int key = (int)rowaddr;
        bool tail = true;
        if (rounds > 0) {
          int kc = key;
          float sv[D];
#pragma unroll
          for (int c = 0; c < D; ++c) sv[c] = v[c];
#pragma nounroll
          for (int r = 0; r < rounds; ++r) {
            kc = __builtin_amdgcn_update_dpp(-1, kc, 0x138, 0xf, 0xf, false);
            const bool same = (kc == key);
#pragma unroll
            for (int c = 0; c < D; ++c) {
              sv[c] = __int_as_float(
                  __builtin_amdgcn_update_dpp(0, __float_as_int(sv[c]), 0x138, 0xf, 0xf, false));
              v[c] += same ? sv[c] : 0.0f;
            }
          }
          const int knext = __builtin_amdgcn_update_dpp(-1, key, 0x130, 0xf, 0xf, false);  // wave_shl:1
          tail = knext != key;
        }
        if (tail) {
          float acc[D];
          ring_ld<D>(L + GR_OFF + rowaddr, acc);
#pragma unroll
          for (int c = 0; c < D; ++c) acc[c] += v[c];
          ring_st<D>(L + GR_OFF + rowaddr, acc);
        }
      };

      for (int base = -PF; base < niter; base += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
          const int it = base + k;
          if (it >= 0 && it < niter) {
            const uint32_t h = (uint32_t)__builtin_amdgcn_readfirstlane((int)hd[k]);
            const int m = (int)(h & 0xffffu), need = m + (int)((h >> 16) & 15u);
            if (m != published) {
              // (every earlier read of chunks < m has been issued, and the LDS executes in order)
              ring_ctrl_store(MDE_RING_CTRL_PROG + 4u * (uint32_t)wave, m);
              published = m;
            }
            if (need >= ready && !(dbg & 1)) {
              for (;;) {
                const int fl = ring_ctrl_load(MDE_RING_CTRL_F + 4u * (uint32_t)(lane & 1));
                ready = min(__builtin_amdgcn_readlane(fl, 0), __builtin_amdgcn_readlane(fl, 1));
                if (need < ready) break;
                __builtin_amdgcn_s_sleep(1);
              }
              asm volatile("" ::: "memory");
            }
            const float p1 = a1_arr ? a1[(size_t)(ib + it) * 64 + lane] : a1s;
            if (!(dbg & 4)) process(pk[k], (a0_scalar || CB) ? a0s : wv[k], p1, (int)((h >> 20) & 63u), (h >> 26) & 1u);
            else loss += __uint_as_float(pk[k]) * 0.0f;
          }
          load_slot(k, it + PF);
        }
      }
    }
    ring_ctrl_store(MDE_RING_CTRL_PROG + 4u * (uint32_t)wave, MDE_RING_DONE);
    // (the two roles are laid out one after the other: leave no counted load pending here, or
    // hipcc carries the stream prefetches into the producer code as waits -- see above)
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  __syncthreads();
  if (HAS_GRAD) {
    // Q == 1: the rows are final.  Q > 1: unscaled per-group partials, summed by k_ring_combine
    float* grow = (Q == 1) ? grad + (size_t)(row_lo + r0) * D : partial + ((size_t)qg * nloc + r0) * D;
    const float sc = (Q == 1) ? grad_scale : 1.0f;
    for (int i = tid; i < nr * D; i += BS) grow[i] = GR[i] * sc;
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
    for (int i = 0; i < NCW; ++i) s += red[i];
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
      for (int i = 0; i < BS / 64; ++i) s += red[i];
      *loss_out = (float)(s * loss_scale);
      *ticket = 0u;  // ready for the next launch (stream order)
    }
  }
}

// grad[row] = scale * sum_q partial[q][row]  (fixed order)
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_combine(int64_t nlocD, int Q, const float* __restrict__ partial,
                                                            float scale, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < nlocD;
       i += (int64_t)gridDim.x * MDE_BLOCK) {
    float s = 0.0f;
    for (int q = 0; q < Q; ++q) s += partial[(size_t)q * nlocD + i];
    out[i] = s * scale;
  }
}

struct RingArgs {
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
static int launch_ring(const RingArgs& A, const Fn& fn, int* nblocks) {
  const mde_ring_layout& L = A.plan->ring;
  const bool cb = A.a0_scalar == 2;
  if (cb && D != 2) {
    mde_set_error("codebook parameter streams exist for d = 2 only");
    return MDE_E_INVALID;
  }
  if (reinterpret_cast<uintptr_t>(A.X) & 15) {
    mde_set_error("the LDS-ring kernel needs a 16-byte aligned embedding matrix");
    return MDE_E_INVALID;
  }
  auto kern = A.grad ? k_fused_ring<D, Fn, true, false> : k_fused_ring<D, Fn, false, false>;
  if constexpr (D == 2) {
    if (cb) kern = A.grad ? k_fused_ring<D, Fn, true, true> : k_fused_ring<D, Fn, false, true>;
  }
  // codebook form: a0 = [H packed words | 8 values]
  const uint32_t* stream = cb ? reinterpret_cast<const uint32_t*>(A.a0) : L.packed;
  const float* a0 = cb ? A.a0 + L.H : A.a0;
  const int Q = L.col_groups;
  *nblocks = L.n_row_blocks * Q;
  hipLaunchKernelGGL(kern, dim3(L.n_row_blocks * Q), dim3(MDE_RING_BS), 0, A.st,
                     (int)(A.plan->row_hi - A.plan->row_lo), (int)A.plan->row_lo, (int)A.plan->n,
                     L.rows_per_block, Q, L.n_chunks, L.wave_iter, L.hdr, stream, a0, A.a1, A.a0_scalar,
                     A.a1_scalar, A.X, A.grad, L.partial, A.plan->partials, fn, A.inv_p, A.grad_scale,
                     A.loss_out, A.loss_scale, getenv("MDE_RING_DBG") ? atoi(getenv("MDE_RING_DBG")) : 0);
  MDE_LAUNCH_CHECK();
  if (Q > 1 && A.grad) {
    const int64_t nlocD = (A.plan->row_hi - A.plan->row_lo) * (int64_t)D;
    hipLaunchKernelGGL(k_ring_combine, dim3(mde_grid(nlocD, MDE_BLOCK, 2048)), dim3(MDE_BLOCK), 0, A.st,
                       nlocD, Q, L.partial, A.grad_scale, A.grad + (size_t)A.plan->row_lo * D);
    MDE_LAUNCH_CHECK();
  }
  return MDE_OK;
}

static MdeFuncArgs ring_func_args(const mde_func* f) {
  MdeFuncArgs a;
  a.kind = f->kind;
  a.kind_neg = f->kind_neg;
  a.S = {f->s0, f->s1, f->s2};
  a.N = {f->n0, f->n1, f->n2};
  return a;
}

// Called by mde_average_distortion.  Returns 1 when the ring kernel was launched (it also writes
// *loss_out = loss_scale * sum of the workgroups' partials; nblocks = number of partials), 0 when
// the caller should use the CSR kernel, < 0 on error.
int mde_ring_try(mde_plan* plan, const float* X, int d, const mde_func* f, float grad_scale,
                 float* grad, float inv_p, hipStream_t st, int* nblocks, float* loss_out,
                 double loss_scale) {
  if (!plan->ring.packed || plan->ring.d != d) return 0;
  RingArgs A{plan, X, d, f->a0, f->a1, f->a0_scalar, f->a1_scalar, grad, inv_p, grad_scale, st,
             loss_out, loss_scale};
  const MdeFuncArgs a = ring_func_args(f);
  const int ea = mde_exp_class(f->s0), en = mde_exp_class(f->n0);
  int rc = MDE_OK;
#define RING(FN)                                               \
  do {                                                         \
    FN fn{a};                                                  \
    if (d == 2)                                                \
      rc = launch_ring<2, FN>(A, fn, nblocks);                 \
    else if (d == 3)                                           \
      rc = launch_ring<3, FN>(A, fn, nblocks);                 \
    else if (d == 1)                                           \
      rc = launch_ring<1, FN>(A, fn, nblocks);                 \
    else                                                       \
      rc = launch_ring<4, FN>(A, fn, nblocks);                 \
    return rc == MDE_OK ? 1 : rc;                              \
  } while (0)
  if (d == 2 || d == 3) {
    if (f->kind_neg == MDE_F_NONE) {
      if (f->kind == MDE_F_LOG1P && ea == 2) RING(FnSingle<MDE_F_LOG1P COMMA 2>);
      if (f->kind == MDE_F_QUADRATIC) RING(FnSingle<MDE_F_QUADRATIC COMMA 0>);
    } else if (f->kind == MDE_F_LOG1P && ea == 2) {
      if (f->kind_neg == MDE_F_LOG && en == 1)
        RING(FnPushPull<MDE_F_LOG1P COMMA 2 COMMA MDE_F_LOG COMMA 1>);
      if (f->kind_neg == MDE_F_LOGRATIO && en == 3)
        RING(FnPushPull<MDE_F_LOG1P COMMA 2 COMMA MDE_F_LOGRATIO COMMA 3>);
    }
  }
  RING(FnRuntime);
#undef RING
}
