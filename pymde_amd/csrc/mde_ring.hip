// mde_ring.hip -- LDS-resident variant of the fused average-distortion kernel for small
// embedding dimensions (d <= 4) on graphs whose embedding table does not fit L2 (layout 1): the
// layout builder and the dispatcher.  The kernel itself is mde_ring_kernel.h, instantiated per family
// of distortion functions in mde_ring_k_*.hip.
// [ref: pymde/average_distortion.py:62-106 -- the same loss and gradient]
//
// Why: the CSR kernel's only random access is the gather of x_u, and random 8-byte gathers are
// bound by the L2 request rate (~100 G/s from an 8 MB table: >= 1 ms for 10^8 half-edges whatever
// the HBM rate).  Here every random access is served by LDS:
//
//   * a workgroup of 12 waves owns a block of R rows (R <= 7872 at d = 2) and one of Q column
//     groups: the rows' x_v and their fp32 gradient accumulators stay in LDS for the whole kernel
//     (2 x 63 KB); with Q > 1 the groups' partial gradients are added by k_ring_combine;
//   * the embedding table streams through a ring of S chunk slots in what is left of the 160 KB (9
//     slots of 4 KB at d = 2), staged by 4 producer waves through their VGPRs (global_load_dwordx4 ->
//     ds_write_b128, two chunks in flight per wave: the bytes in flight need no ring slot);
//   * the other 8 waves are consumers.  Each owns a fixed range of the block's rows and walks
//     its own contiguous stream of packed half-edges (absolute LDS addresses of x_v and x_u, 4
//     bytes, chunk-major) -- because a row has exactly one wave that ever touches its accumulator,
//     the update is a plain LDS read-add-write, the summation order is fixed by the layout
//     (bitwise reproducible), and the waves need NO workgroup barrier between prologue and epilogue;
//   * producers and consumers synchronise through a handful of LDS words: a consumer publishes
//     the oldest chunk it still reads (prog[w]), a producer the next chunk it has not landed yet
//     (F[p]); one hand-shake per PAIR of wave iterations, the pair's slots released as soon as its
//     LDS reads are issued;
//   * LDS fp32 atomics would remove the one-writer rule, but ds_add_f32 retires ~0.16 lanes per
//     clock per CU on gfx950 (tools/ldsprobe) -- 20x slower than read-add-write.  The layout
//     builder therefore makes the rows of a wave iteration DISTINCT (an entry whose row is taken
//     waits for the next iteration), caps the entries per LDS bank class on the row and on the
//     column side, and deals the entries to the lanes so that every 32-lane read pass and every
//     16-lane write pass is as shallow as the classes allow;
//   * every edge is evaluated from both endpoints (owner-computes gradient rows), but its LOSS term
//     is added once: by the entry whose row is the smaller vertex.  The header says per BLOCK of four
//     iterations whether none / all / some of its entries do (three copies of the block body, one
//     branch per block: half of the blocks skip the loss arithmetic).
// What bounds it (round 4, profiles/r04_att_summary.md): the consumers' loop is bound by the
// instructions it issues (VALU busy ~75 % of the launch), the staging by its bytes in flight; the two
// add up (0.10 ms + 0.078 ms per GB staged at config 4) because both are work on the same SIMDs.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

#include "mde_ring.h"

// ---------------------------------------------------------------- layout construction
// Row -> consumer wave map of a workgroup (row block rb, column group g).  A row has ONE wave (nobody else
// touches its accumulator), but WHICH wave is free: the packed words carry absolute LDS addresses, so a
// wave's ownership of a row exists only as "which stream its entries are in".  Rounds 2-4 cut the block into 8
// contiguous row ranges of equal half-edge count.  On a random graph the entries a range finds in a chunk are
// Poisson, the ranges' positions along the column sweep drift apart like random walks (at equal iteration
// index the 8 waves of a config-4 workgroup are spread over 5 chunks on average, 8 at the 90th percentile),
// and a ring slot is free only when ALL consumers are past it: with 9 slots and a 5-chunk pair window the
// fast waves wait for the slow ones most of the time (a third of the consumer loop was polling).
// Here the rows are dealt to the waves one by one, greedily: the column sweep of the group is cut into
// MDE_RING_BUCKETS buckets, P_w[b] is the number of entries wave w holds in buckets <= b, and a row with
// prefix counts c[b] goes to the wave that minimises sum_b (P_w[b] - mean_w P[b]) c[b] -- the steepest descent
// of sum_w sum_b (P_w[b] - mean)^2.  Every wave then reaches every bucket boundary with (almost) the same
// number of entries behind it: spread 1.9 chunks on average, 3 at the 90th percentile (simulation and
// MDE_RING_STATS), which the ring's slack of 4 covers.  One wave per workgroup, rows in order.
// wmap[(rb * Q + g) * R + local row] = wave | rank of the row inside its wave << 8; wrows[stream] = rows.
#ifndef MDE_RING_BUCKETS
#define MDE_RING_BUCKETS 32
#endif
__global__ __launch_bounds__(64) void k_ring_assign(int nloc, int R, int Q, int NC, int C, int contiguous,
                                                    const int32_t* __restrict__ rowptr,
                                                    const int32_t* __restrict__ nbr, uint32_t* __restrict__ wmap,
                                                    int32_t* __restrict__ wrows) {
  constexpr int B = MDE_RING_BUCKETS, NCWP = MDE_RING_NCW <= 8 ? 8 : 16, LPW = 64 / NCWP, BPL = B / LPW;
  static_assert(MDE_RING_NCW <= 16 && B % LPW == 0 && B <= 64, "lane map of k_ring_assign");
  __shared__ int hist[B];
  __shared__ float cpre[B];
  const int wg = blockIdx.x, rb = wg / Q, g = wg % Q, lane = threadIdx.x;
  const int j_lo = (int)(((int64_t)g * NC + Q - 1) / Q), j_hi = (int)(((int64_t)(g + 1) * NC + Q - 1) / Q);
  const int nj = max(1, j_hi - j_lo);
  const int r0 = rb * R, nr = min(R, nloc - r0);
  const int w = lane / LPW, bg = lane % LPW;
  uint32_t* out = wmap + (size_t)wg * R;
  if (contiguous) {
    // (MDE_RING_ASSIGN=0, design comparison: the contiguous ranges of equal half-edge count of rounds 2-4)
    const int64_t lo = rowptr[r0], hi = rowptr[r0 + nr];
    int bd[MDE_RING_NCW + 1];
    for (int t = 0; t <= MDE_RING_NCW; ++t) {
      const int64_t target = lo + ((hi - lo) * t) / MDE_RING_NCW;
      int a = r0, b = r0 + nr;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (rowptr[mid] >= target) b = mid; else a = mid + 1;
      }
      bd[t] = (t == MDE_RING_NCW) ? r0 + nr : a;
    }
    for (int r = lane; r < nr; r += 64) {
      int ww = 0;
      for (int t = 1; t < MDE_RING_NCW; ++t) ww += (bd[t] <= r0 + r);
      out[r] = (uint32_t)ww | ((uint32_t)(r0 + r - bd[ww]) << 8);
    }
    if (lane < MDE_RING_NCW) wrows[wg * MDE_RING_NCW + lane] = bd[lane + 1] - bd[lane];
    return;
  }
  float P[BPL], T[BPL];
#pragma unroll
  for (int k = 0; k < BPL; ++k) P[k] = T[k] = 0.0f;
  int myrows = 0;
  for (int r = 0; r < nr; ++r) {
    const int beg = rowptr[r0 + r], end = rowptr[r0 + r + 1];
    if (lane < B) hist[lane] = 0;
    __syncthreads();
    for (int q = beg + lane; q < end; q += 64) {
      const int j = nbr[q] / C;
      if (j >= j_lo && j < j_hi) atomicAdd(&hist[(int)(((int64_t)(j - j_lo) * B) / nj)], 1);
    }
    __syncthreads();
    int h = lane < B ? hist[lane] : 0;
#pragma unroll
    for (int o = 1; o < B; o <<= 1) {
      const int t = __shfl_up(h, o, 64);
      if (lane >= o) h += t;
    }
    if (lane < B) cpre[lane] = (float)h;
    __syncthreads();
    float c[BPL], cost = 0.0f;
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
      c[k] = cpre[bg * BPL + k];
      cost = fmaf(P[k] - T[k] * (1.0f / MDE_RING_NCW), c[k], cost);
    }
#pragma unroll
    for (int o = 1; o < LPW; o <<= 1) cost += __shfl_xor(cost, o, 64);
    // the best wave: lowest cost, then fewest rows (rows without entries in this group are dealt evenly), then lowest id
    float bc = w < MDE_RING_NCW ? cost : 3.0e38f;
    int bn = myrows, bw = w;
#pragma unroll
    for (int o = LPW; o < 64; o <<= 1) {
      const float oc = __shfl_xor(bc, o, 64);
      const int on = __shfl_xor(bn, o, 64), ow = __shfl_xor(bw, o, 64);
      const bool take = oc < bc || (oc == bc && (on < bn || (on == bn && ow < bw)));
      bc = take ? oc : bc;
      bn = take ? on : bn;
      bw = take ? ow : bw;
    }
    if (lane == 0) out[r] = (uint32_t)bw | ((uint32_t)bn << 8);
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
      T[k] += c[k];
      if (w == bw) P[k] += c[k];
    }
    myrows += (w == bw);
  }
  if (bg == 0 && w < MDE_RING_NCW) wrows[wg * MDE_RING_NCW + w] = myrows;
}

// key = ((rb * Q + column group) * NCW + wave) << JB | chunk; CSR order (row, edge id) is kept
// inside equal keys by the stable sort
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_keys(int nrows, const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ nbr,
                                                         const uint32_t* __restrict__ wmap, int R, int Q,
                                                         int NC, int C, int JB, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals, int32_t* __restrict__ hrow) {
  constexpr int G = 16;  // (>= the largest Q: lane t of a group holds the row's wave in column group t)
  const int lig = threadIdx.x & (G - 1);
  const int group = (blockIdx.x * MDE_BLOCK + threadIdx.x) / G;
  const int ngroups = (gridDim.x * MDE_BLOCK) / G;
  const int rmax = ((nrows + ngroups - 1) / ngroups) * ngroups;  // (every group runs the same trips: shuffles below)
  for (int r = group; r < rmax; r += ngroups) {
    const bool has = r < nrows;
    const int beg = has ? rowptr[r] : 0, end = has ? rowptr[r + 1] : 0;
    const int rb = has ? r / R : 0;
    const uint32_t wv = (has && lig < Q) ? (wmap[((size_t)rb * Q + lig) * R + (r - rb * R)] & 0xffu) : 0u;
    for (int q0 = beg; q0 < end; q0 += G) {
      const int q = q0 + lig;
      const bool in = q < end;
      const uint32_t j = in ? (uint32_t)(nbr[q] / C) : 0u;
      const uint32_t g = (uint32_t)(((uint64_t)j * (uint64_t)Q) / (uint64_t)NC);
      const uint32_t w = (uint32_t)__shfl((int)wv, (int)g, G);
      if (in) {
        keys[q] = ((((uint32_t)rb * (uint32_t)Q + g) * MDE_RING_NCW + w) << JB) | j;
        vals[q] = (uint32_t)q;
        hrow[q] = r;
      }
    }
  }
}

// ---- round 6: layouts with PEELED hub rows and / or PERMUTED row blocks (plan_rows below)
// row_slot[r] = rb * R + s: LDS slot s of row block rb holds local row r; -1: the row is peeled off to the hub
// kernel.  slot_cum[t] = half-edges in the slots below t.  The builder kernels below are the slot-indexed twins of
// k_ring_assign (contiguous mode) and k_ring_keys; downstream of them `hrow` holds SLOTS, and k_ring_meta /
// k_ring_pack run unchanged (slot - rb * R is the address of the row's x_v and accumulator).
__global__ __launch_bounds__(64) void k_ring_assign_slots(int R, int Q, const int32_t* __restrict__ slot_cum,
                                                          uint32_t* __restrict__ wmap, int32_t* __restrict__ wrows) {
  const int wg = blockIdx.x, rb = wg / Q, lane = threadIdx.x;
  const int s0 = rb * R;
  uint32_t* out = wmap + (size_t)wg * R;
  const int64_t lo = slot_cum[s0], hi = slot_cum[s0 + R];
  int bd[MDE_RING_NCW + 1];
  for (int t = 0; t <= MDE_RING_NCW; ++t) {
    const int64_t target = lo + ((hi - lo) * t) / MDE_RING_NCW;
    int a = s0, b = s0 + R;
    while (a < b) {
      const int mid = (a + b) >> 1;
      if (slot_cum[mid] >= target) b = mid; else a = mid + 1;
    }
    bd[t] = (t == MDE_RING_NCW) ? s0 + R : a;
  }
  for (int r = lane; r < R; r += 64) {
    int ww = 0;
    for (int t = 1; t < MDE_RING_NCW; ++t) ww += (bd[t] <= s0 + r);
    out[r] = (uint32_t)ww | ((uint32_t)(s0 + r - bd[ww]) << 8);
  }
  if (lane < MDE_RING_NCW) wrows[wg * MDE_RING_NCW + lane] = bd[lane + 1] - bd[lane];
}

// in: hrow[q] = local row of CSR position q (k_ring_hrow).  out: the stream key of q (the sentinel stream `nseg`,
// which no wave walks, for the entries of peeled rows), vals[q] = q, hrow[q] = the row's slot
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_keys_slots(int64_t H, const int32_t* __restrict__ nbr,
                                                               const int32_t* __restrict__ row_slot,
                                                               const uint32_t* __restrict__ wmap, int R, int Q, int NC, int C,
                                                               int JB, uint32_t nseg, uint32_t* __restrict__ keys,
                                                               uint32_t* __restrict__ vals, int32_t* __restrict__ hrow) {
  for (int64_t q = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; q < H; q += (int64_t)gridDim.x * MDE_BLOCK) {
    const int sl = row_slot[hrow[q]];
    vals[q] = (uint32_t)q;
    if (sl < 0) {
      keys[q] = nseg << JB;
      hrow[q] = 0;
      continue;
    }
    const int rb = sl / R;
    const uint32_t j = (uint32_t)(nbr[q] / C);
    const uint32_t g = (uint32_t)(((uint64_t)j * (uint64_t)Q) / (uint64_t)NC);
    const uint32_t w = wmap[((size_t)rb * Q + g) * R + (sl - rb * R)] & 0xffu;
    keys[q] = ((((uint32_t)rb * (uint32_t)Q + g) * MDE_RING_NCW + w) << JB) | j;
    hrow[q] = sl;
  }
}
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_rows_to_slots(int64_t H, const int32_t* __restrict__ row_slot,
                                                                  int32_t* __restrict__ hrow) {
  for (int64_t q = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; q < H; q += (int64_t)gridDim.x * MDE_BLOCK)
    hrow[q] = max(row_slot[hrow[q]], 0);
}

__global__ __launch_bounds__(MDE_BLOCK) void k_ring_gather_u32(int64_t H, const uint32_t* __restrict__ idx,
                                                               const uint32_t* __restrict__ in,
                                                               uint32_t* __restrict__ out) {
  for (int64_t h = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; h < H; h += (int64_t)gridDim.x * MDE_BLOCK)
    out[h] = in[idx[h]];
}

// hrow[q] = local row of CSR position q
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_hrow(int nrows, const int32_t* __restrict__ rowptr,
                                                         int32_t* __restrict__ hrow) {
  constexpr int G = 16;
  const int lig = threadIdx.x & (G - 1);
  const int group = (blockIdx.x * MDE_BLOCK + threadIdx.x) / G;
  const int ngroups = (gridDim.x * MDE_BLOCK) / G;
  for (int r = group; r < nrows; r += ngroups)
    for (int q = rowptr[r] + lig; q < rowptr[r + 1]; q += G) hrow[q] = r;
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

// Cut each (block, group, wave) stream into wave iterations: up to 64 entries, taken in stream
// order, whose chunks lie within SPAN of the oldest one (the ring holds them all at once).
//   * The rows of an iteration are DISTINCT (always): every lane does its own LDS read-add-write
//     of its row's accumulator, nothing is folded across lanes.  An entry whose row is taken waits
//     for the next iteration (a row's entries still reach its accumulator in stream order); a hub
//     row therefore costs padding, and mde_plan_layout gives up on the ring when that gets out of
//     hand (auto mode).
//   * Bank classes: an iteration takes at most `cap` entries per row bank class and per column
//     bank class (the 8-byte LDS slot mod 32 at d = 2) -- with the lane placement of k_ring_pack
//     every 32-lane pass of its LDS reads is then at most cap/2 deep (tools/sched_sim.cpp: cap 4
//     costs ~1 % padding and takes the modelled LDS clocks per iteration from 26 to 22; the
//     unplaced, uncapped layout of round 2 measured ~40).  An entry that lost `force` times goes
//     out regardless of the caps (never against a taken row).
// Lanes freed by waiting entries are refilled from the stream (a few rounds), so iterations stay
// full.  One wave per stream; FILL = false counts the iterations, FILL = true writes, per iteration,
// the sorted positions of its entries, their number and the oldest chunk still needed (it_m).
#ifndef MDE_RING_FORCE
#define MDE_RING_FORCE 3
#endif
#ifndef MDE_RING_CARRY
#define MDE_RING_CARRY 256  // waiting entries a stream can hold
#endif
#ifndef MDE_RING_ROUNDS
#define MDE_RING_ROUNDS 6    // batches of new entries offered per iteration
#endif
__host__ __device__ constexpr int ring_cls_shift(int d) { return d == 2 ? 3 : (d == 4 ? 4 : 2); }
__host__ __device__ constexpr int ring_cls_mask(int d) { return d == 4 ? 15 : 31; }
// meta[pos] for every sorted position: rank of the row inside its wave (bits 15:0) | row bank class (23:16) | column bank
// class (31:24) -- everything the scheduler needs of an entry besides its chunk (keys[pos] & JM), in
// stream order: the scheduler's loads are contiguous (round 3 chased pos -> vals -> hrow / nbr, three
// dependent random loads per candidate: 2 x 43 ms at config 4)
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_meta(int64_t H, const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals,
                                                         const int32_t* __restrict__ hrow,
                                                         const int32_t* __restrict__ nbr,
                                                         const uint32_t* __restrict__ wmap, int JB, int R, int Q, int d,
                                                         uint32_t* __restrict__ meta) {
  const int sh = ring_cls_shift(d), cm = ring_cls_mask(d), rowbytes = 4 * d;
  for (int64_t pos = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; pos < H; pos += (int64_t)gridDim.x * MDE_BLOCK) {
    const uint32_t q = vals[pos];
    const int wg = (int)((keys[pos] >> JB) / MDE_RING_NCW), rb = wg / Q;
    const uint32_t row = (uint32_t)(hrow[q] - rb * R);
    const uint32_t rc = ((row * (uint32_t)rowbytes) >> sh) & (uint32_t)cm;
    const uint32_t cc = (((uint32_t)nbr[q] * (uint32_t)rowbytes) >> sh) & (uint32_t)cm;  // (chunks start on class 0)
    meta[pos] = (wmap[(size_t)wg * R + row] >> 8) | (rc << 16) | (cc << 24);
  }
}

template <bool FILL>
__global__ __launch_bounds__(64) void k_ring_schedule(int nseg, const int32_t* __restrict__ seg,
                                                      const int32_t* __restrict__ wrows,
                                                      const uint32_t* __restrict__ keys,
                                                      const uint32_t* __restrict__ meta, uint32_t JM, int SPAN,
                                                      int R, int Q, int NC, int d, int cap, int bail_factor,
                                                      int32_t* __restrict__ iters,
                                                      const int32_t* __restrict__ iter_base,
                                                      int32_t* __restrict__ it_ent, int32_t* __restrict__ it_cnt,
                                                      int32_t* __restrict__ it_m, int flag_rows,
                                                      const int32_t* __restrict__ caps) {
  extern __shared__ int flag[];                  // [flag_rows] per row of the WAVE: priority of the entry that holds it
  __shared__ int cq_pos[2][MDE_RING_CARRY];      // waiting entries (sorted position), double buffered,
  __shared__ uint32_t cq_meta[2][MDE_RING_CARRY];  // ... their meta word,
  __shared__ int cq_chunk[2][MDE_RING_CARRY];    // ... their chunk
  __shared__ int cq_age[2][MDE_RING_CARRY];
  __shared__ int em[64], em_row[64];             // entries of the iteration being formed
  __shared__ int rcnt[32], ccnt[32];             // entries per bank class in it
  const int i = blockIdx.x, lane = threadIdx.x;
  if (i >= nseg) return;
  for (int r = lane; r < flag_rows; r += 64) flag[r] = 0x7fffffff;
  __syncthreads();
  const int beg = seg[i], end = seg[i + 1];
  constexpr int row_base = 0;  // (the meta words carry the rank of a row inside its wave)
  const int first = FILL ? iter_base[i] : 0;
  int out = first, next = beg, nc = 0, cur = 0;
  // counting pass, auto mode: a stream that needs several times the iterations its entries would fill
  // (a hub row: one entry per iteration) makes the caller give the layout up -- stop counting there
  // (an iteration holds at most one entry per row of the wave: a short tail block is not a hub)
  const int nrows_w = max(1, wrows[i]);
  const int per_it = nrows_w < 64 ? nrows_w : 64;
  // (a sparse stream -- the short last row block -- needs its iterations for the chunk windows it has to
  // walk, a pair of iterations per SPAN + 1 chunks: that is not a hub either)
  const int walk = 2 * ((NC / Q + SPAN + 1) / (SPAN + 1));
  // FILL with `caps`: the single-pass build -- the stream writes into a region of caps[i] iterations
  // (the same bound) and reports how many it used
  const int bail = caps ? caps[i] - 4
                        : ((!FILL && bail_factor > 0) ? bail_factor * max((end - beg + per_it - 1) / per_it, walk) + 64 : 0x7fffffff);
  int last_chunk = (int)(((int64_t)((i / MDE_RING_NCW) % Q) * NC + Q - 1) / Q);
  int m_lead = 0;
  while (nc > 0 || next < end) {
    const int m = nc > 0 ? cq_chunk[cur][0] : (int)(keys[next] & JM);  // oldest candidate's chunk
    // the chunk window is anchored at the first iteration of a PAIR (the kernel does one hand-shake
    // per pair: it waits for the newest chunk of both)
    if (((out - first) & 1) == 0) m_lead = m;
    const int lim = m_lead + SPAN;
    int nem = 0, nnew = 0, mx = 0;  // emitted so far, waiting so far (into buffer cur ^ 1), newest chunk emitted
    if (lane < 32) rcnt[lane] = ccnt[lane] = 0;
    __syncthreads();
    // offer a batch of candidates (stream order = ascending priority)
    auto offer = [&](int pos, uint32_t mw, int chunk, int age, int prio) {
      const int row = pos >= 0 ? (int)(mw & 0xffffu) - row_base : 0;  // (relative to the wave's first row)
      const int rc = pos >= 0 ? (int)((mw >> 16) & 0xffu) : -1, cc = pos >= 0 ? (int)(mw >> 24) : -2;
      if (pos >= 0) atomicMin(&flag[row], prio);
      __syncthreads();
      const bool rowwin = pos >= 0 && flag[row] == prio;
      // how many earlier row winners of this batch share my classes (whether or not they pass
      // their own caps: slightly pessimistic, deterministic and parallel)
      int rkR = 0, rkC = 0;
      const int rcw = rowwin ? rc : -1, ccw = rowwin ? cc : -2;
      for (int c = 0; c < 32; ++c) {
        const unsigned long long mr = __ballot(rcw == c), mc = __ballot(ccw == c);
        if (rc == c) rkR = __popcll(mr & ((1ull << lane) - 1ull));
        if (cc == c) rkC = __popcll(mc & ((1ull << lane) - 1ull));
      }
      const bool capok = rowwin && (rcnt[rc] + rkR < cap) && (ccnt[cc] + rkC < cap);
      const bool win = rowwin && (capok || age >= MDE_RING_FORCE);
      // winners beyond the 64th wait as well (they keep their turn: age unchanged)
      const unsigned long long wm = __ballot(win);
      const int widx = nem + __popcll(wm & ((1ull << lane) - 1ull));
      const bool emit = win && widx < 64;
      if (emit) {
        em[widx] = pos;
        em_row[widx] = row;
        atomicAdd(&rcnt[rc], 1);
        atomicAdd(&ccnt[cc], 1);
      }
      mx = max(mx, emit ? chunk : 0);
      const bool lose = pos >= 0 && !emit;
      const unsigned long long lm = __ballot(lose);
      const int lidx = nnew + __popcll(lm & ((1ull << lane) - 1ull));
      if (lose && lidx < MDE_RING_CARRY) {
        cq_pos[cur ^ 1][lidx] = pos;
        cq_meta[cur ^ 1][lidx] = mw;
        cq_chunk[cur ^ 1][lidx] = chunk;
        cq_age[cur ^ 1][lidx] = win ? age : age + 1;
      }
      // (overflow of the waiting queue cannot happen: a batch is only offered while
      // nnew + 64 <= MDE_RING_CARRY)
      nem += __popcll(__ballot(emit));
      nnew += __popcll(lm);
      __syncthreads();
    };
    int prio = 0;
    // the waiting entries first, 64 at a time
    for (int c0 = 0; c0 < nc; c0 += 64) {
      const int k = c0 + lane;
      const bool has = k < nc;
      offer(has ? cq_pos[cur][k] : -1, has ? cq_meta[cur][k] : 0u, has ? cq_chunk[cur][k] : 0, has ? cq_age[cur][k] : 0,
            prio + lane);
      prio += 64;
    }
    // then new entries while lanes are free and the window allows (a few rounds: entries that
    // have to wait leave their lane to the next ones)
    for (int round = 0; round < MDE_RING_ROUNDS && nem < 64 && next < end && nnew + 64 <= MDE_RING_CARRY; ++round) {
      const int want = 64 - nem;
      const int cand = next + lane;
      const bool in = lane < want && cand < end;
      const int chunk = in ? (int)(keys[cand] & JM) : 0x7fffffff;
      const bool take = in && chunk <= lim;
      const int ntake = __popcll(__ballot(take));
      if (ntake == 0) break;
      offer(take ? cand : -1, take ? meta[cand] : 0u, chunk, 0, prio + lane);
      prio += 64;
      next += ntake;
    }
    // the rows claimed in this iteration are released (a waiting entry keeps its row claimed until
    // here: later entries of that row do not overtake it)
    if (lane < nem) flag[em_row[lane]] = 0x7fffffff;
    for (int k = lane; k < nnew; k += 64) flag[(int)(cq_meta[cur ^ 1][k] & 0xffffu) - row_base] = 0x7fffffff;
    mx = mde_wave_max(mx);
    if (nem > 0) last_chunk = mx;  // newest chunk the iteration references
    if (FILL) {
      if (lane < nem) it_ent[(size_t)out * 64 + lane] = em[lane];
      if (lane == 0) {
        it_cnt[out] = nem | (last_chunk << 8);
        it_m[out] = m;
      }
    }
    ++out;
    nc = nnew;
    cur ^= 1;
    __syncthreads();
    if (out - first > bail) {
      if (lane == 0) iters[i] = 1 << 25;  // (x 64 entries = 2^31: over the 32-bit position limit checked below)
      return;
    }
  }
  // a stream is stored in blocks of 4 iterations (one 16-byte load per lane): pad with empty ones
  while ((out - first) & 3) {
    if (FILL && lane == 0) {
      it_cnt[out] = 0 | (last_chunk << 8);
      it_m[out] = last_chunk;
    }
    ++out;
  }
  if (iters && lane == 0) iters[i] = out - first;
}

// caps[i] = iterations reserved for stream i by the single-pass build: `factor` times what its entries
// (or its chunk windows) need at least, a multiple of 4
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_caps(int nseg, const int32_t* __restrict__ seg,
                                                         const int32_t* __restrict__ wrows, int R, int Q, int NC, int SPAN,
                                                         int factor, int32_t* __restrict__ caps) {
  const int i = blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (i > nseg) return;
  if (i == nseg) {
    caps[i] = 0;
    return;
  }
  const int nrows_w = max(1, wrows[i]);
  const int per_it = nrows_w < 64 ? nrows_w : 64;
  const int walk = 2 * ((NC / Q + SPAN + 1) / (SPAN + 1));
  const int need = max((seg[i + 1] - seg[i] + per_it - 1) / per_it, walk);
  caps[i] = (factor * need + 64 + 4 + 3) & ~3;
}

// Lane placement of one iteration, wave-parallel (round 3 ran a serial Euler split on one lane per
// iteration: 170 ms of the layout build).  LDS bank rules of gfx950 (tools/valuprobe): a wave64 b64 read
// is served in two passes of 32 lanes and costs, per pass, the deepest bank class (8-byte slot mod
// 32); a b64 write in four passes of 16 lanes over 16 classes.  The row side carries three of the
// four random accesses of an entry (x_v, accumulator read, accumulator write), so the ROW classes
// are dealt evenly: entries of a class alternate between the two 32-lane halves, and inside a half the
// entries of a class mod 16 alternate between its two 16-lane quarters (classes with an odd count hand
// their extra entry to the halves / quarters in turn).  The column classes are capped by the
// scheduler and otherwise left as they fall.
// in: act (lane holds an entry), rc (its row class, 0..31).  out: the lane that handles the entry.
__device__ __forceinline__ int ring_place_wave(bool act, int rc, int lane) {
  const unsigned long long lt = (1ull << lane) - 1ull;
  int rank = 0, par = 0, oddacc = 0;
  for (int c = 0; c < 32; ++c) {
    const unsigned long long m = __ballot(act && rc == c);
    if (rc == c) {
      rank = __popcll(m & lt);
      par = oddacc & 1;
    }
    oddacc += __popcll(m) & 1;
  }
  const int half = (rank + par) & 1;
  int quarter = 0;
  for (int h = 0; h < 2; ++h) {
    int oa = 0;
    for (int c = 0; c < 16; ++c) {
      const unsigned long long m = __ballot(act && half == h && (rc & 15) == c);
      if (half == h && (rc & 15) == c) quarter = (__popcll(m & lt) + (oa & 1)) & 1;
      oa += __popcll(m) & 1;
    }
  }
  int pos = 0;
  for (int g = 0; g < 4; ++g) {
    const unsigned long long m = __ballot(act && half * 2 + quarter == g);
    if (half * 2 + quarter == g) pos = __popcll(m & lt);
  }
  // (a quarter holds at most 16: a half gets at most ceil(cnt / 2) <= 32 entries, a quarter half of that)
  return half * 32 + quarter * 16 + pos;
}

// One wave per iteration: deal its <= 64 entries to the lanes, pad to 64 with dummies, write
// packed words, edge ids and the header.
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_pack(int64_t nit, int nseg, const int32_t* __restrict__ iter_base,
                                                         const int32_t* __restrict__ src_base,
                                                         const int32_t* __restrict__ it_ent,
                                                         const int32_t* __restrict__ it_cnt,
                                                         const int32_t* __restrict__ it_m,
                                                         const uint32_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ vals,
                                                         const int32_t* __restrict__ hrow,
                                                         const int32_t* __restrict__ nbr,
                                                         const int32_t* __restrict__ eid, int R, int Q, int C, int S,
                                                         int ring_off, int JB, int d, int row_lo, int place, int count_all,
                                                         uint32_t* __restrict__ packed,
                                                         int32_t* __restrict__ peid, uint32_t* __restrict__ hdr) {
  __shared__ uint8_t s_taken[MDE_BLOCK / 64][64], s_free[MDE_BLOCK / 64][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t nw = ((int64_t)gridDim.x * MDE_BLOCK) >> 6;
  const uint32_t JM = (1u << JB) - 1u;
  const int sh = ring_cls_shift(d), cm = ring_cls_mask(d);
  const uint32_t cbytes = (uint32_t)C * 4u * (uint32_t)d;
  // (the waves of a block run the same number of trips: the barriers below are block-wide)
  for (int64_t itb = (int64_t)blockIdx.x * (MDE_BLOCK / 64); itb < nit; itb += nw) {
    const int64_t it = itb + wv;
    const bool valid = it < nit;
    // where the scheduler left iteration `it` (the single-pass build reserves a region per stream):
    // stream s = the last one with iter_base[s] <= it
    int64_t src = it;
    if (valid && src_base) {
      int lo = 0, hi = nseg;  // iter_base[lo] <= it < iter_base[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)iter_base[mid] <= it) lo = mid; else hi = mid;
      }
      src = (int64_t)src_base[lo] + (it - (int64_t)iter_base[lo]);
    }
    const int cw = valid ? __builtin_amdgcn_readfirstlane(it_cnt[src]) : 0;  // entries | newest chunk << 8
    const int cnt = cw & 0xff;
    const bool act = valid && lane < cnt;
    uint32_t key = 0, q = 0, col = 0, rowaddr = 0, ring = 0;
    int grow = 0, rcls = 0;
    if (act) {
      const int pos = it_ent[(size_t)src * 64 + lane];
      key = keys[pos];
      q = vals[pos];
      const int rb = (int)((key >> JB) / MDE_RING_NCW) / Q;
      grow = hrow[q];
      rowaddr = (uint32_t)(grow - rb * R) * 4u * (uint32_t)d;
      const uint32_t j = key & JM;
      col = (uint32_t)nbr[q];
      ring = (uint32_t)ring_off + (j % (uint32_t)S) * cbytes + (col - j * (uint32_t)C) * 4u * (uint32_t)d;
      rcls = (int)((rowaddr >> sh) & (uint32_t)cm);
    }
    s_taken[wv][lane] = 0;
    __syncthreads();
    int l = lane;
    if (place) l = ring_place_wave(act, rcls, lane);
    if (act) s_taken[wv][l] = 1;
    __syncthreads();
    {
      // the lanes nobody took, in ascending order, for the padding entries
      const bool fr = !s_taken[wv][lane];
      const unsigned long long fm = __ballot(fr);
      if (fr) s_free[wv][__popcll(fm & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    }
    __syncthreads();
    // whose loss terms: the entry whose row is the smaller vertex of the edge (grow is the local row there); a
    // permuted layout (grow is a slot) adds every entry's with weight 1/2
    const bool counts = act && (count_all || (uint32_t)(row_lo + grow) < col);
    const int ncount = __popcll(__ballot(counts));
    const uint32_t lclass = ncount == 0 ? 0u : (ncount == cnt ? 1u : 2u);
    // chunk window of the iteration: the oldest chunk its stream still needs (waiting entries
    // included) .. the newest chunk it references
    const uint32_t m = valid ? (uint32_t)__builtin_amdgcn_readfirstlane(it_m[src]) : 0u;
    const uint32_t need = max((uint32_t)(cw >> 8), m);
    // the padding lanes read a column that is resident for sure: the first one of the oldest chunk
    // of the PAIR of iterations (the kernel waits for that chunk before it issues the pair's reads)
    const uint32_t mpad = valid ? (uint32_t)__builtin_amdgcn_readfirstlane(it_m[src & ~(int64_t)1]) : 0u;  // (regions start on multiples of 4)
    // element (iteration it, lane l) of a stream lives at ((it / 4) * 64 + l) * 4 + it % 4
    const size_t base = ((size_t)(it >> 2) * 64) * 4 + (size_t)(it & 3);
    if (act) {
      packed[base + (size_t)l * 4] = ring_pack_word(d, rowaddr, ring);
      peid[base + (size_t)l * 4] = eid[q];
    } else if (valid) {
      // padding: a dummy row slot of the lane's own bank class, a resident column
      const int lf = s_free[wv][lane - cnt];
      packed[base + (size_t)lf * 4] = ring_pack_word(d, (uint32_t)(R + (lf & 31)) * 4u * (uint32_t)d,
                                                     (uint32_t)ring_off + (mpad % (uint32_t)S) * cbytes);
      peid[base + (size_t)lf * 4] = -1;
    }
    if (lane == 0 && valid) {
      hdr[MDE_RING_HW * it] = m;
      hdr[MDE_RING_HW * it + 1] = need;
      hdr[MDE_RING_HW * it + 2] = lclass;
      hdr[MDE_RING_HW * it + 3] = (cnt < 64 ? MDE_RING_H3_PAD : 0u) | ((uint32_t)cnt << 8);
    }
    __syncthreads();
  }
}

// loss class of every block of four iterations (header word 3 of its first iteration, bits 1:0)
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_block_class(int64_t nblocks, uint32_t* __restrict__ hdr) {
  const int64_t b = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (b >= nblocks) return;
  bool any0 = false, any1 = false, any2 = false;
  for (int q = 0; q < 4; ++q) {
    const uint32_t* h = hdr + (size_t)MDE_RING_HW * (4 * b + q);
    if ((h[3] >> 8) == 0u) continue;  // an empty iteration adds nothing whatever the class
    any0 |= h[2] == 0u;
    any1 |= h[2] == 1u;
    any2 |= h[2] == 2u;
  }
  const uint32_t cls = (any2 || (any0 && any1)) ? 2u : (any1 ? 1u : 0u);
  hdr[(size_t)MDE_RING_HW * 4 * b + 3] = (hdr[(size_t)MDE_RING_HW * 4 * b + 3] & ~3u) | cls;
  // one hand-shake per pair of iterations: the first one's `need` covers both.  An EMPTY iteration
  // reads no column (its lanes sit on the first column of the pair's oldest chunk, k_ring_pack) and
  // asks for nothing: on a sparse stream its own m may lie far beyond the pair's window.
  for (int t = 0; t < 2; ++t) {
    uint32_t* h0 = hdr + (size_t)MDE_RING_HW * (4 * b + 2 * t);
    uint32_t* h1 = h0 + MDE_RING_HW;
    uint32_t need = h0[0];
    if ((h0[3] >> 8) != 0u) need = max(need, h0[1]);
    if ((h1[3] >> 8) != 0u) need = max(need, h1[1]);
    h0[1] = need;
  }
}

// stats[0] = iterations with padding, stats[1] = iterations that add loss terms for every entry,
// stats[2] = iterations with the per-lane loss test
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_stats(int64_t nit, const uint32_t* __restrict__ hdr,
                                                          unsigned long long* __restrict__ stats) {
  unsigned long long a = 0, b = 0, c = 0;
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < nit; i += (int64_t)gridDim.x * MDE_BLOCK) {
    a += (hdr[MDE_RING_HW * i + 3] & MDE_RING_H3_PAD) ? 1u : 0u;
    b += hdr[MDE_RING_HW * i + 2] == 1u;
    c += hdr[MDE_RING_HW * i + 2] == 2u;
  }
  a = mde_wave_sum(a);
  b = mde_wave_sum(b);
  c = mde_wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&stats[0], a);
    atomicAdd(&stats[1], b);
    atomicAdd(&stats[2], c);
  }
}

// How evenly a workgroup's consumer waves advance along the column sweep (round 6, late).  The ring couples the
// eight waves of a workgroup: nobody can be more than the ring's slack (~4 chunks) ahead of the slowest.  On a uniform
// graph every wave finds about the same number of entries in every chunk and the coupling costs nothing; on a graph
// whose links stay near the diagonal of the vertex order (data sorted by class: 2/3 of the links inside clusters of
// 1000 consecutive items) each wave's entries sit in the few chunks that hold ITS rows' clusters, the waves take
// turns instead of working side by side, and the launch takes what ONE wave at a time would: n = 1M, 50M such edges
// ran 1.28 ms on the ring kernel against 0.66 on the CSR kernels (0.16 for the uniform graph), n = 100k 0.22 against
// 0.05 (tools/r6_clusters.sh).  crit[wg] = sum over groups of GW chunks of the LARGEST entry count among the
// workgroup's waves, tot[wg] = all its entries: crit / (tot / NCW) is ~1.1 for a uniform graph (the maximum of eight
// Poisson counts), up to NCW when the waves' entries never share a chunk group.  One thread per (workgroup, group):
// sixteen binary searches in the sorted keys.
__global__ __launch_bounds__(MDE_BLOCK) void k_ring_sweep_balance(int nwg, int ngroups, int GW, int JB,
                                                                  const uint32_t* __restrict__ keys,
                                                                  const int32_t* __restrict__ seg,
                                                                  unsigned long long* __restrict__ crit,
                                                                  unsigned long long* __restrict__ tot) {
  const int64_t idx = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x;
  if (idx >= (int64_t)nwg * ngroups) return;
  const int wg = (int)(idx / ngroups), g = (int)(idx % ngroups);
  unsigned long long mx = 0, sm = 0;
  for (int w = 0; w < MDE_RING_NCW; ++w) {
    const uint32_t s = (uint32_t)wg * MDE_RING_NCW + (uint32_t)w;
    const int32_t lo = seg[s], hi = seg[s + 1];
    const uint64_t k0 = ((uint64_t)s << JB) + (uint64_t)g * GW, k1 = k0 + (uint64_t)GW;
    auto lower = [&](uint64_t key) {
      int32_t a = lo, b = hi;  // first position in [lo, hi) whose key >= `key`
      while (a < b) {
        const int32_t m = a + ((b - a) >> 1);
        if ((uint64_t)keys[m] < key) a = m + 1; else b = m;
      }
      return a;
    };
    const unsigned long long c = (unsigned long long)(lower(k1) - lower(k0));
    mx = c > mx ? c : mx;
    sm += c;
  }
  if (sm) {
    atomicAdd(&crit[wg], mx);
    atomicAdd(&tot[wg], sm);
  }
}

// MDE_PANEL env: unset / -1 auto, 0 never, 1 whenever the layout is feasible (read at every layout
// decision, so a test can switch it inside one process)
static int panel_mode() {
  const char* e = getenv("MDE_PANEL");
  return e ? atoi(e) : -1;
}

static int bits_for_u64(uint64_t maxval) {
  int b = 1;
  while (b < 64 && (maxval >> b)) ++b;
  return b;
}

struct RingSizes {
  int R, NRB, Q, C, NC, S, JB, ring_off, span, qmax;
};

// What an evaluation costs on either kernel, in microseconds -- the rule that picks the layout since round 6.  A ring
// workgroup takes whichever is longer: its consumer waves' iterations (~0.21 us each: config 4, 776 per wave,
// 0.16 ms) or the chunks that pass through its ring (~0.14 us each -- the hand-shake rate, not the bytes: n = 2M at
// degree 50 runs 3907 chunks per workgroup in 0.55 ms whatever the producers' depth or number, tools/r6_prod_sweep.sh);
// the launch runs ceil(workgroups / 256) rounds of them.  The CSR kernels gather x_u from L2 / HBM: 1.1-1.7 ms per
// 1e8 half-edges when the table overflows L2 (profiles/r06_cliff_probe.txt), ~0.6 when it does not.
static double ring_time_us(double iters_per_wave, int NC, int Q, int nwg) {
  const double rounds = std::ceil((double)nwg / 256.0);
  return rounds * std::max(0.21 * iters_per_wave, 0.14 * (double)NC / (double)Q);
}
static double csr_time_us(const mde_plan* plan, int d) {
  return (double)plan->H * ((int64_t)plan->n * d * 4 >= (6 << 20) ? 1.3e-5 : 0.6e-5);
}
// The regime between the MNIST-sized problems and the benchmark shape (round 6, late): the table sits in L2 and there
// are fewer than 16 M half-edges -- rounds 3-6 kept the CSR kernels there without asking.  tools/r6_midsize2.sh, ring
// forced against auto, d = 2, ms per evaluation: n = 150k at degree 50 0.0935 -> 0.0356, 100k 0.0604 -> 0.0273, 50k
// 0.0357 -> 0.0187; at degree 20 150k 0.0402 -> 0.0241, 100k 0.0312 -> 0.0214, 50k 0.0186 -> 0.0164, 20k 0.0162 -> 0.0151:
// the ring's iterations cost what they cost at config 4, in front of ~11 us of fixed cost (two launches, a prologue
// that fills 61 KB of LDS per workgroup, the groups' partial rows), the CSR kernels' 0.6e-5 us per half-edge sit on ~7.
// d = 2 and 3 only: d = 1 and 4 run the run-time functor on the ring kernel (2.7 x the compile-time kinds).
static bool mid_regime(const mde_plan* plan, int d) {
  return (int64_t)plan->n * d * 4 < (6 << 20) && plan->H < ((int64_t)16 << 20);
}
#define MDE_RING_FIXED_US 11.0
#define MDE_CSR_FIXED_US 7.0
// ... and the ring kernel's iterations cost what the FUNCTION costs (profiles/r05_function_sweep.txt, ms per step at
// config-4 size: Quadratic 0.157 .. Huber 0.178, Logistic / Power / Log / SoftFractional 0.20-0.23, PushAndPull 0.2065,
// the run-time functor 0.44), while the CSR kernels' gathers hide most of it: a neighbour-preserving problem of 100k
// items (3.9 M half-edges, PushAndPull(Log1p, LogRatio)) runs 0.0296 ms on the CSR kernel and 0.0329 on the ring, at
// 200k items 0.0509 against 0.0432 (tools/r6_pn_scale.py).  mde_plan_function_hint tells the plan which function is
// coming; the scale multiplies the ring's estimate in the mid regime only.
static double function_cost_scale(int kind, int kind_neg) {
  if (kind_neg != MDE_F_NONE) return 1.3;
  switch (kind) {
    case MDE_F_LOG1P: case MDE_F_QUADRATIC: case MDE_F_LINEAR: case MDE_F_CUBIC: case MDE_F_INVPOWER: case MDE_F_HUBER:
    case MDE_F_L_QUADRATIC: case MDE_F_L_WEIGHTED_QUADRATIC: case MDE_F_L_ABSOLUTE: case MDE_F_L_HUBER: case MDE_F_L_CUBIC:
      return 1.05;
    case MDE_F_LOGISTIC: case MDE_F_POWER: case MDE_F_LOG: case MDE_F_SIGMOID: case MDE_F_HINGE: case MDE_F_LOGRATIO:
    case MDE_F_L_POWER: case MDE_F_L_LOGISTIC: case MDE_F_L_FRACTIONAL: case MDE_F_L_SOFT_FRACTIONAL:
      return 1.35;
    default:
      return 2.7;  // private kinds: the run-time functor
  }
}
extern "C" int mde_plan_function_hint(mde_plan* plan, int32_t kind, int32_t kind_neg) {
  if (!plan) return MDE_E_INVALID;
  plan->ring.cost_scale = (float)function_cost_scale(kind, kind_neg);
  return MDE_OK;
}

// Decide the block height and the column groups for dimension d; false when the layout is not
// worthwhile (the caller keeps the CSR kernel).
// Column groups per row block.  Default: one workgroup per CU (256 / row blocks).  Round 6: more than that when the
// cost model says so -- 193 row blocks (d = 3 at n = 1M) leave a quarter of the CUs idle with Q = 1 and need two
// rounds with Q = 2; with Q = 5 the 965 workgroups run 4 rounds of a fifth of the table each: 0.496 -> 0.444 ms
// (tools/r6_d3_sweep.sh), which the model predicts (547 -> 452 us).  A workgroup's fixed cost (prologue, epilogue,
// partial rows) is priced at 4 us; the default stays unless another Q is 5 % better.  MDE_RING_Q: design probe.
static int choose_col_groups(const mde_plan* plan, int64_t nrb, int qmax, int64_t nc) {
  int Q = (int)std::min<int64_t>(qmax, std::max<int64_t>(1, 256 / nrb));
  const double its1 = 1.4 * (double)plan->H / ((double)nrb * MDE_RING_NCW * 64.0);
  auto cost = [&](int q) {
    const double rounds = std::ceil((double)(nrb * q) / 256.0);
    return ring_time_us(its1 / q, (int)nc, q, (int)(nrb * q)) + 4.0 * rounds;
  };
  const double base = cost(Q);
  double best = base;
  for (int q = 1; q <= qmax; ++q)
    if ((int64_t)nrb * q <= MDE_MAX_PARTIALS && cost(q) < 0.95 * base && cost(q) < best) {
      best = cost(q);
      Q = q;
    }
  if (getenv("MDE_RING_Q")) Q = std::max(1, std::min(qmax, atoi(getenv("MDE_RING_Q"))));
  return Q;
}

static bool choose_sizes(const mde_plan* plan, int d, RingSizes* z) {
  const int64_t nloc = plan->row_hi - plan->row_lo;
  if (d < 1 || d > 4 || nloc <= 0 || plan->H <= 0) return false;
  const int mode = panel_mode();
  if (mode == 0) return false;
  // Geometry: NRB row blocks x Q column groups ~ one workgroup per CU (256).  Tall blocks first:
  // the staging volume is NRB x table bytes (every workgroup streams the table past its rows), and
  // many rows per consumer wave keep the distinct-rows iterations full.  The block height is capped
  // by what fits the LDS next to a ring of >= 8 chunks (ring_row_cap); the CUs left over are filled
  // with column groups of at least `minch` chunks each (one ring window), whose partial rows a second
  // launch adds.  Config 4 (n = 1M, d = 2): 128 blocks of 7872 rows x 2 column groups; an 8-way shard
  // of it: 16 blocks x 16 groups; a dense 40k-node problem: 32 blocks of 1280 rows x 8 groups.
  const int C = ring_chunk_cols(d);
  const int64_t nc = (plan->n + C - 1) / C;
  if (nc > 65535 || nc < 2) return false;
  // (a stream ends with a few thin iterations -- the waiting entries drain under the distinct-rows rule --
  // and is padded to whole blocks: a column group must be long enough to amortise that; 10 chunks are
  // the column counts round 3 tuned with chunks twice as wide)
  const int minch = getenv("MDE_RING_MINCH") ? std::max(1, atoi(getenv("MDE_RING_MINCH"))) : 10;
  const int qmax = (int)std::min<int64_t>(16, std::max<int64_t>(1, nc / minch));
  int64_t pr = (nloc * qmax + 255) / 256;
  if (pr < 64 * MDE_RING_NCW) pr = 64 * MDE_RING_NCW;  // >= 64 rows per consumer wave, even if CUs stay empty
  if (getenv("MDE_RING_ROWS")) pr = atoi(getenv("MDE_RING_ROWS"));
  if (pr > nloc) pr = nloc;
  pr = ((pr + 63) / 64) * 64;
  const int64_t pr_max = ring_row_cap(d);
  if (pr > pr_max) pr = pr_max;
  const int64_t nrb = (nloc + pr - 1) / pr;
  int Q = choose_col_groups(plan, nrb, qmax, nc);
  if (nrb * Q > MDE_MAX_PARTIALS) return false;
  const int jb = bits_for_u64((uint64_t)nc - 1);
  if (bits_for_u64((uint64_t)(nrb * Q * MDE_RING_NCW)) + jb > 32) return false;
  if (mode != 1) {
    // auto: a 64-entry iteration must fit the ring window, and the launch must be worth its fixed
    // costs: either the table overflows L2 (the CSR kernel's gathers go to HBM), or there are enough
    // half-edges that 64 B of L2 traffic per 8-byte gather is what the CSR kernel spends its time on
    // (40k nodes, 100M half-edges: 0.60 -> 0.30 ms per evaluation)
    // Round 6: in that regime the cost model is asked too, with both sides' fixed costs (below); d = 1 and 4 stay
    if (mid_regime(plan, d) && d != 2 && d != 3) return false;
  }
  const int S = ring_slots_for(d, (int)pr);
  if (S < 6) return false;
  int span = ring_max_span(S);
  if (getenv("MDE_RING_SPAN")) span = std::min(S - 2, std::max(1, atoi(getenv("MDE_RING_SPAN"))));
  // auto (rounds 3-5): "a pair of 64-entry iterations must fit the ring window", i.e. >= 128 / (0.8 span) entries
  // per consumer wave and chunk -- which sent d = 3 at n = 1M, n >= 1.6M at degree 50 and everything sparser to
  // the CSR kernels at 1.1-1.7 ms per 1e8 half-edges, although the ring kernel with half-filled iterations takes
  // 0.27-0.5 there (profiles/r06_cliff_probe.txt).  Round 6: the cost model decides (sparse streams pad, so their
  // iterations are priced at 1.4 x the minimum; the count after scheduling is checked again in build_ring).
  if (mode != 1) {
    const double its = 1.4 * (double)plan->H / ((double)nrb * Q * MDE_RING_NCW * 64.0);
    if (ring_time_us(its, (int)nc, Q, (int)(nrb * Q)) > 0.75 * csr_time_us(plan, d)) return false;
    if (mid_regime(plan, d) &&
        plan->ring.cost_scale * ring_time_us(its, (int)nc, Q, (int)(nrb * Q)) + MDE_RING_FIXED_US >
            0.7 * (csr_time_us(plan, d) + MDE_CSR_FIXED_US))
      return false;
  }
  z->qmax = qmax;
  z->R = (int)pr;
  z->NRB = (int)nrb;
  z->Q = Q;
  z->C = C;
  z->NC = (int)nc;
  z->S = S;
  z->ring_off = ring_off_for(d, (int)pr);
  z->span = span;
  z->JB = jb;
  return true;
}

static void ring_free(mde_ring_layout& L) {
  if (L.packed) (void)hipFree(L.packed);
  if (L.eid) (void)hipFree(L.eid);
  if (L.hdr) (void)hipFree(L.hdr);
  if (L.wave_iter) (void)hipFree(L.wave_iter);
  if (L.partial) (void)hipFree(L.partial);
  void* extra[] = {L.slot_row, L.hub_rows, L.hub_seg, L.hub_first, L.hub_partial};
  for (void* q : extra)
    if (q) (void)hipFree(q);
  L = mde_ring_layout();
}
void mde_ring_release(mde_plan* plan) { ring_free(plan->ring); }

// ---------------------------------------------------------------- round 6: which rows go where
// The layout of rounds 3-5 cut the vertex order into blocks of R consecutive rows and put every row's entries into
// the ring streams.  Two kinds of graph broke it (profiles/r06_cliff_probe.txt):
//   * HUB rows.  The rows of a wave iteration are distinct (one lane = one accumulator), so a row with more
//     entries than its wave's stream has iterations stretches that stream: one vertex of degree 5e5 in a
//     config-4 graph made one wave run 250 336 iterations where the others run 772 (28 ms per evaluation, and auto
//     mode gave the whole layout up for the CSR kernel at 1.4 ms).  Rows with more than T = max(256, H / (blocks x
//     NCW x 128)) half-edges -- half a stream's iterations per column group -- are PEELED: no ring stream holds
//     their entries (sentinel key), and k_hub_rows / k_hub_finish evaluate them from the CSR plan behind the ring
//     kernel.  Their rows are disjoint from the ring's: still one writer per row, fixed order, no atomics.
//   * degrees that DRIFT along the vertex order (preferential attachment: the early vertices collect the edges).
//     Blocks of consecutive rows then hold unequal work and the launch runs as long as its heaviest workgroup
//     (0.80 ms on a 1M-vertex preferential-attachment graph against 0.16 uniform).  When the heaviest block holds
//     more than 1.08 x the mean, the rows are DEALT to the blocks instead: sorted by degree (counting sort) and
//     handed out boustrophedon, every block gets the same number of rows and, to within one row's degree, the same
//     number of half-edges; inside a block the rows keep ascending order.  slot_row records which row sits in which
//     LDS slot; the kernel gathers x_v and scatters the gradient rows through it.
// Host code: one pass over the row pointers (copied to the host: 4 bytes per row) -- the ring layout is built once
// per edge list.
struct RowPlan {
  bool mapped = false, permuted = false;
  int nrb = 0;
  int64_t H_ring = 0;
  std::vector<int32_t> row_slot;   // [nloc]
  std::vector<int32_t> slot_row;   // [nrb * R] (permuted only)
  std::vector<int32_t> slot_cum;   // [nrb * R + 1]
  std::vector<int32_t> hub_rows, hub_first, hub_seg;  // hub_seg: [segments][4] = hub index, first position, end position, 0
  int64_t hub_half_edges = 0;
  int threshold = 0;
};

static int plan_rows(const mde_plan* plan, const RingSizes& z, hipStream_t st, RowPlan* rp) {
  const int64_t nloc = plan->row_hi - plan->row_lo;
  const int R = z.R;
  rp->nrb = z.NRB;
  rp->H_ring = plan->H;
  const char* e_hub = getenv("MDE_RING_HUB");          // 0: never peel; N > 0: peel rows with more than N half-edges
  const char* e_perm = getenv("MDE_RING_PERMUTE");     // 0 / 1: never / always deal the rows to the blocks
  const int hub_env = e_hub ? atoi(e_hub) : -1, perm_env = e_perm ? atoi(e_perm) : -1;
  if (hub_env == 0 && perm_env == 0) return MDE_OK;
  std::vector<int32_t> rowptr((size_t)nloc + 1);
  MDE_HIP(hipMemcpyAsync(rowptr.data(), plan->rowptr, rowptr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  MDE_HIP(hipStreamSynchronize(st));
  const int64_t H = plan->H;
  int T = (int)std::max<int64_t>(256, H / ((int64_t)z.NRB * MDE_RING_NCW * 128));
  if (hub_env > 0) T = hub_env;
  rp->threshold = T;
  // hub rows
  int64_t hub_he = 0;
  int32_t max_deg = 0;
  if (hub_env != 0) {
    for (int64_t r = 0; r < nloc; ++r) {
      const int32_t deg = rowptr[r + 1] - rowptr[r];
      max_deg = std::max(max_deg, deg);
      if (deg > T) {
        rp->hub_rows.push_back((int32_t)r);
        hub_he += deg;
      }
    }
    if (2 * hub_he > H) {  // (most of the graph in "hub" rows: a dense problem, not a hub -- the ring takes it whole)
      rp->hub_rows.clear();
      hub_he = 0;
    }
  }
  const bool peel = !rp->hub_rows.empty();
  // work per block of R consecutive rows (peeled rows count nothing)
  bool permute = perm_env == 1;
  if (perm_env < 0) {
    int64_t heaviest = 0;
    size_t hi = 0;
    for (int b = 0; b < z.NRB; ++b) {
      const int64_t r0 = (int64_t)b * R, r1 = std::min<int64_t>(nloc, r0 + R);
      int64_t w = rowptr[r1] - rowptr[r0];
      while (hi < rp->hub_rows.size() && rp->hub_rows[hi] < r1) {
        w -= rowptr[rp->hub_rows[hi] + 1] - rowptr[rp->hub_rows[hi]];
        ++hi;
      }
      heaviest = std::max(heaviest, w);
    }
    permute = z.NRB > 1 && (double)heaviest > 1.08 * (double)(H - hub_he) / (double)z.NRB;
  }
  if (!peel && !permute) return MDE_OK;
  rp->mapped = true;
  rp->permuted = permute;
  rp->hub_half_edges = hub_he;
  rp->H_ring = H - hub_he;
  rp->row_slot.assign((size_t)nloc, -1);
  std::vector<uint8_t> is_hub((size_t)nloc, 0);
  for (int32_t r : rp->hub_rows) is_hub[(size_t)r] = 1;
  if (!permute) {
    // slots = rows; the peeled ones hold nothing
    rp->slot_cum.assign((size_t)z.NRB * R + 1, 0);
    int64_t cum = 0;
    for (int64_t r = 0; r < (int64_t)z.NRB * R; ++r) {
      rp->slot_cum[(size_t)r] = (int32_t)cum;
      if (r < nloc && !is_hub[(size_t)r]) {
        rp->row_slot[(size_t)r] = (int32_t)r;
        cum += rowptr[r + 1] - rowptr[r];
      }
    }
    rp->slot_cum[(size_t)z.NRB * R] = (int32_t)cum;
  } else {
    // counting sort by degree, heaviest first; boustrophedon deal; rows ascending inside a block
    const int64_t nring = nloc - (int64_t)rp->hub_rows.size();
    const int nrb = (int)std::max<int64_t>(1, (nring + R - 1) / R);
    rp->nrb = nrb;
    int32_t dmax = 0;
    for (int64_t r = 0; r < nloc; ++r)
      if (!is_hub[(size_t)r]) dmax = std::max(dmax, rowptr[r + 1] - rowptr[r]);
    std::vector<int64_t> start((size_t)dmax + 2, 0);
    for (int64_t r = 0; r < nloc; ++r)
      if (!is_hub[(size_t)r]) ++start[(size_t)(dmax - (rowptr[r + 1] - rowptr[r])) + 1];
    for (size_t k = 1; k < start.size(); ++k) start[k] += start[k - 1];
    std::vector<int32_t> block((size_t)nloc, -1);
    for (int64_t r = 0; r < nloc; ++r) {
      if (is_hub[(size_t)r]) continue;
      const int64_t rank = start[(size_t)(dmax - (rowptr[r + 1] - rowptr[r]))]++;
      const int64_t pos = rank % (2 * (int64_t)nrb);
      block[(size_t)r] = (int32_t)(pos < nrb ? pos : 2 * (int64_t)nrb - 1 - pos);
    }
    std::vector<int32_t> fill((size_t)nrb, 0);
    rp->slot_row.assign((size_t)nrb * R, -1);
    std::vector<int32_t> slot_deg((size_t)nrb * R, 0);
    for (int64_t r = 0; r < nloc; ++r) {
      const int b = block[(size_t)r];
      if (b < 0) continue;
      const int sidx = fill[(size_t)b]++;
      if (sidx >= R) {
        mde_set_error("ring layout: a row block overflowed while the rows were dealt (internal error)");
        return MDE_E_INVALID;
      }
      const size_t sl = (size_t)b * R + (size_t)sidx;
      rp->slot_row[sl] = (int32_t)r;
      rp->row_slot[(size_t)r] = (int32_t)sl;
      slot_deg[sl] = rowptr[r + 1] - rowptr[r];
    }
    rp->slot_cum.assign((size_t)nrb * R + 1, 0);
    int64_t cum = 0;
    for (size_t t = 0; t < (size_t)nrb * R; ++t) {
      rp->slot_cum[t] = (int32_t)cum;
      cum += slot_deg[t];
    }
    rp->slot_cum[(size_t)nrb * R] = (int32_t)cum;
  }
  // segments of the hub rows
  rp->hub_first.reserve(rp->hub_rows.size() + 1);
  for (size_t i = 0; i < rp->hub_rows.size(); ++i) {
    rp->hub_first.push_back((int32_t)(rp->hub_seg.size() / 4));
    const int32_t r = rp->hub_rows[i];
    for (int64_t b = rowptr[r]; b < rowptr[r + 1]; b += MDE_HUB_SEG) {
      rp->hub_seg.push_back((int32_t)i);
      rp->hub_seg.push_back((int32_t)b);
      rp->hub_seg.push_back((int32_t)std::min<int64_t>(rowptr[r + 1], b + MDE_HUB_SEG));
      rp->hub_seg.push_back(0);
    }
  }
  rp->hub_first.push_back((int32_t)(rp->hub_seg.size() / 4));
  if (getenv("MDE_RING_STATS"))
    fprintf(stderr, "[mde ring] rows: threshold %d half-edges, max degree %d, %zu hub rows with %lld half-edges (%.2f%%) peeled into %zu "
            "segments; blocks %s (%d)\n", T, max_deg, rp->hub_rows.size(), (long long)hub_he, 100.0 * (double)hub_he / (double)std::max<int64_t>(H, 1),
            rp->hub_seg.size() / 4, permute ? "DEALT by degree" : "of consecutive rows", rp->nrb);
  return MDE_OK;
}

// A stream that runs out of its region marks itself with 2^25 iterations (k_ring_schedule), which the callers
// recognise in the SUM of all streams' counts -- 2^25 x 64 entries are past the 32-bit position limit.  The sum is a
// 32-bit scan: 128 marked streams add up to 2^32 = 0 and the overflow went unseen (round 6: the planted-cluster graph
// at n = 1M with MDE_RING_ASSIGN=1 marked exactly 1024 streams; the regions were then used as if they had sufficed and
// the pack kernel faulted).  The counts themselves are looked at: synchronises, and sets *total to -1 when any
// stream is marked.
static hipError_t any_stream_bailed(const int32_t* iters, int nseg, hipStream_t st, int32_t* total) {
  std::vector<int32_t> h((size_t)nseg);
  hipError_t e = hipMemcpyAsync(h.data(), iters, h.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) return e;
  for (int i = 0; i < nseg; ++i)
    if (h[(size_t)i] >= (1 << 25)) {
      *total = -1;
      break;
    }
  return hipSuccess;
}

static int build_ring(mde_plan* plan, int d, hipStream_t st) {
  RingSizes z;
  if (!choose_sizes(plan, d, &z)) return 0;
  const int64_t nloc = plan->row_hi - plan->row_lo;
  const int64_t H = plan->H;
  // which rows are peeled off to the hub kernel, and whether the rest are dealt to the row blocks (round 6)
  RowPlan rp;
  {
    const int rc = plan_rows(plan, z, st, &rp);
    if (rc != MDE_OK) return rc;
    if (rp.mapped && rp.nrb != z.NRB) {
      z.NRB = rp.nrb;
      z.Q = choose_col_groups(plan, z.NRB, z.qmax, z.NC);
      if ((int64_t)z.NRB * z.Q > MDE_MAX_PARTIALS || bits_for_u64((uint64_t)((int64_t)z.NRB * z.Q * MDE_RING_NCW)) + z.JB > 32) return 0;
    }
  }
  const int64_t H_ring = rp.H_ring;  // half-edges the ring streams hold (the rest belongs to peeled rows)
  if (H_ring <= 0) return 0;
  const int nseg = z.NRB * z.Q * MDE_RING_NCW;
  const uint32_t JM = (1u << z.JB) - 1u;
  uint32_t *keys = nullptr, *vals = nullptr, *keys2 = nullptr, *vals2 = nullptr, *packed = nullptr, *hdr = nullptr, *wmap = nullptr;
  int32_t *hrow = nullptr, *wrows = nullptr, *seg = nullptr, *iters = nullptr, *iter_base = nullptr;
  int32_t *it_ent = nullptr, *it_cnt = nullptr, *it_m = nullptr, *peid = nullptr;
  int32_t *row_slot = nullptr, *slot_cum = nullptr, *slot_row = nullptr, *hub_rows = nullptr, *hub_seg = nullptr, *hub_first = nullptr;
  double* hub_partial = nullptr;
  float* partial = nullptr;
  void* tmp = nullptr;
  hipError_t e = hipSuccess;
  auto release = [&](bool all) {
    void* scratch[] = {keys, vals, keys2, vals2, hrow, wrows, wmap, seg, iters, it_ent, it_cnt, it_m, tmp, row_slot, slot_cum};
    for (void* p : scratch)
      if (p) (void)hipFree(p);
    if (all) {
      void* outs[] = {packed, hdr, peid, iter_base, partial, slot_row, hub_rows, hub_seg, hub_first, hub_partial};
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
  // (MDE_RING_STATS: where the build's time goes, host side included)
  const bool stats_on = getenv("MDE_RING_STATS") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto tick = [&](const char* what) {
    if (!stats_on) return;
    (void)hipStreamSynchronize(st);
    const auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[mde ring build] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t_last).count());
    t_last = t1;
  };
  const size_t hb = (size_t)H * sizeof(uint32_t);
  RB(hipMalloc(&keys, hb));
  RB(hipMalloc(&vals, hb));
  RB(hipMalloc(&keys2, hb));
  RB(hipMalloc(&vals2, hb));
  RB(hipMalloc(&hrow, hb));
  RB(hipMalloc(&wrows, ((size_t)nseg + 1) * sizeof(int32_t)));
  RB(hipMalloc(&wmap, (size_t)z.NRB * z.Q * z.R * sizeof(uint32_t)));
  RB(hipMalloc(&seg, ((size_t)nseg + 1) * sizeof(int32_t)));
  RB(hipMalloc(&iters, ((size_t)nseg + 1) * sizeof(int32_t)));
  RB(hipMalloc(&iter_base, ((size_t)nseg + 1) * sizeof(int32_t)));
  tick("scratch allocations");
  // MDE_RING_ASSIGN=1: the greedy, sweep-balanced row -> wave map (14.6 ms at config 4, no faster: see k_ring_assign);
  // default: contiguous row ranges of equal half-edge count
  const int contiguous = getenv("MDE_RING_ASSIGN") ? atoi(getenv("MDE_RING_ASSIGN")) == 0 : 1;
  auto upload = [&](int32_t** dst, const std::vector<int32_t>& src) -> hipError_t {
    hipError_t er = hipMalloc(dst, std::max<size_t>(src.size(), 1) * sizeof(int32_t));
    if (er == hipSuccess && !src.empty()) er = hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
    return er;
  };
  if (rp.mapped) {
    // (the host vectors stay alive until the build's last synchronisation)
    RB(upload(&row_slot, rp.row_slot));
    RB(upload(&slot_cum, rp.slot_cum));
    if (rp.permuted) RB(upload(&slot_row, rp.slot_row));
    if (!rp.hub_rows.empty()) {
      RB(upload(&hub_rows, rp.hub_rows));
      RB(upload(&hub_seg, rp.hub_seg));
      RB(upload(&hub_first, rp.hub_first));
      RB(hipMalloc(&hub_partial, (rp.hub_seg.size() / 4) * 8 * sizeof(double)));
    }
    hipLaunchKernelGGL(k_ring_assign_slots, dim3(z.NRB * z.Q), dim3(64), 0, st, z.R, z.Q, slot_cum, wmap, wrows);
    RB(hipGetLastError());
    tick("slot -> wave assignment");
    hipLaunchKernelGGL(k_ring_hrow, dim3(mde_grid(nloc * 16, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, (int)nloc, plan->rowptr, hrow);
    RB(hipGetLastError());
    hipLaunchKernelGGL(k_ring_keys_slots, dim3(mde_grid(H, MDE_BLOCK, 8192)), dim3(MDE_BLOCK), 0, st, H, plan->nbr, row_slot, wmap, z.R,
                       z.Q, z.NC, z.C, z.JB, (uint32_t)nseg, keys, vals, hrow);
    RB(hipGetLastError());
  } else {
    hipLaunchKernelGGL(k_ring_assign, dim3(z.NRB * z.Q), dim3(64), 0, st, (int)nloc, z.R, z.Q, z.NC, z.C, contiguous, plan->rowptr,
                       plan->nbr, wmap, wrows);
    RB(hipGetLastError());
    tick("row -> wave assignment");
    hipLaunchKernelGGL(k_ring_keys, dim3(mde_grid(nloc * 16, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, (int)nloc,
                       plan->rowptr, plan->nbr, wmap, z.R, z.Q, z.NC, z.C, z.JB, keys, vals, hrow);
    RB(hipGetLastError());
  }
  tick("keys kernel");
  size_t tmp_bytes = 0, scan_bytes = 0;
  const int end_bit = std::min(32, bits_for_u64((uint64_t)nseg) + z.JB);
  RB(mde_sort_pairs_u32(nullptr, tmp_bytes, keys, keys2, vals, vals2, (int)H, 0, end_bit, st));
  RB(mde_exclusive_sum_i32(nullptr, scan_bytes, iters, iter_base, nseg + 1, st));
  if (scan_bytes > tmp_bytes) tmp_bytes = scan_bytes;
  RB(mde_sort_pairs_u32(nullptr, scan_bytes, keys, keys2, vals, vals2, (int)H, 0,
                                        bits_for_u64((uint64_t)plan->n), st));
  if (scan_bytes > tmp_bytes) tmp_bytes = scan_bytes;
  RB(hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16));
  tick("sort size queries + temp allocation");
  // Dense graphs (many entries per row and chunk): in CSR order the entries of ONE row follow each
  // other inside a chunk, and an iteration -- distinct rows -- would find a handful of rows in its
  // look-ahead.  Order the entries by column first (stable): inside a (stream, chunk) they then come
  // column by column, and the rows adjacent to one column are distinct.
  const bool by_column = getenv("MDE_RING_BYCOL") ? atoi(getenv("MDE_RING_BYCOL")) != 0
                                                  : (double)H_ring > 0.5 * (double)nloc * (double)z.NC;
  if (by_column) {
    // keys2 = columns, sorted with vals -> (keys, vals2); then keys2 = stream keys in that order
    RB(hipMemcpyAsync(keys2, plan->nbr, hb, hipMemcpyDeviceToDevice, st));
    uint32_t* skey = reinterpret_cast<uint32_t*>(hrow);  // (hrow is rebuilt below)
    RB(hipMemcpyAsync(skey, keys, hb, hipMemcpyDeviceToDevice, st));
    RB(mde_sort_pairs_u32(tmp, tmp_bytes, keys2, keys, vals, vals2, (int)H, 0,
                                          bits_for_u64((uint64_t)plan->n), st));
    hipLaunchKernelGGL(k_ring_gather_u32, dim3(mde_grid(H, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, H, vals2, skey, keys);
    RB(hipGetLastError());
    RB(hipMemcpyAsync(vals, vals2, hb, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_ring_hrow, dim3(mde_grid(nloc * 16, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, (int)nloc,
                       plan->rowptr, hrow);
    RB(hipGetLastError());
    if (rp.mapped) {
      hipLaunchKernelGGL(k_ring_rows_to_slots, dim3(mde_grid(H, MDE_BLOCK, 8192)), dim3(MDE_BLOCK), 0, st, H, row_slot, hrow);
      RB(hipGetLastError());
    }
  }
  RB(mde_sort_pairs_u32(tmp, tmp_bytes, keys, keys2, vals, vals2, (int)H, 0, end_bit, st));
  tick("radix sort");
  hipLaunchKernelGGL(k_ring_seg, dim3(mde_grid(H + 1, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, H, (uint32_t)nseg,
                     z.JB, keys2, seg);
  RB(hipGetLastError());
  RB(hipMemsetAsync(iters, 0, ((size_t)nseg + 1) * sizeof(int32_t), st));
  tick("keys, sorts, segments");
  // bank-class cap of the scheduler and the lane placement (environment: design ablations only)
  const int cap_default = d == 4 ? 8 : 4;
  const int cap = getenv("MDE_RING_CAP") ? std::max(1, atoi(getenv("MDE_RING_CAP"))) : cap_default;
  const int place = getenv("MDE_RING_PLACE") ? atoi(getenv("MDE_RING_PLACE")) : 1;
  const int span = z.span;
  // (keys is free from here on: it holds the scheduler's meta words)
  uint32_t* meta = keys;
  // (the sorted positions [0, H_ring) are the ring's entries; the entries of peeled rows sort behind them)
  hipLaunchKernelGGL(k_ring_meta, dim3(mde_grid(H_ring, MDE_BLOCK, 8192)), dim3(MDE_BLOCK), 0, st, H_ring, keys2, vals2, hrow, plan->nbr,
                     wmap, z.JB, z.R, z.Q, d, meta);
  RB(hipGetLastError());
  // rows of the largest wave range (the scheduler keeps one LDS word per row of its wave)
  int flag_rows = 64;
  {
    std::vector<int32_t> hwr((size_t)nseg);
    RB(hipMemcpyAsync(hwr.data(), wrows, hwr.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    RB(hipStreamSynchronize(st));
    for (int i = 0; i < nseg; ++i) flag_rows = std::max(flag_rows, hwr[(size_t)i]);
    flag_rows = (flag_rows + 63) / 64 * 64;
  }
  const size_t flag_bytes = (size_t)flag_rows * sizeof(int);
  // Single pass: every stream schedules into a region of caps[i] iterations (2 x what its entries or its
  // chunk windows need at least) and reports how many it used; k_ring_pack compacts.  When a stream
  // overflows its region (a hub row) the build takes the exact two-pass route instead: count (with the
  // give-up limit of auto mode), allocate, fill.  (Round 3 always counted first: the scheduler ran twice.)
  int32_t *caps = nullptr, *cap_base = nullptr;
  RB(hipMalloc(&caps, ((size_t)nseg + 1) * sizeof(int32_t)));
  RB(hipMalloc(&cap_base, ((size_t)nseg + 1) * sizeof(int32_t)));
  auto drop_caps = [&]() {
    if (caps) (void)hipFree(caps);
    if (cap_base) (void)hipFree(cap_base);
    caps = cap_base = nullptr;
  };
  int32_t total_iters = 0, total_cap = 0;
  // (regions of 2 x the minimum: the scratch they take is what the build's time goes into -- fresh device
  // memory costs ~10 ms per GB to map on first use, more than the kernels that fill it)
  hipLaunchKernelGGL(k_ring_caps, dim3((nseg + 1 + MDE_BLOCK - 1) / MDE_BLOCK), dim3(MDE_BLOCK), 0, st, nseg, seg, wrows, z.R, z.Q,
                     z.NC, span, 2, caps);
  e = hipGetLastError();
  if (e == hipSuccess) e = mde_exclusive_sum_i32(tmp, tmp_bytes, caps, cap_base, nseg + 1, st);
  if (e == hipSuccess) e = hipMemcpyAsync(&total_cap, cap_base + nseg, sizeof(int32_t), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  bool single = e == hipSuccess && total_cap > 0 && (int64_t)total_cap * 64 < ((int64_t)1 << 33);
  if (e != hipSuccess) {
    drop_caps();
    return fail(e, "ring layout: stream capacities");
  }
  if (single) {
    e = hipMalloc(&it_ent, (size_t)total_cap * 64 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&it_cnt, (size_t)total_cap * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&it_m, (size_t)total_cap * sizeof(int32_t));
    if (e != hipSuccess) {
      (void)hipGetLastError();
      single = false;  // (not enough memory for the generous regions: count first)
      void* ps[] = {it_ent, it_cnt, it_m};
      for (void* q : ps)
        if (q) (void)hipFree(q);
      it_ent = it_cnt = it_m = nullptr;
    }
  }
  if (single) {
    hipLaunchKernelGGL(k_ring_schedule<true>, dim3(nseg), dim3(64), flag_bytes, st, nseg, seg, wrows, keys2, meta, JM, span, z.R,
                       z.Q, z.NC, d, cap, 4, iters, cap_base, it_ent, it_cnt, it_m, flag_rows, caps);
    e = hipGetLastError();
    if (e == hipSuccess) e = mde_exclusive_sum_i32(tmp, tmp_bytes, iters, iter_base, nseg + 1, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&total_iters, iter_base + nseg, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = any_stream_bailed(iters, nseg, st, &total_iters);
    if (e != hipSuccess) {
      drop_caps();
      return fail(e, "ring layout: single-pass schedule");
    }
    tick("meta + single scheduling pass");
    if (total_iters <= 0 || (int64_t)total_iters * 64 >= ((int64_t)1 << 31) - 64) {
      // a stream overflowed its region (a row with many times the average degree; or the layout is too
      // large for 32-bit positions): count first -- that pass decides whether the layout is given up
      single = false;
      void* ps[] = {it_ent, it_cnt, it_m};
      for (void* q : ps)
        if (q) (void)hipFree(q);
      it_ent = it_cnt = it_m = nullptr;
    }
  }
  if (!single) {
    hipLaunchKernelGGL(k_ring_schedule<false>, dim3(nseg), dim3(64), flag_bytes, st, nseg, seg, wrows, keys2, meta, JM, span,
                       z.R, z.Q, z.NC, d, cap, panel_mode() == 1 ? 0 : 4, iters, nullptr, nullptr, nullptr, nullptr, flag_rows,
                       nullptr);
    e = hipGetLastError();
    if (e == hipSuccess) e = mde_exclusive_sum_i32(tmp, tmp_bytes, iters, iter_base, nseg + 1, st);
    if (e == hipSuccess) e = hipMemcpyAsync(&total_iters, iter_base + nseg, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = any_stream_bailed(iters, nseg, st, &total_iters);
    if (e != hipSuccess) {
      drop_caps();
      return fail(e, "ring layout: counting pass");
    }
    tick("meta + counting pass");
  }
  const int64_t Hp = (int64_t)total_iters * 64;  // padded half-edge count
  if (total_iters <= 0 || Hp >= ((int64_t)1 << 31) - 64) {
    drop_caps();
    release(true);
    return 0;  // too large for 32-bit positions: the caller keeps the CSR layout
  }
  if (panel_mode() != 1) {
    // Auto mode, with the iterations counted: is the layout still worth it?  (Rounds 3-5 gave up beyond 35 % padding --
    // hub rows, which are peeled now, and sparse streams, whose half-filled iterations still beat the CSR kernel's
    // gathers several times over.)  The streams' mean length is priced; a stream far beyond it would be a hub the
    // threshold let through, and 4 x padding says as much.
    // ... and how much of that runs side by side: the slowest workgroup's critical path along the sweep over the
    // mean stream (k_ring_sweep_balance; ~1.1 on a uniform graph, which the 0.21 us per iteration already contain)
    double serial = 1.0;
    {
      const int nwg = z.NRB * z.Q, GW = 4, ngroups = (z.NC + GW - 1) / GW;
      unsigned long long* bal = nullptr;
      e = hipMalloc(&bal, (size_t)2 * nwg * sizeof(unsigned long long));
      if (e == hipSuccess) e = hipMemsetAsync(bal, 0, (size_t)2 * nwg * sizeof(unsigned long long), st);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_ring_sweep_balance, dim3((unsigned)(((int64_t)nwg * ngroups + MDE_BLOCK - 1) / MDE_BLOCK)), dim3(MDE_BLOCK),
                           0, st, nwg, ngroups, GW, z.JB, keys2, seg, bal, bal + nwg);
        e = hipGetLastError();
      }
      std::vector<unsigned long long> hb((size_t)2 * nwg);
      if (e == hipSuccess) e = hipMemcpyAsync(hb.data(), bal, hb.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
      if (e == hipSuccess) e = hipStreamSynchronize(st);
      if (bal) (void)hipFree(bal);
      if (e != hipSuccess) {
        drop_caps();
        return fail(e, "ring layout: sweep balance");
      }
      unsigned long long cmax = 0, tsum = 0;
      for (int i = 0; i < nwg; ++i) {
        cmax = std::max(cmax, hb[(size_t)i]);
        tsum += hb[(size_t)nwg + i];
      }
      if (tsum > 0) serial = (double)cmax / ((double)tsum / (double)nseg);
    }
    const double t_ring = std::max(1.0, serial / 1.15) *
                              ring_time_us((double)total_iters / (double)nseg, z.NC, z.Q, z.NRB * z.Q) +
                          (double)rp.hub_half_edges * 1.3e-5;
    if (getenv("MDE_RING_STATS"))
      fprintf(stderr, "[mde ring] sweep balance: critical path of the slowest workgroup / mean stream = %.2f\n", serial);
    // (mid regime: what the dealt layout and the hub launches add -- the slot map's gathers and a loss term on every
    // entry ~30 % per iteration, two more launches ~4 us: preferential attachment at n = 100k, out-degree 20 ran
    // 0.042 ms on the ring against 0.031 on the CSR kernel, tools/r6_midsize_skew.sh)
    const double mid_ring = plan->ring.cost_scale * (rp.permuted ? 1.3 : 1.0) * t_ring + MDE_RING_FIXED_US +
                            (rp.hub_half_edges > 0 ? 4.0 : 0.0);
    const bool mid_bad = mid_regime(plan, d) && mid_ring > 0.85 * (csr_time_us(plan, d) + MDE_CSR_FIXED_US);
    if ((double)Hp > 4.0 * (double)H_ring || t_ring > 0.9 * csr_time_us(plan, d) || mid_bad) {
      drop_caps();
      release(true);
      return 0;
    }
  }
  e = hipSuccess;
  if (!single) {
    e = hipMalloc(&it_ent, (size_t)total_iters * 64 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&it_cnt, (size_t)total_iters * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&it_m, (size_t)total_iters * sizeof(int32_t));
  }
  if (e == hipSuccess) e = hipMalloc(&packed, (size_t)Hp * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc(&peid, (size_t)Hp * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc(&hdr, (size_t)total_iters * MDE_RING_HW * sizeof(uint32_t));
  // (+ two words per row block behind the rows: ticket and flag of the in-launch sum of Q = 2 groups, zero between launches)
  const size_t partial_floats = (size_t)z.Q * (size_t)nloc * (size_t)d;
  if (e == hipSuccess && z.Q > 1) e = hipMalloc(&partial, sizeof(float) * (partial_floats + 2 * (size_t)z.NRB));
  if (e == hipSuccess && z.Q > 1) e = hipMemsetAsync(partial + partial_floats, 0, sizeof(float) * 2 * (size_t)z.NRB, st);
  if (e != hipSuccess) {
    drop_caps();
    return fail(e, "ring layout: output allocations");
  }
  tick("output allocations");
  if (!single) {
    hipLaunchKernelGGL(k_ring_schedule<true>, dim3(nseg), dim3(64), flag_bytes, st, nseg, seg, wrows, keys2, meta, JM, span, z.R,
                       z.Q, z.NC, d, cap, 0, nullptr, iter_base, it_ent, it_cnt, it_m, flag_rows, nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_ring_pack, dim3(mde_grid((int64_t)total_iters * 64, MDE_BLOCK, 8192)), dim3(MDE_BLOCK), 0, st,
                       (int64_t)total_iters, nseg, iter_base, single ? cap_base : nullptr, it_ent, it_cnt, it_m, keys2, vals2, hrow,
                       plan->nbr, plan->eid, z.R, z.Q, z.C, z.S, z.ring_off, z.JB, d, (int)plan->row_lo, place, rp.permuted ? 1 : 0,
                       packed, peid, hdr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // (cap_base is read by the pack kernel)
  drop_caps();
  if (e != hipSuccess) return fail(e, "ring layout: fill / pack");
  hipLaunchKernelGGL(k_ring_block_class, dim3((unsigned)((total_iters / 4 + MDE_BLOCK - 1) / MDE_BLOCK)), dim3(MDE_BLOCK), 0, st,
                     (int64_t)(total_iters / 4), hdr);
  RB(hipGetLastError());
  RB(hipStreamSynchronize(st));
  if (getenv("MDE_RING_STATS")) {
    unsigned long long* dstat = reinterpret_cast<unsigned long long*>(iters);  // scratch, >= 3 words
    unsigned long long hstat[3] = {0, 0, 0};
    RB(hipMemsetAsync(dstat, 0, sizeof(hstat), st));
    hipLaunchKernelGGL(k_ring_stats, dim3(256), dim3(MDE_BLOCK), 0, st, (int64_t)total_iters, hdr, dstat);
    RB(hipMemcpyAsync(hstat, dstat, sizeof(hstat), hipMemcpyDeviceToHost, st));
    RB(hipStreamSynchronize(st));
    {
      // how even are the consumer waves' streams?  (the slowest wave sets the kernel's duration)
      std::vector<int32_t> hb((size_t)nseg + 1);
      RB(hipMemcpy(hb.data(), iter_base, hb.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
      int mx = 0, mn = 1 << 30, wg_mx = 0;
      for (int i = 0; i < nseg; ++i) {
        mx = std::max(mx, hb[i + 1] - hb[i]);
        mn = std::min(mn, hb[i + 1] - hb[i]);
      }
      for (int g = 0; g + MDE_RING_NCW <= nseg; g += MDE_RING_NCW) wg_mx = std::max(wg_mx, hb[g + MDE_RING_NCW] - hb[g]);
      fprintf(stderr, "[mde ring] iterations per consumer wave: min %d mean %.1f max %d; per workgroup: mean %.1f max %d\n", mn,
              (double)total_iters / nseg, mx, (double)total_iters * MDE_RING_NCW / nseg, wg_mx);
    }
    {
      // self-check of the hand-shake contract: per pair of iterations, newest chunk - published chunk <= window
      std::vector<uint32_t> hh((size_t)total_iters * MDE_RING_HW);
      RB(hipMemcpy(hh.data(), hdr, hh.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
      long long bad = 0;
      int worst = 0;
      for (int64_t it = 0; it + 1 < total_iters; it += 2) {
        const int gap = (int)hh[it * MDE_RING_HW + 1] - (int)hh[it * MDE_RING_HW];
        if (gap > span || hh[(it + 1) * MDE_RING_HW] < hh[it * MDE_RING_HW]) {
          if (bad < 8)
            fprintf(stderr, "[mde ring] pair at iteration %lld: m %u need %u | next m %u need %u\n", (long long)it, hh[it * MDE_RING_HW],
                    hh[it * MDE_RING_HW + 1], hh[(it + 1) * MDE_RING_HW], hh[(it + 1) * MDE_RING_HW + 1]);
          ++bad;
        }
        worst = std::max(worst, gap);
      }
      fprintf(stderr, "[mde ring] hand-shake contract: %lld of %d pairs exceed the window (worst gap %d chunks)\n", bad, total_iters / 2, worst);
      // how far apart are the consumer waves of a workgroup along the column sweep?  At the same fraction of
      // their streams: newest minus oldest chunk over the waves (a ring slot is free when ALL of them are past it)
      {
        std::vector<int32_t> hb2((size_t)nseg + 1);
        RB(hipMemcpy(hb2.data(), iter_base, hb2.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        std::vector<int> sp;
        for (int g0 = 0; g0 + MDE_RING_NCW <= nseg; g0 += MDE_RING_NCW) {
          int len = 1 << 30;
          for (int w = 0; w < MDE_RING_NCW; ++w) len = std::min(len, hb2[g0 + w + 1] - hb2[g0 + w]);
          if (len < 16) continue;
          for (int k = 0; k < len; k += 2) {
            int lo = 1 << 30, hi2 = 0;
            for (int w = 0; w < MDE_RING_NCW; ++w) {
              const int nit_w = hb2[g0 + w + 1] - hb2[g0 + w];
              const int64_t it = hb2[g0 + w] + (((int64_t)k * nit_w / len) & ~(int64_t)1);
              const int m = (int)hh[it * MDE_RING_HW];
              lo = std::min(lo, m);
              hi2 = std::max(hi2, m);
            }
            sp.push_back(hi2 - lo);
          }
        }
        if (!sp.empty()) {
          std::sort(sp.begin(), sp.end());
          double mean = 0;
          for (int v : sp) mean += v;
          fprintf(stderr, "[mde ring] spread of a workgroup's consumer waves along the column sweep (chunks): mean %.2f median %d p90 %d max %d\n",
                  mean / sp.size(), sp[sp.size() / 2], sp[sp.size() * 9 / 10], sp.back());
        }
      }
    }
    fprintf(stderr, "[mde ring] d=%d R=%d NRB=%d Q=%d chunks=%d x %d cols, window %d, cap %d, placement %d: %d iterations for %lld half-edges "
            "(%.1f%% padding), %.1f%% with padding lanes; loss terms: %.1f%% of the iterations add all, %.2f%% test per lane\n",
            d, z.R, z.NRB, z.Q, z.NC, z.C, span, cap, place, total_iters, (long long)H_ring,
            100.0 * ((double)Hp - (double)H_ring) / (double)H_ring, 100.0 * hstat[0] / total_iters, 100.0 * hstat[1] / total_iters,
            100.0 * hstat[2] / total_iters);
  }
#undef RB
  tick("fill pass, pack, classes");
  release(false);
  tick("scratch release");
  mde_ring_layout& L = plan->ring;
  ring_free(L);
  L.d = d;
  L.rows_per_block = z.R;
  L.n_row_blocks = z.NRB;
  L.col_groups = z.Q;
  L.chunk_cols = z.C;
  L.n_chunks = z.NC;
  L.ring_off = z.ring_off;
  L.slots = z.S;
  L.n_iters = total_iters;
  L.H = Hp;
  L.packed = packed;
  L.eid = peid;
  L.hdr = hdr;
  L.wave_iter = iter_base;
  L.partial = partial;
  L.slot_row = slot_row;
  L.count_all = rp.permuted ? 1 : 0;
  L.n_hub_rows = (int)rp.hub_rows.size();
  L.n_hub_segs = (int)(rp.hub_seg.size() / 4);
  L.hub_half_edges = rp.hub_half_edges;
  L.hub_rows = hub_rows;
  L.hub_seg = hub_seg;
  L.hub_first = hub_first;
  L.hub_partial = hub_partial;
  return 1;
}

// layout the fused kernel will use for dimension d: 0 = CSR, 1 = LDS ring (built on first
// request).  Negative: error.
extern "C" int mde_plan_layout(mde_plan* plan, int32_t d, void* stream) {
  if (!plan || d <= 0) return MDE_E_INVALID;
  if (plan->ring.packed && plan->ring.d == d) return 1;
  RingSizes z;
  if (!choose_sizes(plan, d, &z)) return 0;
  if (plan->ring.rejected_d == d && panel_mode() != 1) return 0;
  const int rc = build_ring(plan, d, mde_stream(stream));
  if (rc == 0) plan->ring.rejected_d = d;
  return rc;
}

__global__ __launch_bounds__(MDE_BLOCK) void k_expand_ring(int64_t H, const int32_t* __restrict__ eid,
                                                           const float* __restrict__ in,
                                                           float* __restrict__ out) {
  for (int64_t q = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; q < H;
       q += (int64_t)gridDim.x * MDE_BLOCK)
    out[q] = eid[q] >= 0 ? in[eid[q]] : 0.0f;  // padding entries: weight 0 (f = 0 for every penalty)
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
// can hold an index into a table of 8 values: the kernel then streams 4 bytes per half-edge
// instead of 8 (packed word + fp32 parameter) and looks the parameter up in LDS.  Entry 0 is the
// weight 0 of padding lanes, entries 1..7 the (at most 7) distinct values of the array.
#define MDE_CB_EMPTY 0xFFFFFFFFu  // (a NaN pattern: NaN parameters simply disable the codebook)

// distinct bit patterns of in[0..p): inserted into table[1..8) with compare-and-swap; *overflow is
// set when an 8th value (or the EMPTY pattern) shows up.  A value goes through the workgroup's own
// table first (LDS compare-and-swap) and only the thread that put it THERE carries it to the global
// one: <= 8 global atomics per workgroup.  (Round 4 until then: every thread met an empty table at
// its first element and went to the global one -- half a million compare-and-swaps on one address,
// 9.4 ms for a scan that reads 200 MB.)
__global__ __launch_bounds__(MDE_BLOCK) void k_codebook_scan(int64_t p, const float* __restrict__ in,
                                                             unsigned int* __restrict__ table,
                                                             int* __restrict__ overflow) {
  __shared__ unsigned int stb[MDE_RING_CB_VALUES];
  if (threadIdx.x < MDE_RING_CB_VALUES) stb[threadIdx.x] = MDE_CB_EMPTY;
  __syncthreads();
  unsigned int last0 = MDE_CB_EMPTY, last1 = MDE_CB_EMPTY;
  const int64_t stride = (int64_t)gridDim.x * MDE_BLOCK;
  constexpr int U = 4;  // independent loads in flight per thread
  for (int64_t i0 = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i0 < p; i0 += U * stride) {
    unsigned int vv[U];
    bool in_range[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int64_t i = i0 + k * stride;
      in_range[k] = i < p;
      vv[k] = in_range[k] ? __float_as_uint(in[i]) : 0u;
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const unsigned int v = vv[k];
      if (!in_range[k]) continue;
      if (v == MDE_CB_EMPTY) {  // (the pattern that marks a free slot: a NaN, no codebook)
        *overflow = 1;
        return;
      }
      if (v == last0 || v == last1) continue;
      last1 = last0;
      last0 = v;
      bool placed = false, mine = false;
      for (int s = 0; s < MDE_RING_CB_VALUES && !placed; ++s) {
        const unsigned int old = atomicCAS(&stb[s], MDE_CB_EMPTY, v);
        mine = (old == MDE_CB_EMPTY);
        placed = mine || old == v;
      }
      if (placed && !mine) continue;  // some thread of this workgroup has carried it already
      placed = false;
      for (int s = 1; s < MDE_RING_CB_VALUES && !placed; ++s) {
        unsigned int cur = __hip_atomic_load(&table[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == MDE_CB_EMPTY) cur = atomicCAS(&table[s], MDE_CB_EMPTY, v);
        placed = (cur == MDE_CB_EMPTY || cur == v);
      }
      if (!placed) {
        *overflow = 1;
        return;
      }
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
      w |= idx;  // (padding entries keep index 0 = weight 0)
    }
    out[q] = w;
  }
}

// Try to put a per-edge parameter array into codebook form for layout 1.  On success
// (*n_values_host in 1..7 at d = 2, 1..3 at d = 3) out_half holds the H packed words with the value index
// in their low bits, followed by the 8-entry value table; pass it as mde_func.a0 with a0_scalar = 2.
// *n_values_host = 0: not applicable (another d, too many distinct values, NaNs) -- nothing is
// written and the caller uses mde_plan_expand_layout.  SYNC.
extern "C" int mde_plan_expand_codebook(const mde_plan* plan, const float* in_edge, float* out_half,
                                        int32_t* n_values_host, void* stream) {
  if (!plan || !in_edge || !out_half || !n_values_host) return MDE_E_INVALID;
  *n_values_host = 0;
  const mde_ring_layout& L = plan->ring;
  // (the index rides in the bits the row address leaves free: 3 at d = 2 -- up to 7 values --, 2 at d = 3 --
  // up to 3, enough for a neighbour graph's {1, 2, -1})
  if (!L.packed || (L.d != 2 && L.d != 3) || L.H == 0 || plan->p == 0) return MDE_OK;
  const int max_values = L.d == 2 ? 7 : 3;
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
  vals[0] = 0u;  // +0.0f: the padding lanes' weight
  for (int s = 1; s < MDE_RING_CB_VALUES; ++s)
    if (host_tb[s] != MDE_CB_EMPTY) vals[1 + nv++] = host_tb[s];
  if (nv == 0 || nv > max_values) return MDE_OK;
  // (finite values of ordinary size only: the kernel skips the NaN/Inf fix-up of f'/d for codebook
  // streams -- the functors that keep f'/d finite at d = 0 do so with a reciprocal of up to 1e30, which a
  // weight beyond 2e8 would carry to Inf (times x_v - x_u = 0: NaN where the reference has 0); weights
  // beyond 1e6 stream as fp32 next to the packed word instead, with the fix-up)
  for (int s = 1; s <= nv; ++s) {
    float fv;
    memcpy(&fv, &vals[s], sizeof(float));
    if (!std::isfinite(fv) || std::fabs(fv) > 1.0e6f) return MDE_OK;
  }
  for (int a = 2; a <= nv; ++a)
    for (int b = a; b > 1 && vals[b - 1] > vals[b]; --b) {
      const unsigned int t = vals[b];
      vals[b] = vals[b - 1];
      vals[b - 1] = t;
    }
  for (int s = nv + 1; s < MDE_RING_CB_VALUES; ++s) vals[s] = MDE_CB_EMPTY;
  MDE_HIP(hipMemcpyAsync(table, vals, sizeof(vals), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_codebook_pack, dim3(mde_grid(L.H, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, L.H, L.packed,
                     L.eid, in_edge, table, reinterpret_cast<uint32_t*>(out_half));
  MDE_LAUNCH_CHECK();
  MDE_HIP(hipStreamSynchronize(st));  // `vals` is a stack buffer
  *n_values_host = nv;
  return MDE_OK;
}

// ---------------------------------------------------------------- byte-index parameter streams
// Up to 255 distinct per-edge parameters (the hop counts of a distance-preserving problem on a graph are tens of
// distinct integers; quantised similarity weights): one index BYTE per entry beside the packed words -- 5 B per
// half-edge instead of 8 -- and a 256-entry value table that the kernel keeps in the 1 KB behind its chunk ring.
// The distinct bit patterns are collected in a hash table (per workgroup in LDS first, the winners carried to the
// global one), the host sorts them (canonical order: ascending bit patterns, entry 0 = the padding lanes' +0.0),
// and k_bytes_pack looks every entry's value up by bisection.
#define MDE_BX_SLOTS 1024  // hash slots (a power of two, >= 4 x the values a table holds)
__device__ __forceinline__ uint32_t bx_hash(uint32_t v) {
  v ^= v >> 16;
  v *= 0x7feb352du;
  v ^= v >> 15;
  v *= 0x846ca68bu;
  v ^= v >> 16;
  return v & (MDE_BX_SLOTS - 1);
}
// insert v into an open-addressing table; returns 1 when v was not there before, 0 when it was, -1 when the table is full
__device__ __forceinline__ int bx_insert(unsigned int* tb, uint32_t v) {
  uint32_t h = bx_hash(v);
  for (int probe = 0; probe < MDE_BX_SLOTS; ++probe) {
    const unsigned int old = atomicCAS(&tb[h], MDE_CB_EMPTY, v);
    if (old == MDE_CB_EMPTY) return 1;
    if (old == v) return 0;
    h = (h + 1) & (MDE_BX_SLOTS - 1);
  }
  return -1;
}
// state[0] = number of distinct values in the global table, state[1] = overflow flag
__global__ __launch_bounds__(MDE_BLOCK) void k_bytes_scan(int64_t p, const float* __restrict__ in,
                                                          unsigned int* __restrict__ table, int* __restrict__ state) {
  __shared__ unsigned int stb[MDE_BX_SLOTS];
  __shared__ int s_count;
  for (int i = threadIdx.x; i < MDE_BX_SLOTS; i += MDE_BLOCK) stb[i] = MDE_CB_EMPTY;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  unsigned int last0 = MDE_CB_EMPTY, last1 = MDE_CB_EMPTY;
  const int64_t stride = (int64_t)gridDim.x * MDE_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; i < p; i += stride) {
    // (continuous weights: some workgroup sees its 256th value within its first few elements -- everybody leaves)
    if (__hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const unsigned int v = __float_as_uint(in[i]);
    if (v == last0 || v == last1) continue;
    last1 = last0;
    last0 = v;
    if (v == MDE_CB_EMPTY) {  // (the pattern that marks a free slot: a NaN, no table)
      state[1] = 1;
      return;
    }
    const int fresh = bx_insert(stb, v);
    if (fresh == 0) continue;  // some thread of this workgroup has carried it already
    if (fresh < 0 || atomicAdd(&s_count, 1) >= MDE_RING_BX_VALUES - 1) {
      state[1] = 1;
      return;
    }
    const int g = bx_insert(table, v);
    if (g < 0 || (g == 1 && atomicAdd(&state[0], 1) >= MDE_RING_BX_VALUES - 1)) {
      state[1] = 1;
      return;
    }
  }
}

// out[q] = index of in[eid[q]] in the sorted table (entries 1..nv; padding entries keep index 0)
__global__ __launch_bounds__(MDE_BLOCK) void k_bytes_pack(int64_t H, const int32_t* __restrict__ eid,
                                                          const float* __restrict__ in,
                                                          const unsigned int* __restrict__ table, int nv,
                                                          uint8_t* __restrict__ out) {
  __shared__ unsigned int tb[MDE_RING_BX_VALUES];
  for (int i = threadIdx.x; i < MDE_RING_BX_VALUES; i += MDE_BLOCK) tb[i] = table[i];
  __syncthreads();
  for (int64_t q = (int64_t)blockIdx.x * MDE_BLOCK + threadIdx.x; q < H; q += (int64_t)gridDim.x * MDE_BLOCK) {
    uint32_t idx = 0;
    if (eid[q] >= 0) {
      const unsigned int v = __float_as_uint(in[eid[q]]);
      int lo = 1, hi = nv;  // tb[1..nv] ascending, v is one of them
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tb[mid] < v) lo = mid + 1; else hi = mid;
      }
      idx = (uint32_t)lo;
    }
    out[q] = (uint8_t)idx;
  }
}

// Try to put a per-edge parameter array into byte-index form for layout 1 (d = 2, 3).  On success
// (*n_values_host in 1..255) out_half holds the H index bytes (in the order of the packed words: one 32-bit word
// per lane and block of four iterations) followed by the 256-entry value table; pass it as mde_func.a0 with
// a0_scalar = 3.  *n_values_host = 0: not applicable (another d, more distinct values, NaNs, values beyond 1e6,
// no room behind the ring) -- nothing usable is written and the caller streams fp32.  SYNC.
extern "C" int mde_plan_expand_bytes(const mde_plan* plan, const float* in_edge, float* out_half,
                                     int32_t* n_values_host, void* stream) {
  if (!plan || !in_edge || !out_half || !n_values_host) return MDE_E_INVALID;
  *n_values_host = 0;
  const mde_ring_layout& L = plan->ring;
  if (!L.packed || (L.d != 2 && L.d != 3) || L.H < 4 * MDE_RING_BX_VALUES || plan->p == 0) return MDE_OK;
  if (MDE_RING_LDS_BYTES - (L.ring_off + L.slots * ring_chunk_bytes(L.d)) < 4 * MDE_RING_BX_VALUES) return MDE_OK;
  const char* e = getenv("MDE_BYTE_STREAM");
  if (e && atoi(e) == 0) return MDE_OK;
  hipStream_t st = mde_stream(stream);
  unsigned int* dtable = nullptr;
  int* dstate = nullptr;
  MDE_HIP(hipMalloc(&dtable, MDE_BX_SLOTS * sizeof(unsigned int) + 2 * sizeof(int)));
  dstate = reinterpret_cast<int*>(dtable + MDE_BX_SLOTS);
  hipError_t err = hipMemsetAsync(dtable, 0xFF, MDE_BX_SLOTS * sizeof(unsigned int), st);
  if (err == hipSuccess) err = hipMemsetAsync(dstate, 0, 2 * sizeof(int), st);
  std::vector<unsigned int> host_tb(MDE_BX_SLOTS);
  int host_state[2] = {0, 0};
  if (err == hipSuccess) {
    hipLaunchKernelGGL(k_bytes_scan, dim3(mde_grid(plan->p, MDE_BLOCK, 2048)), dim3(MDE_BLOCK), 0, st, plan->p, in_edge, dtable,
                       dstate);
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipMemcpyAsync(host_tb.data(), dtable, MDE_BX_SLOTS * sizeof(unsigned int), hipMemcpyDeviceToHost, st);
  if (err == hipSuccess) err = hipMemcpyAsync(host_state, dstate, sizeof(host_state), hipMemcpyDeviceToHost, st);
  if (err == hipSuccess) err = hipStreamSynchronize(st);
  if (err != hipSuccess) {
    (void)hipFree(dtable);
    return mde_hip_fail(err, "byte-index stream: value scan", __FILE__, __LINE__);
  }
  std::vector<unsigned int> vals;
  if (!host_state[1])
    for (unsigned int v : host_tb)
      if (v != MDE_CB_EMPTY) vals.push_back(v);
  bool ok = !host_state[1] && !vals.empty() && (int)vals.size() <= MDE_RING_BX_VALUES - 1;
  // (finite values of ordinary size only, as for the codebook: the kernel skips the NaN / Inf fix-up of f'/d)
  for (size_t i = 0; ok && i < vals.size(); ++i) {
    float fv;
    memcpy(&fv, &vals[i], sizeof(float));
    if (!std::isfinite(fv) || std::fabs(fv) > 1.0e6f) ok = false;
  }
  if (!ok) {
    (void)hipFree(dtable);
    return MDE_OK;
  }
  std::sort(vals.begin(), vals.end());
  const int nv = (int)vals.size();
  std::vector<unsigned int> tb(MDE_RING_BX_VALUES, 0u);  // entry 0 = +0.0f (padding lanes), unused entries 0 too
  for (int i = 0; i < nv; ++i) tb[1 + i] = vals[i];
  unsigned int* table = reinterpret_cast<unsigned int*>(out_half) + L.H / 4;  // behind the H index bytes
  err = hipMemcpyAsync(table, tb.data(), MDE_RING_BX_VALUES * sizeof(unsigned int), hipMemcpyHostToDevice, st);
  if (err == hipSuccess) {
    hipLaunchKernelGGL(k_bytes_pack, dim3(mde_grid(L.H, MDE_BLOCK, 4096)), dim3(MDE_BLOCK), 0, st, L.H, L.eid, in_edge, table, nv,
                       reinterpret_cast<uint8_t*>(out_half));
    err = hipGetLastError();
  }
  if (err == hipSuccess) err = hipStreamSynchronize(st);  // (`tb` is a host buffer)
  (void)hipFree(dtable);
  if (err != hipSuccess) return mde_hip_fail(err, "byte-index stream: pack", __FILE__, __LINE__);
  *n_values_host = nv;
  return MDE_OK;
}

// dst[0] (device double) <- the loss of the plan's last mde_average_distortion in double, before its rounding to float
extern "C" int mde_plan_loss_double(const mde_plan* plan, double* dst, void* stream) {
  if (!plan || !dst) return MDE_E_INVALID;
  MDE_HIP(hipMemcpyAsync(dst, plan->partials + MDE_PARTIALS_LOSS_D, sizeof(double), hipMemcpyDeviceToDevice, mde_stream(stream)));
  return MDE_OK;
}

extern "C" int mde_plan_ring_info(const mde_plan* plan, int64_t* info) {
  if (!plan || !info) return MDE_E_INVALID;
  const mde_ring_layout& L = plan->ring;
  for (int i = 0; i < 16; ++i) info[i] = 0;
  if (!L.packed) return MDE_OK;
  info[0] = 1;
  info[1] = L.d;
  info[2] = L.rows_per_block;
  info[3] = L.n_row_blocks;
  info[4] = L.col_groups;
  info[5] = L.n_chunks;
  info[6] = L.slots;
  info[7] = L.n_iters;
  info[8] = plan->H - L.hub_half_edges;
  info[9] = L.H;
  info[10] = L.slot_row ? 1 : 0;
  info[11] = L.n_hub_rows;
  info[12] = L.hub_half_edges;
  info[13] = L.n_hub_segs;
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

// ---------------------------------------------------------------- round 6: the peeled hub rows
// One workgroup per segment of MDE_HUB_SEG consecutive CSR positions of ONE hub row: gather x_u, evaluate (run-time
// functor; the parameters come from the caller's per-edge arrays through the plan's edge ids -- mde_func.e0 / e1 --,
// the ring-order streams do not hold these entries), sum g (x_v - x_u) and the loss terms over the segment in a fixed
// order (lane-strided partial sums, wave butterflies, waves in order; double).  k_hub_finish adds a row's segments in
// order, writes the gradient row (overwriting the zeros the ring kernel's epilogue left there) and adds the hub rows'
// loss to the ring kernel's: same stream, behind it.  count_all: the layout's rule for who adds an edge's loss term.
template <int D>
__global__ __launch_bounds__(MDE_BLOCK) void k_hub_rows(const int32_t* __restrict__ hub_rows, const int32_t* __restrict__ hub_seg,
                                                        int row_lo, const int32_t* __restrict__ nbr,
                                                        const int32_t* __restrict__ eid, const float* __restrict__ e0,
                                                        const float* __restrict__ e1, int e0_scalar, int e1_scalar,
                                                        const float* __restrict__ X, FnRuntime fn, float inv_p, int count_all,
                                                        double* __restrict__ partial) {
  __shared__ double smem[8];
  const int sgi = blockIdx.x;
  const int hi = hub_seg[4 * sgi], beg = hub_seg[4 * sgi + 1], end = hub_seg[4 * sgi + 2];
  const int64_t v = (int64_t)row_lo + hub_rows[hi];
  float xv[D];
#pragma unroll
  for (int c = 0; c < D; ++c) xv[c] = X[v * D + c];
  const float e0s = e0_scalar ? e0[0] : 0.0f;
  const float e1s = (e1 && e1_scalar) ? e1[0] : 0.0f;
  double acc[D], loss = 0.0;
#pragma unroll
  for (int c = 0; c < D; ++c) acc[c] = 0.0;
  for (int h = beg + (int)threadIdx.x; h < end; h += MDE_BLOCK) {
    const int u = nbr[h], k = eid[h];
    const float p0 = e0_scalar ? e0s : e0[k];
    const float p1 = (e1 && !e1_scalar) ? e1[k] : e1s;
    float diff[D], ss = 0.0f;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      diff[c] = xv[c] - X[(size_t)u * D + c];
      ss = fmaf(diff[c], diff[c], ss);
    }
    float f, gd;
    fn.eval(ss, p0, p1, f, gd);
    const float g = mde_fix_g(gd * inv_p);
    if (count_all || v < (int64_t)u) loss += (double)f;
#pragma unroll
    for (int c = 0; c < D; ++c) acc[c] += (double)(g * diff[c]);
  }
#pragma unroll
  for (int c = 0; c < D; ++c) {
    const double t = mde_block_sum(acc[c], smem);
    if (threadIdx.x == 0) partial[(size_t)sgi * 8 + c] = t;
  }
  const double tl = mde_block_sum(loss, smem);
  if (threadIdx.x == 0) partial[(size_t)sgi * 8 + 4] = tl;
}

// One wave per hub row (blocks 0 .. n_hub - 1): the lanes add the row's segments lane-strided, a butterfly adds the lanes
// (fixed order); the last block adds the segments' loss terms the same way.  (The first version gave a row to ONE
// thread: the 977 segments of a 5e5-degree hub were 977 dependent loads, 148 us.)
template <int D>
__global__ __launch_bounds__(64) void k_hub_finish(int n_hub, int n_seg, const int32_t* __restrict__ hub_rows,
                                                   const int32_t* __restrict__ hub_first,
                                                   const double* __restrict__ partial, int row_lo, float grad_scale,
                                                   float* __restrict__ grad, double loss_scale, float* __restrict__ loss_out,
                                                   double* __restrict__ loss_d) {
  const int lane = threadIdx.x;
  if ((int)blockIdx.x < n_hub) {
    if (!grad) return;
    const int i = blockIdx.x;
    double a[D];
#pragma unroll
    for (int c = 0; c < D; ++c) a[c] = 0.0;
    for (int sg = hub_first[i] + lane; sg < hub_first[i + 1]; sg += 64) {
#pragma unroll
      for (int c = 0; c < D; ++c) a[c] += partial[(size_t)sg * 8 + c];
    }
#pragma unroll
    for (int c = 0; c < D; ++c) a[c] = mde_wave_sum(a[c]);
    if (lane == 0) {
      const int64_t v = (int64_t)row_lo + hub_rows[i];
#pragma unroll
      for (int c = 0; c < D; ++c) grad[v * D + c] = (float)a[c] * grad_scale;
    }
    return;
  }
  double t = 0.0;
  for (int sg = lane; sg < n_seg; sg += 64) t += partial[(size_t)sg * 8 + 4];
  const double tot = mde_wave_sum(t);
  // (the ring kernel left its part of the loss in *loss_d in double: the hub rows' part is added before the one rounding)
  if (lane == 0) {
    *loss_d += tot * loss_scale;
    *loss_out = (float)*loss_d;
  }
}

template <int D>
static int hub_launch_d(const mde_plan* plan, const float* X, const mde_func* f, const float* e0, const float* e1, int e0_scalar,
                        int e1_scalar, float grad_scale, float* grad, float inv_p, hipStream_t st, float* loss_out, double loss_scale) {
  const mde_ring_layout& L = plan->ring;
  FnRuntime fn{ring_func_args(f)};
  hipLaunchKernelGGL(k_hub_rows<D>, dim3(L.n_hub_segs), dim3(MDE_BLOCK), 0, st, L.hub_rows, L.hub_seg, (int)plan->row_lo, plan->nbr,
                     plan->eid, e0, e1, e0_scalar, e1_scalar, X, fn, inv_p, L.count_all, L.hub_partial);
  MDE_LAUNCH_CHECK();
  const int nb = L.n_hub_rows + 1;
  // (the ring kernel's rule: the smaller endpoint adds f -> 2 x the caller's half-weight; count_all: every entry f / 2)
  hipLaunchKernelGGL(k_hub_finish<D>, dim3(nb), dim3(64), 0, st, L.n_hub_rows, L.n_hub_segs, L.hub_rows, L.hub_first,
                     L.hub_partial, (int)plan->row_lo, grad_scale, grad, (L.count_all ? 1.0 : 2.0) * loss_scale, loss_out,
                     plan->partials + MDE_PARTIALS_LOSS_D);
  MDE_LAUNCH_CHECK();
  return MDE_OK;
}

static int hub_launch(const mde_plan* plan, const float* X, int d, const mde_func* f, float grad_scale, float* grad, float inv_p,
                      hipStream_t st, float* loss_out, double loss_scale) {
  // the hub rows read their parameters in the caller's edge order
  const float *e0 = f->e0, *e1 = f->e1;
  int e0_scalar = 0, e1_scalar = 0;
  if (f->a0_scalar == 1) {
    e0 = f->a0;
    e0_scalar = 1;
  }
  if (f->a1 && f->a1_scalar) {
    e1 = f->a1;
    e1_scalar = 1;
  }
  if (!e0 || (f->a1 && !e1)) {
    mde_set_error("this plan's LDS-ring layout peels %d hub rows off to the CSR hub kernel: mde_func.e0 (and e1 for two-parameter "
                  "functions) must point at the per-edge arrays in the caller's edge order", plan->ring.n_hub_rows);
    return MDE_E_INVALID;
  }
  switch (d) {
    case 1: return hub_launch_d<1>(plan, X, f, e0, e1, e0_scalar, e1_scalar, grad_scale, grad, inv_p, st, loss_out, loss_scale);
    case 2: return hub_launch_d<2>(plan, X, f, e0, e1, e0_scalar, e1_scalar, grad_scale, grad, inv_p, st, loss_out, loss_scale);
    case 3: return hub_launch_d<3>(plan, X, f, e0, e1, e0_scalar, e1_scalar, grad_scale, grad, inv_p, st, loss_out, loss_scale);
    default: return hub_launch_d<4>(plan, X, f, e0, e1, e0_scalar, e1_scalar, grad_scale, grad, inv_p, st, loss_out, loss_scale);
  }
}

// ---------------------------------------------------------------- dispatch
// Called by mde_average_distortion.  Returns 1 when the ring kernel was launched (it also writes
// *loss_out = loss_scale * sum of the workgroups' partials; nblocks = number of partials), 0 when
// the caller should use the CSR kernel, < 0 on error.
int mde_ring_try(mde_plan* plan, const float* X, int d, const mde_func* f, float grad_scale,
                 float* grad, float inv_p, hipStream_t st, int* nblocks, float* loss_out,
                 double loss_scale) {
  if (!plan->ring.packed || plan->ring.d != d) return 0;
  const RingArgs A{plan, X, d, f->a0, f->a1, f->a0_scalar, f->a1_scalar, grad, inv_p, grad_scale, st,
                   loss_out, loss_scale};
  // the family's unit first (one code object touched per process), the run-time functor otherwise
  int rc = 0;
  if (f->kind_neg != MDE_F_NONE)
    rc = mde_ring_launch_pushpull(A, f, nblocks);
  else if (f->kind == MDE_F_LOG1P)
    rc = mde_ring_launch_log1p(A, f, nblocks);
  else if (f->kind == MDE_F_L_QUADRATIC || f->kind == MDE_F_L_WEIGHTED_QUADRATIC || f->kind == MDE_F_L_ABSOLUTE ||
           f->kind == MDE_F_L_HUBER)
    rc = mde_ring_launch_loss(A, f, nblocks);
  else if (f->kind == MDE_F_L_CUBIC || f->kind == MDE_F_L_POWER || f->kind == MDE_F_L_LOGISTIC ||
           f->kind == MDE_F_L_FRACTIONAL || f->kind == MDE_F_L_SOFT_FRACTIONAL)
    rc = mde_ring_launch_loss2(A, f, nblocks);
  else if (f->kind == MDE_F_LOGISTIC || f->kind == MDE_F_SIGMOID || f->kind == MDE_F_HINGE || f->kind == MDE_F_POWER ||
           f->kind == MDE_F_INVPOWER || f->kind == MDE_F_LOGRATIO)
    rc = mde_ring_launch_penalty2(A, f, nblocks);
  else
    rc = mde_ring_launch_penalty(A, f, nblocks);
  if (rc == 0) rc = mde_ring_launch_runtime(A, f, nblocks);
  if (rc == 1 && plan->ring.n_hub_rows > 0) {
    const int hrc = hub_launch(plan, X, d, f, grad_scale, grad, inv_p, st, loss_out, loss_scale);
    if (hrc != MDE_OK) return hrc;
  }
  return rc;
}
