// mde_plan.h -- internal definition of the edge plan (shared by the translation units).
#pragma once
#include "mde_common.h"

// LDS-ring layout of the small-d kernel (mde_ring.hip; built lazily, per embedding dim).
// A workgroup (row block rb, column group qg) runs MDE_RING_NCW consumer waves; the
// stream of consumer wave w is the wave iterations [wave_iter[s], wave_iter[s + 1]) with
// s = (rb * col_groups + qg) * NCW + w, 64 packed half-edges each, chunk-major.
struct mde_ring_layout {
  int d = 0;                    // embedding dimension the sizes were chosen for
  int rejected_d = 0;           // dimension for which the ring layout was tried and given up (too much padding)
  float cost_scale = 1.0f;      // what an iteration of the coming function costs relative to Log1p (mde_plan_function_hint)
  int rows_per_block = 0, n_row_blocks = 0;
  int col_groups = 1;           // Q: workgroups per row block, each walking 1/Q of the chunks
  int chunk_cols = 0, n_chunks = 0;
  int ring_off = 0, slots = 0;  // LDS byte offset of the chunk ring and its number of slots (depend on rows_per_block)
  int64_t n_iters = 0;          // wave iterations of all streams
  int64_t H = 0;                // padded entry count = 64 * n_iters
  uint32_t* packed = nullptr;   // [H] absolute LDS addresses of x_v and x_u (ring_pack_word, mde_ring.hip)
  int32_t* eid = nullptr;       // [H] original edge id (parameter expansion), -1 for padding
  uint32_t* hdr = nullptr;      // [2 * n_iters] chunk window | padding flag and loss class of an iteration
  int32_t* wave_iter = nullptr; // [n_row_blocks * col_groups * NCW + 1]
  float* partial = nullptr;     // [Q * nloc * d] per-group gradient partials (Q > 1 only)
  // Round 6 -- graphs the equal-rows, everybody-in-the-ring layout gave up on (mde_ring.hip, plan_rows):
  //  * PERMUTED row blocks: slot_row[rb * rows_per_block + s] = local row whose x_v / accumulator live in LDS slot s
  //    of row block rb (-1: unused), rows dealt to the blocks so that every block holds the same number of
  //    half-edges whatever the degrees do along the vertex order; nullptr: slot s of block rb is row rb * R + s.
  //    A permuted layout adds EVERY entry's loss term with weight 1/2 (the rows of a wave are no longer a range of
  //    the vertex order, so "the entry whose row is the smaller vertex" would be a per-lane test everywhere).
  //  * PEELED hub rows: rows with more half-edges than a wave's stream can take one per iteration are left out of
  //    the ring streams; k_hub_rows / k_hub_finish (mde_ring.hip) evaluate them from the CSR plan in segments of
  //    MDE_HUB_SEG positions, behind the ring kernel on the same stream (one writer per row, fixed order).
  int32_t* slot_row = nullptr;
  int count_all = 0;            // 1: every entry adds f / 2 (permuted layouts); 0: the smaller endpoint adds f
  int n_hub_rows = 0, n_hub_segs = 0;
  int64_t hub_half_edges = 0;
  int32_t* hub_rows = nullptr;  // [n_hub_rows] local row ids, ascending
  int32_t* hub_seg = nullptr;   // [n_hub_segs + 1][2]: index into hub_rows | first CSR position (segments of a row are consecutive; the extra entry closes the last one)
  int32_t* hub_first = nullptr; // [n_hub_rows + 1] first segment of each hub row
  double* hub_partial = nullptr;  // [n_hub_segs][8]: sum g (x_v - x_u) [<= 4] | loss
};
#define MDE_HUB_SEG 512

struct mde_plan {
  int64_t n = 0, p = 0, H = 0, row_lo = 0, row_hi = 0;
  int32_t* rowptr = nullptr;
  int32_t* nbr = nullptr;
  int32_t* eid = nullptr;
  double* partials = nullptr;  // [MDE_MAX_PARTIALS] loss partial sums of the fused kernel
  float avg_degree = 0.f;
  mde_ring_layout ring;        // empty until mde_plan_layout builds it
  // Edge-balanced ("flat") schedule of the CSR kernel, built on first use (mde_plan_flat):
  // tiles of MDE_FLAT_T consecutive half-edge positions, aligned to GLOBAL positions (h_offset =
  // half-edges of the rows below row_lo), so a row is summed in the same order whoever owns it.
  int64_t h_offset = 0;
  int32_t* hrow = nullptr;     // [H] local row of each half-edge position
  float* flat_rec = nullptr;   // [n_tiles][2][8] boundary runs of a tile: row, ends?, sum[<= 4]
  int64_t n_tiles = 0;
  int has_empty = 0;           // some row has no half-edge (its gradient row is zero-filled)
  // Processing order of the rows for the general-d kernel (round 6, mde_plan_row_order): order[q] = local row
  // evaluated q-th.  Rows sorted by breadth-first level (ties by row id), kept only when it brings the two ends of
  // an edge closer together than the caller's numbering does.  Results do not depend on it (a row's sum is the
  // row's own business); which rows are in the caches at the same time does.
  int32_t* order = nullptr;
  int order_state = 0;         // 0 not tried, 1 adopted, -1 tried and rejected
  double order_before = 0.0, order_after = 0.0;  // mean |position(v) - position(u)| over the local half-edges
  int order_levels = 0;
};

#define MDE_FLAT_U 4                    // wave iterations per tile
#define MDE_FLAT_T (64 * MDE_FLAT_U)    // half-edge positions per tile (one wave)
int mde_plan_flat(mde_plan* plan, hipStream_t st);  // mde_plan.hip

// device-wide sort / scan shared by the plan and layout builders (defined in mde_plan.hip)
hipError_t mde_sort_pairs_u32(void* tmp, size_t& tmp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                              const uint32_t* vals_in, uint32_t* vals_out, int n, int begin_bit, int end_bit,
                              hipStream_t st);
hipError_t mde_exclusive_sum_i32(void* tmp, size_t& tmp_bytes, const int32_t* in, int32_t* out, int n, hipStream_t st);
