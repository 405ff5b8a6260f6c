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
};

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
};

#define MDE_FLAT_U 4                    // wave iterations per tile
#define MDE_FLAT_T (64 * MDE_FLAT_U)    // half-edge positions per tile (one wave)
int mde_plan_flat(mde_plan* plan, hipStream_t st);  // mde_plan.hip

// device-wide sort / scan shared by the plan and layout builders (defined in mde_plan.hip)
hipError_t mde_sort_pairs_u32(void* tmp, size_t& tmp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                              const uint32_t* vals_in, uint32_t* vals_out, int n, int begin_bit, int end_bit,
                              hipStream_t st);
hipError_t mde_exclusive_sum_i32(void* tmp, size_t& tmp_bytes, const int32_t* in, int32_t* out, int n, hipStream_t st);
