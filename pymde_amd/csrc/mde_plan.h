// mde_plan.h -- internal definition of the edge plan (shared by the translation units).
#pragma once
#include "mde_common.h"

// Column-panel layout for the LDS-tiled small-d kernel (built lazily, per embedding dim).
// Half-edges are grouped into tiles (row block rb, column panel cp), sorted by row inside a
// tile; tile t = rb * n_panels + cp covers [tile_ptr[t], tile_ptr[t+1]).
#ifndef MDE_PANEL_WAVES
#define MDE_PANEL_WAVES 16  // waves per workgroup of the panel kernel (64 * WAVES threads)
#endif
struct mde_panel_layout {
  int d = 0;            // embedding dimension the tile sizes were chosen for
  int rows_per_block = 0, cols_per_panel = 0;
  int n_row_blocks = 0, n_panels = 0;
  int64_t H = 0;                // padded entry count: 64 * (wave iterations of all sub-ranges)
  uint32_t* packed = nullptr;   // [H] LDS row address << 17 | LDS panel offset (padding: MDE_PANEL_DUMMY)
  int32_t* eid = nullptr;       // [H] original edge id (parameter expansion), -1 for padding
  int32_t* next_tile = nullptr; // [n_row_blocks * (n_panels + 1)] first non-empty panel >= cp of a row block
  int32_t* sub_off = nullptr;   // [n_tiles * MDE_PANEL_WAVES + 1] first wave iteration of each
                                // (tile, wave) sub-range; entries [64 * sub_off[i], 64 * sub_off[i+1])
  int col_groups = 1;           // Q: workgroups per row block, each walking 1/Q of the panels
  float* partial = nullptr;     // [Q * nloc * d] per-group gradient partials (Q > 1 only)
};

struct mde_plan {
  int64_t n = 0, p = 0, H = 0, row_lo = 0, row_hi = 0;
  int32_t* rowptr = nullptr;
  int32_t* nbr = nullptr;
  int32_t* eid = nullptr;
  double* partials = nullptr;  // [MDE_MAX_PARTIALS] loss partial sums of the fused kernel
  float avg_degree = 0.f;
  mde_panel_layout panel;      // empty until mde_plan_build_panels succeeds
};
