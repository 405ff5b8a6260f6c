/*
 * mde_hip.h -- C ABI of libmde_hip.so, the MI355X (gfx950 / CDNA4) implementation of
 * the minimum-distortion-embedding hot path.
 *
 * Every entry point below replaces one seam of the reference (cvxgrp/pymde v0.2.1); the
 * reference location it stands in for is cited as  [ref: file:line].  The reference has
 * no native boundary of its own on this path (it is torch ops called from Python), so
 * this header IS the FFI a maintainer binds with ctypes -- see INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - Every `*_dev` / unqualified data pointer is a DEVICE pointer on the device that
 *     is current on the calling thread.  `mde_func`, `mde_lbfgs_coef` are HOST structs.
 *   - All work is enqueued asynchronously on `stream` (a hipStream_t passed as void*);
 *     nothing synchronises unless documented ("SYNC").
 *   - The library borrows caller memory for the duration of the enqueued work and never
 *     frees it.  Objects created here (mde_plan, mde_lbfgs) own their device memory.
 *   - Return value: MDE_OK (0) or a negative MDE_E_* code.  No exceptions cross the ABI.
 *     mde_last_error() returns a thread-local human-readable message for the last failure.
 *   - Embeddings / gradients are contiguous row-major float32 [n, d].  Edge lists are
 *     int64 [p, 2] at the boundary (the reference's dtype, problem.py:130-132) and int32
 *     inside a plan.
 */
#ifndef MDE_HIP_H_
#define MDE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDE_ABI_VERSION 2   /* 2 (round 6): mde_func.e0 / e1, mde_plan_ring_info */

/* ------------------------------------------------------------------ error codes */
#define MDE_OK 0
#define MDE_E_INVALID (-1)     /* bad argument (null pointer, negative size, d out of range) */
#define MDE_E_SELF_EDGE (-2)   /* an edge (i, i) was found        [ref: problem.py:134-140]  */
#define MDE_E_RANGE (-3)       /* an endpoint is outside [0, n)                              */
#define MDE_E_TOO_LARGE (-4)   /* n >= 2^31 or 2p >= 2^31 (int32 plan)                       */
#define MDE_E_HIP (-5)         /* a HIP runtime call failed; see mde_last_error()            */
#define MDE_E_UNSUPPORTED (-6) /* function kind / dimension not handled by the fused path    */

const char* mde_last_error(void);
int mde_abi_version(void);

/* ------------------------------------------------------------------ distortion functions
 * The per-edge distortion f_k(d_k) and its derivative are evaluated inside the kernels.
 * `kind` selects the closed form; `a0` / `a1` are the per-edge parameter arrays
 * (weights w_k for penalties, deviations delta_k for losses; a1 = weights of the
 * weighted losses); s0..s2 are the scalar hyper-parameters.
 * [ref: pymde/functions/penalties.py:112-400, pymde/functions/losses.py:61-239]
 *
 *   penalties (a0 = w)                          s0          s1      s2
 *   MDE_F_LINEAR            w d
 *   MDE_F_QUADRATIC         w d^2
 *   MDE_F_CUBIC             w d^3
 *   MDE_F_POWER             w d^e                e
 *   MDE_F_HUBER             w d^2/2 | w t(d-t/2) t (threshold)     (strict <, :230)
 *   MDE_F_LOGISTIC          w log(1+exp(a(d-t))) t          a
 *   MDE_F_SIGMOID           w sigmoid(a(d-t))    t          a
 *   MDE_F_HINGE             max(0,w(d-(t-sgn(w)s))) t       s (sigma)
 *   MDE_F_LOG1P             w log1p(d^e)         e
 *   MDE_F_LOG               w log(-expm1(-d^e))  e
 *   MDE_F_INVPOWER          |w| / d^e            e
 *   MDE_F_LOGRATIO          w log(d^e/(1+d^e))   e
 *   MDE_F_DEADZONE_QUADRATIC  w (d<t ? 0 : d^2)  t
 *   MDE_F_DEADZONE_CUBIC      w (d<t ? 0 : d^3)  t
 *   MDE_F_CLIPPED_QUADRATIC   w min(d^2,(t+1)^2) t
 *   losses (a0 = delta, r = |delta - d|)
 *   MDE_F_L_QUADRATIC       (delta-d)^2
 *   MDE_F_L_WEIGHTED_QUADRATIC  a1 (delta-d)^2
 *   MDE_F_L_HUBER           r^2 | t(2r-t)        t                   (strict <, losses.py:122)
 *   MDE_F_L_CUBIC           r^3
 *   MDE_F_L_POWER           r^e                  e
 *   MDE_F_L_WEIGHTED_POWER  a1 r^e               e
 *   MDE_F_L_ABSOLUTE        r
 *   MDE_F_L_LOGISTIC        log(1+exp(r))
 *   MDE_F_L_FRACTIONAL      max(delta/d, d/delta) - 1
 *   MDE_F_L_SOFT_FRACTIONAL (1/g)(lse(g delta/d, g d/delta) - log 2 - g)   s0 = g (gamma)
 *   MDE_F_L_CLIPPED_QUADRATIC   min((delta-d)^2,(t+1)^2)  t
 *   MDE_F_L_LOG1P               log(1 + (d-delta)^e)      e   (losses._Log1p; torch.pow semantics:
 *                               NaN for d < delta unless e is an integer)
 *
 * PushAndPull [ref: penalties.py:372-400]: set `kind` to the attractive penalty and
 * `kind_neg` to the repulsive one (with its scalars in n0..n2); an edge uses `kind` when
 * w_k >= 0 and `kind_neg` when w_k < 0 (zero weight is attractive, penalties.py:390).
 * kind_neg = MDE_F_NONE means a plain (single) function.
 */
enum {
  MDE_F_NONE = 0,
  MDE_F_LINEAR = 1,
  MDE_F_QUADRATIC = 2,
  MDE_F_CUBIC = 3,
  MDE_F_POWER = 4,
  MDE_F_HUBER = 5,
  MDE_F_LOGISTIC = 6,
  MDE_F_SIGMOID = 7,
  MDE_F_HINGE = 8,
  MDE_F_LOG1P = 9,
  MDE_F_LOG = 10,
  MDE_F_INVPOWER = 11,
  MDE_F_LOGRATIO = 12,
  MDE_F_DEADZONE_QUADRATIC = 13,
  MDE_F_DEADZONE_CUBIC = 14,
  MDE_F_CLIPPED_QUADRATIC = 15,
  MDE_F_L_QUADRATIC = 32,
  MDE_F_L_WEIGHTED_QUADRATIC = 33,
  MDE_F_L_HUBER = 34,
  MDE_F_L_CUBIC = 35,
  MDE_F_L_POWER = 36,
  MDE_F_L_WEIGHTED_POWER = 37,
  MDE_F_L_ABSOLUTE = 38,
  MDE_F_L_LOGISTIC = 39,
  MDE_F_L_FRACTIONAL = 40,
  MDE_F_L_SOFT_FRACTIONAL = 41,
  MDE_F_L_CLIPPED_QUADRATIC = 42,
  MDE_F_L_LOG1P = 43
};

typedef struct mde_func {
  int32_t kind;      /* MDE_F_* (attractive branch for PushAndPull)                       */
  int32_t kind_neg;  /* MDE_F_NONE, or the repulsive branch of a PushAndPull              */
  const float* a0;   /* device; per-edge weights / deviations.  Order: see each call      */
  const float* a1;   /* device; second per-edge array (weighted losses) or NULL           */
  int32_t a0_scalar; /* 1: a0 points at ONE value broadcast to every edge (nelement()==1);
                        2: a0 is a codebook stream written by mde_plan_expand_codebook (layout 1);
                        3: a0 is a byte-index stream written by mde_plan_expand_bytes (layout 1)  */
  int32_t a1_scalar;
  float s0, s1, s2;  /* scalars of `kind`                                                 */
  float n0, n1, n2;  /* scalars of `kind_neg`                                             */
  int32_t layout;    /* order of a0/a1 for mde_average_distortion: 0 = CSR plan order,
                        1 = LDS-ring order (mde_plan_layout / mde_plan_expand_layout)        */
  const float* e0;   /* device; the per-edge arrays behind a0 / a1 in the CALLER'S edge order [p] (the   */
  const float* e1;   /* order of the edge list given to mde_plan_create), or NULL.  Needed with layout 1
                        when the plan's ring layout peels hub rows off to the CSR hub kernel
                        (mde_plan_ring_info: those rows read their parameters through the plan's edge
                        ids); ignored otherwise, and for parameters that are one broadcast scalar    */
} mde_func;

/* ------------------------------------------------------------------ the edge plan
 * One-time device preprocessing of an edge list [ref: problem.py:129-170, the `edges`,
 * `_lhs`, `_rhs` buffers].  The plan stores the SYMMETRISED incidence structure in CSR
 * form: for every vertex v in [row_lo, row_hi) the list of its incident half-edges
 * (neighbour u, original edge id k), so the gradient row of v is a pure gather
 *      grad[v] = sum_{h in row v} g_h (x_v - x_{nbr[h]})
 * with no atomics and a fixed summation order (bitwise reproducible).  Half-edges of a
 * row keep the order of their original edge ids (stable sort).
 *
 * row_lo/row_hi select the vertex range this plan (rank) owns; pass 0, n for a
 * single-GPU plan.  Validation (self edges, range) always covers the full edge list.
 * SYNC: returns after the plan is built (it must read back counts).
 */
typedef struct mde_plan mde_plan;

int mde_plan_create(int64_t n, int64_t p, const int64_t* edges, int64_t row_lo, int64_t row_hi,
                    void* stream, mde_plan** out);
int mde_plan_destroy(mde_plan* plan);
int64_t mde_plan_n(const mde_plan* plan);
int64_t mde_plan_p(const mde_plan* plan);           /* edges in the full problem           */
int64_t mde_plan_half_edges(const mde_plan* plan);  /* half-edges stored (= 2p if full)    */
int64_t mde_plan_row_lo(const mde_plan* plan);
int64_t mde_plan_row_hi(const mde_plan* plan);
const int32_t* mde_plan_rowptr(const mde_plan* plan); /* [row_hi-row_lo+1], offsets into nbr */
const int32_t* mde_plan_nbr(const mde_plan* plan);    /* [half_edges] neighbour vertex      */
const int32_t* mde_plan_eid(const mde_plan* plan);    /* [half_edges] original edge id      */

/* Copy the plan arrays into caller buffers (rowptr [nloc+1], nbr [H], eid [H]; any may be NULL). */
int mde_plan_export(const mde_plan* plan, int32_t* rowptr_out, int32_t* nbr_out, int32_t* eid_out,
                    void* stream);

/* Balanced vertex-range boundaries for `world` ranks: bounds[r]..bounds[r+1] holds about
 * 2p/world half-edges.  `bounds_host` is a HOST array of world+1 int64.  SYNC. */
int mde_shard_bounds(int64_t n, int64_t p, const int64_t* edges, int32_t world,
                     int64_t* bounds_host, void* stream);

/* out_half[h] = in_edge[eid[h]] : put a per-edge parameter array into plan (CSR) order. */
int mde_plan_expand(const mde_plan* plan, const float* in_edge, float* out_half, void* stream);

/* Layout the fused kernel prefers for embedding dimension d: 0 = the CSR order above,
 * 1 = the LDS-ring layout of mde_ring.hip (d <= 4, embedding table larger than L2: per-wave streams
 * of packed half-edges, chunk-major; built on first request, SYNC).
 * Per-edge parameters for layout 1 are permuted with mde_plan_expand_layout and flagged with
 * mde_func.layout = 1.  Negative return: error. */
int mde_plan_layout(mde_plan* plan, int32_t d, void* stream);
/* Tell the plan which distortion function its next layout decision is for (mde_func.kind / kind_neg): for problems
 * whose table fits L2 and that have fewer than 16 M half-edges the choice between the two layouts is close, and the
 * ring kernel pays for an expensive function (PushAndPull, Log, the run-time functor of private kinds) where the CSR
 * kernels hide it behind their gathers.  Optional (default: a Log1p-class function); call it before mde_plan_layout.
 * Never changes a result, only which kernel computes it. */
int mde_plan_function_hint(mde_plan* plan, int32_t kind, int32_t kind_neg);
/* Entries of a per-half-edge parameter array in `layout` (layout 1 stores whole 64-entry wave
 * iterations, padded where a stream ends or its chunk window closes, so it is larger than
 * mde_plan_half_edges). */
int64_t mde_plan_layout_half_edges(const mde_plan* plan, int32_t layout);
/* Parameter codebook for layout 1 at d = 2 and d = 3: when `in_edge` [p] holds at most 7 (d = 2) / 3
 * (d = 3) distinct values (k-NN weights 1 / 2, -1 for repulsive pairs, ...) write to out_half
 * [mde_plan_layout_half_edges(plan, 1)] the packed half-edge words with the value index in the low
 * bits their row address leaves free (3 / 2), followed by the 8-entry value table (entry 0 = 0.0, the weight of padding lanes;
 * entries 1..7 the values in ascending bit order), and set *n_values_host to the number of values;
 * the fused kernel then streams 4 instead of 8 bytes per half-edge (mde_func.a0 = out_half,
 * a0_scalar = 2).  *n_values_host = 0: not applicable, nothing written -- use
 * mde_plan_expand_layout (also when a value is NaN or infinite: the kernel skips the NaN/Inf -> 1
 * fix-up of f'/d for codebook streams).  Results are identical either way.  SYNC. */
int mde_plan_expand_codebook(const mde_plan* plan, const float* in_edge, float* out_half,
                             int32_t* n_values_host, void* stream);
/* Byte-index parameter stream for layout 1 at d = 2 and d = 3: when `in_edge` [p] holds at most 255 distinct
 * values (the hop-count deviations of preserve_distances on a graph [ref: pymde/recipes.py:194-215,
 * preprocess/graph.py:402-474], quantised weights) write to out_half [mde_plan_layout_half_edges(plan, 1)
 * floats of space] one index BYTE per entry of the layout (in the order of the packed half-edge words),
 * followed at byte offset H by the 256-entry value table (entry 0 = 0.0, the weight of padding lanes; entries
 * 1..n the values in ascending bit order), and set *n_values_host = n; the fused kernel then streams 5 instead
 * of 8 bytes per half-edge and keeps the table in LDS (mde_func.a0 = out_half, a0_scalar = 3).
 * *n_values_host = 0: not applicable (more values, a NaN / infinite / > 1e6 value, a layout without 1 KB of LDS
 * behind its chunk ring) -- use mde_plan_expand_layout.  Results are identical either way.  SYNC. */
int mde_plan_expand_bytes(const mde_plan* plan, const float* in_edge, float* out_half,
                          int32_t* n_values_host, void* stream);
int mde_plan_expand_layout(const mde_plan* plan, int32_t layout, const float* in_edge,
                           float* out_half, void* stream);
/* What the LDS-ring layout of the plan looks like (after mde_plan_layout returned 1), 16 HOST int64:
 * [0] built (0 / 1), [1] embedding dimension, [2] rows (LDS slots) per row block, [3] row blocks, [4] column groups,
 * [5] chunks, [6] ring slots, [7] wave iterations, [8] half-edges in the ring streams, [9] padded entries,
 * [10] row blocks PERMUTED (rows dealt to the blocks by degree; every entry adds f / 2), [11] hub rows PEELED off to
 * the CSR hub kernel, [12] their half-edges, [13] their segments, [14], [15] reserved (0). */
int mde_plan_ring_info(const mde_plan* plan, int64_t* info_host);
/* Processing order of the plan's rows for the general-d kernel (d = 5 .. 512; round 6).  The reference evaluates
 * the edges in the caller's order whatever the numbering of the items [ref: pymde/average_distortion.py:62-106]; here
 * the rows of X gathered at the same time share the caches only when neighbours are close in the order the rows are
 * evaluated in.  The call runs a breadth-first search over the plan's rows and sorts them by (level, row id); nothing
 * is renumbered -- X, the gradient and the plan arrays keep the caller's numbering, results are identical with and
 * without an order.  mode 1: keep the order when the mean distance between the positions of an edge's two ends at
 * least halves; 2: keep it regardless; 0: drop it.  mde_average_distortion calls it with mode 1 on a plan's first
 * evaluation at such a d (env MDE_ROW_ORDER=0 / 2 overrides).  info_host (may be NULL): 4 HOST doubles --
 * [0] an order is in use, [1] mean |v - u| over the local half-edges in the caller's numbering, [2] the same in the
 * order built (0: the search was abandoned -- a level held more than an eighth of the rows, or more than 256
 * components), [3] breadth-first levels.  SYNC; one launch and one 4-byte read-back per level. */
int mde_plan_row_order(mde_plan* plan, int32_t mode, void* stream, double* info_host);
/* dst[0] (DEVICE double) <- the loss the plan's last mde_average_distortion wrote to loss_out, in double, before the
 * rounding to float.  A solve whose rows are sharded across ranks sums the ranks' shares in double and rounds once,
 * like the single-GPU kernel does (the reference's line search branches on the last bit of the loss).  ASYNC. */
int mde_plan_loss_double(const mde_plan* plan, double* dst, void* stream);

/* ------------------------------------------------------------------ edge-list preprocessing
 *  [ref: pymde/preprocess/preprocess.py:11-129]
 * De-duplicate an edge list: rows are put in (min, max) order and the unique rows are written to
 * edges_out [>= p, 2] sorted by (i, j) -- np.unique(axis=0)'s order; *count_host = their number. SYNC. */
int mde_edges_deduplicate(int64_t n, int64_t p, const int64_t* edges, int64_t* edges_out,
                          int64_t* count_host, void* stream);
/* Unique edges (i < j, sorted) with their multiplicity as a float weight; rows with i == j or a
 * negative index are dropped.  This is how the k-NN graph gets weight 2 on mutual neighbours
 * [ref: preprocess/data_matrix.py:141-178, graph.py:51-72].  SYNC. */
int mde_edges_count_unique(int64_t n, int64_t p, const int64_t* edges, int64_t* edges_out,
                           float* weights_out, int64_t* count_host, void* stream);
/* Exact k nearest neighbours (Euclidean) of every row of data [n, nf]
 * [ref: preprocess/data_matrix.py:91-140].  idx_out [n, k] int32 (self excluded; -1 when fewer than
 * k other rows exist), d2_out [n, k] squared distances ascending per row; 1 <= k <= 64;
 * sqn_work: n floats of scratch.  The Gram tiles run on the f32 matrix cores. */
int mde_knn(int64_t n, int32_t nf, const float* data, int32_t k, int32_t* idx_out, float* d2_out,
            float* sqn_work, void* stream);
/* Shortest-path distances on the graph whose edges built `plan` (a FULL plan; its symmetrised CSR
 * is the adjacency) [ref: preprocess/graph.py:286-474, _graph.pyx:10-52].
 * w: per-half-edge edge lengths in plan (CSR) order (mde_plan_expand), or NULL for unit lengths
 * (BFS).  For every pair i < j at finite positive distance <= max_length (<= 0: no limit) the pair is
 * kept with probability retain_fraction (>= 1: all), decided by a hash of (seed, i, j).  Writes the
 * kept pairs sorted by (i, j) to edges_out [capacity, 2] / dist_out [capacity]; *count_host = their
 * number (fails with MDE_E_INVALID, count still reported, when it exceeds `capacity`).  SYNC. */
int mde_graph_shortest_paths(const mde_plan* plan, const float* w, float max_length,
                             double retain_fraction, uint64_t seed, int64_t capacity,
                             int64_t* edges_out, float* dist_out, int64_t* count_host, void* stream);
/* Edge list of directed neighbour lists: pairs_out[r * k + c] = (r, idx[r][c]); empty slots (idx < 0)
 * and entries with val > max_value (val may be NULL) become self pairs, which
 * mde_edges_count_unique drops [ref: preprocess/data_matrix.py:147-175]. */
int mde_knn_pairs(int64_t n, int32_t k, const int32_t* idx, const float* val, float max_value,
                  int64_t* pairs_out, void* stream);
/* k nearest neighbours of every vertex under the graph's shortest-path metric (direct = 0) or among
 * its graph neighbours by edge length (direct = 1) [ref: preprocess/graph.py:502-587].  plan / w /
 * max_length as for mde_graph_shortest_paths.  idx_out [n, k] int32 (-1 = fewer than k targets
 * within max_length), dist_out [n, k] ascending; ties go to the smaller index.  SYNC. */
int mde_graph_knn(const mde_plan* plan, const float* w, float max_length, int32_t direct, int32_t k,
                  int32_t* idx_out, float* dist_out, void* stream);
/* Sample at most num_edges distinct edges i < j uniformly at random from the edges NOT in
 * `exclude` [n_exclude, 2] (NULL / 0: no exclusion); edges_out must hold num_edges rows, sorted by
 * (i, j); *count_host = number written (== num_edges unless the complement is nearly exhausted).
 * Same `seed` -> same edges.  SYNC. */
int mde_sample_edges(int64_t n, int64_t num_edges, uint64_t seed, const int64_t* exclude,
                     int64_t n_exclude, int64_t* edges_out, int64_t* count_host, void* stream);

/* ------------------------------------------------------------------ the hot kernel
 * Fused forward + backward of the average distortion
 *   E(X) = (1/p) sum_k f_k(||x_ik - x_jk||),  dE/dX          [ref: average_distortion.py:62-106]
 * `f->a0/a1` are in PLAN (half-edge) order (mde_plan_expand) unless *_scalar is set.
 * Writes rows [row_lo,row_hi) of grad (other rows untouched) when grad != NULL, and the
 * plan's share of E into *loss_out (device float; full E for a full plan).  With
 * grad == NULL this is the forward-only evaluation (average_distortion.py:90-91).
 * g_k = f'_k(d_k)/(p d_k) with NaN -> 1, Inf -> 1 as in average_distortion.py:81-88.
 * `grad_scale` multiplies the gradient (upstream grad_output, average_distortion.py:105).
 */
int mde_average_distortion(mde_plan* plan, const float* X, int32_t d, const mde_func* f,
                           float grad_scale, float* grad, float* loss_out, void* stream);

/* ------------------------------------------------------------------ edge-order evaluators
 * [ref: problem.py:246-308 differences / distances / distortions; average_distortion.py:38-55]
 * These work on the caller's ORIGINAL edge order and need no plan. */
int mde_differences(int64_t n, int64_t p, const int64_t* edges, const float* X, int32_t d,
                    float* diff_out /* [p,d] */, void* stream);
int mde_distances(int64_t n, int64_t p, const int64_t* edges, const float* X, int32_t d,
                  float* dist_out /* [p] */, void* stream);
/* backward of distances: grad_X += sum_k gout_k (x_i - x_j)/d_k (+ to i, - to j), NaN -> 0
 * (average_distortion.py:46-52).  Needs a full plan of the same edges; gout in edge order. */
int mde_distances_backward(const mde_plan* plan, const float* X, int32_t d, const float* gout,
                           float* grad /* [n,d], rows of the plan overwritten */, void* stream);
/* out_k = f_k(dist_k) and, when dout != NULL, dout_k = f'_k(dist_k); parameters in EDGE order. */
int mde_distortions(int64_t p, const float* dist, const mde_func* f, float* out, float* dout,
                    void* stream);

/* Unfused fallback for arbitrary Python callables (average_distortion.py:73-105):
 * grad[v] = scale * sum_h gfix(gnorm[eid[h]] / dist[eid[h]]) (x_v - x_nbr), where
 * gnorm = d(mean f)/d(dist) from torch autograd and gfix maps NaN/Inf to 1.0 (:81-88). */
int mde_scatter(const mde_plan* plan, const float* X, int32_t d, const float* gnorm,
                const float* dist, float scale, float* grad, void* stream);

/* ------------------------------------------------------------------ constraints
 * [ref: constraints.py:94-200, util.py:129-171] */
/* Z -= column mean (Centered retraction, constraints.py:106-111). `work` >= mde_work_doubles(d). */
int mde_center(int64_t n, int32_t d, float* Z, double* work, void* stream);
/* Anchored (constraints.py:143-164): rows[anchors] = values (or 0 when values == NULL). */
int mde_anchor_rows(int64_t n_anchors, int32_t d, const int64_t* anchors, const float* values,
                    float* Z, void* stream);
/* Rows on a sphere (constraints.py:203-231, `_Sphere`: private in the reference, used by no recipe).
 * X == NULL: the retraction Z[r] <- (Z[r] / |Z[r]|) * radius (:225-231); else the tangent projection
 * Z[r] -= (1 / radius) (Z[r] . X[r]) X[r] (:214-223; the reference's own scale, 1 / radius). */
int mde_sphere_rows(int64_t n, int32_t d, const float* X, float* Z, float radius, void* stream);
/* Standardized tangent projection Z -= (1/n) X (Z^T X)  (constraints.py:186-192). */
int mde_std_tangent(int64_t n, int32_t d, const float* X, float* Z, double* work, void* stream);
/* Standardized retraction: Z <- sqrt(n) * polar factor of (Z - mean)  (util.py:129-161),
 * computed as sqrt(n) (Z-mean) C^{-1/2}, C = (Z-mean)^T (Z-mean), with C^{-1/2} from a closed
 * form (d <= 2) or a coupled Newton-Schulz iteration in double on the device.  status_dev (device
 * int32, may be NULL) is set non-zero when C is numerically singular.  demean = 0 skips the
 * centring (util.py:130-134).  ASYNC for d < 32; for d >= 32 the iteration's residual words are
 * read back after every batch of steps (one stream synchronisation per 6 steps, usually one). */
int mde_std_retract(int64_t n, int32_t d, float* Z, int32_t demean, double* work,
                    int32_t* status_dev, void* stream);
/* The line search's trial point in one call: Z <- retraction(X + t dir) for the Centered and the
 * Standardized constraint [ref: optim.py:135-136 after lbfgs.py:551: `X += t d` then
 * `constraint.project_onto_constraint(X)`].  At small d the step is folded into the first pass of the
 * retraction (one launch less per trial); results are those of mde_axpy followed by mde_center /
 * mde_std_retract, bit for bit. */
int mde_center_step(int64_t n, int32_t d, const float* X, const float* dir, float t, float* Z, double* work,
                    void* stream);
int mde_std_retract_step(int64_t n, int32_t d, const float* X, const float* dir, float t, float* Z,
                         int32_t demean, double* work, int32_t* status_dev, void* stream);
/* mde_center_step in two halves, for a solve whose rows are sharded across ranks: _begin forms this rank's n rows of
 * Z = X + t dir (dir == NULL: Z as it is) and leaves the column MEANS over those rows in work[0 .. d) (device doubles);
 * the caller sums them across the ranks in place (one small all-reduce); _end subtracts scale x work[0 .. d) from the
 * rows (scale = n / n_total, this rank's share of the rows).  In a world of one: _begin + _end(scale 1) = mde_center_step. */
int mde_center_step_begin(int64_t n, int32_t d, const float* X, const float* dir, float t, float* Z, double* work,
                          void* stream);
int mde_center_step_end(int64_t n, int32_t d, float* Z, double* work, double scale, void* stream);
/* mde_std_tangent(X, Z) followed by mde_vec_stats(Z, dir, X) (dir may be NULL): the projected gradient
 * and the statistics the line search reads, with the statistics folded into the projection's second
 * pass at d <= 4.  ASYNC. */
int mde_std_tangent_stats(int64_t n, int32_t d, const float* X, float* Z, const float* dir, double* stats,
                          double* work, void* stream);
/* Gram matrix out[d_a, d_b] (double, device) = A^T B for A [n,d_a], B [n,d_b]; uses the
 * f32 MFMA path when both widths are multiples of 32. */
int mde_gram(int64_t n, int32_t da, int32_t db, const float* A, const float* B, double* out,
             double* work, void* stream);
/* Z = A M, M [d, d2] device double row-major (small). */
int mde_right_multiply(int64_t n, int32_t d, int32_t d2, const float* A, const double* M,
                       float* out, void* stream);
/* out = base + alpha * A M  (base may be NULL or alias out; out may alias A only when d == d2). */
int mde_right_multiply_add(int64_t n, int32_t d, int32_t d2, const float* A, const double* M,
                           float alpha, const float* base, float* out, void* stream);
/* Z[r, :] += shift (d device doubles): the translation part of util.align [ref: util.py:302-331]. */
int mde_shift_rows(int64_t n, int32_t d, const double* shift, float* Z, void* stream);
/* Z[r, :] *= scale[r]  (row scaling, e.g. the Jacobi preconditioner of the spectral initialiser). */
int mde_row_scale(int64_t n, int32_t d, const float* scale, float* Z, void* stream);
/* out[v] = sum of the per-edge weights (plan / CSR order) over the half-edges of row v: the diagonal
 * of the graph Laplacian [ref: quadratic.py:47-68].  Rows outside the plan's range are untouched. */
int mde_weighted_degree(const mde_plan* plan, const float* w_plan_order, float* out, void* stream);
/* number of doubles of scratch the constraint / gram / vector calls need for width d.  ZERO the
 * buffer once after allocating it: a few of its words are the arrival counters of the kernels that
 * finish their reduction in the last workgroup (they are back at zero after every call).  One
 * buffer serves one stream at a time. */
int64_t mde_work_doubles(int32_t d);

/* ------------------------------------------------------------------ vector kernels
 * [ref: lbfgs.py:350-376, 461-530; optim.py:94-147]  flat float32 vectors of length N */
/* out = y + alpha x  (lbfgs.py:350-357 `_add_grad`);  out may alias y */
int mde_axpy(int64_t N, float alpha, const float* x, const float* y, float* out, void* stream);
/* stats[0..7] (device doubles) = { g.d, g.g, sum|g|, max|g|, #non-finite(g), d.d, max|d|, x.x }
 * (d and x may be NULL: their entries are 0).  One pass.  lbfgs.py:59, 82, 523, 527; optim.py:96, 130 */
int mde_vec_stats(int64_t N, const float* g, const float* d, const float* x, double* stats,
                  double* work, void* stream);

/* ------------------------------------------------------------------ L-BFGS memory
 * [ref: lbfgs.py:461-507]  Device-resident history of (s, y) pairs with one spare slot. */
typedef struct mde_lbfgs mde_lbfgs;
int mde_lbfgs_create(int64_t N, int32_t history, mde_lbfgs** out);
int mde_lbfgs_destroy(mde_lbfgs* o);
int mde_lbfgs_reset(mde_lbfgs* o);                   /* lbfgs.py:378-388 */
int32_t mde_lbfgs_count(const mde_lbfgs* o);
/* Stage the candidate pair y = g - g_prev, s = t d in the spare slot, set g_prev <- g, and
 * compute every inner product the two-loop recursion needs.  dots (device doubles):
 *   [0] y.s  [1] y.y  [2] s.g  [3] y.g
 *   then for each stored pair j (oldest first): s_j.y*, y_j.y*, s*.y_j, s_j.g, y_j.g
 * `dots` must hold 4 + 5*history doubles.  g_prev may be NULL on the first call (nothing is
 * staged; only g_prev <- g is skipped too -- use mde_lbfgs_set_prev). */
int mde_lbfgs_stage(mde_lbfgs* o, const float* g, float* g_prev, const float* d, float t,
                    double* dots, double* work, void* stream);
/* accept != 0: the staged pair becomes the newest pair (dropping the oldest when full). */
int mde_lbfgs_commit(mde_lbfgs* o, int32_t accept);
/* d_out = c_g g + sum_j (cs[j] s_j + cy[j] y_j) over stored pairs (oldest first); also
 * stats as mde_vec_stats(g, d_out, NULL).  cs, cy: HOST arrays of mde_lbfgs_count floats. */
int mde_lbfgs_combine(mde_lbfgs* o, const float* g, float c_g, const float* cs, const float* cy,
                      float* d_out, double* stats, double* work, void* stream);
/* Device-driven variant: the whole direction update of one iteration without a host round trip
 * [ref: lbfgs.py:468-507].  mde_lbfgs_dev_reset empties the history (lbfgs.py:378-388);
 * mde_lbfgs_dev_step stages y = g - g_prev, s = t d (and sets g_prev <- g), accepts the pair iff
 * y.s > 1e-10 (dropping the oldest when the history is full), runs the two-loop recursion in
 * coefficient form on the device and writes the new direction to d_out (may alias d) and the
 * statistics of (g, d_out) to stats as mde_vec_stats(g, d_out, NULL) does.  ASYNC.  Do not mix with
 * the host-driven stage / commit / combine calls on one object.  mde_lbfgs_dev_info reads back the
 * number of stored pairs and whether the last pair was accepted (tests; SYNC). */
int mde_lbfgs_dev_reset(mde_lbfgs* o, void* stream);
int mde_lbfgs_dev_step(mde_lbfgs* o, const float* g, float* g_prev, const float* d, float t,
                       float* d_out, double* stats, double* work, void* stream);
int mde_lbfgs_dev_info(const mde_lbfgs* o, int32_t* count_host, int32_t* accepted_host, void* stream);
/* The same step in two halves, for a solve whose vectors are SHARDED BY ROWS across ranks (every rank keeps the
 * history of its own rows: mde_lbfgs_create(N = its elements)) [ref: lbfgs.py:461-507, the loop being sharded]:
 * mde_lbfgs_dev_stage stages the pair and leaves the mde_lbfgs_dev_dots(o) = 4 + 5 * history inner products over the
 * rank's elements in work[0 ..) (device doubles); the caller sums them across the ranks IN PLACE (one small
 * all-reduce); mde_lbfgs_dev_finish takes the accept / drop-oldest decision and runs the two-loop recursion on the
 * sums (every rank: the same arithmetic on the same numbers) and writes the rank's elements of the new direction to
 * d_out and the statistics of (g, d_out) over ITS elements to stats (partial: see mde_rank_reduce).  In a world of
 * one, stage + finish = mde_lbfgs_dev_step's four-launch form.  ASYNC. */
int mde_lbfgs_dev_stage(mde_lbfgs* o, const float* g, float* g_prev, const float* d, float t, double* work, void* stream);
int mde_lbfgs_dev_finish(mde_lbfgs* o, const float* g, float* d_out, double* stats, double* work, void* stream);
int32_t mde_lbfgs_dev_dots(const mde_lbfgs* o);
/* out[q] = sum -- or max, where bit q of max_mask is set -- over the ranks r (in rank order) of in[r * count + q],
 * q < count <= 64, device doubles; loss_slot >= 0: that result is also written to *loss_out as a float.  Reduces the
 * per-rank records of a sharded solve (statistics boards + loss shares, exchanged with ONE all-gather) in one launch. */
int mde_rank_reduce(int32_t world, int32_t count, uint64_t max_mask, const double* in, double* out,
                    int32_t loss_slot, float* loss_out, void* stream);
/* Which form mde_lbfgs_dev_step takes (process-wide; read from the environment on first use:
 * MDE_LB_UNFUSED, MDE_LB_DEBUG = "blocks,spins,lds_bytes").  For N <= 2^18 the step is ONE launch whose
 * workgroups meet at grid-wide arrival points; an ordinary launch cannot guarantee that they are all
 * resident, so a workgroup that waits longer than `spins` polls gives up and a rescue kernel queued
 * behind redoes the step from the staged sums -- the direction is the four-launch path's either way.
 * unfused != 0 forces the four launches; blocks / spins / lds_bytes > 0 (tests) launch that many
 * workgroups, with that spin limit and that much dynamic LDS each, so that the give-up path can be
 * provoked.  A negative argument leaves that setting as it is.
 * TEST HOOK, not part of the solver's interface: the settings are one unsynchronised process-wide record (do not
 * call it while another thread is inside mde_lbfgs_dev_step), and the dynamic-LDS attribute it sets on the kernel
 * goes back to 0 only when lds_bytes = 0 is passed again. */
int mde_lbfgs_debug_knobs(int32_t unfused, int32_t blocks, int32_t spins, int32_t lds_bytes);

/* ---- the solver's read-back -------------------------------------------------------------------------
 * device -> pinned-host copy on the stream: [loss | status | statistics board] travels in one copy per
 * objective evaluation [ref: the loop accelerated is optim.py:100-175 + lbfgs.py:390-590, which
 * synchronises >= 12 times per iteration].  (Round 2 also exported mde_capture_* to replay an
 * iteration as one HIP graph; measured slower than the launches it replaced, removed in round 3.) */
int mde_copy_to_host(void* dst_host, const void* src_dev, int64_t bytes, void* stream);

/* ---- one steady-state solver iteration as two calls ----------------------------------------------------
 * The usual L-BFGS iteration accepts its first trial point (t = 1).  Everything such an iteration launches
 * -- direction update (mde_lbfgs_dev_step), trial point retraction(X + dir) (mde_center_step /
 * mde_std_retract_step), objective + gradient (mde_average_distortion), tangent projection + statistics,
 * the read-back -- is described once by mde_turn_desc; mde_turn_enqueue launches it, and mde_turn_wait
 * waits for it, applies the strong-Wolfe acceptance test of the first trial [ref: lbfgs.py:44-253, first
 * pass of the bracketing loop: Armijo with c1 and |g_new.d| <= -c2 g.d] and, when the point is accepted
 * and the caller allows it, launches the NEXT iteration at once -- the host's bookkeeping then runs while
 * the GPU works.  A trial that is not accepted is handed back to the caller's line search unchanged.
 * kind: 0 Centered, 1 Standardized.  X[cur] is the accepted iterate, X[1 - cur] receives the trial. */
typedef struct mde_turn_desc {
  mde_plan* plan;
  const mde_func* func;
  int64_t n;
  int32_t d, kind;
  float* X[2];
  float* g;        /* n*d floats, followed by the loss word (loss_dev) */
  float* g_prev;
  float* dir;
  float* loss_dev;
  double* board;   /* statistics board: [0,8) trial statistics, [16,24) direction statistics */
  double* work;
  int32_t* status;
  mde_lbfgs* lbfgs;
  void* host_dst;            /* pinned mirror of [loss | status | pad | board[0,24)] */
  const void* tail_src;
  int64_t read_bytes;
  const float* host_loss;
  const int32_t* host_status;
  const double* host_board;  /* >= 80 doubles of pinned memory: the last kernel of an iteration writes the 24 mirrored board
                                entries, its sequence number behind them (mde_turn_wait polls that word), and -- in the
                                first 64-byte-aligned 32 doubles at or behind host_board + 32 -- the record again as four
                                lines of seven values + the sequence number, which is what mde_turn_wait reads: a line and
                                its tag reach host memory together, the lines in any order */
  double seq;                /* (library state: sequence number of the iteration enqueued last) */
  double pre_id;             /* (library state: > 0 the L-BFGS step of the iteration after it is queued behind it, -1 that step has run and is pending) */
} mde_turn_desc;
/* f0: the loss at X[cur]; the acceptance test of the trial (c1, c2) is made by the iteration's last kernel,
 * on the device.  allow_pre: also queue the L-BFGS step of the iteration AFTER this one behind it -- its
 * kernels leave at once unless that test accepts the trial (the host's turn-around between two iterations
 * then overlaps with that step); pass 0 when the solve would stop after this iteration anyway. */
int mde_turn_enqueue(mde_turn_desc* T, int32_t cur, float t_prev, double f0, double c1, double c2,
                     int32_t allow_pre, void* stream);
/* out (>= 24 doubles): [0] f_new, [1] accepted, [2] next iteration enqueued, [3] status word,
 * [4,12) trial statistics, [12,20) direction statistics, [20] 1 when the gated L-BFGS step of the NEXT iteration
 * has run (the trial was accepted and the iteration was enqueued with allow_pre) but allow_next = 0 kept that
 * iteration from being launched: the step is then PENDING -- g_prev, the history, dir and out[12,20) are already
 * those of the next iteration, the next mde_turn_enqueue on this descriptor skips its own step (t_prev = 1 is
 * what the pending step used), and mde_lbfgs_dev_step must not be called on the same history in between.
 * eps_pre >= 0: the iteration enqueued here gets
 * allow_pre when the accepted point's gradient norm is above eps_pre; < 0: never.  SYNC (waits for the
 * stream). */
int mde_turn_wait(mde_turn_desc* T, int32_t cur, double f0, int32_t allow_next, double c1, double c2,
                  double eps_pre, double* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDE_HIP_H_ */
