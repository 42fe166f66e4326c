/*
 * include/rtrec_amd.h -- C-ABI of the MI355X-native SLIM engine (librtrec_amd.so).
 *
 * rtrec (the reference) is pure Python and has no FFI of its own; the interface these
 * entry points replace is the method set of rtrec.models.internal.slim_elastic.SLIMElastic
 * and the third-party native code it drives.  Each function cites
 * the reference lines it stands in for.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - Every pointer named d_* is a DEVICE pointer (HBM of the current HIP device).  Nothing
 *    here allocates, frees or synchronises; the caller owns all buffers and the stream.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *    enqueued on it and is stream-ordered.
 *  - Return value: 0 on success, negative rtrec_status otherwise.  No exceptions cross the
 *    boundary.  The library keeps no mutable state of its own and reads no environment variables:
 *    everything a call depends on is in its arguments (optional ones in the rtrec_*_opts structs, a
 *    timing bracket in a caller-owned rtrec_timer object), so calls on different streams / threads
 *    are independent as long as they do not share buffers.  (The one thing it remembers is errno-style and
 *    thread-local: the HIP error code behind the calling thread's last RTREC_ERR_LAUNCH, for rtrec_amd_last_error.)
 *  - Index arrays are int32, values float32 (rtrec/utils/interactions.py:276,303 builds
 *    float32 matrices whose scipy index dtype is int32).
 */
#ifndef RTREC_AMD_H
#define RTREC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    RTREC_OK = 0,
    RTREC_ERR_INVALID_ARG = -1,   /* NULL pointer, negative size, inconsistent shapes  */
    RTREC_ERR_UNSUPPORTED = -2,   /* parameter outside what the kernels were built for */
    RTREC_ERR_WORKSPACE = -3,     /* workspace too small (see *_workspace_bytes)       */
    RTREC_ERR_LAUNCH = -4         /* hipGetLastError() != hipSuccess after a launch    */
} rtrec_status;

/* ElasticNet hyper-parameters as they reach sklearn's Cython solver
 * (slim_elastic.py:197-208 -> _coordinate_descent.py:653-654 -> _cd_fast.pyx:276). */
typedef struct {
    float    l1_reg;        /* (float)(alpha * l1_ratio * n_users)        */
    float    l2_reg;        /* (float)(alpha * (1 - l1_ratio) * n_users)  */
    float    tol;           /* (float)tol                                  */
    int32_t  max_iter;
    uint32_t seed;          /* RandomState(random_state).randint(0, 2**31-1); 43 -> 494155588 */
    int32_t  positive;      /* positive_only                               */
    int32_t  top_features;  /* nn_feature_selection; <= 0: every item is a feature */
} rtrec_fit_cfg;

/* Library identification; returns a static string such as "rtrec_amd 0.1 gfx950". */
const char *rtrec_amd_version(void);

/* hipGetErrorString() of the HIP error behind the calling thread's most recent
 * RTREC_ERR_LAUNCH (diagnostics only). */
const char *rtrec_amd_last_error(void);

/* ---------------------------------------------------------------------------------------
 * FIT  (replaces slim_elastic.py:229-281 fit, :283-454 fit_in_parallel/_fit_items,
 *       :510-564 partial_fit_items, :139-154 FeatureSelectionWrapper.fit, and sklearn
 *       _cd_fast.pyx:276-561 sparse_enet_coordinate_descent)
 * ------------------------------------------------------------------------------------- */

/* Per-column sequential float32 sum of squares of a CSC matrix (norm_cols_X of
 * _cd_fast.pyx:394-410).  d_sqnorm[n_items]. */
int rtrec_slim_column_sqnorms(int32_t n_items, const int32_t *d_csc_ptr, const float *d_csc_val,
                              float *d_sqnorm, void *stream);

/* Left-to-right float32 sums of n_sums segments of d_values: d_out[s] = (...((0 + v[o_s]) + v[o_s + 1]) + ...),
 * o = d_offsets[n_sums + 1] -- the accumulation order of the reference's dot products (_cd_fast.pyx:464-466
 * `tmp += R[X_indices[jj]] * X_data[jj]`, :506-509 XtA; scipy csr_matvec).  mode 0: the binade-speculative fold the
 * fit kernels use (csrc/fold_spec.hip.h: integer prefix sums inside a binade, bit-identical to the chain), one
 * 256-entry group per step; mode 2 / 3: the same with 2 / 4 groups per step (the running value stays an integer
 * from group to group); mode 1: the literal chain of dependent float additions.  One wave per segment. */
int rtrec_slim_ordered_sums(const float *d_values, const int64_t *d_offsets, int32_t n_sums, int32_t mode,
                            float *d_out, void *stream);

/* Bytes of scratch rtrec_slim_fit_columns needs for `n_slots` concurrently fitted targets. */
size_t rtrec_slim_fit_workspace_bytes(int32_t n_users, int32_t n_items, int32_t n_slots,
                                      int32_t top_features);

/* One-time initialisation of a fit workspace (sets the scratch invariants the kernel keeps
 * between targets).  Must be called once after allocation, before the first fit call. */
int rtrec_slim_fit_workspace_init(void *d_workspace, size_t workspace_bytes,
                                  int32_t n_users, int32_t n_items, int32_t n_slots,
                                  int32_t top_features, void *stream);

/* Fit `n_targets` item columns of the U x I interaction matrix X.
 *   X is given twice: CSC (d_csc_*: ptr[I+1], row[nnz], val[nnz], rows ascending per column)
 *   and CSR (d_csr_*: ptr[U+1], col[nnz], val[nnz], columns ascending per row).  Inputs are
 *   never modified: the target column is masked by index instead of being zeroed in place
 *   (slim_elastic.py:266/438 zero and restore it).
 *   d_targets[n_targets]: target column ids, processed in the given order by a device-side
 *   work queue (put long columns first for balance).
 * Outputs, slot t = position in d_targets, `cap` entries per target:
 *   top_features > 0 : cap >= min(top_features, I).  d_out_items[t*cap + p] is the p-th
 *       selected feature (descending X^T y, ties -> higher id), d_out_coef the matching
 *       coefficient -- i.e. model.sparse_coef_ INCLUDING explicit zeros (slim_elastic.py:153);
 *       d_out_count[t] = min(top_features, I).
 *   top_features <= 0: any cap >= 1.  Non-zero coefficients only, ascending item id
 *       (sparse.csr_matrix(coef_), _coordinate_descent.py:1133-1136); d_out_count[t] = nnz.  When nnz
 *       exceeds cap only the first cap entries are stored: refit that target with cap >= nnz
 *       (cap = I can never overflow).
 *   d_out_n_iter[t] = sklearn's n_iter_.
 * d_queue: one int32 work-queue counter, must be zero on entry (the call resets it to zero
 * on the stream before launching). */
int rtrec_slim_fit_columns(int32_t n_users, int32_t n_items,
                           const int32_t *d_csc_ptr, const int32_t *d_csc_row, const float *d_csc_val,
                           const int32_t *d_csr_ptr, const int32_t *d_csr_col, const float *d_csr_val,
                           const float *d_sqnorm,
                           const int32_t *d_targets, int32_t n_targets,
                           const rtrec_fit_cfg *cfg,
                           int32_t *d_out_items, float *d_out_coef, int32_t *d_out_count,
                           int32_t *d_out_n_iter, int32_t cap,
                           void *d_workspace, size_t workspace_bytes, int32_t n_slots,
                           int32_t *d_queue, void *stream);

/* Gram matrix of `n_top` item columns for rtrec_fit_opts.d_gram: d_gram[a * P64 + b] = X[:, top[a]] .
 * X[:, top[b]] accumulated in float64 (exact products, relative error of the sums <= ~n_users 2^-53),
 * P64 = n_top rounded up to a multiple of 64 = the row stride to pass as rtrec_fit_opts.gram_n; rows /
 * columns beyond n_top are zero.  d_gram holds P64 * P64 doubles; the workspace (a dense float32 copy of
 * the columns) is rtrec_slim_gram_workspace_bytes(n_users, n_top). */
size_t rtrec_slim_gram_workspace_bytes(int32_t n_users, int32_t n_top);
int rtrec_slim_gram_matrix(int32_t n_users, int32_t n_items,
                           const int32_t *d_csc_ptr, const int32_t *d_csc_row, const float *d_csc_val,
                           const int32_t *d_top_items, int32_t n_top,
                           void *d_workspace, size_t workspace_bytes, double *d_gram, void *stream);

/* Optional inputs / outputs of a fit call. */
typedef struct {
    /* per-target trace for profiling (tools/fit_trace.py), or NULL: d_trace[t*8 + 0..3] = start, end of
     * the X^T y / feature-selection step, end of the target (ticks of the 100 MHz constant clock) and
     * the number of column entries folded in order by the coordinate descent; [4..7] are kernel-
     * specific phase clocks (every-item path: draw generation, lane-private dots, commit loop, gap;
     * latency-mode kernel: ordered folds, residual updates, duality gaps, s_memtime cycles in the folds) */
    int64_t       *d_trace;
    /* Gram matrix of the gram_n most popular items, or NULL: d_gram[a * gram_n + b] = X[:, item_a] .
     * X[:, item_b] in float64 with relative error <= gram_rel_err (< 1e-6), d_gram_index[i] = row of item
     * i or -1.  ONLY for a non-negative X: the kernel then tracks X_p . R of zero coordinates through
     * the coordinate updates (D_p -= dw G_pq, with rigorous rounding bounds) and skips their passes
     * over memory while the interval decides; results are unchanged (csrc/fit.hip, "Gram tracking"). */
    const double  *d_gram;
    const int32_t *d_gram_index;
    int32_t        gram_n;
    double         gram_rel_err;
    /* Tolerance mode (0 = exact, the default).  The exact mode folds every dot product in the reference's
     * left-to-right float32 order: coefficients and n_iter_ are bit-identical to scikit-learn's.  The tolerance
     * modes keep the algorithm (same X^T y and feature selection, same coordinate sequence, same stopping rules)
     * but let go of that order:
     *   1  the dot products are tree-reduced (64 lane partial sums + a shuffle tree); the float32 residual is
     *      updated exactly as the reference updates it.  Coefficients agree to ~1e-6 relative.
     *   2  in addition a target whose features all have a row in d_gram is solved in Gram form (float64 state,
     *      no residual: sklearn's `precompute` solver).  The float32 reference itself carries ~1e-5 of
     *      accumulated rounding in its residual, so this mode agrees with it to a few 1e-5 relative.
     * In either mode a stopping test that flips by a rounding can cost a target one sweep (a change of the order
     * of the solver's own tolerance `tol`). */
    int32_t        fast;
    /* Tuning / test knobs; 0 selects the built-in default.
     *   kernel            1: one wave per target (throughput), 2: one 8-wave workgroup per target (latency);
     *                     default: latency mode for calls of <= 2048 targets
     *   colwalk_min_rows  latency kernel: targets with at least this many users take the column-walk X^T y
     *   screen_min        columns with at least this many entries are screened before an ordered fold
     *   lane_max          every-item path: columns up to this length are folded one per lane (< 0: none) */
    int32_t        kernel;
    int32_t        colwalk_min_rows;
    int32_t        screen_min;
    int32_t        lane_max;
    /* Optional scratch for calls of <= 2048 targets with feature selection (online partial_fit): with it the call
     * computes X^T y of ALL its targets in one pass over X (xty_batch_kernel: a wave per item column, the sums of every
     * target in LDS, products folded in the reference's ascending-user order) instead of one walk per target; results
     * are unchanged.  rtrec_slim_xty_workspace_bytes(n_users, n_items, nnz, n_targets) bytes, nnz = stored entries of X. */
    /* PRECONDITION with d_xty_ws: the call's targets are DISTINCT (the one-pass X^T y maps a target to one slot of the call;
     * a repeated target would be fitted without features).  Callers that may repeat a target pass NULL here. */
    void          *d_xty_ws;
    size_t         xty_ws_bytes;
    int64_t        nnz;
    const int32_t *d_col_order;   /* optional int32[n_items]: item ids by descending column length (the one-pass X^T y
                                     starts its longest columns first); any permutation gives the same results */
    /* How the ordered dot products are evaluated (results are bit-identical either way; A/B runs and tests):
     *   0, 1  the literal chain of dependent float additions (the default: measured faster inside both fit kernels,
     *         DESIGN.md section 3.4 "the binade-speculative fold")
     *   2     the binade-speculative fold (csrc/fold_spec.hip.h: integer prefix sums inside a binade, one real float
     *         addition where the running sum changes binade) for every column of at least 64 entries (tests: small
     *         matrices exercise it)
     *   3     the speculative fold for columns of at least 512 entries */
    int32_t        fold;
} rtrec_fit_opts;

size_t rtrec_slim_xty_workspace_bytes(int32_t n_users, int32_t n_items, int64_t nnz, int32_t n_targets);

/* rtrec_slim_fit_columns with options; opts == NULL behaves exactly like rtrec_slim_fit_columns. */
int rtrec_slim_fit_columns_opt(int32_t n_users, int32_t n_items,
                               const int32_t *d_csc_ptr, const int32_t *d_csc_row, const float *d_csc_val,
                               const int32_t *d_csr_ptr, const int32_t *d_csr_col, const float *d_csr_val,
                               const float *d_sqnorm,
                               const int32_t *d_targets, int32_t n_targets,
                               const rtrec_fit_cfg *cfg,
                               int32_t *d_out_items, float *d_out_coef, int32_t *d_out_count,
                               int32_t *d_out_n_iter, int32_t cap,
                               void *d_workspace, size_t workspace_bytes, int32_t n_slots,
                               int32_t *d_queue, void *stream, const rtrec_fit_opts *opts);

/* ---------------------------------------------------------------------------------------
 * SCORE + TOP-K  (replaces slim_elastic.py:566-626 predict*, :628-741 recommend /
 *       recommend_batch, :744-818 _dense/_sparse_topk_indicies and scipy's csr_matmat)
 *
 * W (item_similarity, I x I) is consumed in a column-tiled CSR layout: the shard's columns
 * [col_offset, col_offset + n_cols) are cut into tiles of `tile_cols` columns; tile t holds a
 * CSR over ALL I rows restricted to its columns:
 *     d_tile_ptr[t * (n_items + 1) + i] .. [.. + i + 1]  -> range in d_w_col / d_w_val
 *     d_w_col[e] = column index LOCAL to the tile (0 .. tile_cols-1), ascending per row
 *     d_w_val[e] = float32 weight
 * so a user row is accumulated tile by tile in LDS in exactly scipy's csr_matmat order
 * (ascending row item, then ascending column).
 * The layout may be COMPACTED to the columns that hold at least one weight (only those can
 * score != 0, i.e. only those can be recommended in SPARSE mode): then d_col_ids[n_cols] maps a
 * layout column to its global item id (ascending) and d_col_map[n_items] maps a global item id
 * to its layout column or -1; pass both NULL for the plain layout whose column c is item
 * col_offset + c.
 * Well-filled row segments may be stored DENSE instead: d_dense_idx[t * n_items + i] >= 0 names
 * a block of tile_cols floats in d_dense_val (zero where W[i, c] is not stored) and the CSR range
 * of that (tile, row) is then empty; pass both NULL when no segment is dense.  Adding x * 0 never
 * changes an accumulator, so results are identical; dense blocks are updated 4 columns per lane.
 * Optionally the per-item lookups are given once more as ONE record table, d_row_hdr[(t * n_items + i) * 4
 * + 0..3] = { d_tile_ptr[t][i], d_tile_ptr[t][i + 1], dense block or -1, tile-local layout column of
 * item i if it lies in tile t else -1 } (16-byte aligned): scoring a user row then costs one memory
 * sector per item instead of three.  NULL = use the separate tables.
 * ------------------------------------------------------------------------------------- */

typedef enum {
    RTREC_TOPK_SPARSE = 0,   /* int ids: only non-zero sums compete (slim_elastic.py:782-818)  */
    RTREC_TOPK_DENSE = 1,    /* str ids: every column competes        (slim_elastic.py:744-779)  */
    RTREC_TOPK_CANDIDATES = 2 /* candidate_item_ids given: d_col_rank[c] >= 0 marks candidates,
                                 interacted items are NOT filtered     (slim_elastic.py:661-672) */
} rtrec_topk_mode;

/* Measurement hook (bench.py's roofline leg): a caller-owned pair of HIP events that a scoring call records on
 * its launch stream around its dominant kernel when the object is passed in rtrec_score_opts.timer.
 * rtrec_timer_read waits for the last recorded pair, adds it up and returns the total kernel time / the
 * number of brackets collected so far (either pointer may be NULL); reset != 0 zeroes the totals afterwards. */
int  rtrec_timer_create(void **out_timer);
int  rtrec_timer_read(void *timer, double *total_ms, int64_t *launches, int32_t reset);
void rtrec_timer_destroy(void *timer);

/* Optional inputs of a scoring call. */
typedef struct {
    /* Rows of the CSR matrix d_xb_*: a job whose row id is outside [0, n_x_rows) scores as an EMPTY row
     * instead of reading out of bounds.  <= 0: unknown, no check (the caller vouches for the ids). */
    int32_t        n_x_rows;
    /* "Feature-row" form of the shard (optional; used for SPARSE mode with float32 accumulation and
     * top_k <= 15, ignored otherwise).  Only items that some column selected with a non-zero weight have a
     * row in W; when those rows are few (fr_rows <= 128) the shard is also given as the dense matrix of
     * those rows over the columns that hold a weight, in an order of the caller's choice:
     *     d_fr_col_ids[n_cols]  layout column -> item id,   d_fr_col_map[n_items]  item id -> layout column or -1
     * (n_cols as in the compacted tiled layout), cut into fr_n_tiles = ceil(n_cols / fr_tile_cols) tiles of
     * fr_tile_cols (256 or 128) columns.  d_fr_map[n_items] = row f of an item or -1.  Only the rows that hold a
     * weight in a tile are stored: the tile's SLICE is a 512-byte header (max |w| of each row over the tile's columns,
     * one float per row: sum_f |x_f| max|w_f| bounds every score a user can have in the tile, and a tile that cannot
     * beat the user's current (k+1)-th best is skipped) and those rows in ascending order, fr_tile_cols floats each
     * (0 where W[row, column] is not stored).  Blocks without a weight are never visited, so an order of the columns that
     * clusters the rows' weights saves work without changing a sum.  For staging in LDS a slice is cut into
     * FRAGMENTS (consecutive rows of it; fr_n_frags in all, in tile order): d_fr_frag_tile[g] = tile | first << 24 |
     * last << 25, d_fr_tile_rows[g * 2 + h] bit f = 1 iff the fragment holds row 64 h + f, d_fr_tile_off[g] = its byte
     * offset inside its super-tile.  Consecutive fragments (at most 64) form fr_n_super super-tiles, fragments
     * d_fr_super_tile[s] .. d_fr_super_tile[s + 1] - 1, each staged as one piece: d_fr_w holds the super-tiles back
     * to back, each a whole number of KiB starting at KiB d_fr_super_kb[s] (fr_n_super + 1 entries both).
     * fr_buf_bytes (a multiple of 1024, >= the largest super-tile) is the size of one LDS staging buffer.  Two forms:
     *   streaming  fr_buf_bytes >= 32 KiB and 2 * (2 * fr_buf_bytes + 5136) <= 160 KiB: two 8-wave workgroups per CU,
     *              two buffers each (a tile may continue in the next super-tile: its sums stay in registers);
     *   resident   fr_n_super == 1, fr_n_tiles <= 64 and everything fits next to the per-wave setup scratch
     *              (fr_buf_bytes + 16 * ceil256(fr_n_tiles * fr_tile_cols / 8 + 768) + 9232 <= 160 KiB): one 16-wave
     *              workgroup per CU loads W once and keeps it.
     * d_fr_scratch: rtrec_slim_score_fr_scratch_bytes() bytes of device scratch.  Scores and ids are identical to
     * the tiled-CSR path; accumulators live in registers and the matrix is streamed through LDS
     * (csrc/score.hip, score_frows_kernel). */
    const int32_t *d_fr_map;
    const int32_t *d_fr_col_ids;
    const int32_t *d_fr_col_map;
    const float   *d_fr_w;
    const uint64_t *d_fr_tile_rows;
    const int32_t *d_fr_tile_off;
    const int32_t *d_fr_super_kb;
    const int32_t *d_fr_super_tile;
    const int32_t *d_fr_frag_tile;
    int32_t        fr_rows, fr_tile_cols, fr_n_tiles, fr_n_frags, fr_n_super, fr_buf_bytes;
    void          *d_fr_scratch;
    size_t         fr_scratch_bytes;
    /* Optional work order for the feature-row kernel: a permutation of 0 .. n_rows-1; job position p scores row
     * d_row_order[p] (outputs stay at row index d_row_order[p]).  Handing the rows over longest-first levels the
     * kernel's waves and shortens its tail; results do not depend on it. */
    const int32_t *d_row_order;
    void          *timer;         /* rtrec_timer object or NULL */
    int32_t        diagnostics;   /* bits 0-7: ablation switches of tools/score_ablate.sh; honoured by diagnostic builds
                                     (-DRTREC_DIAGNOSTICS) only, ignored by the release library.  bits 8-11: users per wave
                                     of the feature-row kernel (8, 4 or 2; 0 = chosen from the batch size); bits 12-23: v > 0
                                     = the segment path gives users of more than v - 1 items (at most 512) a workgroup of
                                     their own instead of a wave (0 = chosen from the batch size) -- test / tuning knobs,
                                     results do not depend on them */
    int32_t       *d_rescored;    /* optional int32[1] on the device: receives the number of rows the exact-tie pass
                                     re-scored (SPARSE mode; rows whose fast-pass list held an exact tie reaching its
                                     (k+1)-th entry -- ties inside the leading k are ordered in place and not counted) */
    int32_t        row_order_grouped;  /* d_row_order is sorted by similarity (rows that rate the same rows of W are
                                     neighbours): a wave then takes eight CONSECUTIVE positions instead of a strided deal */
    /* "Segment" form of the shard (optional; the fast path for a GENERAL W -- any number of non-empty rows -- in SPARSE
     * mode with float32 accumulation and top_k <= 63; used when the feature-row form above is absent or does not apply).
     * The same compacted columns as the tiled layout (sg_n_cols == n_cols) in an order of the caller's choice --
     * rtrec_amd/seg_layout.py orders them by cluster (label propagation over W's graph) so that the columns a row of W
     * weights sit together -- cut into sg_n_tiles = ceil(n_cols / sg_tile_cols) <= 128 tiles of sg_tile_cols columns
     * (a power of two, 256 .. 4096):
     *     d_sg_col_ids[n_cols]      layout column -> item id
     *     d_sg_info[n_items][2]     item -> { its row r in the tables below or -1 (no weight in its row of W),
     *                               its layout column or -1 }     (8-byte aligned)
     *     d_sg_ptr[sg_rows][sg_n_tiles + 1]   the entries of row r that fall into tile t are records d_sg_ptr[r][t] ..
     *                               d_sg_ptr[r][t + 1] - 1 of d_sg_ent[sg_nnz][2] = { column INSIDE the tile (ascending),
     *                               float32 bits of the weight }   (8-byte aligned, sg_nnz < 2^28)
     *     d_sg_bound[sg_rows][64]   word l = bfloat16(max |w| of row r in tile 2 l) | bfloat16(... tile 2 l + 1) << 16,
     *                               each ROUNDED UP (0 for an empty segment / a tile beyond sg_n_tiles)
     * sum_i |x_ui| bound[i][t] bounds every score user u can have in tile t: the kernel opens a user's tiles in
     * descending bound order and stops when the best remaining bound cannot beat the user's current (k+1)-th best
     * score.  Opened tiles are accumulated in scipy's csr_matmat order (ascending item, one rounded product and one
     * rounded add per entry), so ids, scores and counts are identical to the tiled-CSR path
     * (csrc/score_seg.hip.h, score_seg_kernel).  d_row_order applies to this kernel as well. */
    const int32_t *d_sg_info;
    const int32_t *d_sg_ptr;
    const uint32_t *d_sg_ent;
    int64_t        sg_nnz;
    const uint32_t *d_sg_bound;
    const int32_t *d_sg_col_ids;
    int32_t        sg_tile_cols, sg_n_tiles, sg_rows, sg_n_cols;
    /* Heavy pass of the segment kernel (optional; all three or none).  A user with more items than a wave's LDS lists hold
     * (512 items) is scored by a whole workgroup from the TILE's side:
     *     d_sg_trow_ptr[sg_n_tiles + 1], d_sg_trow[...][4] = { item, begin, end, 0 }   the non-empty segments of tile t,
     *                               ascending item (16-byte aligned)
     *     d_sg_scratch              rtrec_slim_score_sg_scratch_bytes() bytes, ZERO when the call starts; the kernel
     *                               leaves it zero (a dense ratings vector and column flags per workgroup)
     * Without them such users are scored by the per-user wave as well, re-reading their items per tile (slow, same answer). */
    const int32_t *d_sg_trow_ptr;
    const int32_t *d_sg_trow;
    void          *d_sg_scratch;
    size_t         sg_scratch_bytes;
    int32_t        row_order_longest_first;   /* d_row_order is sorted by row length, longest first: the heavy pass then finds
                                     the long users at its head instead of walking all rows */
    /* Scoring WITHOUT the tiled layout (d_tile_ptr == NULL; SPARSE mode, float32 accumulation, a usable feature-row or
     * segment form above): the fast pass runs alone and, instead of re-scoring the rows whose lists hold an exact score tie,
     * reports them: d_flagged[0] = their number, d_flagged[1 ..] = the rows (room for 1 + n_rows int32).  The caller scores
     * exactly those rows again WITH the tiled layout (whose exact-tie pass orders them like the reference) -- and has to
     * build that layout only when a call flags something.
     * DENSE mode (mode = RTREC_TOPK_DENSE, every column competes -- slim_elastic.py:745-778) may be asked for in the same way:
     * the fast pass then also flags every row whose leading top_k scores are not all POSITIVE (positives outrank every
     * zero-score column and zeros outrank negatives, so any other row's list is final) and every tie (DENSE mode orders ties
     * by item id, not by first touch); the flagged rows are the caller's to score with the tiled layout in DENSE mode. */
    int32_t       *d_flagged;
    /* Optional second HIP stream of the caller (NULL: none).  The segment path then runs its workgroup-per-long-user kernel
     * on it, beside the main kernel, for passes of 8192 rows or more: forked from and joined to `stream` by events inside the
     * call, so the caller sees the usual stream-ordered semantics on `stream` and need not synchronise anything itself. */
    void          *aux_stream;
} rtrec_score_opts;

size_t rtrec_slim_score_fr_scratch_bytes(int32_t fr_n_tiles, int32_t fr_tile_cols);
size_t rtrec_slim_score_sg_scratch_bytes(int32_t n_items, int32_t sg_n_tiles, int32_t sg_tile_cols);

/* Bytes of scratch for rtrec_slim_score_topk. */
size_t rtrec_slim_score_workspace_bytes(int32_t n_rows, int32_t n_tiles, int32_t top_k);

/* Score `n_rows` user rows of the CSR matrix d_xb_* (ptr, col, val; columns ascending) against
 * the shard and return each row's local top-k.  Job r scores CSR row d_row_ids[r] (or row r
 * when d_row_ids is NULL), so a device-resident interaction matrix is scored in place without
 * slicing it on the host (the reference slices: slim_elastic.py:707):
 *   d_out_ids[n_rows*top_k]   global column ids, -1 padded
 *   d_out_scores[n_rows*top_k] float32 scores (float64 accumulations are rounded on output),
 *                             -inf padded;  d_out_scores64 (may be NULL) receives the float64
 *                             values when acc_f64 != 0
 *   d_out_aux[n_rows*top_k]   tie-break key that travels with each entry (first-touch rank in
 *                             SPARSE mode, candidate rank in CANDIDATES mode, 0 otherwise) so
 *                             shards can be merged with rtrec_slim_merge_topk
 *   d_out_count[n_rows]       number of valid entries (<= top_k)
 * Order: score descending; ties resolved exactly as the reference resolves them in SPARSE
 * mode (stable sort over scipy's reverse-first-touch product order), by higher column id in
 * DENSE mode and by higher candidate rank in CANDIDATES mode (numpy's unstable argsort leaves
 * those two unspecified -- DESIGN.md D1).
 * acc_f64 != 0 accumulates in float64 (W built by the serial SLIMElastic.fit is a float64
 * matrix, slim_elastic.py:252).
 * Limits: top_k <= 1023 and n_tiles * (top_k + 1) <= 1024 (RTREC_ERR_UNSUPPORTED otherwise); up to
 * top_k = 63 the selection is a two-pass threshold filter, beyond that one scan per result. */
int rtrec_slim_score_topk(int32_t n_rows, const int32_t *d_row_ids,
                          const int32_t *d_xb_ptr, const int32_t *d_xb_col, const float *d_xb_val,
                          int32_t n_items, int32_t n_cols, int32_t col_offset,
                          const int32_t *d_col_ids, const int32_t *d_col_map,
                          int32_t tile_cols, int32_t n_tiles,
                          const int32_t *d_tile_ptr, const uint16_t *d_w_col, const float *d_w_val,
                          const int32_t *d_dense_idx, const float *d_dense_val,
                          const int32_t *d_row_hdr,
                          const int32_t *d_col_rank,
                          int32_t top_k, int32_t filter_interacted, int32_t mode, int32_t acc_f64,
                          int32_t *d_out_ids, float *d_out_scores, double *d_out_scores64,
                          uint32_t *d_out_aux, int32_t *d_out_count,
                          void *d_workspace, size_t workspace_bytes, void *stream);

/* rtrec_slim_score_topk with options; opts == NULL behaves exactly like rtrec_slim_score_topk. */
int rtrec_slim_score_topk_opt(int32_t n_rows, const int32_t *d_row_ids,
                              const int32_t *d_xb_ptr, const int32_t *d_xb_col, const float *d_xb_val,
                              int32_t n_items, int32_t n_cols, int32_t col_offset,
                              const int32_t *d_col_ids, const int32_t *d_col_map,
                              int32_t tile_cols, int32_t n_tiles,
                              const int32_t *d_tile_ptr, const uint16_t *d_w_col, const float *d_w_val,
                              const int32_t *d_dense_idx, const float *d_dense_val,
                              const int32_t *d_row_hdr,
                              const int32_t *d_col_rank,
                              int32_t top_k, int32_t filter_interacted, int32_t mode, int32_t acc_f64,
                              int32_t *d_out_ids, float *d_out_scores, double *d_out_scores64,
                              uint32_t *d_out_aux, int32_t *d_out_count,
                              void *d_workspace, size_t workspace_bytes, void *stream,
                              const rtrec_score_opts *opts);

/* Score-vector export (replaces slim_elastic.py:566-626 predict / predict_selected / predict_all):
 * d_out[r * out_stride + c] = sum_i X[row_r, i] * W[i, col_offset + c] for the n_cols columns of a
 * PLAIN (not compacted, no dense blocks) tiled layout, accumulated in csr_matmat order;
 * d_out is float32, or float64 when acc_f64 != 0. */
int rtrec_slim_score_rows(int32_t n_rows, const int32_t *d_row_ids,
                          const int32_t *d_xb_ptr, const int32_t *d_xb_col, const float *d_xb_val,
                          int32_t n_items, int32_t n_cols, int32_t col_offset,
                          int32_t tile_cols, int32_t n_tiles,
                          const int32_t *d_tile_ptr, const uint16_t *d_w_col, const float *d_w_val,
                          int32_t acc_f64, void *d_out, int64_t out_stride, void *stream);

/* Merge `n_lists` per-shard top-k lists per row (layout [n_lists][n_rows][top_k], as produced
 * by an all-gather of rtrec_slim_score_topk outputs) into one top-k per row, using the same
 * (score, aux, id) order.  d_in_scores64 may be NULL (then float32 scores are compared). */
int rtrec_slim_merge_topk(int32_t n_rows, int32_t n_lists, int32_t top_k,
                          const int32_t *d_in_ids, const float *d_in_scores, const double *d_in_scores64,
                          const uint32_t *d_in_aux, const int32_t *d_in_count,
                          int32_t *d_out_ids, float *d_out_scores, int32_t *d_out_count,
                          void *stream);

/* The same merge over lists that live inside a larger buffer -- e.g. one all-gathered block of
 * packed per-user records [n_lists][n_rows][scores | ids | aux | count].  Strides are in ELEMENTS of
 * the respective array: entry r of list l for row u is at  l * list_stride + u * row_stride + r
 * (ids, float32 scores and aux share list_stride/row_stride; the float64 scores and the counts
 * have their own). */
int rtrec_slim_merge_topk_strided(int32_t n_rows, int32_t n_lists, int32_t top_k,
                                  const int32_t *d_in_ids, const float *d_in_scores,
                                  const double *d_in_scores64, const uint32_t *d_in_aux,
                                  const int32_t *d_in_count,
                                  int64_t list_stride, int64_t row_stride,
                                  int64_t score64_list_stride, int64_t score64_row_stride,
                                  int64_t count_list_stride, int64_t count_row_stride,
                                  int32_t *d_out_ids, float *d_out_scores, int32_t *d_out_count,
                                  void *stream);

/* ---------------------------------------------------------------------------------------
 * SIMILAR ITEMS  (replaces slim_elastic.py:820-857)
 * W in CSC (d_wc_ptr[I+1], d_wc_row, d_wc_val).  For each query item: stored entries of its
 * column, the item itself dropped, top_k by score descending (ties: lower row id first).
 * Outputs [n_queries*top_k], -1 / -inf padded, and counts.
 * ------------------------------------------------------------------------------------- */
int rtrec_slim_similar_topk(int32_t n_queries, const int32_t *d_queries,
                            const int32_t *d_wc_ptr, const int32_t *d_wc_row, const float *d_wc_val,
                            int32_t top_k,
                            int32_t *d_out_ids, float *d_out_scores, int32_t *d_out_count,
                            void *stream);

/* ---------------------------------------------------------------------------------------
 * INTERACTION STORE, host side  (replaces the per-interaction dict updates of
 * rtrec/utils/interactions.py:81-119 for the columnar store of rtrec_amd/utils/interactions.py)
 * HOST pointers.  Merge two sorted blocks (key = user << 32 | item, each with distinct keys) into
 * out_* (room for n_a + n_b entries); on equal keys the entry of b wins.  Returns the number of
 * entries written, or -1 on invalid arguments.  n_threads <= 0: up to 16 hardware threads.
 * ------------------------------------------------------------------------------------- */
int64_t rtrec_store_merge_sorted(const int64_t *a_key, const double *a_val, const double *a_ts, int64_t n_a,
                                 const int64_t *b_key, const double *b_val, const double *b_ts, int64_t n_b,
                                 int64_t *out_key, double *out_val, double *out_ts, int32_t n_threads);

/* Positions of ASCENDING needles[m] in the sorted hay[n] (host pointers): pos[i] = lower bound of needles[i]
 * clamped to n - 1, found[i] = (hay[pos[i]] == needles[i]).  0 on success, -1 on invalid arguments. */
int rtrec_store_find_sorted(const int64_t *hay, int64_t n, const int64_t *needles, int64_t m,
                            int64_t *pos, uint8_t *found, int32_t n_threads);

/* Hot-item bookkeeping (replaces one LRUFreqSet.add per interaction, rtrec/utils/lru.py:11-60): replay
 * values[n] in order on a capacity-bounded recency list with hit counts.  The list comes in
 * (state_keys/state_counts[n_state]) and goes out (out_keys/out_counts, room for `capacity`) oldest first;
 * all keys are ids in [0, id_bound).  Returns the length of the list, or -1 on invalid arguments. */
int64_t rtrec_lru_replay(const int64_t *state_keys, const int64_t *state_counts, int64_t n_state,
                         const int64_t *values, int64_t n, int64_t capacity, int64_t id_bound,
                         int64_t *out_keys, int64_t *out_counts);

/* One round of a batch on distinct (user, item) pairs, no time decay (rtrec/utils/interactions.py:81-119):
 * out_val[k] = clip(old[k] + delta[order[k]], lo, hi) -- or delta[order[k]] when old is NULL (upsert) --
 * and out_ts[k] = tstamp[order[k]].  Host pointers; 0 on success, -1 on invalid arguments. */
int rtrec_store_apply_round(const int64_t *order, int64_t n, const double *delta, const double *tstamp,
                            const double *old, double lo, double hi, double *out_val, double *out_ts,
                            int32_t n_threads);

/* Time decay of stored values (rtrec/utils/interactions.py:62-79), host pointers:
 * out[k] = val[k] * pow(rate, ((now_arr ? now_arr[k] : now) - ts[k]) / 86400.0) in float64 with libm's pow (what
 * CPython's `**` calls); out64 and / or out32 (rounded to float32) may be NULL.  0 on success. */
int rtrec_store_decay(const double *val, const double *ts, int64_t n, double rate, const double *now_arr,
                      double now, double *out64, float *out32, int32_t n_threads);

/* The same decay for a RESIDENT matrix (device pointers): d_out32[k] = (float)(d_val[k] * pow(rate, (now - d_ts[k]) / 86400.0)).
 * The device pow is not libm's bit for bit; a float64 product still rounds to the reference's float32 unless it lies
 * within the pow's error of a float32 rounding boundary.  Entries closer than 4096 float64 ulps to one are listed in
 * d_unsafe_idx (first `cap` of them; *d_unsafe_count = how many there were, ~n * 2^-17): the caller re-evaluates those
 * with rtrec_store_decay and patches them, every other value is provably the reference's (csrc/store_device.hip). */
int rtrec_store_decay_device(const double *d_val, const double *d_ts, int64_t n, double rate, double now,
                             float *d_out32, int32_t *d_unsafe_idx, int32_t *d_unsafe_count, int32_t cap, void *stream);

/* ---------------------------------------------------------------------------------------
 * CANDIDATES MODE FOR REQUEST-SIZED CALLS  (replaces recommend_batch(..., candidate_item_ids=...),
 * rtrec/models/internal/slim_elastic.py:723-735: scores = X[users] @ W[:, candidates], argsort()[-top_k:][::-1]; zeros compete,
 * filter_interacted is ignored, ties: the later candidate first -- DESIGN.md D1).
 * d_cands[n_cands] item ids in the caller's order (duplicates allowed: every position competes, as in the reference;
 * n_cands <= 8192); W in CSC form d_wc_* (rows ascending, float32 values).  One wave per row computes the candidates'
 * scores with the reference's summation order (ascending item; one rounded product and one rounded add per addend; float32,
 * or float64 accumulation when acc_f64: then d_out_scores64 is required) and emits the best top_k:
 * d_out_ids / d_out_scores [n_rows][top_k] (-1 / -inf beyond min(top_k, n_cands)), d_out_count[n_rows].  Meant for
 * n_rows x n_cands of a request; rtrec_slim_score_topk in RTREC_TOPK_CANDIDATES mode serves bulk calls.
 * ------------------------------------------------------------------------------------- */
int rtrec_slim_score_candidates(int32_t n_rows, const int32_t *d_row_ids, const int32_t *d_xb_ptr, const int32_t *d_xb_col,
                                const float *d_xb_val, int32_t n_x_rows, int32_t n_items, const int32_t *d_wc_ptr,
                                const int32_t *d_wc_row, const float *d_wc_val, const int32_t *d_cands, int32_t n_cands,
                                int32_t top_k, int32_t acc_f64, int32_t *d_out_ids, float *d_out_scores,
                                double *d_out_scores64, int32_t *d_out_count, void *stream);

/* ---------------------------------------------------------------------------------------
 * FLOAT64 ANSWERS FROM THE FLOAT32 FAST PASS  (SPARSE mode; a W that is float64 on the host -- the reference's serial fit,
 * slim_elastic.py:252 -- whose values are float32 numbers; ratings and weights all >= 0 and large enough that no product
 * underflows in float32: the CALLER checks both).
 * d_in_ids / d_in_scores [n_rows][top_k + 1], d_in_count [n_rows]: the lists of rtrec_slim_score_topk_opt asked for
 * top_k + 1 columns with float32 accumulation (item ids, scores descending).  Every candidate's float64 score is recomputed
 * like the reference's csr_matmat (W in CSC form d_wc_*, rows ascending; one rounded product and one rounded add per
 * addend, ascending item order), the candidates are sorted, and d_out_* [n_rows][top_k] receive ids, float32 casts and the
 * float64 scores.  A row is final when its top_k-th float64 score exceeds m32 * (1 + rel_margin), m32 = the list's
 * (top_k + 1)-th float32 score -- no column outside the list can then reach it (rel_margin >= 2 (n + 1) 2^-24 for columns
 * of at most n weights) -- or when the list holds every non-zero column (d_in_count <= top_k).  Other rows, and rows with
 * two equal float64 scores among the candidates, are appended to d_flagged (d_flagged[0] = running count, rows from
 * d_flagged[1]; room for every row on top of what is there) for the caller to score with the float64 tiled kernel.
 * d_abs_slack (NULL: the non-negative case above) -- SIGNED weights / ratings: double[n_x_rows], per row of X a bound
 * >= 2 (n_u + 2) 2^-24 sum_i |x_ui| max_c |w_ic| on how far a float32 score can be from the float64 one; a row is then final
 * when its top_k-th float64 score exceeds max(m32, 0) + d_abs_slack[row of X] (a column outside the list may also be one
 * whose float32 sum cancelled to 0), its list is full and no candidate's float64 sum is exactly 0.
 * ------------------------------------------------------------------------------------- */
int rtrec_slim_refine_topk_f64(int32_t n_rows, const int32_t *d_row_ids, const int32_t *d_xb_ptr, const int32_t *d_xb_col,
                               const float *d_xb_val, int32_t n_x_rows, int32_t n_items, const int32_t *d_wc_ptr,
                               const int32_t *d_wc_row, const float *d_wc_val, int32_t top_k, const int32_t *d_in_ids,
                               const float *d_in_scores, const int32_t *d_in_count, double rel_margin,
                               const double *d_abs_slack,
                               int32_t *d_out_ids, float *d_out_scores, double *d_out_scores64, int32_t *d_out_count,
                               int32_t *d_flagged, void *stream);

/* ---------------------------------------------------------------------------------------
 * SPARSE MODE OVER COLUMN SHARDS: THE TIE KEY OF EVERY LIST ENTRY  (the order _sparse_topk_indicies, slim_elastic.py:782-818,
 * gives columns with equal scores: a stable sort over scipy's reverse-first-touch product order).
 * d_ids / d_count: the lists of rtrec_slim_score_topk(_opt) for this rank's columns ([n_rows][top_k], item ids).  d_aux
 * [n_rows][top_k] receives, per valid entry, the position in the user's row of X (rows sorted by item id) of the first item
 * whose row of W stores a weight in the entry's column (W in CSC form: d_wc_ptr[n_items + 1], d_wc_row ascending per column;
 * a rank may hold only its own columns' entries), 0 for the others -- the key rtrec_slim_merge_topk orders equal scores by.
 * On one shard the score call computes it only for rows whose own list holds a tie; two columns of DIFFERENT shards can tie
 * without either shard seeing one, so a rank that holds part of the columns calls this before the exchange.
 * ------------------------------------------------------------------------------------- */
int rtrec_slim_first_touch_aux(int32_t n_rows, const int32_t *d_row_ids, const int32_t *d_xb_ptr, const int32_t *d_xb_col,
                               int32_t n_x_rows, int32_t n_items, const int32_t *d_wc_ptr, const int32_t *d_wc_row,
                               int32_t top_k, const int32_t *d_ids, const int32_t *d_count, uint32_t *d_aux, void *stream);

/* ---------------------------------------------------------------------------------------
 * DENSE MODE: ZERO-SCORE COLUMNS BEHIND A SHORT FAST-PASS LIST  (the tail of _dense_topk_indicies, slim_elastic.py:745-778:
 * argsort over ALL columns -- after a user's positive scores come the zero-score columns, the higher column id first
 * (DESIGN.md D1), interacted items excluded).  For a shard [col_lo, col_hi) whose weights, like the ratings, are all positive
 * and normal (the CALLER checks both: then a list that is not full holds every column the user's row touches, and every other
 * column scores exactly +0.0).
 * d_flagged_in: the rows rtrec_slim_score_topk_opt flagged in DENSE mode without a tiled layout ([0] = count, rows from [1]).
 * A flagged row whose list (d_out_* [n_rows][top_k], d_out_count) is shorter than top_k, all positive and free of equal
 * neighbours is completed in place -- ids = the highest column ids of the shard that are neither listed nor (filter_interacted)
 * rated by the user, scores +0.0, d_out_count raised -- every other flagged row is copied to d_flagged_out (same form; must
 * not alias d_flagged_in) for the tiled DENSE kernel.  top_k <= 64.
 * ------------------------------------------------------------------------------------- */
int rtrec_slim_dense_fill(int32_t n_rows, const int32_t *d_row_ids, const int32_t *d_xb_ptr, const int32_t *d_xb_col,
                          int32_t n_x_rows, int32_t col_lo, int32_t col_hi, int32_t top_k, int32_t filter_interacted,
                          int32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_aux, int32_t *d_out_count,
                          const int32_t *d_flagged_in, int32_t *d_flagged_out, void *stream);

/* ---------------------------------------------------------------------------------------
 * SEGMENT LAYOUT BUILDER  (the d_sg_* arrays of rtrec_score_opts from a W resident on the device; what has to happen
 * between a mini-batch refit -- rtrec/models/slim.py:29-43 writes W's columns, slim_elastic.py:371-374 -- and the next
 * recommend -- slim_elastic.py:707-708).  Specification: rtrec_amd/seg_layout.py::build_seg_layout.
 * W as COO triples sorted by (column, row): d_rows / d_cols int64[nnz], d_vals float32[nnz]; the shard is the columns
 * [col_lo, col_hi); d_labels int64[n_items] in [0, n_items) orders the shard's columns (stable by label, then item id).
 *
 * rtrec_slim_seg_plan   marks the shard's columns and rows, orders the columns, and SYNCHRONISES the stream once to
 *                       return h_out[4] = {n_cols, n_rows (items that hold a weight), tile_cols (0: more than 128 tiles
 *                       of 4096 columns -- no segment layout), n_tiles}.  The workspace must stay untouched until
 *                       rtrec_slim_seg_fill has been enqueued.
 * rtrec_slim_seg_fill   enqueues the rest (no synchronisation) into caller-allocated arrays:
 *                       d_info int32[n_items][2] (8-byte aligned), d_seg_ptr int32[n_rows][n_tiles + 1],
 *                       d_ent int32[ent_capacity][2] with ent_capacity >= 2 * nnz (the records in use are the first
 *                       d_seg_ptr[n_rows - 1][n_tiles]; the rest are pads), d_bound uint32[n_rows][64],
 *                       d_col_ids int32[n_cols], d_trow_ptr int32[n_tiles + 1], d_trow int32[trow_capacity][4] (16-byte
 *                       aligned) with trow_capacity >= min(nnz, n_rows * n_tiles).
 * Requires nnz < 2^27.  RTREC_OK or a negative RTREC_ERR_* code.
 * ------------------------------------------------------------------------------------- */
size_t rtrec_slim_seg_plan_workspace_bytes(int32_t n_items);
size_t rtrec_slim_seg_fill_workspace_bytes(int32_t n_items, int64_t nnz, int32_t n_rows, int32_t n_tiles);
int rtrec_slim_seg_plan(int32_t n_items, int64_t nnz, const int64_t *d_rows, const int64_t *d_cols,
                        int32_t col_lo, int32_t col_hi, const int64_t *d_labels, void *d_workspace, size_t workspace_bytes,
                        int32_t *h_out, void *stream);
int rtrec_slim_seg_fill(int32_t n_items, int64_t nnz, const int64_t *d_rows, const int64_t *d_cols, const float *d_vals,
                        int32_t col_lo, int32_t col_hi, const void *d_plan_workspace, int32_t n_cols, int32_t n_rows,
                        int32_t tile_cols, int32_t n_tiles, void *d_workspace, size_t workspace_bytes,
                        int32_t *d_info, int32_t *d_seg_ptr, int32_t *d_ent, int64_t ent_capacity, uint32_t *d_bound,
                        int32_t *d_col_ids, int32_t *d_trow_ptr, int32_t *d_trow, int64_t trow_capacity, void *stream);

/* Bulk ingest on the device (replaces one add_interaction per DataFrame row, rtrec/utils/interactions.py:81-119 as driven by
 * rtrec/recommender.py:203-223).  Device pointers.  The batch is given sorted by (user, item, arrival): d_order[k] = arrival
 * index of the k-th interaction in that order, d_start[n_groups + 1] = the runs of the distinct pairs.  Per pair, in arrival
 * order: v = d_old ? d_old[g] : 0.0, then v = max(lo, min(v + d_delta[i], hi)) per occurrence with Python's min / max (a NaN
 * sum ends as lo) -- or, with upsert != 0, v = the last occurrence's delta; d_out_ts[g] = the last occurrence's tstamp.
 * d_out_val32 (may be NULL) receives (float)v, the value the resident matrix carries.  No time decay (the decayed current
 * value of interactions.py:62-79 depends on the running max_timestamp: such stores ingest on the host). */
int rtrec_store_fold_device(const int64_t *d_order, const int64_t *d_start, int64_t n_groups,
                            const double *d_delta, const double *d_tstamp, const double *d_old, double lo, double hi,
                            int32_t upsert, double *d_out_val, double *d_out_ts, float *d_out_val32, void *stream);

/* ---------------------------------------------------------------------------------------
 * FIT with optim="sgd"  (replaces scikit-learn's SGDRegressor as slim_elastic.py:209-222 configures it, behind
 * FeatureSelectionWrapper slim_elastic.py:139-154: sklearn/linear_model/_sgd_fast.pyx.tp _plain_sgd32, float32 weights,
 * squared loss, elastic-net penalty with the truncated-gradient L1, invscaling learning rate, shuffle=True, n_iter_no_change=5).
 * The feature selection is rtrec_slim_fit_columns with max_iter = 1 (its item lists, in selection order); the solver runs in
 * blocks of epochs:
 *
 * rtrec_slim_sgd_schedule -- HOST routine, host pointers: everything of the solver that does not depend on the target, for
 * the next n_epochs epochs.  sample_order[n_samples] in/out (identity before epoch 0), state[3] in/out = {wscale, u, t}
 * ({1, 0, 1} before epoch 0).  Out: time_of[n_epochs][n_samples] = position of sample i in the epoch's shuffled order
 * (_seq_dataset.pyx.tp:137-145, our_rand_r, the same seed value every epoch); per global step g of the block eta[g],
 * the weight scale before / after the step's WeightVector32.scale(), the cumulative L1 penalty u after the step;
 * reset_cnt[n_epochs * n_samples + 1] = number of reset_wscale() calls at steps < g and reset_mult[] their float factors in
 * order (RTREC_ERR_WORKSPACE when more than reset_cap).
 *
 * rtrec_slim_fit_sgd_epochs -- device pointers: runs those epochs for every unfinished target.  d_ttime / d_tval
 * [n_epochs][nnz]: per epoch, the entries of every column of X (CSC order of the columns, d_csc_ptr) sorted by time_of[row],
 * as (time, value).  d_sel[n_targets][cap] / d_sel_count: the selected features in selection order.  State across blocks:
 * d_w, d_q [n_targets][cap] (zero before the first block), d_best_loss (+inf), d_no_improve (0), d_n_iter (0 = running; set
 * to n_iter_ when the target stops, -1 on a non-finite weight: scikit-learn raises ValueError there).  d_unfinished[1]
 * receives the number of targets still running.  cap <= 256 (up to 64: one feature per wave lane, else two or four).  When a target stops, d_w holds its coef_.
 * ------------------------------------------------------------------------------------- */
int rtrec_slim_sgd_schedule(int32_t n_samples, int32_t n_epochs, uint32_t seed,
                            double alpha, double l1_ratio, double eta0, double power_t,
                            int32_t *sample_order, double *state,
                            int32_t *time_of, double *eta, double *ws_before, double *ws_after, double *u_after,
                            int32_t *reset_cnt, float *reset_mult, int32_t reset_cap, int32_t *n_resets);
int rtrec_slim_fit_sgd_epochs(int32_t n_users, int32_t n_items, const int32_t *d_csc_ptr,
                              const int32_t *d_ttime, const float *d_tval, int64_t nnz,
                              const int32_t *d_targets, int32_t n_targets,
                              const int32_t *d_sel, const int32_t *d_sel_count, int32_t cap,
                              int32_t first_epoch, int32_t n_epochs, int32_t max_iter, double tol,
                              const double *d_eta, const double *d_ws_before, const double *d_ws_after,
                              const double *d_u_after, const int32_t *d_reset_cnt, const float *d_reset_mult,
                              float *d_w, float *d_q, double *d_best_loss, int32_t *d_no_improve,
                              int32_t *d_n_iter, int32_t *d_unfinished, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RTREC_AMD_H */
