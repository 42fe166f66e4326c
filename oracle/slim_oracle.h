/*
 * oracle/slim_oracle.h -- CPU restatement of the rtrec SLIM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker for the HIP kernels in
 * rtrec_amd/csrc; it is never linked into, imported by, or called from the product
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity status: PINNED against golden vectors generated in the build container by
 * importing the real reference (rtrec @ /root/reference + scikit-learn 1.7.2 +
 * scipy 1.15.3 + numpy 2.2.6) -- see tools/gen_golden.py and tests/golden/.
 *
 * The arithmetic lives in third-party code that rtrec calls:
 *   - scikit-learn 1.7.2  sklearn/linear_model/_cd_fast.pyx:276-561
 *       sparse_enet_coordinate_descent (float32 specialisation, no sample weights,
 *       X_mean == 0)                                    -> slim_oracle_cd()
 *   - scikit-learn 1.7.2  sklearn/utils/_random.pxd:20-34 our_rand_r (xorshift32)
 *   - scipy 1.15.3 sparsetools csr_matvec (X.T.dot(y))  -> slim_oracle_feature_scores()
 *   - scipy 1.15.3 sparsetools csr_matmat               -> slim_oracle_score_row_*()
 * and in rtrec itself:
 *   - rtrec/models/internal/slim_elastic.py:139-154 FeatureSelectionWrapper.fit
 *   - rtrec/models/internal/slim_elastic.py:434-447 / 544-560 per-column fit loop
 *   - rtrec/models/internal/slim_elastic.py:782-818 _sparse_topk_indicies
 *   - rtrec/models/internal/slim_elastic.py:744-779 _dense_topk_indicies
 *   - rtrec/models/internal/slim_elastic.py:820-857 similar_items
 *
 * All float32 arithmetic is performed with one rounding per operation (compile with
 * -ffp-contract=off and without -mfma), mirroring the SSE2-baseline scipy/sklearn wheels.
 */
#ifndef SLIM_ORACLE_H
#define SLIM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float    l1_reg;      /* (float)(alpha * l1_ratio * n_samples), _coordinate_descent.py:653 */
    float    l2_reg;      /* (float)(alpha * (1 - l1_ratio) * n_samples), :654 */
    float    tol;         /* (float)tol */
    int32_t  max_iter;
    uint32_t seed;        /* RandomState(random_state).randint(0, 2**31-1); 43 -> 494155588 */
    int32_t  positive;
    int32_t  top_features; /* nn_feature_selection, <=0 means "all features" */
} slim_oracle_cfg;

/* xorshift32 of sklearn/utils/_random.pxd:20-34; returns the draw, advances *state */
uint32_t slim_oracle_rand_r(uint32_t *state);

/* Literal restatement of _cd_fast.pyx:327-561 for float32, CSC features.
 * w[n_features] in/out (zero on entry as in a cold ElasticNet.fit); R is caller scratch of
 * n_samples floats; XtA scratch of n_features floats.  Returns n_iter (already +1). */
int32_t slim_oracle_cd(int32_t n_samples, int32_t n_features,
                       const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                       const float *y, const slim_oracle_cfg *cfg,
                       float *w, float *R, float *XtA, float *gap_out);

/* scores[i] = sum_{jj in col i, ascending} data[jj] * y[row[jj]]   (csr_matvec of X.T) with the
 * target column `skip_col` treated as all-zero (slim_elastic.py:266/438 zeroes it in place). */
void slim_oracle_feature_scores(int32_t n_items, const float *X_data, const int32_t *X_indices,
                                const int32_t *X_indptr, const float *y, int32_t skip_col,
                                float *scores);

/* Top-K of scores, descending; ties broken towards the HIGHER index (the order
 * np.argsort(kind="stable")[-1:-1-K:-1] yields; numpy's default unstable sort leaves tie
 * order unspecified -- documented divergence D1 in DESIGN.md).  Returns count = min(K, n). */
int32_t slim_oracle_select_topk(int32_t n, const float *scores, int32_t K, int32_t *sel);

/* Fit one target column j of the U x I CSC matrix X (slim_elastic.py:544-560):
 *   y = X[:, j]; X[:, j] := 0; [feature selection]; elastic-net CD; restore.
 * Output: the entries of model.sparse_coef_ for this column, ascending item index:
 *   K set  -> exactly min(K, I) entries, explicit zeros included (slim_elastic.py:153)
 *   K<=0   -> only non-zero coefficients (sparse.csr_matrix(coef_), _coordinate_descent.py:1133)
 * out_idx/out_val must hold n_items entries.  Returns the number of entries; *n_iter_out
 * receives sklearn's n_iter_. scratch: caller provides nothing; function mallocs. */
int32_t slim_oracle_fit_column(int32_t n_users, int32_t n_items,
                               const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                               int32_t j, double alpha, double l1_ratio, double tol,
                               int32_t max_iter, uint32_t seed, int32_t positive, int32_t top_features,
                               int32_t *out_idx, float *out_val, int32_t *n_iter_out, float *gap_out);

/* One row of scipy csr_matmat: A row (a_idx ascending, a_val) times B = W in CSR.
 * Emits the product row exactly as scipy stores it: entries in REVERSE first-touch order,
 * zero sums dropped.  acc/next are caller scratch of n_cols entries (acc zero, next -1 on
 * entry; restored on exit).  Returns the number of entries written to out_idx/out_val. */
int32_t slim_oracle_score_row_f32(int32_t n_a, const int32_t *a_idx, const float *a_val,
                                  const int32_t *W_indptr, const int32_t *W_indices, const float *W_data,
                                  int32_t n_cols, float *acc, int32_t *next,
                                  int32_t *out_idx, float *out_val);
int32_t slim_oracle_score_row_f64(int32_t n_a, const int32_t *a_idx, const float *a_val,
                                  const int32_t *W_indptr, const int32_t *W_indices, const float *W_data,
                                  int32_t n_cols, double *acc, int32_t *next,
                                  int32_t *out_idx, double *out_val);

/* _sparse_topk_indicies (slim_elastic.py:782-818): drop interacted (if filter), stable sort
 * by score descending, first top_k.  `n`, idx, val as produced by slim_oracle_score_row_*.
 * interacted: ascending item ids of the user row.  Returns count written to out_idx/out_val. */
int32_t slim_oracle_topk_sparse_f32(int32_t n, const int32_t *idx, const float *val,
                                    int32_t n_inter, const int32_t *inter, int32_t filter,
                                    int32_t top_k, int32_t *out_idx, float *out_val);
int32_t slim_oracle_topk_sparse_f64(int32_t n, const int32_t *idx, const double *val,
                                    int32_t n_inter, const int32_t *inter, int32_t filter,
                                    int32_t top_k, int32_t *out_idx, double *out_val);

/* _dense_topk_indicies (slim_elastic.py:744-779) over a dense score vector of n_cols
 * (interacted -> -inf, argsort, last top_k reversed, -inf dropped).  Ties: higher index first
 * (stable-argsort order; D1).  scores is modified in place like the reference does. */
int32_t slim_oracle_topk_dense_f32(int32_t n_cols, float *scores,
                                   int32_t n_inter, const int32_t *inter, int32_t filter,
                                   int32_t top_k, int32_t *out_idx, float *out_val);
int32_t slim_oracle_topk_dense_f64(int32_t n_cols, double *scores,
                                   int32_t n_inter, const int32_t *inter, int32_t filter,
                                   int32_t top_k, int32_t *out_idx, double *out_val);

/* similar_items (slim_elastic.py:838-857): stored entries of W[:, item] (CSC), drop the
 * item itself, argsort(-score)[:top_k]; ties: lower position first (stable order; D1). */
int32_t slim_oracle_similar_items(const int32_t *Wc_indptr, const int32_t *Wc_indices, const float *Wc_data,
                                  int32_t item, int32_t top_k, int32_t *out_idx, float *out_val);

/* Whole-matrix helpers used by the CPU baseline: fit columns cols[0..n_cols) and write
 * ragged results (out_ptr has n_cols+1 entries; out_idx/out_val sized n_cols * cap where
 * cap = (top_features > 0 ? min(top_features, n_items) : n_items)). Single-threaded. */
int64_t slim_oracle_fit_columns(int32_t n_users, int32_t n_items,
                                const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                                int32_t n_cols, const int32_t *cols,
                                double alpha, double l1_ratio, double tol,
                                int32_t max_iter, uint32_t seed, int32_t positive, int32_t top_features,
                                int64_t *out_ptr, int32_t *out_idx, float *out_val, int32_t *n_iter_out);

/* Score + sparse top-k for a batch of user rows (CSR Xb) against W (CSR, f32).  ids/scores
 * are [n_rows, top_k], padded with -1 / -inf; counts[n_rows].  dense != 0 selects the
 * dense top-k semantics. */
void slim_oracle_recommend_batch(int32_t n_rows, const int32_t *Xb_indptr, const int32_t *Xb_indices,
                                 const float *Xb_data,
                                 const int32_t *W_indptr, const int32_t *W_indices, const float *W_data,
                                 int32_t n_cols, int32_t top_k, int32_t filter, int32_t dense,
                                 int32_t use_f64,
                                 int32_t *ids, float *scores, int32_t *counts);

/* Multi-core variants for the CPU baseline (POSIX threads over item columns / blocks of user rows -- the
 * reference's own parallel axis, slim_elastic.py:296-301,358-366).  Per-column and per-row arithmetic is the
 * single-threaded code, so results are identical for every thread count.  Fit output has a fixed stride:
 * column c's coefficients at out_idx/out_val[c * cap ...], out_cnt[c] of them.  Return the threads used. */
int32_t slim_oracle_fit_columns_mt(int32_t n_users, int32_t n_items,
                                   const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                                   int32_t n_cols, const int32_t *cols,
                                   double alpha, double l1_ratio, double tol,
                                   int32_t max_iter, uint32_t seed, int32_t positive, int32_t top_features,
                                   int32_t cap, int32_t *out_cnt, int32_t *out_idx, float *out_val,
                                   int32_t *n_iter_out, int32_t n_threads);
int32_t slim_oracle_recommend_batch_mt(int32_t n_rows, const int32_t *Xb_indptr, const int32_t *Xb_indices,
                                       const float *Xb_data,
                                       const int32_t *W_indptr, const int32_t *W_indices, const float *W_data,
                                       int32_t n_cols, int32_t top_k, int32_t filter, int32_t dense,
                                       int32_t use_f64,
                                       int32_t *ids, float *scores, int32_t *counts, int32_t n_threads);

/* optim="sgd" (slim_elastic.py:209-222): sklearn/linear_model/_sgd_fast.pyx.tp _plain_sgd32 as SGDRegressor(loss="squared_error",
 * penalty="elasticnet", learning_rate="invscaling", fit_intercept=False, average=False) runs it; CSR rows over the selected
 * features.  Returns n_iter_ (epochs), -1 on a non-finite weight. */
int32_t slim_oracle_sgd(int32_t n_samples, int32_t n_features,
                        const float *X_data, const int32_t *X_indices, const int32_t *X_indptr, const float *y,
                        double alpha, double l1_ratio, double eta0, double power_t, double tol, int32_t max_iter,
                        uint32_t seed, float *w);
/* FeatureSelectionWrapper(SGDRegressor).fit for one target column (slim_elastic.py:139-154); -1 without feature selection */
int32_t slim_oracle_fit_column_sgd(int32_t n_users, int32_t n_items,
                                   const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                                   int32_t j, double alpha, double l1_ratio, double eta0, double tol,
                                   int32_t max_iter, uint32_t seed, int32_t top_features,
                                   int32_t *out_idx, float *out_val, int32_t *n_iter_out);

#ifdef __cplusplus
}
#endif
#endif
