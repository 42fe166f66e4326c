/*
 * oracle/slim_oracle.c -- CPU restatement of the rtrec SLIM hot path (TEST INFRASTRUCTURE).
 * See slim_oracle.h for what each function follows in the reference.  Build with
 *   gcc -O2 -std=c11 -fPIC -shared -ffp-contract=off -fno-fast-math (no -march flags)
 * so every float operation rounds once, like the SSE2-baseline scipy / scikit-learn wheels.
 */
#include "slim_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- sklearn/utils/_random.pxd:20-34 ------------------------------------------------ */
uint32_t slim_oracle_rand_r(uint32_t *state)
{
    if (*state == 0) *state = 1; /* DEFAULT_SEED */
    *state ^= (uint32_t)(*state << 13);
    *state ^= (uint32_t)(*state >> 17);
    *state ^= (uint32_t)(*state << 5);
    return *state % ((uint32_t)2147483647 + 1u);
}

/* _cd_fast.pyx:29-31 rand_int */
static inline uint32_t rand_int(uint32_t end, uint32_t *state)
{
    return slim_oracle_rand_r(state) % end;
}

/* BLAS level-1 stand-ins.  The reference calls OpenBLAS sdot/sasum here (_cd_fast.pyx:426,
 * 522,531,540,543); their internal summation order depends on the CPU kernel and thread
 * count, so it cannot be restated.  These terms feed ONLY the duality-gap stop test
 * (gap < tol), never the coefficients.  Plain ascending-index float32 accumulation is the
 * oracle's (and the GPU kernel's) canonical order.  -- documented divergence D2. */
static float seq_dot(int32_t n, const float *a, const float *b)
{
    float s = 0.0f;
    for (int32_t i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}
static float seq_asum(int32_t n, const float *a)
{
    float s = 0.0f;
    for (int32_t i = 0; i < n; i++) s += fabsf(a[i]);
    return s;
}

/* Test switch: evaluate the coordinate-descent dot products (tmp, XtA) with the CPU model of the device's
 * binade-speculative fold (fold_model.c) instead of the literal loop.  Off by default; tests turn it on to show
 * that the speculative fold leaves every coefficient bit and sweep count of the goldens unchanged, and to count
 * how many entries of real sums take the integer path.  Per thread. */
typedef struct { int64_t entries, spec_entries, serial_entries, passes; } fold_model_stats;
float fold_model_fold(float acc, const float *p, int64_t n, fold_model_stats *st);
static __thread int g_fold_model = 0;
static __thread fold_model_stats g_fold_stats;
void slim_oracle_set_fold_model(int on) { g_fold_model = on; memset(&g_fold_stats, 0, sizeof g_fold_stats); }
void slim_oracle_fold_model_stats(int64_t out[4])
{
    out[0] = g_fold_stats.entries; out[1] = g_fold_stats.spec_entries;
    out[2] = g_fold_stats.serial_entries; out[3] = g_fold_stats.passes;
}
/* sum_{jj} a(jj) * b(jj) through the fold model: products rounded exactly like the literal loop's */
static float model_dot(const float *R, const float *X_data, const int32_t *X_indices, int32_t b, int32_t e, int r_first)
{
    float *p = (float *)malloc(sizeof(float) * (size_t)(e > b ? e - b : 1));
    for (int32_t jj = b; jj < e; jj++)
        p[jj - b] = r_first ? R[X_indices[jj]] * X_data[jj] : X_data[jj] * R[X_indices[jj]];
    const float v = fold_model_fold(0.0f, p, e - b, &g_fold_stats);
    free(p);
    return v;
}

/* ---- _cd_fast.pyx:327-561, float32, no sample weights, X_mean == 0 -------------------- */
int32_t slim_oracle_cd(int32_t n_samples, int32_t n_features,
                       const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                       const float *y, const slim_oracle_cfg *cfg,
                       float *w, float *R, float *XtA, float *gap_out)
{
    const float alpha = cfg->l1_reg, beta = cfg->l2_reg;
    float tol = cfg->tol;
    const float d_w_tol = tol;
    float gap = tol + 1.0f;
    uint32_t rng = cfg->seed;
    float *norm_cols_X = (float *)calloc((size_t)(n_features > 0 ? n_features : 1), sizeof(float));
    int32_t n_iter = 0;

    memcpy(R, y, (size_t)n_samples * sizeof(float)); /* :381 R = y.copy() */

    /* :394-428  column norms and R = y - X w */
    for (int32_t ii = 0; ii < n_features; ii++) {
        float normalize_sum = 0.0f;
        const float w_ii = w[ii];
        for (int32_t jj = X_indptr[ii]; jj < X_indptr[ii + 1]; jj++) {
            normalize_sum += X_data[jj] * X_data[jj];          /* (x - 0)**2 */
            R[X_indices[jj]] -= X_data[jj] * w_ii;
        }
        norm_cols_X[ii] = normalize_sum;                        /* + (n - nnz) * 0**2 */
    }

    tol *= seq_dot(n_samples, y, y);                            /* :426 */

    for (n_iter = 0; n_iter < cfg->max_iter; n_iter++) {
        float w_max = 0.0f, d_w_max = 0.0f;
        for (int32_t f_iter = 0; f_iter < n_features; f_iter++) {
            const int32_t ii = (int32_t)rand_int((uint32_t)n_features, &rng);   /* :435 */
            if (norm_cols_X[ii] == 0.0f) continue;                               /* :439 */
            const int32_t startptr = X_indptr[ii], endptr = X_indptr[ii + 1];
            const float w_ii = w[ii];

            if (w_ii != 0.0f)                                                    /* :447 */
                for (int32_t jj = startptr; jj < endptr; jj++)
                    R[X_indices[jj]] += X_data[jj] * w_ii;

            float tmp = 0.0f;                                                    /* :464 */
            if (g_fold_model) tmp = model_dot(R, X_data, X_indices, startptr, endptr, 1);
            else
            for (int32_t jj = startptr; jj < endptr; jj++)
                tmp += R[X_indices[jj]] * X_data[jj];

            if (cfg->positive && tmp < 0.0f) {                                   /* :471 */
                w[ii] = 0.0f;
            } else {
                /* :474  fsign(tmp) * fmax(fabs(tmp) - alpha, 0) / (norm_cols_X[ii] + beta)
                 * libc fabs() is double -> the numerator is formed in double, the
                 * denominator is a float sum, the quotient is rounded to float once. */
                const double num = fabs((double)tmp) - (double)alpha;
                const double sgn = (tmp == 0.0f) ? 0.0 : (tmp > 0.0f ? 1.0 : -1.0);
                const float den = norm_cols_X[ii] + beta;
                w[ii] = (float)(sgn * (num > 0.0 ? num : 0.0) / (double)den);
            }

            if (w[ii] != 0.0f) {                                                 /* :477 */
                const float wn = w[ii];
                for (int32_t jj = startptr; jj < endptr; jj++)
                    R[X_indices[jj]] -= X_data[jj] * wn;
            }

            const float d_w_ii = fabsf(w[ii] - w_ii);                            /* :493 */
            if (d_w_ii > d_w_max) d_w_max = d_w_ii;
            if (fabsf(w[ii]) > w_max) w_max = fabsf(w[ii]);
        }

        if (w_max == 0.0f || d_w_max / w_max < d_w_tol || n_iter == cfg->max_iter - 1) { /* :499 */
            for (int32_t ii = 0; ii < n_features; ii++) {                        /* :506 */
                float s = 0.0f;
                if (g_fold_model) s = model_dot(R, X_data, X_indices, X_indptr[ii], X_indptr[ii + 1], 0);
                else
                for (int32_t kk = X_indptr[ii]; kk < X_indptr[ii + 1]; kk++)
                    s += X_data[kk] * R[X_indices[kk]];
                s -= beta * w[ii];
                XtA[ii] = s;
            }
            float dual_norm_XtA;
            if (cfg->positive) {                                                 /* :515 max() */
                dual_norm_XtA = XtA[0];
                for (int32_t i = 1; i < n_features; i++)
                    if (XtA[i] > dual_norm_XtA) dual_norm_XtA = XtA[i];
            } else {                                                             /* abs_max() */
                dual_norm_XtA = fabsf(XtA[0]);
                for (int32_t i = 1; i < n_features; i++)
                    if (fabsf(XtA[i]) > dual_norm_XtA) dual_norm_XtA = fabsf(XtA[i]);
            }
            const float R_norm2 = seq_dot(n_samples, R, R);                      /* :522 */
            const float w_norm2 = seq_dot(n_features, w, w);                     /* :531 */
            float const_;
            if (dual_norm_XtA > alpha) {                                         /* :532 */
                const_ = alpha / dual_norm_XtA;
                const float A_norm2 = R_norm2 * (const_ * const_);
                gap = (float)(0.5 * (double)(R_norm2 + A_norm2));
            } else {
                const_ = 1.0f;
                gap = R_norm2;
            }
            const float l1_norm = seq_asum(n_features, w);                       /* :540 */
            /* :542-544  float terms first, the 0.5*beta*(...)*w_norm2 product in double
             * (0.5 is a C double literal), sum added to gap in double, rounded to float. */
            const float t12 = alpha * l1_norm - const_ * seq_dot(n_samples, R, y);
            const double t3 = 0.5 * (double)beta * (double)(1.0f + const_ * const_) * (double)w_norm2;
            gap = (float)((double)gap + ((double)t12 + t3));
            if (gap < tol) break;                                                /* :546 */
        }
    }
    free(norm_cols_X);
    if (gap_out) *gap_out = gap;
    /* `for n_iter in range(max_iter)` leaves n_iter = max_iter-1 when exhausted; returns +1 */
    return (n_iter < cfg->max_iter ? n_iter : cfg->max_iter - 1) + 1;
}

/* ---- scipy csr_matvec on X.T (slim_elastic.py:141) ----------------------------------- */
void slim_oracle_feature_scores(int32_t n_items, const float *X_data, const int32_t *X_indices,
                                const int32_t *X_indptr, const float *y, int32_t skip_col,
                                float *scores)
{
    for (int32_t i = 0; i < n_items; i++) {
        float sum = 0.0f;
        if (i != skip_col)
            for (int32_t jj = X_indptr[i]; jj < X_indptr[i + 1]; jj++)
                sum += X_data[jj] * y[X_indices[jj]];
        scores[i] = sum;
    }
}

/* ---- np.argsort(scores)[-1:-1-K:-1] with the stable-sort tie order (D1) --------------- */
typedef struct { float s; int32_t i; } fs_pair;
static int cmp_desc_hi(const void *a, const void *b)
{
    const fs_pair *x = (const fs_pair *)a, *y = (const fs_pair *)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) ? -1 : (x->i < y->i ? 1 : 0);
}
int32_t slim_oracle_select_topk(int32_t n, const float *scores, int32_t K, int32_t *sel)
{
    fs_pair *p = (fs_pair *)malloc((size_t)(n > 0 ? n : 1) * sizeof(fs_pair));
    for (int32_t i = 0; i < n; i++) { p[i].s = scores[i]; p[i].i = i; }
    qsort(p, (size_t)n, sizeof(fs_pair), cmp_desc_hi);
    const int32_t cnt = K < n ? K : n;
    for (int32_t i = 0; i < cnt; i++) sel[i] = p[i].i;
    free(p);
    return cnt;
}

/* ---- one target column (slim_elastic.py:139-154, 544-560) ---------------------------- */
int32_t slim_oracle_fit_column(int32_t n_users, int32_t n_items,
                               const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                               int32_t j, double alpha, double l1_ratio, double tol,
                               int32_t max_iter, uint32_t seed, int32_t positive, int32_t top_features,
                               int32_t *out_idx, float *out_val, int32_t *n_iter_out, float *gap_out)
{
    slim_oracle_cfg cfg;
    cfg.l1_reg = (float)(alpha * l1_ratio * (double)n_users);
    cfg.l2_reg = (float)(alpha * (1.0 - l1_ratio) * (double)n_users);
    cfg.tol = (float)tol;
    cfg.max_iter = max_iter;
    cfg.seed = seed;
    cfg.positive = positive;
    cfg.top_features = top_features;

    float *y = (float *)calloc((size_t)n_users, sizeof(float));
    float *R = (float *)malloc((size_t)n_users * sizeof(float));
    for (int32_t jj = X_indptr[j]; jj < X_indptr[j + 1]; jj++) y[X_indices[jj]] = X_data[jj];

    int32_t n_out = 0;
    if (top_features > 0) {
        float *scores = (float *)malloc((size_t)n_items * sizeof(float));
        int32_t *sel = (int32_t *)malloc((size_t)n_items * sizeof(int32_t));
        slim_oracle_feature_scores(n_items, X_data, X_indices, X_indptr, y, j, scores);
        const int32_t K = slim_oracle_select_topk(n_items, scores, top_features, sel);
        /* X[:, sel] with the target column zeroed (values 0, structure kept) */
        int64_t nnz = 0;
        for (int32_t p = 0; p < K; p++) nnz += X_indptr[sel[p] + 1] - X_indptr[sel[p]];
        float *Fd = (float *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
        int32_t *Fi = (int32_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t));
        int32_t *Fp = (int32_t *)malloc((size_t)(K + 1) * sizeof(int32_t));
        int32_t pos = 0;
        Fp[0] = 0;
        for (int32_t p = 0; p < K; p++) {
            const int32_t c = sel[p];
            for (int32_t jj = X_indptr[c]; jj < X_indptr[c + 1]; jj++) {
                Fd[pos] = (c == j) ? 0.0f : X_data[jj];
                Fi[pos] = X_indices[jj];
                pos++;
            }
            Fp[p + 1] = pos;
        }
        float *w = (float *)calloc((size_t)(K > 0 ? K : 1), sizeof(float));
        float *XtA = (float *)malloc((size_t)(K > 0 ? K : 1) * sizeof(float));
        int32_t n_iter = slim_oracle_cd(n_users, K, Fd, Fi, Fp, y, &cfg, w, R, XtA, gap_out);
        if (n_iter_out) *n_iter_out = n_iter;
        /* csr_matrix((coef, (0, sel))) -> indices ascending, explicit zeros kept (:153) */
        fs_pair *o = (fs_pair *)malloc((size_t)(K > 0 ? K : 1) * sizeof(fs_pair));
        for (int32_t p = 0; p < K; p++) { o[p].s = w[p]; o[p].i = sel[p]; }
        for (int32_t a = 1; a < K; a++) { /* insertion sort by index */
            fs_pair t = o[a]; int32_t b = a - 1;
            while (b >= 0 && o[b].i > t.i) { o[b + 1] = o[b]; b--; }
            o[b + 1] = t;
        }
        for (int32_t p = 0; p < K; p++) { out_idx[p] = o[p].i; out_val[p] = o[p].s; }
        n_out = K;
        free(o); free(XtA); free(w); free(Fp); free(Fi); free(Fd); free(sel); free(scores);
    } else {
        /* all I columns are features; the target column's values are zero */
        const int64_t nnz = X_indptr[n_items];
        float *Fd = (float *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
        memcpy(Fd, X_data, (size_t)nnz * sizeof(float));
        for (int32_t jj = X_indptr[j]; jj < X_indptr[j + 1]; jj++) Fd[jj] = 0.0f;
        float *w = (float *)calloc((size_t)n_items, sizeof(float));
        float *XtA = (float *)malloc((size_t)n_items * sizeof(float));
        int32_t n_iter = slim_oracle_cd(n_users, n_items, Fd, X_indices, X_indptr, y, &cfg, w, R, XtA, gap_out);
        if (n_iter_out) *n_iter_out = n_iter;
        for (int32_t i = 0; i < n_items; i++)
            if (w[i] != 0.0f) { out_idx[n_out] = i; out_val[n_out] = w[i]; n_out++; }
        free(XtA); free(w); free(Fd);
    }
    free(R); free(y);
    return n_out;
}

int64_t slim_oracle_fit_columns(int32_t n_users, int32_t n_items,
                                const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                                int32_t n_cols, const int32_t *cols,
                                double alpha, double l1_ratio, double tol,
                                int32_t max_iter, uint32_t seed, int32_t positive, int32_t top_features,
                                int64_t *out_ptr, int32_t *out_idx, float *out_val, int32_t *n_iter_out)
{
    int64_t pos = 0;
    out_ptr[0] = 0;
    for (int32_t c = 0; c < n_cols; c++) {
        int32_t it = 0; float gap = 0.0f;
        const int32_t n = slim_oracle_fit_column(n_users, n_items, X_data, X_indices, X_indptr,
                                                 cols[c], alpha, l1_ratio, tol, max_iter, seed,
                                                 positive, top_features,
                                                 out_idx + pos, out_val + pos, &it, &gap);
        if (n_iter_out) n_iter_out[c] = it;
        pos += n;
        out_ptr[c + 1] = pos;
    }
    return pos;
}

/* ---- optim="sgd": scikit-learn SGDRegressor as slim_elastic.py:209-222 configures it ----------------------------
 * SGDRegressor(loss="squared_error", penalty="elasticnet", alpha, l1_ratio, fit_intercept=False, max_iter, tol,
 * random_state, learning_rate="invscaling", eta0, average=False) with the defaults power_t=0.25, shuffle=True,
 * early_stopping=False, n_iter_no_change=5 -> sklearn/linear_model/_stochastic_gradient.py:1664-1734 (_fit_regressor) ->
 * sklearn/linear_model/_sgd_fast.pyx.tp:_plain_sgd32 (X is float32, so coef_ is: _stochastic_gradient.py:227,1701) with
 * WeightVector32 (sklearn/utils/_weight_vector.pyx.tp: float weights, DOUBLE wscale, reset threshold 1e-6),
 * CSRDataset32.shuffle (sklearn/utils/_seq_dataset.pyx.tp:137-145: Fisher-Yates with our_rand_r, the SAME seed value every
 * epoch, applied to the order the previous epoch left) and CyHalfSquaredError (sklearn/_loss/_loss.pyx.tp:310-321).
 * Every expression below keeps the C types the Cython source gives it (float products, double accumulators, float
 * parameters of scale() / add()).  X: CSR rows over the K selected features, entries in ascending feature position.
 * w[n_features] zero on entry.  Returns n_iter_ (epochs run), or -1 on a non-finite weight (sklearn raises ValueError). */
int32_t slim_oracle_sgd(int32_t n_samples, int32_t n_features,
                        const float *X_data, const int32_t *X_indices, const int32_t *X_indptr, const float *y,
                        double alpha, double l1_ratio, double eta0, double power_t, double tol, int32_t max_iter,
                        uint32_t seed, float *w)
{
    float *q = (float *)calloc((size_t)(n_features > 0 ? n_features : 1), sizeof(float));
    int32_t *index = (int32_t *)malloc((size_t)(n_samples > 0 ? n_samples : 1) * sizeof(int32_t));
    for (int32_t i = 0; i < n_samples; i++) index[i] = i;
    double wscale = 1.0, u = 0.0, t = 1.0, best_loss = INFINITY, eta = eta0;
    const double intercept = 0.0;
    int no_improvement_count = 0, infinity = 0;
    const unsigned int train_count = (unsigned int)n_samples;
    int32_t epoch = 0;
    for (epoch = 0; epoch < max_iter; epoch++) {
        double sumloss = 0.0;
        {   /* dataset.shuffle(seed): seed is passed by value */
            uint32_t s = seed;
            for (int32_t i = 0; i < n_samples - 1; i++) {
                const int32_t j = i + (int32_t)(slim_oracle_rand_r(&s) % (uint32_t)(n_samples - i));
                const int32_t tmp = index[i]; index[i] = index[j]; index[j] = tmp;
            }
        }
        for (int32_t i = 0; i < n_samples; i++) {
            const int32_t si = index[i];
            const float *xd = X_data + X_indptr[si];
            const int32_t *xi = X_indices + X_indptr[si];
            const int32_t xnnz = X_indptr[si + 1] - X_indptr[si];
            const float yv = y[si];
            /* p = w.dot(x) + intercept: float products, double sum, the float return value of dot() */
            double innerprod = 0.0;
            for (int32_t j = 0; j < xnnz; j++) innerprod += w[xi[j]] * xd[j];
            innerprod *= wscale;
            const double p = (double)(float)innerprod + intercept;
            eta = eta0 / pow(t, power_t);
            sumloss += 0.5 * (p - (double)yv) * (p - (double)yv);
            double dloss = p - (double)yv;
            if (dloss < -1e12) dloss = -1e12; else if (dloss > 1e12) dloss = 1e12;
            double update = -eta * dloss;
            { const float class_weight = 1.0f, sample_weight = 1.0f; update *= class_weight * sample_weight; }
            {   /* w.scale(max(0, 1.0 - ((1.0 - l1_ratio) * eta * alpha))): the argument becomes a float parameter */
                const double a = 1.0 - ((1.0 - l1_ratio) * eta * alpha);
                const float c = (float)(a > 0.0 ? a : 0.0);
                wscale *= c;
                if (wscale < 1e-6) {                 /* reset_wscale: sscal with a float alpha */
                    const float ws = (float)wscale;
                    for (int32_t f = 0; f < n_features; f++) w[f] = ws * w[f];
                    wscale = 1.0;
                }
            }
            if (update != 0.0) {                     /* w.add(x, update): c and the local wscale are floats */
                const float c = (float)update, wsf = (float)wscale;
                for (int32_t j = 0; j < xnnz; j++) {
                    const double val = xd[j];
                    w[xi[j]] = (float)((double)w[xi[j]] + val * (double)(c / wsf));
                }
            }
            u += (l1_ratio * eta * alpha);
            for (int32_t j = 0; j < xnnz; j++) {     /* l1penalty32 (truncated gradient) */
                const int32_t idx = xi[j];
                const double z = w[idx];
                if (wscale * z > 0.0) {
                    const double v = (double)w[idx] - ((u + (double)q[idx]) / wscale);
                    w[idx] = (float)(v > 0.0 ? v : 0.0);
                } else if (wscale * z < 0.0) {
                    const double v = (double)w[idx] + ((u - (double)q[idx]) / wscale);
                    w[idx] = (float)(v < 0.0 ? v : 0.0);
                }
                q[idx] = (float)((double)q[idx] + wscale * ((double)w[idx] - z));
            }
            t += 1.0;
        }
        for (int32_t f = 0; f < n_features; f++) if (!isfinite(w[f])) infinity = 1;
        if (infinity) break;
        if (tol > -INFINITY && sumloss > best_loss - tol * train_count) no_improvement_count++;
        else no_improvement_count = 0;
        if (sumloss < best_loss) best_loss = sumloss;
        if (no_improvement_count >= 5) break;        /* n_iter_no_change; learning_rate is not "adaptive" */
    }
    {   const float ws = (float)wscale;              /* w.reset_wscale() */
        for (int32_t f = 0; f < n_features; f++) w[f] = ws * w[f];
    }
    free(index); free(q);
    if (infinity) return -1;
    return (epoch < max_iter ? epoch : max_iter - 1) + 1;
}

/* One target column with optim="sgd" behind FeatureSelectionWrapper (slim_elastic.py:139-154): the same X^T y / top-K
 * selection as the coordinate-descent path, X[:, sel] converted to CSR (check_array(accept_sparse="csr") on the CSC slice:
 * a row's entries come out in ascending position of the selected features), then slim_oracle_sgd.  The zeroed target
 * column's explicit zeros change no state (products, updates and penalties of 0) and are dropped.  Output: the K selected
 * items ascending with their coefficients, explicit zeros kept (:153). */
int32_t slim_oracle_fit_column_sgd(int32_t n_users, int32_t n_items,
                                   const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                                   int32_t j, double alpha, double l1_ratio, double eta0, double tol,
                                   int32_t max_iter, uint32_t seed, int32_t top_features,
                                   int32_t *out_idx, float *out_val, int32_t *n_iter_out)
{
    if (top_features <= 0) return -1;     /* the reference itself fails without nn_feature_selection (no sparse_coef_) */
    float *y = (float *)calloc((size_t)n_users, sizeof(float));
    for (int32_t jj = X_indptr[j]; jj < X_indptr[j + 1]; jj++) y[X_indices[jj]] = X_data[jj];
    float *scores = (float *)malloc((size_t)n_items * sizeof(float));
    int32_t *sel = (int32_t *)malloc((size_t)n_items * sizeof(int32_t));
    slim_oracle_feature_scores(n_items, X_data, X_indices, X_indptr, y, j, scores);
    const int32_t K = slim_oracle_select_topk(n_items, scores, top_features, sel);
    int32_t *Rp = (int32_t *)calloc((size_t)n_users + 1, sizeof(int32_t));
    int64_t nnz = 0;
    for (int32_t p = 0; p < K; p++) {
        if (sel[p] == j) continue;
        for (int32_t jj = X_indptr[sel[p]]; jj < X_indptr[sel[p] + 1]; jj++) { Rp[X_indices[jj] + 1]++; nnz++; }
    }
    for (int32_t r = 0; r < n_users; r++) Rp[r + 1] += Rp[r];
    float *Rd = (float *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(float));
    int32_t *Ri = (int32_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int32_t));
    int32_t *fill = (int32_t *)malloc((size_t)(n_users > 0 ? n_users : 1) * sizeof(int32_t));
    memcpy(fill, Rp, (size_t)n_users * sizeof(int32_t));
    for (int32_t p = 0; p < K; p++) {           /* ascending feature position -> sorted rows */
        if (sel[p] == j) continue;
        for (int32_t jj = X_indptr[sel[p]]; jj < X_indptr[sel[p] + 1]; jj++) {
            const int32_t r = X_indices[jj];
            Rd[fill[r]] = X_data[jj]; Ri[fill[r]] = p; fill[r]++;
        }
    }
    float *w = (float *)calloc((size_t)(K > 0 ? K : 1), sizeof(float));
    const int32_t n_iter = slim_oracle_sgd(n_users, K, Rd, Ri, Rp, y, alpha, l1_ratio, eta0, 0.25, tol, max_iter, seed, w);
    if (n_iter_out) *n_iter_out = n_iter;
    fs_pair *o = (fs_pair *)malloc((size_t)(K > 0 ? K : 1) * sizeof(fs_pair));
    for (int32_t p = 0; p < K; p++) { o[p].s = w[p]; o[p].i = sel[p]; }
    for (int32_t a = 1; a < K; a++) {
        fs_pair t = o[a]; int32_t b = a - 1;
        while (b >= 0 && o[b].i > t.i) { o[b + 1] = o[b]; b--; }
        o[b + 1] = t;
    }
    for (int32_t p = 0; p < K; p++) { out_idx[p] = o[p].i; out_val[p] = o[p].s; }
    free(o); free(w); free(fill); free(Ri); free(Rd); free(Rp); free(sel); free(scores); free(y);
    return K;
}

/* ---- scipy sparsetools csr_matmat, one row ------------------------------------------- */
#define DEFINE_SCORE_ROW(NAME, T)                                                              \
int32_t NAME(int32_t n_a, const int32_t *a_idx, const float *a_val,                            \
             const int32_t *W_indptr, const int32_t *W_indices, const float *W_data,           \
             int32_t n_cols, T *acc, int32_t *next, int32_t *out_idx, T *out_val)              \
{                                                                                              \
    (void)n_cols;                                                                              \
    int32_t head = -2, length = 0;                                                             \
    for (int32_t jj = 0; jj < n_a; jj++) {                                                     \
        const int32_t j = a_idx[jj];                                                           \
        const T v = (T)a_val[jj];                                                              \
        for (int32_t kk = W_indptr[j]; kk < W_indptr[j + 1]; kk++) {                           \
            const int32_t k = W_indices[kk];                                                   \
            acc[k] += v * (T)W_data[kk];                                                       \
            if (next[k] == -1) { next[k] = head; head = k; length++; }                         \
        }                                                                                      \
    }                                                                                          \
    int32_t nnz = 0;                                                                           \
    for (int32_t jj = 0; jj < length; jj++) {                                                  \
        if (acc[head] != 0) { out_idx[nnz] = head; out_val[nnz] = acc[head]; nnz++; }          \
        const int32_t tmp = head;                                                              \
        head = next[head];                                                                     \
        next[tmp] = -1;                                                                        \
        acc[tmp] = 0;                                                                          \
    }                                                                                          \
    return nnz;                                                                                \
}
DEFINE_SCORE_ROW(slim_oracle_score_row_f32, float)
DEFINE_SCORE_ROW(slim_oracle_score_row_f64, double)

static int in_sorted(int32_t n, const int32_t *a, int32_t v)
{
    int32_t lo = 0, hi = n;
    while (lo < hi) { int32_t m = (lo + hi) >> 1; if (a[m] < v) lo = m + 1; else hi = m; }
    return lo < n && a[lo] == v;
}

/* ---- _sparse_topk_indicies: Python's sorted(..., reverse=True) is stable --------------- */
#define DEFINE_TOPK_SPARSE(NAME, T)                                                            \
int32_t NAME(int32_t n, const int32_t *idx, const T *val, int32_t n_inter, const int32_t *inter,\
             int32_t filter, int32_t top_k, int32_t *out_idx, T *out_val)                      \
{                                                                                              \
    int32_t *ci = (int32_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int32_t));                \
    T *cv = (T *)malloc((size_t)(n > 0 ? n : 1) * sizeof(T));                                  \
    int32_t m = 0;                                                                             \
    for (int32_t i = 0; i < n; i++) {                                                          \
        if (filter && in_sorted(n_inter, inter, idx[i])) continue;                             \
        ci[m] = idx[i]; cv[m] = val[i]; m++;                                                   \
    }                                                                                          \
    /* stable insertion-style selection of the first top_k in descending order */              \
    int32_t cnt = 0;                                                                           \
    for (int32_t i = 0; i < m; i++) {                                                          \
        int32_t pos = cnt;                                                                     \
        while (pos > 0 && out_val[pos - 1] < cv[i]) pos--;                                     \
        if (pos >= top_k) continue;                                                            \
        const int32_t last = cnt < top_k ? cnt : top_k - 1;                                    \
        for (int32_t q = last; q > pos; q--) { out_val[q] = out_val[q - 1]; out_idx[q] = out_idx[q - 1]; } \
        out_val[pos] = cv[i]; out_idx[pos] = ci[i];                                            \
        if (cnt < top_k) cnt++;                                                                \
    }                                                                                          \
    free(ci); free(cv);                                                                        \
    return cnt;                                                                                \
}
DEFINE_TOPK_SPARSE(slim_oracle_topk_sparse_f32, float)
DEFINE_TOPK_SPARSE(slim_oracle_topk_sparse_f64, double)

/* ---- _dense_topk_indicies -------------------------------------------------------------- */
#define DEFINE_TOPK_DENSE(NAME, T)                                                             \
int32_t NAME(int32_t n_cols, T *scores, int32_t n_inter, const int32_t *inter, int32_t filter, \
             int32_t top_k, int32_t *out_idx, T *out_val)                                      \
{                                                                                              \
    if (filter) for (int32_t i = 0; i < n_inter; i++) scores[inter[i]] = (T)-INFINITY;         \
    /* argsort(scores)[-top_k:][::-1] under a stable sort: descending, ties -> higher index */ \
    int32_t cnt = 0;                                                                           \
    for (int32_t i = 0; i < n_cols; i++) {                                                     \
        int32_t pos = cnt;                                                                     \
        while (pos > 0 && out_val[pos - 1] <= scores[i]) pos--;                                \
        if (pos >= top_k) continue;                                                            \
        const int32_t last = cnt < top_k ? cnt : top_k - 1;                                    \
        for (int32_t q = last; q > pos; q--) { out_val[q] = out_val[q - 1]; out_idx[q] = out_idx[q - 1]; } \
        out_val[pos] = scores[i]; out_idx[pos] = i;                                            \
        if (cnt < top_k) cnt++;                                                                \
    }                                                                                          \
    int32_t m = 0;                                                                             \
    for (int32_t i = 0; i < cnt; i++)                                                          \
        if (out_val[i] != (T)-INFINITY) { out_val[m] = out_val[i]; out_idx[m] = out_idx[i]; m++; } \
    return m;                                                                                  \
}
DEFINE_TOPK_DENSE(slim_oracle_topk_dense_f32, float)
DEFINE_TOPK_DENSE(slim_oracle_topk_dense_f64, double)

/* ---- similar_items (slim_elastic.py:838-857) ----------------------------------------- */
int32_t slim_oracle_similar_items(const int32_t *Wc_indptr, const int32_t *Wc_indices, const float *Wc_data,
                                  int32_t item, int32_t top_k, int32_t *out_idx, float *out_val)
{
    int32_t cnt = 0;
    for (int32_t kk = Wc_indptr[item]; kk < Wc_indptr[item + 1]; kk++) {
        const int32_t i = Wc_indices[kk];
        if (i == item) continue;
        const float v = Wc_data[kk];
        /* argsort(-v) stable: descending v, ties keep stored (ascending index) order */
        int32_t pos = cnt;
        while (pos > 0 && out_val[pos - 1] < v) pos--;
        if (pos >= top_k) continue;
        const int32_t last = cnt < top_k ? cnt : top_k - 1;
        for (int32_t q = last; q > pos; q--) { out_val[q] = out_val[q - 1]; out_idx[q] = out_idx[q - 1]; }
        out_val[pos] = v; out_idx[pos] = i;
        if (cnt < top_k) cnt++;
    }
    return cnt;
}

/* ---- batch recommend: score rows + top-k --------------------------------------------- */
void slim_oracle_recommend_batch(int32_t n_rows, const int32_t *Xb_indptr, const int32_t *Xb_indices,
                                 const float *Xb_data,
                                 const int32_t *W_indptr, const int32_t *W_indices, const float *W_data,
                                 int32_t n_cols, int32_t top_k, int32_t filter, int32_t dense,
                                 int32_t use_f64,
                                 int32_t *ids, float *scores, int32_t *counts)
{
    float *acc32 = (float *)calloc((size_t)n_cols, sizeof(float));
    double *acc64 = (double *)calloc((size_t)n_cols, sizeof(double));
    int32_t *next = (int32_t *)malloc((size_t)n_cols * sizeof(int32_t));
    int32_t *pidx = (int32_t *)malloc((size_t)n_cols * sizeof(int32_t));
    float *pv32 = (float *)malloc((size_t)n_cols * sizeof(float));
    double *pv64 = (double *)malloc((size_t)n_cols * sizeof(double));
    int32_t *oi = (int32_t *)malloc((size_t)(top_k > 0 ? top_k : 1) * sizeof(int32_t));
    float *ov32 = (float *)malloc((size_t)(top_k > 0 ? top_k : 1) * sizeof(float));
    double *ov64 = (double *)malloc((size_t)(top_k > 0 ? top_k : 1) * sizeof(double));
    for (int32_t k = 0; k < n_cols; k++) next[k] = -1;

    for (int32_t r = 0; r < n_rows; r++) {
        const int32_t s = Xb_indptr[r], n_a = Xb_indptr[r + 1] - s;
        int32_t cnt = 0;
        if (!use_f64) {
            int32_t n = slim_oracle_score_row_f32(n_a, Xb_indices + s, Xb_data + s, W_indptr, W_indices,
                                                  W_data, n_cols, acc32, next, pidx, pv32);
            if (dense) {
                float *d = (float *)calloc((size_t)n_cols, sizeof(float));
                for (int32_t i = 0; i < n; i++) d[pidx[i]] = pv32[i];
                cnt = slim_oracle_topk_dense_f32(n_cols, d, n_a, Xb_indices + s, filter, top_k, oi, ov32);
                free(d);
            } else {
                cnt = slim_oracle_topk_sparse_f32(n, pidx, pv32, n_a, Xb_indices + s, filter, top_k, oi, ov32);
            }
            for (int32_t i = 0; i < cnt; i++) scores[(int64_t)r * top_k + i] = ov32[i];
        } else {
            int32_t n = slim_oracle_score_row_f64(n_a, Xb_indices + s, Xb_data + s, W_indptr, W_indices,
                                                  W_data, n_cols, acc64, next, pidx, pv64);
            if (dense) {
                double *d = (double *)calloc((size_t)n_cols, sizeof(double));
                for (int32_t i = 0; i < n; i++) d[pidx[i]] = pv64[i];
                cnt = slim_oracle_topk_dense_f64(n_cols, d, n_a, Xb_indices + s, filter, top_k, oi, ov64);
                free(d);
            } else {
                cnt = slim_oracle_topk_sparse_f64(n, pidx, pv64, n_a, Xb_indices + s, filter, top_k, oi, ov64);
            }
            for (int32_t i = 0; i < cnt; i++) scores[(int64_t)r * top_k + i] = (float)ov64[i];
        }
        for (int32_t i = 0; i < cnt; i++) ids[(int64_t)r * top_k + i] = oi[i];
        for (int32_t i = cnt; i < top_k; i++) {
            ids[(int64_t)r * top_k + i] = -1;
            scores[(int64_t)r * top_k + i] = -INFINITY;
        }
        counts[r] = cnt;
    }
    free(ov64); free(ov32); free(oi); free(pv64); free(pv32); free(pidx); free(next); free(acc64); free(acc32);
}


/* ---- multi-core variants for the CPU baseline ------------------------
 * The reference's own parallel axis is the item column (slim_elastic.py:296-301, 358-366: a
 * process pool over chunks of columns); users are independent in recommend_batch.  POSIX
 * threads pull columns / row blocks from an atomic counter; every worker owns its scratch, so
 * the per-column / per-row arithmetic is the single-threaded code above, unchanged.
 * Fit output: fixed stride `cap` per column (out_idx/out_val[c * cap ...], out_cnt[c]). */
#include <pthread.h>
#include <stdatomic.h>

typedef struct {
    int32_t n_users, n_items;
    const float *X_data; const int32_t *X_indices; const int32_t *X_indptr;
    int32_t n_cols; const int32_t *cols;
    double alpha, l1_ratio, tol; int32_t max_iter; uint32_t seed; int32_t positive, top_features;
    int32_t cap; int32_t *out_cnt; int32_t *out_idx; float *out_val; int32_t *n_iter_out;
    atomic_int next;
} fit_mt_job;

static void *fit_mt_worker(void *arg)
{
    fit_mt_job *J = (fit_mt_job *)arg;
    for (;;) {
        const int32_t c = atomic_fetch_add(&J->next, 1);
        if (c >= J->n_cols) break;
        int32_t it = 0; float gap = 0.0f;
        J->out_cnt[c] = slim_oracle_fit_column(J->n_users, J->n_items, J->X_data, J->X_indices, J->X_indptr,
                                               J->cols[c], J->alpha, J->l1_ratio, J->tol, J->max_iter, J->seed,
                                               J->positive, J->top_features,
                                               J->out_idx + (int64_t)c * J->cap, J->out_val + (int64_t)c * J->cap,
                                               &it, &gap);
        if (J->n_iter_out) J->n_iter_out[c] = it;
    }
    return NULL;
}

int32_t slim_oracle_fit_columns_mt(int32_t n_users, int32_t n_items,
                                   const float *X_data, const int32_t *X_indices, const int32_t *X_indptr,
                                   int32_t n_cols, const int32_t *cols,
                                   double alpha, double l1_ratio, double tol,
                                   int32_t max_iter, uint32_t seed, int32_t positive, int32_t top_features,
                                   int32_t cap, int32_t *out_cnt, int32_t *out_idx, float *out_val,
                                   int32_t *n_iter_out, int32_t n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    fit_mt_job J = { n_users, n_items, X_data, X_indices, X_indptr, n_cols, cols, alpha, l1_ratio, tol, max_iter,
                     seed, positive, top_features, cap, out_cnt, out_idx, out_val, n_iter_out, 0 };
    pthread_t th[256];
    int32_t started = 0;
    for (int32_t t = 0; t < n_threads - 1; t++)
        if (pthread_create(&th[started], NULL, fit_mt_worker, &J) == 0) started++;
    fit_mt_worker(&J);
    for (int32_t t = 0; t < started; t++) pthread_join(th[t], NULL);
    return started + 1;
}

typedef struct {
    int32_t n_rows; const int32_t *Xb_indptr; const int32_t *Xb_indices; const float *Xb_data;
    const int32_t *W_indptr; const int32_t *W_indices; const float *W_data;
    int32_t n_cols, top_k, filter, dense, use_f64;
    int32_t *ids; float *scores; int32_t *counts;
    int32_t block;
    atomic_int next;
} rec_mt_job;

static void *rec_mt_worker(void *arg)
{
    rec_mt_job *J = (rec_mt_job *)arg;
    for (;;) {
        const int32_t b = atomic_fetch_add(&J->next, 1);
        const int64_t r0 = (int64_t)b * J->block;
        if (r0 >= J->n_rows) break;
        const int32_t n = (int32_t)((r0 + J->block <= J->n_rows) ? J->block : J->n_rows - r0);
        /* rows [r0, r0 + n): the row pointers are absolute offsets into Xb_indices / Xb_data */
        slim_oracle_recommend_batch(n, J->Xb_indptr + r0, J->Xb_indices, J->Xb_data, J->W_indptr, J->W_indices,
                                    J->W_data, J->n_cols, J->top_k, J->filter, J->dense, J->use_f64,
                                    J->ids + r0 * J->top_k, J->scores + r0 * J->top_k, J->counts + r0);
    }
    return NULL;
}

int32_t slim_oracle_recommend_batch_mt(int32_t n_rows, const int32_t *Xb_indptr, const int32_t *Xb_indices,
                                       const float *Xb_data,
                                       const int32_t *W_indptr, const int32_t *W_indices, const float *W_data,
                                       int32_t n_cols, int32_t top_k, int32_t filter, int32_t dense,
                                       int32_t use_f64,
                                       int32_t *ids, float *scores, int32_t *counts, int32_t n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    rec_mt_job J = { n_rows, Xb_indptr, Xb_indices, Xb_data, W_indptr, W_indices, W_data, n_cols, top_k, filter,
                     dense, use_f64, ids, scores, counts, 256, 0 };
    pthread_t th[256];
    int32_t started = 0;
    for (int32_t t = 0; t < n_threads - 1; t++)
        if (pthread_create(&th[started], NULL, rec_mt_worker, &J) == 0) started++;
    rec_mt_worker(&J);
    for (int32_t t = 0; t < started; t++) pthread_join(th[t], NULL);
    return started + 1;
}
