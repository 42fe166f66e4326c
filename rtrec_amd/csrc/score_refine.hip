// rtrec_amd/csrc/score_refine.hip -- float64 answers from a float32 fast pass (SPARSE mode).
//
// A W fitted serially is float64 on the host (/root/reference/rtrec/models/internal/slim_elastic.py:252; its values are
// float32 numbers: scikit-learn keeps float32 coefficients for a float32 X), so the reference's `X[users] @ W`
// (:707-708) accumulates float64 sums.  The fast kernels (csrc/score_seg.hip.h, score_frows_kernel) accumulate
// float32.  For NON-NEGATIVE ratings and weights their lists still pin the float64 answer down:
//
//   * the fast pass is asked for top_k + 1 columns per row (by float32 score, descending);
//   * this kernel recomputes the float64 score of those candidates exactly as the reference does -- column c of W in
//     CSC form (rows ascending), the entries whose row the user rates (binary search in the user's ascending item list),
//     acc = acc + (double)x * (double)w with one rounded product and one rounded add, ascending item order: scipy's
//     csr_matmat order per output column -- and sorts them;
//   * every column OUTSIDE the list has a float32 score <= m32, the (top_k + 1)-th of the list; all addends are >= 0,
//     so its float64 score is at most m32 * (1 + margin), margin >= 2 (n + 1) 2^-24 for columns of at most n weights
//     (float32 sum >= exact (1 - (n + 1) u), float64 sum <= exact (1 + (n + 1) u^2...)).  When the top_k-th best float64
//     score of the candidates is above that bound, no outsider can be among the best top_k: the sorted candidates are
//     the reference's answer.  Otherwise -- and when two leading candidates tie exactly -- the row is flagged and the
//     caller scores it with the float64 tiled kernel.
// Rows with fewer than top_k + 1 non-zero columns need no margin: the list holds every non-zero column (the caller
// makes sure products cannot underflow in float32: non-zero in float64 <=> non-zero in float32).
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

namespace rtrec {
namespace {

constexpr int kRfWaves = 4;
constexpr int kRfItems = 1024;      // items of a row staged in LDS per wave (8 KB); longer rows are searched in global memory

__global__ __launch_bounds__(kRfWaves * 64) void refine_f64_kernel(
    int n_rows, const int *__restrict__ row_ids, const int *__restrict__ xb_ptr, const int *__restrict__ xb_col,
    const float *__restrict__ xb_val, int n_x_rows, int n_items, const int *__restrict__ wc_ptr, const int *__restrict__ wc_row,
    const float *__restrict__ wc_val, int top_k, const int *__restrict__ in_ids, const float *__restrict__ in_scores,
    const int *__restrict__ in_count, double rel_margin, int *__restrict__ out_ids, float *__restrict__ out_scores,
    double *__restrict__ out_scores64, int *__restrict__ out_count, int *__restrict__ flagged) {
    __shared__ int s_col[kRfWaves][kRfItems];
    __shared__ float s_val[kRfWaves][kRfItems];
    const int lane = lane_id();
    int *lcol = s_col[static_cast<int>(threadIdx.x) >> 6];
    float *lval = s_val[static_cast<int>(threadIdx.x) >> 6];
    const int wave = (static_cast<int>(blockIdx.x) * kRfWaves) + (static_cast<int>(threadIdx.x) >> 6);
    const int n_waves = static_cast<int>(gridDim.x) * kRfWaves;
    const int kin = top_k + 1;
    const double ninf = -__builtin_huge_val();
    for (int row = wave; row < n_rows; row += n_waves) {
        const int xrow = row_ids ? row_ids[row] : row;
        int a0 = 0, n_a = 0;
        if (xrow >= 0 && xrow < n_x_rows) { a0 = xb_ptr[xrow]; n_a = xb_ptr[xrow + 1] - a0; }
        const int n = min(in_count[row], kin);
        // the row's items in LDS: every candidate's entries are looked up in them (a wave's LDS traffic is program-ordered)
        const bool staged = n_a <= kRfItems;
        if (staged) for (int q = lane; q < n_a; q += 64) { lcol[q] = xb_col[a0 + q]; lval[q] = xb_val[a0 + q]; }
        // ---- exact float64 score of candidate `lane`
        double e = ninf;
        int c = -1;
        if (lane < n) {
            c = in_ids[static_cast<long long>(row) * kin + lane];
            double acc = 0.0;
            if (c >= 0 && c < n_items) {
                for (int q = wc_ptr[c]; q < wc_ptr[c + 1]; ++q) {
                    const int i = wc_row[q];
                    int lo = 0, hi = n_a;                                   // first position with item >= i
                    if (staged) {
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (lcol[mid] < i) lo = mid + 1; else hi = mid;
                        }
                        if (lo < n_a && lcol[lo] == i)
                            acc = __dadd_rn(acc, __dmul_rn(static_cast<double>(lval[lo]), static_cast<double>(wc_val[q])));
                    } else {
                        while (lo < hi) {
                            const int mid = (lo + hi) >> 1;
                            if (xb_col[a0 + mid] < i) lo = mid + 1; else hi = mid;
                        }
                        if (lo < n_a && xb_col[a0 + lo] == i)
                            acc = __dadd_rn(acc, __dmul_rn(static_cast<double>(xb_val[a0 + lo]), static_cast<double>(wc_val[q])));
                    }
                }
            }
            e = acc;
        }
        // ---- rank by (score descending, list position ascending); exact ties among the candidates are the tiled kernel's
        int rank = 0;
        bool tie = false;
        for (int j = 0; j < n; ++j) {
            const double o = readlane_d(e, j);
            if (lane < n && j != lane) {
                rank += (o > e || (o == e && j < lane)) ? 1 : 0;
                tie = tie || (o == e);
            }
        }
        const bool any_tie = __ballot(lane < n && tie) != 0ull;
        // the top_k-th best float64 score against what a column outside the list can reach
        bool unsafe = false;
        if (n == kin) {
            const unsigned long long at = __ballot(lane < n && rank == top_k - 1);
            const double e_k = readlane_d(e, static_cast<int>(__builtin_ctzll(at)));
            const double m32 = static_cast<double>(in_scores[static_cast<long long>(row) * kin + top_k]);
            unsafe = !(e_k > m32 * (1.0 + rel_margin));
        }
        const int n_fin = min(n, top_k);
        if (lane < n && rank < top_k) {
            const long long o = static_cast<long long>(row) * top_k + rank;
            out_ids[o] = c;
            out_scores[o] = static_cast<float>(e);
            out_scores64[o] = e;
        }
        if (lane >= n_fin && lane < top_k) {
            const long long o = static_cast<long long>(row) * top_k + lane;
            out_ids[o] = -1;
            out_scores[o] = -__builtin_huge_valf();
            out_scores64[o] = ninf;
        }
        if (lane == 0) {
            out_count[row] = n_fin;
            if (any_tie || unsafe) flagged[1 + atomicAdd(flagged, 1)] = row;
        }
    }
}

}  // namespace
}  // namespace rtrec

extern "C" int rtrec_slim_refine_topk_f64(int32_t n_rows, const int32_t *d_row_ids, const int32_t *d_xb_ptr, const int32_t *d_xb_col,
                                          const float *d_xb_val, int32_t n_x_rows, int32_t n_items, const int32_t *d_wc_ptr,
                                          const int32_t *d_wc_row, const float *d_wc_val, int32_t top_k, const int32_t *d_in_ids,
                                          const float *d_in_scores, const int32_t *d_in_count, double rel_margin,
                                          int32_t *d_out_ids, float *d_out_scores, double *d_out_scores64, int32_t *d_out_count,
                                          int32_t *d_flagged, void *stream) {
    if (n_rows < 0 || top_k <= 0 || top_k > 63 || n_items <= 0 || n_x_rows < 0 || !(rel_margin >= 0.0)) return RTREC_ERR_INVALID_ARG;
    if (n_rows == 0) return RTREC_OK;
    if (!d_xb_ptr || !d_xb_col || !d_xb_val || !d_wc_ptr || !d_wc_row || !d_wc_val || !d_in_ids || !d_in_scores || !d_in_count ||
        !d_out_ids || !d_out_scores || !d_out_scores64 || !d_out_count || !d_flagged)
        return RTREC_ERR_INVALID_ARG;
    (void)hipGetLastError();
    const long long want = (static_cast<long long>(n_rows) + rtrec::kRfWaves - 1) / rtrec::kRfWaves;
    const unsigned grid = static_cast<unsigned>(want < 16384 ? want : 16384);
    hipLaunchKernelGGL(rtrec::refine_f64_kernel, dim3(grid), dim3(rtrec::kRfWaves * 64), 0, static_cast<hipStream_t>(stream), n_rows,
                       d_row_ids, d_xb_ptr, d_xb_col, d_xb_val, n_x_rows, n_items, d_wc_ptr, d_wc_row, d_wc_val, top_k, d_in_ids,
                       d_in_scores, d_in_count, rel_margin, d_out_ids, d_out_scores, d_out_scores64, d_out_count, d_flagged);
    return rtrec::launch_status();
}
