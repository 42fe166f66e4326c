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
//
// SIGNED weights or ratings (round 4; positive_only=False, slim_elastic.py:187): the sign argument is gone, but a float32
// sum of n rounded products is within (n + 1) 2^-24 (1 + ...) sum |x w| of the exact one, and sum |x w| <= B_u = sum_i
// |x_ui| max_c |w_ic| for EVERY column.  With abs_slack[user] >= 2 (n_u + 2) 2^-24 B_u (+ an underflow term) from the caller:
// a column outside the list has a float32 score <= m32 or a float32 sum of exactly 0 (SPARSE mode: not a candidate there,
// but its float64 sum may be non-zero), hence a float64 score <= max(m32, 0) + slack; when the top_k-th float64 score of
// the candidates is above that, the sorted candidates are the answer.  Flagged for the float64 tiled kernel: rows that fail
// the test, rows whose list is not full (a cancelled float32 sum may hide a column), a candidate whose float64 sum is
// exactly 0 (no stored product in the reference), exact ties.
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

namespace rtrec {
namespace {

constexpr int kRfWaves = 4;
constexpr int kRfItems = 1024;      // item slots staged in LDS per wave (8 KB), divided among the rows the wave works on; a longer
                                    // row is searched in global memory

__device__ __forceinline__ double rf_shfl_d(double v, int src) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl(static_cast<int>(b & 0xffffffffll), src, 64);
    const int hi = __shfl(static_cast<int>(b >> 32), src, 64);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}

// P lanes per row (the smallest power of two >= top_k + 1), 64 / P rows per wave: lane = slot * P + candidate.
template <int P>
__global__ __launch_bounds__(kRfWaves * 64) void refine_f64_kernel(
    int n_rows, const int *__restrict__ row_ids, const int *__restrict__ xb_ptr, const int *__restrict__ xb_col,
    const float *__restrict__ xb_val, int n_x_rows, int n_items, const int *__restrict__ wc_ptr, const int *__restrict__ wc_row,
    const float *__restrict__ wc_val, int top_k, const int *__restrict__ in_ids, const float *__restrict__ in_scores,
    const int *__restrict__ in_count, double rel_margin, const double *__restrict__ abs_slack, int *__restrict__ out_ids,
    float *__restrict__ out_scores, double *__restrict__ out_scores64, int *__restrict__ out_count, int *__restrict__ flagged) {
    constexpr int RPW = 64 / P;                 // rows per wave
    constexpr int CAP = kRfItems / RPW;         // staged items per row
    __shared__ int s_col[kRfWaves][kRfItems];
    __shared__ float s_val[kRfWaves][kRfItems];
    const int lane = lane_id();
    const int slot = lane / P, cand = lane % P;
    int *lcol = s_col[static_cast<int>(threadIdx.x) >> 6] + slot * CAP;
    float *lval = s_val[static_cast<int>(threadIdx.x) >> 6] + slot * CAP;
    const long long wave = (static_cast<long long>(blockIdx.x) * kRfWaves) + (static_cast<int>(threadIdx.x) >> 6);
    const long long n_waves = static_cast<long long>(gridDim.x) * kRfWaves;
    const int kin = top_k + 1;
    const double ninf = -__builtin_huge_val();
    const unsigned long long gmask = (P == 64 ? ~0ull : ((1ull << P) - 1ull)) << (slot * P);
    for (long long base = wave * RPW; base < n_rows; base += n_waves * RPW) {
        const long long row = base + slot;
        const bool live = row < n_rows;
        int a0 = 0, n_a = 0, n = 0;
        double slack = 0.0;
        if (live) {
            const int xrow = row_ids ? row_ids[row] : static_cast<int>(row);
            if (xrow >= 0 && xrow < n_x_rows) { a0 = xb_ptr[xrow]; n_a = xb_ptr[xrow + 1] - a0; if (abs_slack) slack = abs_slack[xrow]; }
            n = min(in_count[row], kin);
        }
        // the row's items in LDS: its candidates' entries are looked up in them (a wave's LDS traffic is program-ordered)
        const bool staged = n_a <= CAP;
        if (staged) for (int q = cand; q < n_a; q += P) { lcol[q] = xb_col[a0 + q]; lval[q] = xb_val[a0 + q]; }
        // the lanes of a row read what OTHER lanes of the wave have just staged: tell the compiler
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- exact float64 score of candidate `cand` of row `slot`
        const bool has = live && cand < n;
        double e = ninf;
        int c = -1;
        if (has) {
            c = in_ids[row * kin + cand];
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
        // ---- rank inside the row's lane group by (score descending, list position ascending); exact ties are the tiled kernel's
        int rank = 0;
        bool tie = false;
        for (int j = 0; j < P; ++j) {
            const int src = slot * P + j;
            const double o = rf_shfl_d(e, src);
            const bool ov = __shfl(has ? 1 : 0, src, 64) != 0;
            if (has && ov && j != cand) {
                rank += (o > e || (o == e && j < cand)) ? 1 : 0;
                tie = tie || (o == e);
            }
        }
        const bool any_tie = (__ballot(has && tie) & gmask) != 0ull;
        // the top_k-th best float64 score against what a column outside the list can reach
        const unsigned long long at = __ballot(has && rank == top_k - 1) & gmask;
        const double e_k = rf_shfl_d(e, at ? static_cast<int>(__builtin_ctzll(at)) : lane);
        bool unsafe = false;
        if (abs_slack) {                  // signed W / ratings: absolute slack, and no shortcut for lists that are not full
            unsafe = live && n_a > 0;         // (an empty list too: float32 products may all have underflowed to 0)
            if (live && n == kin) {
                const double m32 = static_cast<double>(in_scores[row * kin + top_k]);
                unsafe = !(e_k > (m32 > 0.0 ? m32 : 0.0) + slack);
            }
            if ((__ballot(has && e == 0.0) & gmask) != 0ull) unsafe = true;      // no stored product in the reference
        } else if (live && n == kin) {
            const double m32 = static_cast<double>(in_scores[row * kin + top_k]);
            unsafe = !(e_k > m32 * (1.0 + rel_margin));
        }
        const int n_fin = min(n, top_k);
        if (has && rank < top_k) {
            const long long o = row * top_k + rank;
            out_ids[o] = c;
            out_scores[o] = static_cast<float>(e);
            out_scores64[o] = e;
        }
        if (live && cand >= n_fin && cand < top_k) {
            const long long o = row * top_k + cand;
            out_ids[o] = -1;
            out_scores[o] = -__builtin_huge_valf();
            out_scores64[o] = ninf;
        }
        if (live && cand == 0) {
            out_count[row] = n_fin;
            if (any_tie || unsafe) flagged[1 + atomicAdd(flagged, 1)] = static_cast<int>(row);
        }
    }
}

}  // namespace
}  // namespace rtrec

extern "C" int rtrec_slim_refine_topk_f64(int32_t n_rows, const int32_t *d_row_ids, const int32_t *d_xb_ptr, const int32_t *d_xb_col,
                                          const float *d_xb_val, int32_t n_x_rows, int32_t n_items, const int32_t *d_wc_ptr,
                                          const int32_t *d_wc_row, const float *d_wc_val, int32_t top_k, const int32_t *d_in_ids,
                                          const float *d_in_scores, const int32_t *d_in_count, double rel_margin,
                                          const double *d_abs_slack,
                                          int32_t *d_out_ids, float *d_out_scores, double *d_out_scores64, int32_t *d_out_count,
                                          int32_t *d_flagged, void *stream) {
    if (n_rows < 0 || top_k <= 0 || top_k > 63 || n_items <= 0 || n_x_rows < 0 || !(rel_margin >= 0.0)) return RTREC_ERR_INVALID_ARG;
    if (n_rows == 0) return RTREC_OK;
    if (!d_xb_ptr || !d_xb_col || !d_xb_val || !d_wc_ptr || !d_wc_row || !d_wc_val || !d_in_ids || !d_in_scores || !d_in_count ||
        !d_out_ids || !d_out_scores || !d_out_scores64 || !d_out_count || !d_flagged)
        return RTREC_ERR_INVALID_ARG;
    (void)hipGetLastError();
    int P = 2;
    while (P < top_k + 1) P *= 2;                       // lanes per row
    const long long rows_per_wg = static_cast<long long>(rtrec::kRfWaves) * (64 / P);
    const long long want = (static_cast<long long>(n_rows) + rows_per_wg - 1) / rows_per_wg;
    const unsigned grid = static_cast<unsigned>(want < 16384 ? want : 16384);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define RTREC_RF_LAUNCH(P_)                                                                                                      \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(rtrec::refine_f64_kernel<P_>), dim3(grid), dim3(rtrec::kRfWaves * 64), 0, st, n_rows, d_row_ids,  \
                       d_xb_ptr, d_xb_col, d_xb_val, n_x_rows, n_items, d_wc_ptr, d_wc_row, d_wc_val, top_k, d_in_ids, d_in_scores,     \
                       d_in_count, rel_margin, d_abs_slack, d_out_ids, d_out_scores, d_out_scores64, d_out_count, d_flagged)
    switch (P) {
        case 2: RTREC_RF_LAUNCH(2); break;
        case 4: RTREC_RF_LAUNCH(4); break;
        case 8: RTREC_RF_LAUNCH(8); break;
        case 16: RTREC_RF_LAUNCH(16); break;
        case 32: RTREC_RF_LAUNCH(32); break;
        default: RTREC_RF_LAUNCH(64); break;
    }
#undef RTREC_RF_LAUNCH
    return rtrec::launch_status();
}
