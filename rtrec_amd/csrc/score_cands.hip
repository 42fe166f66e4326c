// rtrec_amd/csrc/score_cands.hip -- CANDIDATES mode for request-sized calls: rank a given list of items for a few users.
//
// Reference: SLIMElastic.recommend_batch with candidate_item_ids (/root/reference/rtrec/models/internal/slim_elastic.py:723-735):
//     scores = X[users] @ W[:, candidates]   (dense, zeros included; filter_interacted is ignored)
//     np.argsort(scores)[-top_k:][::-1]       -> candidates by score descending; ties: the LATER candidate first (DESIGN D1)
// The tiled kernel serves this mode by scoring every column of W and masking with a rank array of n_items entries that the
// host builds and uploads per call.  For a re-ranking request (one user, a few hundred candidates) that is all overhead:
// here one wave takes one row, stages the user's items in LDS, and every lane computes the scores of its candidates from
// W's CSC columns -- the entries whose row the user rates (binary search), acc = acc + x * w with one rounded product and
// one rounded add in ascending item order: scipy's csr_matmat order per output column, bit-identical sums -- then the
// top_k are taken out one by one (wave arg-max over the LDS score array).
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

namespace rtrec {
namespace {

constexpr int kCdItems = 2048;          // items of a row staged in LDS (longer rows: searched in global memory)
constexpr int kCdMaxCands = 8192;       // scores kept in LDS (float32: 32 KB, float64: 64 KB)

template <typename ACC>
__global__ __launch_bounds__(256) void score_cands_kernel(
    int n_rows, const int *__restrict__ row_ids, const int *__restrict__ xb_ptr, const int *__restrict__ xb_col,
    const float *__restrict__ xb_val, int n_x_rows, int n_items, const int *__restrict__ wc_ptr, const int *__restrict__ wc_row,
    const float *__restrict__ wc_val, const int *__restrict__ cands, int n_cands, int top_k, int *__restrict__ out_ids,
    float *__restrict__ out_scores, double *__restrict__ out_scores64, int *__restrict__ out_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int *lcol = reinterpret_cast<int *>(smem);
    float *lval = reinterpret_cast<float *>(lcol + kCdItems);
    ACC *sc = reinterpret_cast<ACC *>(lval + kCdItems);
    const int lane = lane_id();
    const int tid = static_cast<int>(threadIdx.x), nthreads = static_cast<int>(blockDim.x);
    for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const int xrow = row_ids ? row_ids[row] : row;
        int a0 = 0, n_a = 0;
        if (xrow >= 0 && xrow < n_x_rows) { a0 = xb_ptr[xrow]; n_a = xb_ptr[xrow + 1] - a0; }
        const bool staged = n_a <= kCdItems;
        if (staged) for (int q = tid; q < n_a; q += nthreads) { lcol[q] = xb_col[a0 + q]; lval[q] = xb_val[a0 + q]; }
        // ---- scores of the candidates (thread t: candidates t, t + threads, ...).  A column's entries are requested four at a
        //      time (rows and weights of the next four entries in flight together), looked up in the row's items, and added in
        //      entry order
        auto lookup = [&](int i, float &x) -> bool {
            int lo = 0, hi = n_a;
            if (staged) {
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (lcol[mid] < i) lo = mid + 1; else hi = mid; }
                if (lo < n_a && lcol[lo] == i) { x = lval[lo]; return true; }
            } else {
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (xb_col[a0 + mid] < i) lo = mid + 1; else hi = mid; }
                if (lo < n_a && xb_col[a0 + lo] == i) { x = xb_val[a0 + lo]; return true; }
            }
            return false;
        };
        auto add = [&](ACC &acc, float x, float w) {
            if constexpr (sizeof(ACC) == 4) acc = __fadd_rn(acc, __fmul_rn(x, w));
            else acc = __dadd_rn(acc, __dmul_rn(static_cast<double>(x), static_cast<double>(w)));
        };
        __syncthreads();                                   // the staged items are visible to every wave
        for (int p = tid; p < n_cands; p += nthreads) {
            const int c = cands[p];
            ACC acc = static_cast<ACC>(0);
            if (c >= 0 && c < n_items) {
                const int qe = wc_ptr[c + 1];
                int q = wc_ptr[c];
                for (; q + 4 <= qe; q += 4) {
                    int ii[4];
                    float ww[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { ii[j] = wc_row[q + j]; ww[j] = wc_val[q + j]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x;
                        if (lookup(ii[j], x)) add(acc, x, ww[j]);
                    }
                }
                for (; q < qe; ++q) {
                    float x;
                    if (lookup(wc_row[q], x)) add(acc, x, wc_val[q]);
                }
            }
            sc[p] = acc;
        }
        __syncthreads();
        if (tid >= 64) { __syncthreads(); continue; }      // selection: wave 0 (the others wait at the row's last barrier)
        // ---- the best top_k, one by one: (score descending, position descending)
        const int n_fin = min(top_k, n_cands);
        for (int r = 0; r < n_fin; ++r) {
            ACC best = static_cast<ACC>(0);
            int bp = -1;
            for (int p = lane; p < n_cands; p += 64) {
                const ACC v = sc[p];
                if (bp < 0 ? (v == v) : (v > best || (v == best && p > bp))) { best = v; bp = p; }      // taken entries hold NaN: never chosen
            }
            // wave arg-max
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const ACC ob = shfl_xor_t(best, m);
                const int op = shfl_xor_t(bp, m);
                if (op >= 0 && (bp < 0 || ob > best || (ob == best && op > bp))) { best = ob; bp = op; }
            }
            if (lane == 0) {
                const long long o = static_cast<long long>(row) * top_k + r;
                out_ids[o] = cands[bp];
                out_scores[o] = static_cast<float>(best);
                if (out_scores64) out_scores64[o] = static_cast<double>(best);
            }
            if (lane == (bp & 63)) sc[bp] = static_cast<ACC>(__builtin_nanf(""));
        }
        for (int r = n_fin + lane; r < top_k; r += 64) {
            const long long o = static_cast<long long>(row) * top_k + r;
            out_ids[o] = -1;
            out_scores[o] = -__builtin_huge_valf();
            if (out_scores64) out_scores64[o] = -__builtin_huge_val();
        }
        if (lane == 0) out_count[row] = n_fin;
        __syncthreads();                                   // the row is done: LDS may be overwritten
    }
}

}  // namespace
}  // namespace rtrec

extern "C" int rtrec_slim_score_candidates(int32_t n_rows, const int32_t *d_row_ids, const int32_t *d_xb_ptr, const int32_t *d_xb_col,
                                           const float *d_xb_val, int32_t n_x_rows, int32_t n_items, const int32_t *d_wc_ptr,
                                           const int32_t *d_wc_row, const float *d_wc_val, const int32_t *d_cands, int32_t n_cands,
                                           int32_t top_k, int32_t acc_f64, int32_t *d_out_ids, float *d_out_scores,
                                           double *d_out_scores64, int32_t *d_out_count, void *stream) {
    if (n_rows < 0 || top_k <= 0 || n_cands <= 0 || n_cands > rtrec::kCdMaxCands || n_items <= 0 || n_x_rows < 0) return RTREC_ERR_INVALID_ARG;
    if (n_rows == 0) return RTREC_OK;
    if (!d_xb_ptr || !d_xb_col || !d_xb_val || !d_wc_ptr || !d_wc_row || !d_wc_val || !d_cands || !d_out_ids || !d_out_scores ||
        !d_out_count || (acc_f64 && !d_out_scores64))
        return RTREC_ERR_INVALID_ARG;
    (void)hipGetLastError();
    const unsigned grid = static_cast<unsigned>(n_rows < 16384 ? n_rows : 16384);
    const size_t lds = static_cast<size_t>(rtrec::kCdItems) * 8 + static_cast<size_t>(n_cands) * (acc_f64 ? 8 : 4);
    if (lds > 64u * 1024u) return RTREC_ERR_UNSUPPORTED;           // (float64: at most 6,144 candidates)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned threads = n_cands > 128 ? 256u : (n_cands > 64 ? 128u : 64u);       // one candidate per thread where the list allows
    if (acc_f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rtrec::score_cands_kernel<double>), dim3(grid), dim3(threads), lds, st, n_rows, d_row_ids, d_xb_ptr,
                           d_xb_col, d_xb_val, n_x_rows, n_items, d_wc_ptr, d_wc_row, d_wc_val, d_cands, n_cands, top_k, d_out_ids,
                           d_out_scores, d_out_scores64, d_out_count);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rtrec::score_cands_kernel<float>), dim3(grid), dim3(threads), lds, st, n_rows, d_row_ids, d_xb_ptr,
                           d_xb_col, d_xb_val, n_x_rows, n_items, d_wc_ptr, d_wc_row, d_wc_val, d_cands, n_cands, top_k, d_out_ids,
                           d_out_scores, nullptr, d_out_count);
    return rtrec::launch_status();
}
