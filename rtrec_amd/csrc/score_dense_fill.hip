// rtrec_amd/csrc/score_dense_fill.hip -- DENSE mode: complete a short fast-pass list with zero-score columns.
//
// Replaces (reference): the tail of _dense_topk_indicies (slim_elastic.py:745-778) for a user with fewer than top_k positive
// scores: `argsort(scores)[-k:][::-1]` over ALL columns -- after the positives come the zero-score columns, ordered by the
// canonical tie rule of DESIGN.md D1 (what a stable argsort yields: the higher column id first); interacted items are -inf.
//
// The fast pass (feature-row / segment kernels with dense_rule) lists non-zero sums only and flags every row whose leading
// top_k entries are not all positive.  On a full W few rows are flagged; on a COLUMN SHARD nearly all are (a user's positive
// scores sit in a few shards), and handing those to the tiled DENSE kernel costs more than the fast pass saved.  For a W and
// ratings that are all positive and normal (>= 1e-18: no product underflows, no sum cancels) a list that is not full holds
// EVERY column the user's row touches, each with a positive score -- so the rest of the shard's columns score exactly +0.0 and
// the answer is the list followed by the highest column ids that are neither in it nor (filter_interacted) rated by the user.
// One wave per flagged row: rows it can complete (list shorter than top_k, all positive, no equal neighbours) are completed
// in place, the others (ties, non-positive entries) are passed on in d_flagged_out for the tiled kernel.
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

namespace rtrec {

struct FillArgs {
    const int *row_ids; const int *xb_ptr; const int *xb_col; int n_x_rows;
    int col_lo, col_hi, top_k, filter;
    int *out_id; float *out_score; uint32_t *out_aux; int *out_cnt;
    const int *flag_in; int *flag_out;
};

constexpr int kFillWindow = 2048;        // column ids per step: 32 per lane

__global__ __launch_bounds__(64) void dense_fill_kernel(FillArgs a) {
    __shared__ uint32_t bm[64];
    const int lane = lane_id();
    const int n_flag = a.flag_in[0];
    for (int f = blockIdx.x; f < n_flag; f += gridDim.x) {
        const int row = a.flag_in[1 + f];
        const int cnt = a.out_cnt[row];
        const long long o = static_cast<long long>(row) * a.top_k;
        const float s = lane < cnt ? a.out_score[o + lane] : 0.0f;
        const int id = lane < cnt ? a.out_id[o + lane] : -1;
        const float below = __shfl_down(s, 1, 64);
        const bool ok = cnt < a.top_k && !__ballot(lane < cnt && !(s > 0.0f)) && !__ballot(lane + 1 < cnt && s == below);
        if (!ok) {
            if (lane == 0) a.flag_out[1 + atomicAdd(a.flag_out, 1)] = row;
            continue;
        }
        const int xr = a.row_ids ? a.row_ids[row] : row;
        int a0 = 0, n_a = 0;
        if (a.filter && xr >= 0 && xr < a.n_x_rows) { a0 = a.xb_ptr[xr]; n_a = a.xb_ptr[xr + 1] - a0; }
        int have = cnt;
        int hi = a.col_hi;
        while (have < a.top_k && hi > a.col_lo) {
            const int lo = hi - kFillWindow > a.col_lo ? hi - kFillWindow : a.col_lo;
            bm[lane] = 0u;
            __syncthreads();
            for (int i = lane; i < n_a; i += 64) {              // the user's own items in the window
                const int item = a.xb_col[a0 + i];
                if (item >= lo && item < hi) atomicOr(&bm[(hi - 1 - item) >> 5], 1u << ((hi - 1 - item) & 31));
            }
            if (id >= lo && id < hi) atomicOr(&bm[(hi - 1 - id) >> 5], 1u << ((hi - 1 - id) & 31));      // ... and the listed columns
            __syncthreads();
            const int span = hi - lo, base = lane * 32;         // bit b of the window = column hi - 1 - b
            uint32_t allowed = ~bm[lane];
            if (base >= span) allowed = 0u;
            else if (span - base < 32) allowed &= (1u << (span - base)) - 1u;
            const int n = __builtin_popcount(allowed);
            int incl = n;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int up = __shfl_up(incl, d, 64);
                incl += lane >= d ? up : 0;
            }
            int pos = have + incl - n;
            while (allowed && pos < a.top_k) {
                const int j = __builtin_ctz(allowed);
                allowed &= allowed - 1u;
                a.out_id[o + pos] = hi - 1 - (base + j);
                a.out_score[o + pos] = 0.0f;
                if (a.out_aux) a.out_aux[o + pos] = 0u;
                ++pos;
            }
            const int total = have + readlane_i(incl, 63);
            have = total < a.top_k ? total : a.top_k;
            hi = lo;
            __syncthreads();
        }
        if (lane == 0) a.out_cnt[row] = have;
    }
}

}  // namespace rtrec

using namespace rtrec;

extern "C" int rtrec_slim_dense_fill(int32_t n_rows, const int32_t *d_row_ids, const int32_t *d_xb_ptr, const int32_t *d_xb_col,
                                     int32_t n_x_rows, int32_t col_lo, int32_t col_hi, int32_t top_k, int32_t filter_interacted,
                                     int32_t *d_out_ids, float *d_out_scores, uint32_t *d_out_aux, int32_t *d_out_count,
                                     const int32_t *d_flagged_in, int32_t *d_flagged_out, void *stream) {
    if (n_rows < 0 || top_k <= 0 || top_k > 64 || col_lo < 0 || col_hi < col_lo || n_x_rows < 0) return RTREC_ERR_INVALID_ARG;
    if (!d_out_ids || !d_out_scores || !d_out_count || !d_flagged_in || !d_flagged_out || d_flagged_in == d_flagged_out)
        return RTREC_ERR_INVALID_ARG;
    if (filter_interacted && (!d_xb_ptr || !d_xb_col)) return RTREC_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (hipMemsetAsync(d_flagged_out, 0, 4, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    if (n_rows == 0) return RTREC_OK;
    FillArgs a{};
    a.row_ids = d_row_ids; a.xb_ptr = d_xb_ptr; a.xb_col = d_xb_col; a.n_x_rows = n_x_rows;
    a.col_lo = col_lo; a.col_hi = col_hi; a.top_k = top_k; a.filter = filter_interacted ? 1 : 0;
    a.out_id = d_out_ids; a.out_score = d_out_scores; a.out_aux = d_out_aux; a.out_cnt = d_out_count;
    a.flag_in = d_flagged_in; a.flag_out = d_flagged_out;
    const int grid = n_rows < 8192 ? n_rows : 8192;
    hipLaunchKernelGGL(dense_fill_kernel, dim3(grid), dim3(64), 0, st, a);
    return rtrec::launch_status();
}
