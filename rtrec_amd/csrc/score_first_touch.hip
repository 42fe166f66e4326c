// rtrec_amd/csrc/score_first_touch.hip -- SPARSE mode over COLUMN SHARDS: the tie key of every list entry.
//
// Replaces (reference): the order `_sparse_topk_indicies` (slim_elastic.py:782-818) gives columns with EQUAL scores -- a
// stable sort over scipy's csr_matmat output order, which is the reverse of the order in which the user's row first touched
// the columns.  The kernels' tie key is therefore aux = the position, in the user's row of X, of the first item whose row of
// W holds a weight in the column (higher position first, then higher id: common.hip.h cand_better).
//
// On one GPU a row's key is only computed when its list holds a tie (the exact-tie pass / fr_ties_kernel); every other entry
// carries aux = 0.  Across COLUMN SHARDS that is not enough: two columns of different shards can tie while neither shard sees
// a tie, and merge_topk_kernel would then order them by id.  (Found by tools/fuzz_score.py's column-shard draws, round 4:
// integer ratings and weights.)  A rank that holds only part of W's columns therefore completes the key for every entry of
// its lists before the exchange: one lane per entry walks the column's stored rows (ascending item id, CSC) and looks each
// up in the user's row (ascending item id: binary search); the first hit is the lowest item, i.e. the lowest position.
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

namespace rtrec {

struct FirstTouchArgs {
    int n_rows; const int *row_ids; const int *xb_ptr; const int *xb_col; int n_x_rows;
    int n_items; const int *wc_ptr; const int *wc_row;
    int top_k; const int *ids; const int *cnt; uint32_t *aux;
};

__global__ __launch_bounds__(256) void first_touch_aux_kernel(FirstTouchArgs a) {
    const long long total = static_cast<long long>(a.n_rows) * a.top_k;
    for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < total;
         e += static_cast<long long>(gridDim.x) * blockDim.x) {
        const int row = static_cast<int>(e / a.top_k);
        const int slot = static_cast<int>(e - static_cast<long long>(row) * a.top_k);
        uint32_t key = 0u;
        const int c = slot < a.cnt[row] ? a.ids[e] : -1;
        if (c >= 0 && c < a.n_items) {
            const int xr = a.row_ids ? a.row_ids[row] : row;
            if (xr >= 0 && xr < a.n_x_rows) {
                const int a0 = a.xb_ptr[xr], n_a = a.xb_ptr[xr + 1] - a0;
                const int *items = a.xb_col + a0;
                const int j1 = a.wc_ptr[c + 1];
                for (int j = a.wc_ptr[c]; j < j1; ++j) {
                    const int r = a.wc_row[j];
                    int lo = 0, hi = n_a;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (items[mid] < r) lo = mid + 1; else hi = mid;
                    }
                    if (lo < n_a && items[lo] == r) { key = static_cast<uint32_t>(lo); break; }
                }
            }
        }
        a.aux[e] = key;
    }
}

}  // namespace rtrec

using namespace rtrec;

extern "C" int rtrec_slim_first_touch_aux(int32_t n_rows, const int32_t *d_row_ids, const int32_t *d_xb_ptr, const int32_t *d_xb_col,
                                          int32_t n_x_rows, int32_t n_items, const int32_t *d_wc_ptr, const int32_t *d_wc_row,
                                          int32_t top_k, const int32_t *d_ids, const int32_t *d_count, uint32_t *d_aux, void *stream) {
    if (n_rows < 0 || n_x_rows < 0 || n_items <= 0 || top_k <= 0) return RTREC_ERR_INVALID_ARG;
    if (n_rows == 0) return RTREC_OK;
    if (!d_xb_ptr || !d_xb_col || !d_wc_ptr || !d_wc_row || !d_ids || !d_count || !d_aux) return RTREC_ERR_INVALID_ARG;
    FirstTouchArgs a{};
    a.n_rows = n_rows; a.row_ids = d_row_ids; a.xb_ptr = d_xb_ptr; a.xb_col = d_xb_col; a.n_x_rows = n_x_rows;
    a.n_items = n_items; a.wc_ptr = d_wc_ptr; a.wc_row = d_wc_row; a.top_k = top_k; a.ids = d_ids; a.cnt = d_count; a.aux = d_aux;
    const long long total = static_cast<long long>(n_rows) * top_k;
    const long long blocks = (total + 255) / 256;
    (void)hipGetLastError();
    hipLaunchKernelGGL(first_touch_aux_kernel, dim3(static_cast<unsigned>(blocks < 65536 ? blocks : 65536)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    return rtrec::launch_status();
}
