// rtrec_amd/csrc/score.hip -- fused user-row x W accumulation + interacted filter + top-k.
//
// Replaces (reference): SLIMElastic.recommend / recommend_batch (slim_elastic.py:628-741), the
// scipy csr_matmat product behind safe_sparse_dot (slim_elastic.py:585,608,626,708,726) and the
// two top-k helpers (_sparse_topk_indicies :782-818, _dense_topk_indicies :744-779).
//
// Layout: W is cut into column tiles of `tile_cols` columns; each tile is a CSR over all item
// rows with tile-local uint16 column ids (6 bytes per stored weight).  One wavefront owns one
// (user row, tile) job: it streams the W rows of the user's items IN ASCENDING ITEM ORDER and
// adds x_ui * W[i, c] into an LDS accumulator with ds_add_f32 / ds_add_f64.  LDS operations of
// one wave execute in program order and a W row never repeats a column, so every accumulator
// receives its addends in exactly scipy's csr_matmat order -> bit-identical scores.  The tile is
// then reduced to its top-(k+1) in registers/LDS and a second kernel merges the tiles of a row.
//
// Tie order of the SPARSE mode (Python's stable sorted() over scipy's reverse-first-touch
// product order) needs the first-touch rank of a column, which costs a second LDS array.  The
// fast pass therefore runs without it, the merge flags rows whose leading k+1 scores contain an
// exact tie, and only those rows are re-scored by the FT (first-touch tracking) instantiation.
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

namespace rtrec {

struct ScoreArgs {
    int n_rows;
    const int *row_ids;   // optional: job r scores CSR row row_ids[r]
    const int *xb_ptr;
    const int *xb_col;
    const float *xb_val;
    int n_items;
    int n_cols;
    int col_offset;
    int tile_cols;
    int n_tiles;
    const int *tile_ptr;
    const uint16_t *w_col;
    const float *w_val;
    const int *col_rank;
    int kk;      // candidates kept per tile (top_k or top_k + 1)
    int filter;
    int mode;
    // per (row, tile) candidate lists
    void *cand_score;  // ACC[n_rows * n_tiles * kk]
    int *cand_id;
    uint32_t *cand_aux;
    int *cand_cnt;     // [n_rows * n_tiles]
    // FT pass: list of rows to re-score and its length (device)
    const int *row_list;
    const int *row_list_len;
    int *queue;
};

constexpr int kListCap = 1024;  // LDS candidate list entries (uint16 column ids)

template <typename ACC>
__device__ __forceinline__ void lds_add(ACC *p, ACC v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Key of an accumulator for selection: invalid columns map to -inf.
template <typename ACC>
__device__ __forceinline__ ACC sel_key(ACC v, bool zero_is_valid) {
    return (!zero_is_valid && v == ACC(0)) ? NegInf<ACC>::value() : v;
}

template <typename ACC, bool FT>
__device__ void score_tile_job(const ScoreArgs &a, int row, int tile, unsigned char *smem) {
    const int lane = lane_id();
    const int S = a.tile_cols;
    ACC *acc = reinterpret_cast<ACC *>(smem);
    uint32_t *ft = reinterpret_cast<uint32_t *>(smem + static_cast<size_t>(S) * sizeof(ACC));
    uint16_t *clist = reinterpret_cast<uint16_t *>(smem + static_cast<size_t>(S) * (sizeof(ACC) + (FT ? 4 : 0)));
    const ACC ninf = NegInf<ACC>::value();

    const int t0 = tile * S;                                   // first shard-local column of the tile
    const int ncol = min(S, a.n_cols - t0);                    // valid columns in this tile
    const int xrow = a.row_ids ? a.row_ids[row] : row;
    const int a0 = a.xb_ptr[xrow];
    const int n_a = a.xb_ptr[xrow + 1] - a0;
    const bool zero_valid = (a.mode != RTREC_TOPK_SPARSE);

    // ---- init accumulators (S is a multiple of 256) ----
    for (int c = lane * 4; c < S; c += 256) {
        acc[c + 0] = ACC(0); acc[c + 1] = ACC(0); acc[c + 2] = ACC(0); acc[c + 3] = ACC(0);
        if (FT) { ft[c + 0] = 0xffffffffu; ft[c + 1] = 0xffffffffu; ft[c + 2] = 0xffffffffu; ft[c + 3] = 0xffffffffu; }
    }

    // ---- accumulate: rows of W for the user's items, ascending item order ----
    const int *tp = a.tile_ptr + static_cast<size_t>(tile) * (a.n_items + 1);
    for (int base = 0; base < n_a; base += 64) {
        const int p = base + lane;
        float x = 0.0f;
        int s = 0, e = 0;
        if (p < n_a) {
            const int item = a.xb_col[a0 + p];
            x = a.xb_val[a0 + p];
            if (item < a.n_items) {   // items newer than W have no row yet
                s = tp[item];
                e = tp[item + 1];
            }
        }
        unsigned long long live = __ballot(e > s);
        while (live) {
            const int q = __builtin_ctzll(live);
            live &= live - 1;
            const int ss = readlane_i(s, q), ee = readlane_i(e, q);
            const ACC xx = static_cast<ACC>(readlane_f(x, q));
            const uint32_t pos = static_cast<uint32_t>(base + q);
            int o = ss + lane;
            // 4 independent loads in flight per lane on long rows
            for (; o + 192 < ee; o += 256) {
                const int c0 = a.w_col[o], c1 = a.w_col[o + 64], c2 = a.w_col[o + 128], c3 = a.w_col[o + 192];
                const float v0 = a.w_val[o], v1 = a.w_val[o + 64], v2 = a.w_val[o + 128], v3 = a.w_val[o + 192];
                lds_add(&acc[c0], xx * static_cast<ACC>(v0));
                lds_add(&acc[c1], xx * static_cast<ACC>(v1));
                lds_add(&acc[c2], xx * static_cast<ACC>(v2));
                lds_add(&acc[c3], xx * static_cast<ACC>(v3));
                if (FT) { atomicMin(&ft[c0], pos); atomicMin(&ft[c1], pos); atomicMin(&ft[c2], pos); atomicMin(&ft[c3], pos); }
            }
            for (; o < ee; o += 64) {
                const int c = a.w_col[o];
                const float v = a.w_val[o];
                lds_add(&acc[c], xx * static_cast<ACC>(v));
                if (FT) atomicMin(&ft[c], pos);
            }
        }
    }

    // ---- invalidate: tile padding, interacted items, non-candidates ----
    for (int c = ncol + lane; c < S; c += 64) acc[c] = ninf;
    if (a.mode == RTREC_TOPK_CANDIDATES) {
        for (int c = lane; c < ncol; c += 64)
            if (a.col_rank[a.col_offset + t0 + c] < 0) acc[c] = ninf;
    } else if (a.filter) {
        const int lo = a.col_offset + t0, hi = lo + ncol;
        for (int p = lane; p < n_a; p += 64) {
            const int item = a.xb_col[a0 + p];
            if (item >= lo && item < hi) acc[item - lo] = ninf;
        }
    }

    // ---- pass 1: per-lane best key ----
    ACC best = ninf;
    for (int c = lane * 4; c < S; c += 256) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const ACC k = sel_key(acc[c + j], zero_valid);
            best = k > best ? k : best;
        }
    }
    // tau = kk-th largest lane best (a lower bound of the kk-th largest key overall)
    ACC tau = ninf;
    {
        ACC cur = best;
        for (int r = 0; r < a.kk; ++r) {
            const ACC m = wave_max(cur);
            tau = m;
            if (m == ninf) break;
            const unsigned long long eq = __ballot(cur == m);
            if (lane == __builtin_ctzll(eq)) cur = ninf;
        }
    }

    // ---- pass 2: collect columns with key >= tau ----
    int cnt = 0;
    for (int c = lane * 4; c < S; c += 256) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const ACC k = sel_key(acc[c + j], zero_valid);
            const bool hit = (k != ninf) && (k >= tau);
            const unsigned long long m = __ballot(hit);
            if (m) {
                const int pos = cnt + lane_prefix(m);
                if (hit && pos < kListCap) clist[pos] = static_cast<uint16_t>(c + j);
                cnt += __builtin_popcountll(m);
            }
        }
    }

    ACC *out_s = reinterpret_cast<ACC *>(a.cand_score) + (static_cast<size_t>(row) * a.n_tiles + tile) * a.kk;
    int *out_i = a.cand_id + (static_cast<size_t>(row) * a.n_tiles + tile) * a.kk;
    uint32_t *out_a = a.cand_aux + (static_cast<size_t>(row) * a.n_tiles + tile) * a.kk;
    const int gbase = a.col_offset + t0;

    auto make_cand = [&](int c) {
        Cand<ACC> x;
        x.score = acc[c];
        x.id = gbase + c;
        x.aux = FT ? ft[c] : (a.mode == RTREC_TOPK_CANDIDATES ? static_cast<uint32_t>(a.col_rank[gbase + c]) : 0u);
        return x;
    };

    int n_out = 0;
    if (cnt <= 64) {
        // one candidate per lane, rank by counting
        Cand<ACC> mine;
        mine.id = -1; mine.score = ninf; mine.aux = 0u;
        if (lane < cnt) mine = make_cand(clist[lane]);
        int rank = 0;
        for (int t = 0; t < cnt; ++t) {
            const Cand<ACC> o = cand_readlane<ACC>(mine, t);
            rank += cand_better(o, mine) ? 1 : 0;
        }
        if (lane < cnt && rank < a.kk) {
            out_s[rank] = mine.score; out_i[rank] = mine.id; out_a[rank] = mine.aux;
        }
        n_out = min(cnt, a.kk);
    } else if (cnt <= kListCap) {
        for (int r = 0; r < a.kk; ++r) {
            Cand<ACC> b; b.id = -1; b.score = ninf; b.aux = 0u;
            int bt = -1;
            for (int t = lane; t < cnt; t += 64) {
                const uint16_t c = clist[t];
                if (c == 0xffffu) continue;
                const Cand<ACC> x = make_cand(c);
                if (cand_better(x, b)) { b = x; bt = t; }
            }
            const Cand<ACC> w = wave_best(b);
            if (w.id < 0) break;
            if (b.id == w.id && bt >= 0) clist[bt] = 0xffffu;
            if (lane == 0) { out_s[r] = w.score; out_i[r] = w.id; out_a[r] = w.aux; }
            n_out = r + 1;
        }
    } else {
        // more exact-threshold candidates than the list holds: successive full scans, each
        // bounded above by the previously emitted candidate
        Cand<ACC> last; last.id = -1; last.score = ninf; last.aux = 0u;
        for (int r = 0; r < a.kk; ++r) {
            Cand<ACC> b; b.id = -1; b.score = ninf; b.aux = 0u;
            for (int c = lane; c < ncol; c += 64) {
                if (sel_key(acc[c], zero_valid) == ninf) continue;
                const Cand<ACC> x = make_cand(c);
                if (last.id >= 0 && !cand_better(last, x)) continue;   // x must be strictly below last
                if (cand_better(x, b)) b = x;
            }
            const Cand<ACC> w = wave_best(b);
            if (w.id < 0) break;
            last = w;
            if (lane == 0) { out_s[r] = w.score; out_i[r] = w.id; out_a[r] = w.aux; }
            n_out = r + 1;
        }
    }
    if (lane == 0) a.cand_cnt[static_cast<size_t>(row) * a.n_tiles + tile] = n_out;
}

// Fast pass: one (row, tile) job per workgroup.  Work items are ordered tile-major and dealt
// to the 8 XCDs in contiguous ranges (blocks b and b+8 share an XCD and its L2), so an XCD
// streams one tile's slice of W at a time out of its own L2.
template <typename ACC>
__global__ __launch_bounds__(64) void score_tiles_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const long long total = static_cast<long long>(a.n_rows) * a.n_tiles;
    const long long per_xcd = (total + 7) / 8;
    const long long w = static_cast<long long>(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (blockIdx.x / 8 >= per_xcd || w >= total) return;
    const int tile = static_cast<int>(w / a.n_rows);
    const int row = static_cast<int>(w % a.n_rows);
    score_tile_job<ACC, false>(a, row, tile, smem);
}

// Exact-tie pass: block b re-scores (flagged row b / n_tiles, tile b % n_tiles); blocks beyond
// the flagged count exit at once (the count only exists on the device).
template <typename ACC>
__global__ __launch_bounds__(64) void score_tiles_ft_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_flag = *a.row_list_len;
    const int f = blockIdx.x / a.n_tiles;
    if (f >= n_flag) return;
    score_tile_job<ACC, true>(a, a.row_list[f], blockIdx.x % a.n_tiles, smem);
}

struct MergeArgs {
    int n_rows;
    int n_lists;      // lists per row
    int kk;           // entries per list
    int top_k;
    long long list_stride;  // element stride between consecutive lists of one row
    long long row_stride;   // element stride between rows
    long long cnt_list_stride, cnt_row_stride;
    const void *in_score;   // ACC
    const int *in_id;
    const uint32_t *in_aux;
    const int *in_cnt;
    int *out_id;
    float *out_score;
    double *out_score64;    // may be null
    uint32_t *out_aux;      // may be null
    int *out_cnt;
    int detect_ties;        // flag rows whose leading top_k+1 scores contain an exact tie
    int *flag_list;         // compacted flagged rows
    int *flag_len;
    const int *row_list;    // if non-null: merge only these rows
    const int *row_list_len;
};

// One wave per row: the row's n_lists * kk candidates -> top_k by (score, aux, id).
template <typename ACC>
__global__ __launch_bounds__(64) void merge_topk_kernel(MergeArgs m) {
    const int lane = lane_id();
    int row = blockIdx.x;
    if (m.row_list) {
        if (row >= *m.row_list_len) return;
        row = m.row_list[row];
    }
    const ACC *sc = reinterpret_cast<const ACC *>(m.in_score);
    const ACC ninf = NegInf<ACC>::value();
    constexpr int kMaxPerLane = 16;   // n_lists * kk <= 1024
    Cand<ACC> mine[kMaxPerLane];
    const int total = m.n_lists * m.kk;
#pragma unroll
    for (int j = 0; j < kMaxPerLane; ++j) {
        mine[j].id = -1; mine[j].score = ninf; mine[j].aux = 0u;
        const int t = lane + 64 * j;
        if (t < total) {
            const int l = t / m.kk, r = t % m.kk;
            const int c = m.in_cnt[l * m.cnt_list_stride + row * m.cnt_row_stride];
            if (r < c) {
                const long long off = l * m.list_stride + row * m.row_stride + r;
                mine[j].score = sc[off];
                mine[j].id = m.in_id[off];
                mine[j].aux = m.in_aux ? m.in_aux[off] : 0u;
            }
        }
    }
    const int want = m.detect_ties ? m.top_k + 1 : m.top_k;
    int n_out = 0;
    bool tie = false;
    ACC prev = ninf;
    for (int r = 0; r < want; ++r) {
        Cand<ACC> b; b.id = -1; b.score = ninf; b.aux = 0u;
        int bj = -1;
#pragma unroll
        for (int j = 0; j < kMaxPerLane; ++j)
            if (cand_better(mine[j], b)) { b = mine[j]; bj = j; }
        const Cand<ACC> w = wave_best(b);
        if (w.id < 0) break;
        if (b.id == w.id && b.aux == w.aux && bj >= 0) {
#pragma unroll
            for (int j = 0; j < kMaxPerLane; ++j)
                if (j == bj) mine[j].id = -1;
        }
        if (r > 0 && w.score == prev) tie = true;
        prev = w.score;
        if (r < m.top_k) {
            if (lane == 0) {
                const long long o = static_cast<long long>(row) * m.top_k + r;
                m.out_id[o] = w.id;
                m.out_score[o] = static_cast<float>(w.score);
                if (m.out_score64) m.out_score64[o] = static_cast<double>(w.score);
                if (m.out_aux) m.out_aux[o] = w.aux;
            }
            n_out = r + 1;
        }
    }
    if (lane == 0) {
        for (int r = n_out; r < m.top_k; ++r) {
            const long long o = static_cast<long long>(row) * m.top_k + r;
            m.out_id[o] = -1;
            m.out_score[o] = -__builtin_huge_valf();
            if (m.out_score64) m.out_score64[o] = -__builtin_huge_val();
            if (m.out_aux) m.out_aux[o] = 0u;
        }
        m.out_cnt[row] = n_out;
        if (m.detect_ties && tie) {
            const int slot = atomicAdd(m.flag_len, 1);
            m.flag_list[slot] = row;
        }
    }
}

// similar_items (slim_elastic.py:838-857): one wave per query column of W (CSC).
__global__ __launch_bounds__(64) void similar_topk_kernel(int n_queries, const int *queries,
                                                          const int *wc_ptr, const int *wc_row, const float *wc_val,
                                                          int top_k, int *out_id, float *out_score, int *out_cnt) {
    const int lane = lane_id();
    const int qi = blockIdx.x;
    if (qi >= n_queries) return;
    const int item = queries[qi];
    const int s = wc_ptr[item], e = wc_ptr[item + 1];
    const float ninf = -__builtin_huge_valf();
    Cand<float> last; last.id = -1; last.score = ninf; last.aux = 0u;
    int n_out = 0;
    // aux = (0xffffffff - position) so that, among equal scores, the earlier stored entry
    // (lower row id) wins: the order a stable argsort of the negated scores produces.
    for (int r = 0; r < top_k; ++r) {
        Cand<float> b; b.id = -1; b.score = ninf; b.aux = 0u;
        for (int o = s + lane; o < e; o += 64) {
            const int i = wc_row[o];
            if (i == item) continue;
            Cand<float> x; x.score = wc_val[o]; x.id = i; x.aux = 0xffffffffu - static_cast<uint32_t>(o - s);
            if (last.id >= 0 && !cand_better(last, x)) continue;
            if (cand_better(x, b)) b = x;
        }
        const Cand<float> w = wave_best(b);
        if (w.id < 0) break;
        last = w;
        if (lane == 0) { out_id[static_cast<long long>(qi) * top_k + r] = w.id; out_score[static_cast<long long>(qi) * top_k + r] = w.score; }
        n_out = r + 1;
    }
    if (lane == 0) {
        for (int r = n_out; r < top_k; ++r) { out_id[static_cast<long long>(qi) * top_k + r] = -1; out_score[static_cast<long long>(qi) * top_k + r] = ninf; }
        out_cnt[qi] = n_out;
    }
}

}  // namespace rtrec

// ------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------
using namespace rtrec;

#include <cstdio>
#include <cstdlib>

namespace {
// RTREC_AMD_DEBUG=1: synchronise after every launch and report the stage (diagnostics only).
inline void debug_stage(hipStream_t st, const char *what) {
    static const bool on = std::getenv("RTREC_AMD_DEBUG") != nullptr;
    if (!on) return;
    const hipError_t e = hipStreamSynchronize(st);
    std::fprintf(stderr, "[rtrec_amd] %s: %s\n", what, hipGetErrorString(e));
    std::fflush(stderr);
}
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct ScoreWs {
    size_t cand_score, cand_id, cand_aux, cand_cnt, flag_list, flag_len, queue, tmp_id, tmp_score, tmp_aux, tmp_cnt, total;
};
ScoreWs score_ws_layout(int n_rows, int n_tiles, int top_k) {
    ScoreWs w;
    const size_t kk = static_cast<size_t>(top_k) + 1;
    const size_t n = static_cast<size_t>(n_rows) * n_tiles * kk;
    size_t o = 0;
    w.cand_score = o; o = align_up(o + n * sizeof(double), 256);
    w.cand_id = o;    o = align_up(o + n * sizeof(int), 256);
    w.cand_aux = o;   o = align_up(o + n * sizeof(uint32_t), 256);
    w.cand_cnt = o;   o = align_up(o + static_cast<size_t>(n_rows) * n_tiles * sizeof(int), 256);
    w.flag_list = o;  o = align_up(o + static_cast<size_t>(n_rows) * sizeof(int), 256);
    w.flag_len = o;   o = align_up(o + 256, 256);
    w.queue = o;      o = align_up(o + 256, 256);
    w.total = o;
    return w;
}

template <typename ACC>
int score_impl(const ScoreArgs &base, int top_k, int acc_bytes, int32_t *d_out_ids, float *d_out_scores,
               double *d_out_scores64, uint32_t *d_out_aux, int32_t *d_out_count,
               unsigned char *ws, const ScoreWs &L, hipStream_t st) {
    ScoreArgs a = base;
    const bool sparse = (a.mode == RTREC_TOPK_SPARSE);
    a.kk = sparse ? top_k + 1 : top_k;
    a.cand_score = ws + L.cand_score;
    a.cand_id = reinterpret_cast<int *>(ws + L.cand_id);
    a.cand_aux = reinterpret_cast<uint32_t *>(ws + L.cand_aux);
    a.cand_cnt = reinterpret_cast<int *>(ws + L.cand_cnt);
    int *flag_list = reinterpret_cast<int *>(ws + L.flag_list);
    int *flag_len = reinterpret_cast<int *>(ws + L.flag_len);
    int *queue = reinterpret_cast<int *>(ws + L.queue);
    a.row_list = flag_list;
    a.row_list_len = flag_len;
    a.queue = queue;
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (hipMemsetAsync(flag_len, 0, 4, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    if (hipMemsetAsync(queue, 0, 4, st) != hipSuccess) return RTREC_ERR_LAUNCH;

    const size_t lds_fast = static_cast<size_t>(a.tile_cols) * acc_bytes + kListCap * 2;
    const size_t lds_ft = static_cast<size_t>(a.tile_cols) * (acc_bytes + 4) + kListCap * 2;
    const long long total = static_cast<long long>(a.n_rows) * a.n_tiles;
    const long long per_xcd = (total + 7) / 8;
    const unsigned grid = static_cast<unsigned>(per_xcd * 8);
    debug_stage(st, "score: begin");
    hipLaunchKernelGGL(HIP_KERNEL_NAME(score_tiles_kernel<ACC>), dim3(grid), dim3(64), lds_fast, st, a);
    debug_stage(st, "score_tiles_kernel");

    MergeArgs m{};
    m.n_rows = a.n_rows; m.n_lists = a.n_tiles; m.kk = a.kk; m.top_k = top_k;
    m.list_stride = a.kk; m.row_stride = static_cast<long long>(a.n_tiles) * a.kk;
    m.cnt_list_stride = 1; m.cnt_row_stride = a.n_tiles;
    m.in_score = a.cand_score; m.in_id = a.cand_id; m.in_aux = a.cand_aux; m.in_cnt = a.cand_cnt;
    m.out_id = d_out_ids; m.out_score = d_out_scores; m.out_score64 = d_out_scores64; m.out_aux = d_out_aux;
    m.out_cnt = d_out_count;
    m.detect_ties = sparse ? 1 : 0;
    m.flag_list = flag_list; m.flag_len = flag_len;
    m.row_list = nullptr; m.row_list_len = nullptr;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(merge_topk_kernel<ACC>), dim3(a.n_rows), dim3(64), 0, st, m);
    debug_stage(st, "merge_topk_kernel");

    if (sparse) {
        // exact tie order for the flagged rows only
        ScoreArgs f = a;
        f.kk = top_k;
        const unsigned ft_grid = static_cast<unsigned>(total);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(score_tiles_ft_kernel<ACC>), dim3(ft_grid), dim3(64), lds_ft, st, f);
        debug_stage(st, "score_tiles_ft_kernel");
        MergeArgs mf = m;
        mf.kk = top_k;
        mf.list_stride = top_k; mf.row_stride = static_cast<long long>(a.n_tiles) * top_k;
        mf.detect_ties = 0;
        mf.row_list = flag_list; mf.row_list_len = flag_len;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(merge_topk_kernel<ACC>), dim3(a.n_rows), dim3(64), 0, st, mf);
        debug_stage(st, "merge_topk_kernel (exact ties)");
    }
    return rtrec::launch_status();
}
}  // namespace

extern "C" size_t rtrec_slim_score_workspace_bytes(int32_t n_rows, int32_t n_tiles, int32_t top_k) {
    if (n_rows < 0 || n_tiles <= 0 || top_k <= 0) return 0;
    return score_ws_layout(n_rows, n_tiles, top_k).total;
}

extern "C" int rtrec_slim_score_topk(int32_t n_rows, const int32_t *d_row_ids,
                                     const int32_t *d_xb_ptr, const int32_t *d_xb_col, const float *d_xb_val,
                                     int32_t n_items, int32_t n_cols, int32_t col_offset,
                                     int32_t tile_cols, int32_t n_tiles,
                                     const int32_t *d_tile_ptr, const uint16_t *d_w_col, const float *d_w_val,
                                     const int32_t *d_col_rank,
                                     int32_t top_k, int32_t filter_interacted, int32_t mode, int32_t acc_f64,
                                     int32_t *d_out_ids, float *d_out_scores, double *d_out_scores64,
                                     uint32_t *d_out_aux, int32_t *d_out_count,
                                     void *d_workspace, size_t workspace_bytes, void *stream) {
    if (n_rows < 0 || n_items <= 0 || n_cols <= 0 || top_k <= 0) return RTREC_ERR_INVALID_ARG;
    if (n_rows == 0) return RTREC_OK;
    if (!d_xb_ptr || !d_tile_ptr || !d_out_ids || !d_out_scores || !d_out_count || !d_workspace) return RTREC_ERR_INVALID_ARG;
    if (mode < 0 || mode > 2) return RTREC_ERR_INVALID_ARG;
    if (mode == RTREC_TOPK_CANDIDATES && !d_col_rank) return RTREC_ERR_INVALID_ARG;
    if (tile_cols < 256 || tile_cols > 65536 || (tile_cols % 256) != 0) return RTREC_ERR_UNSUPPORTED;
    if (n_tiles != (n_cols + tile_cols - 1) / tile_cols) return RTREC_ERR_INVALID_ARG;
    const int acc_bytes = acc_f64 ? 8 : 4;
    // the first-touch instantiation needs tile_cols * (acc + 4) + list bytes of LDS (160 KiB / CU)
    if (static_cast<size_t>(tile_cols) * (acc_bytes + 4) + kListCap * 2 > 160u * 1024u) return RTREC_ERR_UNSUPPORTED;
    if (top_k + 1 > 64 || static_cast<long long>(n_tiles) * (top_k + 1) > 1024) return RTREC_ERR_UNSUPPORTED;
    const ScoreWs L = score_ws_layout(n_rows, n_tiles, top_k);
    if (workspace_bytes < L.total) return RTREC_ERR_WORKSPACE;

    ScoreArgs a{};
    a.n_rows = n_rows; a.row_ids = d_row_ids; a.xb_ptr = d_xb_ptr; a.xb_col = d_xb_col; a.xb_val = d_xb_val;
    a.n_items = n_items; a.n_cols = n_cols; a.col_offset = col_offset;
    a.tile_cols = tile_cols; a.n_tiles = n_tiles;
    a.tile_ptr = d_tile_ptr; a.w_col = d_w_col; a.w_val = d_w_val; a.col_rank = d_col_rank;
    a.filter = filter_interacted; a.mode = mode;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char *ws = static_cast<unsigned char *>(d_workspace);
    if (acc_f64)
        return score_impl<double>(a, top_k, 8, d_out_ids, d_out_scores, d_out_scores64, d_out_aux, d_out_count, ws, L, st);
    return score_impl<float>(a, top_k, 4, d_out_ids, d_out_scores, d_out_scores64, d_out_aux, d_out_count, ws, L, st);
}

extern "C" int rtrec_slim_merge_topk(int32_t n_rows, int32_t n_lists, int32_t top_k,
                                     const int32_t *d_in_ids, const float *d_in_scores, const double *d_in_scores64,
                                     const uint32_t *d_in_aux, const int32_t *d_in_count,
                                     int32_t *d_out_ids, float *d_out_scores, int32_t *d_out_count,
                                     void *stream) {
    if (n_rows < 0 || n_lists <= 0 || top_k <= 0) return RTREC_ERR_INVALID_ARG;
    if (n_rows == 0) return RTREC_OK;
    if (!d_in_ids || !d_in_scores || !d_in_count || !d_out_ids || !d_out_scores || !d_out_count) return RTREC_ERR_INVALID_ARG;
    if (static_cast<long long>(n_lists) * top_k > 1024) return RTREC_ERR_UNSUPPORTED;
    MergeArgs m{};
    m.n_rows = n_rows; m.n_lists = n_lists; m.kk = top_k; m.top_k = top_k;
    m.list_stride = static_cast<long long>(n_rows) * top_k; m.row_stride = top_k;
    m.cnt_list_stride = n_rows; m.cnt_row_stride = 1;
    m.in_id = d_in_ids; m.in_aux = d_in_aux; m.in_cnt = d_in_count;
    m.out_id = d_out_ids; m.out_score = d_out_scores; m.out_score64 = nullptr; m.out_aux = nullptr; m.out_cnt = d_out_count;
    m.detect_ties = 0; m.flag_list = nullptr; m.flag_len = nullptr; m.row_list = nullptr; m.row_list_len = nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (d_in_scores64) {
        m.in_score = d_in_scores64;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(merge_topk_kernel<double>), dim3(n_rows), dim3(64), 0, st, m);
    } else {
        m.in_score = d_in_scores;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(merge_topk_kernel<float>), dim3(n_rows), dim3(64), 0, st, m);
    }
    return rtrec::launch_status();
}

extern "C" int rtrec_slim_similar_topk(int32_t n_queries, const int32_t *d_queries,
                                       const int32_t *d_wc_ptr, const int32_t *d_wc_row, const float *d_wc_val,
                                       int32_t top_k,
                                       int32_t *d_out_ids, float *d_out_scores, int32_t *d_out_count,
                                       void *stream) {
    if (n_queries < 0 || top_k <= 0) return RTREC_ERR_INVALID_ARG;
    if (n_queries == 0) return RTREC_OK;
    if (!d_queries || !d_wc_ptr || !d_out_ids || !d_out_scores || !d_out_count) return RTREC_ERR_INVALID_ARG;
    (void)hipGetLastError();
    hipLaunchKernelGGL(similar_topk_kernel, dim3(n_queries), dim3(64), 0, static_cast<hipStream_t>(stream),
                       n_queries, d_queries, d_wc_ptr, d_wc_row, d_wc_val, top_k, d_out_ids, d_out_scores, d_out_count);
    return rtrec::launch_status();
}
