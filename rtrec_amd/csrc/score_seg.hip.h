// rtrec_amd/csrc/score_seg.hip.h -- SPARSE-mode scoring for a GENERAL W: "segment" layout (included by score.hip).
//
// Replaces (reference): X[users] @ W (slim_elastic.py:707-708, scipy csr_matmat) + _sparse_topk_indicies
// (slim_elastic.py:782-818) when W has MANY non-empty rows -- what a fit on data with item-item structure
// yields (ML-20M shape in 80 item clusters: 3.4k rows, 357k weights, a user's row gathers 12k of them and its
// score vector has ~5k non-zeros).  The feature-row kernel above needs <= 128 rows of W; the tiled-CSR kernel
// pays per (user, tile, item) and selects from a touched list of thousands of columns.
//
// Layout (rtrec_amd/seg_layout.py, rtrec_score_opts.seg): the columns that hold a weight, ordered by cluster
// (label propagation over W's graph), in tiles of T columns; per row of W its entries sorted by layout column,
// seg_ptr[row][t] .. [t + 1] = its SEGMENT in tile t; bound[row][t] = max |w| of that segment (bfloat16, rounded up).
//
// One wave scores one user at a time (persistent waves, users claimed longest first):
//   1. setup: the user's items -> (row of W, layout column) by one 8-byte gather each; items that have a row are
//      compacted into LDS; B[t] = sum_i |x_ui| bound[i][t] for all <= 128 tiles (lane l holds tiles 2l, 2l + 1; one
//      coalesced 256-byte load per item);
//   2. tiles in DESCENDING B order: B[t] bounds every score of the tile, so the loop ends as soon as the best
//      remaining B cannot beat the user's current (k+1)-th best score (ML-20M shape, clustered: 5 of 86 tiles are
//      opened per user).  An opened tile: the user's own columns are set to -inf (interacted filter; -inf absorbs),
//      the user's segments of the tile are added IN ASCENDING ITEM ORDER into T floats of LDS -- plain
//      read-modify-write, one rounded product and one rounded add per entry; a wave's LDS operations execute in
//      program order and a segment never repeats a column, so every column receives its addends exactly in scipy's
//      csr_matmat order: bit-identical sums -- then the tile is read back 4 columns per lane, zeroed, and what beats
//      the (k+1)-th best goes into the user's sorted list (one register pair across the lanes, insertion by DPP shift);
//   3. the list is the answer.  A list with an exact score tie among its leading k + 1 entries is handed to the
//      exact-tie pass (score_sparse_kernel<ACC, true>: first-touch order, as for the other fast paths).
// Skipping a tile never changes an answer: nothing in it can enter the list (margin for the float32 rounding of
// both sides below), and zero sums are no candidates in SPARSE mode anyway.

struct SegArgs {
    int n_rows; const int *row_ids; const int *order; int n_x_rows;
    const int *xb_ptr; const int *xb_col; const float *xb_val;
    int n_items;
    const int2 *info;          // [n_items] {row of W or -1, layout column or -1}
    const int *seg_ptr;        // [R][n_tiles + 1]
    const uint16_t *w_col;     // column inside the tile
    const float *w_val;
    const uint32_t *bound;     // [R][64]: lane l -> bfloat16 bounds of tiles 2l (low half) and 2l + 1 (high half)
    const int *col_ids;        // layout column -> item id
    int n_cols, T, n_tiles, R;
    int kk, top_k, filter;
    int *out_id; float *out_score; uint32_t *out_aux; int *out_cnt;
    int *flag_list; int *flag_len;
    int *queue;
};

constexpr int kSgWaves = 4;          // waves per workgroup; they share nothing (no barrier in the kernel)
constexpr int kSgCapAll = 512;       // layout columns of the user's items kept in LDS (interacted filter)
constexpr int kSgCapRow = 256;       // (row, rating) pairs of the user's items that have a row in W kept in LDS
constexpr int kSgQueueChunk = 4;     // users per queue claim
constexpr int kSgMaxKk = 64;         // top_k + 1 list entries: one per lane

__host__ __device__ constexpr size_t sg_wave_lds(int T) {
    return static_cast<size_t>(T) * 4 + kSgCapAll * 4 + kSgCapRow * 8;
}

__device__ __forceinline__ float sg_wave_max(float v) {       // uniform maximum of the 64 lane values
    const float ninf = -__builtin_huge_valf();
    v = fmaxf(v, fr_dpp_f<0x111>(v, ninf));
    v = fmaxf(v, fr_dpp_f<0x112>(v, ninf));
    v = fmaxf(v, fr_dpp_f<0x114>(v, ninf));
    v = fmaxf(v, fr_dpp_f<0x118>(v, ninf));
    v = fmaxf(v, fr_dpp_f<0x142>(v, ninf));      // row_bcast:15
    v = fmaxf(v, fr_dpp_f<0x143>(v, ninf));      // row_bcast:31
    return readlane_f(v, 63);
}

// kk-th largest of the 64 lane values (-inf = none; -inf when fewer than kk lanes hold one): kk rounds of "take the
// maximum out".  Uniform.  Runs while a user's list is still filling, i.e. in the first scan step of its first tile.
__device__ __forceinline__ float sg_kth_lane_best(float v, int kk) {
    const float ninf = -__builtin_huge_valf();
    const int lane = lane_id();
    float tau = ninf;
    for (int r = 0; r < kk; ++r) {
        tau = sg_wave_max(v);
        if (tau == ninf) break;
        const unsigned long long at = __ballot(v == tau);
        v = lane == static_cast<int>(__builtin_ctzll(at)) ? ninf : v;
    }
    return tau;
}

template <int GROUP>
__global__ __launch_bounds__(kSgWaves * 64) void score_seg_kernel(SegArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = static_cast<int>(threadIdx.x) >> 6;
    const int T = a.T;
    unsigned char *wbase = smem + static_cast<size_t>(wave) * sg_wave_lds(T);
    float *acc = reinterpret_cast<float *>(wbase);
    vf4 *acc4 = reinterpret_cast<vf4 *>(wbase);
    int *lcl = reinterpret_cast<int *>(wbase + static_cast<size_t>(T) * 4);
    int *rr = lcl + kSgCapAll;
    float *rx = reinterpret_cast<float *>(rr + kSgCapRow);
    const float ninf = -__builtin_huge_valf();
    const vf4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int kk = a.kk;
    const unsigned long long kkmask = kk >= 64 ? ~0ull : ((1ull << kk) - 1ull);
    for (int c = lane; c < (T >> 2); c += 64) acc4[c] = zero4;

    int w_next = 0, w_end = 0;
    for (;;) {
        if (w_next >= w_end) {
            int w0 = 0;
            if (lane == 0) w0 = atomicAdd(a.queue, kSgQueueChunk);
            w_next = readfirst_i(w0);
            if (w_next >= a.n_rows) break;
            w_end = min(w_next + kSgQueueChunk, a.n_rows);
        }
        const int p = w_next++;
        const int row = a.order ? a.order[p] : p;
        const int xrow = a.row_ids ? a.row_ids[row] : row;
        const bool xok = xrow >= 0 && xrow < a.n_x_rows;       // anything else scores as an empty row
        const int a0 = readfirst_i(xok ? a.xb_ptr[xrow] : 0);
        const int n_a = readfirst_i(xok ? a.xb_ptr[xrow + 1] - a0 : 0);

        // ---- 1. setup: rows of W the user rates, layout columns of its items, per-tile score bounds
        float B0 = 0.0f, B1 = 0.0f;
        int n_r = 0;
        const bool fits = n_a <= kSgCapAll;
        for (int base = 0; base < n_a; base += 64) {
            const int idx = base + lane;
            int r = -1, lc = -1;
            float x = 0.0f;
            if (idx < n_a) {
                const int item = a.xb_col[a0 + idx];
                x = a.xb_val[a0 + idx];
                if (item < a.n_items) { const int2 f = a.info[item]; r = f.x; lc = f.y; }      // items newer than W have neither
            }
            if (fits && idx < n_a) lcl[idx] = lc;
            unsigned long long m = __ballot(r >= 0);
            if (r >= 0) {
                const int pos = n_r + lane_prefix(m);
                if (pos < kSgCapRow) { rr[pos] = r; rx[pos] = x; }
            }
            n_r += static_cast<int>(__builtin_popcountll(m));
            while (m) {
                int rq[4];
                float aq[4];
                uint32_t bq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    rq[j] = -1; aq[j] = 0.0f;
                    if (m) {
                        const int q = __builtin_ctzll(m);
                        m &= m - 1;
                        rq[j] = readlane_i(r, q);
                        aq[j] = fabsf(readlane_f(x, q));
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) bq[j] = rq[j] >= 0 ? a.bound[static_cast<size_t>(rq[j]) * 64 + lane] : 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    B0 = __builtin_fmaf(aq[j], __uint_as_float(bq[j] << 16), B0);
                    B1 = __builtin_fmaf(aq[j], __uint_as_float(bq[j] & 0xffff0000u), B1);
                }
            }
        }
        const bool in_lds = fits && n_r <= kSgCapRow;        // else: the user's items are re-read from global per tile
        const int n_ch = in_lds ? (n_r + 63) >> 6 : (n_a + 63) >> 6;
        // a computed score is a float32 sum of rounded products, the bound a float32 sum of |x| * (rounded-up max):
        // score <= B * (1 + (2 n + 2) 2^-24) with n <= n_r addends
        const float slack = 1.0f + 1e-5f + static_cast<float>(n_r) * 1.3e-7f;

        // ---- 2. tiles by descending bound
        float ls = ninf;        // lane j: score of the j-th best column so far
        int lcid = -1;          //         its layout column
        int n_list = 0;
        float theta = ninf;     // score of the (k+1)-th best once the list is full
        for (;;) {
            const float bm = fmaxf(B0, B1);
            const float bmax = sg_wave_max(bm);
            if (!(bmax > 0.0f)) break;
            if (n_list >= kk && bmax * slack < theta) break;
            const int ql = static_cast<int>(__builtin_ctzll(__ballot(bm == bmax)));
            const int which = readlane_f(B0, ql) == bmax ? 0 : 1;
            const int t = 2 * ql + which;
            if (lane == ql) { if (which == 0) B0 = 0.0f; else B1 = 0.0f; }
            const int t0 = t * T;
            const int ncol = min(T, a.n_cols - t0);

            if (a.filter) {         // the user's own columns leave the race: -inf + p = -inf
                for (int base = 0; base < n_a; base += 64) {
                    const int idx = base + lane;
                    int lc = -1;
                    if (idx < n_a) {
                        if (fits) lc = lcl[idx];
                        else { const int item = a.xb_col[a0 + idx]; if (item < a.n_items) lc = a.info[item].y; }
                    }
                    lc -= t0;
                    if (lc >= 0 && lc < ncol) acc[lc] = ninf;
                }
            }

            for (int ch = 0; ch < n_ch; ++ch) {
                const int idx = (ch << 6) + lane;
                int r = -1;
                float x = 0.0f;
                if (in_lds) {
                    if (idx < n_r) { r = rr[idx]; x = rx[idx]; }
                } else if (idx < n_a) {
                    const int item = a.xb_col[a0 + idx];
                    if (item < a.n_items) { r = a.info[item].x; x = a.xb_val[a0 + idx]; }
                }
                int s = 0, e = 0;
                if (r >= 0) {
                    const int *pp = a.seg_ptr + static_cast<size_t>(r) * (a.n_tiles + 1) + t;
                    s = pp[0]; e = pp[1];
                }
                unsigned long long live = __ballot(e > s);
                while (live) {
                    // the next GROUP segments (ascending item order): their first 64 entries are requested together,
                    // the accumulator updates then go out segment by segment
                    int ss[GROUP], ee[GROUP], cc[GROUP];
                    float xx[GROUP], vv[GROUP];
#pragma unroll
                    for (int j = 0; j < GROUP; ++j) {
                        ss[j] = 0; ee[j] = 0; xx[j] = 0.0f;
                        if (live) {
                            const int q = __builtin_ctzll(live);
                            live &= live - 1;
                            ss[j] = readlane_i(s, q);
                            ee[j] = readlane_i(e, q);
                            xx[j] = readlane_f(x, q);
                        }
                        cc[j] = -1; vv[j] = 0.0f;
                        if (ss[j] + lane < ee[j]) { cc[j] = a.w_col[ss[j] + lane]; vv[j] = a.w_val[ss[j] + lane]; }
                    }
#pragma unroll
                    for (int j = 0; j < GROUP; ++j) {
                        if (ee[j] == ss[j]) break;
                        if (cc[j] >= 0) acc[cc[j]] = acc[cc[j]] + xx[j] * vv[j];
                        for (int ob = ss[j] + 64; ob < ee[j]; ob += 64) {       // rest of a long segment
                            const int o = ob + lane;
                            if (o < ee[j]) { const int c = a.w_col[o]; acc[c] = acc[c] + xx[j] * a.w_val[o]; }
                        }
                    }
                }
            }

            // read the tile back (4 columns per lane and step), zero it, keep what can enter the list
            for (int c4 = lane; c4 < (T >> 2); c4 += 64) {
                const vf4 v = acc4[c4];
                acc4[c4] = zero4;
                bool h[4];
                if (n_list >= kk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) h[j] = v[j] != 0.0f && v[j] > theta;
                } else {
                    // list still filling: at least kk of this step's values are >= the kk-th largest lane maximum,
                    // so nothing below it can end up among the best kk
                    float lb = ninf;
#pragma unroll
                    for (int j = 0; j < 4; ++j) lb = (v[j] != 0.0f && v[j] > lb) ? v[j] : lb;
                    const float cut = sg_kth_lane_best(lb, kk);
#pragma unroll
                    for (int j = 0; j < 4; ++j) h[j] = v[j] != 0.0f && v[j] > ninf && v[j] >= cut;
                }
                if (!__ballot(h[0] || h[1] || h[2] || h[3])) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned long long mj = __ballot(h[j]);
                    while (mj) {
                        const int q = __builtin_ctzll(mj);
                        mj &= mj - 1;
                        const float sc = readlane_f(v[j], q);
                        if (n_list >= kk && !(sc > theta)) continue;
                        const int col = t0 + 4 * (c4 - lane + q) + j;
                        // entries that stay ahead: a higher score, or the same score and a lower layout column
                        const unsigned long long ahead = __ballot(ls > sc || (ls == sc && lcid < col)) & kkmask;
                        const int pos = static_cast<int>(__builtin_popcountll(ahead));
                        if (pos >= kk) continue;
                        const float us = fr_shift_up(ls, ninf);
                        const int uc = fr_shift_up(lcid, -1);
                        if (lane > pos) { ls = us; lcid = uc; }
                        else if (lane == pos) { ls = sc; lcid = col; }
                        n_list = min(n_list + 1, kk);
                        theta = n_list >= kk ? readlane_f(ls, kk - 1) : ninf;
                    }
                }
            }
        }

        // ---- 3. emit
        const int n_fin = min(n_list, a.top_k);
        if (lane < a.top_k) {
            const long long o = static_cast<long long>(row) * a.top_k + lane;
            const bool ok = lane < n_fin;
            a.out_id[o] = ok ? a.col_ids[lcid] : -1;
            a.out_score[o] = ok ? ls : ninf;
            if (a.out_aux) a.out_aux[o] = 0u;
        }
        const float below = fr_shift_down(ls, ninf);
        const unsigned long long tie = __ballot(lane + 1 < n_list && ls == below);
        if (lane == 0) {
            a.out_cnt[row] = n_fin;
            if (tie) a.flag_list[atomicAdd(a.flag_len, 1)] = row;
        }
    }
}
