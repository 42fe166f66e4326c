// rtrec_amd/csrc/score.hip -- fused user-row x W accumulation + interacted filter + top-k.
//
// Replaces (reference): SLIMElastic.recommend / recommend_batch (slim_elastic.py:628-741), the
// scipy csr_matmat product behind safe_sparse_dot (slim_elastic.py:585,608,626,708,726) and the
// two top-k helpers (_sparse_topk_indicies :782-818, _dense_topk_indicies :744-779).
//
// Layout: W is cut into column tiles of `tile_cols` columns; each tile is a CSR over all item
// rows with tile-local uint16 column ids (6 bytes per stored weight; well-filled row segments are
// zero-padded dense blocks).  One wavefront owns one (user row, tile) job: it streams the W rows of
// the user's items IN ASCENDING ITEM ORDER and adds x_ui * W[i, c] into an LDS accumulator with a
// plain read-modify-write (no atomics: one wave owns the tile).  LDS operations of one wave execute
// in program order and a W row never repeats a column, so every accumulator receives its addends
// in exactly scipy's csr_matmat order -> bit-identical scores.  The tile is then reduced to its
// top-(k+1) in registers/LDS and a second kernel merges the tiles of a row.
//
// Tie order of the SPARSE mode (Python's stable sorted() over scipy's reverse-first-touch
// product order) needs the first-touch rank of a column, which costs a second LDS array.  The
// fast pass therefore runs without it, the merge flags rows whose leading k+1 scores contain an
// exact tie, and only those rows are re-scored by the FT (first-touch tracking) instantiation.
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

#include <type_traits>
#include <utility>

namespace rtrec {

struct ScoreArgs {
    int n_rows;
    const int *row_ids;   // optional: job r scores CSR row row_ids[r]
    int n_x_rows;         // rows of the CSR matrix: a row id outside [0, n_x_rows) scores as an empty row
    const int *xb_ptr;
    const int *xb_col;
    const float *xb_val;
    int n_items;          // rows of W
    int n_cols;           // columns of this layout (shard width, or number of active columns)
    int col_offset;       // global id of local column 0 when col_ids == nullptr
    const int *col_ids;   // optional: local column -> global item id (ascending)
    const int *col_map;   // optional: global item id -> local column or -1
    int tile_cols;
    int n_tiles;
    const int *tile_ptr;
    const uint16_t *w_col;
    const float *w_val;
    const int *dense_idx;  // optional: [n_tiles * n_items] dense block of (tile, row) or -1
    const float *dense_val; // optional: [n_dense * tile_cols]
    const int4 *row_hdr;    // optional: [n_tiles * n_items] {ptr begin, ptr end, dense block, tile-local column of the item}
    const int *col_rank;
    int kk;               // entries kept per (row, tile): top_k, or top_k + 1 when ties are detected
    int top_k;
    int filter;
    int mode;
    // per (row, tile) candidate lists, used when n_tiles > 1
    void *cand_score;     // ACC[n_rows * n_tiles * kk]
    int *cand_id;
    uint32_t *cand_aux;
    int *cand_cnt;        // [n_rows * n_tiles]
    // final outputs, written directly when n_tiles == 1
    int direct;
    int *out_id;
    float *out_score;
    double *out_score64;
    uint32_t *out_aux;
    int *out_cnt;
    int detect_ties;
    int *flag_list;
    int *flag_len;
    // exact-tie pass: rows to re-score
    const int *row_list;
    const int *row_list_len;
    int *queue;
    int *rescored;        // exact-tie pass, optional: receives the number of rows it re-scores
    int ablate;           // diagnostics: bit0 skip accumulate, bit1 skip select, bit2 skip reset, bit3 skip filter
};

// Diagnostic build (-DSCORE_PROFILE, tools/ab_build.sh): per-phase shader-clock totals of the sparse kernel,
// summed over all waves in g_score_prof (read with rtrec_amd_score_profile; not part of the release ABI).
#ifdef SCORE_PROFILE
enum { PF_JOBS, PF_ROWPTR, PF_HDR, PF_GROUP, PF_DENSE, PF_SPARSE, PF_SELECT, PF_EMIT, PF_RESET, PF_QUEUE,
       PF_N_DENSE, PF_N_SPARSE_ROWS, PF_N_SPARSE_CHUNKS, PF_N_OVERFLOW, PF_TOTAL, PF_COUNT };
__device__ unsigned long long g_score_prof[16];
#define PF_DECL unsigned long long pf_[16] = {0}; unsigned long long pf_t_ = __builtin_amdgcn_s_memtime();
#define PF_MARK(slot) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pf_[slot] += n_ - pf_t_; pf_t_ = n_; }
#define PF_ADD(slot, v) { pf_[slot] += (v); }
#define PF_PARAMS , unsigned long long *pf_, unsigned long long &pf_t_
#define PF_ARGS , pf_, pf_t_
#define PF_NOARGS , pf_dummy_, pf_dummy_t_
#define PF_DUMMY unsigned long long pf_dummy_[16]; unsigned long long pf_dummy_t_ = 0;
#else
#define PF_DECL
#define PF_MARK(slot)
#define PF_ADD(slot, v)
#define PF_PARAMS
#define PF_ARGS
#define PF_NOARGS
#define PF_DUMMY
#endif

#ifndef SCORE_LIST_CAP
#define SCORE_LIST_CAP 256
#endif
#ifndef SCORE_TOUCH_CAP
#define SCORE_TOUCH_CAP 1024
#endif
constexpr int kListCap = SCORE_LIST_CAP;    // threshold-candidate list (uint16 columns)
constexpr int kTouchCap = SCORE_TOUCH_CAP;  // touched-column list of the sparse kernel (uint16 columns)
constexpr int kResCap = 64;      // result slots for top_k + 1 <= 64; larger k: res_cap() slots
constexpr int kMaxTopK = 1023;
constexpr int kQueueChunk = 8;   // jobs claimed per work-queue atomic
constexpr int kRowGroup = 8;     // W rows whose first loads are issued together
constexpr int kStreamDepth = 8;  // 64-entry chunks of a long W row requested per round trip
constexpr int kDenseUnroll = 4;  // 256-column steps of a dense W block per pipeline stage
constexpr int kDenseBatch = 8;   // ... per round trip in the generic (exact-tie / float64) form of that loop
typedef float vf4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int res_cap(int kk) { return kk <= kResCap ? kResCap : (kk + 63) / 64 * 64; }
__host__ __device__ constexpr size_t score_lds_bytes(int tile_cols, int acc_bytes, bool ft, bool touched, int kk = kResCap) {
    return static_cast<size_t>(tile_cols) * (acc_bytes + (ft ? 4 : 0)) + (touched ? kTouchCap * 2 : 0) + kListCap * 2 +
           res_cap(kk) * (8 + 4 + 4) + 16;
}

template <typename ACC>
__device__ __forceinline__ void lds_add(ACC *p, ACC v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <typename ACC>
__device__ __forceinline__ ACC lds_add_rtn(ACC *p, ACC v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// "No contribution yet" marker of the sparse kernel's accumulators: -0.0.  (-0) + p == p for
// every p != 0, a float sum that left -0 never returns to it under round-to-nearest, and
// -0 == 0 compares as a zero score (excluded in SPARSE mode exactly like scipy's `!= 0`).
__device__ __forceinline__ float untouched_value(float) { return __uint_as_float(0x80000000u); }
__device__ __forceinline__ double untouched_value(double) { return __longlong_as_double(static_cast<long long>(0x8000000000000000ull)); }
__device__ __forceinline__ bool is_untouched(float v) { return __float_as_uint(v) == 0x80000000u; }
__device__ __forceinline__ bool is_untouched(double v) { return static_cast<unsigned long long>(__double_as_longlong(v)) == 0x8000000000000000ull; }

// Key of an accumulator for selection: invalid columns map to -inf.
template <typename ACC>
__device__ __forceinline__ ACC sel_key(ACC v, bool zero_is_valid) {
    return (!zero_is_valid && v == ACC(0)) ? NegInf<ACC>::value() : v;
}

struct IdxAll {    // every column of the tile
    static constexpr bool kAll = true;
    __device__ __forceinline__ int operator()(int t) const { return t; }
};
struct IdxList {   // columns named by a uint16 list in LDS
    static constexpr bool kAll = false;
    const uint16_t *l;
    __device__ __forceinline__ int operator()(int t) const { return l[t]; }
};

template <typename ACC>
struct TileLds {
    ACC *acc;
    uint32_t *ft;
    uint16_t *tlist;
    uint16_t *clist;
    int *ccnt;       // length of clist while it is being filled
    ACC *res_s;
    int *res_i;
    uint32_t *res_a;
};

template <typename ACC>
__device__ __forceinline__ TileLds<ACC> carve_lds(unsigned char *smem, int S, bool ft, bool touched, int kk) {
    TileLds<ACC> L;
    const int rc = res_cap(kk);
    unsigned char *p = smem;
    L.res_s = reinterpret_cast<ACC *>(p);          p += rc * 8;
    L.acc = reinterpret_cast<ACC *>(p);            p += static_cast<size_t>(S) * sizeof(ACC);
    L.ft = reinterpret_cast<uint32_t *>(p);        if (ft) p += static_cast<size_t>(S) * 4;
    L.res_i = reinterpret_cast<int *>(p);          p += rc * 4;
    L.res_a = reinterpret_cast<uint32_t *>(p);     p += rc * 4;
    L.tlist = reinterpret_cast<uint16_t *>(p);     if (touched) p += kTouchCap * 2;
    L.clist = reinterpret_cast<uint16_t *>(p);     p += kListCap * 2;
    L.ccnt = reinterpret_cast<int *>(p);
    return L;
}

__device__ __forceinline__ int global_col(const ScoreArgs &a, int local) {
    return a.col_ids ? a.col_ids[local] : a.col_offset + local;
}

// Top-kk of the columns idx(0..n_idx) of one tile by (score, aux, global id), written sorted to
// L.res_*; returns the number of entries.  All 64 lanes must call it.
template <typename ACC, bool FT, typename IDX>
__device__ __forceinline__ int select_topk(const ScoreArgs &a, const TileLds<ACC> &L, IDX idx, int n_idx, int t0, bool zero_valid) {
    const int lane = lane_id();
    const ACC ninf = NegInf<ACC>::value();
    const ACC *acc = L.acc;

    // Candidates carry the LAYOUT column t0 + c instead of the item id: col_ids is ascending, so both
    // order alike and the (global-memory) translation is done once for the <= kk winners in
    // emit_result instead of sitting on the critical path of every comparison.
    auto make_cand = [&](int c) {
        Cand<ACC> x;
        x.score = acc[c];
        x.id = t0 + c;
        x.aux = FT ? L.ft[c]
                   : (a.mode == RTREC_TOPK_CANDIDATES ? static_cast<uint32_t>(a.col_rank[global_col(a, t0 + c)]) : 0u);
        return x;
    };
    // Rank of every candidate = number of better ones; only lanes that HOLD a candidate are visited (a
    // light user's touched list is mostly its own, excluded, items), and in SPARSE mode without
    // first-touch tracking every aux is 0, so the comparison is (score, id) alone.  Selection is the
    // per-user cost of a pass (C2: 43 % of it), and it is VALU-issue bound.
    auto rank_and_store = [&](bool have, const Cand<ACC> &mine) {
        const bool use_aux = FT || a.mode == RTREC_TOPK_CANDIDATES;
        unsigned long long m = __ballot(have);
        const int n_have = static_cast<int>(__builtin_popcountll(m));
        int rank = 0;
        if (use_aux) {
            for (; m; m &= m - 1) {
                const Cand<ACC> o = cand_readlane<ACC>(mine, __builtin_ctzll(m));
                rank += cand_better(o, mine) ? 1 : 0;
            }
        } else {
            for (; m; m &= m - 1) {
                const int t = __builtin_ctzll(m);
                const ACC os = readlane_t(mine.score, t);
                const int oi = readlane_i(mine.id, t);
                rank += (os > mine.score || (os == mine.score && oi > mine.id)) ? 1 : 0;
            }
        }
        if (have && rank < a.kk) { L.res_s[rank] = mine.score; L.res_i[rank] = mine.id; L.res_a[rank] = mine.aux; }
        return min(n_have, a.kk);
    };

    if (n_idx <= 64) {
        Cand<ACC> mine; mine.id = -1; mine.score = ninf; mine.aux = 0u;
        bool have = false;
        if (lane < n_idx) {
            const int c = idx(lane);
            if (sel_key(acc[c], zero_valid) != ninf) { mine = make_cand(c); have = true; }
        }
        return rank_and_store(have, mine);
    }

    if (a.kk > 64) {
        // more results than lanes: the lane-best threshold below cannot bound the answer.  Successive
        // scans, each bounded above by the previously emitted candidate (rare: top_k >= 64 requests)
        Cand<ACC> last; last.id = -1; last.score = ninf; last.aux = 0u;
        int n_big = 0;
        for (int r = 0; r < a.kk; ++r) {
            Cand<ACC> b; b.id = -1; b.score = ninf; b.aux = 0u;
            for (int t = lane; t < n_idx; t += 64) {
                const int c = idx(t);
                if (sel_key(acc[c], zero_valid) == ninf) continue;
                const Cand<ACC> x = make_cand(c);
                if (last.id >= 0 && !cand_better(last, x)) continue;
                if (cand_better(x, b)) b = x;
            }
            const Cand<ACC> w = wave_best(b);
            if (w.id < 0) break;
            last = w;
            if (lane == 0) { L.res_s[r] = w.score; L.res_i[r] = w.id; L.res_a[r] = w.aux; }
            n_big = r + 1;
        }
        return n_big;
    }

    // pass 1: per-lane best key, then tau = kk-th largest lane best (lower bound of the answer)
    ACC best = ninf;
    if (IDX::kAll) {   // whole tile: 4 columns per lane and step
        for (int c = lane * 4; c < n_idx; c += 256) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const ACC k = (c + j < n_idx) ? sel_key(acc[c + j], zero_valid) : ninf;
                best = k > best ? k : best;
            }
        }
    } else {
        for (int t = lane; t < n_idx; t += 64) {
            const ACC k = sel_key(acc[idx(t)], zero_valid);
            best = k > best ? k : best;
        }
    }
    // rank of this lane's best among the 64 lane bests (ties broken by lane id): pure VALU
    // (v_readlane + compare), no cross-lane LDS traffic
    ACC tau = ninf;
    if (a.kk <= 16) {
        // Any lower bound of the kk-th largest key serves (a looser one only admits a few more candidates to
        // pass 2).  The kk-th largest of the 16 QUAD maxima is one -- kk quads hold a key >= it -- and costs 16
        // rank steps instead of 64.
        ACC q = best;
        { const ACC o = shfl_xor_t(q, 1); q = o > q ? o : q; }
        { const ACC o = shfl_xor_t(q, 2); q = o > q ? o : q; }
        int rank = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const ACC o = readlane_t(q, 4 * t);
            rank += (o > q || (o == q && t < (lane >> 2))) ? 1 : 0;
        }
        const unsigned long long at = __ballot(rank == a.kk - 1);
        if (at) tau = readlane_t(q, __builtin_ctzll(at));
    } else {
        int rank = 0;
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            const ACC o = readlane_t(best, t);
            rank += (o > best || (o == best && t < lane)) ? 1 : 0;
        }
        const unsigned long long at = __ballot(rank == a.kk - 1);
        if (at) tau = readlane_t(best, __builtin_ctzll(at));
    }
    // pass 2: collect the columns with key >= tau (about kk of them): one wave-wide test per
    // 256-column step, the per-column ballots only where something was found
    int cnt = 0;
    if (IDX::kAll) {
        for (int c0 = lane * 4; c0 < n_idx + 256; c0 += 256) {   // uniform trip count
            if (c0 - lane * 4 >= n_idx) break;
            bool hit[4];
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + j;
                hit[j] = false;
                if (c < n_idx) {
                    const ACC k = sel_key(acc[c], zero_valid);
                    hit[j] = (k != ninf) && (k >= tau);
                }
                any = any || hit[j];
            }
            if (!__ballot(any)) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned long long m = __ballot(hit[j]);
                if (m) {
                    const int pos = cnt + lane_prefix(m);
                    if (hit[j] && pos < kListCap) L.clist[pos] = static_cast<uint16_t>(c0 + j);
                    cnt += __builtin_popcountll(m);
                }
            }
        }
    } else
    for (int tb = 0; tb < n_idx; tb += 64) {
        const int t = tb + lane;
        int c = 0;
        bool hit = false;
        if (t < n_idx) {
            c = idx(t);
            const ACC k = sel_key(acc[c], zero_valid);
            hit = (k != ninf) && (k >= tau);
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
            const int pos = cnt + lane_prefix(m);
            if (hit && pos < kListCap) L.clist[pos] = static_cast<uint16_t>(c);
            cnt += __builtin_popcountll(m);
        }
    }
    if (cnt <= 64) {
        Cand<ACC> mine; mine.id = -1; mine.score = ninf; mine.aux = 0u;
        const bool have = lane < cnt;
        if (have) mine = make_cand(L.clist[lane]);
        return rank_and_store(have, mine);
    }
    int n_out = 0;
    if (cnt <= kListCap) {
        for (int r = 0; r < a.kk; ++r) {
            Cand<ACC> b; b.id = -1; b.score = ninf; b.aux = 0u;
            int bt = -1;
            for (int t = lane; t < cnt; t += 64) {
                const uint16_t c = L.clist[t];
                if (c == 0xffffu) continue;
                const Cand<ACC> x = make_cand(c);
                if (cand_better(x, b)) { b = x; bt = t; }
            }
            const Cand<ACC> w = wave_best(b);
            if (w.id < 0) break;
            if (b.id == w.id && bt >= 0) L.clist[bt] = 0xffffu;
            if (lane == 0) { L.res_s[r] = w.score; L.res_i[r] = w.id; L.res_a[r] = w.aux; }
            n_out = r + 1;
        }
        return n_out;
    }
    // more threshold candidates than the list holds (mass ties): successive scans, each bounded
    // above by the previously emitted candidate
    Cand<ACC> last; last.id = -1; last.score = ninf; last.aux = 0u;
    for (int r = 0; r < a.kk; ++r) {
        Cand<ACC> b; b.id = -1; b.score = ninf; b.aux = 0u;
        for (int t = lane; t < n_idx; t += 64) {
            const int c = idx(t);
            if (sel_key(acc[c], zero_valid) == ninf) continue;
            const Cand<ACC> x = make_cand(c);
            if (last.id >= 0 && !cand_better(last, x)) continue;   // strictly below `last`
            if (cand_better(x, b)) b = x;
        }
        const Cand<ACC> w = wave_best(b);
        if (w.id < 0) break;
        last = w;
        if (lane == 0) { L.res_s[r] = w.score; L.res_i[r] = w.id; L.res_a[r] = w.aux; }
        n_out = r + 1;
    }
    return n_out;
}

// Write the tile's sorted result (L.res_*, n_out entries) either as the row's final answer
// (single-tile layouts) or as this tile's candidate list for the merge kernel.
template <typename ACC>
__device__ __forceinline__ void emit_result(const ScoreArgs &a, const TileLds<ACC> &L, int row, int tile, int n_out) {
    const int lane = lane_id();
    if (a.direct) {
        const int n_fin = min(n_out, a.top_k);
        for (int l = lane; l < a.top_k; l += 64) {
            const long long o = static_cast<long long>(row) * a.top_k + l;
            const bool ok = l < n_fin;
            const ACC sc = ok ? L.res_s[l] : NegInf<ACC>::value();
            a.out_id[o] = ok ? global_col(a, L.res_i[l]) : -1;
            a.out_score[o] = static_cast<float>(sc);
            if (a.out_score64) a.out_score64[o] = static_cast<double>(sc);
            if (a.out_aux) a.out_aux[o] = ok ? L.res_a[l] : 0u;
        }
        bool tie = false;
        if (a.detect_ties) {
            for (int l = lane; l + 1 < n_out; l += 64) tie = tie || (L.res_s[l] == L.res_s[l + 1]);
        }
        const unsigned long long any_tie = __ballot(tie);
        if (lane == 0) {
            a.out_cnt[row] = n_fin;
            if (any_tie) a.flag_list[atomicAdd(a.flag_len, 1)] = row;
        }
    } else {
        const size_t base = (static_cast<size_t>(row) * a.n_tiles + tile) * a.kk;
        for (int l = lane; l < n_out; l += 64) {
            reinterpret_cast<ACC *>(a.cand_score)[base + l] = L.res_s[l];
            a.cand_id[base + l] = global_col(a, L.res_i[l]);
            a.cand_aux[base + l] = L.res_a[l];
        }
        if (lane == 0) a.cand_cnt[static_cast<size_t>(row) * a.n_tiles + tile] = n_out;
    }
}

// ---- accumulate the W rows of one user's items into the tile, ascending item order ----------
// The accumulators are updated with plain LDS read-modify-write, not atomics: one wave owns the
// tile, its LDS operations execute in program order, and a W row never repeats a column, so the
// lanes of one instruction never collide.  (ds_add_f32 measures ~190 cycles per wave-instruction
// per CU on gfx950, a read+write pair ~14: tools/microbench/lds_atomic_rate.hip.)
// TOUCH: the value read back tells whether this is the first contribution to the column; such
// columns are appended to tlist.  FT: the first-touch position of the column is recorded.
// PREFILTER (sparse kernel, filter_interacted): the user's own items are taken out of the race while
// their row headers are loaded -- the accumulator of an interacted column is set to -inf (an
// absorbing value: -inf + p = -inf) and the column joins the touched list so that it is reset with
// the rest.  This replaces a separate pass (two dependent global loads per job) after the
// accumulation; scores of the other columns are untouched.
template <typename ACC, bool FT, bool TOUCH, bool PREFILTER = false>
__device__ __forceinline__ int accumulate_tile(const ScoreArgs &a, const TileLds<ACC> &L, int a0, int n_a, int tile,
                                               int t0, int ncol PF_PARAMS) {
    const int lane = lane_id();
    ACC *acc = L.acc;
    const int *tp = a.tile_ptr + static_cast<size_t>(tile) * (a.n_items + 1);
    int tcnt = 0;
    // Once a user is known to touch most of the tile (a long W row, or the list is full) the
    // touched list is pointless: stop maintaining it and let the caller scan / reset the whole tile.
    bool track = TOUCH;
    // One rounded product.  Where the accumulators start from the -0.0 "untouched" marker (TOUCH / FT) a product
    // of -0.0 -- a stored -0.0 rating (a decayed negative rating that underflowed), or a negative denormal rating
    // times a small coefficient -- would leave the marker in place: (-0) + (-0) = -0, the column would look
    // untouched to its next contribution and enter the touched list twice.  p + (+0) turns -0 into +0 and changes
    // nothing else, which is also what scipy computes (its sums start at +0, and (+0) + (-0) = +0).
    auto prod = [](ACC x, float v) {
        ACC p = x * static_cast<ACC>(v);
        if constexpr (TOUCH || FT) p = p + ACC(0);
        return p;
    };
    auto push = [&](bool first, int c) {
        if (!track) return;
        const unsigned long long m = __ballot(first);
        if (m) {
            const int pos = tcnt + lane_prefix(m);
            if (first && pos < kTouchCap) L.tlist[pos] = static_cast<uint16_t>(c);
            tcnt += __builtin_popcountll(m);
            if (tcnt > kTouchCap) track = false;
        }
    };
    for (int base = 0; base < n_a; base += 64) {
        const int p = base + lane;
        float x = 0.0f;
        int s = 0, e = 0, d = -1;
        int lc = -1;
        if (p < n_a) {
            const int item = a.xb_col[a0 + p];
            x = a.xb_val[a0 + p];
            if (a.row_hdr) {
                // one 16-byte record per (tile, item) instead of three gathers from three tables:
                // a user row costs one memory sector per item
                if (item < a.n_items) {
                    const int4 h = a.row_hdr[static_cast<size_t>(tile) * a.n_items + item];
                    s = h.x; e = h.y; d = h.z;
                    if (PREFILTER && a.filter) lc = h.w;
                }
            } else {
                if (item < a.n_items) {   // items newer than W have no row yet
                    s = tp[item];
                    e = tp[item + 1];
                    if (a.dense_idx) d = a.dense_idx[static_cast<size_t>(tile) * a.n_items + item];
                }
                if (PREFILTER && a.filter)
                    lc = (a.col_map ? (item < a.n_items ? a.col_map[item] : -1) : item - a.col_offset) - t0;
            }
        }
        if (PREFILTER && a.filter) {
            const bool mine = lc >= 0 && lc < ncol;        // the items of one row are distinct: no two lanes collide
            bool first = false;
            if (mine) {
                first = is_untouched(acc[lc]);
                acc[lc] = NegInf<ACC>::value();
            }
            push(first, lc);
        }
        unsigned long long live = __ballot(e > s || d >= 0);
        PF_MARK(PF_HDR)
        if (TOUCH && track && __ballot(e - s >= kTouchCap / 4 || d >= 0)) { track = false; tcnt = kTouchCap + 1; }
        while (live) {
            // Take the next kRowGroup non-empty rows (ascending item order) and issue the loads of
            // their first 64 entries together, so one memory round trip serves the whole group;
            // the accumulator updates are then issued row by row (scipy's accumulation order).
            int ss[kRowGroup], ee[kRowGroup], cc[kRowGroup], dd[kRowGroup];
            float vv[kRowGroup];
            ACC xx[kRowGroup];
            uint32_t ps[kRowGroup];
#pragma unroll
            for (int j = 0; j < kRowGroup; ++j) {
                ss[j] = 0; ee[j] = 0; dd[j] = -1; xx[j] = ACC(0); ps[j] = 0u;
                if (live) {
                    const int q = __builtin_ctzll(live);
                    live &= live - 1;
                    ss[j] = readlane_i(s, q);
                    ee[j] = readlane_i(e, q);
                    dd[j] = readlane_i(d, q);
                    xx[j] = static_cast<ACC>(readlane_f(x, q));
                    ps[j] = static_cast<uint32_t>(base + q);
                }
                cc[j] = -1; vv[j] = 0.0f;
                if (ss[j] + lane < ee[j]) { cc[j] = a.w_col[ss[j] + lane]; vv[j] = a.w_val[ss[j] + lane]; }
            }
            PF_MARK(PF_GROUP)
#pragma unroll
            for (int j = 0; j < kRowGroup; ++j) {
                if (dd[j] >= 0) {
                    // dense block: 4 consecutive accumulators per lane and step.  x * 0 is +-0 and
                    // never changes a sum, so the zero padding is inert.
                    const float *dv = a.dense_val + static_cast<size_t>(dd[j]) * a.tile_cols;
                    const ACC xj = xx[j];
                    if (!FT && sizeof(ACC) == 4) {
                        // float tile: whole-vector arithmetic (v_pk_mul_f32 / v_pk_add_f32 on the
                        // registers the 128-bit loads filled, no repacking moves) and a two-stage
                        // pipeline -- the W block of steps s+4..s+7 is in flight while steps s..s+3
                        // are read from LDS, updated (one rounded product, one rounded add per
                        // column, as in the scalar form) and written back.
                        const vf4 *dv4 = reinterpret_cast<const vf4 *>(dv) + lane;
                        vf4 *acc4 = reinterpret_cast<vf4 *>(acc) + lane;
                        const int steps = a.tile_cols >> 8;          // tile_cols is a multiple of 256
                        const float xs = static_cast<float>(xj);
                        int sidx = 0;
                        if (steps >= kDenseUnroll) {
                            vf4 wn[kDenseUnroll];
#pragma unroll
                            for (int u = 0; u < kDenseUnroll; ++u) wn[u] = dv4[64 * u];
                            for (; sidx + kDenseUnroll <= steps; sidx += kDenseUnroll) {
                                vf4 w[kDenseUnroll], o[kDenseUnroll];
#pragma unroll
                                for (int u = 0; u < kDenseUnroll; ++u) w[u] = wn[u];
                                if (sidx + 2 * kDenseUnroll <= steps) {
#pragma unroll
                                    for (int u = 0; u < kDenseUnroll; ++u) wn[u] = dv4[64 * (sidx + kDenseUnroll + u)];
                                }
#pragma unroll
                                for (int u = 0; u < kDenseUnroll; ++u) o[u] = acc4[64 * (sidx + u)];
#pragma unroll
                                for (int u = 0; u < kDenseUnroll; ++u) acc4[64 * (sidx + u)] = o[u] + w[u] * xs;
                            }
                        }
                        for (; sidx < steps; ++sidx) acc4[64 * sidx] = acc4[64 * sidx] + dv4[64 * sidx] * xs;
                        PF_MARK(PF_DENSE) PF_ADD(PF_N_DENSE, 1)
                        continue;
                    }
                    // the W block is fetched kDenseBatch steps at a time: one memory round trip per batch instead of one
                    // per step (the exact-tie pass of a single row is a chain of these: 16 steps per row of a 4096-column tile)
                    for (int c0 = lane * 4; c0 < a.tile_cols; c0 += 256 * kDenseBatch) {
                        float4 wb[kDenseBatch];
#pragma unroll
                        for (int u = 0; u < kDenseBatch; ++u) {
                            const int c = c0 + 256 * u;
                            wb[u] = c < a.tile_cols ? *reinterpret_cast<const float4 *>(dv + c) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        }
#pragma unroll
                        for (int u = 0; u < kDenseBatch; ++u) {
                            const int c = c0 + 256 * u;
                            if (c >= a.tile_cols) break;
                            const float4 w4 = wb[u];
                            ACC o0 = acc[c], o1 = acc[c + 1], o2 = acc[c + 2], o3 = acc[c + 3];
                            if (FT) {
                                if (w4.x != 0.0f && is_untouched(o0)) L.ft[c] = ps[j];
                                if (w4.y != 0.0f && is_untouched(o1)) L.ft[c + 1] = ps[j];
                                if (w4.z != 0.0f && is_untouched(o2)) L.ft[c + 2] = ps[j];
                                if (w4.w != 0.0f && is_untouched(o3)) L.ft[c + 3] = ps[j];
                                // a padded zero must not make the column look touched
                                if (w4.x != 0.0f) acc[c] = o0 + prod(xj, w4.x);
                                if (w4.y != 0.0f) acc[c + 1] = o1 + prod(xj, w4.y);
                                if (w4.z != 0.0f) acc[c + 2] = o2 + prod(xj, w4.z);
                                if (w4.w != 0.0f) acc[c + 3] = o3 + prod(xj, w4.w);
                            } else {
                                acc[c] = o0 + xj * static_cast<ACC>(w4.x);
                                acc[c + 1] = o1 + xj * static_cast<ACC>(w4.y);
                                acc[c + 2] = o2 + xj * static_cast<ACC>(w4.z);
                                acc[c + 3] = o3 + xj * static_cast<ACC>(w4.w);
                            }
                        }
                    }
                    continue;
                }
                if (ee[j] == ss[j]) continue;   // group not full
                bool first = false;
                if (cc[j] >= 0) {
                    const ACC old = acc[cc[j]];
                    first = is_untouched(old);
                    acc[cc[j]] = old + prod(xx[j], vv[j]);
                    if (FT && first) L.ft[cc[j]] = ps[j];
                }
                if (TOUCH) push(first, cc[j]);
                // the rest of a long row.  Full batches of kStreamDepth x 64 entries are branch-free:
                // all loads are issued before the first is consumed, and since the columns of one
                // row are distinct the updates go out as reads, then adds, then writes.
                const int eej = ee[j];
                const ACC xj = xx[j];
                const uint32_t pos = ps[j];
                int ob = ss[j] + 64;
                for (; ob + 64 * kStreamDepth <= eej; ob += 64 * kStreamDepth) {
                    const uint16_t *pc = a.w_col + ob + lane;
                    const float *pv = a.w_val + ob + lane;
                    int c[kStreamDepth];
                    float v[kStreamDepth];
                    ACC old[kStreamDepth];
#pragma unroll
                    for (int u = 0; u < kStreamDepth; ++u) { c[u] = pc[u * 64]; v[u] = pv[u * 64]; }
#pragma unroll
                    for (int u = 0; u < kStreamDepth; ++u) old[u] = acc[c[u]];
#pragma unroll
                    for (int u = 0; u < kStreamDepth; ++u) {
                        acc[c[u]] = old[u] + prod(xj, v[u]);
                        if (FT && is_untouched(old[u])) L.ft[c[u]] = pos;
                    }
                    if (TOUCH && track) {
#pragma unroll
                        for (int u = 0; u < kStreamDepth; ++u) push(is_untouched(old[u]), c[u]);
                    }
                }
                for (; ob < eej; ob += 64) {   // ragged tail, one masked chunk at a time
                    const int o = ob + lane;
                    int c = 0;
                    bool f2 = false;
                    if (o < eej) {
                        c = a.w_col[o];
                        const ACC old = acc[c];
                        f2 = is_untouched(old);
                        acc[c] = old + prod(xj, a.w_val[o]);
                        if (FT && f2) L.ft[c] = pos;
                    }
                    if (TOUCH) push(f2, c);
                }
                PF_MARK(PF_SPARSE) PF_ADD(PF_N_SPARSE_ROWS, 1) PF_ADD(PF_N_SPARSE_CHUNKS, (eej - ss[j] + 63) >> 6)
            }
        }
    }
    return tcnt;
}

// ---- DENSE / CANDIDATES modes: one (row, tile) job per workgroup, full-tile scan --------------
template <typename ACC>
__global__ __launch_bounds__(64) void score_tiles_dense_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const long long total = static_cast<long long>(a.n_rows) * a.n_tiles;
    const long long per_xcd = (total + 7) / 8;
    // tile-major work list dealt to the 8 XCDs in contiguous ranges (blocks b and b+8 share an
    // XCD and its L2), so an XCD streams one tile's slice of W at a time
    const long long w = static_cast<long long>(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (blockIdx.x / 8 >= per_xcd || w >= total) return;
    const int tile = static_cast<int>(w / a.n_rows);
    const int row = static_cast<int>(w % a.n_rows);

    const int lane = lane_id();
    const int S = a.tile_cols;
    const TileLds<ACC> L = carve_lds<ACC>(smem, S, false, false, a.kk);
    const ACC ninf = NegInf<ACC>::value();
    const int t0 = tile * S;
    const int ncol = min(S, a.n_cols - t0);
    const int xrow = a.row_ids ? a.row_ids[row] : row;
    const bool xok = xrow >= 0 && xrow < a.n_x_rows;       // anything else scores as an empty row
    const int a0 = xok ? a.xb_ptr[xrow] : 0;
    const int n_a = xok ? a.xb_ptr[xrow + 1] - a0 : 0;

    for (int c = lane * 4; c < S; c += 256) { L.acc[c] = ACC(0); L.acc[c + 1] = ACC(0); L.acc[c + 2] = ACC(0); L.acc[c + 3] = ACC(0); }
    PF_DUMMY
    accumulate_tile<ACC, false, false>(a, L, a0, n_a, tile, 0, 0 PF_NOARGS);

    if (a.mode == RTREC_TOPK_CANDIDATES) {
        for (int c = lane; c < ncol; c += 64)
            if (a.col_rank[global_col(a, t0 + c)] < 0) L.acc[c] = ninf;
    } else if (a.filter) {
        for (int p = lane; p < n_a; p += 64) {
            const int item = a.xb_col[a0 + p];
            const int lc = (a.col_map ? (item < a.n_items ? a.col_map[item] : -1) : item - a.col_offset) - t0;
            if (lc >= 0 && lc < ncol) L.acc[lc] = ninf;
        }
    }
    const int n_out = select_topk<ACC, false>(a, L, IdxAll{}, ncol, t0, /*zero_valid=*/true);
    emit_result<ACC>(a, L, row, tile, n_out);
}

// ---- score-vector export (SLIMElastic.predict / predict_selected / predict_all) ----------------
// Same accumulation as the DENSE mode, but the tile is written out instead of reduced:
// out[row, t0 + c] for the columns of the plain layout.
template <typename ACC>
__global__ __launch_bounds__(64) void score_rows_kernel(ScoreArgs a, ACC *out, long long out_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const long long total = static_cast<long long>(a.n_rows) * a.n_tiles;
    const long long per_xcd = (total + 7) / 8;
    const long long w = static_cast<long long>(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (blockIdx.x / 8 >= per_xcd || w >= total) return;
    const int tile = static_cast<int>(w / a.n_rows);
    const int row = static_cast<int>(w % a.n_rows);
    const int lane = lane_id();
    const int S = a.tile_cols;
    const TileLds<ACC> L = carve_lds<ACC>(smem, S, false, false, a.kk);
    const int t0 = tile * S;
    const int ncol = min(S, a.n_cols - t0);
    const int xrow = a.row_ids ? a.row_ids[row] : row;
    const bool xok = xrow >= 0 && xrow < a.n_x_rows;       // anything else scores as an empty row
    const int a0 = xok ? a.xb_ptr[xrow] : 0;
    const int n_a = xok ? a.xb_ptr[xrow + 1] - a0 : 0;
    for (int c = lane * 4; c < S; c += 256) { L.acc[c] = ACC(0); L.acc[c + 1] = ACC(0); L.acc[c + 2] = ACC(0); L.acc[c + 3] = ACC(0); }
    PF_DUMMY
    accumulate_tile<ACC, false, false>(a, L, a0, n_a, tile, 0, 0 PF_NOARGS);
    ACC *o = out + static_cast<long long>(row) * out_stride + t0;
    for (int c = lane; c < ncol; c += 64) o[c] = L.acc[c];
}

// ---- SPARSE mode: persistent waves, accumulators stay in LDS across jobs ----------------------
// Between jobs every accumulator holds the "untouched" marker; a job only visits, selects from
// and resets the columns it touched, so its cost follows the user's W rows, not the tile width.
template <typename ACC, bool FT>
__global__ __launch_bounds__(64) void score_sparse_kernel(ScoreArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int S = a.tile_cols;
    const TileLds<ACC> L = carve_lds<ACC>(smem, S, FT, true, a.kk);
    const ACC ninf = NegInf<ACC>::value();
    const ACC unt = untouched_value(ACC(0));
    const int n_rows = FT ? *a.row_list_len : a.n_rows;
    const int total = n_rows * a.n_tiles;
    if (FT && a.rescored && blockIdx.x == 0 && lane == 0) *a.rescored = n_rows;
    if (total == 0) return;              // the exact-tie pass usually has nothing to do: leave before touching LDS
    bool lds_ready = false;              // ... and a workgroup that never gets a job never initialises its tile either

    PF_DECL
#ifdef SCORE_PROFILE
    const unsigned long long pf_start_ = pf_t_;
#endif
    // jobs are claimed kQueueChunk at a time: one device-scope counter serves only ~90 claims/us
    int w_next = 0, w_end = 0;
    for (;;) {
        if (w_next >= w_end) {
            int w0 = 0;
            // (the exact-tie pass has a handful of rows: one job per claim spreads them over the chip)
            constexpr int chunk = FT ? 1 : kQueueChunk;
            if (lane == 0) w0 = atomicAdd(a.queue, chunk);
            w_next = readfirst_i(w0);
            PF_MARK(PF_QUEUE)
            if (w_next >= total) break;
            w_end = min(w_next + chunk, total);
        }
        if (!lds_ready) {
            for (int c = lane; c < S; c += 64) { L.acc[c] = unt; if (FT) L.ft[c] = 0xffffffffu; }
            lds_ready = true;
        }
        const int w = w_next++;
        const int tile = w / n_rows;             // tile-major: concurrent waves share a W tile in L2
        const int row = FT ? a.row_list[w % n_rows] : (w % n_rows);
        const int t0 = tile * S;
        const int ncol = min(S, a.n_cols - t0);
        const int xrow = a.row_ids ? a.row_ids[row] : row;
        const bool xok = xrow >= 0 && xrow < a.n_x_rows;   // anything else scores as an empty row
        const int a0 = xok ? a.xb_ptr[xrow] : 0;
        const int n_a = readfirst_i(xok ? a.xb_ptr[xrow + 1] - a0 : 0);
        PF_MARK(PF_ROWPTR) PF_ADD(PF_JOBS, 1)

        // interacted items leave the race inside accumulate_tile (PREFILTER)
        const int tcnt = (a.ablate & 1) ? 0 : accumulate_tile<ACC, FT, true, true>(a, L, a0, n_a, tile, t0, ncol PF_ARGS);
        const bool overflow = tcnt > kTouchCap;
        PF_ADD(PF_N_OVERFLOW, overflow ? 1 : 0)
        int n_out = 0;
        if (tcnt > 0 && !(a.ablate & 2)) {
            if (!overflow) n_out = select_topk<ACC, FT>(a, L, IdxList{L.tlist}, tcnt, t0, /*zero_valid=*/false);
            else n_out = select_topk<ACC, FT>(a, L, IdxAll{}, ncol, t0, /*zero_valid=*/false);
        }
        PF_MARK(PF_SELECT)
        emit_result<ACC>(a, L, row, tile, n_out);
        PF_MARK(PF_EMIT)

        // restore the invariant
        if (a.ablate & 4) {
        } else if (!overflow) {
            for (int t = lane; t < tcnt; t += 64) { const int c = L.tlist[t]; L.acc[c] = unt; if (FT) L.ft[c] = 0xffffffffu; }
        } else {
            for (int c = lane * 4; c < S; c += 256) {
                L.acc[c] = unt; L.acc[c + 1] = unt; L.acc[c + 2] = unt; L.acc[c + 3] = unt;
                if (FT) { L.ft[c] = 0xffffffffu; L.ft[c + 1] = 0xffffffffu; L.ft[c + 2] = 0xffffffffu; L.ft[c + 3] = 0xffffffffu; }
            }
        }
        PF_MARK(PF_RESET)
    }
#ifdef SCORE_PROFILE
    if (!FT && lane == 0) {
        pf_[PF_TOTAL] = __builtin_amdgcn_s_memtime() - pf_start_;
        for (int q = 0; q < PF_COUNT; ++q) atomicAdd(&g_score_prof[q], pf_[q]);
    }
#endif
}

struct MergeArgs {
    int n_rows;
    int n_lists;      // lists per row
    int kk;           // entries per list
    int top_k;
    long long list_stride;  // element stride between consecutive lists of one row (ids, aux)
    long long row_stride;   // element stride between rows (ids, aux)
    long long s_list_stride, s_row_stride;   // the same for the score array (in ACC elements)
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
// NS: candidate slots per lane (n_lists * kk <= 64 * NS) -- every round scans a lane's slots, so the common small merges
// (a few tiles or shards of ~10 entries each: one slot) cost a sixteenth of the general case's slot work.
template <typename ACC, int NS>
__global__ __launch_bounds__(64) void merge_topk_kernel(MergeArgs m) {
    const int lane = lane_id();
    const ACC *sc = reinterpret_cast<const ACC *>(m.in_score);
    const ACC ninf = NegInf<ACC>::value();
    // with a row list (the exact-tie pass) a small grid walks the list, which is usually empty; otherwise block = row
    const int n_work = m.row_list ? *m.row_list_len : m.n_rows;
    for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
    const int row = m.row_list ? m.row_list[work] : work;
    constexpr int kMaxPerLane = NS;   // n_lists * kk <= 1024 = 64 * 16
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
                mine[j].score = sc[l * m.s_list_stride + row * m.s_row_stride + r];
                mine[j].id = m.in_id[off];
                mine[j].aux = m.in_aux ? m.in_aux[off] : 0u;
            }
        }
    }
    const int want = m.detect_ties ? m.top_k + 1 : m.top_k;
    if constexpr (NS == 1) {
        // one candidate per lane: its rank is the number of better candidates -- `total` steps of readlane + compare
        // instead of top_k rounds of a wave-wide arg-max -- and the leading top_k lanes write their own slot
        const Cand<ACC> me = mine[0];
        int rank = 0;
        bool eq = false;
        for (int t = 0; t < total; ++t) {
            const Cand<ACC> o = cand_readlane<ACC>(me, t);
            rank += cand_better(o, me) ? 1 : 0;
            eq = eq || (t != lane && o.id >= 0 && o.score == me.score);
        }
        const bool valid = me.id >= 0;
        const int n_valid = static_cast<int>(__builtin_popcountll(__ballot(valid)));
        const int n_fin = min(n_valid, m.top_k);
        if (valid && rank < m.top_k) {
            const long long o = static_cast<long long>(row) * m.top_k + rank;
            m.out_id[o] = me.id;
            m.out_score[o] = static_cast<float>(me.score);
            if (m.out_score64) m.out_score64[o] = static_cast<double>(me.score);
            if (m.out_aux) m.out_aux[o] = me.aux;
        }
        if (lane >= n_fin && lane < m.top_k) {
            const long long o = static_cast<long long>(row) * m.top_k + lane;
            m.out_id[o] = -1;
            m.out_score[o] = -__builtin_huge_valf();
            if (m.out_score64) m.out_score64[o] = -__builtin_huge_val();
            if (m.out_aux) m.out_aux[o] = 0u;
        }
        // equal scores take consecutive ranks: two of them lie inside the leading `want` iff the first of a group does so
        // with room for a second, i.e. some member of a group has a rank below want - 1
        const unsigned long long tied = __ballot(valid && eq && rank < want - 1);
        if (lane == 0) {
            m.out_cnt[row] = n_fin;
            if (m.detect_ties && tied) m.flag_list[atomicAdd(m.flag_len, 1)] = row;
        }
        continue;
    }
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


// =============================================================================================
// SPARSE mode, "feature-row" form (float32 W): the fast path when W has FEW non-empty rows.
//
// Only items that some target selected as a feature with a non-zero weight have a row in W.  Under
// sklearn's alpha * n_samples scaling that set is small -- 61 rows on the 100k x 50k workload, 66 on the
// ML-20M shape -- and those rows are long (a popular item is a feature of most targets).  The shard of
// W is then a small dense R x n_cols matrix, and a user's score vector is the sum of its <= R rows
// weighted by the user's ratings, added in ascending item order.  This kernel keeps the ACCUMULATORS IN
// REGISTERS and streams that matrix through LDS:
//   * a workgroup of 8 waves scores 128 users (16 per wave) against all column tiles, one tile
//     (64 * REGS columns, R rows, one contiguous slice) at a time; slice t+1 is copied global -> LDS by
//     LDS-DMA (global_load_lds_dwordx4, double buffered) while tile t is computed;
//   * per tile every wave sweeps the rows its users own in ascending row order: one ds_read_b128 of the
//     row serves all 16 users, a user that owns the row adds x * w with two v_pk_mul_f32 + two
//     v_pk_add_f32 per 256 columns (one rounded product, one rounded add per column -- scipy's
//     csr_matmat arithmetic and order; a stored zero is never added by scipy, and adding x * 0 = +-0
//     here never changes a sum that started at +0);
//   * the interacted filter is a bitmap: the user's own columns are marked once per job (built in LDS,
//     parked in a per-wave scratch) and masked out of the candidates of each tile with scalar ops;
//   * top-(k+1) per user is a sorted list in two registers (lane j = rank j) carried across the tiles:
//     a column enters only if it beats the current (k+1)-th score, so after the first tiles a
//     (user, tile) pair costs four compares.  The list is the row's final answer: no per-tile
//     candidate lists, no merge kernel.
// Exact score ties: a tie inside the leading k is ordered from W by fr_ties_kernel (first-touch row of each tied
// column); a tie that reaches the (k+1)-th entry is re-scored by score_sparse_kernel<ACC, true> like before.
// No LDS accumulators, no touched lists, no per-user reset.
// =============================================================================================
struct FrArgs {
    int n_rows; const int *row_ids; int n_x_rows;
    const int *order;       // optional [n_rows]: position p of the work list scores (and writes) row order[p]
    const int *xb_ptr; const int *xb_col; const float *xb_val;
    int n_items;
    const int *fmap;        // [n_items] item -> row of the dense matrix, or -1
    const int *col_map;     // [n_items] item -> layout column, or -1
    const int *col_ids;     // [n_cols]  layout column -> item id (ascending)
    int n_cols, R, n_tiles;
    const float *wd;        // the tiles' slices, super-tile after super-tile: per tile its rows that hold a weight, ascending
    const unsigned long long *tile_rows;   // [n_frags][2]: bit f of word h set <=> the fragment holds row 64 h + f of its tile's slice
    const int *tile_off;    // [n_frags]: byte offset of the fragment inside its super-tile
    const int *frag_tile;   // [n_frags]: tile | first << 24 | last << 25 (a slice may be cut into consecutive fragments)
    const int *st_kb;       // [n_super + 1]: KiB offset of super-tile s in wd (one LDS-DMA wave-instruction moves 1 KiB)
    const int *st_tile;     // [n_super + 1]: first FRAGMENT of super-tile s (at most 64 fragments each)
    int n_super;
    int resident;           // one super-tile that stays in LDS for the life of the workgroup
    int consecutive;        // a wave takes 8 consecutive positions of the work order (pattern-sorted) instead of a strided deal
    int buf_bytes;          // bytes of one LDS buffer (>= the largest super-tile, >= the setup scratch)
    unsigned long long *mscratch;   // [gridDim.x][waves][users][n_tiles * REGS] interacted-column lane masks
    int kk, top_k, filter;
    int *out_id; float *out_score; uint32_t *out_aux; int *out_cnt;
    int dense_rule;                  // DENSE mode through this pass (see sg_emit): short-of-positives rows and every tie are flagged
    int *flag_list; int *flag_len;   // rows for the exact-tie pass
    int *tie_list; int *tie_len;     // rows whose ties fr_ties_kernel orders
    int *queue;
};

constexpr int kFrWaves = 16;                 // one workgroup per CU, four waves per SIMD
constexpr int kFrUsers = 8;                  // users per wave
constexpr int kFrWaveScratch = 4096;         // LDS bytes of setup scratch per wave (inside the second slice buffer)
constexpr int kFrMaskWords = 416;            // 64-bit mask words per user the scratch holds (n_tiles * REGS)
constexpr int kFrUserWords = 192;            // 32-bit words of global scratch per user: 128 ratings, 32 row words, pad
constexpr int kFrMaxKk = 16;                 // top_k + 1 entries of a list fit one DPP row
constexpr int kFrMaxRows = 128;

// 64-bit words of global scratch per wave: kFrUserWords 32-bit words per user, then n_tiles * regs mask words per user
__host__ __device__ constexpr size_t fr_wave_scratch_words(int n_tiles, int regs) {
    return static_cast<size_t>(kFrUsers) * kFrUserWords / 2 + static_cast<size_t>(kFrUsers) * n_tiles * regs;
}
constexpr int kFrCandCap = 64;               // candidates a wave buffers per merge round (one per lane)
#ifndef FR_MASKS_ON_DEMAND
#define FR_MASKS_ON_DEMAND 0
#endif
// 0 (default): every user's dense interacted-column masks are built and parked, as in round 4.  1: masks rebuilt on demand
// from the row's layout columns (WRITE_SIZE 364 -> ~57 MB per ML-20M pass) -- measured SLOWER, 1.29 against 1.22 ms per pass
// (A/B on one box, tools/ab_c3_score.sh): the rebuild costs more instructions than the parking costs bandwidth, and the
// columns kept in registers spill (33 VGPRs at 4 users per wave).  Kept for A/B runs.
constexpr bool kFrOnDemand = FR_MASKS_ON_DEMAND != 0;
constexpr int kFrHeadEntries = 128;          // entries of a user's row whose layout columns stay in registers
constexpr int kFrTailMax = kFrUserWords;     // further entries whose layout columns are parked as a list (4 B each); longer rows: dense masks
// per wave: the candidate buffer (scores, columns) and the interacted-column mask of ONE (user, tile), rebuilt on demand
__host__ __device__ constexpr size_t fr_wave_extra_bytes() { return static_cast<size_t>(kFrCandCap) * 8 + 64; }
constexpr int kFrStep = 2;                   // rows of W per sweep step
constexpr int kFrSetupChunks = 4;            // 64-entry chunks of a long user row whose loads are issued together
constexpr int kFrSetupChunksLong = 8;        // ... for rows of 1,152+ entries in the 2- and 4-user forms of the kernel
constexpr int kFrTileHeaderBytes = 512;      // per tile, in front of its first fragment: max |w| of each row (FR_TILE_HEADER_BYTES)
constexpr int kFrZeroRowBytes = 1024;        // one slice row of +0.0 (the widest tile: 256 columns)
// per-wave LDS setup scratch actually needed: the interacted-column mask words, a pad, 128 ratings
__host__ __device__ constexpr int fr_setup_scratch(int mask_words) { return ((mask_words * 8 + 256 + 512) + 255) / 256 * 256; }
// Streaming layout: two slice buffers (the second doubles as setup scratch).  RESIDENT layout (all of W's slices in ONE
// super-tile that fits next to the setup scratch): one buffer, loaded once per workgroup and kept across its jobs.
constexpr int kFrWavesStream = 8;            // streaming layout: two 8-wave workgroups per CU
__host__ __device__ constexpr size_t fr_lds_bytes(int buf_bytes, bool resident = false, int mask_words = kFrMaskWords) {
    return resident ? static_cast<size_t>(buf_bytes) + kFrWaves * fr_setup_scratch(mask_words) + kFrWaves * fr_wave_extra_bytes() +
                          kFrZeroRowBytes + 16
                    : 2 * static_cast<size_t>(buf_bytes) + kFrWavesStream * fr_wave_extra_bytes() + kFrZeroRowBytes + 16;
}

typedef __attribute__((address_space(3))) void fr_lds_void;
typedef __attribute__((address_space(1))) const void fr_glb_void;

template <int REGS> struct FrVec;
template <> struct FrVec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct FrVec<2> { typedef float type __attribute__((ext_vector_type(2))); };

__device__ __forceinline__ float float_prev(float f) {       // largest float < f (f finite or +inf, not NaN)
    const uint32_t b = __float_as_uint(f);
    if ((b << 1) == 0u) return __uint_as_float(0x80000001u);
    return __uint_as_float((b >> 31) ? b + 1u : b - 1u);
}
__device__ __forceinline__ float fr_shift_up(float v, float fill) {      // lane j <- lane j-1, lane 0 <- fill
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int fr_shift_up(int v, int fill) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ float fr_row_shift_up(float v, float fill) {   // within each 16-lane row: lane j <- lane j-1
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ int fr_row_shift_up(int v, int fill) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false);
}
__device__ __forceinline__ float fr_shift_down(float v, float fill) {    // lane j <- lane j+1, lane 63 <- fill
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}


template <int CTRL> __device__ __forceinline__ float fr_dpp_f(float v, float fill) {   // lanes without a source keep `fill`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// kk-th largest of the 64 per-lane values `best` (-inf = no value), or -inf when fewer than kk lanes hold
// one: a LOWER BOUND of the kk-th largest element of any set whose per-lane maxima these are.  Uniform.
__device__ __forceinline__ float fr_kth_lane_best(float best, int kk) {
    const float ninf = -__builtin_huge_valf();
    const int lane = lane_id();
    float tau = ninf;
    if (kk > 64) {           // kk - 64: over the 64 lane values themselves
        // kk - 64 <= 16 rounds of "take the maximum out": a DPP max-reduction (row_shr 1/2/4/8, row_bcast 15/31: six
        // v_max_f32_dpp, result in lane 63) and one select that retires the first lane holding it -- ~10 vector
        // instructions per round where ranking all 64 values costs ~350
        kk -= 64;
        float v = best;
        for (int r = 0; r < kk; ++r) {
            float t = v;
            t = fmaxf(t, fr_dpp_f<0x111>(t, ninf));
            t = fmaxf(t, fr_dpp_f<0x112>(t, ninf));
            t = fmaxf(t, fr_dpp_f<0x114>(t, ninf));
            t = fmaxf(t, fr_dpp_f<0x118>(t, ninf));
            t = fmaxf(t, fr_dpp_f<0x142>(t, ninf));      // row_bcast:15
            t = fmaxf(t, fr_dpp_f<0x143>(t, ninf));      // row_bcast:31
            tau = readlane_f(t, 63);
            if (tau == ninf) break;                      // fewer than kk values
            const unsigned long long at = __ballot(v == tau);
            v = lane == static_cast<int>(__builtin_ctzll(at)) ? ninf : v;
        }
    } else if (kk <= 16) {   // the kk-th largest of the 16 quad maxima is a bound as well, at a quarter of the steps
        float q = best;
        { const float o = shfl_xor_t(q, 1); q = o > q ? o : q; }
        { const float o = shfl_xor_t(q, 2); q = o > q ? o : q; }
        int rank = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float o = readlane_f(q, 4 * t);
            rank += (o > q || (o == q && t < (lane >> 2))) ? 1 : 0;
        }
        const unsigned long long at = __ballot(rank == kk - 1 && (lane & 3) == 0);
        if (at) tau = readlane_f(q, __builtin_ctzll(at));
    } else {
        int rank = 0;
#pragma unroll
        for (int t = 0; t < 64; ++t) {
            const float o = readlane_f(best, t);
            rank += (o > best || (o == best && t < lane)) ? 1 : 0;
        }
        const unsigned long long at = __ballot(rank == kk - 1);
        if (at) tau = readlane_f(best, __builtin_ctzll(at));
    }
    return tau;
}

// Compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).  The per-user
// register arrays of the kernel below are indexed with these constants only, so every element is a scalar
// the register allocator sees from the start (a runtime-indexed array of this size would live in scratch).
template <class F, int... I>
__device__ __forceinline__ void fr_static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void fr_static_for(F &&f) {
    fr_static_for_impl(static_cast<F &&>(f), std::make_integer_sequence<int, N>{});
}

typedef const __attribute__((address_space(4))) unsigned long long fr_const_u64;

// First-touch row of layout column c for a user who rates the rows own0 / own1 (bit f: row f / 64 + f of W): the lowest
// such row with a weight in that column -- scipy's csr_matmat meets the user's items in ascending order, and rows of W
// ascend with the item id.  Walks the fragments of the column's tile in global memory (rows ascending across and inside
// fragments; a row's index inside a fragment = the number of lower rows the fragment holds).  Rare path (exact ties).
template <int TC>
__device__ int fr_first_touch(const FrArgs &a, int c, unsigned long long own0, unsigned long long own1) {
    const int t = c / TC, cl = c % TC;
    const int n_frags = a.st_tile[a.n_super];
    int sup = 0;
    for (int g = 0; g < n_frags; ++g) {
        while (g >= a.st_tile[sup + 1]) ++sup;
        const int ft = a.frag_tile[g];
        if ((ft & 0xffffff) != t) continue;
        const unsigned long long m0 = a.tile_rows[2 * g], m1 = a.tile_rows[2 * g + 1];
        const float *base = a.wd + (static_cast<size_t>(a.st_kb[sup]) << 8) + (a.tile_off[g] >> 2) + cl;
        for (unsigned long long w = m0 & own0; w; w &= w - 1) {
            const int f = __builtin_ctzll(w);
            const int k = __builtin_popcountll(m0 & ((1ull << f) - 1ull));
            if (base[static_cast<size_t>(k) * TC] != 0.0f) return f;
        }
        const int k1 = __builtin_popcountll(m0);
        for (unsigned long long w = m1 & own1; w; w &= w - 1) {
            const int f = __builtin_ctzll(w);
            const int k = k1 + __builtin_popcountll(m1 & ((1ull << f) - 1ull));
            if (base[static_cast<size_t>(k) * TC] != 0.0f) return 64 + f;
        }
        if (ft & (1 << 25)) break;          // the tile's last fragment
    }
    return -1;
}

// UW: users per wave.  8 is the throughput form (every LDS read of a row of W is shared by 8 users).  A launch cannot be
// shorter than its longest job, though, and a job's length is its waves' serial work for UW users each: passes too small
// to give every workgroup slot several 8-user jobs (a rank's share of a row-sharded pass, mid-sized request batches)
// run the same code with 4 or 2 users per wave -- more, shorter jobs.
template <int REGS, int XR, int UW = kFrUsers>
__global__ __launch_bounds__(kFrWaves * 64, 4) void score_frows_kernel(FrArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef typename FrVec<REGS>::type vec;
    constexpr int TC = 64 * REGS;
    constexpr int ROWB = TC * 4;                            // bytes of one row of a tile
    constexpr int NL = (UW + 3) / 4;                        // list registers: four users per 64 lanes, 16 lanes each
    const int tid = static_cast<int>(threadIdx.x), wave = tid >> 6, lane = tid & 63;
    const int NW = static_cast<int>(blockDim.x) >> 6;       // waves of this workgroup: 16 (resident layout) or 8 (streaming)
    const int kk = a.kk;                                    // <= 16
    unsigned char *buf0 = smem;
    unsigned char *buf1 = smem + a.buf_bytes;               // second slice buffer; also (or, resident: only) setup scratch
    const bool resident = a.resident != 0;                  // W's slices stay in buf0 for the life of the workgroup
    const int wscratch = resident ? fr_setup_scratch(a.n_tiles * REGS) : kFrWaveScratch;
    const size_t lds_front = resident ? static_cast<size_t>(a.buf_bytes) + NW * static_cast<size_t>(wscratch)
                                      : 2 * static_cast<size_t>(a.buf_bytes);
    unsigned char *extra = smem + lds_front + static_cast<size_t>(wave) * fr_wave_extra_bytes();
    float *cv = reinterpret_cast<float *>(extra);                     // [kFrCandCap] candidate scores of one user
    int *cp = reinterpret_cast<int *>(cv + kFrCandCap);               // [kFrCandCap] their layout columns
    const unsigned char *zrow = smem + lds_front + NW * fr_wave_extra_bytes();
    int *s_job = reinterpret_cast<int *>(smem + lds_front + NW * fr_wave_extra_bytes() + kFrZeroRowBytes);
    for (int o = tid * 4; o < kFrZeroRowBytes; o += NW * 64 * 4) *reinterpret_cast<float *>(smem + (zrow - smem) + o) = 0.0f;
    const int n_jobs = (a.n_rows + UW * NW - 1) / (UW * NW);
    const float ninf = -__builtin_huge_valf();
    const int mwords = a.n_tiles * REGS;
    // per-wave global scratch: per user 192 words (its ratings by row of W: 128; pad), then the users' mask words
    const size_t wave_words = fr_wave_scratch_words(a.n_tiles, REGS);
    unsigned long long *sc_wave = a.mscratch + (static_cast<size_t>(blockIdx.x) * NW + wave) * wave_words;
    uint32_t *xs_wave = reinterpret_cast<uint32_t *>(sc_wave);
    unsigned long long *ms_wave = sc_wave + kFrUsers * kFrUserWords / 2;
    const uint32_t lane16 = static_cast<uint32_t>(lane) * (REGS * 4);         // byte offset of this lane in a row

    // super-tile s = tiles [st_tile[s], st_tile[s + 1]): their rows that hold a weight, tile after tile, st_kb[s + 1] - st_kb[s] KiB
    auto load_super = [&](int sidx, unsigned char *dst) {
        const int kb0 = a.st_kb[sidx], kb1 = a.st_kb[sidx + 1];
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.wd) + (static_cast<size_t>(kb0) << 10);
        for (int c = wave; c < kb1 - kb0; c += NW)
            __builtin_amdgcn_global_load_lds((fr_glb_void *)(src + (static_cast<size_t>(c) << 10) + lane * 16),
                                             (fr_lds_void *)(dst + (c << 10)), 16, 0, 0);
    };

    PF_DECL
#ifdef SCORE_PROFILE
    const unsigned long long pf_start_ = pf_t_;
#endif
    if (resident) {                                         // all of W's slices: once per workgroup
        load_super(0, buf0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // Streaming layout: a job = 128 users of the whole workgroup (the waves share the staged slices and meet at its
    // barriers).  Resident layout: W is read-only in LDS, so every WAVE claims its own 8 users and no barrier is left
    // inside the loop -- a wave never waits for a slower neighbour.
    const int n_wave_jobs = (a.n_rows + UW - 1) / UW;
    if (resident) __syncthreads();                          // W has landed
    for (;;) {
        int base, pstride;
        if (resident) {
            PF_MARK(PF_GROUP)
            int j = 0;
            if (lane == 0) j = atomicAdd(a.queue, 1);
            j = readfirst_i(j);
            PF_MARK(PF_QUEUE)
            if (j >= n_wave_jobs) break;
            // positions j, j + n_wave_jobs, ...: with rows handed over longest-first every wave job is the same mix
            base = a.consecutive ? j * UW : j; pstride = a.consecutive ? 1 : n_wave_jobs;
        } else {
            __syncthreads();                                // the previous job has left both buffers
            PF_MARK(PF_QUEUE)
            if (tid == 0) *s_job = atomicAdd(a.queue, 1);
            __syncthreads();
            const int job = *s_job;
            if (job >= n_jobs) break;
            load_super(0, buf0);
            // position p of the job's users goes to wave p % NW, so that with rows handed over longest-first (a.order)
            // every wave of the workgroup gets the same mix of long and short rows and the barriers find the waves level
            base = job * NW * UW + (a.consecutive ? wave * UW : wave); pstride = a.consecutive ? 1 : NW;
        }

        // ---- setup: per user its ratings of the R feature items (dense, lane = row of W) and the interacted-column
        //      masks, built in LDS (the second buffer is free until super-tile 0 starts) and parked in the wave's
        //      global scratch; the ratings come back into registers below ----
        unsigned long long *Ml = reinterpret_cast<unsigned long long *>(buf1 + wave * wscratch);
        float *xl = reinterpret_cast<float *>(Ml + (resident ? mwords : kFrMaskWords)) + 64;   // [128] floats
        // the eight users' row extents: lane u follows user u's chain of dependent loads (work order -> row id -> row
        // pointers), all eight chains in parallel; then the first 64 entries of every row and what they map to
        // (feature row, layout column), again all eight users' loads in flight together -- three memory round trips
        // per job instead of four per user
        int a0_l = 0, na_l = 0;
        if (lane < UW) {
            const int p = base + lane * pstride;
            if (p < a.n_rows) {
                const int r = a.order ? a.order[p] : p;
                const int xrow = a.row_ids ? a.row_ids[r] : r;
                if (xrow >= 0 && xrow < a.n_x_rows) { a0_l = a.xb_ptr[xrow]; na_l = a.xb_ptr[xrow + 1] - a0_l; }
            }
        }
        // (entries 0..63 and 64..127 of every row: a typical row is done after these two)
        int it0[UW], fm0[UW], cm0[UW], it1[UW], fm1[UW], cm1[UW];
        float xv0[UW], xv1[UW];
        fr_static_for<UW>([&](auto Uc) {
            constexpr int u = decltype(Uc)::value;
            const int a0 = readlane_i(a0_l, u), n_a = readlane_i(na_l, u);
            it0[u] = -1; xv0[u] = 0.0f; it1[u] = -1; xv1[u] = 0.0f;
            if (lane < n_a) { it0[u] = a.xb_col[a0 + lane]; xv0[u] = a.xb_val[a0 + lane]; }
            if (64 + lane < n_a) { it1[u] = a.xb_col[a0 + 64 + lane]; xv1[u] = a.xb_val[a0 + 64 + lane]; }
        });
        fr_static_for<UW>([&](auto Uc) {
            constexpr int u = decltype(Uc)::value;
            fm0[u] = -1; cm0[u] = -1; fm1[u] = -1; cm1[u] = -1;
            if (it0[u] >= 0 && it0[u] < a.n_items) {
                fm0[u] = a.fmap[it0[u]];
                if (a.filter) cm0[u] = a.col_map[it0[u]];
            }
            if (it1[u] >= 0 && it1[u] < a.n_items) {
                fm1[u] = a.fmap[it1[u]];
                if (a.filter) cm1[u] = a.col_map[it1[u]];
            }
        });
        auto mark = [&](int c) {      // interacted layout column c -> its bit in the user's mask words
            const int cl = c & (TC - 1);
            atomicOr(&Ml[(c / TC) * REGS + (cl & (REGS - 1))], 1ull << (cl / REGS));
        };
        // The interacted-column masks are needed only where a tile survives the bound test AND holds a candidate -- a few
        // (user, tile) pairs per user -- so they are no longer built and parked for every tile (round 4: 364 MB of stores per
        // ML-20M pass for 11.6 MB of output).  A row of up to 128 entries keeps its layout columns in registers (cm0 / cm1);
        // up to kFrTailMax further entries park theirs as a list (4 B each); the mask of one (user, tile) is rebuilt from those
        // when the selection asks for it.  Only longer rows (a few percent of the users) still build dense masks here.
        float xr[UW][XR];          // lane f: the user's rating of the item of row 64 * h + f of W (0: not owned)
        fr_static_for<UW>([&](auto Uc) {
            constexpr int u = decltype(Uc)::value;
            const int a0 = readlane_i(a0_l, u), n_a = readlane_i(na_l, u);
            const bool dense_u = !kFrOnDemand || n_a > kFrHeadEntries + kFrTailMax;
            uint32_t *tail_u = xs_wave + u * kFrUserWords;
            if (dense_u) for (int w = lane; w < mwords; w += 64) Ml[w] = 0ull;
            fr_static_for<2>([&](auto H) { xl[H() * 64 + lane] = 0.0f; });
            if (fm0[u] >= 0) xl[fm0[u]] = xv0[u];                           // the items of one row are distinct
            if (dense_u && cm0[u] >= 0) mark(cm0[u]);
            if (fm1[u] >= 0) xl[fm1[u]] = xv1[u];
            if (dense_u && cm1[u] >= 0) mark(cm1[u]);
            // rows beyond 128 entries, four chunks per round: their entries, then what they map to, are requested together
            // (two round trips per round).  The 2- and 4-user forms (smaller passes: row shards, API batches -- their length is
            // their longest user's setup) first take a very long row eight chunks at a time (sixteen spill): a 78k-item user of the
            // 1M x 500k shape is 305 rounds of four chunks, ~0.6 ms, in a 125k-user pass of 0.6 ms
            auto setup_round = [&](auto CHc, int b) {
                constexpr int CH = decltype(CHc)::value;
                int item[CH], f[CH], c[CH];
                float xv[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int q = b + 64 * j + lane;
                    item[j] = -1; xv[j] = 0.0f;
                    if (q < n_a) { item[j] = a.xb_col[a0 + q]; xv[j] = a.xb_val[a0 + q]; }
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    f[j] = -1; c[j] = -1;
                    if (item[j] >= 0 && item[j] < a.n_items) {
                        f[j] = a.fmap[item[j]];
                        if (a.filter) c[j] = a.col_map[item[j]];
                    }
                }
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    if (f[j] >= 0) xl[f[j]] = xv[j];
                    if (dense_u) { if (c[j] >= 0) mark(c[j]); }
                    else if (b + 64 * j + lane < n_a) tail_u[b + 64 * j + lane - kFrHeadEntries] = static_cast<uint32_t>(c[j]);
                }
            };
            int b = 128;
            if constexpr (UW < 8) {
                for (; b + 64 * kFrSetupChunksLong <= n_a; b += 64 * kFrSetupChunksLong)
                    setup_round(std::integral_constant<int, kFrSetupChunksLong>{}, b);
            }
            for (; b < n_a; b += 64 * kFrSetupChunks) setup_round(std::integral_constant<int, kFrSetupChunks>{}, b);
            // the ratings by row of W go straight from the LDS image into registers (they used to take a round trip through
            // the wave's global scratch)
            fr_static_for<XR>([&](auto H) { xr[u][H()] = xl[H() * 64 + lane]; });
            if (dense_u) for (int w = lane; w < mwords; w += 64) ms_wave[static_cast<size_t>(u) * mwords + w] = Ml[w];
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_dcache_inv();                   // dense masks are read back through the scalar cache
        // running top-kk of every user, in registers: user u = lanes (u & 3) * 16 .. + kk - 1 of register u >> 2,
        // lane offset j = rank j (a DPP row is 16 lanes, so a list shifts with row_shr:1)
        float ls4[NL];
        int lc4[NL];
        fr_static_for<NL>([&](auto G4) { ls4[G4()] = ninf; lc4[G4()] = -1; });

        PF_MARK(PF_HDR) PF_ADD(PF_JOBS, 1)
        if (!resident) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // super-tile 0 has landed
            __syncthreads();
        }
        PF_MARK(PF_GROUP)

        // rows of W that at least one of the wave's users rates: the sweep below visits only those
        unsigned long long own_or[XR];
        fr_static_for<XR>([&](auto H) {
            constexpr int h = decltype(H)::value;
            unsigned long long m = 0ull;
            fr_static_for<UW>([&](auto Uc) { m |= __ballot(xr[decltype(Uc)::value][h] != 0.0f); });
            own_or[h] = m;
        });

        // the eight users' sums over the current tile: they outlive a super-tile when the tile's slice continues in the next
        vec acc[UW];
        fr_static_for<UW>([&](auto Uc) { acc[decltype(Uc)::value] = vec(0.0f); });
        bool tile_skip = false;    // the current tile cannot place a column in any of the wave's lists (decided at its first fragment)
        // lane u (< 8): user u's sum of |ratings| over the feature rows, and its current (k+1)-th best score (-inf while the
        // list is filling) -- the cheap first level of the tile test below looks at all eight users in one compare
        float l1v = 0.0f, thrv = ninf;
        fr_static_for<UW>([&](auto Uc) {
            constexpr int u = decltype(Uc)::value;
            float b = fabsf(xr[u][0]);
            if constexpr (XR == 2) b = __fadd_rn(b, fabsf(xr[u][1]));
            b = __fadd_rn(b, fr_dpp_f<0x111>(b, 0.0f));
            b = __fadd_rn(b, fr_dpp_f<0x112>(b, 0.0f));
            b = __fadd_rn(b, fr_dpp_f<0x114>(b, 0.0f));
            b = __fadd_rn(b, fr_dpp_f<0x118>(b, 0.0f));
            b = __fadd_rn(b, fr_dpp_f<0x142>(b, 0.0f));
            b = __fadd_rn(b, fr_dpp_f<0x143>(b, 0.0f));
            const float tot = readlane_f(b, 63);
            l1v = lane == u ? tot : l1v;
        });
        for (int sidx = 0; sidx < a.n_super; ++sidx) {
            const unsigned char *wb = (sidx & 1) ? buf1 : buf0;
            if (sidx + 1 < a.n_super) load_super(sidx + 1, (sidx & 1) ? buf0 : buf1);
            const int t_lo = a.st_tile[sidx], t_hi = a.st_tile[sidx + 1];      // fragments of this super-tile
            // the fragments' row masks, offsets and tiles, lane g = fragment t_lo + g: one load per super-tile, read
            // back with v_readlane below (no memory latency inside the loop)
            unsigned long long nzv[2] = {0ull, 0ull};
            int toffv = 0, ftilev = 0;
            if (lane < t_hi - t_lo) {
                nzv[0] = a.tile_rows[(t_lo + lane) * 2];
                nzv[1] = a.tile_rows[(t_lo + lane) * 2 + 1];
                toffv = a.tile_off[t_lo + lane];
                ftilev = a.frag_tile[t_lo + lane];
            }
            const unsigned char *wlane = wb + lane16;             // this lane's columns in a slice row
            const unsigned char *wzero = zrow + lane16;           // a row of +0.0: what a step reads past the last row

            for (int g = t_lo; g < t_hi; ++g) {
                const int ft = readlane_i(ftilev, g - t_lo);
                const int t = ft & 0xffffff;                     // the tile this fragment belongs to
                // ---- tile-major sweep: every row of W that holds a weight in this tile (and that one of the wave's
                //      users rates) is read from LDS ONCE and applied to all eight users: acc_u += x_u * w, one
                //      rounded product and one rounded add per column (two v_pk_mul_f32 + two v_pk_add_f32 per user
                //      and 256 columns), rows ascending = scipy's order.  A user that does not rate the row has
                //      x_u = 0: x * w = +-0 changes no sum (a sum that starts at +0 never becomes -0), and the same
                //      holds for rows and blocks that are skipped altogether. ----
                const int toff = readlane_i(toffv, g - t_lo);
                if (ft & (1 << 24)) {                            // first fragment of a tile
                    // ---- can this tile matter at all?  Its header holds max |w| of every row of W over the tile's
                    //      columns (lane f = row 64 h + f), so  B_u = sum_f |x_uf| max|w_f|  bounds every score user u
                    //      can have in it, rounding of the float32 sums included in the 1e-4 margin.  A column enters a
                    //      list only by BEATING the user's (k+1)-th best: when B_u <= that for all eight users the
                    //      sweep, the selection and the tile's further fragments are skipped -- after the heavy first
                    //      tiles that is the fate of nine tiles in ten (ML-20M shape). ----
                    float wm[XR];
                    fr_static_for<XR>([&](auto H) {
                        wm[H()] = *reinterpret_cast<const float *>(wb + toff - kFrTileHeaderBytes + (H() * 64 + lane) * 4);
                    });
                    // first level, all eight users in one compare: (sum_f |x_uf|) * max_f max|w_f| -- decides 86 % of
                    // the (wave, tile) pairs on the ML-20M shape; only the others pay the per-user reductions below
                    float wmm = wm[0];
                    if constexpr (XR == 2) wmm = fmaxf(wmm, wm[1]);
                    wmm = fmaxf(wmm, fr_dpp_f<0x111>(wmm, 0.0f));
                    wmm = fmaxf(wmm, fr_dpp_f<0x112>(wmm, 0.0f));
                    wmm = fmaxf(wmm, fr_dpp_f<0x114>(wmm, 0.0f));
                    wmm = fmaxf(wmm, fr_dpp_f<0x118>(wmm, 0.0f));
                    wmm = fmaxf(wmm, fr_dpp_f<0x142>(wmm, 0.0f));
                    wmm = fmaxf(wmm, fr_dpp_f<0x143>(wmm, 0.0f));
                    const float wtop = __fmul_rn(readlane_f(wmm, 63), 1.0001f);
                    // (a user with NO rating on a row of W -- an empty row, a position past the end of the batch in the last job --
                    // has bound 0: every sum is +0, no candidate; it must not hold the tile open while its list is empty.  Such a
                    // slot used to keep its wave sweeping EVERY tile: a batch whose size is no multiple of the job size paid one
                    // unpruned wave at its end, 0.06-0.09 ms whatever its size -- tools/row_slice_probe.py, round 4)
                    const unsigned long long open1 = __ballot(lane < UW && l1v > 0.0f && !(thrv >= 0.0f && __fmul_rn(l1v, wtop) <= thrv));
                    bool all_skip = true;
                    if (open1) fr_static_for<UW>([&](auto Uc) {
                        constexpr int u = decltype(Uc)::value;
                        float b = __fmul_rn(fabsf(xr[u][0]), wm[0]);
                        if constexpr (XR == 2) b = __fadd_rn(b, __fmul_rn(fabsf(xr[u][1]), wm[1]));
                        b = __fadd_rn(b, fr_dpp_f<0x111>(b, 0.0f));
                        b = __fadd_rn(b, fr_dpp_f<0x112>(b, 0.0f));
                        b = __fadd_rn(b, fr_dpp_f<0x114>(b, 0.0f));
                        b = __fadd_rn(b, fr_dpp_f<0x118>(b, 0.0f));
                        b = __fadd_rn(b, fr_dpp_f<0x142>(b, 0.0f));      // row_bcast:15
                        b = __fadd_rn(b, fr_dpp_f<0x143>(b, 0.0f));      // row_bcast:31: lane 63 holds the wave's sum
                        const float bound = __fmul_rn(readlane_f(b, 63), 1.0001f);
                        const float thr_u = readlane_f(ls4[u >> 2], (u & 3) * 16 + kk - 1);
                        if (bound > 0.0f && !(thr_u >= 0.0f && bound <= thr_u)) all_skip = false;
                    });
                    tile_skip = all_skip;
                    if (!tile_skip) fr_static_for<UW>([&](auto Uc) { acc[decltype(Uc)::value] = vec(0.0f); });
                }
                if (tile_skip) continue;
                int below = 0;                                   // rows of this tile's slice before half h
                fr_static_for<XR>([&](auto H) {
                    constexpr int h = decltype(H)::value;
                    const unsigned long long nz =
                        (static_cast<unsigned long long>(readlane_u(static_cast<uint32_t>(nzv[h] >> 32), g - t_lo)) << 32) |
                        readlane_u(static_cast<uint32_t>(nzv[h]), g - t_lo);
                    unsigned long long rows = nz & own_or[h];
                    while (rows) {
                        // two rows per step: their LDS reads go out together; eight users' applies (80 vector
                        // instructions) cover the latency of the next pair
                        int f[kFrStep];
                        vec w[kFrStep];
#pragma unroll
                        for (int q = 0; q < kFrStep; ++q) {
                            const bool live = rows != 0ull;
                            f[q] = live ? __builtin_ctzll(rows) : 0;
                            rows &= rows - 1;          // 0 stays 0
                            // slice position of row f: the rows of the tile below it (scalar arithmetic)
                            const int pos = below + static_cast<int>(__builtin_popcountll(nz & ((1ull << f[q]) - 1ull)));
                            const unsigned char *src = live ? wlane + toff + pos * ROWB : wzero;
                            w[q] = *reinterpret_cast<const vec *>(src);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = 0; q < kFrStep; ++q) {
                            fr_static_for<UW>([&](auto Uc) {
                                constexpr int u = decltype(Uc)::value;
                                acc[u] = acc[u] + w[q] * readlane_f(xr[u][h], f[q]);
                            });
                        }
                        PF_ADD(PF_N_DENSE, kFrStep)
                    }
                    below += static_cast<int>(__builtin_popcountll(nz));
                });

                if (!(ft & (1 << 25))) continue;                 // the tile's slice continues in the next super-tile
                // ---- candidates, user after user: columns that beat the user's kk-th score and are not interacted ----
                fr_static_for<UW>([&](auto Uc) {
                    constexpr int u = decltype(Uc)::value;
                    constexpr int g = u >> 2, lb = (u & 3) * 16;
                    float best = acc[u][0];
                    fr_static_for<REGS>([&](auto Rc) { best = acc[u][Rc()] > best ? acc[u][Rc()] : best; });
                    float thr = readlane_f(ls4[g], lb + kk - 1);      // the user's current kk-th best score
                    float tcut = thr;
                    if (thr >= 0.0f) {
                        if (!__ballot(best > thr)) return;        // the common case after the first tiles: nothing enters
                    }
                    // the user's interacted columns in this tile, one 64-lane mask per register of the tile (see the setup)
                    unsigned long long ex[REGS];
                    fr_static_for<REGS>([&](auto Rc) { ex[Rc()] = 0ull; });
                    if (a.filter) {
                        const int n_a_u = readlane_i(na_l, u);
                        if (!kFrOnDemand || n_a_u > kFrHeadEntries + kFrTailMax) {
                            fr_const_u64 *mc = (fr_const_u64 *)(ms_wave + static_cast<size_t>(u) * mwords);
                            fr_static_for<REGS>([&](auto Rc) { ex[Rc()] = mc[t * REGS + Rc()]; });
                        } else {
                            unsigned long long *mx = reinterpret_cast<unsigned long long *>(cp + kFrCandCap);
                            if (lane < REGS) mx[lane] = 0ull;
                            auto put = [&](int c) {
                                if (c >= 0 && c / TC == t) {
                                    const int cl = c & (TC - 1);
                                    atomicOr(&mx[cl & (REGS - 1)], 1ull << (cl / REGS));
                                }
                            };
                            put(cm0[u]);
                            put(cm1[u]);
                            const uint32_t *tail_u = xs_wave + u * kFrUserWords;
                            for (int b = kFrHeadEntries; b < n_a_u; b += 64) {
                                int c = -1;
                                if (b + lane < n_a_u)
                                    c = static_cast<int>(__hip_atomic_load(&tail_u[b + lane - kFrHeadEntries], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                                put(c);
                            }
                            fr_static_for<REGS>([&](auto Rc) {
                                const unsigned long long v = mx[Rc()];       // LDS operations of one wave execute in order
                                ex[Rc()] = (static_cast<unsigned long long>(readfirst_i(static_cast<int>(v >> 32))) << 32) |
                                           static_cast<unsigned int>(readfirst_i(static_cast<int>(v & 0xffffffffull)));
                            });
                        }
                    }
                    if (thr == ninf) {
                        // The list is not full yet (first tile, or a user with few scored columns): everything
                        // non-zero would pass.  Take the tile's own kk-th best admissible score (a bound from the
                        // lane maxima) as the cut instead.
                        PF_ADD(PF_N_OVERFLOW, 1)
                        float bm = ninf;
                        fr_static_for<REGS>([&](auto Rc) {
                            constexpr int r = decltype(Rc)::value;
                            const float v = acc[u][r];
                            const float vm = (v != 0.0f && !((ex[r] >> lane) & 1ull)) ? v : ninf;
                            bm = vm > bm ? vm : bm;
                        });
                        // kk-th largest of the 64 lane maxima (the 16 quad maxima would do as a bound, but a loose one:
                        // half the tile can lie above it); -inf: fewer than kk lanes hold a score, and everything
                        // they hold (< kk * REGS <= kFrCandCap) fits the buffer
                        const float t0 = fr_kth_lane_best(bm, 64 + kk);
                        if (t0 != ninf) tcut = float_prev(t0);      // candidates are the values >= t0
                    }
                    int nc = 0;
                    const int rel = lane - lb;
                    const bool in = rel >= 0 && rel < kk;
                    // merge the buffered candidates into the user's list (lane lb + j = rank j)
                    auto merge = [&]() {
                        const float myv = lane < nc ? cv[lane] : ninf;
                        const int mycol = lane < nc ? cp[lane] : 0;
                        for (int i = 0; i < nc; ++i) {
                            const float v = readlane_f(myv, i);
                            const int col = readlane_i(mycol, i);
                            const float s = ls4[g];
                            const int c = lc4[g];
                            PF_ADD(PF_N_SPARSE_CHUNKS, 1)
                            if (!(v > readlane_f(s, lb + kk - 1))) continue;            // the threshold has risen meanwhile
                            // ties inside the fast pass order by higher column; exact ties are re-scored anyway
                            const bool better = in && ((s > v) || (s == v && c > col));
                            const int pos = static_cast<int>(__builtin_popcountll(__ballot(better)));
                            PF_ADD(PF_N_SPARSE_ROWS, 1)
                            const float s_up = fr_row_shift_up(s, ninf);
                            const int c_up = fr_row_shift_up(c, -1);
                            ls4[g] = !in || rel < pos ? s : (rel == pos ? v : s_up);
                            lc4[g] = !in || rel < pos ? c : (rel == pos ? col : c_up);
                        }
                        nc = 0;
                        const float thr_new = readlane_f(ls4[g], lb + kk - 1);
                        thrv = lane == u ? thr_new : thrv;
                    };
                    fr_static_for<REGS>([&](auto Rc) {
                        constexpr int r = decltype(Rc)::value;
                        const float v = acc[u][r];
                        // only non-zero sums compete (scipy keeps `!= 0`)
                        unsigned long long m = __ballot(v > tcut && v != 0.0f);
                        if (!m) return;
                        m &= ~ex[r];
                        if (!m) return;
                        const int cnt = static_cast<int>(__builtin_popcountll(m));
                        if (nc + cnt > kFrCandCap) merge();      // the buffer holds one ballot's worth (64): make room first
                        if ((m >> lane) & 1ull) {
                            const int pos = nc + lane_prefix(m);
                            cv[pos] = v;
                            cp[pos] = t * TC + lane * REGS + r;
                        }
                        nc += cnt;
                    });
                    if (nc > 0) merge();
                });
            }
            PF_MARK(PF_DENSE)
            if (!resident) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // super-tile s+1 has landed ...
                __syncthreads();                                    // ... and every wave has left super-tile s
            }
            PF_MARK(PF_GROUP)
        }

        // ---- the lists are the rows' answers: lane lb + j of list register g = rank j of user 4 g + lb / 16 ----
        int gid[NL], orow = 0;
        fr_static_for<NL>([&](auto G4) { gid[G4()] = lc4[G4()] >= 0 ? a.col_ids[lc4[G4()]] : -1; });    // all gathers in flight together
        if (lane < UW) {
            const int p = base + lane * pstride;
            orow = p < a.n_rows ? (a.order ? a.order[p] : p) : -1;
        }
        fr_static_for<UW>([&](auto Uc) {
            constexpr int u = decltype(Uc)::value;
            const int r = readlane_i(orow, u);
            if (r < 0) return;
            constexpr int lb = (u & 3) * 16;
            const int rel = lane - lb;
            const bool in = rel >= 0 && rel < kk;
            const float s = ls4[u >> 2];
            const int c = lc4[u >> 2];
            const int n_valid = static_cast<int>(__builtin_popcountll(__ballot(in && c >= 0)));
            const int n_fin = min(n_valid, a.top_k);
            const float nxt = fr_shift_down(s, ninf);
            const unsigned long long tie = __ballot(rel >= 0 && rel + 1 < n_valid && s == nxt);
            if (rel >= 0 && rel < a.top_k) {
                const long long o = static_cast<long long>(r) * a.top_k + rel;
                const bool ok = rel < n_fin;
                a.out_id[o] = ok ? gid[u >> 2] : -1;
                a.out_score[o] = ok ? s : ninf;
                if (a.out_aux) a.out_aux[o] = 0u;
            }
            const bool short_of_positives = a.dense_rule &&
                static_cast<int>(__builtin_popcountll(__ballot(rel >= 0 && rel < n_fin && s > 0.0f))) < a.top_k;
            if (lane == lb) {
                a.out_cnt[r] = n_fin;
                if (short_of_positives) {
                    a.flag_list[atomicAdd(a.flag_len, 1)] = r;
                } else if (tie) {
                    // A tie that reaches the (k+1)-th entry may have lost equal columns at the list's threshold: the row is
                    // re-scored by the exact-tie pass.  Otherwise the answer's SET is right and only the order of the tied
                    // entries is open: fr_ties_kernel settles it from W (SPARSE mode's first-touch order; DENSE mode orders
                    // ties by item id: such a row is re-scored too).
                    const bool boundary = n_valid == kk && ((tie >> (lb + kk - 2)) & 1ull);
                    if (boundary || a.dense_rule) a.flag_list[atomicAdd(a.flag_len, 1)] = r;
                    else a.tie_list[atomicAdd(a.tie_len, 1)] = r;
                }
            }
        });
        PF_MARK(PF_EMIT)
    }
#ifdef SCORE_PROFILE
    if (lane == 0) {
        pf_[PF_TOTAL] = __builtin_amdgcn_s_memtime() - pf_start_;
        for (int q = 0; q < PF_COUNT; ++q) atomicAdd(&g_score_prof[q], pf_[q]);
    }
#endif
}

}  // namespace rtrec

// ------------------------------------------------------------------------------------------
// C-ABI
// ------------------------------------------------------------------------------------------
using namespace rtrec;

#include <cstdio>
#include <cstdlib>
#include <new>

namespace {
// Diagnostic build only (-DRTREC_DEBUG_STAGES): synchronise after every launch and report the stage.
inline void debug_stage(hipStream_t st, const char *what) {
#ifdef RTREC_DEBUG_STAGES
    const hipError_t e = hipStreamSynchronize(st);
    std::fprintf(stderr, "[rtrec_amd] %s: %s\n", what, hipGetErrorString(e));
    std::fflush(stderr);
#else
    (void)st; (void)what;
#endif
}
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// HIP-event bracket around the dominant kernel of a scoring call.  One object per caller (rtrec_timer_create):
// the library itself keeps no state.
struct KernelTimer {
    hipEvent_t start = nullptr, stop = nullptr;
    double total_ms = 0.0;
    long long launches = 0;
    bool pending = false;
};
void timer_collect(KernelTimer &t) {
    if (!t.pending) return;
    float ms = 0.0f;
    if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
        t.total_ms += ms;
        t.launches += 1;
    }
    t.pending = false;
}

// Exact ties inside the leading k of a row scored by score_frows_kernel (a.tie_list): the answer's set is right, the order
// of the tied entries is open.  The reference orders equal scores by reverse first touch, then by higher item id.  One wave
// per row: read the row's list back, take each entry's first-touch row from W (fr_first_touch), turn it into the position
// of that item in the user's row (what the exact-tie pass reports as aux), rank the list by (score, position, item id)
// and rewrite the row.  A user with a stored zero rating on a row of W touches columns the feature-row kernel does not
// see (the rating is indistinguishable from "not rated" there): such a row is handed to the exact-tie pass instead.
template <int TC>
__global__ __launch_bounds__(64) void fr_ties_kernel(FrArgs a) {
    const int lane = lane_id();
    const int n = *a.tie_len;
    const float ninf = -__builtin_huge_valf();
    for (int w = blockIdx.x; w < n; w += gridDim.x) {
        const int r = a.tie_list[w];
        const int n_fin = a.out_cnt[r];
        float s = ninf;
        int g = -1, c = -1;
        if (lane < n_fin) {
            const long long o = static_cast<long long>(r) * a.top_k + lane;
            g = a.out_id[o];
            s = a.out_score[o];
            c = a.col_map[g];
        }
        const int xrow = a.row_ids ? a.row_ids[r] : r;
        const int a0 = a.xb_ptr[xrow], n_a = a.xb_ptr[xrow + 1] - a0;
        unsigned long long own0 = 0ull, own1 = 0ull;
        bool zero_stored = false;
        for (int b = 0; b < n_a; b += 64) {
            int fm = -1;
            float x = 0.0f;
            if (b + lane < n_a) {
                const int item = a.xb_col[a0 + b + lane];
                x = a.xb_val[a0 + b + lane];
                if (item < a.n_items) fm = a.fmap[item];
            }
            if (__ballot(fm >= 0 && x == 0.0f)) zero_stored = true;
            for (unsigned long long m = __ballot(fm >= 0); m; m &= m - 1) {
                const int f = readlane_i(fm, __builtin_ctzll(m));
                if (f < 64) own0 |= 1ull << f; else own1 |= 1ull << (f - 64);
            }
        }
        if (zero_stored) {
            if (lane == 0) a.flag_list[atomicAdd(a.flag_len, 1)] = r;
            continue;
        }
        int ftv = -1;
        if (lane < n_fin) ftv = fr_first_touch<TC>(a, c, own0, own1);
        int pv = 0;
        for (int b = 0; b < n_a; b += 64) {
            int fm = -1;
            if (b + lane < n_a) {
                const int item = a.xb_col[a0 + b + lane];
                if (item < a.n_items) fm = a.fmap[item];
            }
            for (int j = 0; j < n_fin; ++j) {
                const unsigned long long m = __ballot(fm >= 0 && fm == readlane_i(ftv, j));
                if (m && lane == j) pv = b + static_cast<int>(__builtin_ctzll(m));
            }
        }
        int rank = 0;
        for (int j = 0; j < n_fin; ++j) {
            const float sj = readlane_f(s, j);
            const int pj = readlane_i(pv, j), gj = readlane_i(g, j);
            rank += (sj > s || (sj == s && (pj > pv || (pj == pv && gj > g)))) ? 1 : 0;
        }
        if (lane < n_fin) {
            const long long o = static_cast<long long>(r) * a.top_k + rank;
            a.out_id[o] = g;
            a.out_score[o] = s;
            if (a.out_aux) a.out_aux[o] = static_cast<uint32_t>(pv);
        }
    }
}

#ifndef SG_GROUP
#define SG_GROUP 8
#endif
#include "score_seg.hip.h"

template <typename ACC>
void launch_merge_topk(unsigned grid, hipStream_t st, const MergeArgs &m) {
    const int total = m.n_lists * m.kk;
    if (total <= 64) hipLaunchKernelGGL(HIP_KERNEL_NAME(merge_topk_kernel<ACC, 1>), dim3(grid), dim3(64), 0, st, m);
    else if (total <= 128) hipLaunchKernelGGL(HIP_KERNEL_NAME(merge_topk_kernel<ACC, 2>), dim3(grid), dim3(64), 0, st, m);
    else if (total <= 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(merge_topk_kernel<ACC, 4>), dim3(grid), dim3(64), 0, st, m);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(merge_topk_kernel<ACC, 16>), dim3(grid), dim3(64), 0, st, m);
}

struct ScoreWs {
    size_t cand_score, cand_id, cand_aux, cand_cnt, flag_list, tie_list, flag_len, queue, total;
};
ScoreWs score_ws_layout(int n_rows, int n_tiles, int top_k) {
    ScoreWs w;
    const size_t kk = static_cast<size_t>(top_k) + 1;
    const size_t n = n_tiles > 1 ? static_cast<size_t>(n_rows) * n_tiles * kk : 0;
    size_t o = 0;
    w.cand_score = o; o = align_up(o + n * sizeof(double), 256);
    w.cand_id = o;    o = align_up(o + n * sizeof(int), 256);
    w.cand_aux = o;   o = align_up(o + n * sizeof(uint32_t), 256);
    w.cand_cnt = o;   o = align_up(o + (n_tiles > 1 ? static_cast<size_t>(n_rows) * n_tiles * sizeof(int) : 0), 256);
    w.flag_list = o;  o = align_up(o + static_cast<size_t>(n_rows) * sizeof(int), 256);
    w.tie_list = o;   o = align_up(o + static_cast<size_t>(n_rows) * sizeof(int), 256);
    // the per-call counters share one block (one memset): [0] rows for the exact-tie pass, [1] rows for fr_ties_kernel,
    // [2], [3] the job queues of the fast pass and the exact-tie pass
    w.flag_len = o;   w.queue = o + 8;  o = align_up(o + 256, 256);
    w.total = o;
    return w;
}

// Persistent grid: as many single-wave workgroups as the LDS footprint lets a CU hold.
unsigned persistent_grid(size_t lds_bytes, long long jobs) {
    int per_cu = static_cast<int>((160u * 1024u) / (lds_bytes > 0 ? lds_bytes : 1));
    per_cu = per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu);
    const long long g = 256ll * per_cu;
    return static_cast<unsigned>(jobs < g ? (jobs > 0 ? jobs : 1) : g);
}

// Feature-row form of the shard (rtrec_score_opts): usable for SPARSE mode with float32 accumulation.
struct FrLayout {
    const int *fmap = nullptr; const float *wd = nullptr; const int *order = nullptr;
    const int *col_ids = nullptr; const int *col_map = nullptr; const unsigned long long *tile_rows = nullptr;
    int rows = 0, tile_cols = 0, n_tiles = 0, n_frags = 0, n_super = 0, buf_bytes = 0;
    const int *tile_off = nullptr; const int *st_kb = nullptr; const int *st_tile = nullptr; const int *frag_tile = nullptr;
    unsigned long long *scratch = nullptr; size_t scratch_bytes = 0;
    int consecutive = 0;
    int users_per_wave = 0;      // 8 / 4 / 2 forces the kernel form; 0: by batch size
};
size_t fr_scratch_bytes(int n_tiles, int tile_cols) {
    const int regs = tile_cols / 64;
    return static_cast<size_t>(256) * kFrWaves * fr_wave_scratch_words(n_tiles, regs) * sizeof(unsigned long long);
}
bool fr_usable(const FrLayout &F, int kk) {
    if (!F.fmap || !F.wd || !F.scratch || !F.col_ids || !F.col_map || !F.tile_rows || !F.tile_off || !F.st_kb || !F.st_tile ||
        !F.frag_tile) return false;
    if (F.tile_cols != 256 && F.tile_cols != 128) return false;
    const int regs = F.tile_cols / 64;
    if (F.rows <= 0 || F.rows > kFrMaxRows || F.n_tiles <= 0 || F.n_tiles * regs > kFrMaskWords) return false;
    if (F.n_frags < F.n_tiles || F.n_super <= 0 || F.n_super > F.n_frags) return false;
    if (F.buf_bytes <= 0 || (F.buf_bytes & 1023)) return false;
    const int regs_ = F.tile_cols / 64;
    const bool resident = F.n_super == 1 && F.n_tiles <= 64 && fr_lds_bytes(F.buf_bytes, true, F.n_tiles * regs_) <= 160u * 1024u;
    // streaming: two 8-wave workgroups per CU, each with two slice buffers (the second doubles as 4 KiB of setup scratch per wave)
    if (!resident && (F.buf_bytes < kFrWavesStream * kFrWaveScratch || 2 * fr_lds_bytes(F.buf_bytes) > 160u * 1024u)) return false;
    if (F.scratch_bytes < fr_scratch_bytes(F.n_tiles, F.tile_cols)) return false;
    return kk >= 1 && kk <= kFrMaxKk;
}

// Segment form of the shard (rtrec_score_opts.d_sg_*): SPARSE mode, float32 accumulation, top_k <= 63.
struct SgLayout {
    const int2 *info = nullptr; const int *seg_ptr = nullptr; const uint32_t *w_ent = nullptr; long long nnz = 0;
    const uint32_t *bound = nullptr; const int *col_ids = nullptr; const int *order = nullptr;
    int T = 0, n_tiles = 0, rows = 0, n_cols = 0, order_longest_first = 0;
    const int *trow_ptr = nullptr; const int4 *trow = nullptr;      // heavy pass (optional, with the scratch)
    unsigned char *scratch = nullptr; size_t scratch_bytes = 0;
    void *aux_stream = nullptr; // rtrec_score_opts.aux_stream
    int heavy_min = 0;          // tuning knob (rtrec_score_opts.diagnostics bits 12-23): 0 = chosen from the pass size
};
bool sg_usable(const SgLayout &S, int kk) {
    if (!S.info || !S.seg_ptr || !S.w_ent || S.nnz <= 0 || S.nnz >= (1ll << 28) || !S.bound || !S.col_ids) return false;
    if (S.T < 256 || S.T > 4096 || (S.T & (S.T - 1))) return false;
    if (S.n_cols <= 0 || S.rows <= 0 || S.n_tiles != (S.n_cols + S.T - 1) / S.T || S.n_tiles > 128) return false;
    return kk >= 1 && kk <= kSgMaxKk;
}

template <typename ACC>
int score_impl(const ScoreArgs &base, int top_k, int acc_bytes, int32_t *d_out_ids, float *d_out_scores,
               double *d_out_scores64, uint32_t *d_out_aux, int32_t *d_out_count,
               unsigned char *ws, const ScoreWs &L, hipStream_t st, KernelTimer *tmr, const FrLayout &FR, const SgLayout &SG,
               int n_x_rows, int32_t *d_rescored, int32_t *d_flagged) {
    ScoreArgs a = base;
    // DENSE mode without the tiled layout: the fast pass runs as for SPARSE mode (non-zero sums, k + 1 entries) and flags
    // what DENSE semantics could change -- see sg_emit
    const bool dense_fast = (a.mode == RTREC_TOPK_DENSE) && base.tile_ptr == nullptr;
    const bool sparse = (a.mode == RTREC_TOPK_SPARSE) || dense_fast;
    const bool single = (a.n_tiles == 1);
    a.top_k = top_k;
    a.kk = sparse ? top_k + 1 : top_k;
    a.cand_score = ws + L.cand_score;
    a.cand_id = reinterpret_cast<int *>(ws + L.cand_id);
    a.cand_aux = reinterpret_cast<uint32_t *>(ws + L.cand_aux);
    a.cand_cnt = reinterpret_cast<int *>(ws + L.cand_cnt);
    // Without the tiled layout (a fast layout only: rtrec_slim_score_topk_opt) there is no exact-tie pass here: the rows it
    // would re-score are reported to the caller instead (d_flagged[0] = count, d_flagged[1 ..] = rows).
    const bool have_tiled = base.tile_ptr != nullptr;
    int *flag_list = have_tiled ? reinterpret_cast<int *>(ws + L.flag_list) : d_flagged + 1;
    int *flag_len = have_tiled ? reinterpret_cast<int *>(ws + L.flag_len) : d_flagged;
    int *queue = reinterpret_cast<int *>(ws + L.queue);   // [0]: fast pass, [1]: exact-tie pass
    if (!have_tiled && hipMemsetAsync(d_flagged, 0, 4, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    a.direct = single ? 1 : 0;
    a.out_id = d_out_ids; a.out_score = d_out_scores; a.out_score64 = d_out_scores64; a.out_aux = d_out_aux;
    a.out_cnt = d_out_count;
    a.detect_ties = sparse ? 1 : 0;
    a.flag_list = flag_list; a.flag_len = flag_len;
    a.row_list = flag_list; a.row_list_len = flag_len;
    a.queue = queue;
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    if (hipMemsetAsync(ws + L.flag_len, 0, 16, st) != hipSuccess) return RTREC_ERR_LAUNCH;      // both list lengths and both queues

    const long long total = static_cast<long long>(a.n_rows) * a.n_tiles;
    debug_stage(st, "score: begin");
    if (tmr) { timer_collect(*tmr); (void)hipEventRecord(tmr->start, st); }
    bool fr_done = false, sg_done = false;
    FrArgs f_fr{};
    if (sparse && sizeof(ACC) == 4 && fr_usable(FR, a.kk)) {
        // few, long rows in W: accumulators in registers, the dense R x n_cols matrix streamed through LDS
        FrArgs f{};
        f.n_rows = a.n_rows; f.row_ids = a.row_ids; f.n_x_rows = n_x_rows; f.order = FR.order;
        f.xb_ptr = a.xb_ptr; f.xb_col = a.xb_col; f.xb_val = a.xb_val; f.n_items = a.n_items;
        f.fmap = FR.fmap; f.col_map = FR.col_map; f.col_ids = FR.col_ids; f.tile_rows = FR.tile_rows;
        f.n_cols = a.n_cols; f.R = FR.rows; f.n_tiles = FR.n_tiles; f.wd = FR.wd;
        f.tile_off = FR.tile_off; f.st_kb = FR.st_kb; f.st_tile = FR.st_tile; f.n_super = FR.n_super; f.buf_bytes = FR.buf_bytes;
        f.frag_tile = FR.frag_tile;
        f.consecutive = FR.consecutive;
        f.mscratch = FR.scratch;
        f.kk = a.kk; f.top_k = top_k; f.filter = a.filter; f.dense_rule = dense_fast ? 1 : 0;
        f.out_id = d_out_ids; f.out_score = d_out_scores; f.out_aux = d_out_aux; f.out_cnt = d_out_count;
        f.flag_list = flag_list; f.flag_len = flag_len; f.queue = queue;
        f.tie_list = reinterpret_cast<int *>(ws + L.tie_list); f.tie_len = reinterpret_cast<int *>(ws + L.flag_len) + 1;
        // all slices in one super-tile that fits next to the setup scratch: W stays in LDS for the life of a 16-wave
        // workgroup; otherwise two 8-wave workgroups per CU stream the super-tiles (one computes while the other sets a
        // job up or waits at a super-tile barrier)
        f.resident = (FR.n_super == 1 && FR.n_tiles <= 64 &&
                      fr_lds_bytes(FR.buf_bytes, true, FR.n_tiles * (FR.tile_cols / 64)) <= 160u * 1024u) ? 1 : 0;
        const int nw = f.resident ? kFrWaves : kFrWavesStream;
        const int max_grid = f.resident ? 256 : 512;
        // users per wave: 8 when that still gives every workgroup slot a few jobs, else 4, else 2 (FR.users_per_wave forces one)
        const long long slots_rows = static_cast<long long>(max_grid) * nw;            // rows per round at one user per wave
        // (round 4: the STREAMING layout takes 8 users per wave only from ~393k rows on -- a launch cannot be shorter than a few
        // jobs, and a 64-user job sweeps every super-tile of W: 1M x 500k shape, 125k rows (one of 8 row shards): 0.91 -> 0.62 ms
        // with 4 per wave, 250k rows 1.22 -> 1.02 ms; ML-20M shape, all 138k rows 1.42 -> 1.39 ms; tools/fr_uw_sweep.py,
        // profiles/r04_fr_uw_sweep_*.json.  The resident layout's waves claim jobs on their own and keep 8 from 98k rows.)
        const long long uw8_min = f.resident ? 24 : 96;
        int uw = a.n_rows >= uw8_min * slots_rows ? 8 : (a.n_rows >= 6 * slots_rows ? 4 : 2);
        if (FR.users_per_wave == 8 || FR.users_per_wave == 4 || FR.users_per_wave == 2) uw = FR.users_per_wave;
        const int n_jobs = (a.n_rows + uw * nw - 1) / (uw * nw);
        const unsigned grid = static_cast<unsigned>(n_jobs < max_grid ? n_jobs : max_grid);
        const size_t lds = f.resident ? fr_lds_bytes(FR.buf_bytes, true, FR.n_tiles * (FR.tile_cols / 64)) : fr_lds_bytes(FR.buf_bytes);
        const bool two = FR.rows > 64;
#define RTREC_FR_LAUNCH(REGS_, XR_)                                                                                            \
        do {                                                                                                                   \
            if (uw == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(score_frows_kernel<REGS_, XR_, 8>), dim3(grid), dim3(nw * 64), lds, st, f);      \
            else if (uw == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(score_frows_kernel<REGS_, XR_, 4>), dim3(grid), dim3(nw * 64), lds, st, f); \
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(score_frows_kernel<REGS_, XR_, 2>), dim3(grid), dim3(nw * 64), lds, st, f);              \
        } while (0)
        if (FR.tile_cols == 256) {
            if (two) RTREC_FR_LAUNCH(4, 2); else RTREC_FR_LAUNCH(4, 1);
        } else {
            if (two) RTREC_FR_LAUNCH(2, 2); else RTREC_FR_LAUNCH(2, 1);
        }
#undef RTREC_FR_LAUNCH
        f_fr = f;
        fr_done = true;
    } else if (sparse && sizeof(ACC) == 4 && sg_usable(SG, a.kk)) {
        // many rows in W: one wave per user, tiles opened in descending score-bound order (score_seg.hip.h)
        SegArgs g{};
        g.n_rows = a.n_rows; g.row_ids = a.row_ids; g.order = SG.order; g.n_x_rows = n_x_rows;
        g.xb_ptr = a.xb_ptr; g.xb_col = a.xb_col; g.xb_val = a.xb_val; g.n_items = a.n_items;
        g.info = SG.info; g.seg_ptr = SG.seg_ptr; g.w_ent = SG.w_ent; g.nnz = SG.nnz; g.bound = SG.bound;
        g.col_ids = SG.col_ids; g.n_cols = SG.n_cols; g.T = SG.T; g.n_tiles = SG.n_tiles; g.R = SG.rows;
        g.kk = a.kk; g.top_k = top_k; g.filter = a.filter; g.dense_rule = dense_fast ? 1 : 0;
        g.out_id = d_out_ids; g.out_score = d_out_scores; g.out_aux = d_out_aux; g.out_cnt = d_out_count;
        g.flag_list = flag_list; g.flag_len = flag_len; g.queue = queue;
        const bool wide = SG.n_cols >= 0xffff || SG.rows >= 0xffff;        // the LDS lists hold 16-bit columns / rows otherwise
        const size_t wave_lds = sg_wave_lds(SG.T, wide ? 4 : 2);
        int waves_cu = static_cast<int>((160u * 1024u) / wave_lds);
        waves_cu = waves_cu > 4 * SG_OCC ? 4 * SG_OCC : waves_cu;             // SG_OCC (7) waves per SIMD: the kernel's registers
        int wg_cu = waves_cu / kSgWaves;
        wg_cu = wg_cu < 1 ? 1 : wg_cu;
        const long long cap = 256ll * wg_cu;
        // users per claim: kSgQueueChunk when the pass has that many for every wave slot of the chip, fewer for a smaller one
        // (a request-sized batch must not queue four users behind one wave while the other slots idle)
        long long chunk = a.n_rows / (cap * kSgWaves);
        chunk = chunk < 1 ? 1 : (chunk > kSgQueueChunk ? kSgQueueChunk : chunk);
        g.chunk = static_cast<int>(chunk);
        g.n_claims = static_cast<int>((static_cast<long long>(a.n_rows) + chunk - 1) / chunk);
        g.heavy_min = SG.heavy_min > 0 ? (SG.heavy_min - 1 < kSgCap ? SG.heavy_min - 1 : kSgCap)
                                       : sg_heavy_min_for(a.n_rows);
        const long long want = (static_cast<long long>(a.n_rows) + kSgWaves * chunk - 1) / (kSgWaves * chunk);
        const unsigned grid = static_cast<unsigned>(want < cap ? (want > 0 ? want : 1) : cap);
        const bool heavy = SG.trow_ptr && SG.trow && SG.scratch &&
                           SG.scratch_bytes >= sg_heavy_scratch_bytes(a.n_items, SG.n_tiles, SG.T);
        if (heavy) {
            g.trow_ptr = SG.trow_ptr; g.trow = SG.trow;
            g.xs = reinterpret_cast<float *>(SG.scratch);
            g.fl = SG.scratch + static_cast<size_t>(kSgHeavySlots) * a.n_items * 4;
            g.order_longest_first = (SG.order && SG.order_longest_first) ? 1 : 0;
        }
        // users too long for a wave's LDS lists first, one workgroup each (a percent of the rows): the launch is over
        // at once when there are none
        // With an auxiliary stream the heavy pass runs BESIDE the main kernel (fork / join by events): it is short of
        // parallelism (a thousand-odd users, its end is its longest user's critical path), the main kernel's tail too.
        hipStream_t aux = static_cast<hipStream_t>(SG.aux_stream);
        hipEvent_t e_fork = nullptr, e_join = nullptr;
        bool forked = false;
        if (heavy && aux && aux != st && a.n_rows >= kSgForkMinRows &&
            hipEventCreateWithFlags(&e_fork, hipEventDisableTiming) == hipSuccess) {
            if (hipEventCreateWithFlags(&e_join, hipEventDisableTiming) == hipSuccess) {
                forked = hipEventRecord(e_fork, st) == hipSuccess && hipStreamWaitEvent(aux, e_fork, 0) == hipSuccess;
            }
        }
        // (round 4: a FULL pass is bound by the wave-slot time of the two kernels together, not by a long user's latency: four
        // waves per long user leave more slots to the main kernel -- c3s 1.81 -> 1.74 ms over four alternated runs; smaller
        // passes keep eight: there the long user's latency is the pass)
        const int heavy_waves = (a.n_rows >= 49152 && sg_heavy_waves(SG.T) > 4) ? 4 : sg_heavy_waves(SG.T);
        if (heavy) hipLaunchKernelGGL(score_seg_heavy_kernel, dim3(kSgHeavySlots), dim3(heavy_waves * 64),
                                      sg_heavy_lds(SG.T, heavy_waves), forked ? aux : st, g);
        if (forked) (void)hipEventRecord(e_join, aux);
        if (wide) hipLaunchKernelGGL(HIP_KERNEL_NAME(score_seg_kernel<SG_GROUP, int, false>), dim3(grid), dim3(kSgWaves * 64), kSgWaves * wave_lds, st, g);
        else if (SG.T == 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(score_seg_kernel<SG_GROUP, uint16_t, true>), dim3(grid), dim3(kSgWaves * 64), kSgWaves * wave_lds, st, g);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(score_seg_kernel<SG_GROUP, uint16_t, false>), dim3(grid), dim3(kSgWaves * 64), kSgWaves * wave_lds, st, g);
        if (forked) (void)hipStreamWaitEvent(st, e_join, 0);
        if (e_fork) (void)hipEventDestroy(e_fork);          // (released when the recorded work has completed)
        if (e_join) (void)hipEventDestroy(e_join);
        sg_done = true;
    } else if (!have_tiled) {
        return RTREC_ERR_INVALID_ARG;       // no tiled layout and no usable fast layout
    } else if (sparse) {
        const size_t lds = score_lds_bytes(a.tile_cols, acc_bytes, false, true, a.kk);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(score_sparse_kernel<ACC, false>), dim3(persistent_grid(lds, total)), dim3(64),
                           lds, st, a);
    } else {
        const size_t lds = score_lds_bytes(a.tile_cols, acc_bytes, false, false, a.kk);
        const long long per_xcd = (total + 7) / 8;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(score_tiles_dense_kernel<ACC>), dim3(static_cast<unsigned>(per_xcd * 8)),
                           dim3(64), lds, st, a);
    }
    if (tmr) { (void)hipEventRecord(tmr->stop, st); tmr->pending = true; }
    debug_stage(st, "score tiles");
    if (fr_done) {      // ties inside the leading k: ordered from W, row by row (usually none)
        if (FR.tile_cols == 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(fr_ties_kernel<256>), dim3(64), dim3(64), 0, st, f_fr);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(fr_ties_kernel<128>), dim3(64), dim3(64), 0, st, f_fr);
        debug_stage(st, "fr_ties_kernel");
    }

    MergeArgs m{};
    m.n_rows = a.n_rows; m.n_lists = a.n_tiles; m.kk = a.kk; m.top_k = top_k;
    m.list_stride = a.kk; m.row_stride = static_cast<long long>(a.n_tiles) * a.kk;
    m.s_list_stride = m.list_stride; m.s_row_stride = m.row_stride;
    m.cnt_list_stride = 1; m.cnt_row_stride = a.n_tiles;
    m.in_score = a.cand_score; m.in_id = a.cand_id; m.in_aux = a.cand_aux; m.in_cnt = a.cand_cnt;
    m.out_id = d_out_ids; m.out_score = d_out_scores; m.out_score64 = d_out_scores64; m.out_aux = d_out_aux;
    m.out_cnt = d_out_count;
    m.detect_ties = sparse ? 1 : 0;
    m.flag_list = flag_list; m.flag_len = flag_len;
    m.row_list = nullptr; m.row_list_len = nullptr;
    if (!single && !fr_done && !sg_done) {      // the feature-row and segment kernels write final lists themselves
        launch_merge_topk<ACC>(static_cast<unsigned>(a.n_rows), st, m);
        debug_stage(st, "merge_topk_kernel");
    }

    if (!have_tiled) {
        if (d_rescored && hipMemsetAsync(d_rescored, 0, 4, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    } else if (sparse) {
        // exact reference tie order for the flagged rows only (first-touch tracking on)
        ScoreArgs f = a;
        f.kk = top_k;
        f.detect_ties = 0;
        f.queue = queue + 1;
        f.rescored = d_rescored;
        const size_t lds = score_lds_bytes(a.tile_cols, acc_bytes, true, true, f.kk);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(score_sparse_kernel<ACC, true>), dim3(persistent_grid(lds, total)), dim3(64),
                           lds, st, f);
        debug_stage(st, "score_sparse_kernel (exact ties)");
        if (!single) {
            MergeArgs mf = m;
            mf.kk = top_k;
            mf.list_stride = top_k; mf.row_stride = static_cast<long long>(a.n_tiles) * top_k;
            mf.s_list_stride = mf.list_stride; mf.s_row_stride = mf.row_stride;
            mf.detect_ties = 0;
            mf.row_list = flag_list; mf.row_list_len = flag_len;
            launch_merge_topk<ACC>(static_cast<unsigned>(a.n_rows < 256 ? a.n_rows : 256), st, mf);
            debug_stage(st, "merge_topk_kernel (exact ties)");
        }
    } else if (d_rescored && hipMemsetAsync(d_rescored, 0, 4, st) != hipSuccess) {
        return RTREC_ERR_LAUNCH;
    }
    return rtrec::launch_status();
}
}  // namespace

extern "C" int rtrec_timer_create(void **out_timer) {
    if (!out_timer) return RTREC_ERR_INVALID_ARG;
    KernelTimer *t = new (std::nothrow) KernelTimer();
    if (!t) return RTREC_ERR_LAUNCH;
    if (hipEventCreate(&t->start) != hipSuccess || hipEventCreate(&t->stop) != hipSuccess) { delete t; return RTREC_ERR_LAUNCH; }
    *out_timer = t;
    return RTREC_OK;
}

extern "C" int rtrec_timer_read(void *timer, double *total_ms, int64_t *launches, int32_t reset) {
    if (!timer) return RTREC_ERR_INVALID_ARG;
    KernelTimer &t = *static_cast<KernelTimer *>(timer);
    timer_collect(t);
    if (total_ms) *total_ms = t.total_ms;
    if (launches) *launches = t.launches;
    if (reset) { t.total_ms = 0.0; t.launches = 0; }
    return RTREC_OK;
}

extern "C" void rtrec_timer_destroy(void *timer) {
    if (!timer) return;
    KernelTimer *t = static_cast<KernelTimer *>(timer);
    if (t->start) (void)hipEventDestroy(t->start);
    if (t->stop) (void)hipEventDestroy(t->stop);
    delete t;
}

#ifdef SCORE_PROFILE
extern "C" int rtrec_amd_seg_heavy_profile(unsigned long long *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_heavy_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -4;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_heavy_prof), z, sizeof(z)) != hipSuccess) return -4; }
    return 0;
}
extern "C" int rtrec_amd_seg_profile(unsigned long long *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_seg_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -4;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_seg_prof), z, sizeof(z)) != hipSuccess) return -4; }
    return 0;
}
extern "C" int rtrec_amd_score_profile(unsigned long long *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(rtrec::g_score_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -4;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rtrec::g_score_prof), z, sizeof(z)) != hipSuccess) return -4; }
    return 0;
}
#endif

extern "C" size_t rtrec_slim_score_fr_scratch_bytes(int32_t fr_n_tiles, int32_t fr_tile_cols) {
    if (fr_n_tiles <= 0 || (fr_tile_cols != 256 && fr_tile_cols != 128)) return 0;
    return fr_scratch_bytes(fr_n_tiles, fr_tile_cols);
}

extern "C" size_t rtrec_slim_score_sg_scratch_bytes(int32_t n_items, int32_t sg_n_tiles, int32_t sg_tile_cols) {
    if (n_items <= 0 || sg_n_tiles <= 0 || sg_tile_cols <= 0) return 0;
    return sg_heavy_scratch_bytes(n_items, sg_n_tiles, sg_tile_cols);
}

extern "C" size_t rtrec_slim_score_workspace_bytes(int32_t n_rows, int32_t n_tiles, int32_t top_k) {
    if (n_rows < 0 || n_tiles <= 0 || top_k <= 0) return 0;
    return score_ws_layout(n_rows, n_tiles, top_k).total;
}

extern "C" int rtrec_slim_score_topk_opt(int32_t n_rows, const int32_t *d_row_ids,
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
                                     const rtrec_score_opts *opts) {
    if (n_rows < 0 || n_items <= 0 || n_cols <= 0 || top_k <= 0) return RTREC_ERR_INVALID_ARG;
    if (d_row_hdr && (reinterpret_cast<uintptr_t>(d_row_hdr) & 15u)) return RTREC_ERR_INVALID_ARG;
    if (n_rows == 0) return RTREC_OK;
    if (!d_xb_ptr || !d_out_ids || !d_out_scores || !d_out_count || !d_workspace) return RTREC_ERR_INVALID_ARG;
    if (!d_tile_ptr) {
        // fast layout only (feature rows or segments): SPARSE mode, float32; the rows whose lists hold an exact score tie are
        // handed back in opts->d_flagged for the caller to re-score against the tiled layout (which it may build only then)
        if (!opts || !opts->d_flagged || (mode != RTREC_TOPK_SPARSE && mode != RTREC_TOPK_DENSE) || acc_f64) return RTREC_ERR_INVALID_ARG;
        tile_cols = 256; n_tiles = 1;
        d_w_col = nullptr; d_w_val = nullptr; d_dense_idx = nullptr; d_dense_val = nullptr; d_row_hdr = nullptr;
    }
    if (mode < 0 || mode > 2) return RTREC_ERR_INVALID_ARG;
    if (mode == RTREC_TOPK_CANDIDATES && !d_col_rank) return RTREC_ERR_INVALID_ARG;
    if ((d_col_ids == nullptr) != (d_col_map == nullptr)) return RTREC_ERR_INVALID_ARG;
    if ((d_dense_idx == nullptr) != (d_dense_val == nullptr)) return RTREC_ERR_INVALID_ARG;
    if (tile_cols < 256 || tile_cols > 65536 || (tile_cols % 256) != 0) return RTREC_ERR_UNSUPPORTED;
    if (d_tile_ptr && n_tiles != (n_cols + tile_cols - 1) / tile_cols) return RTREC_ERR_INVALID_ARG;
    const int acc_bytes = acc_f64 ? 8 : 4;
    // the exact-tie instantiation keeps an accumulator AND a first-touch word per column in LDS
    if (top_k > kMaxTopK || static_cast<long long>(n_tiles) * (top_k + 1) > 1024) return RTREC_ERR_UNSUPPORTED;
    if (score_lds_bytes(tile_cols, acc_bytes, true, true, top_k + 1) > 160u * 1024u) return RTREC_ERR_UNSUPPORTED;
    if (static_cast<long long>(n_rows) * n_tiles >= (1ll << 31)) return RTREC_ERR_UNSUPPORTED;
    const ScoreWs L = score_ws_layout(n_rows, n_tiles, top_k);
    if (workspace_bytes < L.total) return RTREC_ERR_WORKSPACE;

    ScoreArgs a{};
    a.n_rows = n_rows; a.row_ids = d_row_ids; a.xb_ptr = d_xb_ptr; a.xb_col = d_xb_col; a.xb_val = d_xb_val;
    a.n_items = n_items; a.n_cols = n_cols; a.col_offset = col_offset; a.col_ids = d_col_ids; a.col_map = d_col_map;
    a.tile_cols = tile_cols; a.n_tiles = n_tiles;
    a.tile_ptr = d_tile_ptr; a.w_col = d_w_col; a.w_val = d_w_val; a.col_rank = d_col_rank;
    a.dense_idx = d_dense_idx; a.dense_val = d_dense_val;
    a.row_hdr = reinterpret_cast<const int4 *>(d_row_hdr);
    a.filter = filter_interacted; a.mode = mode;
    a.n_x_rows = (opts && opts->n_x_rows > 0) ? opts->n_x_rows : 0x7fffffff;
#ifdef RTREC_DIAGNOSTICS
    a.ablate = opts ? (opts->diagnostics & 0xff) : 0;       // tools/score_ablate.sh, diagnostic build only
#else
    a.ablate = 0;
#endif
    KernelTimer *tmr = opts ? static_cast<KernelTimer *>(opts->timer) : nullptr;
    FrLayout FR;
    if (opts && opts->d_fr_map && opts->d_fr_w) {
        FR.fmap = opts->d_fr_map; FR.wd = opts->d_fr_w; FR.rows = opts->fr_rows; FR.tile_cols = opts->fr_tile_cols;
        FR.n_tiles = opts->fr_n_tiles; FR.n_frags = opts->fr_n_frags; FR.n_super = opts->fr_n_super;
        FR.buf_bytes = opts->fr_buf_bytes; FR.tile_off = opts->d_fr_tile_off; FR.st_kb = opts->d_fr_super_kb;
        FR.st_tile = opts->d_fr_super_tile; FR.frag_tile = opts->d_fr_frag_tile;
        FR.scratch = static_cast<unsigned long long *>(opts->d_fr_scratch); FR.scratch_bytes = opts->fr_scratch_bytes;
        FR.order = opts->d_row_order; FR.consecutive = (opts->d_row_order && opts->row_order_grouped) ? 1 : 0;
        FR.users_per_wave = (opts->diagnostics >> 8) & 0xf;
        FR.col_ids = opts->d_fr_col_ids; FR.col_map = opts->d_fr_col_map;
        FR.tile_rows = reinterpret_cast<const unsigned long long *>(opts->d_fr_tile_rows);
        if (FR.n_tiles != (n_cols + FR.tile_cols - 1) / (FR.tile_cols > 0 ? FR.tile_cols : 1)) return RTREC_ERR_INVALID_ARG;
    }
    SgLayout SG;
    if (opts && opts->d_sg_info && opts->d_sg_ent) {
        if (reinterpret_cast<uintptr_t>(opts->d_sg_info) & 7u) return RTREC_ERR_INVALID_ARG;
        SG.info = reinterpret_cast<const int2 *>(opts->d_sg_info); SG.seg_ptr = opts->d_sg_ptr; SG.w_ent = opts->d_sg_ent;
        SG.nnz = opts->sg_nnz; SG.bound = opts->d_sg_bound; SG.col_ids = opts->d_sg_col_ids;
        SG.T = opts->sg_tile_cols; SG.n_tiles = opts->sg_n_tiles; SG.rows = opts->sg_rows; SG.n_cols = opts->sg_n_cols;
        SG.order = opts->d_row_order; SG.order_longest_first = opts->row_order_longest_first;
        SG.trow_ptr = opts->d_sg_trow_ptr; SG.trow = reinterpret_cast<const int4 *>(opts->d_sg_trow);
        SG.scratch = static_cast<unsigned char *>(opts->d_sg_scratch); SG.scratch_bytes = opts->sg_scratch_bytes;
        SG.heavy_min = (opts->diagnostics >> 12) & 0xfff;
        SG.aux_stream = opts->aux_stream;
        if (SG.n_cols != n_cols || (reinterpret_cast<uintptr_t>(opts->d_sg_trow) & 15u) ||
            (reinterpret_cast<uintptr_t>(opts->d_sg_scratch) & 15u)) return RTREC_ERR_INVALID_ARG;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char *ws = static_cast<unsigned char *>(d_workspace);
    if (acc_f64)
        return score_impl<double>(a, top_k, 8, d_out_ids, d_out_scores, d_out_scores64, d_out_aux, d_out_count, ws, L, st,
                                  tmr, FR, SG, a.n_x_rows, opts ? opts->d_rescored : nullptr, opts ? opts->d_flagged : nullptr);
    return score_impl<float>(a, top_k, 4, d_out_ids, d_out_scores, d_out_scores64, d_out_aux, d_out_count, ws, L, st,
                             tmr, FR, SG, a.n_x_rows, opts ? opts->d_rescored : nullptr, opts ? opts->d_flagged : nullptr);
}

extern "C" int rtrec_slim_score_topk(int32_t n_rows, const int32_t *d_row_ids,
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
                                     void *d_workspace, size_t workspace_bytes, void *stream) {
    return rtrec_slim_score_topk_opt(n_rows, d_row_ids, d_xb_ptr, d_xb_col, d_xb_val, n_items, n_cols, col_offset,
                                     d_col_ids, d_col_map, tile_cols, n_tiles, d_tile_ptr, d_w_col, d_w_val,
                                     d_dense_idx, d_dense_val, d_row_hdr, d_col_rank, top_k, filter_interacted, mode,
                                     acc_f64, d_out_ids, d_out_scores, d_out_scores64, d_out_aux, d_out_count,
                                     d_workspace, workspace_bytes, stream, nullptr);
}

extern "C" int rtrec_slim_score_rows(int32_t n_rows, const int32_t *d_row_ids,
                                     const int32_t *d_xb_ptr, const int32_t *d_xb_col, const float *d_xb_val,
                                     int32_t n_items, int32_t n_cols, int32_t col_offset,
                                     int32_t tile_cols, int32_t n_tiles,
                                     const int32_t *d_tile_ptr, const uint16_t *d_w_col, const float *d_w_val,
                                     int32_t acc_f64, void *d_out, int64_t out_stride, void *stream) {
    if (n_rows < 0 || n_items <= 0 || n_cols <= 0 || out_stride < n_cols) return RTREC_ERR_INVALID_ARG;
    if (n_rows == 0) return RTREC_OK;
    if (!d_xb_ptr || !d_tile_ptr || !d_out) return RTREC_ERR_INVALID_ARG;
    if (tile_cols < 256 || tile_cols > 16384 || (tile_cols % 256) != 0) return RTREC_ERR_UNSUPPORTED;
    if (n_tiles != (n_cols + tile_cols - 1) / tile_cols) return RTREC_ERR_INVALID_ARG;
    ScoreArgs a{};
    a.n_rows = n_rows; a.row_ids = d_row_ids; a.xb_ptr = d_xb_ptr; a.xb_col = d_xb_col; a.xb_val = d_xb_val;
    a.n_items = n_items; a.n_cols = n_cols; a.col_offset = col_offset;
    a.tile_cols = tile_cols; a.n_tiles = n_tiles; a.tile_ptr = d_tile_ptr; a.w_col = d_w_col; a.w_val = d_w_val;
    a.n_x_rows = 0x7fffffff;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    const long long total = static_cast<long long>(n_rows) * n_tiles;
    const unsigned grid = static_cast<unsigned>(((total + 7) / 8) * 8);
    const size_t lds = score_lds_bytes(tile_cols, acc_f64 ? 8 : 4, false, false);
    if (acc_f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(score_rows_kernel<double>), dim3(grid), dim3(64), lds, st, a,
                           static_cast<double *>(d_out), static_cast<long long>(out_stride));
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(score_rows_kernel<float>), dim3(grid), dim3(64), lds, st, a,
                           static_cast<float *>(d_out), static_cast<long long>(out_stride));
    return rtrec::launch_status();
}

extern "C" int rtrec_slim_merge_topk_strided(int32_t n_rows, int32_t n_lists, int32_t top_k,
                                             const int32_t *d_in_ids, const float *d_in_scores,
                                             const double *d_in_scores64, const uint32_t *d_in_aux,
                                             const int32_t *d_in_count,
                                             int64_t list_stride, int64_t row_stride,
                                             int64_t score64_list_stride, int64_t score64_row_stride,
                                             int64_t count_list_stride, int64_t count_row_stride,
                                             int32_t *d_out_ids, float *d_out_scores, int32_t *d_out_count,
                                             void *stream) {
    if (n_rows < 0 || n_lists <= 0 || top_k <= 0) return RTREC_ERR_INVALID_ARG;
    if (n_rows == 0) return RTREC_OK;
    if (!d_in_ids || !d_in_scores || !d_in_count || !d_out_ids || !d_out_scores || !d_out_count) return RTREC_ERR_INVALID_ARG;
    if (static_cast<long long>(n_lists) * top_k > 1024) return RTREC_ERR_UNSUPPORTED;
    MergeArgs m{};
    m.n_rows = n_rows; m.n_lists = n_lists; m.kk = top_k; m.top_k = top_k;
    m.list_stride = list_stride; m.row_stride = row_stride;
    m.cnt_list_stride = count_list_stride; m.cnt_row_stride = count_row_stride;
    m.in_id = d_in_ids; m.in_aux = d_in_aux; m.in_cnt = d_in_count;
    m.out_id = d_out_ids; m.out_score = d_out_scores; m.out_score64 = nullptr; m.out_aux = nullptr; m.out_cnt = d_out_count;
    m.detect_ties = 0; m.flag_list = nullptr; m.flag_len = nullptr; m.row_list = nullptr; m.row_list_len = nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (d_in_scores64) {
        m.in_score = d_in_scores64;
        m.s_list_stride = score64_list_stride; m.s_row_stride = score64_row_stride;
        launch_merge_topk<double>(static_cast<unsigned>(n_rows), st, m);
    } else {
        m.in_score = d_in_scores;
        m.s_list_stride = list_stride; m.s_row_stride = row_stride;
        launch_merge_topk<float>(static_cast<unsigned>(n_rows), st, m);
    }
    return rtrec::launch_status();
}

extern "C" int rtrec_slim_merge_topk(int32_t n_rows, int32_t n_lists, int32_t top_k,
                                     const int32_t *d_in_ids, const float *d_in_scores, const double *d_in_scores64,
                                     const uint32_t *d_in_aux, const int32_t *d_in_count,
                                     int32_t *d_out_ids, float *d_out_scores, int32_t *d_out_count,
                                     void *stream) {
    const int64_t ls = static_cast<int64_t>(n_rows) * top_k;
    return rtrec_slim_merge_topk_strided(n_rows, n_lists, top_k, d_in_ids, d_in_scores, d_in_scores64, d_in_aux,
                                         d_in_count, ls, top_k, ls, top_k, n_rows, 1,
                                         d_out_ids, d_out_scores, d_out_count, stream);
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
