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
// seg_ptr[row][t] .. [t + 1] = its SEGMENT in tile t ({column, weight} records); bound[row][t] = max |w| of that segment (bfloat16, rounded up).
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
    const uint32_t *w_ent;     // [nnz][2]: {column inside the tile, float32 bits of the weight}
    long long nnz;
    const uint32_t *bound;     // [R][64]: lane l -> bfloat16 bounds of tiles 2l (low half) and 2l + 1 (high half)
    const int *col_ids;        // layout column -> item id
    int n_cols, T, n_tiles, R;
    int kk, top_k, filter;
    int *out_id; float *out_score; uint32_t *out_aux; int *out_cnt;
    int *flag_list; int *flag_len;
    int *queue;
    int chunk;                 // users per queue claim (1 .. kSgQueueChunk): small passes claim fewer, so that every wave gets work
    int n_claims;              // ceil(n_rows / chunk): claim c takes the users at positions c, c + n_claims, c + 2 n_claims, ... of the order
    int heavy_min;             // users with more items than this go to score_seg_heavy_kernel (<= kSgCap; see kSgSmallPass)
    int dense_rule;            // DENSE mode through this pass: a row with fewer than top_k POSITIVE scores is flagged as well
    // long users (more items than a wave's LDS lists hold) are left to score_seg_heavy_kernel, one workgroup per user
    int order_longest_first;   // `order` is sorted by row length, longest first: the long users are its head
    const int *trow_ptr;       // [n_tiles + 1]: the segments of tile t, ascending item ...
    const int4 *trow;          // ... as {item, begin, end, 0}
    float *xs;                 // [slots][n_items] zeroed scratch: a heavy user's ratings by item (null: no heavy pass)
    unsigned char *fl;         // [slots][n_tiles * T] zeroed scratch: 1 = layout column of one of the user's items
};

constexpr int kSgWaves = 4;          // waves per workgroup; they share nothing (no barrier in the kernel)
#ifndef SG_CAP
#define SG_CAP 512
#endif
constexpr int kSgCap = SG_CAP;          // items of a user a wave keeps in LDS (layout columns, rows of W, ratings); longer users: heavy pass
#ifndef SG_NT_USER_ROWS
#define SG_NT_USER_ROWS 0
#endif
constexpr int kSgQueueChunk = 4;     // users per queue claim of a full-size pass (SegArgs::chunk)
constexpr int kSgMaxKk = 64;         // top_k + 1 list entries: one per lane
// A pass with fewer users than the chip has wave slots (~7k) leaves most of it idle, and its latency is its longest
// user's: there a user of more than n_rows / 16 items (at least 32) gets a whole workgroup (score_seg_heavy_kernel: eight
// waves share the user's tiles) instead of one wave.  The threshold rises with the pass size (sg_heavy_min_for); from ~49k rows
// on only users beyond a wave's LDS lists get a workgroup.
constexpr int kSgSmallHeavyMin = 32;
constexpr int kSgForkMinRows = 8192;      // passes from this size on run the heavy pass on the caller's auxiliary stream
// Round 4 (tools/seg_shard_probe.py, the strided user slices of 2 .. 32 row shards of c3s): a pass of 8k-50k users is still
// short of waves, and its long single-wave users set its length -- 17,312 users: threshold 512 0.64 ms, 384 0.51 ms, 256
// 0.57 ms; 8,656 users: 512 0.63 ms, 384 0.50 ms, 256 0.38 ms; 34,624: 0.74 / 0.72 / 0.96 ms; 69k and more: 512 is best
// (the workgroup kernel costs more slot time than it saves latency once the chip is full).
__host__ __device__ constexpr int sg_heavy_min_for(int n_rows) {
    return n_rows >= 49152 ? kSgCap
         : n_rows >= 12288 ? (kSgCap < 384 ? kSgCap : 384)
         : n_rows >= 6144 ? (kSgCap < 256 ? kSgCap : 256)
         : (n_rows / 16 < kSgSmallHeavyMin ? kSgSmallHeavyMin : (n_rows / 16 > kSgCap ? kSgCap : n_rows / 16));
}
#ifndef SG_OCC
#define SG_OCC 7          // waves per SIMD the main kernel is compiled for (72 VGPRs)
#endif
#ifndef SG_HEAVY_SLOTS
#define SG_HEAVY_SLOTS 1024
#endif
#ifndef SG_HEAVY_WAVES
#define SG_HEAVY_WAVES 8
#endif
constexpr int kSgHeavySlots = SG_HEAVY_SLOTS;   // workgroups (= scratch slots) of the heavy pass: 8 waves each, three per CU
                                               // (measured on the ML-20M-shape structured workload: 16 waves x 512 slots 0.95 ms,
                                               // 8 x 1024 0.65 ms, 4 x 2048 0.76 ms for its 1,673 long users)

// per wave: T accumulators, then per item a rating (float), a layout column and a row of W (IDX: uint16_t while both
// counts stay below 65535, else int)
__host__ __device__ constexpr size_t sg_wave_lds(int T, int idx_bytes) {
    return static_cast<size_t>(T + 64) * 4 + static_cast<size_t>(kSgCap) * (4 + 2 * idx_bytes) + 64;   // + 64 junk slots, + claimed chunk
}
__host__ __device__ constexpr int sg_heavy_waves(int T) { return T <= 2048 ? SG_HEAVY_WAVES : (SG_HEAVY_WAVES < 8 ? SG_HEAVY_WAVES : 8); }
__host__ __device__ constexpr size_t sg_heavy_lds(int T, int waves = 0) {
    return static_cast<size_t>(waves > 0 ? waves : sg_heavy_waves(T)) * (static_cast<size_t>(T + 64) * 4 + 128 * 4 + 64 * 8) + 64 +
           static_cast<size_t>(kSgCap) * 12;        // + rating | layout column | row of W of up to kSgCap items (list path)
}
__host__ __device__ constexpr size_t sg_heavy_scratch_bytes(int n_items, int n_tiles, int T) {
    return static_cast<size_t>(kSgHeavySlots) * (static_cast<size_t>(n_items) * 4 + static_cast<size_t>(n_tiles) * T);
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

// A user's best columns so far: lane j holds the j-th best (score, layout column); kk entries at most.
struct SgList {
    float ls; int lc; int n; float theta;       // theta: score of the kk-th entry once the list is full, else -inf
};
__device__ __forceinline__ SgList sg_list_empty() {
    SgList L; L.ls = -__builtin_huge_valf(); L.lc = -1; L.n = 0; L.theta = -__builtin_huge_valf(); return L;
}
__device__ __forceinline__ void sg_list_insert(SgList &L, float sc, int col, int kk, unsigned long long kkmask) {
    const float ninf = -__builtin_huge_valf();
    const int lane = lane_id();
    if (L.n >= kk && !(sc > L.theta)) return;
    // entries that stay ahead: a higher score, or the same score and a lower layout column
    const unsigned long long ahead = __ballot(L.ls > sc || (L.ls == sc && L.lc < col)) & kkmask;
    const int pos = static_cast<int>(__builtin_popcountll(ahead));
    if (pos >= kk) return;
    const float us = fr_shift_up(L.ls, ninf);
    const int uc = fr_shift_up(L.lc, -1);
    if (lane > pos) { L.ls = us; L.lc = uc; }
    else if (lane == pos) { L.ls = sc; L.lc = col; }
    L.n = min(L.n + 1, kk);
    L.theta = L.n >= kk ? readlane_f(L.ls, kk - 1) : ninf;
}

// Raw buffer view of a device array (gfx9 resource word 3; offsets beyond `bytes` read as 0): loads then take a lane
// offset register and a scalar offset, no 64-bit address arithmetic per lane.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sg_buffer(const void *p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, static_cast<int>(bytes > 0xfffffffcull ? 0xfffffffcull : bytes),
                                             0x00020000);
}
typedef unsigned int sg_u2 __attribute__((ext_vector_type(2)));

typedef unsigned int sg_u4 __attribute__((ext_vector_type(4)));

// Lane l holds (s, e, x) of one of the user's rows of W (ascending item order across the lanes), e > s where the row has
// a segment in the tile: add x * w into acc[column] segment by segment, in lane order -- a segment never repeats a column
// and the LDS operations of a wave execute in program order, so every column sees its addends in ascending item order.
// GENERIC form (tiles wider than 256 columns): the first 64 records of the next GROUP segments are requested together,
// the updates then go out segment by segment, a long segment 64 records at a time.
// (Measured and dropped: a two-group software pipeline -- loads of group g + 1 in flight while group g updates -- was no
// faster than this: the other waves of the SIMD already fill the waits.)
template <int GROUP>
__device__ __forceinline__ void sg_accumulate(float *acc, __amdgpu_buffer_rsrc_t went, int s, int e, float x) {
    const int lane = lane_id();
    const int lane8 = lane * 8;
    s &= 0x7fffffff; e &= 0x7fffffff;
    unsigned long long live = __ballot(e > s);
    while (live) {
        int ss[GROUP], len[GROUP];
        float xx[GROUP];
        sg_u2 ent[GROUP];
#pragma unroll
        for (int j = 0; j < GROUP; ++j) {
            ss[j] = 0; len[j] = 0; xx[j] = 0.0f;
            ent[j] = sg_u2{0u, 0u};
            if (live) {
                const int q = __builtin_ctzll(live);
                live &= live - 1;
                ss[j] = readlane_i(s, q);
                len[j] = readlane_i(e, q) - ss[j];
                xx[j] = readlane_f(x, q);
                if (lane < len[j]) ent[j] = __builtin_amdgcn_raw_buffer_load_b64(went, lane8, ss[j] * 8, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < GROUP; ++j) {
            if (len[j] == 0) break;
            const float p = xx[j] * __uint_as_float(ent[j].y);
            if (lane < len[j]) acc[ent[j].x] = acc[ent[j].x] + p;
            for (int ob = 64; ob < len[j]; ob += 64) {       // rest of a long segment
                const sg_u2 en = __builtin_amdgcn_raw_buffer_load_b64(went, lane8, (ss[j] + ob) * 8, 0);
                if (ob + lane < len[j]) acc[en.x] = acc[en.x] + xx[j] * __uint_as_float(en.y);
            }
        }
    }
}

// 256-COLUMN TILES (up to 32,768 active columns per shard: every BASELINE shape but the 500k-item one).  The loop above
// spends most of its instructions on scalar bookkeeping (next set bit, mask updates, exec masks, address arithmetic: the
// counters showed as many scalar as vector instructions, and one scalar unit serves the CU's four SIMDs).  Here
//   * the lanes that hold a segment are COMPACTED to lanes 0 .. n-1 once per chunk (three ds_permute: byte offset | dense
//     flag, entries, rating), so that segment k is read with v_readlane at the constant lane k of an unrolled loop;
//   * a segment is at most one step: sparse (<= 64 records, one 8-byte record per lane) or dense (256 floats, 16 bytes per
//     lane, bit 31 of its begin pointer) -- see seg_layout.py;
//   * nothing is exec-masked: lanes beyond a sparse segment's end update a junk slot behind the tile (acc[256 + lane]),
//     slots beyond the chunk's last segment read past the end of the buffer view (no memory access, zeros).
// Per segment that leaves ~12 vector / LDS / memory instructions and a scalar branch (dense or sparse).
__device__ __forceinline__ void sg_accumulate256(float *acc, __amdgpu_buffer_rsrc_t went, int s_raw, int e, float x) {
    const int lane = lane_id();
    const int sb = s_raw & 0x7fffffff;
    const bool has = (e & 0x7fffffff) > sb;
    const unsigned long long live = __ballot(has);
    if (!live) return;
    const int n = static_cast<int>(__builtin_popcountll(live));
    const int dl = (has ? lane_prefix(live) : n + lane_prefix(~live)) << 2;
    const int cs = __builtin_amdgcn_ds_permute(dl, has ? ((sb << 3) | (s_raw & static_cast<int>(0x80000000u))) : 0x7ffffff8);
    const int cl = __builtin_amdgcn_ds_permute(dl, has ? (e & 0x7fffffff) - sb : 0);
    const float cx = __int_as_float(__builtin_amdgcn_ds_permute(dl, __float_as_int(x)));
    vf4 *acc4 = reinterpret_cast<vf4 *>(acc);
    const int lane8 = lane * 8, lane16 = lane * 16;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        if (g * 8 >= n) break;
        sg_u4 ent[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int so = readlane_i(cs, g * 8 + j);
            if (so < 0) ent[j] = __builtin_amdgcn_raw_buffer_load_b128(went, lane16, so & 0x7fffffff, 0);
            else {
                const sg_u2 t = __builtin_amdgcn_raw_buffer_load_b64(went, lane8, so, 0);
                ent[j] = sg_u4{t.x, t.y, 0u, 0u};
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int so = readlane_i(cs, g * 8 + j);
            const float xj = readlane_f(cx, g * 8 + j);
            if (so < 0) {
                vf4 v = acc4[lane];
                v.x = v.x + xj * __uint_as_float(ent[j].x);
                v.y = v.y + xj * __uint_as_float(ent[j].y);
                v.z = v.z + xj * __uint_as_float(ent[j].z);
                v.w = v.w + xj * __uint_as_float(ent[j].w);
                acc4[lane] = v;
            } else {
                const int ln = readlane_i(cl, g * 8 + j);
                const int c = lane < ln ? static_cast<int>(ent[j].x) : 256 + lane;
                acc[c] = acc[c] + xj * __uint_as_float(ent[j].y);
            }
        }
    }
}

// Read the tile back (4 columns per lane and step), zero it, and insert what can still enter the list: non-zero sums
// above max(L.theta, floor) whose flag byte (optional: the user's own columns) is clear.
__device__ __forceinline__ void sg_scan_tile(vf4 *acc4, int T, int t0, SgList &L, int kk, unsigned long long kkmask,
                                             const unsigned char *flags, float floor) {
    const float ninf = -__builtin_huge_valf();
    const int lane = lane_id();
    const vf4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c4 = lane; c4 < (T >> 2); c4 += 64) {
        const vf4 v = acc4[c4];
        acc4[c4] = zero4;
        uint32_t fw = 0u;
        if (flags) fw = *reinterpret_cast<const uint32_t *>(flags + 4 * c4);
        bool h[4];
        if (L.n >= kk) {
            const float th = fmaxf(L.theta, floor);
#pragma unroll
            for (int j = 0; j < 4; ++j) h[j] = v[j] != 0.0f && v[j] > th && !((fw >> (8 * j)) & 0xffu);
        } else {
            // list still filling: at least kk of this step's values are >= the kk-th largest lane maximum, so nothing
            // below it can end up among the best kk
            float lb = ninf;
#pragma unroll
            for (int j = 0; j < 4; ++j) lb = (v[j] != 0.0f && v[j] > lb && !((fw >> (8 * j)) & 0xffu)) ? v[j] : lb;
            const float cut = sg_kth_lane_best(lb, kk);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                h[j] = v[j] != 0.0f && v[j] > floor && v[j] >= cut && !((fw >> (8 * j)) & 0xffu);       // floor >= -inf: -inf never passes
        }
        if (!__ballot(h[0] || h[1] || h[2] || h[3])) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned long long mj = __ballot(h[j]);
            while (mj) {
                const int q = __builtin_ctzll(mj);
                mj &= mj - 1;
                sg_list_insert(L, readlane_f(v[j], q), t0 + 4 * (c4 - lane + q) + j, kk, kkmask);
            }
        }
    }
}

// The list is the answer: ids through col_ids, counts, and the row goes to the exact-tie pass when two of its leading
// entries have equal scores.
__device__ __forceinline__ void sg_emit(const SegArgs &a, const SgList &L, int row) {
    const float ninf = -__builtin_huge_valf();
    const int lane = lane_id();
    const int n_fin = min(L.n, a.top_k);
    if (lane < a.top_k) {
        const long long o = static_cast<long long>(row) * a.top_k + lane;
        const bool ok = lane < n_fin;
        a.out_id[o] = ok ? a.col_ids[L.lc] : -1;
        a.out_score[o] = ok ? L.ls : ninf;
        if (a.out_aux) a.out_aux[o] = 0u;
    }
    const float below = fr_shift_down(L.ls, ninf);
    const unsigned long long tie = __ballot(lane + 1 < L.n && L.ls == below);
    // DENSE mode (every column competes, zeros included): the list is the answer only when its leading top_k scores are all
    // positive -- positives outrank every zero-score column, zeros outrank negatives; any other row is the caller's to re-score
    const bool short_of_positives = a.dense_rule &&
        static_cast<int>(__builtin_popcountll(__ballot(lane < n_fin && L.ls > 0.0f))) < a.top_k;
    if (lane == 0) {
        a.out_cnt[row] = n_fin;
        if (tie || short_of_positives) a.flag_list[atomicAdd(a.flag_len, 1)] = row;
    }
}

// B0 / B1 += |x| * bound[r][2 lane], [2 lane + 1] for every lane of the chunk that holds a row r >= 0 (eight bound rows in flight)
__device__ __forceinline__ void sg_add_bounds(__amdgpu_buffer_rsrc_t bound, int r, float x, float &B0, float &B1) {
    const int lane4 = lane_id() * 4;
    unsigned long long m = __ballot(r >= 0);
    while (m) {
#ifndef SG_NB
#define SG_NB 8
#endif
        constexpr int NB = SG_NB;
        int rq[NB];
        float aq[NB];
        uint32_t bq[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            rq[j] = -1; aq[j] = 0.0f;
            if (m) {
                const int q = __builtin_ctzll(m);
                m &= m - 1;
                rq[j] = readlane_i(r, q);
                aq[j] = fabsf(readlane_f(x, q));
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) bq[j] = rq[j] >= 0 ? __builtin_amdgcn_raw_buffer_load_b32(bound, lane4, rq[j] * 256, 0) : 0u;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            B0 = __builtin_fmaf(aq[j], __uint_as_float(bq[j] << 16), B0);
            B1 = __builtin_fmaf(aq[j], __uint_as_float(bq[j] & 0xffff0000u), B1);
        }
    }
}

// The tile with the largest remaining bound: returns it (uniform) and retires it from B0 / B1; bmax <= 0: none left.
__device__ __forceinline__ int sg_next_tile(float &B0, float &B1, float &bmax) {
    const int lane = lane_id();
    const float bm = fmaxf(B0, B1);
    bmax = sg_wave_max(bm);
    if (!(bmax > 0.0f)) return -1;
    const int ql = static_cast<int>(__builtin_ctzll(__ballot(bm == bmax)));
    const int which = readlane_f(B0, ql) == bmax ? 0 : 1;
    if (lane == ql) { if (which == 0) B0 = 0.0f; else B1 = 0.0f; }
    return 2 * ql + which;
}

// Diagnostic build (-DSCORE_PROFILE): shader-clock totals per phase of score_seg_kernel, summed over all waves
// (rtrec_amd_seg_profile; not part of the release ABI).
#ifdef SCORE_PROFILE
enum { SP_JOBS, SP_CLAIM, SP_ITEMS, SP_BOUNDS, SP_NEXT, SP_FILTER, SP_SEGPTR, SP_ACC, SP_SCAN, SP_EMIT, SP_N_TILES, SP_N_SEGS, SP_TOTAL, SP_COUNT };
__device__ unsigned long long g_seg_prof[16];
#define SP_DECL unsigned long long sp_[16] = {0}; unsigned long long sp_t_ = __builtin_amdgcn_s_memtime(); const unsigned long long sp_start_ = sp_t_;
#define SP_MARK(slot) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); sp_[slot] += n_ - sp_t_; sp_t_ = n_; }
#define SP_ADD(slot, v) { sp_[slot] += (v); }
#define SP_FLUSH if (lane == 0) { sp_[SP_TOTAL] = __builtin_amdgcn_s_memtime() - sp_start_; for (int q_ = 0; q_ < SP_COUNT; ++q_) atomicAdd(&g_seg_prof[q_], sp_[q_]); }
// workgroup-per-user kernel: wave-clock totals per phase, summed over all waves (rtrec_amd_seg_heavy_profile)
enum { HP_USERS, HP_SETUP, HP_SYNC1, HP_NEXT, HP_TROW, HP_ACC, HP_SCAN, HP_SYNC2, HP_MERGE, HP_CLEAN, HP_TILES, HP_TOTAL, HP_COUNT };
__device__ unsigned long long g_heavy_prof[16];
#define HP_DECL unsigned long long hp_[16] = {0}; unsigned long long hp_t_ = __builtin_amdgcn_s_memtime(); const unsigned long long hp_start_ = hp_t_;
#define HP_MARK(slot) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); hp_[slot] += n_ - hp_t_; hp_t_ = n_; }
#define HP_ADD(slot, v) { hp_[slot] += (v); }
#define HP_FLUSH if (lane == 0) { hp_[HP_TOTAL] = __builtin_amdgcn_s_memtime() - hp_start_; for (int q_ = 0; q_ < HP_COUNT; ++q_) atomicAdd(&g_heavy_prof[q_], hp_[q_]); }
#else
#define SP_DECL
#define SP_MARK(slot)
#define SP_ADD(slot, v)
#define SP_FLUSH
#define HP_DECL
#define HP_MARK(slot)
#define HP_ADD(slot, v)
#define HP_FLUSH
#endif

template <typename IDX> struct SgNone;
template <> struct SgNone<uint16_t> { static constexpr int value = 0xffff; };
template <> struct SgNone<int> { static constexpr int value = -1; };

template <int GROUP, typename IDX, bool T256>
__global__ __launch_bounds__(kSgWaves * 64, SG_OCC) void score_seg_kernel(SegArgs a) {       // 7 waves per SIMD: <= 72 VGPRs
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = static_cast<int>(threadIdx.x) >> 6;
    const int T = a.T;
    unsigned char *wbase = smem + static_cast<size_t>(wave) * sg_wave_lds(T, sizeof(IDX));
    float *acc = reinterpret_cast<float *>(wbase);
    vf4 *acc4 = reinterpret_cast<vf4 *>(wbase);
    float *rx = reinterpret_cast<float *>(wbase + static_cast<size_t>(T + 64) * 4);  // rating of the j-th item that has a row in W
    IDX *lcl = reinterpret_cast<IDX *>(rx + kSgCap);                                  // layout column of the user's j-th item
    IDX *rr = lcl + kSgCap;                                                           // row of W of the j-th item that has one
    constexpr int none = SgNone<IDX>::value;
    const float ninf = -__builtin_huge_valf();
    const vf4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int kk = a.kk;
    const unsigned long long kkmask = kk >= 64 ? ~0ull : ((1ull << kk) - 1ull);
    for (int c = lane; c < (T >> 2); c += 64) acc4[c] = zero4;
    const __amdgpu_buffer_rsrc_t went = sg_buffer(a.w_ent, static_cast<size_t>(a.nnz) * 8);
    const __amdgpu_buffer_rsrc_t bnd = sg_buffer(a.bound, static_cast<size_t>(a.R) * 256);

    SP_DECL
    int w_next = 0, w_end = 0;
    int *claimed = reinterpret_cast<int *>(rr + kSgCap);       // [3][kSgQueueChunk]: output row, first entry, length of the chunk's users
    for (;;) {
        if (w_next >= w_end) {
            int c0 = 0;
            if (lane == 0) c0 = atomicAdd(a.queue, 1);
            const int c = readfirst_i(c0);
            if (c >= a.n_claims) break;
            // A claim's users are STRIDED through the work order -- positions c, c + n_claims, ...: with the order longest
            // first, one user of every length quantile.  (Consecutive positions gave a wave four of the longest users in a
            // row: the pass then lasted as long as those four -- 909 users of 257..512 items among 127,723 shorter ones took
            // it from 1.11 to 1.53 ms, tools/seg_class_probe.py.)
            // the chunk's row pointers in one go: three dependent loads per chunk instead of three per user
            const int pos = c + lane * a.n_claims;
            const bool ok = lane < a.chunk && pos < a.n_rows;
            if (ok) {
                const int c_row = a.order ? a.order[pos] : pos;
                const int xr = a.row_ids ? a.row_ids[c_row] : c_row;
                int c_a0 = 0, c_na = 0;
                if (xr >= 0 && xr < a.n_x_rows) { c_a0 = a.xb_ptr[xr]; c_na = a.xb_ptr[xr + 1] - c_a0; }      // anything else: an empty row
                claimed[lane] = c_row; claimed[kSgQueueChunk + lane] = c_a0; claimed[2 * kSgQueueChunk + lane] = c_na;
            }
            w_next = 0;
            w_end = static_cast<int>(__builtin_popcountll(__ballot(ok)));      // (the valid positions are the leading lanes)
        }
        const int p = w_next++;
        const int row = readfirst_i(claimed[p]);
        const int a0 = readfirst_i(claimed[kSgQueueChunk + p]);
        const int n_a = readfirst_i(claimed[2 * kSgQueueChunk + p]);
        const bool in_lds = n_a <= kSgCap;
        if (a.xs && n_a > a.heavy_min) continue;          // a long user: score_seg_heavy_kernel takes it, one workgroup per user
        SP_MARK(SP_CLAIM) SP_ADD(SP_JOBS, 1)

        // ---- 1. setup: rows of W the user rates, layout columns of its items, per-tile score bounds
        float B0 = 0.0f, B1 = 0.0f;
        int n_r = 0;
        for (int base = 0; base < n_a; base += 64) {
            const int idx = base + lane;
            int r = -1, lc = -1;
            float x = 0.0f;
            if (idx < n_a) {
#if SG_NT_USER_ROWS       // the user-row stream is read once per pass: keep it from evicting W's records and tables out of L2
                const int item = __builtin_nontemporal_load(&a.xb_col[a0 + idx]);
                x = __builtin_nontemporal_load(&a.xb_val[a0 + idx]);
#else
                const int item = a.xb_col[a0 + idx];
                x = a.xb_val[a0 + idx];
#endif
                if (item < a.n_items) { const int2 f = a.info[item]; r = f.x; lc = f.y; }      // items newer than W have neither
            }
            if (in_lds && idx < n_a) lcl[idx] = static_cast<IDX>(lc < 0 ? none : lc);
            const unsigned long long m = __ballot(r >= 0);
            if (in_lds && r >= 0) {
                const int pos = n_r + lane_prefix(m);
                rr[pos] = static_cast<IDX>(r);
                rx[pos] = x;
            }
            n_r += static_cast<int>(__builtin_popcountll(m));
            SP_MARK(SP_ITEMS)
            sg_add_bounds(bnd, r, x, B0, B1);
            SP_MARK(SP_BOUNDS)
        }
        // without the heavy pass a long user's items are re-read from global per tile (slow, same answer)
        const int n_ch = in_lds ? (n_r + 63) >> 6 : (n_a + 63) >> 6;
        // a computed score is a float32 sum of rounded products, the bound a float32 sum of |x| * (rounded-up max):
        // score <= B * (1 + (2 n + 2) 2^-24) with n <= n_r addends
        const float slack = 1.0f + 1e-5f + static_cast<float>(n_r) * 1.3e-7f;

        // ---- 2. tiles by descending bound
        SgList L = sg_list_empty();
        for (;;) {
            float bmax;
            const int t = sg_next_tile(B0, B1, bmax);
            if (t < 0) break;
            if (L.n >= kk && bmax * slack < L.theta) break;
            const int t0 = t * T;
            const int ncol = min(T, a.n_cols - t0);
            SP_MARK(SP_NEXT) SP_ADD(SP_N_TILES, 1)

            if (a.filter) {         // the user's own columns leave the race: -inf + p = -inf
                for (int base = 0; base < n_a; base += 64) {
                    const int idx = base + lane;
                    int lc = -1;
                    if (idx < n_a) {
                        if (in_lds) { lc = lcl[idx]; lc = lc == none ? -1 : lc; }
                        else { const int item = a.xb_col[a0 + idx]; if (item < a.n_items) lc = a.info[item].y; }
                    }
                    lc = lc < 0 ? -1 : lc - t0;
                    if (lc >= 0 && lc < ncol) acc[lc] = ninf;
                }
            }
            SP_MARK(SP_FILTER)
            // (measured and dropped: gathering the segment pointers of two chunks per step -- the second chunk's registers cost
            // a wave per SIMD or spills, and users with several chunks got slower, not faster)
            // (also measured and dropped, round 3 late: requesting chunk ch + 1's pointers before chunk ch's records, so that
            // the two gathers of a step overlap -- 4-8 spilled VGPRs at 7 waves per SIMD (2.06-2.09 ms per c3s pass against
            // 1.95), and no better without spills at 6 waves per SIMD (1.99 ms): occupancy hides those round trips already)
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
                SP_MARK(SP_SEGPTR) SP_ADD(SP_N_SEGS, __builtin_popcountll(__ballot((e & 0x7fffffff) > (s & 0x7fffffff))))
                if constexpr (T256) sg_accumulate256(acc, went, s, e, x);
                else sg_accumulate<GROUP>(acc, went, s, e, x);
                SP_MARK(SP_ACC)
            }
            sg_scan_tile(acc4, T, t0, L, kk, kkmask, nullptr, ninf);
            SP_MARK(SP_SCAN)
        }
        SP_MARK(SP_NEXT)
        sg_emit(a, L, row);
        SP_MARK(SP_EMIT)
    }
    SP_FLUSH
}

// ---- heavy pass: one workgroup per long user ---------------------------------------------------------------------
// A user with thousands of items would keep one wave busy for milliseconds (every opened tile costs a pass over its item
// list).  Here the user's ratings are scattered once into a dense per-workgroup scratch xs[item]; a tile is then walked
// from the TILE's side -- trow lists the rows of W that have a segment in it (ascending item), xs[item] != 0 says the
// user rates it -- the tiles are dealt to the workgroup's waves in descending bound order, every wave keeps a list of its
// own (the best (k+1)-th score any wave has reached is shared through LDS and prunes for all), and wave 0 merges the
// lists.  The user's own columns are flag bytes fl[layout column], tested when a tile is read back.
//
// LIST PATH (round 4): a user of up to kSgCap items keeps its items in LDS lists like the wave-per-user kernel -- no
// scatter into xs, no flag bytes, nothing to restore -- but the workgroup's waves share them: the setup chunks (items ->
// row of W / layout column, bound rows) are dealt to the waves and an opened tile is accumulated by ONE wave from the
// user's side (its rows in ascending item order: scipy's order, bit-identical sums), the tiles dealt to the waves in
// descending bound order.  Why: a full pass is as long as its longest single-wave user -- a 257..512-item user is a chain
// of ~200 dependent L2 round trips (15 opened tiles x (segment pointers + records) per 64-row chunk), ~1.7 ms under load;
// the 127,723 users of up to 256 items take 1.11 ms, adding 909 users of 257..512 items made it 1.53 ms, all 9,097 of
// them 1.72 ms (tools/seg_class_probe.py).  With the workgroup the chain is the setup / 8 + two tiles.
// (measured, round 4: compiled for 8 waves per SIMD -- 64 VGPRs, 12 spilled, all 1024 workgroups resident -- it is no faster:
// 0.62 -> 0.67 ms for the 1,673 long users of c3s alone)
__global__ __launch_bounds__(1024) void score_seg_heavy_kernel(SegArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id();
    const int wave = static_cast<int>(threadIdx.x) >> 6;
    const int nw = static_cast<int>(blockDim.x) >> 6;
    const int T = a.T;
    const float ninf = -__builtin_huge_valf();
    float *acc = reinterpret_cast<float *>(smem + static_cast<size_t>(wave) * (T + 64) * 4);
    vf4 *acc4 = reinterpret_cast<vf4 *>(acc);
    float *bred = reinterpret_cast<float *>(smem + static_cast<size_t>(nw) * (T + 64) * 4);      // [nw][128] partial bounds
    float *lst_s = bred + nw * 128;                                                         // [nw][64] the waves' lists
    int *lst_c = reinterpret_cast<int *>(lst_s + nw * 64);
    int *shared = lst_c + nw * 64;                  // [0]: bits of the best full-list theta (> 0) any wave has reached
    float *t_rx = reinterpret_cast<float *>(shared + 16);      // list path: rating, layout column (-1: none), row of W (-1: none)
    int *t_lcl = reinterpret_cast<int *>(t_rx + kSgCap);       // of the user's idx-th item, shared by the waves
    int *t_rr = t_lcl + kSgCap;
    const vf4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const int kk = a.kk;
    const unsigned long long kkmask = kk >= 64 ? ~0ull : ((1ull << kk) - 1ull);
    for (int c = lane; c < (T >> 2); c += 64) acc4[c] = zero4;
    const __amdgpu_buffer_rsrc_t went = sg_buffer(a.w_ent, static_cast<size_t>(a.nnz) * 8);
    const __amdgpu_buffer_rsrc_t bnd = sg_buffer(a.bound, static_cast<size_t>(a.R) * 256);
    float *xs = a.xs + static_cast<size_t>(blockIdx.x) * a.n_items;
    unsigned char *fl = a.fl + static_cast<size_t>(blockIdx.x) * a.n_tiles * T;
    const int stride = static_cast<int>(blockDim.x);
    HP_DECL

    // the long users are found by their length: with a longest-first work order they are its head and a workgroup
    // leaves at the first user that is not long; without one every workgroup walks its share of the rows
    for (int u = blockIdx.x; u < a.n_rows; u += gridDim.x) {
        const int row = a.order ? a.order[u] : u;
        const int xrow = a.row_ids ? a.row_ids[row] : row;
        const bool xok = xrow >= 0 && xrow < a.n_x_rows;
        const int a0 = readfirst_i(xok ? a.xb_ptr[xrow] : 0);
        const int n_a = readfirst_i(xok ? a.xb_ptr[xrow + 1] - a0 : 0);
        if (n_a <= a.heavy_min) {
            if (a.order_longest_first) break;
            continue;
        }
        if (threadIdx.x == 0) shared[0] = 0;
        HP_MARK(HP_NEXT) HP_ADD(HP_USERS, wave == 0 ? 1 : 0)
        const bool lists = n_a <= kSgCap;           // list path (uniform)
        // ---- 1. scatter the ratings / flags (list path: fill the LDS lists), partial bounds per wave
        float B0 = 0.0f, B1 = 0.0f;
        for (int base = wave * 64; base < n_a; base += stride) {
            const int idx = base + lane;
            int r = -1, lc = -1;
            float x = 0.0f;
            if (idx < n_a) {
                const int item = a.xb_col[a0 + idx];
                x = a.xb_val[a0 + idx];
                if (item < a.n_items) {
                    const int2 f = a.info[item];
                    r = f.x; lc = f.y;
                    if (!lists) {
                        if (r >= 0) xs[item] = x;
                        if (a.filter && f.y >= 0) fl[f.y] = 1;
                    }
                }
                if (lists) { t_rx[idx] = x; t_lcl[idx] = lc; t_rr[idx] = r; }
            }
            sg_add_bounds(bnd, r, x, B0, B1);
        }
        bred[wave * 128 + 2 * lane] = B0;
        bred[wave * 128 + 2 * lane + 1] = B1;
        HP_MARK(HP_SETUP)
        __syncthreads();            // xs / fl / bred of all waves are visible (one CU, one L1)
        HP_MARK(HP_SYNC1)
        B0 = 0.0f; B1 = 0.0f;
        for (int w = 0; w < nw; ++w) { B0 += bred[w * 128 + 2 * lane]; B1 += bred[w * 128 + 2 * lane + 1]; }
        const float slack = 1.0f + 1e-5f + static_cast<float>(n_a) * 1.3e-7f;

        // ---- 2. tiles in descending bound order, dealt to the waves
        SgList L = sg_list_empty();
        int rank = 0;
        for (;;) {
            float bmax;
            const int t = sg_next_tile(B0, B1, bmax);
            if (t < 0) break;
            const int sb = *reinterpret_cast<volatile int *>(shared);
            const float floor = sb ? __int_as_float(sb) : ninf;
            const float th = fmaxf(L.theta, floor);
            if (th > ninf && bmax * slack < th) break;
            if ((rank++ % nw) != wave) continue;
            const int t0 = t * T;
            HP_MARK(HP_NEXT) HP_ADD(HP_TILES, 1)
            if (lists) {
                // the user's side: its own columns leave the race (-inf absorbs), then its rows of W in ascending item order
                const int ncol = min(T, a.n_cols - t0);
                if (a.filter) {
                    for (int idx = lane; idx < n_a; idx += 64) {
                        const int lc = t_lcl[idx] < 0 ? -1 : t_lcl[idx] - t0;
                        if (lc >= 0 && lc < ncol) acc[lc] = ninf;
                    }
                }
                for (int idx = lane; idx - lane < n_a; idx += 64) {
                    int s = 0, e = 0;
                    float x = 0.0f;
                    const int r = idx < n_a ? t_rr[idx] : -1;
                    if (r >= 0) {
                        x = t_rx[idx];
                        const int *pp = a.seg_ptr + static_cast<size_t>(r) * (a.n_tiles + 1) + t;
                        s = pp[0]; e = pp[1];
                    }
                    if (T == 256) sg_accumulate256(acc, went, s, e, x);
                    else sg_accumulate<8>(acc, went, s, e, x);
                }
                sg_scan_tile(acc4, T, t0, L, kk, kkmask, nullptr, floor);
            } else {
            const int i1 = a.trow_ptr[t + 1];
            for (int i0 = a.trow_ptr[t]; i0 < i1; i0 += 64) {
                const int i = i0 + lane;
                int s = 0, e = 0;
                float x = 0.0f;
                if (i < i1) {
                    const int4 rec = a.trow[i];
                    x = xs[rec.x];
                    if (x != 0.0f) { s = rec.y; e = rec.z; }
                }
                HP_MARK(HP_TROW)
                if (T == 256) sg_accumulate256(acc, went, s, e, x);
                else sg_accumulate<8>(acc, went, s, e, x);
                HP_MARK(HP_ACC)
            }
            sg_scan_tile(acc4, T, t0, L, kk, kkmask, a.filter ? fl + t0 : nullptr, floor);
            HP_MARK(HP_SCAN)
            }
            if (L.n >= kk && L.theta > 0.0f && lane == 0) atomicMax(shared, __float_as_int(L.theta));
        }
        // ---- 3. merge the waves' lists in wave 0, emit
        lst_s[wave * 64 + lane] = (lane < L.n) ? L.ls : ninf;
        lst_c[wave * 64 + lane] = L.lc;
        HP_MARK(HP_NEXT)
        __syncthreads();
        HP_MARK(HP_SYNC2)
        if (wave == 0) {
            for (int w = 1; w < nw; ++w) {
                const float os = lst_s[w * 64 + lane];
                const int oc = lst_c[w * 64 + lane];
                unsigned long long mm = __ballot(os > ninf) & kkmask;
                while (mm) {
                    const int q = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    sg_list_insert(L, readlane_f(os, q), readlane_i(oc, q), kk, kkmask);
                }
            }
            sg_emit(a, L, row);
        }
        HP_MARK(HP_MERGE)
        // ---- 4. restore the scratch invariants (all zero; the list path has touched none)
        for (int base = wave * 64; base < n_a && !lists; base += stride) {
            const int idx = base + lane;
            if (idx < n_a) {
                const int item = a.xb_col[a0 + idx];
                if (item < a.n_items) {
                    const int2 f = a.info[item];
                    if (f.x >= 0) xs[item] = 0.0f;
                    if (a.filter && f.y >= 0) fl[f.y] = 0;
                }
            }
        }
        __syncthreads();
        HP_MARK(HP_CLEAN)
    }
    HP_FLUSH
}
