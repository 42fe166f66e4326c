// rtrec_amd/csrc/fit.hip -- per-item-column SLIM fit: X^T y feature selection + elastic-net
// coordinate descent, in the reference's exact float32 operation order.
//
// Replaces (reference): the per-column loop of SLIMElastic.fit / fit_in_parallel /
// partial_fit_items (slim_elastic.py:261-277, 434-447, 544-560), FeatureSelectionWrapper.fit
// (slim_elastic.py:139-154) and scikit-learn's sparse_enet_coordinate_descent
// (sklearn/linear_model/_cd_fast.pyx:276-561, float32, selection='random').
//
// One wavefront fits one target column j at a time, pulled from a device work queue:
//   1. s = X^T y over the co-occurring rows (exactly scipy csr_matvec's per-item order: rows
//      ascending), with a touched list so only the neighbourhood of j is ever visited;
//   2. top-K of s (descending, ties -> higher id), K = nn_feature_selection;
//   3. coordinate descent with sklearn's xorshift32 coordinate sequence.  The dot product
//      tmp = sum_r R[r] * x[r] is the order-sensitive step: the 64 lanes gather and multiply in
//      parallel, then the products are folded strictly left to right (chain_add) so tmp, hence
//      every coefficient and the sweep count, is bit-identical to the Cython loop.  Residual
//      updates are element-wise and stay fully parallel.
// While every coefficient is still zero (R == y) the dot product of feature p equals s[p]
// bit for bit (same products, same order), so targets whose solution is all-zero -- the long
// tail of a Zipf catalogue -- never touch the O(nnz) path at all.
//
// Scratch invariants between targets (set by rtrec_slim_fit_workspace_init): R == 0, s ==
// kUntouched, w_all == 0, ever_flag == 0.
#include "common.hip.h"
#include <type_traits>
#include "fold_spec.hip.h"
#include "../../include/rtrec_amd.h"

namespace rtrec {

constexpr int kRowUnroll = 8;   // 64-item chunks of one user row updated together in the X^T y step

#ifndef FIT_WAVES_PER_SIMD
#define FIT_WAVES_PER_SIMD 5
#endif

// -0.0f can never be produced by (+0) + p or by a float sum that starts at +0 under
// round-to-nearest, so its bit pattern marks "no contribution yet".
constexpr uint32_t kUntouched = 0x80000000u;

struct FitArgs {
    int U, I;
    const int *cptr; const int *crow; const float *cval;
    const int *rptr; const int *rcol; const float *rval;
    const float *sqn;
    const int *col_order;   // optional: item ids by descending column length (column-walk balance)
    int colwalk_min_rows;   // latency mode: targets with at least this many users use the column walk
    int screen_min;         // columns with at least this many entries are screened before an ordered fold
    int lane_max;           // every-item path: columns up to this length are folded one per lane
    const int *targets; int n_targets;
    rtrec_fit_cfg cfg;
    int *out_items; float *out_coef; int *out_count; int *out_niter; int cap;
    float *R;        // [slots][U]
    float *stash;    // [slots][U]   latency mode: the add-back values of the column folded last (mw_fold -> mw_update_stashed)
    float *s;        // [slots][I]
    int *touched;    // [slots][I]
    float *cand_s;   // [slots][I]   K path: candidate scores; ALL path: ever_flag (as int)
    int *cand_i;     // [slots][I]   K path: candidate ids;    ALL path: ever list
    float *w_all;    // [slots][I]   ALL path only
    int *long_list;  // [slots][I]   ALL path only: columns longer than lane_max, per duality-gap evaluation
    int *queue;
    long long *trace;   // optional [n_targets][8]: start, prep end, end (100 MHz ticks), folded entries, 4 kernel-specific phase counters
    const double *gram;       // optional [gram_n][gram_n]: X_p . X_q of the gram_n tracked (popular) items
    const int *gram_index;    // optional [I]: item -> row of `gram`, or -1
    int gram_n;
    double gram_rel_err;      // |gram - exact| <= gram_rel_err * exact
    int fast;                 // tolerance mode (rtrec_fit_opts.fast): 1 tree-reduced dots, 2 also Gram-form CD (fit_gram_cd)
    // optional, latency mode: the non-zero entries of X^T y of every target of the call, computed beforehand by
    // xty_batch_kernel (one pass over X for the whole call); target t owns [t * I, t * I + pre_cnt[t])
    const int *pre_cnt; const int *pre_i; const float *pre_s;
    int spec_min;             // columns of at least this many entries are folded by fold256_spec (csrc/fold_spec.hip.h); INT_MAX: never
};

__device__ __forceinline__ float cd_update(float tmp, float alpha, float beta, float nrm, int positive) {
    // _cd_fast.pyx:471-475; libc fabs() is double, so the numerator and the quotient are
    // formed in double and rounded to float once.
    if (positive && tmp < 0.0f) return 0.0f;
    const double num = fabs(static_cast<double>(tmp)) - static_cast<double>(alpha);
    const double sgn = (tmp == 0.0f) ? 0.0 : (tmp > 0.0f ? 1.0 : -1.0);
    const float den = __fadd_rn(nrm, beta);
    return static_cast<float>(sgn * (num > 0.0 ? num : 0.0) / static_cast<double>(den));
}

// Ordered fold of 64 products (one per lane) through LDS: one ds_write_b32, then 16 uniform-address
// ds_read_b128 (LDS broadcast) feeding the 64 dependent v_add_f32 -- ~1.3 instructions per entry where
// v_readlane + v_add needs 2.  `buf` is a wave-private slice of 64 floats; the LDS operations of one
// wave execute in program order, so no barrier is involved.  Lanes beyond the data hold +0.0, which
// never changes the running sum (it starts at +0 and a float sum cannot become -0 again).
constexpr int kFoldBufBytes = 4 * 64 * 4;   // four slices: the four chunks of one dot_pass batch
__device__ __forceinline__ float fold64_lds(float acc, const float *buf, int n4 = 16) {
    const float4 *b4 = reinterpret_cast<const float4 *>(buf);
    if (n4 >= 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 v = b4[k];
            acc = __fadd_rn(acc, v.x); acc = __fadd_rn(acc, v.y); acc = __fadd_rn(acc, v.z); acc = __fadd_rn(acc, v.w);
        }
        return acc;
    }
    for (int k = 0; k < n4; ++k) {
        const float4 v = b4[k];
        acc = __fadd_rn(acc, v.x); acc = __fadd_rn(acc, v.y); acc = __fadd_rn(acc, v.z); acc = __fadd_rn(acc, v.w);
    }
    return acc;
}

// tmp = sum over column entries [b, e) of (R[r] (+ x*w_old)) * x, strictly left to right.
// Software-pipelined over batches of 4 x 64 entries: while batch k is folded (256 dependent adds),
// the R gathers of batch k+1 and the index/value loads of batch k+2 are already in flight, so the
// wave is bound by the fold itself instead of by two dependent memory round trips per batch.
__device__ float dot_pass(const int *__restrict__ crow, const float *__restrict__ cval, const float *R,
                          int b, int e, float w_old, float *fold_buf) {
    const int lane = lane_id();
    const bool add_back = (w_old != 0.0f);
    float tmp = 0.0f;
    const int n_full = (e - b) >> 8;              // batches of 256 entries
    int o = b;
    if (n_full > 0) {
        int ra[4], rb[4];
        float xa[4], xb[4], va[4];
        // prologue: indices of batch 0 -> gathers of batch 0; indices of batch 1
#pragma unroll
        for (int u = 0; u < 4; ++u) { ra[u] = crow[o + u * 64 + lane]; xa[u] = cval[o + u * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) va[u] = R[ra[u]];
        if (n_full > 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { rb[u] = crow[o + 256 + u * 64 + lane]; xb[u] = cval[o + 256 + u * 64 + lane]; }
        }
        for (int k = 0; k < n_full; ++k) {
            float vn[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            int rc[4] = {0, 0, 0, 0};
            float xc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (k + 1 < n_full) {              // gathers of batch k+1 (its indices arrived during batch k-1)
#pragma unroll
                for (int u = 0; u < 4; ++u) vn[u] = R[rb[u]];
            }
            if (k + 2 < n_full) {              // indices of batch k+2
                const int o2 = o + 512;
#pragma unroll
                for (int u = 0; u < 4; ++u) { rc[u] = crow[o2 + u * 64 + lane]; xc[u] = cval[o2 + u * 64 + lane]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float v = va[u];
                if (add_back) v = __fadd_rn(v, __fmul_rn(xa[u], w_old));
                fold_buf[u * 64 + lane] = __fmul_rn(v, xa[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) tmp = fold64_lds(tmp, fold_buf + u * 64);
#pragma unroll
            for (int u = 0; u < 4; ++u) { va[u] = vn[u]; xa[u] = xb[u]; rb[u] = rc[u]; xb[u] = xc[u]; }
            o += 256;
        }
    }
    for (; o < e; o += 64) {
        const int n = min(64, e - o);
        float prod = 0.0f;
        if (lane < n) {
            const int r = crow[o + lane];
            const float x = cval[o + lane];
            float v = R[r];
            if (add_back) v = __fadd_rn(v, __fmul_rn(x, w_old));
            prod = __fmul_rn(v, x);
        }
        fold_buf[lane] = prod;                       // lanes >= n hold +0.0
        tmp = fold64_lds(tmp, fold_buf, (n + 3) >> 2);
    }
    return tmp;
}

// The same sum by integer prefix sums inside a binade (csrc/fold_spec.hip.h) -- bit-identical to dot_pass, without the
// chain of dependent additions.  Lane L takes entries 4L .. 4L+3 of a 256-entry group (one 16-byte load of the row ids
// and one of the values, 4-byte aligned), so the products arrive in the layout fold256_spec scans; same software
// pipeline as dot_pass (gathers of group k+1 and the index loads of group k+2 in flight while group k is folded).
struct __attribute__((packed, aligned(4))) PackedI4 { int x, y, z, w; };
struct __attribute__((packed, aligned(4))) PackedF4 { float x, y, z, w; };

__device__ float dot_pass_spec(const int *__restrict__ crow, const float *__restrict__ cval, const float *R,
                               int b, int e, float w_old) {
    const int lane = lane_id();
    const bool add_back = (w_old != 0.0f);
    float tmp = 0.0f;
    const int n_full = (e - b) >> 8;
    int o = b;
    auto prods = [&](const PackedF4 &v, const PackedF4 &x, float (&p)[4]) {
        const float vv[4] = {v.x, v.y, v.z, v.w}, xx[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t = vv[u];
            if (add_back) t = __fadd_rn(t, __fmul_rn(xx[u], w_old));
            p[u] = __fmul_rn(t, xx[u]);
        }
    };
    if (n_full > 0) {
        PackedI4 ra, rb{0, 0, 0, 0};
        PackedF4 xa, xb{0.0f, 0.0f, 0.0f, 0.0f}, va;
        ra = *reinterpret_cast<const PackedI4 *>(crow + o + 4 * lane);
        xa = *reinterpret_cast<const PackedF4 *>(cval + o + 4 * lane);
        va.x = R[ra.x]; va.y = R[ra.y]; va.z = R[ra.z]; va.w = R[ra.w];
        if (n_full > 1) {
            rb = *reinterpret_cast<const PackedI4 *>(crow + o + 256 + 4 * lane);
            xb = *reinterpret_cast<const PackedF4 *>(cval + o + 256 + 4 * lane);
        }
        for (int k = 0; k < n_full; ++k) {
            PackedF4 vn{0.0f, 0.0f, 0.0f, 0.0f}, xc{0.0f, 0.0f, 0.0f, 0.0f};
            PackedI4 rc{0, 0, 0, 0};
            if (k + 1 < n_full) { vn.x = R[rb.x]; vn.y = R[rb.y]; vn.z = R[rb.z]; vn.w = R[rb.w]; }
            if (k + 2 < n_full) {
                rc = *reinterpret_cast<const PackedI4 *>(crow + o + 512 + 4 * lane);
                xc = *reinterpret_cast<const PackedF4 *>(cval + o + 512 + 4 * lane);
            }
            float p[4];
            prods(va, xa, p);
            tmp = fold256_spec(tmp, p[0], p[1], p[2], p[3]);
            va = vn; xa = xb; rb = rc; xb = xc;
            o += 256;
        }
    }
    if (o < e) {
        const int n = e - o;
        float p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int oo = 4 * lane + u;
            p[u] = 0.0f;
            if (oo < n) {
                const float x = cval[o + oo];
                float t = R[crow[o + oo]];
                if (add_back) t = __fadd_rn(t, __fmul_rn(x, w_old));
                p[u] = __fmul_rn(t, x);
            }
        }
        tmp = fold256_spec(tmp, p[0], p[1], p[2], p[3], n);
    }
    return tmp;
}

template <bool SPEC>
__device__ __forceinline__ float dot_pass_any(const int *__restrict__ crow, const float *__restrict__ cval, const float *R,
                                              int b, int e, float w_old, float *fold_buf, int spec_min) {
    if (SPEC && e - b >= spec_min) return dot_pass_spec(crow, cval, R, b, e, w_old);
    return dot_pass(crow, cval, R, b, e, w_old, fold_buf);
}

// R[r] <- (R[r] + x*w_old) - x*w_new over the column (element-wise, order free; 4 x 64 entries
// per step so that four gathers are in flight per lane).
__device__ void update_pass(const int *__restrict__ crow, const float *__restrict__ cval, float *R,
                            int b, int e, float w_old, float w_new) {
    const int lane = lane_id();
    int o = b;
    for (; o + 256 <= e; o += 256) {
        int r[4];
        float x[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { r[u] = crow[o + u * 64 + lane]; x[u] = cval[o + u * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = R[r[u]];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (w_old != 0.0f) v[u] = __fadd_rn(v[u], __fmul_rn(x[u], w_old));
            if (w_new != 0.0f) v[u] = __fsub_rn(v[u], __fmul_rn(x[u], w_new));
            R[r[u]] = v[u];
        }
    }
    for (o += lane; o < e; o += 64) {
        const int r = crow[o];
        const float x = cval[o];
        float v = R[r];
        if (w_old != 0.0f) v = __fadd_rn(v, __fmul_rn(x, w_old));
        if (w_new != 0.0f) v = __fsub_rn(v, __fmul_rn(x, w_new));
        R[r] = v;
    }
}

// sum over [b, e) of x * R[r], left to right (XtA of _cd_fast.pyx:506-509).
template <bool SPEC>
__device__ float xta_pass(const int *__restrict__ crow, const float *__restrict__ cval, const float *R, int b, int e,
                          float *fold_buf, int spec_min) {
    return dot_pass_any<SPEC>(crow, cval, R, b, e, 0.0f, fold_buf, spec_min);   // R[r]*x == x*R[r] (one rounding, commutative)
}

// ---------------------------------------------------------------------------------------------
// Screening: an ORDER-FREE evaluation of the same products, with a rigorous bound on how far the
// ordered (left-to-right) float32 sum can be from it.
//
// A coordinate whose coefficient is 0 stays 0 whenever tmp <= alpha (|tmp| <= alpha without
// `positive`), and then nothing else about tmp matters: no residual update, d_w = 0.  Likewise
// the duality gap only needs max_p XtA[p].  So the 64-step dependent fold -- the whole cost of
// this kernel -- is only required for coordinates that are non-zero, close to the threshold, or
// candidates for that maximum; every other column gets one parallel pass (64 lanes, 8 gathers in
// flight, three VALU ops per 64 entries instead of 128).
//
// Bound: both evaluations add the SAME n rounded products p_i.  For any summation order
// |computed - exact| <= gamma_k * S with S = sum |p_i|, gamma_k = k u / (1 - k u), u = 2^-24
// and k the longest chain of additions (n-1 for the ordered sum, at most n/64 + 16 for per-thread
// partial sums + shuffle tree + the 8 wave partials of the multi-wave kernel); float addition is
// exact when the result is subnormal, so there is no underflow term.  With n <= 2^20
// (k u <= 1/16, so gamma_k <= 1.0667 k u) and S <= asum / (1 - gamma_par) <= 1.002 asum:
//   |ordered - parallel| <= 1.0667 * 1.002 * (n - 1 + n/64 + 16) u asum <= (1.0856 n + 17) u asum
//                        <  1.125 (n + 24) u asum =: err.
// The result of a screened decision is therefore bit-identical to the reference's.
// ---------------------------------------------------------------------------------------------
constexpr int kScreenMaxLen = 1 << 20;
constexpr int kSpecMinDefault = 512;     // rtrec_fit_opts.fold = 3: shorter columns keep the chain (the first entries of a fold are serial anyway)
constexpr int kScreenMinDefault = 192;

struct Screen { double lo, hi; };   // the ordered sum lies in [lo, hi]

__device__ __forceinline__ Screen screen_interval(float psum, float pasum, int n) {
    const double err = 1.125 * static_cast<double>(n + 24) * 0x1p-24 * static_cast<double>(pasum);
    Screen r;
    r.lo = static_cast<double>(psum) - err;
    r.hi = static_cast<double>(psum) + err;
    return r;
}

// sum and sum of |.| of R[r]*x[r] over [b, e), any order (one wave).
__device__ void screen_pass(const int *__restrict__ crow, const float *__restrict__ cval, const float *R,
                            int b, int e, float &psum, float &pasum) {
    const int lane = lane_id();
    float s0 = 0.0f, a0 = 0.0f;
    int o = b;
    for (; o + 512 <= e; o += 512) {
        int r[8];
        float x[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { r[u] = crow[o + u * 64 + lane]; x[u] = cval[o + u * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = R[r[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float p = __fmul_rn(v[u], x[u]);
            s0 = __fadd_rn(s0, p);
            a0 = __fadd_rn(a0, fabsf(p));
        }
    }
    for (; o < e; o += 256) {
        int r[4];
        float x[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int oo = o + u * 64 + lane;
            r[u] = -1; x[u] = 0.0f;
            if (oo < e) { r[u] = crow[oo]; x[u] = cval[oo]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (r[u] >= 0) ? R[r[u]] : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float p = __fmul_rn(v[u], x[u]);
            s0 = __fadd_rn(s0, p);
            a0 = __fadd_rn(a0, fabsf(p));
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s0 = __fadd_rn(s0, shfl_xor_t(s0, m));
        a0 = __fadd_rn(a0, shfl_xor_t(a0, m));
    }
    psum = s0; pasum = a0;
}

// Decision for a coordinate whose coefficient is (+-)0: true -> it provably stays zero, w_new set.
__device__ __forceinline__ bool screen_stays_zero(const Screen &S, float alpha, int positive, float &w_new) {
    const double a = static_cast<double>(alpha);
    if (positive) {
        if (S.hi <= a) { w_new = 0.0f; return true; }      // tmp < 0 -> 0; 0 <= tmp <= alpha -> +0
        return false;
    }
    if (S.hi <= a && S.lo >= -a) {                           // |tmp| <= alpha: zero, signed like tmp
        if (S.lo >= 0.0) { w_new = 0.0f; return true; }      // tmp >= 0 -> +0 (sign(0) = 0 -> +0 as well)
        if (S.hi < 0.0) { w_new = -0.0f; return true; }      // tmp < 0 -> -1 * 0 / d = -0
    }
    return false;
}

__device__ __forceinline__ float float_next_down(float f) {   // largest float < f (f finite)
    const uint32_t b = __float_as_uint(f);
    if ((b << 1) == 0u) return __uint_as_float(0x80000001u);
    return __uint_as_float((b >> 31) ? b + 1u : b - 1u);
}
__device__ __forceinline__ float float_next_up(float f) {     // smallest float > f (f finite)
    const uint32_t b = __float_as_uint(f);
    if ((b << 1) == 0u) return __uint_as_float(0x00000001u);
    return __uint_as_float((b >> 31) ? b - 1u : b + 1u);
}

// Interval of fl(xta - beta_w) (or of its absolute value without `positive`) for xta in S, rounded
// outward to float; fl(. - c) and fabs-after-it are monotone, so the ordered value lies inside.
__device__ __forceinline__ void screen_xta_interval(const Screen &S, float bw, int positive, float &lo, float &hi) {
    float lf = static_cast<float>(S.lo);
    if (static_cast<double>(lf) > S.lo) lf = float_next_down(lf);
    float hf = static_cast<float>(S.hi);
    if (static_cast<double>(hf) < S.hi) hf = float_next_up(hf);
    const float vl = __fsub_rn(lf, bw), vh = __fsub_rn(hf, bw);
    if (positive) { lo = vl; hi = vh; return; }
    const float al = fabsf(vl), ah = fabsf(vh);
    hi = al > ah ? al : ah;
    lo = (vl > 0.0f) ? vl : (vh < 0.0f ? -vh : 0.0f);
}

// ---------------------------------------------------------------------------------------------
// Gram tracking (non-negative X, K <= 64): screening WITHOUT touching memory.
//
// For a feature p let D_p = sum_r x_pr R[r] in exact arithmetic over the current float residual.
// A coordinate update of feature q changes R[r] <- fl(fl(R[r] + fl(x_qr w_old)) - fl(x_qr w_new)) =
// R[r] - dw x_qr + eps_r with |eps_r| <= 3.05 u (|R[r]| + x_qr (|w_old| + |w_new|)), hence
//     D_p <- D_p - dw G_pq + sum_r x_pr eps_r,      G_pq = X_p . X_q .
// G is shared by all targets: the host computes it once per fit for the most popular items (the
// features of nearly every target) with a float64 GEMM of the densified columns (products of
// floats are exact in double, so |G^ - G| <= gram_rel_err G with gram_rel_err ~ n 2^-53).
// Lane p keeps  c_p ~ D_p,  r_p >= |c_p - D_p|  and  a_p >= sum_r x_pr (|y_r| + sum_q |w_q| x_qr),
// which bounds A_p = sum_r x_pr |R[r]| up to the rounding already counted in r_p:
//   start (R == y):  c = s_p (the float X^T y entry: the ordered sum of m = |U_p ^ U_j| rounded
//                    products), r = 1.2 (m+1) u s_p, a = s_p (1 + 1.1 (m+1) u);
//   update of q:     c -= dw G^;  a += (|w_new| - |w_old|) G^ (1 +- gram_rel_err);
//                    r += |dw| G^ gram_rel_err' + 3.05 u (a + r + (|w_old| + |w_new|) G^) + double rounding.
// The ordered float32 sum tmp_p of the n_p rounded products then satisfies
//   |tmp_p - c_p| <= r_p + 1.07 (n_p + 1) u (a_p + r_p)
// (u A for rounding the products, gamma_{n-1} (1+u) A for the ordered additions), which feeds the
// same decisions as a screening pass; whenever it does not decide, the screening pass and then the
// ordered fold follow, so the result is bit-identical to the reference's.  An update of a feature
// that is not in G ends the tracking for the target.
// ---------------------------------------------------------------------------------------------
constexpr double kU = 0x1p-24;


struct GramLane {   // per-lane state, lane p = feature p
    double c, r, a;
    int g;          // row of the Gram matrix, -1: not tracked
};

// LDS arrays describing the selected features of the current target (K path).
struct FeatLds {
    int *f_id, *f_b, *f_e, *f_ever, *hist;
    float *f_nrm, *f_w, *f_s;
};
__host__ __device__ constexpr size_t feat_lds_bytes(int K) { return static_cast<size_t>(K) * 7 * 4 + 256 * 4 + 16; }
__device__ __forceinline__ FeatLds carve_feat(unsigned char *smem, int K) {
    FeatLds F;
    F.f_id = reinterpret_cast<int *>(smem);
    F.f_b = F.f_id + K;
    F.f_e = F.f_b + K;
    F.f_nrm = reinterpret_cast<float *>(F.f_e + K);
    F.f_w = F.f_nrm + K;
    F.f_s = F.f_w + K;
    F.f_ever = reinterpret_cast<int *>(F.f_s + K);
    F.hist = F.f_ever + K;
    return F;
}

// ---------------------------------------------------------------------------------------------
// Tolerance mode (rtrec_fit_opts.fast): the same coordinate descent -- same features (X^T y and the
// top-K selection stay exact), same xorshift coordinate sequence, same stopping rules -- but the dot
// products are no longer folded in the reference's left-to-right float32 order:
//   * par_dot: one order-free pass, 64 lanes + a shuffle tree (the wavefront partial sums of the brief);
//   * fit_gram_cd: when all of a target's features are rows of the shared Gram matrix the residual is
//     never formed at all.  With c_p = X_p . R the update of feature p by dw changes every c_q by
//     -dw G_pq, so a sweep costs K^2 multiply-adds on a K x K block of G held in LDS instead of K passes
//     over columns of tens of thousands of entries (sklearn's own `precompute` Gram solver,
//     _cd_fast.pyx enet_coordinate_descent_gram; rtrec asks for it, slim_elastic.py:201, but sklearn
//     ignores it for sparse X).  State in float64.
// Coefficients agree with the exact mode to ~1e-6 relative (a stopping test that flips by a rounding
// can cost one sweep on a rare target); DESIGN.md section 3.4, tests/test_gpu_kernels.py.
// ---------------------------------------------------------------------------------------------
__device__ float par_dot(const int *__restrict__ crow, const float *__restrict__ cval, const float *R,
                         int b, int e, float w_old) {
    const int lane = lane_id();
    const bool add_back = (w_old != 0.0f);
    float s0 = 0.0f;
    int o = b;
    for (; o + 512 <= e; o += 512) {
        int r[8];
        float x[8], v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { r[u] = crow[o + u * 64 + lane]; x[u] = cval[o + u * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = R[r[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float vv = v[u];
            if (add_back) vv = __fadd_rn(vv, __fmul_rn(x[u], w_old));
            s0 = __fadd_rn(s0, __fmul_rn(vv, x[u]));
        }
    }
    for (o += lane; o < e; o += 64) {
        const float x = cval[o];
        float vv = R[crow[o]];
        if (add_back) vv = __fadd_rn(vv, __fmul_rn(x, w_old));
        s0 = __fadd_rn(s0, __fmul_rn(vv, x));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s0 = __fadd_rn(s0, shfl_xor_t(s0, m));
    return s0;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_t(v, m);
    return v;
}

// Gram-form coordinate descent for one target whose Kc <= 64 features (lane p = feature p) all have a row in
// a.gram.  Inputs in LDS: f_id, f_s (= X_p . y, the exact float X^T y entries), f_nrm.  gs: Kc x 64 floats of
// LDS for the feature block of G.  Writes f_w (float32 coefficients) and returns sklearn's n_iter_.
__device__ int fit_gram_cd(const FitArgs &a, int Kc, const FeatLds &F, float *gs, float yy, float tol_s, int g_lane) {
    const int lane = lane_id();
    const double alpha = a.cfg.l1_reg, beta = a.cfg.l2_reg;
    const int positive = a.cfg.positive;
    // feature block of the Gram matrix: row p of gs = G[g_p][g_q] for lane q
    for (int p = 0; p < Kc; ++p) {
        const int gp = readlane_i(g_lane, p);
        float v = 0.0f;
        if (lane < Kc && gp >= 0 && g_lane >= 0) v = static_cast<float>(a.gram[static_cast<size_t>(gp) * a.gram_n + g_lane]);
        gs[p * 64 + lane] = v;
    }
    const bool live = lane < Kc && F.f_nrm[lane < Kc ? lane : 0] != 0.0f;
    const double q = live ? static_cast<double>(F.f_s[lane]) : 0.0;       // X_q . y
    const double nrm = live ? static_cast<double>(F.f_nrm[lane]) : 0.0;
    double c = q, w = 0.0;                                                // c = X_q . R,  R = y - sum_p w_p X_p
    uint32_t rng = a.cfg.seed;
    const int max_iter = a.cfg.max_iter;
    const double tol = a.cfg.tol;
    int n_iter = 0;
    for (; n_iter < max_iter; ++n_iter) {
        double w_max = 0.0, d_w_max = 0.0;
        for (int f = 0; f < Kc; ++f) {
            const int p = static_cast<int>(rand_int(static_cast<uint32_t>(Kc), rng));
            const double nrm_p = readlane_d(nrm, p);
            if (nrm_p == 0.0) continue;
            const double w_old = readlane_d(w, p);
            const double tmp = readlane_d(c, p) + w_old * nrm_p;           // X_p . (R + w_p X_p)
            double w_new;
            if (positive && tmp < 0.0) w_new = 0.0;
            else {
                const double mag = fabs(tmp) - alpha;
                w_new = (tmp > 0.0 ? 1.0 : (tmp < 0.0 ? -1.0 : 0.0)) * (mag > 0.0 ? mag : 0.0) / (nrm_p + beta);
            }
            if (w_new != w_old) {
                c -= (w_new - w_old) * static_cast<double>(gs[p * 64 + lane]);
                if (lane == p) w = w_new;
            }
            const double d = fabs(w_new - w_old);
            d_w_max = d > d_w_max ? d : d_w_max;
            const double aw = fabs(w_new);
            w_max = aw > w_max ? aw : w_max;
        }
        if (w_max == 0.0 || d_w_max / w_max < tol || n_iter == max_iter - 1) {
            // duality gap (_cd_fast.pyx:499-546) from the Gram state: XtA_q = c_q - beta w_q,
            // R.R = y.y - w.q - w.c (G w = q - c),  R.y = y.y - w.q
            double xta = live ? c - beta * w : 0.0;
            xta = positive ? xta : fabs(xta);
            double dn = live ? xta : (positive ? -1e300 : 0.0);
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { const double o = shfl_xor_t(dn, m); dn = o > dn ? o : dn; }
            const double wq = wave_sum_d(w * q), wc = wave_sum_d(w * c), ww = wave_sum_d(w * w), l1 = wave_sum_d(fabs(w));
            // features with a zero column (the target itself) have XtA = 0 exactly like the reference
            const unsigned long long dead = __ballot(lane < Kc && !live);
            if (dead && dn < 0.0) dn = 0.0;
            const double R_norm2 = static_cast<double>(yy) - wq - wc, Ry = static_cast<double>(yy) - wq;
            double cst, gap;
            if (dn > alpha) { cst = alpha / dn; gap = 0.5 * (R_norm2 + R_norm2 * cst * cst); }
            else { cst = 1.0; gap = R_norm2; }
            gap += alpha * l1 - cst * Ry + 0.5 * beta * (1.0 + cst * cst) * ww;
            if (gap < static_cast<double>(tol_s)) break;
        }
    }
    if (lane < Kc) F.f_w[lane] = static_cast<float>(w);
    return (n_iter < max_iter ? n_iter : max_iter - 1) + 1;
}

struct Prep {
    float yy, tol_s;
    int tc, Kc;
};

// Steps 0-2 for target j, executed by ONE wave: y.y, s = X^T y with its touched list, and (K path)
// the top-K feature selection written to the LDS feature arrays.
template <bool ALLF>
__device__ __forceinline__ Prep prep_target(const FitArgs &a, int j, int K, float *s, int *touched, float *cand_s,
                                            int *cand_i, const FeatLds &F, int tc_in = -1, int t_pre = -1) {
    const int lane = lane_id();
    const int I = a.I;
    int *f_id = F.f_id, *f_b = F.f_b, *f_e = F.f_e, *f_ever = F.f_ever;
    float *f_nrm = F.f_nrm, *f_w = F.f_w, *f_s = F.f_s;
    const int yb = a.cptr[j], ye = a.cptr[j + 1];

    // ---- y . y (tolerance scale, _cd_fast.pyx:426) ----
    float yy = 0.0f;
    for (int o = yb; o < ye; o += 64) {
        const int n = min(64, ye - o);
        float prod = 0.0f;
        if (lane < n) { const float y = a.cval[o + lane]; prod = __fmul_rn(y, y); }
        yy = chain_add(yy, prod, n);
    }
    const float tol_s = __fmul_rn(a.cfg.tol, yy);

    // ---- 1. s = X^T y (target column masked), touched list ----
    int tc = tc_in >= 0 ? tc_in : 0;
    if (t_pre >= 0) {    // the call's batched X^T y pass has the non-zero sums: scatter them, nothing to walk
        const int n0 = a.pre_cnt[t_pre];
        const size_t o0 = static_cast<size_t>(t_pre) * I;
        for (int tt = lane; tt < n0; tt += 64) {
            const int i = a.pre_i[o0 + tt];
            s[i] = a.pre_s[o0 + tt];
            touched[tt] = i;
        }
        tc = n0;
        tc_in = n0;
    }
    for (int ob = yb; ob < ye && tc_in < 0; ob += 64) {
        const int n = min(64, ye - ob);
        int rb_l = 0, re_l = 0;
        float y_l = 0.0f;
        if (lane < n) {
            const int u = a.crow[ob + lane];
            y_l = a.cval[ob + lane];
            rb_l = a.rptr[u];
            re_l = a.rptr[u + 1];
        }
        for (int q = 0; q < n; ++q) {
            const int rb = readlane_i(rb_l, q), re = readlane_i(re_l, q);
            const float yv = readlane_f(y_l, q);
            // Items of one row are distinct, so the s updates of a row are independent: 8 x 64 of them
            // are gathered together (rows themselves stay strictly sequential = csr_matvec order).
            for (int o = rb; o < re; o += 64 * kRowUnroll) {
                int it[kRowUnroll];
                float xv[kRowUnroll], old[kRowUnroll];
#pragma unroll
                for (int k = 0; k < kRowUnroll; ++k) {
                    const int oo = o + k * 64 + lane;
                    it[k] = -1; xv[k] = 0.0f;
                    if (oo < re) { it[k] = a.rcol[oo]; xv[k] = a.rval[oo]; }
                    if (it[k] == j) it[k] = -1;
                }
#pragma unroll
                for (int k = 0; k < kRowUnroll; ++k) { old[k] = 0.0f; if (it[k] >= 0) old[k] = s[it[k]]; }
#pragma unroll
                for (int k = 0; k < kRowUnroll; ++k) {
                    if (o + k * 64 >= re) break;
                    bool first = false;
                    if (it[k] >= 0) {
                        first = (__float_as_uint(old[k]) == kUntouched);
                        s[it[k]] = __fadd_rn(first ? 0.0f : old[k], __fmul_rn(xv[k], yv));
                    }
                    const unsigned long long m = __ballot(first);
                    if (m) {
                        if (first) touched[tc + lane_prefix(m)] = it[k];
                        tc += __builtin_popcountll(m);
                    }
                }
            }
        }
    }

    // ---- 2. feature selection: top-K of s, descending, ties -> higher id ----
    int Kc = 0;
    if (!ALLF) {
        // compact the non-zero scores of the neighbourhood
        int cn = 0;
        for (int tb = 0; tb < tc; tb += 64) {
            const int tt = tb + lane;
            int i = -1;
            float v = 0.0f;
            if (tt < tc) { i = touched[tt]; v = s[i]; }
            const bool keep = (tt < tc) && (v != 0.0f);
            const unsigned long long m = __ballot(keep);
            if (m) {
                if (keep) { const int pos = cn + lane_prefix(m); cand_s[pos] = v; cand_i[pos] = i; }
                cn += __builtin_popcountll(m);
            }
        }
        const float ninf = -__builtin_huge_valf();
        // ---- positives: the K best by (score desc, id desc) ----
        // Radix select on the float bits (positive floats order like unsigned ints): four 8-bit
        // histogram passes find the K-th largest score tau; everything above tau is taken, ties at
        // tau are taken by descending id; the (at most K) winners are then ranked in LDS.  This
        // replaces K dependent argmax rounds over global memory (150 us -> a few us per target).
        int *hbin = F.hist;                           // 256 LDS bins
        int npos = 0;
        for (int tb = 0; tb < cn; tb += 64) {
            const bool pos = (tb + lane < cn) && (cand_s[tb + lane] > 0.0f);
            npos += __builtin_popcountll(__ballot(pos));
        }
        uint32_t tau = 0u;        // scores strictly above tau are always selected
        int need_tie = 0;         // how many candidates with score == tau are selected (highest ids)
        if (npos > K) {
            uint32_t prefix = 0u, mask = 0u;
            int need = K;
            for (int shift = 24; shift >= 0; shift -= 8) {
                for (int bnk = lane; bnk < 256; bnk += 64) hbin[bnk] = 0;
                for (int tt = lane; tt < cn; tt += 64) {
                    const float v = cand_s[tt];
                    if (v > 0.0f) {
                        const uint32_t key = __float_as_uint(v);
                        if ((key & mask) == prefix) atomicAdd(&hbin[(key >> shift) & 255u], 1);
                    }
                }
                // lane l owns bins 4l..4l+3; walk from the top bin down until `need` is covered
                const int c0 = hbin[4 * lane], c1 = hbin[4 * lane + 1], c2 = hbin[4 * lane + 2], c3 = hbin[4 * lane + 3];
                const int tot = c0 + c1 + c2 + c3;
                int above = 0, sel_lane = 0;
                for (int l = 63; l >= 0; --l) {
                    const int tl = readlane_i(tot, l);
                    if (above + tl >= need) { sel_lane = l; break; }
                    above += tl;
                }
                const int b3 = readlane_i(c3, sel_lane), b2 = readlane_i(c2, sel_lane), b1 = readlane_i(c1, sel_lane);
                int bin = 4 * sel_lane + 3;
                if (above + b3 < need) { above += b3; bin--; if (above + b2 < need) { above += b2; bin--; if (above + b1 < need) { above += b1; bin--; } } }
                prefix |= static_cast<uint32_t>(bin) << shift;
                mask |= 255u << shift;
                need -= above;
            }
            tau = prefix;
            need_tie = need;      // >= 1
        }
        // collect the winners (unsorted) into f_id / f_s
        int n_sel = 0;
        for (int tb = 0; tb < cn; tb += 64) {
            const int tt = tb + lane;
            float v = 0.0f;
            int id = -1;
            if (tt < cn) { v = cand_s[tt]; id = cand_i[tt]; }
            const uint32_t key = __float_as_uint(v);
            const bool take = (v > 0.0f) && (npos <= K || key > tau);
            const unsigned long long m = __ballot(take);
            if (m) {
                if (take) { const int pos = n_sel + lane_prefix(m); f_id[pos] = id; f_s[pos] = v; }
                n_sel += __builtin_popcountll(m);
            }
        }
        if (npos > K) {
            // ties at tau: highest ids first (exact score ties are rare with float ratings)
            int last_id = 0x7fffffff;
            for (int r = 0; r < need_tie; ++r) {
                int best = -1;
                for (int tt = lane; tt < cn; tt += 64) {
                    const float v = cand_s[tt];
                    if (__float_as_uint(v) == tau && v > 0.0f) { const int id = cand_i[tt]; if (id < last_id && id > best) best = id; }
                }
                best = wave_max(best);
                last_id = best;
                if (lane == 0) { f_id[n_sel] = best; f_s[n_sel] = __uint_as_float(tau); }
                n_sel++;
            }
        }
        // rank the winners: position = number of better entries (strict total order)
        {
            int *tmp_i = f_e;                                   // K ints of scratch
            float *tmp_s = f_w;                                 // K floats of scratch (f_w is reset below)
            for (int pb = 0; pb < n_sel; pb += 64) {
                const int pp = pb + lane;
                Cand<float> me; me.id = -1; me.score = ninf; me.aux = 0u;
                if (pp < n_sel) { me.id = f_id[pp]; me.score = f_s[pp]; }
                int rank = 0;
                for (int q = 0; q < n_sel; ++q) {
                    Cand<float> o; o.id = f_id[q]; o.score = f_s[q]; o.aux = 0u;
                    rank += cand_better(o, me) ? 1 : 0;
                }
                if (pp < n_sel) { tmp_i[rank] = me.id; tmp_s[rank] = me.score; }
            }
            for (int pp = lane; pp < n_sel; pp += 64) { f_id[pp] = tmp_i[pp]; f_s[pp] = tmp_s[pp]; }
        }
        Kc = n_sel;
        // zero scores (untouched items, the target itself, exact-zero sums): higher id first
        for (int hi = I - 1; hi >= 0 && Kc < K; hi -= 64) {
            const int id = hi - lane;
            bool isz = false;
            if (id >= 0) {
                const float v = s[id];
                isz = (__float_as_uint(v) == kUntouched) || (v == 0.0f);
            }
            const unsigned long long m = __ballot(isz);
            if (m) {
                const int pos = Kc + lane_prefix(m);
                if (isz && pos < K) { f_id[pos] = id; f_s[pos] = 0.0f; }
                Kc = min(K, Kc + __builtin_popcountll(m));
            }
        }
        // negatives (only when I - #positive - #zero < K), closest to zero first
        for (; Kc < K; ++Kc) {
            Cand<float> b; b.id = -1; b.score = ninf; b.aux = 0u;
            int bt = -1;
            for (int tt = lane; tt < cn; tt += 64) {
                const float v = cand_s[tt];
                if (!(v < 0.0f) || v == ninf) continue;
                Cand<float> x; x.score = v; x.id = cand_i[tt]; x.aux = 0u;
                if (cand_better(x, b)) { b = x; bt = tt; }
            }
            const Cand<float> w = wave_best(b);
            if (w.id < 0) break;
            if (bt >= 0 && b.id == w.id) cand_s[bt] = ninf;
            if (lane == 0) { f_id[Kc] = w.id; f_s[Kc] = w.score; }
        }
        for (int p = lane; p < Kc; p += 64) {
            const int c = f_id[p];
            f_b[p] = a.cptr[c];
            f_e[p] = a.cptr[c + 1];
            f_nrm[p] = (c == j) ? 0.0f : a.sqn[c];
            f_w[p] = 0.0f;
            f_ever[p] = 0;
        }
    }
    Prep P;
    P.yy = yy; P.tol_s = tol_s; P.tc = tc; P.Kc = Kc;
    return P;
}

template <bool ALLF, bool SPEC>
__device__ void fit_one(const FitArgs &a, int t, int slot, unsigned char *smem) {
    const int lane = lane_id();
    const int U = a.U, I = a.I;
    const int j = a.targets[t];
    float *R = a.R + static_cast<size_t>(slot) * U;
    float *s = a.s + static_cast<size_t>(slot) * I;
    int *touched = a.touched + static_cast<size_t>(slot) * I;
    float *cand_s = a.cand_s + static_cast<size_t>(slot) * I;
    int *cand_i = a.cand_i + static_cast<size_t>(slot) * I;
    float *w_all = ALLF ? a.w_all + static_cast<size_t>(slot) * I : nullptr;
    int *ever_flag = reinterpret_cast<int *>(cand_s);   // ALL path
    int *ever_list = cand_i;                            // ALL path

    const int K = ALLF ? 0 : min(a.cfg.top_features, I);
    float *fold_buf = reinterpret_cast<float *>(smem);           // kFoldBufBytes, then the feature arrays
    const FeatLds F = carve_feat(smem + kFoldBufBytes, K);
    int *f_id = F.f_id, *f_b = F.f_b, *f_e = F.f_e, *f_ever = F.f_ever;
    float *f_nrm = F.f_nrm, *f_w = F.f_w, *f_s = F.f_s;

    const float alpha = a.cfg.l1_reg, beta = a.cfg.l2_reg;
    const int positive = a.cfg.positive;
    const int yb = a.cptr[j], ye = a.cptr[j + 1];
    const int ny = ye - yb;

    const long long tr0 = a.trace ? static_cast<long long>(wall_clock64()) : 0;
    long long tr_folded = 0;
#ifdef RTREC_FIT_PHASES      // diagnostic build (tools/fit_trace.py --phases): where a target's coordinate descent spends its time
    long long ph_fold = 0, ph_upd = 0, ph_screen = 0, ph_gap = 0;
#define FIT_PH_T0 const long long ph_t0 = static_cast<long long>(wall_clock64());
#define FIT_PH_ADD(acc) acc += static_cast<long long>(wall_clock64()) - ph_t0;
#else
#define FIT_PH_T0
#define FIT_PH_ADD(acc)
#endif
    const Prep P = prep_target<ALLF>(a, j, K, s, touched, cand_s, cand_i, F);
    const long long tr1 = a.trace ? static_cast<long long>(wall_clock64()) : 0;
    const float yy = P.yy, tol_s = P.tol_s;
    const int tc = P.tc, Kc = P.Kc;
    const int nf = ALLF ? I : Kc;

    // Gram tracking state (see above); starts with the pristine residual R == y
    bool gram_on = !ALLF && a.gram != nullptr && Kc <= 64;
    GramLane GL; GL.c = 0.0; GL.r = 0.0; GL.a = 0.0; GL.g = -1;
    if (gram_on && lane < Kc && f_nrm[lane] != 0.0f) {
        GL.g = a.gram_index[f_id[lane]];
        if (GL.g >= 0) {
            const double fs = static_cast<double>(f_s[lane]);
            const double m1 = static_cast<double>(min(f_e[lane] - f_b[lane], ny) + 1);
            GL.c = fs;
            GL.r = 1.2 * m1 * kU * fs;
            GL.a = fs * (1.0 + 1.1 * m1 * kU);
            if (!(fs >= 0.0) || m1 * kU > 0x1p-4) GL.g = -1;
        }
    }
    auto gram_interval = [&](int p, int n, Screen &S) -> bool {   // uniform p; true if feature p is tracked
        if (!gram_on) return false;
        if (readlane_i(GL.g, p) < 0) return false;
        const double c = readlane_d(GL.c, p), r = readlane_d(GL.r, p), aa = readlane_d(GL.a, p);
        const double band = r + 1.07 * static_cast<double>(n + 1) * kU * (aa + r);
        S.lo = c - band; S.hi = c + band;
        return n <= kScreenMaxLen;
    };
    auto gram_update = [&](int q, float w_old, float w_new) {     // after the residual update of feature q
        if (!gram_on) return;
        const int gq = readlane_i(GL.g, q);
        if (gq < 0) { gram_on = false; return; }                  // G has no row for q: tracking ends
        if (GL.g >= 0) {
            const double g = a.gram[static_cast<size_t>(gq) * a.gram_n + GL.g];
            const double d = a.gram_rel_err;
            const double awo = fabs(static_cast<double>(w_old)), awn = fabs(static_cast<double>(w_new));
            const double dw = static_cast<double>(w_new) - static_cast<double>(w_old);
            const double da = (awn - awo) * g * (awn > awo ? 1.0 + d : 1.0 - d);
            const double a_new = GL.a + da + 1e-15 * (GL.a + fabs(da));
            const double abar = (a_new > GL.a ? a_new : GL.a) + GL.r;
            GL.r += fabs(dw) * g * d * (1.0 + d) + 3.05 * kU * (abar + (awo + awn) * g * (1.0 + d)) +
                    4e-16 * (fabs(GL.c) + fabs(dw * g));
            GL.c -= dw * g;
            GL.a = a_new;
        }
    };

    // ---- tolerance mode, all features in the Gram matrix: Gram-form coordinate descent, no residual ----
    bool gram_done = false;
    int gram_iters = 0;
    if (!ALLF && a.fast >= 2 && a.gram != nullptr && Kc > 0 && Kc <= 64 && ny > 0) {
        int g_lane = -1;
        if (lane < Kc) g_lane = a.gram_index[f_id[lane]];
        const unsigned long long missing = __ballot(lane < Kc && f_nrm[lane] != 0.0f && g_lane < 0);
        if (!missing) {
            float *gs = reinterpret_cast<float *>(smem + kFoldBufBytes + ((feat_lds_bytes(K) + 15) / 16) * 16);
            gram_iters = fit_gram_cd(a, Kc, F, gs, P.yy, P.tol_s, g_lane);
            gram_done = true;
        }
    }
    if (a.fast) gram_on = false;   // the tracking screens serve the ordered folds only

    // ---- 3. coordinate descent (_cd_fast.pyx:428-548) ----
    bool dirty = false;            // false while every w is still 0, i.e. R == y
    int n_ever = 0;                // ALL path: features whose w was ever non-zero
    float gap = __fadd_rn(a.cfg.tol, 1.0f);
    uint32_t rng = a.cfg.seed;
    int n_iter = 0;
    const int max_iter = a.cfg.max_iter;
    const bool skip_cd = (ny == 0) || (nf == 0) || gram_done;   // y == 0: all coefficients stay 0, max_iter sweeps
    if (skip_cd) n_iter = max_iter > 0 ? max_iter - 1 : 0;
    if (gram_done) n_iter = gram_iters - 1;

    auto s_value = [&](int p) -> float {
        const float v = s[p];
        return (__float_as_uint(v) == kUntouched) ? 0.0f : v;
    };

    for (n_iter = skip_cd ? n_iter : 0; !skip_cd && n_iter < max_iter; ++n_iter) {
        float w_max = 0.0f, d_w_max = 0.0f;
        for (int f = 0; f < nf; ++f) {
            const int p = static_cast<int>(rand_int(static_cast<uint32_t>(nf), rng));
            float nrm;
            int b, e;
            float w_old;
            if (ALLF) {
                nrm = (p == j) ? 0.0f : a.sqn[p];
                if (nrm == 0.0f) continue;
                b = a.cptr[p]; e = a.cptr[p + 1];
                w_old = w_all[p];
            } else {
                nrm = f_nrm[p];
                if (nrm == 0.0f) continue;
                b = f_b[p]; e = f_e[p];
                w_old = f_w[p];
            }
            float tmp = 0.0f, w_new = 0.0f;
            bool screened = false;
            if (!dirty) tmp = ALLF ? s_value(p) : f_s[p];
            else {
                if (a.fast) {
                    tmp = par_dot(a.crow, a.cval, R, b, e, w_old);
                } else if (w_old == 0.0f) {
                    Screen S;
                    if (!ALLF && gram_interval(p, e - b, S)) screened = screen_stays_zero(S, alpha, positive, w_new);
                    if (!screened && e - b >= a.screen_min && e - b <= kScreenMaxLen) {
                        float ps, pa;
                        FIT_PH_T0
                        screen_pass(a.crow, a.cval, R, b, e, ps, pa);
                        FIT_PH_ADD(ph_screen)
                        screened = screen_stays_zero(screen_interval(ps, pa, e - b), alpha, positive, w_new);
                    }
                }
                if (!screened && !a.fast) {
                    FIT_PH_T0
                    tmp = dot_pass_any<SPEC>(a.crow, a.cval, R, b, e, w_old, fold_buf, a.spec_min); tr_folded += e - b;
                    FIT_PH_ADD(ph_fold)
                }
            }
            if (!screened) w_new = cd_update(tmp, alpha, beta, nrm, positive);
            if (w_old != 0.0f || w_new != 0.0f) {
                if (!dirty) {   // materialise R = y
                    for (int o = yb + lane; o < ye; o += 64) R[a.crow[o]] = a.cval[o];
                    dirty = true;
                }
                {
                    FIT_PH_T0
                    update_pass(a.crow, a.cval, R, b, e, w_old, w_new);
                    FIT_PH_ADD(ph_upd)
                }
                if (!ALLF) gram_update(p, w_old, w_new);
                if (ALLF) {
                    if (ever_flag[p] == 0) { if (lane == 0) { ever_flag[p] = 1; ever_list[n_ever] = p; } n_ever++; }
                    if (lane == 0) w_all[p] = w_new;
                } else {
                    if (lane == 0) f_ever[p] = 1;
                }
            }
            // always store: the sign of a zero coefficient follows the last update (w = -1 * 0 / d)
            if (!ALLF && lane == 0) f_w[p] = w_new;
            const float d = fabsf(__fsub_rn(w_new, w_old));
            d_w_max = d > d_w_max ? d : d_w_max;
            const float aw = fabsf(w_new);
            w_max = aw > w_max ? aw : w_max;
        }

        if (w_max == 0.0f || __fdiv_rn(d_w_max, w_max) < a.cfg.tol || n_iter == max_iter - 1) {
            FIT_PH_T0
            // dual norm of XtA = X^T R - beta w
            float dn = 0.0f;
            bool dn_init = false;
            auto dn_take = [&](float xta) {
                const float v = positive ? xta : fabsf(xta);
                if (!dn_init) { dn = v; dn_init = true; } else if (v > dn) dn = v;
            };
            if (ALLF) {
                if (!dirty) {
                    // XtA[p] = s[p] for every feature (0 for untouched ones and the target)
                    float mx = 0.0f;   // I >= 1 and the target itself contributes exactly 0
                    for (int tb = 0; tb < tc; tb += 64) {
                        float v = 0.0f;
                        if (tb + lane < tc) { v = s[touched[tb + lane]]; v = positive ? v : fabsf(v); }
                        const float m = wave_max(v);
                        mx = m > mx ? m : mx;
                    }
                    dn = mx; dn_init = true;
                } else {
                    for (int p = 0; p < I; ++p) {
                        float xta = 0.0f;
                        const int b = a.cptr[p], e = a.cptr[p + 1];
                        if (p != j && b != e) {
                            xta = xta_pass<SPEC>(a.crow, a.cval, R, b, e, fold_buf, a.spec_min);
                            xta = __fsub_rn(xta, __fmul_rn(beta, w_all[p]));
                        }
                        dn_take(xta);
                    }
                }
            } else if (!dirty) {
                for (int p = 0; p < nf; ++p) dn_take(f_nrm[p] != 0.0f ? f_s[p] : 0.0f);
            } else {
                // Only max_p XtA[p] is needed: screen every column, then fold in order only the
                // columns whose interval reaches the best lower bound (the maximum is among them).
                float *x_lo = cand_s, *x_hi = reinterpret_cast<float *>(cand_i);   // free after step 2
                float best_lo = -__builtin_huge_valf();
                for (int p = 0; p < nf; ++p) {
                    float lo = 0.0f, hi = 0.0f;
                    if (f_nrm[p] != 0.0f) {
                        const int b = f_b[p], e = f_e[p];
                        const float bw = __fmul_rn(beta, f_w[p]);
                        Screen S;
                        if (a.fast) {
                            const float xta = __fsub_rn(par_dot(a.crow, a.cval, R, b, e, 0.0f), bw);
                            lo = hi = positive ? xta : fabsf(xta);
                        } else if (gram_interval(p, e - b, S)) {
                            screen_xta_interval(S, bw, positive, lo, hi);
                        } else if (e - b >= a.screen_min && e - b <= kScreenMaxLen) {
                            float ps, pa;
                            screen_pass(a.crow, a.cval, R, b, e, ps, pa);
                            screen_xta_interval(screen_interval(ps, pa, e - b), bw, positive, lo, hi);
                        } else {
                            const float xta = __fsub_rn(xta_pass<SPEC>(a.crow, a.cval, R, b, e, fold_buf, a.spec_min), bw);
                            tr_folded += e - b;
                            lo = hi = positive ? xta : fabsf(xta);
                        }
                    }
                    if (lane == 0) { x_lo[p] = lo; x_hi[p] = hi; }
                    best_lo = lo > best_lo ? lo : best_lo;
                }
                for (int p = 0; p < nf; ++p) {
                    const float lo = x_lo[p], hi = x_hi[p];
                    if (!(hi >= best_lo)) continue;
                    float v = lo;
                    if (lo != hi) {
                        const float xta = __fsub_rn(xta_pass<SPEC>(a.crow, a.cval, R, f_b[p], f_e[p], fold_buf, a.spec_min), __fmul_rn(beta, f_w[p]));
                        tr_folded += f_e[p] - f_b[p];
                        v = positive ? xta : fabsf(xta);
                    }
                    dn_take(v);
                }
            }
            // R.R, R.y, w.w, |w|_1 in ascending index order (canonical order)
            float R_norm2, Ry, w_norm2 = 0.0f, l1 = 0.0f;
            if (!dirty) { R_norm2 = yy; Ry = yy; }
            else {
                R_norm2 = 0.0f;
                if (a.fast) {
                    float acc = 0.0f;
                    for (int o = lane; o < U; o += 64) { const float v = R[o]; acc = __fadd_rn(acc, __fmul_rn(v, v)); }
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) acc = __fadd_rn(acc, shfl_xor_t(acc, m));
                    R_norm2 = acc;
                } else
                for (int o = 0; o < U; o += 64) {
                    float prod = 0.0f;
                    if (o + lane < U) { const float v = R[o + lane]; prod = __fmul_rn(v, v); }
                    if (__ballot(prod != 0.0f)) R_norm2 = chain_add(R_norm2, prod, min(64, U - o));
                }
                Ry = 0.0f;
                for (int o = yb; o < ye; o += 64) {
                    const int n = min(64, ye - o);
                    float prod = 0.0f;
                    if (lane < n) prod = __fmul_rn(R[a.crow[o + lane]], a.cval[o + lane]);
                    Ry = chain_add(Ry, prod, n);
                }
                for (int o = 0; o < nf; o += 64) {
                    float wv = 0.0f;
                    if (o + lane < nf) wv = ALLF ? w_all[o + lane] : f_w[o + lane];
                    if (__ballot(wv != 0.0f)) {
                        const int n = min(64, nf - o);
                        w_norm2 = chain_add(w_norm2, __fmul_rn(wv, wv), n);
                        l1 = chain_add(l1, fabsf(wv), n);
                    }
                }
            }
            float cst;
            if (dn > alpha) {
                cst = __fdiv_rn(alpha, dn);
                const float A_norm2 = __fmul_rn(R_norm2, __fmul_rn(cst, cst));
                gap = static_cast<float>(0.5 * static_cast<double>(__fadd_rn(R_norm2, A_norm2)));
            } else {
                cst = 1.0f;
                gap = R_norm2;
            }
            const float t12 = __fsub_rn(__fmul_rn(alpha, l1), __fmul_rn(cst, Ry));
            const double t3 = 0.5 * static_cast<double>(beta) * static_cast<double>(__fadd_rn(1.0f, __fmul_rn(cst, cst))) *
                              static_cast<double>(w_norm2);
            gap = static_cast<float>(static_cast<double>(gap) + (static_cast<double>(t12) + t3));
            FIT_PH_ADD(ph_gap)
            if (gap < tol_s) break;
        }
    }
    const int n_iter_out = (n_iter < max_iter ? n_iter : max_iter - 1) + 1;

    // ---- outputs ----
    int *oi = a.out_items + static_cast<size_t>(t) * a.cap;
    float *oc = a.out_coef + static_cast<size_t>(t) * a.cap;
    if (ALLF) {
        int cnt = 0;
        // non-zero coefficients ascending by id; n_ever is small, so rank the ever-list
        for (int k = lane; k < n_ever; k += 64) {
            const int p = ever_list[k];
            const float wv = w_all[p];
            if (wv != 0.0f) {
                int rank = 0;
                for (int q = 0; q < n_ever; ++q) {
                    const int pq = ever_list[q];
                    if (pq < p && w_all[pq] != 0.0f) rank++;
                }
                oi[rank] = p; oc[rank] = wv;
            }
        }
        for (int k = 0; k < n_ever; k += 64) {
            bool nz = false;
            if (k + lane < n_ever) nz = (w_all[ever_list[k + lane]] != 0.0f);
            cnt += __builtin_popcountll(__ballot(nz));
        }
        if (lane == 0) a.out_count[t] = cnt;
    } else {
        for (int p = lane; p < Kc; p += 64) { oi[p] = f_id[p]; oc[p] = f_w[p]; }
        if (lane == 0) a.out_count[t] = Kc;
    }
    if (lane == 0) a.out_niter[t] = n_iter_out;

    // ---- restore the scratch invariants ----
    if (dirty) {
        for (int o = yb + lane; o < ye; o += 64) R[a.crow[o]] = 0.0f;
        if (ALLF) {
            for (int k = 0; k < n_ever; ++k) {
                const int p = ever_list[k];
                for (int o = a.cptr[p] + lane; o < a.cptr[p + 1]; o += 64) R[a.crow[o]] = 0.0f;
            }
        } else {
            for (int p = 0; p < Kc; ++p) {
                if (f_ever[p] == 0) continue;
                for (int o = f_b[p] + lane; o < f_e[p]; o += 64) R[a.crow[o]] = 0.0f;
            }
        }
    }
    if (ALLF) {
        for (int k = lane; k < n_ever; k += 64) { const int p = ever_list[k]; w_all[p] = 0.0f; ever_flag[p] = 0; }
    }
    for (int tt = lane; tt < tc; tt += 64) s[touched[tt]] = __uint_as_float(kUntouched);
    if (a.trace && lane == 0) {
        long long *tr = a.trace + static_cast<size_t>(t) * 8;
        tr[0] = tr0; tr[1] = tr1; tr[2] = static_cast<long long>(wall_clock64()); tr[3] = tr_folded;
#ifdef RTREC_FIT_PHASES
        tr[4] = ph_fold; tr[5] = ph_upd; tr[6] = ph_screen; tr[7] = ph_gap;
#endif
    }
}
#undef FIT_PH_T0
#undef FIT_PH_ADD


// =============================================================================================
// Every-item-is-a-feature path (nn_feature_selection = None, the reference's default): I draws per
// sweep, nearly all of them over SHORT columns whose coefficient is and stays zero.
//
// The draws of a sweep are taken 64 at a time: the xorshift sequence is advanced 64 steps, the
// per-draw metadata (norm, column range, current coefficient) arrives in one round trip, and every
// lane folds ITS OWN column strictly in order -- a lane-private sequential accumulation is exactly
// the reference's operation order, so for columns of up to kLaneMax entries the batch yields 64 exact
// dot products at once instead of one.  The draws are then committed in order: zero-stays-zero draws
// are no-ops; the first draw that changes the residual is applied (wave-wide update pass), after
// which the lane-private dots of the LATER draws are recomputed, because they saw the old residual.
// Columns longer than kLaneMax keep the wave-wide screening pass + ordered fold.
// The duality gap's max_p XtA[p] is computed the same way: exact per lane for short columns,
// screened (and folded only if they can be the maximum) for long ones.
// =============================================================================================
constexpr int kLaneMaxDefault = 256;   // FitArgs::lane_max

__device__ __forceinline__ float lane_dot(const int *__restrict__ crow, const float *__restrict__ cval, const float *R,
                                          int b, int e, float w_old) {
    // 8 entries per step, software-pipelined: while the residual gathers of group g are in flight the
    // (independent) index / value loads of group g+1 are issued, so a step costs one memory round trip
    const bool add_back = (w_old != 0.0f);
    float tmp = 0.0f;
    int o = b;
    if (o + 8 <= e) {
        int r[8];
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { r[u] = crow[o + u]; x[u] = cval[o + u]; }
        for (;;) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = R[r[u]];
            const int on = o + 8;
            const bool more = on + 8 <= e;
            int rn[8];
            float xn[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { rn[u] = 0; xn[u] = 0.0f; }
            if (more) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { rn[u] = crow[on + u]; xn[u] = cval[on + u]; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                float vv = v[u];
                if (add_back) vv = __fadd_rn(vv, __fmul_rn(x[u], w_old));
                tmp = __fadd_rn(tmp, __fmul_rn(vv, x[u]));
            }
            o = on;
            if (!more) break;
#pragma unroll
            for (int u = 0; u < 8; ++u) { r[u] = rn[u]; x[u] = xn[u]; }
        }
    }
    for (; o < e; ++o) {
        const float x = cval[o];
        float vv = R[crow[o]];
        if (add_back) vv = __fadd_rn(vv, __fmul_rn(x, w_old));
        tmp = __fadd_rn(tmp, __fmul_rn(vv, x));
    }
    return tmp;
}

template <bool SPEC>
__device__ void fit_one_allf(const FitArgs &a, int t, int slot, unsigned char *smem) {
    const int lane = lane_id();
    const int U = a.U, I = a.I;
    const int j = a.targets[t];
    float *R = a.R + static_cast<size_t>(slot) * U;
    float *s = a.s + static_cast<size_t>(slot) * I;
    int *touched = a.touched + static_cast<size_t>(slot) * I;
    float *cand_s = a.cand_s + static_cast<size_t>(slot) * I;
    int *cand_i = a.cand_i + static_cast<size_t>(slot) * I;
    float *w_all = a.w_all + static_cast<size_t>(slot) * I;
    int *ever_flag = reinterpret_cast<int *>(cand_s);
    int *ever_list = cand_i;
    float *fold_buf = reinterpret_cast<float *>(smem);
    const FeatLds F = carve_feat(smem + kFoldBufBytes, 0);
    int *oi = a.out_items + static_cast<size_t>(t) * a.cap;
    float *oc = a.out_coef + static_cast<size_t>(t) * a.cap;
    int *long_list = a.long_list + static_cast<size_t>(slot) * I;

    const float alpha = a.cfg.l1_reg, beta = a.cfg.l2_reg;
    const int positive = a.cfg.positive;
    const int yb = a.cptr[j], ye = a.cptr[j + 1];
    const int ny = ye - yb;

    const long long tr0 = a.trace ? static_cast<long long>(wall_clock64()) : 0;
    long long tr_folded = 0;
    const Prep P = prep_target<true>(a, j, 0, s, touched, cand_s, cand_i, F);
    const long long tr1 = a.trace ? static_cast<long long>(wall_clock64()) : 0;
    const float yy = P.yy, tol_s = P.tol_s;
    const int tc = P.tc;
    const int nf = I;

    bool dirty = false;
    int n_ever = 0;
    float gap = __fadd_rn(a.cfg.tol, 1.0f);
    uint32_t rng = a.cfg.seed;
    const int max_iter = a.cfg.max_iter;
    const bool skip_cd = (ny == 0) || (nf == 0);
    int n_iter = skip_cd ? (max_iter > 0 ? max_iter - 1 : 0) : 0;
    const float ninf = -__builtin_huge_valf();

    auto s_value = [&](int p) -> float {
        const float v = s[p];
        return (__float_as_uint(v) == kUntouched) ? 0.0f : v;
    };
    // wave-wide exact / screened evaluation of one long column (uniform arguments)
    auto long_tmp = [&](int p, int b, int e, float w_old, bool &screened, float &w_new) -> float {
        screened = false;
        if (!dirty) return s_value(p);
        if (w_old == 0.0f && e - b >= a.screen_min && e - b <= kScreenMaxLen) {
            float ps, pa;
            screen_pass(a.crow, a.cval, R, b, e, ps, pa);
            screened = screen_stays_zero(screen_interval(ps, pa, e - b), alpha, positive, w_new);
            if (screened) return 0.0f;
        }
        tr_folded += e - b;
        return dot_pass_any<SPEC>(a.crow, a.cval, R, b, e, w_old, fold_buf, a.spec_min);
    };

    long long ph_rng = 0, ph_dots = 0, ph_loop = 0, ph_gap = 0;      // phase clocks (trace only)
    const bool prof = a.trace != nullptr;
    for (; !skip_cd && n_iter < max_iter; ++n_iter) {
        float w_max = 0.0f, d_w_max = 0.0f;
        for (int f0 = 0; f0 < nf; f0 += 64) {
            const int nb = min(64, nf - f0);
            int p_l = 0;
            const long long c0 = prof ? static_cast<long long>(wall_clock64()) : 0;
            for (int k = 0; k < nb; ++k) {
                const int p = static_cast<int>(rand_int(static_cast<uint32_t>(nf), rng));
                if (lane == k) p_l = p;
            }
            float nrm_l = 0.0f, wold_l = 0.0f, tmp_l = 0.0f;
            int b_l = 0, e_l = 0;
            if (lane < nb) {
                nrm_l = (p_l == j) ? 0.0f : a.sqn[p_l];
                b_l = a.cptr[p_l]; e_l = a.cptr[p_l + 1];
                wold_l = w_all[p_l];
            }
            const bool cand_l = lane < nb && nrm_l != 0.0f;
            const bool short_l = cand_l && (e_l - b_l) <= a.lane_max;
            auto lane_dots = [&](int from) {
                if (short_l && lane >= from)
                    tmp_l = dirty ? lane_dot(a.crow, a.cval, R, b_l, e_l, wold_l) : s_value(p_l);
            };
            const long long c1 = prof ? static_cast<long long>(wall_clock64()) : 0;
            lane_dots(0);
            const long long c2 = prof ? static_cast<long long>(wall_clock64()) : 0;
            ph_rng += c1 - c0; ph_dots += c2 - c1;
            int k = 0;
            while (k < nb) {
                float wnew_l = 0.0f;
                if (short_l) wnew_l = cd_update(tmp_l, alpha, beta, nrm_l, positive);
                const bool need_l = cand_l && lane >= k && (!short_l || wold_l != 0.0f || wnew_l != 0.0f);
                const unsigned long long m = __ballot(need_l);
                if (!m) break;
                const int q = __builtin_ctzll(m);
                const int p = readlane_i(p_l, q);
                const float nrm = readlane_f(nrm_l, q);
                const int b = readlane_i(b_l, q), e = readlane_i(e_l, q);
                const float w_old = readlane_f(wold_l, q);
                float w_new = 0.0f;
                if (e - b <= a.lane_max) {
                    w_new = readlane_f(wnew_l, q);
                } else {
                    bool screened = false;
                    const float tmp = long_tmp(p, b, e, w_old, screened, w_new);
                    if (!screened) w_new = cd_update(tmp, alpha, beta, nrm, positive);
                }
                if (w_old != 0.0f || w_new != 0.0f) {
                    if (!dirty) {   // materialise R = y
                        for (int o = yb + lane; o < ye; o += 64) R[a.crow[o]] = a.cval[o];
                        dirty = true;
                    }
                    update_pass(a.crow, a.cval, R, b, e, w_old, w_new);
                    if (ever_flag[p] == 0) { if (lane == 0) { ever_flag[p] = 1; ever_list[n_ever] = p; } n_ever++; }
                    if (lane == 0) w_all[p] = w_new;
                    // later draws of the batch: the same coordinate sees the new coefficient, and every
                    // lane-private dot is stale because the residual changed
                    if (lane > q && lane < nb && p_l == p) wold_l = w_new;
                    lane_dots(q + 1);
                }
                const float d = fabsf(__fsub_rn(w_new, w_old));
                d_w_max = d > d_w_max ? d : d_w_max;
                const float aw = fabsf(w_new);
                w_max = aw > w_max ? aw : w_max;
                k = q + 1;
            }
            if (prof) ph_loop += static_cast<long long>(wall_clock64()) - c2;
        }

        const long long g0 = prof ? static_cast<long long>(wall_clock64()) : 0;
        if (w_max == 0.0f || __fdiv_rn(d_w_max, w_max) < a.cfg.tol || n_iter == max_iter - 1) {
            float dn = 0.0f;       // the target itself contributes XtA = 0, so the maximum is >= 0
            float R_norm2, Ry, w_norm2 = 0.0f, l1 = 0.0f;
            if (!dirty) {
                // XtA[p] = s[p] for every feature (0 for untouched ones and the target)
                for (int tb = 0; tb < tc; tb += 64) {
                    float v = 0.0f;
                    if (tb + lane < tc) { v = s[touched[tb + lane]]; v = positive ? v : fabsf(v); }
                    const float mx = wave_max(v);
                    dn = mx > dn ? mx : dn;
                }
                R_norm2 = yy; Ry = yy;
            } else {
                // short columns: exact per lane; long columns: listed, screened, folded if they can win
                float lmax = 0.0f;
                int n_long = 0;
                for (int p0 = 0; p0 < I; p0 += 64) {
                    const int p = p0 + lane;
                    int b = 0, e = 0;
                    if (p < I && p != j) { b = a.cptr[p]; e = a.cptr[p + 1]; }
                    const bool is_long = (e - b) > a.lane_max;
                    if (e > b && !is_long) {
                        const float xta = __fsub_rn(lane_dot(a.crow, a.cval, R, b, e, 0.0f), __fmul_rn(beta, w_all[p]));
                        const float v = positive ? xta : fabsf(xta);
                        lmax = v > lmax ? v : lmax;
                    }
                    const unsigned long long m = __ballot(is_long);
                    if (m) {
                        if (is_long) long_list[n_long + lane_prefix(m)] = p;
                        n_long += __builtin_popcountll(m);
                    }
                }
                dn = wave_max(lmax);
                float best_lo = dn;
                for (int pass = 0; pass < 2; ++pass) {
                    for (int q = 0; q < n_long; ++q) {
                        const int p = long_list[q];
                        const int b = a.cptr[p], e = a.cptr[p + 1];
                        const float bw = __fmul_rn(beta, w_all[p]);
                        float lo = ninf, hi = __builtin_huge_valf();
                        if (e - b >= a.screen_min && e - b <= kScreenMaxLen) {
                            float ps, pa;
                            screen_pass(a.crow, a.cval, R, b, e, ps, pa);
                            screen_xta_interval(screen_interval(ps, pa, e - b), bw, positive, lo, hi);
                        }
                        if (pass == 0) { best_lo = lo > best_lo ? lo : best_lo; continue; }
                        if (!(hi >= best_lo)) continue;
                        float v = lo;
                        if (lo != hi) {
                            const float xta = __fsub_rn(xta_pass<SPEC>(a.crow, a.cval, R, b, e, fold_buf, a.spec_min), bw);
                            tr_folded += e - b;
                            v = positive ? xta : fabsf(xta);
                        }
                        dn = v > dn ? v : dn;
                    }
                }
                R_norm2 = 0.0f;
                for (int o = 0; o < U; o += 64) {
                    float prod = 0.0f;
                    if (o + lane < U) { const float v = R[o + lane]; prod = __fmul_rn(v, v); }
                    if (__ballot(prod != 0.0f)) R_norm2 = chain_add(R_norm2, prod, min(64, U - o));
                }
                Ry = 0.0f;
                for (int o = yb; o < ye; o += 64) {
                    const int n = min(64, ye - o);
                    float prod = 0.0f;
                    if (lane < n) prod = __fmul_rn(R[a.crow[o + lane]], a.cval[o + lane]);
                    Ry = chain_add(Ry, prod, n);
                }
                for (int o = 0; o < nf; o += 64) {
                    float wv = 0.0f;
                    if (o + lane < nf) wv = w_all[o + lane];
                    if (__ballot(wv != 0.0f)) {
                        const int n = min(64, nf - o);
                        w_norm2 = chain_add(w_norm2, __fmul_rn(wv, wv), n);
                        l1 = chain_add(l1, fabsf(wv), n);
                    }
                }
            }
            float cst;
            if (dn > alpha) {
                cst = __fdiv_rn(alpha, dn);
                const float A_norm2 = __fmul_rn(R_norm2, __fmul_rn(cst, cst));
                gap = static_cast<float>(0.5 * static_cast<double>(__fadd_rn(R_norm2, A_norm2)));
            } else {
                cst = 1.0f;
                gap = R_norm2;
            }
            const float t12 = __fsub_rn(__fmul_rn(alpha, l1), __fmul_rn(cst, Ry));
            const double t3 = 0.5 * static_cast<double>(beta) * static_cast<double>(__fadd_rn(1.0f, __fmul_rn(cst, cst))) *
                              static_cast<double>(w_norm2);
            gap = static_cast<float>(static_cast<double>(gap) + (static_cast<double>(t12) + t3));
            if (prof) ph_gap += static_cast<long long>(wall_clock64()) - g0;
            if (gap < tol_s) break;
        }
    }
    const int n_iter_out = (n_iter < max_iter ? n_iter : max_iter - 1) + 1;

    // ---- outputs: non-zero coefficients ascending by id; n_ever is small, so rank the ever-list.
    //      Only the first `cap` are stored; out_count reports all of them (the caller refits a
    //      target whose count exceeds its cap with a larger one) ----
    for (int k = lane; k < n_ever; k += 64) {
        const int p = ever_list[k];
        const float wv = w_all[p];
        if (wv != 0.0f) {
            int rank = 0;
            for (int q = 0; q < n_ever; ++q) {
                const int pq = ever_list[q];
                if (pq < p && w_all[pq] != 0.0f) rank++;
            }
            if (rank < a.cap) { oi[rank] = p; oc[rank] = wv; }
        }
    }
    int cnt = 0;
    for (int k = 0; k < n_ever; k += 64) {
        bool nz = false;
        if (k + lane < n_ever) nz = (w_all[ever_list[k + lane]] != 0.0f);
        cnt += __builtin_popcountll(__ballot(nz));
    }
    if (lane == 0) { a.out_count[t] = cnt; a.out_niter[t] = n_iter_out; }

    // ---- restore the scratch invariants ----
    if (dirty) {
        for (int o = yb + lane; o < ye; o += 64) R[a.crow[o]] = 0.0f;
        for (int k = 0; k < n_ever; ++k) {
            const int p = ever_list[k];
            for (int o = a.cptr[p] + lane; o < a.cptr[p + 1]; o += 64) R[a.crow[o]] = 0.0f;
        }
    }
    for (int k = lane; k < n_ever; k += 64) { const int p = ever_list[k]; w_all[p] = 0.0f; ever_flag[p] = 0; }
    for (int tt = lane; tt < tc; tt += 64) s[touched[tt]] = __uint_as_float(kUntouched);
    if (a.trace && lane == 0) {
        long long *tr = a.trace + static_cast<size_t>(t) * 8;
        tr[0] = tr0; tr[1] = tr1; tr[2] = static_cast<long long>(wall_clock64()); tr[3] = tr_folded;
        tr[4] = ph_rng; tr[5] = ph_dots; tr[6] = ph_loop; tr[7] = ph_gap;
    }
}

// SPEC: long columns may fold by integer prefix sums (csrc/fold_spec.hip.h); the chain-only form is a kernel of its own
template <bool ALLF, bool SPEC>
__global__ __launch_bounds__(64, FIT_WAVES_PER_SIMD) void fit_columns_kernel(FitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slot = blockIdx.x;
    for (;;) {
        int t = 0;
        if (lane_id() == 0) t = atomicAdd(a.queue, 1);
        t = readfirst_i(t);
        if (t >= a.n_targets) return;
        if (ALLF) fit_one_allf<SPEC>(a, t, slot, smem);
        else fit_one<false, SPEC>(a, t, slot, smem);
    }
}

// =============================================================================================
// Latency mode (K path): one multi-wave workgroup per target, used when a call fits few targets
// (online partial_fit: a mini-batch touches ~1k items and the heaviest one is the critical path).
//
// The only sequential part of a coordinate update is folding the products R[r]*x[r] left to right.
// Wave 0 (consumer) does nothing else: it pops 64-product chunks from an LDS ring in order and adds
// them one by one.  The other waves (producers) run ahead: coalesced loads of the column, gather
// of R, add-back, multiply, push to the ring (each keeps kProdDepth chunks in flight).  The
// element-wise residual update, the R = y materialisation and the scratch reset are split over
// all threads.  Results are bit-identical to the single-wave kernel.
// =============================================================================================
#ifndef MW_FOLD_GROUP
#define MW_FOLD_GROUP 4
#endif
constexpr int kFoldGroup = MW_FOLD_GROUP;   // chunks the consumer folds per loop iteration (power of two)
constexpr int kMwWaves = 8;
constexpr int kMwThreads = kMwWaves * 64;
#ifndef MW_IDLE_WAVE4
#define MW_IDLE_WAVE4 0
#endif
// MW_IDLE_WAVE4 (A/B): wave 4 shares SIMD 0 with the consumer (waves go to SIMD wave % 4); it then takes no part in the folds
constexpr int kProducers = MW_IDLE_WAVE4 ? kMwWaves - 2 : kMwWaves - 1;
constexpr int kProdDepth = 8;    // chunks a producer wave keeps in flight
constexpr int kMwMaxTargets = 2048;  // calls with at most this many targets use the multi-wave kernel
constexpr int kColWalkMinRows = 1024;  // targets with at least this many users take the column-walk X^T y
constexpr int kRing = 128;       // ring slots of 64 products (>= kProducers * kProdDepth, power of 2)
#ifndef MW_CHAIN_ALL_LANES
#define MW_CHAIN_ALL_LANES 0
#endif
#ifndef MW_SPEC_GROUPS
#define MW_SPEC_GROUPS 4
#endif
constexpr int kSpecGroups = MW_SPEC_GROUPS;   // 256-entry groups the speculative consumer folds per turn

struct MwLds {
    float *ring;   // [kRing][64]
    int *ready;    // [kRing]  tag = global chunk number + 1
    int *done;     // [1]      chunks consumed so far
    float *bc_f;   // [8]      broadcast floats
    int *bc_i;     // [8]      broadcast ints
    float *red;    // [2 * kMwWaves] per-wave partial sums of a screening pass
};
__host__ __device__ constexpr size_t mw_lds_bytes(int K) {
    return ((feat_lds_bytes(K) + 15) / 16) * 16 + kRing * 64 * 4 + kRing * 4 + 16 + 32 + 32 + 2 * kMwWaves * 4;
}
__device__ __forceinline__ MwLds carve_mw(unsigned char *smem, int K) {
    MwLds M;
    unsigned char *p = smem + ((feat_lds_bytes(K) + 15) / 16) * 16;
    M.ring = reinterpret_cast<float *>(p);  p += kRing * 64 * 4;
    M.ready = reinterpret_cast<int *>(p);   p += kRing * 4;
    M.done = reinterpret_cast<int *>(p);    p += 16;
    M.bc_f = reinterpret_cast<float *>(p);  p += 32;
    M.bc_i = reinterpret_cast<int *>(p);    p += 32;
    M.red = reinterpret_cast<float *>(p);
    return M;
}

__device__ __forceinline__ int lds_load_acquire(int *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store_release(int *p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Ordered fold of one stream of 64-element chunks.
//   MODE 0: sum over column entries [b, e) of (R[row] (+ x*w_old)) * x      (dot / XtA)
//   MODE 1: sum over r in [b, e) of R[r]*R[r]                                (R . R)
// Chunks are padded with +0.0 products, which never change the sum.  `seq` is the workgroup-wide
// running chunk number (identical in every wave).  Returns the sum in wave 0, 0 elsewhere.
template <int MODE, bool SPEC>
__device__ float mw_fold(const int *__restrict__ crow, const float *__restrict__ cval, const float *R, const MwLds &M,
                         int b, int e, float w_old, int wave, int lane, int &seq, int spec_min, long long *wait_ticks = nullptr,
                         float *stash = nullptr) {
    // padded to whole groups of kFoldGroup chunks: the extra products are +0.0 and never change the sum
    const int n_chunks = (((e - b + 63) >> 6) + kFoldGroup - 1) & ~(kFoldGroup - 1);
    float tmp = 0.0f;
    if (wave == 0) {
        // The chain of dependent v_add_f32 is the critical path of a popular target (~1e8 entries), so
        // nothing else may sit on it: the ring is read half a chunk (8 x ds_read_b128, uniform address =
        // LDS broadcast) AHEAD of the adds, the ready flag of the next chunk is requested before the
        // adds of this one and tested after them, and `done` is published every fourth chunk.
        // Consumer: lanes 0..15 hold a chunk (one ds_read_b128 each = ONE LDS instruction per chunk), lane 0
        // folds it with 64 DPP adds (chain64_dpp).  A wave issues in order, so every other instruction
        // sits on the chain: the loop therefore works in groups of four chunks (n_chunks is padded to a
        // multiple of four with +0.0 products) -- the ring reads of group g+1 and the ready flags of group
        // g+2 are requested before the 256 adds of group g, and `done` is published once per group.
        __builtin_amdgcn_s_setprio(3);
        if (SPEC && n_chunks > 0 && e - b >= spec_min) {
            // Binade-speculative consumer (csrc/fold_spec.hip.h): four chunks are 256 consecutive products in the ring (groups
            // are aligned: seq and n_chunks are multiples of four); lane L reads products 4L .. 4L+3 of a group with one
            // ds_read_b128 and the wave folds kSpecGroups groups per turn -- one integer scan per group instead of 256 dependent
            // additions, the running value in an SGPR from group to group.  A stream whose chunk count is not a multiple of
            // 4 * kSpecGroups ends with groups the producers never fill: those are not waited for and fold as +0.0.
            static_assert(kFoldGroup == 4, "fold_groups_spec consumes groups of 4 x 64 products");
            constexpr int SG = kSpecGroups, Cp = 4 * SG;                       // chunks per turn
            const float4 *ring4 = reinterpret_cast<const float4 *>(M.ring);
            auto slot_of = [&](int c) { return (seq + c) & (kRing - 1); };
            auto wait_chunks = [&](int c0) {                                   // chunks [c0, c0 + Cp) below n_chunks are ready
                for (;;) {
                    bool bad = false;
#pragma unroll
                    for (int k = 0; k < Cp; k += 4)       // one flag per group of four chunks: tag = global group number + 1
                        if (c0 + k < n_chunks)
                            bad |= (__hip_atomic_load(&M.ready[slot_of(c0 + k) >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ((seq + c0 + k) >> 2) + 1);
                    if (!bad) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            };
            auto read_groups = [&](int c0, float (&P)[SG][4]) {
#pragma unroll
                for (int g = 0; g < SG; ++g) {
                    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (c0 + 4 * g < n_chunks) v = ring4[slot_of(c0 + 4 * g) * 16 + lane];
                    P[g][0] = v.x; P[g][1] = v.y; P[g][2] = v.z; P[g][3] = v.w;
                }
            };
            float A[SG][4], B[SG][4];
            wait_chunks(0);
            read_groups(0, A);
            for (int c = 0; c < n_chunks; c += Cp) {
                const bool more = c + Cp < n_chunks;
                if (more) {
                    if (wait_ticks) {
                        const long long w0 = static_cast<long long>(wall_clock64());
                        wait_chunks(c + Cp);
                        *wait_ticks += static_cast<long long>(wall_clock64()) - w0;
                    } else {
                        wait_chunks(c + Cp);
                    }
                    read_groups(c + Cp, B);                                    // in flight while this turn's groups are folded
                }
                tmp = fold_groups_spec<SG>(tmp, A);
                if (more) {
#pragma unroll
                    for (int g = 0; g < SG; ++g) { A[g][0] = B[g][0]; A[g][1] = B[g][1]; A[g][2] = B[g][2]; A[g][3] = B[g][3]; }
                }
                if (lane == 0) lds_store_release(M.done, seq + min(c + Cp, n_chunks));
            }
        } else if (n_chunks > 0 && (MW_CHAIN_ALL_LANES || lane < 16)) {
            // A group's four chunks are four consecutive ring slots (seq and every group start are multiples of four, the ring
            // holds 128): one base address per group for the products and one for the flags.  Two groups per loop turn with the
            // registers of A and B swapping roles (round 5: the sixteen v_mov of `A = B`, the per-chunk slot arithmetic, one loop
            // branch and one `done` store per group were a quarter of the instructions between two chains -- every one of them
            // sits ON the chain of a wave that issues in order).
            constexpr int Gp = kFoldGroup;
            static_assert(Gp == 4 && (kRing & 3) == 0, "a group is four consecutive ring slots");
            const float4 *ring4 = reinterpret_cast<const float4 *>(M.ring);
            const int l15 = lane & 15;
            const int sq = __builtin_amdgcn_readfirstlane(seq);     // uniform: the slot arithmetic stays on the scalar unit
            float4 A[Gp], B[Gp];
            // one ready flag per GROUP (a producer wave owns whole groups): tag = global group number + 1
            const int gq = sq >> 2;
            int f;
            auto read_flag = [&](int g) {
                f = __hip_atomic_load(&M.ready[(gq + g) & (kRing / 4 - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            {
                const int nb = sq & (kRing - 1);
                read_flag(0);
                while (f != gq + 1) { __builtin_amdgcn_s_sleep(1); read_flag(0); }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
                for (int k = 0; k < Gp; ++k) A[k] = ring4[(nb + k) * 16 + l15];
                read_flag(1);
            }
            // folds the group held in X (group g: chunks 4g .. 4g+3) after requesting group g+1 into Y and the flag of g+2.
            // A taken branch costs a wave that runs alone an instruction refetch, so the expected path (the flag is set, more
            // groups follow) is laid out as the fall-through and the spin loop out of line.
            const int n_groups = n_chunks >> 2;
            auto step = [&](auto sure, float4 (&X)[Gp], float4 (&Y)[Gp], int g) {      // sure: group g+1 exists, no test
                if (decltype(sure)::value || g + 1 < n_groups) {
                    if (__builtin_expect(f != gq + g + 2, 0)) {
#ifdef MW_TRACE_WAIT      // diagnostic build: ticks the chain consumer waits for its producers (tr[7] of the trace)
                        const long long w0 = static_cast<long long>(wall_clock64());
#endif
                        do { __builtin_amdgcn_s_sleep(1); read_flag(g + 1); } while (f != gq + g + 2);
#ifdef MW_TRACE_WAIT
                        if (wait_ticks) *wait_ticks += static_cast<long long>(wall_clock64()) - w0;
#endif
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const float4 *src = ring4 + ((sq + 4 * g + 4) & (kRing - 1)) * 16 + l15;
#pragma unroll
                for (int k = 0; k < Gp; ++k) Y[k] = src[k * 16];                 // past the end: stale, unused
                read_flag(g + 2);
#pragma unroll
                for (int k = 0; k < Gp; ++k) tmp = chain64_dpp(tmp, X[k]);
            };
            int g = 0;
            for (; g + 2 < n_groups; g += 2) {            // both groups of the turn have a successor: no tests on the chain
                step(std::true_type{}, A, B, g);
                step(std::true_type{}, B, A, g + 1);
                if (lane == 0) lds_store_release(M.done, sq + 4 * (g + 2));
            }
            step(std::false_type{}, A, B, g);             // the last one or two groups
            if (g + 1 < n_groups) step(std::false_type{}, B, A, g + 1);
            if (lane == 0) lds_store_release(M.done, sq + n_chunks);
        }
        tmp = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(tmp)));
        __builtin_amdgcn_s_setprio(0);
    } else {
        // Producers are latency machines: a round is one index load and one dependent gather, so the
        // indices of round k+1 are requested before the gathers of round k are consumed (one memory
        // round trip per round instead of two) and a round carries kProdDepth chunks per wave.
        // A wave owns whole groups of four chunks (two per round): the consumer then tests ONE ready flag per group.
        static_assert(kProdDepth == 8 && kFoldGroup == 4, "a producer round is two groups of four chunks");
        const bool add_back = (w_old != 0.0f);
        constexpr int kGroupStride = kProducers * (kProdDepth / 4);       // groups per round of all producers
        const int n_groups = n_chunks >> 2;
        auto chunk_of = [&](int g0, int u) { return 4 * (g0 + kProducers * (u >> 2)) + (u & 3); };   // < n_chunks iff its group < n_groups
        int rr[kProdDepth], rn[kProdDepth];
        float xx[kProdDepth], xn[kProdDepth];
        auto load_idx = [&](int g0, int (&r)[kProdDepth], float (&x)[kProdDepth]) {
#pragma unroll
            for (int u = 0; u < kProdDepth; ++u) {
                const int c = chunk_of(g0, u);
                const int o = b + c * 64 + lane;
                r[u] = -1; x[u] = 0.0f;
                if (c < n_chunks && o < e) {
                    if (MODE == 0) { r[u] = crow[o]; x[u] = cval[o]; }
                    else r[u] = o;
                }
            }
        };
        int g0 = MW_IDLE_WAVE4 ? (wave < 4 ? wave - 1 : (wave == 4 ? n_groups : wave - 2)) : wave - 1;
        if (g0 < n_groups) load_idx(g0, rr, xx);
        for (; g0 < n_groups; g0 += kGroupStride) {
            float prod[kProdDepth];
#pragma unroll
            for (int u = 0; u < kProdDepth; ++u) prod[u] = (rr[u] >= 0) ? R[rr[u]] : 0.0f;     // gathers
            if (g0 + kGroupStride < n_groups) load_idx(g0 + kGroupStride, rn, xn);
#pragma unroll
            for (int u = 0; u < kProdDepth; ++u) {
                if (MODE == 0) {
                    float v = prod[u];
                    if (add_back) v = __fadd_rn(v, __fmul_rn(xx[u], w_old));
                    // the residual update that follows this fold starts from exactly this value: kept (coalesced, fire and
                    // forget) so that the update streams it instead of gathering R again (mw_update_stashed)
                    if (stash && rr[u] >= 0) stash[chunk_of(g0, u) * 64 + lane] = v;
                    prod[u] = (rr[u] >= 0) ? __fmul_rn(v, xx[u]) : 0.0f;
                } else {
                    prod[u] = __fmul_rn(prod[u], prod[u]);
                }
            }
#pragma unroll
            for (int h = 0; h < kProdDepth / 4; ++h) {
                const int c = chunk_of(g0, 4 * h);
                if (c >= n_chunks) break;
                const int G = seq + c, slot = G & (kRing - 1);                            // the group's first chunk and ring slot
                while (lds_load_acquire(M.done) <= G + 3 - kRing) __builtin_amdgcn_s_sleep(1);   // its slots are still in use
#pragma unroll
                for (int k = 0; k < 4; ++k) M.ring[(slot + k) * 64 + lane] = prod[4 * h + k];
                if (lane == 0) lds_store_release(&M.ready[slot >> 2], (G >> 2) + 1);
            }
#pragma unroll
            for (int u = 0; u < kProdDepth; ++u) { rr[u] = rn[u]; xx[u] = xn[u]; }
        }
    }
    seq += n_chunks;
    return tmp;
}

// R[r] <- (R[r] + x*w_old) - x*w_new over the column by all threads of the workgroup (element-wise,
// order free).  Latency-bound like the producers: kUpdDepth entries per thread and round, and the
// indices of the next round are requested before the gathers of this one are consumed.
constexpr int kUpdDepth = 8;   // 16 shortens the update by 10 % and lengthens the fold by 3 % (register pressure): no gain
__device__ void mw_update(const int *__restrict__ crow, const float *__restrict__ cval, float *R, int b, int e,
                          float w_old, float w_new, int tid) {
    int r[kUpdDepth], rn[kUpdDepth];
    float x[kUpdDepth], xn[kUpdDepth];
    auto load_idx = [&](int o0, int (&rr)[kUpdDepth], float (&xx)[kUpdDepth]) {
#pragma unroll
        for (int k = 0; k < kUpdDepth; ++k) {
            const int o = o0 + k * kMwThreads;
            rr[k] = -1; xx[k] = 0.0f;
            if (o < e) { rr[k] = crow[o]; xx[k] = cval[o]; }
        }
    };
    int o = b + tid;
    if (o < e) load_idx(o, r, x);
    for (; o < e; o += kUpdDepth * kMwThreads) {
        float v[kUpdDepth];
#pragma unroll
        for (int k = 0; k < kUpdDepth; ++k) v[k] = (r[k] >= 0) ? R[r[k]] : 0.0f;
        if (o + kUpdDepth * kMwThreads < e) load_idx(o + kUpdDepth * kMwThreads, rn, xn);
#pragma unroll
        for (int k = 0; k < kUpdDepth; ++k) {
            if (r[k] < 0) continue;
            float t = v[k];
            if (w_old != 0.0f) t = __fadd_rn(t, __fmul_rn(x[k], w_old));
            if (w_new != 0.0f) t = __fsub_rn(t, __fmul_rn(x[k], w_new));
            R[r[k]] = t;
        }
#pragma unroll
        for (int k = 0; k < kUpdDepth; ++k) { r[k] = rn[k]; x[k] = xn[k]; }
    }
}

// The same update after mw_fold<0> of the same column left v = R[r] (+ x*w_old) in stash[o - b]: R[r] <- v (- x*w_new).
// Three coalesced streams and a scatter, no dependent gather: the update of a popular column is one memory round trip
// per kUpdDepth * kMwThreads entries instead of two (round 5: 25 % of the heaviest target of a mini-batch was this pass).
__device__ void mw_update_stashed(const int *__restrict__ crow, const float *__restrict__ cval, const float *__restrict__ stash,
                                  float *R, int b, int e, float w_new, int tid) {
    // Whole rounds (kUpdDepth * kMwThreads entries) are straight-line code: per-entry bounds tests put every load and store in
    // a basic block of its own, and the compiler then waits for ALL outstanding memory operations -- the previous store
    // included -- before each store (eight serial store round trips per round; measured 37 us per 52 k-entry update).
    // The loads of round k+1 are issued BEFORE the scatter of round k: memory operations retire in order, so waiting for
    // those loads leaves the stores in flight.
    constexpr int kRound = kUpdDepth * kMwThreads;
    const int full = (e - b) / kRound;
    crow += b + tid; cval += b + tid; stash += tid;
    int r[kUpdDepth], rn[kUpdDepth];
    float x[kUpdDepth], xn[kUpdDepth], v[kUpdDepth], vn[kUpdDepth];
    auto load = [&](int i, int (&rr)[kUpdDepth], float (&xx)[kUpdDepth], float (&vv)[kUpdDepth]) {
#pragma unroll
        for (int k = 0; k < kUpdDepth; ++k) {
            const int oo = i * kRound + k * kMwThreads;
            rr[k] = crow[oo]; xx[k] = cval[oo]; vv[k] = stash[oo];
        }
    };
    auto scatter = [&](const int (&rr)[kUpdDepth], const float (&xx)[kUpdDepth], const float (&vv)[kUpdDepth]) {
#pragma unroll
        for (int k = 0; k < kUpdDepth; ++k) R[rr[k]] = (w_new != 0.0f) ? __fsub_rn(vv[k], __fmul_rn(xx[k], w_new)) : vv[k];
    };
    if (full > 0) {
        load(0, r, x, v);
        for (int i = 1; i < full; ++i) {
            load(i, rn, xn, vn);
            scatter(r, x, v);
#pragma unroll
            for (int k = 0; k < kUpdDepth; ++k) { r[k] = rn[k]; x[k] = xn[k]; v[k] = vn[k]; }
        }
        scatter(r, x, v);
    }
    const int rest = (e - b) - full * kRound - tid;     // entries of the last, partial round from this thread on
#pragma unroll
    for (int k = 0; k < kUpdDepth; ++k) {
        const int oo = full * kRound + k * kMwThreads;
        if (k * kMwThreads < rest) { r[k] = crow[oo]; x[k] = cval[oo]; v[k] = stash[oo]; }
    }
#pragma unroll
    for (int k = 0; k < kUpdDepth; ++k)
        if (k * kMwThreads < rest) R[r[k]] = (w_new != 0.0f) ? __fsub_rn(v[k], __fmul_rn(x[k], w_new)) : v[k];
}

// Screening pass (see screen_pass) by all threads of the workgroup; every thread returns the
// same interval.  Contains two barriers.
__device__ Screen mw_screen(const int *__restrict__ crow, const float *__restrict__ cval, const float *R,
                            const MwLds &M, int b, int e, int tid) {
    float s0 = 0.0f, a0 = 0.0f;
    for (int o = b + tid; o < e; o += 4 * kMwThreads) {
        int r[4];
        float x[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int oo = o + u * kMwThreads;
            r[u] = -1; x[u] = 0.0f;
            if (oo < e) { r[u] = crow[oo]; x[u] = cval[oo]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (r[u] >= 0) ? R[r[u]] : 0.0f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float p = __fmul_rn(v[u], x[u]);
            s0 = __fadd_rn(s0, p);
            a0 = __fadd_rn(a0, fabsf(p));
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s0 = __fadd_rn(s0, shfl_xor_t(s0, m));
        a0 = __fadd_rn(a0, shfl_xor_t(a0, m));
    }
    if ((tid & 63) == 0) { M.red[2 * (tid >> 6)] = s0; M.red[2 * (tid >> 6) + 1] = a0; }
    __syncthreads();
    float ps = 0.0f, pa = 0.0f;
#pragma unroll
    for (int w = 0; w < kMwWaves; ++w) { ps = __fadd_rn(ps, M.red[2 * w]); pa = __fadd_rn(pa, M.red[2 * w + 1]); }
    __syncthreads();
    return screen_interval(ps, pa, e - b);
}

// ---------------------------------------------------------------------------------------------
// Batched X^T y for a call of few targets (online partial_fit): ONE pass over X serves every target.
//   s_t[i] = sum over users u (ascending) of X[u, i] * X[u, t]        (csr_matvec order, the target's own item masked)
// Per target this is a walk over the users of t and their rows (prep_target) or over all of X (column walk): for a
// mini-batch of ~850 popular items that is 850 passes over a 170 MB matrix.  Here a wave owns an item column i, keeps
// s_.[i] for ALL targets of the call in LDS (n_t floats) and streams the column once: for entry (u, x) it adds x * y to
// the sums of the targets user u has rated -- read from a compacted copy of the rows restricted to the call's targets
// (xty_compact_kernel).  Users ascend along a column, and a user's targets are distinct, so every (i, t) sum sees its
// products in exactly the reference order.  Non-zero sums are appended to the target's candidate list (cand_i, cand_s).
// ---------------------------------------------------------------------------------------------
constexpr int kXtyWaves = 8;
constexpr int kXtyMaxTargets = 2048;     // = kMwMaxTargets: 8 KiB of LDS per wave
constexpr int kXtyBatch = 16;            // users whose target lists are fetched together (memory-level parallelism of a wave)

struct XtyArgs {
    int U, I, n_t;
    const int *cptr; const int *crow; const float *cval;
    const int *rptr; const int *rcol; const float *rval;
    const int *col_order;
    const int *targets;
    int *tmap;              // [I]  item -> index in `targets` or -1
    int *ypos; int *ylen;   // [U]  the user's compacted target list in yt / yv
    int *yt; float *yv;     // [<= nnz]
    int *cursor;            // [2]  compaction cursor, column queue
    int *cand_cnt; int *cand_i; float *cand_s;
};

// S[t] += p as one LDS float atomic without return (ds_add_f32): the LDS applies a wave's atomics in issue order
// with one IEEE round-to-nearest add each, so the sum is the same left-to-right fold a read-add-write would give,
// without the wave waiting a round trip per user.
__device__ __forceinline__ void lds_add_f32(float *p, float v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(256) void xty_tmap_kernel(XtyArgs a) {
    const int g = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
    if (g < a.n_t) { a.tmap[a.targets[g]] = g; a.cand_cnt[g] = 0; }
    if (g == 0) { a.cursor[0] = 0; a.cursor[1] = 0; }
}

// one wave per user row: keep the entries whose item is a target of the call
__global__ __launch_bounds__(256) void xty_compact_kernel(XtyArgs a) {
    const int lane = lane_id();
    const int u = static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6);
    if (u >= a.U) return;
    const int rb = a.rptr[u], re = a.rptr[u + 1];
    int cnt = 0;
    for (int o = rb; o < re; o += 64) cnt += __builtin_popcountll(__ballot(o + lane < re && a.tmap[a.rcol[o + lane]] >= 0));
    int pos = 0;
    if (lane == 0 && cnt > 0) pos = atomicAdd(&a.cursor[0], cnt);
    pos = readfirst_i(pos);
    if (lane == 0) { a.ypos[u] = pos; a.ylen[u] = cnt; }
    for (int o = rb; o < re && cnt > 0; o += 64) {
        int t = -1;
        float y = 0.0f;
        if (o + lane < re) { t = a.tmap[a.rcol[o + lane]]; y = a.rval[o + lane]; }
        const unsigned long long m = __ballot(t >= 0);
        if (t >= 0) { const int q = pos + lane_prefix(m); a.yt[q] = t; a.yv[q] = y; }
        pos += __builtin_popcountll(m);
    }
}

__global__ __launch_bounds__(kXtyWaves * 64) void xty_batch_kernel(XtyArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = lane_id(), wave = static_cast<int>(threadIdx.x >> 6);
    float *S = reinterpret_cast<float *>(smem) + static_cast<size_t>(wave) * a.n_t;
    for (int t = lane; t < a.n_t; t += 64) S[t] = 0.0f;
    for (;;) {
        int pos = 0;
        if (lane == 0) pos = atomicAdd(&a.cursor[1], 1);
        pos = readfirst_i(pos);
        if (pos >= a.I) break;
        const int i = a.col_order ? a.col_order[pos] : pos;
        const int b = a.cptr[i], e = a.cptr[i + 1];
        if (b == e) continue;
        const int t_self = a.tmap[i];                      // fitting item i masks its own column
        bool any = false;
        for (int ob = b; ob < e; ob += 64) {
            const int n = min(64, e - ob);
            int yp_l = 0, yl_l = 0;
            float x_l = 0.0f;
            if (lane < n) {
                const int u = a.crow[ob + lane];
                x_l = a.cval[ob + lane];
                yp_l = a.ypos[u];
                yl_l = a.ylen[u];
            }
            unsigned long long live = __ballot(yl_l > 0);       // users that rated one of the targets
            while (live) {
                // the target lists of the next kXtyBatch users are fetched together, then applied user after user
                int q[kXtyBatch], len[kXtyBatch], tt[kXtyBatch];
                float yy[kXtyBatch], xx[kXtyBatch];
#pragma unroll
                for (int k = 0; k < kXtyBatch; ++k) {
                    q[k] = live ? __builtin_ctzll(live) : -1;
                    live &= live - 1;
                    len[k] = 0; tt[k] = -1; yy[k] = 0.0f; xx[k] = 0.0f;
                    if (q[k] >= 0) {
                        const int p0 = readlane_i(yp_l, q[k]);
                        len[k] = readlane_i(yl_l, q[k]);
                        xx[k] = readlane_f(x_l, q[k]);
                        if (lane < len[k]) { tt[k] = a.yt[p0 + lane]; yy[k] = a.yv[p0 + lane]; }
                    }
                }
#pragma unroll
                for (int k = 0; k < kXtyBatch; ++k) {
                    if (q[k] < 0) break;
                    any = true;
                    if (tt[k] >= 0 && tt[k] != t_self) lds_add_f32(&S[tt[k]], __fmul_rn(xx[k], yy[k]));
                    if (len[k] > 64) {                          // a user with more than 64 of the call's targets
                        const int p0 = readlane_i(yp_l, q[k]);
                        for (int c = 64; c < len[k]; c += 64) {
                            if (c + lane < len[k]) {
                                const int t2 = a.yt[p0 + c + lane];
                                if (t2 != t_self) lds_add_f32(&S[t2], __fmul_rn(xx[k], a.yv[p0 + c + lane]));
                            }
                        }
                    }
                }
            }
        }
        if (!any) continue;
        // non-zero sums -> the targets' candidate lists; the accumulators return to +0
        for (int tb = 0; tb < a.n_t; tb += 64) {
            const int t = tb + lane;
            float v = 0.0f;
            if (t < a.n_t) v = S[t];
            if (v != 0.0f) {
                const int slot = atomicAdd(&a.cand_cnt[t], 1);
                a.cand_i[static_cast<size_t>(t) * a.I + slot] = i;
                a.cand_s[static_cast<size_t>(t) * a.I + slot] = v;
                S[t] = 0.0f;
            }
        }
    }
}

// X^T y by COLUMN walk, all waves of the workgroup (latency mode, popular targets).
// The row walk of prep_target visits the rows of U_j one after the other (two dependent memory
// round trips each: 70k rows -> 0.3 s for the most popular item).  Here y is first scattered into
// the dense residual buffer (R = y, which the coordinate descent needs anyway) and every wave
// takes whole item columns: 64 coalesced entries per step, gather y[row], and the products with
// y != 0 are folded in ascending row order -- exactly csr_matvec's order for that item.  Cost:
// one streaming pass over X plus |U_i ^ U_j| dependent adds per item, spread over the waves.
__device__ int xty_colwalk_mw(const FitArgs &a, int j, float *R, float *s, int *touched, int *lds_tc,
                              int wave, int lane, int n_waves) {
    const int I = a.I;
    const int yb = a.cptr[j], ye = a.cptr[j + 1];
    for (int o = yb + static_cast<int>(threadIdx.x); o < ye; o += n_waves * 64) R[a.crow[o]] = a.cval[o];
    if (threadIdx.x == 0) *lds_tc = 0;
    __syncthreads();
    for (int pos = wave; pos < I; pos += n_waves) {
        const int i = a.col_order ? a.col_order[pos] : pos;     // similar lengths run side by side
        if (i == j) continue;
        const int b = a.cptr[i], e = a.cptr[i + 1];
        if (b == e) continue;
        float sum = 0.0f;
        bool any = false;
        for (int ob = b; ob < e; ob += 256) {
            int r[4];
            float x[4], yv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int o = ob + k * 64 + lane;
                r[k] = -1; x[k] = 0.0f;
                if (o < e) { r[k] = a.crow[o]; x[k] = a.cval[o]; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) yv[k] = (r[k] >= 0) ? R[r[k]] : 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned long long m = __ballot(yv[k] != 0.0f);
                if (m) {
                    any = true;
                    const float prod = __fmul_rn(x[k], yv[k]);
                    while (m) {
                        const int l = __builtin_ctzll(m);
                        m &= m - 1;
                        sum = __fadd_rn(sum, readlane_f(prod, l));
                    }
                }
            }
        }
        if (any && lane == 0) {
            s[i] = sum;                       // (+0) + products: never the -0.0 marker
            touched[atomicAdd(lds_tc, 1)] = i;
        }
    }
    __syncthreads();
    return *lds_tc;
}

template <bool SPEC>
__device__ void fit_one_mw(const FitArgs &a, int t, int slot, unsigned char *smem) {
    const int tid = static_cast<int>(threadIdx.x), wave = tid >> 6, lane = tid & 63;
    const int U = a.U, I = a.I;
    const int j = a.targets[t];
    float *R = a.R + static_cast<size_t>(slot) * U;
    float *stash = a.stash + static_cast<size_t>(slot) * U;
    float *s = a.s + static_cast<size_t>(slot) * I;
    int *touched = a.touched + static_cast<size_t>(slot) * I;
    float *cand_s = a.cand_s + static_cast<size_t>(slot) * I;
    int *cand_i = a.cand_i + static_cast<size_t>(slot) * I;
    const int K = min(a.cfg.top_features, I);
    const FeatLds F = carve_feat(smem, K);
    const MwLds M = carve_mw(smem, K);
    int *f_id = F.f_id, *f_b = F.f_b, *f_e = F.f_e, *f_ever = F.f_ever;
    float *f_nrm = F.f_nrm, *f_w = F.f_w, *f_s = F.f_s;

    const float alpha = a.cfg.l1_reg, beta = a.cfg.l2_reg;
    const int positive = a.cfg.positive;
    const int yb = a.cptr[j], ye = a.cptr[j + 1];
    const int ny = ye - yb;

    const long long tr0 = a.trace ? static_cast<long long>(wall_clock64()) : 0;
    long long tr_folded = 0, ph_fold = 0, ph_upd = 0, ph_gap = 0, ph_cyc = 0, ph_wait = 0;
    if (tid < kRing) M.ready[tid] = 0;
    if (tid == 0) *M.done = 0;
    int seq = 0;
    // popular targets: column walk by all waves (R = y is materialised as a by-product)
    const bool pre = a.pre_cnt != nullptr;
    const bool col_walk = !pre && ny >= a.colwalk_min_rows;
    int tc_cw = -1;
    if (col_walk) tc_cw = xty_colwalk_mw(a, j, R, s, touched, &M.bc_i[4], wave, lane, kMwWaves);
    if (wave == 0) {
        const Prep P = prep_target<false>(a, j, K, s, touched, cand_s, cand_i, F, tc_cw, pre ? t : -1);
        if (lane == 0) { M.bc_f[0] = P.yy; M.bc_f[1] = P.tol_s; M.bc_i[0] = P.tc; M.bc_i[1] = P.Kc; }
    }
    __syncthreads();
    const long long tr1 = a.trace ? static_cast<long long>(wall_clock64()) : 0;
    const float yy = M.bc_f[0], tol_s = M.bc_f[1];
    const int tc = M.bc_i[0], Kc = M.bc_i[1];
    const int nf = Kc;

    // tolerance mode (rtrec_fit_opts.fast = 2), all features in the Gram matrix: wave 0 runs the Gram-form
    // coordinate descent (fit_gram_cd: K^2 multiply-adds per sweep, no residual); X^T y and the selection above
    // were exact and multi-wave, which is what a popular target of a small call needs.  Otherwise the ordered
    // folds below run as in exact mode (trivially within the tolerance).
    bool gram_done = false;
    if (a.fast >= 2 && a.gram != nullptr && Kc > 0 && Kc <= 64 && ny > 0) {
        if (wave == 0) {
            int g_lane = -1;
            if (lane < Kc) g_lane = a.gram_index[f_id[lane]];
            const unsigned long long missing = __ballot(lane < Kc && f_nrm[lane] != 0.0f && g_lane < 0);
            int iters = -1;
            if (!missing) {
                float *gs = reinterpret_cast<float *>(smem + ((mw_lds_bytes(K) + 15) / 16) * 16);
                iters = fit_gram_cd(a, Kc, F, gs, yy, tol_s, g_lane);
            }
            if (lane == 0) M.bc_i[3] = iters;
        }
        __syncthreads();
        gram_done = M.bc_i[3] >= 0;
    }

    bool dirty = false;
    bool r_is_y = col_walk;     // R already holds y (column walk): nothing to materialise later
    float gap = __fadd_rn(a.cfg.tol, 1.0f);
    uint32_t rng = a.cfg.seed;
    const int max_iter = a.cfg.max_iter;
    const bool skip_cd = (ny == 0) || (nf == 0) || gram_done;
    int n_iter = skip_cd ? (max_iter > 0 ? max_iter - 1 : 0) : 0;
    if (gram_done) n_iter = M.bc_i[3] - 1;
    int upd = 0;   // parity of the broadcast slot

    for (; !skip_cd && n_iter < max_iter; ++n_iter) {
        float w_max = 0.0f, d_w_max = 0.0f;
        for (int f = 0; f < nf; ++f) {
            const int p = static_cast<int>(rand_int(static_cast<uint32_t>(nf), rng));
            const float nrm = f_nrm[p];
            if (nrm == 0.0f) continue;
            const int b = f_b[p], e = f_e[p];
            const float w_old = f_w[p];
            float w_new = 0.0f;
            bool stashed = false;      // the fold of this draw left R[r] (+ x*w_old) of the column in `stash`
            if (!dirty) {
                w_new = cd_update(f_s[p], alpha, beta, nrm, positive);     // identical in every wave
            } else {
                bool screened = false;
                if (w_old == 0.0f && e - b >= a.screen_min && e - b <= kScreenMaxLen)
                    screened = screen_stays_zero(mw_screen(a.crow, a.cval, R, M, b, e, tid), alpha, positive, w_new);
                if (!screened) {
                    const long long c0 = a.trace ? static_cast<long long>(wall_clock64()) : 0;
                    const long long k0 = a.trace ? static_cast<long long>(__builtin_amdgcn_s_memtime()) : 0;
                    const float tmp = mw_fold<0, SPEC>(a.crow, a.cval, R, M, b, e, w_old, wave, lane, seq, a.spec_min, a.trace ? &ph_wait : nullptr,
                                                       w_old != 0.0f ? stash : nullptr);
                    stashed = w_old != 0.0f;     // a zero coefficient mostly stays zero: nothing to update, nothing kept
                    if (a.trace) { ph_fold += static_cast<long long>(wall_clock64()) - c0; ph_cyc += static_cast<long long>(__builtin_amdgcn_s_memtime()) - k0; }
                    tr_folded += e - b;
                    upd ^= 1;
                    if (tid == 0) M.bc_f[2 + upd] = cd_update(tmp, alpha, beta, nrm, positive);
                    __syncthreads();
                    w_new = M.bc_f[2 + upd];
                }
            }
            const bool changed = __float_as_uint(w_new) != __float_as_uint(w_old);
            const bool touch_r = (w_old != 0.0f || w_new != 0.0f);
            if (!dirty && (changed || touch_r)) __syncthreads();   // every wave has read f_w[p] before it changes
            if (changed && tid == 0) f_w[p] = w_new;
            if (touch_r) {
                if (!dirty) {   // materialise R = y
                    if (!r_is_y) for (int o = yb + tid; o < ye; o += kMwThreads) R[a.crow[o]] = a.cval[o];
                    dirty = true;
                    __syncthreads();
                }
                const long long c0 = a.trace ? static_cast<long long>(wall_clock64()) : 0;
                if (stashed) mw_update_stashed(a.crow, a.cval, stash, R, b, e, w_new, tid);
                else mw_update(a.crow, a.cval, R, b, e, w_old, w_new, tid);
                if (a.trace) { __syncthreads(); ph_upd += static_cast<long long>(wall_clock64()) - c0; }
#ifdef MW_TRACE_UPD
                ph_gap += e - b; ph_wait += 1;
#endif
                if (tid == 0) f_ever[p] = 1;
            }
            if (changed || touch_r) __syncthreads();
            const float d = fabsf(__fsub_rn(w_new, w_old));
            d_w_max = d > d_w_max ? d : d_w_max;
            const float aw = fabsf(w_new);
            w_max = aw > w_max ? aw : w_max;
        }

        const long long g0 = a.trace ? static_cast<long long>(wall_clock64()) : 0;
        if (w_max == 0.0f || __fdiv_rn(d_w_max, w_max) < a.cfg.tol || n_iter == max_iter - 1) {
            float dn = 0.0f;
            bool dn_init = false;
            auto dn_take = [&](float xta) {
                const float v = positive ? xta : fabsf(xta);
                if (!dn_init) { dn = v; dn_init = true; } else if (v > dn) dn = v;
            };
            float R_norm2, Ry, w_norm2 = 0.0f, l1 = 0.0f;
            if (!dirty) {
                for (int p = 0; p < nf; ++p) dn_take(f_nrm[p] != 0.0f ? f_s[p] : 0.0f);
                R_norm2 = yy; Ry = yy;
            } else {
                // screen every column, fold in order only the candidates for the maximum (see fit_one)
                float *x_lo = cand_s, *x_hi = reinterpret_cast<float *>(cand_i);
                float best_lo = -__builtin_huge_valf();
                for (int p = 0; p < nf; ++p) {
                    float lo = 0.0f, hi = 0.0f;
                    if (f_nrm[p] != 0.0f) {
                        const int b = f_b[p], e = f_e[p];
                        const float bw = __fmul_rn(beta, f_w[p]);
                        if (e - b >= a.screen_min && e - b <= kScreenMaxLen) {
                            screen_xta_interval(mw_screen(a.crow, a.cval, R, M, b, e, tid), bw, positive, lo, hi);
                        } else {
                            lo = -__builtin_huge_valf(); hi = __builtin_huge_valf();   // short column: always folded
                        }
                    }
                    if (tid == 0) { x_lo[p] = lo; x_hi[p] = hi; }
                    best_lo = lo > best_lo ? lo : best_lo;
                }
                __syncthreads();
                for (int p = 0; p < nf; ++p) {
                    const float lo = x_lo[p], hi = x_hi[p];
                    if (!(hi >= best_lo)) continue;
                    float v = lo;
                    if (lo != hi) {
                        const float xta = __fsub_rn(mw_fold<0, SPEC>(a.crow, a.cval, R, M, f_b[p], f_e[p], 0.0f, wave, lane, seq, a.spec_min),
                                                    __fmul_rn(beta, f_w[p]));
                        tr_folded += f_e[p] - f_b[p];
                        v = positive ? xta : fabsf(xta);
                    }
                    dn_take(v);
                }
                R_norm2 = mw_fold<1, SPEC>(a.crow, a.cval, R, M, 0, U, 0.0f, wave, lane, seq, a.spec_min);
                Ry = 0.0f;
                if (wave == 0) {   // short folds stay on the consumer wave
                    for (int o = yb; o < ye; o += 64) {
                        const int n = min(64, ye - o);
                        float prod = 0.0f;
                        if (lane < n) prod = __fmul_rn(R[a.crow[o + lane]], a.cval[o + lane]);
                        Ry = chain_add(Ry, prod, n);
                    }
                    for (int o = 0; o < nf; o += 64) {
                        float wv = 0.0f;
                        if (o + lane < nf) wv = f_w[o + lane];
                        if (__ballot(wv != 0.0f)) {
                            const int n = min(64, nf - o);
                            w_norm2 = chain_add(w_norm2, __fmul_rn(wv, wv), n);
                            l1 = chain_add(l1, fabsf(wv), n);
                        }
                    }
                }
            }
            float cst;
            if (dn > alpha) {
                cst = __fdiv_rn(alpha, dn);
                const float A_norm2 = __fmul_rn(R_norm2, __fmul_rn(cst, cst));
                gap = static_cast<float>(0.5 * static_cast<double>(__fadd_rn(R_norm2, A_norm2)));
            } else {
                cst = 1.0f;
                gap = R_norm2;
            }
            const float t12 = __fsub_rn(__fmul_rn(alpha, l1), __fmul_rn(cst, Ry));
            const double t3 = 0.5 * static_cast<double>(beta) * static_cast<double>(__fadd_rn(1.0f, __fmul_rn(cst, cst))) *
                              static_cast<double>(w_norm2);
            gap = static_cast<float>(static_cast<double>(gap) + (static_cast<double>(t12) + t3));
            bool stop = gap < tol_s;
            if (dirty) {   // only wave 0 holds the folded sums: broadcast its decision
                if (tid == 0) M.bc_i[2] = stop ? 1 : 0;
                __syncthreads();
                stop = M.bc_i[2] != 0;
                __syncthreads();
            }
#ifndef MW_TRACE_UPD
            if (a.trace) ph_gap += static_cast<long long>(wall_clock64()) - g0;
#endif
            if (stop) break;
        }
    }
    const int n_iter_out = (n_iter < max_iter ? n_iter : max_iter - 1) + 1;

    int *oi = a.out_items + static_cast<size_t>(t) * a.cap;
    float *oc = a.out_coef + static_cast<size_t>(t) * a.cap;
    for (int p = tid; p < Kc; p += kMwThreads) { oi[p] = f_id[p]; oc[p] = f_w[p]; }
    if (tid == 0) { a.out_count[t] = Kc; a.out_niter[t] = n_iter_out; }

    if (dirty || r_is_y) for (int o = yb + tid; o < ye; o += kMwThreads) R[a.crow[o]] = 0.0f;
    if (dirty) {
        for (int p = 0; p < Kc; ++p) {
            if (f_ever[p] == 0) continue;
            for (int o = f_b[p] + tid; o < f_e[p]; o += kMwThreads) R[a.crow[o]] = 0.0f;
        }
    }
    for (int tt = tid; tt < tc; tt += kMwThreads) s[touched[tt]] = __uint_as_float(kUntouched);
    __syncthreads();
    if (a.trace && tid == 0) {
        long long *tr = a.trace + static_cast<size_t>(t) * 8;
        tr[0] = tr0; tr[1] = tr1; tr[2] = static_cast<long long>(wall_clock64()); tr[3] = tr_folded;
        tr[4] = ph_fold; tr[5] = ph_upd; tr[6] = ph_gap;
#if defined(MW_TRACE_WAIT) || defined(MW_TRACE_UPD)   // MW_TRACE_UPD (diagnostic build): tr[6] entries updated, tr[7] updates
        tr[7] = ph_wait;
#else
        tr[7] = SPEC && a.spec_min < 0x7fffffff ? ph_wait : ph_cyc;   // speculative fold: ticks the consumer waited for its producers
#endif
    }
}

// SPEC: the consumer wave may fold by integer prefix sums (csrc/fold_spec.hip.h); the chain-only form is a kernel of its
// own so that its register allocation is the one the literal chain was tuned with
template <bool SPEC>
__global__ __launch_bounds__(kMwThreads, 4) void fit_columns_mw_kernel(FitArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slot = blockIdx.x;
    const int K = min(a.cfg.top_features, a.I);
    const MwLds M = carve_mw(smem, K);
    for (;;) {
        if (threadIdx.x == 0) M.bc_i[3] = atomicAdd(a.queue, 1);
        __syncthreads();
        const int t = M.bc_i[3];
        __syncthreads();
        if (t >= a.n_targets) return;
        fit_one_mw<SPEC>(a, t, slot, smem);
    }
}

__global__ __launch_bounds__(64) void column_sqnorms_kernel(int n_items, const int *cptr, const float *cval, float *sqn) {
    const int lane = lane_id();
    for (int c = blockIdx.x; c < n_items; c += gridDim.x) {
        const int b = cptr[c], e = cptr[c + 1];
        float acc = 0.0f;
        for (int o = b; o < e; o += 64) {
            const int n = min(64, e - o);
            float prod = 0.0f;
            if (lane < n) { const float x = cval[o + lane]; prod = __fmul_rn(x, x); }
            acc = chain_add(acc, prod, n);
        }
        if (lane == 0) sqn[c] = acc;
    }
}


// ---------------------------------------------------------------------------------------------
// Gram matrix of the most popular items (input of the Gram tracking above): G = X_P^T X_P in float64.
// The P columns are first densified into XP[U][P64] (float32, zero where absent), then each workgroup
// accumulates one 64 x 64 tile of G over a chunk of rows -- 32-row slabs staged through LDS, 4 x 4
// double accumulators per thread (products of float32 values are exact in float64) -- and adds it to
// G with float64 atomics; only tiles on or above the diagonal are computed, the mirror image is
// written with them.
// ---------------------------------------------------------------------------------------------
constexpr int kGramTile = 64;
constexpr int kGramSlab = 32;
constexpr int kGramChunkRows = 4096;

__global__ __launch_bounds__(64) void gram_densify_kernel(int n_top, int p64, const int *top_items, const int *cptr,
                                                          const int *crow, const float *cval, float *xp) {
    const int lane = lane_id();
    for (int p = blockIdx.x; p < n_top; p += gridDim.x) {
        const int c = top_items[p];
        for (int o = cptr[c] + lane; o < cptr[c + 1]; o += 64) xp[static_cast<size_t>(crow[o]) * p64 + p] = cval[o];
    }
}

__global__ __launch_bounds__(256) void gram_tile_kernel(int n_users, int p64, const float *__restrict__ xp, double *G) {
    __shared__ __attribute__((aligned(16))) float As[kGramSlab][kGramTile];
    __shared__ __attribute__((aligned(16))) float Bs[kGramSlab][kGramTile];
    // blockIdx.x enumerates tile pairs (ti <= tj)
    const int T = p64 / kGramTile;
    int ti = 0, rem = static_cast<int>(blockIdx.x);
    while (rem >= T - ti) { rem -= T - ti; ++ti; }
    const int tj = ti + rem;
    const int tid = static_cast<int>(threadIdx.x), ty = tid >> 4, tx = tid & 15;
    const int r0 = static_cast<int>(blockIdx.y) * kGramChunkRows;
    const int r1 = min(n_users, r0 + kGramChunkRows);
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[i][k] = 0.0;
    for (int rb = r0; rb < r1; rb += kGramSlab) {
        // 32 x 64 floats per slab = 2048: 8 per thread, coalesced 256-byte rows
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + q * 256, rr = e >> 6, cc = e & 63;
            const int row = rb + rr;
            float va = 0.0f, vb = 0.0f;
            if (row < r1) {
                va = xp[static_cast<size_t>(row) * p64 + ti * kGramTile + cc];
                vb = xp[static_cast<size_t>(row) * p64 + tj * kGramTile + cc];
            }
            As[rr][cc] = va; Bs[rr][cc] = vb;
        }
        __syncthreads();
#pragma unroll 8
        for (int rr = 0; rr < kGramSlab; ++rr) {
            const float4 a4 = *reinterpret_cast<const float4 *>(&As[rr][ty * 4]);
            const float4 b4 = *reinterpret_cast<const float4 *>(&Bs[rr][tx * 4]);
            const double av[4] = {a4.x, a4.y, a4.z, a4.w};
            const double bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[i][k] += av[i] * bv[k];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (acc[i][k] == 0.0) continue;
            const int gi = ti * kGramTile + ty * 4 + i, gj = tj * kGramTile + tx * 4 + k;
            atomicAdd(&G[static_cast<size_t>(gi) * p64 + gj], acc[i][k]);
            if (ti != tj) atomicAdd(&G[static_cast<size_t>(gj) * p64 + gi], acc[i][k]);
        }
}

__global__ void fill_u32_kernel(uint32_t *p, size_t n, uint32_t v) {
    for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<size_t>(gridDim.x) * blockDim.x)
        p[i] = v;
}

}  // namespace rtrec

using namespace rtrec;

namespace {
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct FitWs { size_t R, stash, s, touched, cand_s, cand_i, w_all, long_list, total; };
FitWs fit_ws_layout(int U, int I, int slots, int top_features) {
    FitWs w;
    size_t o = 0;
    const size_t sl = static_cast<size_t>(slots);
    w.R = o;       o = align_up(o + sl * U * 4, 256);
    w.stash = o;   o = align_up(o + sl * U * 4, 256);
    w.s = o;       o = align_up(o + sl * I * 4, 256);
    w.touched = o; o = align_up(o + sl * I * 4, 256);
    w.cand_s = o;  o = align_up(o + sl * I * 4, 256);
    w.cand_i = o;  o = align_up(o + sl * I * 4, 256);
    w.w_all = o;   if (top_features <= 0) o = align_up(o + sl * I * 4, 256);
    w.long_list = o; if (top_features <= 0) o = align_up(o + sl * I * 4, 256);
    w.total = o;
    return w;
}
}  // namespace

extern "C" const char *rtrec_amd_version(void) { return "rtrec_amd 0.1 gfx950"; }

extern "C" const char *rtrec_amd_last_error(void) {
    return hipGetErrorString(static_cast<hipError_t>(rtrec::last_hip_error()));
}

extern "C" int rtrec_slim_column_sqnorms(int32_t n_items, const int32_t *d_csc_ptr, const float *d_csc_val,
                                         float *d_sqnorm, void *stream) {
    if (n_items < 0 || !d_csc_ptr || !d_sqnorm) return RTREC_ERR_INVALID_ARG;
    if (n_items == 0) return RTREC_OK;
    (void)hipGetLastError();   // drop stale errors of earlier, unrelated runtime calls
    const int grid = n_items < 8192 ? n_items : 8192;
    hipLaunchKernelGGL(column_sqnorms_kernel, dim3(grid), dim3(64), 0, static_cast<hipStream_t>(stream),
                       n_items, d_csc_ptr, d_csc_val, d_sqnorm);
    return rtrec::launch_status();
}

extern "C" size_t rtrec_slim_gram_workspace_bytes(int32_t n_users, int32_t n_top) {
    if (n_users <= 0 || n_top <= 0) return 0;
    const size_t p64 = (static_cast<size_t>(n_top) + 63) / 64 * 64;
    return static_cast<size_t>(n_users) * p64 * sizeof(float);
}

extern "C" int rtrec_slim_gram_matrix(int32_t n_users, int32_t n_items,
                                      const int32_t *d_csc_ptr, const int32_t *d_csc_row, const float *d_csc_val,
                                      const int32_t *d_top_items, int32_t n_top,
                                      void *d_workspace, size_t workspace_bytes, double *d_gram, void *stream) {
    if (n_users <= 0 || n_items <= 0 || n_top <= 0 || n_top > 4096) return RTREC_ERR_INVALID_ARG;
    if (!d_csc_ptr || !d_csc_row || !d_csc_val || !d_top_items || !d_workspace || !d_gram) return RTREC_ERR_INVALID_ARG;
    const int p64 = (n_top + 63) / 64 * 64;
    const size_t need = static_cast<size_t>(n_users) * p64 * sizeof(float);
    if (workspace_bytes < need) return RTREC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (hipMemsetAsync(d_workspace, 0, need, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    if (hipMemsetAsync(d_gram, 0, static_cast<size_t>(p64) * p64 * sizeof(double), st) != hipSuccess) return RTREC_ERR_LAUNCH;
    float *xp = static_cast<float *>(d_workspace);
    hipLaunchKernelGGL(gram_densify_kernel, dim3(n_top < 4096 ? n_top : 4096), dim3(64), 0, st, n_top, p64, d_top_items,
                       d_csc_ptr, d_csc_row, d_csc_val, xp);
    const int T = p64 / kGramTile;
    const unsigned chunks = static_cast<unsigned>((n_users + kGramChunkRows - 1) / kGramChunkRows);
    hipLaunchKernelGGL(gram_tile_kernel, dim3(static_cast<unsigned>(T * (T + 1) / 2), chunks), dim3(256), 0, st, n_users, p64,
                       xp, d_gram);
    return rtrec::launch_status();
}

extern "C" size_t rtrec_slim_fit_workspace_bytes(int32_t n_users, int32_t n_items, int32_t n_slots, int32_t top_features) {
    if (n_users <= 0 || n_items <= 0 || n_slots <= 0) return 0;
    return fit_ws_layout(n_users, n_items, n_slots, top_features).total;
}

extern "C" int rtrec_slim_fit_workspace_init(void *d_workspace, size_t workspace_bytes, int32_t n_users, int32_t n_items,
                                             int32_t n_slots, int32_t top_features, void *stream) {
    if (!d_workspace || n_users <= 0 || n_items <= 0 || n_slots <= 0) return RTREC_ERR_INVALID_ARG;
    const FitWs L = fit_ws_layout(n_users, n_items, n_slots, top_features);
    if (workspace_bytes < L.total) return RTREC_ERR_WORKSPACE;
    (void)hipGetLastError();
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char *ws = static_cast<unsigned char *>(d_workspace);
    if (hipMemsetAsync(ws, 0, L.total, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    const size_t n = static_cast<size_t>(n_slots) * n_items;
    hipLaunchKernelGGL(fill_u32_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<uint32_t *>(ws + L.s), n, kUntouched);
    return rtrec::launch_status();
}

struct XtyWs { size_t tmap, ypos, ylen, cursor, cand_cnt, yt, yv, cand_i, cand_s, total; };
static XtyWs xty_ws_layout(int n_users, int n_items, long long nnz, int n_targets) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    XtyWs w;
    size_t o = 0;
    w.tmap = o;     o = up(o + static_cast<size_t>(n_items) * 4);
    w.ypos = o;     o = up(o + static_cast<size_t>(n_users) * 4);
    w.ylen = o;     o = up(o + static_cast<size_t>(n_users) * 4);
    w.cursor = o;   o = up(o + 256);
    w.cand_cnt = o; o = up(o + static_cast<size_t>(n_targets) * 4);
    w.yt = o;       o = up(o + static_cast<size_t>(nnz) * 4);
    w.yv = o;       o = up(o + static_cast<size_t>(nnz) * 4);
    w.cand_i = o;   o = up(o + static_cast<size_t>(n_targets) * n_items * 4);
    w.cand_s = o;   o = up(o + static_cast<size_t>(n_targets) * n_items * 4);
    w.total = o;
    return w;
}

extern "C" size_t rtrec_slim_xty_workspace_bytes(int32_t n_users, int32_t n_items, int64_t nnz, int32_t n_targets) {
    if (n_users <= 0 || n_items <= 0 || nnz <= 0 || n_targets <= 0 || n_targets > kXtyMaxTargets) return 0;
    return xty_ws_layout(n_users, n_items, nnz, n_targets).total;
}

static int fit_columns_impl(int32_t n_users, int32_t n_items,
                            const int32_t *d_csc_ptr, const int32_t *d_csc_row, const float *d_csc_val,
                            const int32_t *d_csr_ptr, const int32_t *d_csr_col, const float *d_csr_val,
                            const float *d_sqnorm,
                            const int32_t *d_targets, int32_t n_targets,
                            const rtrec_fit_cfg *cfg,
                            int32_t *d_out_items, float *d_out_coef, int32_t *d_out_count,
                            int32_t *d_out_n_iter, int32_t cap,
                            void *d_workspace, size_t workspace_bytes, int32_t n_slots,
                            int32_t *d_queue, void *stream, const rtrec_fit_opts *opts) {
    if (n_users <= 0 || n_items <= 0 || n_targets < 0 || !cfg || n_slots <= 0) return RTREC_ERR_INVALID_ARG;
    if (n_targets == 0) return RTREC_OK;
    if (!d_csc_ptr || !d_csr_ptr || !d_sqnorm || !d_targets || !d_out_items || !d_out_coef || !d_out_count ||
        !d_out_n_iter || !d_workspace || !d_queue)
        return RTREC_ERR_INVALID_ARG;
    const bool allf = cfg->top_features <= 0;
    const int K = allf ? n_items : (cfg->top_features < n_items ? cfg->top_features : n_items);
    if (cap < (allf ? 1 : K)) return RTREC_ERR_INVALID_ARG;
    if (!allf && K > 4096) return RTREC_ERR_UNSUPPORTED;   // 7 LDS arrays of K entries per wave
    if (cfg->max_iter <= 0) return RTREC_ERR_INVALID_ARG;
    const FitWs L = fit_ws_layout(n_users, n_items, n_slots, cfg->top_features);
    if (workspace_bytes < L.total) return RTREC_ERR_WORKSPACE;

    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned char *ws = static_cast<unsigned char *>(d_workspace);
    FitArgs a{};
    a.U = n_users; a.I = n_items;
    a.cptr = d_csc_ptr; a.crow = d_csc_row; a.cval = d_csc_val;
    a.rptr = d_csr_ptr; a.rcol = d_csr_col; a.rval = d_csr_val;
    a.sqn = d_sqnorm; a.targets = d_targets; a.n_targets = n_targets; a.cfg = *cfg;
    a.out_items = d_out_items; a.out_coef = d_out_coef; a.out_count = d_out_count; a.out_niter = d_out_n_iter; a.cap = cap;
    a.R = reinterpret_cast<float *>(ws + L.R);
    a.stash = reinterpret_cast<float *>(ws + L.stash);
    a.s = reinterpret_cast<float *>(ws + L.s);
    a.touched = reinterpret_cast<int *>(ws + L.touched);
    a.cand_s = reinterpret_cast<float *>(ws + L.cand_s);
    a.cand_i = reinterpret_cast<int *>(ws + L.cand_i);
    a.w_all = allf ? reinterpret_cast<float *>(ws + L.w_all) : nullptr;
    a.long_list = allf ? reinterpret_cast<int *>(ws + L.long_list) : nullptr;
    a.queue = d_queue;
    if (opts) {
        a.trace = reinterpret_cast<long long *>(opts->d_trace);
        if (opts->d_gram && opts->d_gram_index && opts->gram_n > 0 && opts->gram_rel_err >= 0.0 && opts->gram_rel_err < 1e-6) {
            a.gram = opts->d_gram; a.gram_index = opts->d_gram_index; a.gram_n = opts->gram_n;
            a.gram_rel_err = opts->gram_rel_err;
        }
        a.fast = opts->fast < 0 ? 0 : (opts->fast > 2 ? 2 : opts->fast);
    }
    // tuning / test knobs arrive in rtrec_fit_opts (0 = the built-in default); the library reads no environment
    a.colwalk_min_rows = (opts && opts->colwalk_min_rows > 0) ? opts->colwalk_min_rows : kColWalkMinRows;
    a.screen_min = (opts && opts->screen_min > 0) ? opts->screen_min : kScreenMinDefault;
    a.lane_max = (opts && opts->lane_max != 0) ? (opts->lane_max < 0 ? 0 : opts->lane_max) : kLaneMaxDefault;
    {   // how the ordered dot products are evaluated (rtrec_fit_opts.fold); results are bit-identical either way
        const int fold = opts ? opts->fold : 0;
        a.spec_min = fold == 2 ? 64 : (fold == 3 ? kSpecMinDefault : 0x7fffffff);
    }
    (void)hipGetLastError();
    if (hipMemsetAsync(d_queue, 0, 4, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    const int grid = n_slots < n_targets ? n_slots : n_targets;
    if (allf) {
        if (a.spec_min < 0x7fffffff) hipLaunchKernelGGL(HIP_KERNEL_NAME(fit_columns_kernel<true, true>), dim3(grid), dim3(64), kFoldBufBytes + 16, st, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(fit_columns_kernel<true, false>), dim3(grid), dim3(64), kFoldBufBytes + 16, st, a);
    } else {
        // few targets (online partial_fit): the heaviest target is the critical path -> latency mode (multi-wave
        // kernel); opts->kernel forces one of the two (A/B runs, tests).  The tolerance mode has no ordered fold to
        // feed, so it always takes the one-wave-per-target kernel.
        const int force = opts ? opts->kernel : 0;
        // tolerance mode: the multi-wave kernel serves Gram-form CD only (fast = 2); tree-reduced dots are single-wave
        const bool latency_mode = a.fast != 1 && (force == 2 || (force == 0 && n_targets <= kMwMaxTargets));
        if (latency_mode && opts && opts->d_xty_ws && opts->nnz > 0 && n_targets <= kXtyMaxTargets) {
            // X^T y of every target of the call in one pass over X (xty_batch_kernel); the fit kernel then only selects
            const XtyWs X = xty_ws_layout(n_users, n_items, opts->nnz, n_targets);
            if (opts->xty_ws_bytes < X.total) return RTREC_ERR_WORKSPACE;
            unsigned char *xw = static_cast<unsigned char *>(opts->d_xty_ws);
            XtyArgs x{};
            x.U = n_users; x.I = n_items; x.n_t = n_targets;
            x.cptr = d_csc_ptr; x.crow = d_csc_row; x.cval = d_csc_val;
            x.rptr = d_csr_ptr; x.rcol = d_csr_col; x.rval = d_csr_val;
            x.col_order = opts->d_col_order; x.targets = d_targets;
            x.tmap = reinterpret_cast<int *>(xw + X.tmap); x.ypos = reinterpret_cast<int *>(xw + X.ypos);
            x.ylen = reinterpret_cast<int *>(xw + X.ylen); x.cursor = reinterpret_cast<int *>(xw + X.cursor);
            x.cand_cnt = reinterpret_cast<int *>(xw + X.cand_cnt);
            x.yt = reinterpret_cast<int *>(xw + X.yt); x.yv = reinterpret_cast<float *>(xw + X.yv);
            x.cand_i = reinterpret_cast<int *>(xw + X.cand_i); x.cand_s = reinterpret_cast<float *>(xw + X.cand_s);
            if (hipMemsetAsync(x.tmap, 0xff, static_cast<size_t>(n_items) * 4, st) != hipSuccess) return RTREC_ERR_LAUNCH;
            hipLaunchKernelGGL(xty_tmap_kernel, dim3((n_targets + 255) / 256), dim3(256), 0, st, x);
            hipLaunchKernelGGL(xty_compact_kernel, dim3((n_users + 3) / 4), dim3(256), 0, st, x);
            const size_t xl = static_cast<size_t>(kXtyWaves) * n_targets * 4;
            const int per_cu = xl * 2 <= 160u * 1024u ? 2 : 1;
            hipLaunchKernelGGL(xty_batch_kernel, dim3(256 * per_cu), dim3(kXtyWaves * 64), xl, st, x);
            a.pre_cnt = x.cand_cnt; a.pre_i = x.cand_i; a.pre_s = x.cand_s;
        }
        if (latency_mode) {
            size_t mw_lds = mw_lds_bytes(K);
#ifdef MW_LDS_PAD       // diagnostic build: extra dynamic LDS so that only one workgroup fits a CU
            mw_lds += MW_LDS_PAD;
#endif
            if (a.fast >= 2 && a.gram && K <= 64) mw_lds = ((mw_lds + 15) / 16) * 16 + static_cast<size_t>(K) * 64 * 4;
            if (a.spec_min < 0x7fffffff)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(fit_columns_mw_kernel<true>), dim3(grid), dim3(kMwThreads), mw_lds, st, a);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(fit_columns_mw_kernel<false>), dim3(grid), dim3(kMwThreads), mw_lds, st, a);
        } else {
            size_t lds = kFoldBufBytes + feat_lds_bytes(K);
            if (a.fast >= 2 && a.gram && K <= 64) lds = kFoldBufBytes + ((feat_lds_bytes(K) + 15) / 16) * 16 + static_cast<size_t>(K) * 64 * 4;
            if (a.spec_min < 0x7fffffff) hipLaunchKernelGGL(HIP_KERNEL_NAME(fit_columns_kernel<false, true>), dim3(grid), dim3(64), lds, st, a);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(fit_columns_kernel<false, false>), dim3(grid), dim3(64), lds, st, a);
        }
    }
    return rtrec::launch_status();
}

extern "C" int rtrec_slim_fit_columns(int32_t n_users, int32_t n_items,
                                      const int32_t *d_csc_ptr, const int32_t *d_csc_row, const float *d_csc_val,
                                      const int32_t *d_csr_ptr, const int32_t *d_csr_col, const float *d_csr_val,
                                      const float *d_sqnorm,
                                      const int32_t *d_targets, int32_t n_targets,
                                      const rtrec_fit_cfg *cfg,
                                      int32_t *d_out_items, float *d_out_coef, int32_t *d_out_count,
                                      int32_t *d_out_n_iter, int32_t cap,
                                      void *d_workspace, size_t workspace_bytes, int32_t n_slots,
                                      int32_t *d_queue, void *stream) {
    return fit_columns_impl(n_users, n_items, d_csc_ptr, d_csc_row, d_csc_val, d_csr_ptr, d_csr_col, d_csr_val, d_sqnorm,
                            d_targets, n_targets, cfg, d_out_items, d_out_coef, d_out_count, d_out_n_iter, cap,
                            d_workspace, workspace_bytes, n_slots, d_queue, stream, nullptr);
}

extern "C" int rtrec_slim_fit_columns_opt(int32_t n_users, int32_t n_items,
                                          const int32_t *d_csc_ptr, const int32_t *d_csc_row, const float *d_csc_val,
                                          const int32_t *d_csr_ptr, const int32_t *d_csr_col, const float *d_csr_val,
                                          const float *d_sqnorm,
                                          const int32_t *d_targets, int32_t n_targets,
                                          const rtrec_fit_cfg *cfg,
                                          int32_t *d_out_items, float *d_out_coef, int32_t *d_out_count,
                                          int32_t *d_out_n_iter, int32_t cap,
                                          void *d_workspace, size_t workspace_bytes, int32_t n_slots,
                                          int32_t *d_queue, void *stream, const rtrec_fit_opts *opts) {
    return fit_columns_impl(n_users, n_items, d_csc_ptr, d_csc_row, d_csc_val, d_csr_ptr, d_csr_col, d_csr_val, d_sqnorm,
                            d_targets, n_targets, cfg, d_out_items, d_out_coef, d_out_count, d_out_n_iter, cap,
                            d_workspace, workspace_bytes, n_slots, d_queue, stream, opts);
}
