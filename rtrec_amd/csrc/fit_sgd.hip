// rtrec_amd/csrc/fit_sgd.hip -- optim="sgd": scikit-learn's SGDRegressor behind FeatureSelectionWrapper, per target column.
//
// Replaces (reference): SLIMElastic.get_model's SGDRegressor(loss="squared_error", penalty="elasticnet",
// learning_rate="invscaling", fit_intercept=False, average=False) (slim_elastic.py:209-222) fitted on X[:, selected]
// (slim_elastic.py:139-154), i.e. sklearn/linear_model/_sgd_fast.pyx.tp:_plain_sgd32 with WeightVector32
// (sklearn/utils/_weight_vector.pyx.tp), CSRDataset32.shuffle (sklearn/utils/_seq_dataset.pyx.tp:137-145) and
// CyHalfSquaredError (sklearn/_loss/_loss.pyx.tp:310-321).  Restated on the CPU in oracle/slim_oracle.c:slim_oracle_sgd.
//
// The solver is one strictly sequential pass over ALL U samples per epoch and target (an empty row still advances the learning
// rate, the weight scale and the cumulative L1 penalty).  What makes it tractable on the device:
//   * everything that does not depend on the target is a SCHEDULE computed once on the host (rtrec_slim_sgd_schedule, with
//     the C library's pow(), like Cython's): per global step g the learning rate eta_g, the weight scale before / after the
//     step's w.scale(), the cumulative L1 penalty u, and the steps at which reset_wscale() ran with its float factor; per
//     epoch the shuffled sample order as time_of[sample];
//   * a sample whose row holds none of the target's K features and whose y is 0 changes NOTHING of the target's own state
//     (p = 0, loss += 0, update = -0.0): it is skipped, and the schedule supplies eta / wscale / u at the next sample that
//     matters (pending resets are applied in order);
//   * per epoch the CSC copy of X is sorted by time inside every column (shared by all targets: the caller builds it with
//     one device sort), so a target's samples come out of a K-way merge: lane f holds a cursor into feature f's column.
// One wave per target: lane = position of the feature in the selection (= X[:, selected]'s column order, hence the order of a
// row's entries and of the double-precision dot product; K <= 64: one position per lane, up to K = 256 two or four), y is a
// uniform cursor.  All arithmetic keeps the C types of the
// Cython source: float products, double sums, the float parameters of WeightVector.scale() / add().
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace rtrec {

constexpr int kSgdInf = 0x7fffffff;
constexpr int kSgdMaxFeatures = 256;       // selected features per target: up to four per wave lane

struct SgdArgs {
    int U, I;
    const int *cptr;
    const int *ttime; const float *tval; long long nnz;      // [n_epochs][nnz]: per epoch, every column's entries by time
    const int *targets; int n_targets;
    const int *sel; const int *sel_count; int cap;
    int first_epoch, n_epochs, max_iter;
    double tol, l1_ratio_unused;
    const double *eta, *ws_before, *ws_after, *u_after;      // [n_epochs * U]
    const int *reset_cnt;                                    // [n_epochs * U + 1]: resets at steps < g (block-relative)
    const float *reset_mult;                                 // the resets of the block in order: float(wscale) at the reset
    float *w, *q;                                            // [n_targets][cap] state across blocks
    double *best_loss; int *no_improve;                      // [n_targets]
    int *n_iter;                                             // [n_targets] 0 = still running, > 0 epochs run, -1 non-finite
    int *unfinished;
};

// uniform minimum of the 64 lane values: four row_shr steps + two row broadcasts in the DPP path (no LDS crossbar round
// trips: this reduction runs once per solver step)
template <int CTRL> __device__ __forceinline__ int sgd_dpp(int v, int fill) {
    return __builtin_amdgcn_update_dpp(fill, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, sgd_dpp<0x111>(v, kSgdInf));
    v = min(v, sgd_dpp<0x112>(v, kSgdInf));
    v = min(v, sgd_dpp<0x114>(v, kSgdInf));
    v = min(v, sgd_dpp<0x118>(v, kSgdInf));
    v = min(v, sgd_dpp<0x142>(v, kSgdInf));      // row_bcast:15
    v = min(v, sgd_dpp<0x143>(v, kSgdInf));      // row_bcast:31
    return readlane_i(v, 63);
}

// F features per lane (K <= 64 F): the feature at position f * 64 + lane of the selection is lane's f-th.  A row's entries
// are summed position by position, i.e. f outer, lanes inner.
template <int F>
__global__ __launch_bounds__(64) void fit_sgd_kernel(SgdArgs a) {
    const int lane = lane_id();
    const int U = a.U;
    for (int t = blockIdx.x; t < a.n_targets; t += gridDim.x) {
        if (a.n_iter[t] != 0) continue;                      // finished in an earlier block of epochs
        const int j = a.targets[t];
        const int Kc = a.sel_count[t];
        int cb[F], ce[F];
        float w[F], q[F];
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int pos = f * 64 + lane;
            const int col = pos < Kc ? a.sel[static_cast<size_t>(t) * a.cap + pos] : -1;
            const bool live = col >= 0 && col != j;          // (the zeroed target column's explicit zeros change nothing)
            cb[f] = live ? a.cptr[col] : 0; ce[f] = live ? a.cptr[col + 1] : 0;
            w[f] = pos < a.cap ? a.w[static_cast<size_t>(t) * a.cap + pos] : 0.0f;
            q[f] = pos < a.cap ? a.q[static_cast<size_t>(t) * a.cap + pos] : 0.0f;
        }
        const int yb = a.cptr[j], ye = a.cptr[j + 1];
        double best_loss = a.best_loss[t];
        int no_imp = a.no_improve[t];
        int done = 0;
        for (int ep = 0; ep < a.n_epochs && !done; ++ep) {
            const int *tt = a.ttime + static_cast<size_t>(ep) * a.nnz;
            const float *tv = a.tval + static_cast<size_t>(ep) * a.nnz;
            const long long g0 = static_cast<long long>(ep) * U;
            int cur[F], my_t[F], nx_t[F];
            float my_x[F], nx_x[F];
#pragma unroll
            for (int f = 0; f < F; ++f) {
                cur[f] = cb[f];
                my_t[f] = cur[f] < ce[f] ? tt[cur[f]] : kSgdInf;
                my_x[f] = cur[f] < ce[f] ? tv[cur[f]] : 0.0f;
                nx_t[f] = cur[f] + 1 < ce[f] ? tt[cur[f] + 1] : kSgdInf;   // one entry ahead: an advance does not wait for memory
                nx_x[f] = cur[f] + 1 < ce[f] ? tv[cur[f] + 1] : 0.0f;
            }
            int yc = yb;
            int y_t = yc < ye ? tt[yc] : kSgdInf;
            float y_v = yc < ye ? tv[yc] : 0.0f;
            double sumloss = 0.0;
            int ri = a.reset_cnt[g0];
            // the schedule of 64 consecutive steps in registers (lane l: step gw + l): a target whose samples are dense in time
            // reads its tables once per 64 steps instead of six dependent broadcast loads per step
            long long gw = -(1ll << 40);
            double w_eta = 0.0, w_wsb = 0.0, w_wsa = 0.0, w_ua = 0.0;
            int w_rc0 = 0, w_rc1 = 0;
            const long long g_last = static_cast<long long>(a.n_epochs) * U - 1;
            for (;;) {
                int lmin = my_t[0];
#pragma unroll
                for (int f = 1; f < F; ++f) lmin = min(lmin, my_t[f]);
                const int tmin = min(wave_min_i(lmin), y_t);
                if (tmin == kSgdInf) break;
                const long long g = g0 + tmin;
                if (g - gw >= 64) {
                    gw = g;
                    const long long gl = min(gw + lane, g_last);
                    w_eta = a.eta[gl]; w_wsb = a.ws_before[gl]; w_wsa = a.ws_after[gl]; w_ua = a.u_after[gl];
                    w_rc0 = a.reset_cnt[gl]; w_rc1 = a.reset_cnt[gl + 1];
                }
                const int wl = static_cast<int>(g - gw);
                const int rc = readlane_i(w_rc0, wl);
                while (ri < rc) {                                   // reset_wscale() of the skipped steps
                    const float rm = a.reset_mult[ri];
#pragma unroll
                    for (int f = 0; f < F; ++f) w[f] = __fmul_rn(rm, w[f]);
                    ++ri;
                }
                bool part[F];
                double innerprod = 0.0;
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    part[f] = my_t[f] == tmin;
                    const float prod = part[f] ? __fmul_rn(w[f], my_x[f]) : 0.0f;
                    for (unsigned long long mm = __ballot(part[f]); mm; mm &= mm - 1)
                        innerprod = __dadd_rn(innerprod, static_cast<double>(readlane_f(prod, static_cast<int>(__builtin_ctzll(mm)))));
                }
                innerprod = __dmul_rn(innerprod, readlane_d(w_wsb, wl));
                const double p = static_cast<double>(static_cast<float>(innerprod));
                const double yv = (y_t == tmin) ? static_cast<double>(y_v) : 0.0;
                const double eta = readlane_d(w_eta, wl);
                const double d = __dsub_rn(p, yv);
                sumloss = __dadd_rn(sumloss, __dmul_rn(__dmul_rn(0.5, d), d));
                double dloss = d;
                if (dloss < -1e12) dloss = -1e12; else if (dloss > 1e12) dloss = 1e12;
                const double update = __dmul_rn(-eta, dloss);        // (x class_weight x sample_weight = 1.0f: exact)
                if (readlane_i(w_rc1, wl) > rc) {                    // this step's w.scale() reset
                    const float rm = a.reset_mult[ri];
#pragma unroll
                    for (int f = 0; f < F; ++f) w[f] = __fmul_rn(rm, w[f]);
                    ++ri;
                }
                const double wsd = readlane_d(w_wsa, wl);
                const double u = readlane_d(w_ua, wl);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    if (update != 0.0 && part[f]) {
                        const float c = static_cast<float>(update), wsf = static_cast<float>(wsd);
                        w[f] = static_cast<float>(__dadd_rn(static_cast<double>(w[f]),
                                                            __dmul_rn(static_cast<double>(my_x[f]), static_cast<double>(__fdiv_rn(c, wsf)))));
                    }
                    if (part[f]) {                                       // l1penalty32
                        const double z = static_cast<double>(w[f]);
                        const double sz = __dmul_rn(wsd, z);
                        if (sz > 0.0) {
                            const double v = __dsub_rn(static_cast<double>(w[f]), __ddiv_rn(__dadd_rn(u, static_cast<double>(q[f])), wsd));
                            w[f] = static_cast<float>(v > 0.0 ? v : 0.0);
                        } else if (sz < 0.0) {
                            const double v = __dadd_rn(static_cast<double>(w[f]), __ddiv_rn(__dsub_rn(u, static_cast<double>(q[f])), wsd));
                            w[f] = static_cast<float>(v < 0.0 ? v : 0.0);
                        }
                        q[f] = static_cast<float>(__dadd_rn(static_cast<double>(q[f]), __dmul_rn(wsd, __dsub_rn(static_cast<double>(w[f]), z))));
                        // advance this lane's cursor
                        ++cur[f];
                        my_t[f] = nx_t[f]; my_x[f] = nx_x[f];
                        nx_t[f] = cur[f] + 1 < ce[f] ? tt[cur[f] + 1] : kSgdInf;
                        nx_x[f] = cur[f] + 1 < ce[f] ? tv[cur[f] + 1] : 0.0f;
                    }
                }
                if (y_t == tmin) {
                    ++yc;
                    y_t = yc < ye ? tt[yc] : kSgdInf;
                    y_v = yc < ye ? tv[yc] : 0.0f;
                }
            }
            {   // the epoch's remaining (skipped) steps
                const int rc = a.reset_cnt[g0 + U];
                while (ri < rc) {
                    const float rm = a.reset_mult[ri];
#pragma unroll
                    for (int f = 0; f < F; ++f) w[f] = __fmul_rn(rm, w[f]);
                    ++ri;
                }
            }
            const int epoch = a.first_epoch + ep;
            bool bad = false;
#pragma unroll
            for (int f = 0; f < F; ++f) bad = bad || (f * 64 + lane < Kc && !isfinite(w[f]));
            if (__ballot(bad)) { done = -1; break; }      // any_nonfinite(weights): sklearn raises ValueError
            if (sumloss > __dsub_rn(best_loss, __dmul_rn(a.tol, static_cast<double>(static_cast<unsigned int>(U))))) ++no_imp;
            else no_imp = 0;
            if (sumloss < best_loss) best_loss = sumloss;
            if (no_imp >= 5 || epoch == a.max_iter - 1) {
                const float rm = static_cast<float>(a.ws_after[g0 + U - 1]);     // w.reset_wscale()
#pragma unroll
                for (int f = 0; f < F; ++f) w[f] = __fmul_rn(rm, w[f]);
                done = epoch + 1;
            }
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int pos = f * 64 + lane;
            if (pos < a.cap) { a.w[static_cast<size_t>(t) * a.cap + pos] = w[f]; a.q[static_cast<size_t>(t) * a.cap + pos] = q[f]; }
        }
        if (lane == 0) {
            a.best_loss[t] = best_loss; a.no_improve[t] = no_imp;
            if (done) a.n_iter[t] = done; else atomicAdd(a.unfinished, 1);
        }
    }
}

}  // namespace rtrec

using namespace rtrec;

namespace {
// sklearn/utils/_random.pxd:20-34 (our_rand_r)
inline uint32_t sgd_rand_r(uint32_t *state) {
    if (*state == 0) *state = 1;
    *state ^= static_cast<uint32_t>(*state << 13);
    *state ^= static_cast<uint32_t>(*state >> 17);
    *state ^= static_cast<uint32_t>(*state << 5);
    return *state % (static_cast<uint32_t>(2147483647) + 1u);
}
}  // namespace

// HOST routine (no device work): the target-independent schedule of `n_epochs` epochs of _plain_sgd32, continuing from
// `sample_order` (identity before epoch 0; on return the order the last epoch left) and `state` = {wscale, u, t}.
extern "C" int rtrec_slim_sgd_schedule(int32_t n_samples, int32_t n_epochs, uint32_t seed,
                                       double alpha, double l1_ratio, double eta0, double power_t,
                                       int32_t *sample_order, double *state,
                                       int32_t *time_of, double *eta, double *ws_before, double *ws_after, double *u_after,
                                       int32_t *reset_cnt, float *reset_mult, int32_t reset_cap, int32_t *n_resets) {
    if (n_samples <= 0 || n_epochs <= 0 || !sample_order || !state || !time_of || !eta || !ws_before || !ws_after || !u_after ||
        !reset_cnt || !reset_mult || !n_resets)
        return RTREC_ERR_INVALID_ARG;
    double wscale = state[0], u = state[1], t = state[2];
    int32_t nr = 0;
    long long g = 0;
    for (int32_t ep = 0; ep < n_epochs; ++ep) {
        uint32_t s = seed;                                   // dataset.shuffle(seed): the seed is passed by value
        for (int32_t i = 0; i < n_samples - 1; ++i) {
            const int32_t jx = i + static_cast<int32_t>(sgd_rand_r(&s) % static_cast<uint32_t>(n_samples - i));
            const int32_t tmp = sample_order[i]; sample_order[i] = sample_order[jx]; sample_order[jx] = tmp;
        }
        int32_t *tof = time_of + static_cast<size_t>(ep) * n_samples;
        for (int32_t i = 0; i < n_samples; ++i) tof[sample_order[i]] = i;
        for (int32_t i = 0; i < n_samples; ++i, ++g) {
            const double e = eta0 / pow(t, power_t);
            eta[g] = e;
            ws_before[g] = wscale;
            reset_cnt[g] = nr;
            const double arg = 1.0 - ((1.0 - l1_ratio) * e * alpha);
            const float c = static_cast<float>(arg > 0.0 ? arg : 0.0);      // the float parameter of WeightVector32.scale()
            wscale *= c;
            if (wscale < 1e-6) {
                if (nr >= reset_cap) return RTREC_ERR_WORKSPACE;
                reset_mult[nr++] = static_cast<float>(wscale);
                wscale = 1.0;
            }
            ws_after[g] = wscale;
            u += (l1_ratio * e * alpha);
            u_after[g] = u;
            t += 1.0;
        }
    }
    reset_cnt[g] = nr;
    *n_resets = nr;
    state[0] = wscale; state[1] = u; state[2] = t;
    return RTREC_OK;
}

extern "C" int rtrec_slim_fit_sgd_epochs(int32_t n_users, int32_t n_items, const int32_t *d_csc_ptr,
                                         const int32_t *d_ttime, const float *d_tval, int64_t nnz,
                                         const int32_t *d_targets, int32_t n_targets,
                                         const int32_t *d_sel, const int32_t *d_sel_count, int32_t cap,
                                         int32_t first_epoch, int32_t n_epochs, int32_t max_iter, double tol,
                                         const double *d_eta, const double *d_ws_before, const double *d_ws_after,
                                         const double *d_u_after, const int32_t *d_reset_cnt, const float *d_reset_mult,
                                         float *d_w, float *d_q, double *d_best_loss, int32_t *d_no_improve,
                                         int32_t *d_n_iter, int32_t *d_unfinished, void *stream) {
    if (n_users <= 0 || n_items <= 0 || n_targets < 0 || n_epochs <= 0 || max_iter <= 0 || cap <= 0 || cap > kSgdMaxFeatures)
        return cap > kSgdMaxFeatures ? RTREC_ERR_UNSUPPORTED : RTREC_ERR_INVALID_ARG;
    if (n_targets == 0) return RTREC_OK;
    if (!d_csc_ptr || !d_ttime || !d_tval || !d_targets || !d_sel || !d_sel_count || !d_eta || !d_ws_before || !d_ws_after ||
        !d_u_after || !d_reset_cnt || !d_reset_mult || !d_w || !d_q || !d_best_loss || !d_no_improve || !d_n_iter || !d_unfinished)
        return RTREC_ERR_INVALID_ARG;
    SgdArgs a{};
    a.U = n_users; a.I = n_items; a.cptr = d_csc_ptr; a.ttime = d_ttime; a.tval = d_tval; a.nnz = nnz;
    a.targets = d_targets; a.n_targets = n_targets; a.sel = d_sel; a.sel_count = d_sel_count; a.cap = cap;
    a.first_epoch = first_epoch; a.n_epochs = n_epochs; a.max_iter = max_iter; a.tol = tol;
    a.eta = d_eta; a.ws_before = d_ws_before; a.ws_after = d_ws_after; a.u_after = d_u_after;
    a.reset_cnt = d_reset_cnt; a.reset_mult = d_reset_mult;
    a.w = d_w; a.q = d_q; a.best_loss = d_best_loss; a.no_improve = d_no_improve; a.n_iter = d_n_iter; a.unfinished = d_unfinished;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (hipMemsetAsync(d_unfinished, 0, 4, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    const int grid = n_targets < 16384 ? n_targets : 16384;
    if (cap <= 64) hipLaunchKernelGGL(fit_sgd_kernel<1>, dim3(grid), dim3(64), 0, st, a);
    else if (cap <= 128) hipLaunchKernelGGL(fit_sgd_kernel<2>, dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL(fit_sgd_kernel<4>, dim3(grid), dim3(64), 0, st, a);
    return rtrec::launch_status();
}
