// rtrec_amd/csrc/store_device.hip -- device side of the interaction store: time decay of the resident matrix, and the
// per-pair fold of a bulk batch (fold_kernel below).
//
// Replaces (reference): UserItemInteractions._apply_decay (rtrec/utils/interactions.py:62-79), evaluated per stored
// entry by every export (to_csr :259-289, to_csc :291-303):
//     value * decay_rate ** ((max_timestamp - tstamp) / 86400.0)      in float64 (CPython: libm pow), cast to float32.
// With time decay every value of X is a function of max_timestamp, which moves with every mini-batch, so a
// resident X has to be re-evaluated on the device: one pass over (raw value, timestamp) pairs, 16 B read and 4 B
// written per interaction.
//
// Bit-exactness: X is float32, the reference's arithmetic float64.  The device's pow() is not libm's bit for bit
// (both are within an ulp or so of the true value), but a float64 product rounds to the same float32 unless it lies
// within that error of a float32 rounding boundary -- about one value in 2^29 / margin.  The kernel therefore
// flags every entry whose product is closer than kMarginUlps float64 ulps to a boundary; the host re-evaluates
// just those with libm (rtrec_store_decay) and patches them.  Every other float32 is provably the reference's.
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

namespace rtrec {

constexpr double kMarginUlps = 4096.0;    // float64 ulps of slack granted to the device pow (its error is ~1)

__global__ __launch_bounds__(256) void decay_kernel(const double *__restrict__ val, const double *__restrict__ ts, long long n,
                                                    double rate, double now, float *__restrict__ out,
                                                    int *__restrict__ unsafe_idx, int *__restrict__ unsafe_count, int cap) {
    for (long long k = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; k < n;
         k += static_cast<long long>(gridDim.x) * blockDim.x) {
        const double elapsed_days = (now - ts[k]) / 86400.0;
        const double v = val[k] * pow(rate, elapsed_days);
        const float f = static_cast<float>(v);
        out[k] = f;
        // distance of v to the nearest float32 rounding boundary (the midpoints between f and its neighbours)
        const double fd = static_cast<double>(f);
        if (!(fabs(v) < 1e300) || v == 0.0) continue;                        // inf / nan / an exact zero: nothing to decide
        if (fd == 0.0 || fabs(v) < 0x1p-148) {
            // at the underflow boundary (|v| around 2^-150: float32 rounds to 0 or to its smallest denormal): the spacing
            // test below does not apply, the host decides with libm
            const int slot = atomicAdd(unsafe_count, 1);
            if (slot < cap) unsafe_idx[slot] = static_cast<int>(k);
            continue;
        }
        const float up = __uint_as_float(__float_as_uint(fabsf(f)) + 1u), dn = __uint_as_float(__float_as_uint(fabsf(f)) - 1u);
        const double a = fabs(v), m_up = 0.5 * (fabs(fd) + static_cast<double>(up)), m_dn = 0.5 * (fabs(fd) + static_cast<double>(dn));
        const double dist = fmin(fabs(a - m_up), fabs(a - m_dn));
        if (dist <= kMarginUlps * 0x1p-52 * a) {
            const int slot = atomicAdd(unsafe_count, 1);
            if (slot < cap) unsafe_idx[slot] = static_cast<int>(k);
        }
    }
}

// Bulk ingest (rtrec/utils/interactions.py:81-119 applied to a DataFrame-sized batch, rtrec/recommender.py:203-223).
// The batch arrives sorted by (user, item, arrival): order[k] = arrival index of the k-th interaction in that order,
// start[g] .. start[g + 1] = the run of the g-th distinct pair.  One thread folds one pair in arrival order with the
// reference's own float64 arithmetic,
//     current = stored value or 0.0;  new = max(lo, min(current + delta, hi))      (Python's min / max: a NaN sum ends as lo)
// or, with upsert, keeps the last (delta, tstamp).  Runs are short (a user rates an item a handful of times); a long
// one (a replayed stream) is loaded eight occurrences at a time so its gathers overlap, the adds stay sequential.
// Traffic per interaction: 8 B of order + 2 x 8 B gathered (a 64-B sector each when the arrival order is random).
__global__ __launch_bounds__(256) void fold_kernel(const long long *__restrict__ order, const long long *__restrict__ start,
                                                   long long n_groups, const double *__restrict__ delta,
                                                   const double *__restrict__ tstamp, const double *__restrict__ old,
                                                   double lo, double hi, int upsert, double *__restrict__ out_val,
                                                   double *__restrict__ out_ts, float *__restrict__ out_val32) {
    for (long long g = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; g < n_groups;
         g += static_cast<long long>(gridDim.x) * blockDim.x) {
        const long long q0 = start[g], q1 = start[g + 1];
        const long long last = order[q1 - 1];
        double v;
        if (upsert) {
            v = delta[last];
        } else {
            v = old ? old[g] : 0.0;
            long long q = q0;
            for (; q + 8 <= q1; q += 8) {
                double d[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = delta[order[q + j]];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v += d[j];
                    v = hi < v ? hi : v;
                    v = v > lo ? v : lo;
                }
            }
            for (; q < q1; ++q) {
                v += delta[order[q]];
                v = hi < v ? hi : v;
                v = v > lo ? v : lo;
            }
        }
        out_val[g] = v;
        out_ts[g] = tstamp[last];
        if (out_val32) out_val32[g] = static_cast<float>(v);
    }
}

}  // namespace rtrec

extern "C" int rtrec_store_fold_device(const int64_t *d_order, const int64_t *d_start, int64_t n_groups,
                                       const double *d_delta, const double *d_tstamp, const double *d_old, double lo, double hi,
                                       int32_t upsert, double *d_out_val, double *d_out_ts, float *d_out_val32, void *stream) {
    if (n_groups < 0) return RTREC_ERR_INVALID_ARG;
    if (n_groups == 0) return RTREC_OK;
    if (!d_order || !d_start || !d_delta || !d_tstamp || !d_out_val || !d_out_ts) return RTREC_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    const long long blocks = (n_groups + 255) / 256;
    hipLaunchKernelGGL(rtrec::fold_kernel, dim3(static_cast<unsigned>(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st,
                       reinterpret_cast<const long long *>(d_order), reinterpret_cast<const long long *>(d_start),
                       static_cast<long long>(n_groups), d_delta, d_tstamp, d_old, lo, hi, static_cast<int>(upsert),
                       d_out_val, d_out_ts, d_out_val32);
    return rtrec::launch_status();
}

extern "C" int rtrec_store_decay_device(const double *d_val, const double *d_ts, int64_t n, double rate, double now,
                                        float *d_out32, int32_t *d_unsafe_idx, int32_t *d_unsafe_count, int32_t cap,
                                        void *stream) {
    if (n < 0 || n >= (1ll << 31) || cap < 0) return RTREC_ERR_INVALID_ARG;
    if (n == 0) return RTREC_OK;
    if (!d_val || !d_ts || !d_out32 || !d_unsafe_idx || !d_unsafe_count) return RTREC_ERR_INVALID_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    if (hipMemsetAsync(d_unsafe_count, 0, 4, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(rtrec::decay_kernel, dim3(static_cast<unsigned>(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st,
                       d_val, d_ts, static_cast<long long>(n), rate, now, d_out32, d_unsafe_idx, d_unsafe_count, cap);
    return rtrec::launch_status();
}
