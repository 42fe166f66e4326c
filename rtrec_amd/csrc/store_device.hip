// rtrec_amd/csrc/store_device.hip -- device side of the interaction store: time decay of the resident matrix.
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

}  // namespace rtrec

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
