// rtrec_amd/csrc/ordered_fold.hip -- left-to-right float32 sums of array segments (one wave per segment).
//
// The primitive every order-sensitive reduction of the fit path is built from (csrc/fold_spec.hip.h: the reference's
// sequential accumulation, sklearn _cd_fast.pyx:464-466, without the dependent-add chain).  Exported so that the
// speculative fold can be checked entry stream by entry stream against the literal chain (tests/test_gpu_kernels.py:
// ties, cancellations, infinities, long drifting sums) and timed on its own (tools/fold_bench.py).
#include "common.hip.h"
#include "fold_spec.hip.h"
#include "../../include/rtrec_amd.h"

namespace rtrec {

struct __attribute__((packed, aligned(4))) PackedF4 { float x, y, z, w; };

// MODE 0: fold256_spec (one group per call), 1: the literal chain, 2 / 3: fold_groups_spec<2> / <4> (several groups per
// call, the running value in an SGPR from group to group)
template <int MODE>
__global__ __launch_bounds__(64) void ordered_sums_kernel(const float *__restrict__ v, const long long *__restrict__ off,
                                                          int n_sums, float *__restrict__ out) {
    const int lane = lane_id();
    for (int sidx = blockIdx.x; sidx < n_sums; sidx += gridDim.x) {
        const long long b = off[sidx], e = off[sidx + 1];
        float acc = 0.0f;
        if (MODE == 1) {
            for (long long o = b; o < e; o += 64) {
                const int n = static_cast<int>(e - o < 64 ? e - o : 64);
                const float p = lane < n ? v[o + lane] : 0.0f;
                acc = chain_add(acc, p, n);
            }
        } else {
            constexpr int G = MODE == 0 ? 1 : (MODE == 2 ? 2 : 4);
            constexpr int W = G * kFoldGroupEntries;
            auto load = [&](long long o, float (&p)[G][4]) {                 // a full window, 16 bytes per lane and group
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const PackedF4 q = *reinterpret_cast<const PackedF4 *>(v + o + g * kFoldGroupEntries + 4 * lane);
                    p[g][0] = q.x; p[g][1] = q.y; p[g][2] = q.z; p[g][3] = q.w;
                }
            };
            long long o = b;
            if (o + W <= e) {                    // one window ahead of the fold: the loads are off the critical path
                float p[G][4], p1[G][4];
                load(o, p);
                for (; o + W <= e; o += W) {
                    const bool more = o + 2 * W <= e;
                    if (more) load(o + W, p1);
                    acc = fold_groups_spec<G>(acc, p);
                    if (more) {
#pragma unroll
                        for (int g = 0; g < G; ++g)
#pragma unroll
                            for (int k = 0; k < 4; ++k) p[g][k] = p1[g][k];
                    }
                }
            }
            if (o < e) {
                const int n = static_cast<int>(e - o);
                float p[G][4];
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = g * kFoldGroupEntries + 4 * lane + k;
                        p[g][k] = i < n ? v[o + i] : 0.0f;
                    }
                acc = fold_groups_spec<G>(acc, p, n);
            }
        }
        if (lane == 0) out[sidx] = acc;
    }
}

}  // namespace rtrec

using namespace rtrec;

extern "C" int rtrec_slim_ordered_sums(const float *d_values, const int64_t *d_offsets, int32_t n_sums, int32_t mode,
                                       float *d_out, void *stream) {
    if (n_sums < 0 || mode < 0 || mode > 3) return RTREC_ERR_INVALID_ARG;
    if (n_sums == 0) return RTREC_OK;
    if (!d_values || !d_offsets || !d_out) return RTREC_ERR_INVALID_ARG;
    (void)hipGetLastError();
    const int grid = n_sums < 8192 ? n_sums : 8192;
    const long long *off = reinterpret_cast<const long long *>(d_offsets);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (mode) {
    case 0: hipLaunchKernelGGL(HIP_KERNEL_NAME(ordered_sums_kernel<0>), dim3(grid), dim3(64), 0, st, d_values, off, n_sums, d_out); break;
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(ordered_sums_kernel<1>), dim3(grid), dim3(64), 0, st, d_values, off, n_sums, d_out); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(ordered_sums_kernel<2>), dim3(grid), dim3(64), 0, st, d_values, off, n_sums, d_out); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(ordered_sums_kernel<3>), dim3(grid), dim3(64), 0, st, d_values, off, n_sums, d_out); break;
    }
    return rtrec::launch_status();
}
