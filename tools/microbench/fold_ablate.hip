// Ablation of the clean speculative pass: which part costs what (clock ticks per 256-entry group, one wave alone).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 fold_ablate.hip -o fold_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../rtrec_amd/csrc/fold_spec.hip.h"
using namespace rtrec;

// PARTS bit 0: quotient arithmetic (mul, rndne, sub, cvt, lane prefix)   bit 1: wave scan   bit 2: tie compares
//       bit 3: range maximum + compare   bit 4: ballot + branch   bit 5: state decode (readfirstlane, exponent) + result encode
template <int PARTS>
__global__ __launch_bounds__(64) void k(int iters, float *out, long long *clk) {
    const int lane = threadIdx.x & 63;
    float p0 = 1e-3f * lane, p1 = 2e-3f * lane, p2 = 1.5e-3f * lane - 0.01f, p3 = 0.7e-3f * lane;
    float acc = 1.0e6f, sink = 0.0f;
    constexpr uint32_t kLo = 0x800101u, kSpan = 0x7ffdffu;
    const long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float a = acc;
        asm volatile("" : "+v"(a));
        uint32_t bits = 0x49742400u;
        if (PARTS & 32) bits = static_cast<uint32_t>(readfirst_i(__float_as_int(a)));
        const uint32_t ex = (bits >> 23) & 255u, sign = bits & 0x80000000u;
        const float scale = __uint_as_float(((277u - ex) << 23) | sign), ulp = __uint_as_float(((ex - 23u) << 23) | sign);
        const uint32_t M0 = (bits & 0x7fffffu) | 0x800000u;
        uint32_t q0 = 1, q1 = 2, q2 = 3, q3 = 4;
        float d0 = 0.1f, d1 = 0.1f, d2 = 0.1f, d3 = 0.1f;
        if (PARTS & 1) {
            const float s0 = __fmul_rn(p0, scale), s1 = __fmul_rn(p1, scale), s2 = __fmul_rn(p2, scale), s3 = __fmul_rn(p3, scale);
            const float r0 = __builtin_rintf(s0), r1 = __builtin_rintf(s1), r2 = __builtin_rintf(s2), r3 = __builtin_rintf(s3);
            d0 = __fsub_rn(s0, r0); d1 = __fsub_rn(s1, r1); d2 = __fsub_rn(s2, r2); d3 = __fsub_rn(s3, r3);
            q0 = cvt_i32_sat(r0); q1 = cvt_i32_sat(r1); q2 = cvt_i32_sat(r2); q3 = cvt_i32_sat(r3);
        }
        const uint32_t a1 = q0 + q1, a2 = a1 + q2, a3 = a2 + q3;
        uint32_t incl = a3;
        if (PARTS & 2) incl = static_cast<uint32_t>(wave_scan_incl_i(static_cast<int>(a3)));
        const uint32_t baseB = incl - a3 + (M0 - kLo);
        const uint32_t U0 = baseB + q0, U1 = baseB + a1, U2 = baseB + a2, U3 = baseB + a3;
        bool any_odd = false;
        if (PARTS & 4) any_odd = !(fabsf(d0) < 0.5f) || !(fabsf(d1) < 0.5f) || !(fabsf(d2) < 0.5f) || !(fabsf(d3) < 0.5f);
        bool rng = false;
        if (PARTS & 8) rng = max(max(max(U0, U1), U2), U3) >= kSpan;
        uint32_t res = U3;
        if (PARTS & 16) {
            if ((__ballot(any_odd) | __ballot(rng)) != 0ull) res = 0;
        } else {
            res += (any_odd ? 1u : 0u) + (rng ? 1u : 0u);
        }
        if (PARTS & 32) a = __fmul_rn(static_cast<float>(static_cast<int>(readlane_u(res, 63) + kLo)), ulp);
        else a = __uint_as_float(res);
        sink += a;
    }
    const long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x] = c1 - c0;
    if (threadIdx.x == 0) out[blockIdx.x] = sink;
}

template <int PARTS>
static void run(const char *name) {
    const int iters = 20000, grid = 64;
    float *out; long long *clk;
    hipMalloc(&out, grid * 4); hipMalloc(&clk, grid * 8);
    hipLaunchKernelGGL(k<PARTS>, dim3(grid), dim3(64), 0, 0, 100, out, clk);
    hipLaunchKernelGGL(k<PARTS>, dim3(grid), dim3(64), 0, 0, iters, out, clk);
    hipDeviceSynchronize();
    long long h[grid];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
    printf("%-44s %.0f clock ticks per group\n", name, s / grid / iters);
    hipFree(out); hipFree(clk);
}

int main() {
    run<0>("loop skeleton");
    run<1>("quotients");
    run<2>("scan");
    run<3>("quotients + scan");
    run<7>("+ tie compares");
    run<15>("+ range max");
    run<31>("+ ballot/branch");
    run<63>("+ state decode/encode (full clean pass)");
    run<32>("state decode/encode only");
    run<16>("ballot/branch only");
    return 0;
}
