// Microbenchmark: clock ticks of fold_groups_spec<G> (csrc/fold_spec.hip.h) with the products in registers -- no memory
// in the loop.  Variants: clean groups (one scan each, no failure), one failure per call, and the literal chain
// (chain64_dpp per 64-entry row), one wave alone on its SIMD.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 fold_pass.hip -o fold_pass && ./fold_pass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../rtrec_amd/csrc/fold_spec.hip.h"

using namespace rtrec;

// KIND 0: clean speculative groups; 1: one failure in the middle of group 0 (a product that leaves the binade);
//      2: chain64_dpp rows (the literal chain)
template <int KIND, int G>
__global__ __launch_bounds__(256) void k(int iters, float *out, long long *clk) {
    const int lane = threadIdx.x & 63;
    float p[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int u = 0; u < 4; ++u) p[g][u] = 1e-3f * static_cast<float>(((4 * lane + u + 7 * g) * 37) % 101) - 0.02f;
    if (KIND == 1 && lane == 30) p[0][1] = 3.0e6f;          // entry 121 throws the sum into another binade
    float acc = 1.0e6f, sink = 0.0f;
    const long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float a = acc;
        asm volatile("" : "+v"(a));                        // a fresh, opaque start value every iteration
        if (KIND == 2) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float4 m; m.x = p[g][0]; m.y = p[g][1]; m.z = p[g][2]; m.w = p[g][3];
                for (int row = 0; row < 4; ++row) a = readlane_f(chain64_dpp(a, m), row * 16);
            }
        } else {
            a = fold_groups_spec<G>(a, p);
        }
        sink += a;
    }
    const long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
    if (threadIdx.x == 0) out[blockIdx.x] = sink;
}

template <int KIND, int G>
static void run(const char *name, int waves) {
    const int iters = 20000, grid = 256;
    float *out; long long *clk;
    hipMalloc(&out, grid * 4); hipMalloc(&clk, grid * 4 * 8);
    hipLaunchKernelGGL((k<KIND, G>), dim3(grid), dim3(64 * waves), 0, 0, 100, out, clk);
    hipLaunchKernelGGL((k<KIND, G>), dim3(grid), dim3(64 * waves), 0, 0, iters, out, clk);
    hipDeviceSynchronize();
    long long h[grid * 4];
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid; ++i) s += h[i * 4];
    printf("%-26s G=%d waves/WG %d: %.0f ticks per 256-entry group = %.2f ticks/entry\n", name, G, waves, s / grid / iters / G, s / grid / iters / G / 256);
    hipFree(out); hipFree(clk);
}

int main() {
    run<0, 1>("clean", 1); run<0, 2>("clean", 1); run<0, 4>("clean", 1);
    run<1, 1>("one failure per call", 1); run<1, 2>("one failure per call", 1); run<1, 4>("one failure per call", 1);
    run<2, 1>("chain64_dpp", 1);
    run<0, 4>("clean", 4); run<2, 1>("chain64_dpp", 4);
    return 0;
}
