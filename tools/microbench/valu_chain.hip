// Microbenchmark: cost of a strictly dependent v_add_f32 chain on gfx950 (the ordered float32 fold
// that pins the SLIM fit kernel's critical path).  Reports core cycles (s_memtime) and ns
// (s_memrealtime, 100 MHz) per add for
//   dep1     one dependent chain, all 64 lanes
//   dep1_l1  one dependent chain, EXEC = lane 0 only
//   indep2   two independent chains interleaved (issue rate)
//   dep1x8   eight waves per workgroup (two per SIMD), every wave one chain
//   dep1_dpp one dependent chain of v_add_f32_dpp (row_shl on the non-chain operand)
//   dep1_sgpr one dependent chain with the addend in an SGPR
//   dep1+bg  wave 0 chains at s_setprio 3, seven waves run independent VALU work at priority 0
// hipcc --offload-arch=gfx950 -O3 valu_chain.hip -o valu_chain
#include <hip/hip_runtime.h>
#include <cstdio>

#define ADD16(acc, x) \
    asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n" \
                 "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n" \
                 "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n" \
                 "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" \
                 : "+v"(acc) : "v"(x))
#define ADD16x2(a, b, x) \
    asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n" \
                 "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n" \
                 "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n" \
                 "v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" \
                 : "+v"(a), "+v"(b) : "v"(x))

#define ADD16DPP(acc, x) \
    asm volatile("v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n" \
                 "v_add_f32_dpp %0, %1, %0 row_shl:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n" \
                 "v_add_f32_dpp %0, %1, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:6 row_mask:0xf bank_mask:0xf\n" \
                 "v_add_f32_dpp %0, %1, %0 row_shl:7 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n" \
                 "v_add_f32_dpp %0, %1, %0 row_shl:9 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n" \
                 "v_add_f32_dpp %0, %1, %0 row_shl:11 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:12 row_mask:0xf bank_mask:0xf\n" \
                 "v_add_f32_dpp %0, %1, %0 row_shl:13 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:14 row_mask:0xf bank_mask:0xf\n" \
                 "v_add_f32_dpp %0, %1, %0 row_shl:15 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf" \
                 : "+v"(acc) : "v"(x))
#define ADD16S(acc, s) \
    asm volatile("v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n" \
                 "v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n" \
                 "v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n" \
                 "v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0" \
                 : "+v"(acc) : "s"(s))
// MODE 0 dep1, 1 dep1_l1, 2 indep2, 3 all waves chain, 4 wave 0 chains + background VALU
template <int MODE>
__global__ void k(int n_iter, float x, float *out, long long *clk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = lane, acc2 = 1.0f;
    __syncthreads();
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 4 && wave != 0) {
        float b0 = lane, b1 = 1, b2 = 2, b3 = 3;
        for (int it = 0; it < n_iter; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { ADD16x2(b0, b1, x); ADD16x2(b2, b3, x); }
        }
        acc = b0 + b1 + b2 + b3;
    } else {
        if (MODE == 4) __builtin_amdgcn_s_setprio(3);
        if (MODE == 1) {
            if (lane == 0)
                for (int it = 0; it < n_iter; ++it) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) ADD16(acc, x);
                }
        } else if (MODE == 5) {
            for (int it = 0; it < n_iter; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ADD16DPP(acc, x);
            }
        } else if (MODE == 6) {
            const uint32_t sx = __builtin_amdgcn_readfirstlane(__float_as_uint(x));
            for (int it = 0; it < n_iter; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ADD16S(acc, sx);
            }
        } else if (MODE == 2) {
            for (int it = 0; it < n_iter; ++it) {
#pragma unroll
                for (int u = 0; u < 2; ++u) ADD16x2(acc, acc2, x);
            }
        } else {
            for (int it = 0; it < n_iter; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u) ADD16(acc, x);
            }
        }
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + acc2;
    if (lane == 0 && wave == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int n_iter = 20000;
    float *o; long long *c;
    hipMalloc(&o, 1024 * 512 * 4); hipMalloc(&c, 1024 * 16);
    const char *names[7] = {"dep1", "dep1_l1", "indep2", "dep1x8", "dep1+bg", "dep1_dpp", "dep1_sgpr"};
    for (int mode = 0; mode < 7; ++mode) {
        const int threads = (mode == 3 || mode == 4) ? 512 : 64, blocks = 256;
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k<0><<<blocks, threads>>>(n_iter, 1e-3f, o, c);
            if (mode == 1) k<1><<<blocks, threads>>>(n_iter, 1e-3f, o, c);
            if (mode == 2) k<2><<<blocks, threads>>>(n_iter, 1e-3f, o, c);
            if (mode == 3) k<3><<<blocks, threads>>>(n_iter, 1e-3f, o, c);
            if (mode == 4) k<4><<<blocks, threads>>>(n_iter, 1e-3f, o, c);
            if (mode == 5) k<5><<<blocks, threads>>>(n_iter, 1e-3f, o, c);
            if (mode == 6) k<6><<<blocks, threads>>>(n_iter, 1e-3f, o, c);
            hipDeviceSynchronize();
        }
        long long h[2];
        hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        const double adds = (double)n_iter * 64;
        printf("%-8s %.2f s_memtime ticks/add  %.2f ns/add  (per chain add; indep2 counts both chains)\n", names[mode],
               h[0] / adds, h[1] * 10.0 / adds);
    }
    return 0;
}
