// Microbenchmark: is a long straight-line chain of dependent adds bound by the VALU or by instruction fetch?
// One wave runs `iters` passes over a body of 1024 dependent v_add_f32, encoded as
//   e32   4-byte VOP2            dpp   8-byte VOP2 + DPP dword (row_shl)          e64   8-byte VOP3
//   mix   3 x e32 + 1 x dpp per four adds (5 bytes per add on average)
// and reports s_memtime cycles per add.   hipcc --offload-arch=gfx950 -O3 fetch_bound.hip -o fetch_bound
#include <hip/hip_runtime.h>
#include <cstdio>

#define A4_E32 "v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n"
#define A4_E64 "v_add_f32_e64 %0, %1, %0\n v_add_f32_e64 %0, %1, %0\n v_add_f32_e64 %0, %1, %0\n v_add_f32_e64 %0, %1, %0\n"
#define A4_DPP "v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n" \
               "v_add_f32_dpp %0, %1, %0 row_shl:3 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n"
#define A4_MIX "v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32 %0, %1, %0\n v_add_f32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n"
#define X4(a) a a a a
#define X16(a) X4(X4(a))
#define X256(a) X16(X16(a))

template <int KIND, int LANES>
__global__ void k(int iters, float x, float *out, long long *clk) {
    float acc = threadIdx.x;
    const long long c0 = __builtin_amdgcn_s_memtime();
    if (static_cast<int>(threadIdx.x) < LANES) {
        for (int it = 0; it < iters; ++it) {
            if (KIND == 0) asm volatile(X256(A4_E32) : "+v"(acc) : "v"(x));
            if (KIND == 1) asm volatile(X256(A4_DPP) : "+v"(acc) : "v"(x));
            if (KIND == 2) asm volatile(X256(A4_E64) : "+v"(acc) : "v"(x));
            if (KIND == 3) asm volatile(X256(A4_MIX) : "+v"(acc) : "v"(x));
        }
    }
    const long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int KIND, int LANES>
static void run(const char *name, int blocks) {
    float *out; long long *clk;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&clk, blocks * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<KIND, LANES>), dim3(blocks), dim3(64), 0, 0, iters, 1e-3f, out, clk); hipDeviceSynchronize(); }
    long long h[1024];
    hipMemcpy(h, clk, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int b = 0; b < blocks; ++b) s += h[b];
    printf("%-30s blocks %4d: %.3f cycles per add\n", name, blocks, s / blocks / iters / 1024.0);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int blocks : {1, 256}) {
        run<0, 64>("e32 (4 B)", blocks);
        run<1, 64>("dpp (8 B)", blocks);
        run<1, 16>("dpp (8 B), 16 lanes", blocks);
        run<2, 64>("e64 (8 B)", blocks);
        run<3, 64>("3 e32 + 1 dpp (5 B)", blocks);
    }
    return 0;
}
