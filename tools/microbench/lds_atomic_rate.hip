// Microbenchmark: LDS float accumulate throughput on gfx950, one 64-lane wave per workgroup,
// random (conflict-light) column addresses.  Variants: ds_add_f32 (no return), ds_add_rtn_f32,
// plain read-modify-write, ds_add_u32.   hipcc --offload-arch=gfx950 -O3 lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void k(const unsigned short *cols, int n_iter, float *out) {
    __shared__ float acc[4096];
    const int lane = threadIdx.x;
    for (int c = lane; c < 4096; c += 64) acc[c] = 0.0f;
    float sink = 0.0f;
    const unsigned short *p = cols + (blockIdx.x % 64) * 4096;
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const int c = p[j * 64 + lane];
            if (MODE == 0) __hip_atomic_fetch_add(&acc[c], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 1) sink += __hip_atomic_fetch_add(&acc[c], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 2) acc[c] = acc[c] + 1.0f;
            else if (MODE == 3) __hip_atomic_fetch_add(reinterpret_cast<unsigned *>(&acc[c]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    float s = sink;
    for (int c = lane; c < 4096; c += 64) s += acc[c];
    out[blockIdx.x * 64 + lane] = s;
}

int main() {
    const int blocks = 256 * 8, n_iter = 50;
    std::vector<unsigned short> h(64 * 4096);
    unsigned x = 12345;
    for (int b = 0; b < 64; ++b)
        for (int j = 0; j < 64; ++j) {
            // 64 distinct columns per instruction (like one W row chunk), random over 4096
            for (int l = 0; l < 64; ++l) { x = x * 1664525u + 1013904223u; h[b * 4096 + j * 64 + l] = (unsigned short)(((x >> 8) % 64) * 64 + l); }
        }
    unsigned short *d; float *o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, blocks * 64 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = {"ds_add_f32", "ds_add_rtn_f32", "read-modify-write", "ds_add_u32"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<blocks, 64>>>(d, n_iter, o);
            if (mode == 1) k<1><<<blocks, 64>>>(d, n_iter, o);
            if (mode == 2) k<2><<<blocks, 64>>>(d, n_iter, o);
            if (mode == 3) k<3><<<blocks, 64>>>(d, n_iter, o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)blocks * n_iter * 64;
            if (rep) printf("%-18s %.3f ms  %.1f Gentries/s  %.1f cycles/wave-instr/CU @2.4GHz\n", names[mode], ms,
                            instr * 64 / ms / 1e6, ms * 1e-3 * 2.4e9 / (instr / 256));
        }
    }
    return 0;
}
