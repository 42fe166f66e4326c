// Microbenchmark: the latency-mode fit kernel's consumer loop (chain64_dpp over an LDS ring) in
// isolation.  One workgroup of 8 waves per CU (x WGS_PER_CU); wave 0 consumes a pre-filled ring,
// waves 1..7 are "noise" of a selectable kind:
//   0 idle (exit)   1 poll an LDS word with s_sleep 1 (ring-full producers)
//   2 stream global loads (producers gathering)   3 both, alternating
// Reports ns per product of the consumer.   hipcc --offload-arch=gfx950 -O3 consumer_loop.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kRing = 128;

__device__ __forceinline__ float chain64_dpp(float acc, const float4 &p) {
    asm volatile(
        "v_add_f32 %0, %1, %0\n v_add_f32 %0, %2, %0\n v_add_f32 %0, %3, %0\n v_add_f32 %0, %4, %0\n"
#define R(n) "v_add_f32_dpp %0, %1, %0 row_shl:" #n " row_mask:0xf bank_mask:0xf\n" \
             "v_add_f32_dpp %0, %2, %0 row_shl:" #n " row_mask:0xf bank_mask:0xf\n" \
             "v_add_f32_dpp %0, %3, %0 row_shl:" #n " row_mask:0xf bank_mask:0xf\n" \
             "v_add_f32_dpp %0, %4, %0 row_shl:" #n " row_mask:0xf bank_mask:0xf\n"
        R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9) R(10) R(11) R(12) R(13) R(14) R(15)
        : "+v"(acc) : "v"(p.x), "v"(p.y), "v"(p.z), "v"(p.w) : "memory");
    return acc;
}

template <int NOISE, bool PRIO>
__global__ __launch_bounds__(512) void k(int n_rounds, const float *g, size_t g_n, float *out, long long *clk) {
    __shared__ __attribute__((aligned(16))) float ring[kRing * 64];
    __shared__ int ready[kRing];
    __shared__ int done, stop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kRing * 64; i += 512) ring[i] = 1e-3f * (i & 63);
    if (tid < kRing) ready[tid] = tid + 1;
    if (tid == 0) { done = 0; stop = 0; }
    __syncthreads();
    float tmp = 0.0f;
    if (wave == 0) {
        const long long r0 = __builtin_amdgcn_s_memrealtime();
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        for (int round = 0; round < n_rounds; ++round) {
            const int seq = 0, n_chunks = kRing;
            if (lane < 16) {
                const float4 *ring4 = reinterpret_cast<const float4 *>(ring);
                auto ready_flag = [&](int c) {
                    return __hip_atomic_load(&ready[(seq + c) & (kRing - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                };
                auto wait_ready = [&](int c, int flag) {
                    while (flag != ((seq + c) & (kRing - 1)) + 1) { __builtin_amdgcn_s_sleep(1); flag = ready_flag(c); }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                };
                auto ring_read = [&](int c) { return ring4[((seq + c) & (kRing - 1)) * 16 + lane]; };
                wait_ready(0, ready_flag(0));
                float4 p0 = ring_read(0);
                wait_ready(1, ready_flag(1));
                float4 p1 = ring_read(1);
                int flag = ready_flag(2);
                for (int c = 0; c < n_chunks; ++c) {
                    if (c + 2 < n_chunks) wait_ready(c + 2, flag);
                    const float4 p2 = ring_read(c + 2);
                    flag = ready_flag(c + 3);
                    tmp = chain64_dpp(tmp, p0);
                    p0 = p1; p1 = p2;
                    if (((c & 3) == 3 || c + 1 == n_chunks) && lane == 0)
                        __hip_atomic_store(&done, seq + c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        const long long r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            clk[blockIdx.x] = r1 - r0;
            __hip_atomic_store(&stop, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else if (NOISE != 0) {
        size_t o = (static_cast<size_t>(blockIdx.x) * 512 + tid) * 4 % g_n;
        int it = 0;
        while (__hip_atomic_load(&stop, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
            if (NOISE == 1 || (NOISE == 3 && (it & 1))) {
                __builtin_amdgcn_s_sleep(1);
            } else {
                float a = 0;
#pragma unroll
                for (int u = 0; u < 8; ++u) { a += g[o]; o = (o + 512 * 1024 + 64) % g_n; }
                tmp += a;
            }
            ++it;
        }
    }
    out[blockIdx.x * 512 + tid] = tmp;
}

template <int NOISE, bool PRIO>
static void run(const char *name, int blocks, const float *g, size_t g_n, float *o, long long *c) {
    const int n_rounds = 200;
    for (int rep = 0; rep < 2; ++rep) {
        k<NOISE, PRIO><<<blocks, 512>>>(n_rounds, g, g_n, o, c);
        hipDeviceSynchronize();
    }
    long long h[8];
    hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double t = 0;
    for (int i = 0; i < 8; ++i) t += h[i];
    printf("%-34s blocks=%4d  %.2f ns/product\n", name, blocks, t / 8 * 10.0 / (double(n_rounds) * kRing * 64));
}

int main() {
    const size_t g_n = size_t(1) << 28;     // 1 GiB of floats: the noise loads miss the caches
    float *g, *o; long long *c;
    hipMalloc(&g, g_n * 4); hipMemset(g, 0, g_n * 4);
    hipMalloc(&o, 1024 * 512 * 4); hipMalloc(&c, 1024 * 8);
    for (int blocks : {256, 512}) {
        run<0, true>("consumer alone", blocks, g, g_n, o, c);
        run<1, true>("+7 waves polling LDS (s_sleep 1)", blocks, g, g_n, o, c);
        run<2, true>("+7 waves streaming global loads", blocks, g, g_n, o, c);
        run<3, true>("+7 waves polling and loading", blocks, g, g_n, o, c);
        run<3, false>("same, no s_setprio", blocks, g, g_n, o, c);
    }
    return 0;
}
