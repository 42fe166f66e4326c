// Microbenchmark: the latency-mode ordered fold (rtrec::mw_fold<0, false> of csrc/fit.hip: seven producer waves, the LDS
// ring, the chain consumer) and the residual update that follows it, as the fit kernel runs them, on one synthetic popular
// column -- the inner loop of the heaviest target of a mini-batch without the rest of the kernel around it.
// Reports consumer clock cycles (s_memtime) and ns (wall clock) per folded entry and per updated entry.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off mw_fold_bench.hip -o mw_fold_bench
//   ./mw_fold_bench [blocks=1] [entries=52000] [users=138493] [folds=200]
#include "../../rtrec_amd/csrc/fit.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace rtrec;

// MODE 0: fold only; 1: fold + stashed update; 2: fold + gathered update
template <int MODE>
__global__ __launch_bounds__(kMwThreads, 4) void k(const int *crow, const float *cval, float *R_all, float *stash_all, int n, int U,
                                                   int folds, float *out, long long *clk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const MwLds M = carve_mw(smem, 0);
    float *R = R_all + static_cast<size_t>(blockIdx.x) * U;
    float *stash = stash_all + static_cast<size_t>(blockIdx.x) * U;
    if (tid < kRing) M.ready[tid] = 0;
    if (tid == 0) *M.done = 0;
    int seq = 0;
    __syncthreads();
    float acc = 0.0f;
    long long cyc_fold = 0, cyc_upd = 0, t_fold = 0, t_upd = 0;
    for (int f = 0; f < folds; ++f) {
        const float w_old = 0.25f;
        long long c0 = __builtin_amdgcn_s_memtime(), r0 = static_cast<long long>(wall_clock64());
        const float tmp = mw_fold<0, false>(crow, cval, R, M, 0, n, w_old, wave, lane, seq, 0x7fffffff, nullptr, MODE == 1 ? stash : nullptr);
        __syncthreads();
        cyc_fold += __builtin_amdgcn_s_memtime() - c0; t_fold += static_cast<long long>(wall_clock64()) - r0;
        acc += tmp;
        if (MODE != 0) {
            c0 = __builtin_amdgcn_s_memtime(); r0 = static_cast<long long>(wall_clock64());
            const float w_new = 0.25f;        // R returns to what it was: every fold sees the same sums
            if (MODE == 1) mw_update_stashed(crow, cval, stash, R, 0, n, w_new, tid);
            else mw_update(crow, cval, R, 0, n, w_old, w_new, tid);
            __syncthreads();
            cyc_upd += __builtin_amdgcn_s_memtime() - c0; t_upd += static_cast<long long>(wall_clock64()) - r0;
        }
    }
    if (tid == 0) {
        out[blockIdx.x] = acc;
        clk[blockIdx.x * 4 + 0] = cyc_fold; clk[blockIdx.x * 4 + 1] = t_fold;
        clk[blockIdx.x * 4 + 2] = cyc_upd;  clk[blockIdx.x * 4 + 3] = t_upd;
    }
}

template <int MODE>
static void run(const char *name, int blocks, const int *crow, const float *cval, float *R, float *stash, int n, int U, int folds,
                float *out, long long *clk) {
    const size_t lds = mw_lds_bytes(0);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(kMwThreads), lds, 0, crow, cval, R, stash, n, U, rep == 0 ? 3 : folds, out, clk);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); exit(1); }
    }
    std::vector<long long> h(static_cast<size_t>(blocks) * 4);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    float o0;
    hipMemcpy(&o0, out, 4, hipMemcpyDeviceToHost);
    double c[4] = {0, 0, 0, 0};
    for (int b = 0; b < blocks; ++b) for (int i = 0; i < 4; ++i) c[i] += static_cast<double>(h[b * 4 + i]) / blocks;
    const double e = static_cast<double>(folds) * n;
    printf("%-28s blocks %4d  fold %.3f cycles/entry %.3f ns/entry   update %.3f cycles/entry %.3f ns/entry (%.1f us each)   sum %.9g\n", name,
           blocks, c[0] / e, c[1] * 10.0 / e, c[2] / e, c[3] * 10.0 / e, c[3] * 0.01 / folds, o0);
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 1;
    const int n = argc > 2 ? atoi(argv[2]) : 52000;
    const int U = argc > 3 ? atoi(argv[3]) : 138493;
    const int folds = argc > 4 ? atoi(argv[4]) : 200;
    std::vector<int> rows(n);
    std::vector<float> vals(n), Rh(static_cast<size_t>(U));
    unsigned s = 12345u;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    {   // n distinct sorted rows out of U
        std::vector<char> pick(U, 0);
        int got = 0;
        while (got < n) { const int r = rnd() % U; if (!pick[r]) { pick[r] = 1; ++got; } }
        int o = 0;
        for (int r = 0; r < U; ++r) if (pick[r]) rows[o++] = r;
    }
    for (int i = 0; i < n; ++i) vals[i] = 0.5f * static_cast<float>(1 + rnd() % 10);
    for (int r = 0; r < U; ++r) Rh[r] = 0.5f * static_cast<float>(rnd() % 10) - 1.0f;
    int *crow; float *cval, *R, *stash, *out; long long *clk;
    hipMalloc(&crow, n * 4); hipMalloc(&cval, n * 4);
    hipMalloc(&R, static_cast<size_t>(blocks) * U * 4); hipMalloc(&stash, static_cast<size_t>(blocks) * U * 4);
    hipMalloc(&out, blocks * 4); hipMalloc(&clk, static_cast<size_t>(blocks) * 32);
    hipMemcpy(crow, rows.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(cval, vals.data(), n * 4, hipMemcpyHostToDevice);
    for (int b = 0; b < blocks; ++b) hipMemcpy(R + static_cast<size_t>(b) * U, Rh.data(), static_cast<size_t>(U) * 4, hipMemcpyHostToDevice);
    printf("column of %d entries over %d users, %d folds per workgroup\n", n, U, folds);
    run<0>("fold", blocks, crow, cval, R, stash, n, U, folds, out, clk);
    run<2>("fold + gathered update", blocks, crow, cval, R, stash, n, U, folds, out, clk);
    run<1>("fold + stashed update", blocks, crow, cval, R, stash, n, U, folds, out, clk);
    return 0;
}
