// rtrec_amd/csrc/common.hip.h -- wave-level helpers shared by the gfx950 kernels.
//
// All kernels here run ONE 64-lane wavefront per workgroup (blockDim.x == 64): the SLIM hot
// path is made of order-sensitive float32 reductions (sklearn's sequential CD dot products,
// scipy's csr_matmat accumulation order), and a single wave gives program-ordered LDS and
// vector-memory traffic without barriers.  Parallelism comes from thousands of independent
// (target column) / (user, tile) jobs in flight, not from wide workgroups.
//
// Compile with -ffp-contract=off: every float op below must round once (no FMA fusion).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtrec {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return static_cast<int>(threadIdx.x) & 63; }

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint32_t readlane_u(uint32_t v, int lane) {
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane(static_cast<int>(b & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane(static_cast<int>(b >> 32), lane);
    return __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
}
__device__ __forceinline__ float readfirst_f(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ int readfirst_i(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ float readlane_t(float v, int lane) { return readlane_f(v, lane); }
__device__ __forceinline__ double readlane_t(double v, int lane) { return readlane_d(v, lane); }

// Exclusive count of set bits of `mask` below this lane.
__device__ __forceinline__ int lane_prefix(unsigned long long mask) {
    return static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<unsigned int>(mask >> 32),
                                                    __builtin_amdgcn_mbcnt_lo(static_cast<unsigned int>(mask), 0u)));
}

// acc <- (...((acc + p[0]) + p[1]) + ... + p[n-1]) with p[k] = `prod` of lane k: the strictly
// sequential float32 accumulation of sklearn's `tmp += R[X_indices[jj]] * X_data[jj]`
// (_cd_fast.pyx:464-466) and of scipy's csr_matvec.  Every lane computes the same value.
__device__ __forceinline__ float chain_add_full(float acc, float prod) {
#pragma unroll
    for (int k = 0; k < 64; ++k) acc = __fadd_rn(acc, readlane_f(prod, k));
    return acc;
}
__device__ __forceinline__ float chain_add(float acc, float prod, int n) {
    if (n >= 64) return chain_add_full(acc, prod);
    for (int k = 0; k < n; ++k) acc = __fadd_rn(acc, readlane_f(prod, k));
    return acc;
}

template <typename T> struct NegInf;
template <> struct NegInf<float> { __device__ static float value() { return -__builtin_huge_valf(); } };
template <> struct NegInf<double> { __device__ static double value() { return -__builtin_huge_val(); } };

__device__ __forceinline__ float shfl_xor_t(float v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ double shfl_xor_t(double v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ int shfl_xor_t(int v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ uint32_t shfl_xor_t(uint32_t v, int m) {
    return static_cast<uint32_t>(__shfl_xor(static_cast<int>(v), m, 64));
}

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const T o = shfl_xor_t(v, m);
        v = o > v ? o : v;
    }
    return v;
}

// Candidate of a top-k selection: (score, aux, id) ordered lexicographically, larger first.
// id < 0 marks "no candidate".
template <typename T>
struct Cand {
    T score;
    uint32_t aux;
    int id;
};

template <typename T>
__device__ __forceinline__ bool cand_better(const Cand<T> &a, const Cand<T> &b) {
    if (a.id < 0) return false;
    if (b.id < 0) return true;
    if (a.score > b.score) return true;
    if (a.score < b.score) return false;
    if (a.aux != b.aux) return a.aux > b.aux;
    return a.id > b.id;
}

template <typename T>
__device__ __forceinline__ Cand<T> wave_best(Cand<T> c) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Cand<T> o;
        o.score = shfl_xor_t(c.score, m);
        o.aux = shfl_xor_t(c.aux, m);
        o.id = shfl_xor_t(c.id, m);
        if (cand_better(o, c)) c = o;
    }
    return c;
}

template <typename T>
__device__ __forceinline__ Cand<T> cand_readlane(const Cand<T> &c, int lane);
template <>
__device__ __forceinline__ Cand<float> cand_readlane<float>(const Cand<float> &c, int lane) {
    Cand<float> o;
    o.score = readlane_f(c.score, lane);
    o.aux = readlane_u(c.aux, lane);
    o.id = readlane_i(c.id, lane);
    return o;
}
template <>
__device__ __forceinline__ Cand<double> cand_readlane<double>(const Cand<double> &c, int lane) {
    Cand<double> o;
    o.score = readlane_d(c.score, lane);
    o.aux = readlane_u(c.aux, lane);
    o.id = readlane_i(c.id, lane);
    return o;
}

// xorshift32 of sklearn/utils/_random.pxd:20-34 followed by rand_int's `% end`
// (_cd_fast.pyx:29-31).
__device__ __forceinline__ uint32_t rand_int(uint32_t end, uint32_t &state) {
    if (state == 0u) state = 1u;
    state ^= state << 13;
    state ^= state >> 17;
    state ^= state << 5;
    return (state % 2147483648u) % end;
}

// Host side: status of the launches issued since the entry point cleared the error state.
inline int &last_hip_error() {
    static thread_local int e = 0;
    return e;
}
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    last_hip_error() = static_cast<int>(e);
    return e == hipSuccess ? 0 : -4;
}

}  // namespace rtrec
