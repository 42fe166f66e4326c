// rtrec_amd/csrc/fold_spec.hip.h -- the ordered float32 fold without the chain of dependent additions.
//
// The reference accumulates its dot products strictly left to right in float32
// (sklearn/linear_model/_cd_fast.pyx:464-466 `tmp += R[X_indices[jj]] * X_data[jj]`, :506-509 XtA; scipy csr_matvec),
// and every coefficient bit and sweep count depends on that order.  A literal transcription is one dependent
// v_add_f32 per entry (4.56 cycles each on gfx950, csrc/common.hip.h chain_add) -- the critical path of a popular
// target.  This file computes THE SAME float, bit for bit, with integer prefix sums:
//
//   While the running sum stays inside one binade, acc = +-M u with u = 2^(e-23) and M an integer in [2^23, 2^24).
//   The exact sum acc + p = +-(M + p/u) u is rounded to the grid u, so RN(acc + p) = +-(M + RNI(p/u)) u unless p/u
//   is exactly half way between two integers.  A run of sequentially rounded additions is therefore an INTEGER
//   prefix sum of r_i = rndne(p_i / u) -- associative, one wave-wide scan per 256 entries.
//   * A tie (|p_i/u - r_i| == 1/2) rounds to the EVEN grid point: M -> M + r_i when M is even (r_i is the even
//     neighbour of p_i/u), M + r_i + 2 d_i when M is odd; either way the value after a tie is even.  With plain
//     prefixes A_i = M0 + sum_{k<=i} r_k the true running value is A_i + C_i, C_i the sum of the corrections
//     c_t = 2 d_t ((A_{t-1} + C_{t-1}) & 1) over the ties t <= i: a scalar walk over the tie entries only.
//   * Every running value must stay strictly inside (2^23, 2^24) (same binade, same grid).  |C_i| <= 256, so the
//     test is made on A_i with a margin of 256 on both sides; the first entry that fails it (or whose quotient is
//     not finite) is added with ONE real float addition and the run restarts from the new sum.
//   * A running sum without an integer image (zero, subnormal, below 2^-103, inf, nan) and stretches where the sum
//     crosses a binade every few additions (the first entries of a fold, a sum hovering at a power of two) are added
//     with real float additions, whole 64-entry rows at a time (chain64_dpp: one dependent v_add_f32_dpp per entry).
//   Precondition: the fold starts from +0.0 (or any value other than -0.0), as every dot product of the reference does.
// The control flow is restated for the CPU in oracle/fold_model.c (test infrastructure) and checked there against
// the plain sequential loop on 1e6+ generated sums and against the oracle's coordinate descent on the goldens;
// tests/test_gpu_kernels.py feeds the same streams (ties, cancellations, infinities) to the device function.
#pragma once

#include "common.hip.h"

namespace rtrec {

constexpr int kFoldGroupEntries = 256;     // one call: lane L holds entries 4L .. 4L+3
constexpr int kFoldMinAdvance = 32;        // a pass that absorbed fewer entries is followed by a serial stretch
constexpr int kFoldSerialLead = 16;        // a serial stretch from pos ends with the 64-entry row that holds entry pos + 16

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int fold_dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}

// inclusive prefix sums of one int per lane (wrapping): four row_shr steps inside each row of 16 lanes, then the two
// row broadcasts -- six DPP adds, no LDS round trip
__device__ __forceinline__ int wave_scan_incl_i(int x) {
    x += fold_dpp_i<0x111, 0xf>(x);      // row_shr:1
    x += fold_dpp_i<0x112, 0xf>(x);      // row_shr:2
    x += fold_dpp_i<0x114, 0xf>(x);      // row_shr:4
    x += fold_dpp_i<0x118, 0xf>(x);      // row_shr:8
    x += fold_dpp_i<0x142, 0xa>(x);      // row_bcast:15 -> rows 1, 3
    x += fold_dpp_i<0x143, 0xc>(x);      // row_bcast:31 -> rows 2, 3
    return x;
}

__device__ __forceinline__ int cvt_i32_sat(float q) {      // v_cvt_i32_f32: saturates, nan -> 0 (a C++ cast would be undefined)
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(q));
    return r;
}

// Ordered fold of one 64-product chunk held as one float4 per lane in the 16 lanes of a DPP row (product i = component
// i%4 of row lane i/4): the row's first lane adds them left to right, fetching row lane n's components with the DPP
// modifier row_shl:n on the product operand (the first DPP read comes four instructions after the block starts, so a
// VALU-written operand has passed its DPP wait states); the accumulator is the plain second source.  64 instructions
// per 64 products and nothing else on the chain: no broadcast reads, no v_readlane, no SGPR traffic.  The row's 16
// lanes must be enabled in EXEC (DPP does not read disabled lanes); only the row's first lane holds the result.
// Every row of the wave folds ITS OWN 64 products from the `acc` its first lane was given.
__device__ __forceinline__ float chain64_dpp(float acc, const float4 &p) {
    asm volatile(
        "v_add_f32 %0, %1, %0\n\t"
        "v_add_f32 %0, %2, %0\n\t"
        "v_add_f32 %0, %3, %0\n\t"
        "v_add_f32 %0, %4, %0\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %1, %0 row_shl:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %2, %0 row_shl:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %3, %0 row_shl:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %4, %0 row_shl:15 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(p.x), "v"(p.y), "v"(p.z), "v"(p.w)
        : "memory");
    return acc;
}


// Entries [from, to) of the group added one after the other with real float additions; `to` is a multiple of 64 or the
// end of the data (uniform bounds, from < to).  Lane L's four entries lie in DPP row L / 16, so a 64-entry row of the group
// is one chain64_dpp; entries before `from` are replaced by +0.0 (acc + 0 == acc: the running sum is never -0), entries
// past the data are +0.0 already.
__device__ __forceinline__ float fold_serial_rows(float acc, float p0, float p1, float p2, float p3, int from, int to) {
    const int lin0 = 4 * lane_id();
    float4 m;
    m.x = lin0 + 0 >= from ? p0 : 0.0f; m.y = lin0 + 1 >= from ? p1 : 0.0f;
    m.z = lin0 + 2 >= from ? p2 : 0.0f; m.w = lin0 + 3 >= from ? p3 : 0.0f;
    for (int row = from >> 6; row * 64 < to; ++row)
        acc = readlane_f(chain64_dpp(acc, m), row * 16);
    return acc;
}
__device__ __forceinline__ int fold_stretch_end(int pos, int n) { return min(((pos + kFoldSerialLead) | 63) + 1, n); }

struct FoldQuant { uint32_t q0, a1, a2, a3, incl; bool odd; };

// quotients of one group's four entries per lane, their lane prefixes and the wave scan of the lane totals
template <bool MASKED>
__device__ __forceinline__ FoldQuant fold_quantise(const float (&p)[4], float scale, int lin0, int lim) {
    const float s0 = (!MASKED || lin0 + 0 >= lim) ? __fmul_rn(p[0], scale) : 0.0f, s1 = (!MASKED || lin0 + 1 >= lim) ? __fmul_rn(p[1], scale) : 0.0f;
    const float s2 = (!MASKED || lin0 + 2 >= lim) ? __fmul_rn(p[2], scale) : 0.0f, s3 = (!MASKED || lin0 + 3 >= lim) ? __fmul_rn(p[3], scale) : 0.0f;
    const float r0 = __builtin_rintf(s0), r1 = __builtin_rintf(s1), r2 = __builtin_rintf(s2), r3 = __builtin_rintf(s3);
    FoldQuant Q;
    Q.odd = !(fabsf(__fsub_rn(s0, r0)) < 0.5f) || !(fabsf(__fsub_rn(s1, r1)) < 0.5f) ||
            !(fabsf(__fsub_rn(s2, r2)) < 0.5f) || !(fabsf(__fsub_rn(s3, r3)) < 0.5f);
    Q.q0 = static_cast<uint32_t>(cvt_i32_sat(r0));
    Q.a1 = Q.q0 + static_cast<uint32_t>(cvt_i32_sat(r1));
    Q.a2 = Q.a1 + static_cast<uint32_t>(cvt_i32_sat(r2));
    Q.a3 = Q.a2 + static_cast<uint32_t>(cvt_i32_sat(r3));
    Q.incl = static_cast<uint32_t>(wave_scan_incl_i(static_cast<int>(Q.a3)));
    return Q;
}

// ---------------------------------------------------------------------------------------------------------------------
// One 256-entry group against the running value M u (entries below `pos` already in it), given its quantisation Q.
//
// Range test: with B = A - (2^23 + 257) every running value lies in (2^23 + 256, 2^24 - 256) iff max B <u 2^23 - 513 (values
// below the range wrap to huge unsigned numbers) -- one unsigned maximum per lane instead of a compare per entry.  A clean
// group returns done = true with the new sum.  Otherwise the group is finished with the data at hand: tie corrections by
// a scalar walk over the tie entries; the first failing entry j by ONE real addition (done = false, pos = j + 1: the later
// entries must be quantised again in the new binade).  The clean path is straight-line code with not-taken branches only:
// a taken branch costs a solo wave an instruction fetch (measured, tools/microbench/fold_pass.hip: the same pass took 824
// instead of 468 clock ticks with a dozen of them on it).
// ---------------------------------------------------------------------------------------------------------------------
struct FoldStep { float acc; int pos; int adv; bool done; };

template <bool MASKED>
__device__ __forceinline__ FoldStep fold_group_finish(const float (&p)[4], const FoldQuant &Q, uint32_t M, float scale, float ulp, int pos) {
    constexpr uint32_t kLo = 0x800101u, kSpan = 0x7ffdffu;
    const int lin0 = 4 * lane_id();
    const uint32_t baseB = Q.incl - Q.a3 + (M - kLo);                            // B before this lane's entries
    const uint32_t U0 = baseB + Q.q0, U1 = baseB + Q.a1, U2 = baseB + Q.a2, U3 = baseB + Q.a3;
    // masked entries must not fail the range test (a running value inside the margin would never advance)
    const uint32_t B0 = (!MASKED || lin0 + 0 >= pos) ? U0 : 0u, B1 = (!MASKED || lin0 + 1 >= pos) ? U1 : 0u;
    const uint32_t B2 = (!MASKED || lin0 + 2 >= pos) ? U2 : 0u, B3 = (!MASKED || lin0 + 3 >= pos) ? U3 : 0u;
    const uint32_t bmax = max(max(max(B0, B1), B2), B3);
    const unsigned long long odd_m = __ballot(Q.odd), range_m = __ballot(bmax >= kSpan);
    FoldStep st;
    st.pos = kFoldGroupEntries; st.adv = kFoldGroupEntries; st.done = true;
    if (__builtin_expect((odd_m | range_m) == 0ull, 1)) {                        // the common case: no addition at all
        st.acc = __fmul_rn(static_cast<float>(static_cast<int>(readlane_u(U3, 63) + kLo)), ulp);
        return st;
    }
    // ---- first entry that leaves the binade (or is not finite), ties before it ----
    bool f0 = B0 >= kSpan, f1 = B1 >= kSpan, f2 = B2 >= kSpan, f3 = B3 >= kSpan;
    bool t0 = false, t1 = false, t2 = false, t3 = false;
    float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
    if (__builtin_expect(odd_m != 0ull, 0)) {                                    // exact ties, or inf / nan quotients
        const float s0 = (!MASKED || lin0 + 0 >= pos) ? __fmul_rn(p[0], scale) : 0.0f, s1 = (!MASKED || lin0 + 1 >= pos) ? __fmul_rn(p[1], scale) : 0.0f;
        const float s2 = (!MASKED || lin0 + 2 >= pos) ? __fmul_rn(p[2], scale) : 0.0f, s3 = (!MASKED || lin0 + 3 >= pos) ? __fmul_rn(p[3], scale) : 0.0f;
        d0 = __fsub_rn(s0, __builtin_rintf(s0)); d1 = __fsub_rn(s1, __builtin_rintf(s1));
        d2 = __fsub_rn(s2, __builtin_rintf(s2)); d3 = __fsub_rn(s3, __builtin_rintf(s3));
        const bool o0 = !(fabsf(d0) < 0.5f), o1 = !(fabsf(d1) < 0.5f), o2 = !(fabsf(d2) < 0.5f), o3 = !(fabsf(d3) < 0.5f);
        t0 = o0 && fabsf(d0) == 0.5f; t1 = o1 && fabsf(d1) == 0.5f; t2 = o2 && fabsf(d2) == 0.5f; t3 = o3 && fabsf(d3) == 0.5f;
        f0 = f0 || (o0 && !t0); f1 = f1 || (o1 && !t1); f2 = f2 || (o2 && !t2); f3 = f3 || (o3 && !t3);
    }
    // Per lane: index of its first failing entry (4: none), the running value before that entry and the entry itself --
    // selected by the lane's own flags, so that locating j needs no uniform branch: one ballot, one s_ff1, three v_readlane.
    const int kf = f0 ? 0 : (f1 ? 1 : (f2 ? 2 : (f3 ? 3 : 4)));
    const uint32_t Vb = f0 ? baseB : (f1 ? U0 : (f2 ? U1 : U2));
    const float Pf = f0 ? p[0] : (f1 ? p[1] : (f2 ? p[2] : p[3]));
    const unsigned long long fm = __ballot(kf < 4);
    const int jl = fm ? static_cast<int>(__builtin_ctzll(fm)) : 63;
    const int j = fm ? 4 * jl + readlane_i(kf, jl) : kFoldGroupEntries;
    int C = 0;
    if (__builtin_expect(odd_m != 0ull, 0)) {                                    // tie corrections of the entries before j, in entry order
        const unsigned long long m0 = __ballot(t0 && lin0 + 0 < j), m1 = __ballot(t1 && lin0 + 1 < j);
        const unsigned long long m2 = __ballot(t2 && lin0 + 2 < j), m3 = __ballot(t3 && lin0 + 3 < j);
        unsigned long long m = m0 | m1 | m2 | m3;
        if (m) {
            // parity of the plain running value BEFORE the entry (A = B + 2^23 + 257: opposite parity), direction of the remainder
            const unsigned long long e0 = __ballot(!(baseB & 1u)), e1 = __ballot(!(U0 & 1u)), e2 = __ballot(!(U1 & 1u)), e3 = __ballot(!(U2 & 1u));
            const unsigned long long g0 = __ballot(d0 > 0.0f), g1 = __ballot(d1 > 0.0f), g2 = __ballot(d2 > 0.0f), g3 = __ballot(d3 > 0.0f);
            while (m) {
                const int L = __builtin_ctzll(m);
                m &= m - 1;
                if (((m0 >> L) & 1ull) && ((((e0 >> L) & 1ull) != 0) != ((C & 1) != 0))) C += ((g0 >> L) & 1ull) ? 1 : -1;
                if (((m1 >> L) & 1ull) && ((((e1 >> L) & 1ull) != 0) != ((C & 1) != 0))) C += ((g1 >> L) & 1ull) ? 1 : -1;
                if (((m2 >> L) & 1ull) && ((((e2 >> L) & 1ull) != 0) != ((C & 1) != 0))) C += ((g2 >> L) & 1ull) ? 1 : -1;
                if (((m3 >> L) & 1ull) && ((((e3 >> L) & 1ull) != 0) != ((C & 1) != 0))) C += ((g3 >> L) & 1ull) ? 1 : -1;
            }
        }
        if (j >= kFoldGroupEntries) {                                            // ties only: the group is absorbed
            st.acc = __fmul_rn(static_cast<float>(static_cast<int>(readlane_u(U3, 63) + kLo) + C), ulp);
            return st;
        }
    }
    // running value before entry j, then the one real addition (a range failure always has a failing lane: fm != 0 here)
    st.acc = __fadd_rn(__fmul_rn(static_cast<float>(static_cast<int>(readlane_u(Vb, jl) + kLo) + C), ulp), readlane_f(Pf, jl));
    st.adv = j - pos;
    st.pos = max(j, pos) + 1;                  // j >= pos by construction; the max only makes progress unconditional
    st.done = false;
    return st;
}

// One 256-entry group folded into acc from entry `pos` on: a speculative pass per turn until the group is absorbed.  A sum
// that changes binade every few entries (the start of a fold, a sum hovering at a power of two) gets real additions.
__device__ __forceinline__ float fold_group_loop(float acc, const float (&p)[4], int n, int pos, int adv) {
    const int lin0 = 4 * lane_id();
    for (;;) {
        if ((adv < kFoldMinAdvance || n - pos <= 24) && pos < n) {
            const int to = fold_stretch_end(pos, n);
            acc = fold_serial_rows(acc, p[0], p[1], p[2], p[3], pos, to);
            pos = to;
        }
        if (pos >= n) return acc;
        const uint32_t bits = static_cast<uint32_t>(readfirst_i(__float_as_int(acc)));
        const uint32_t ex = (bits >> 23) & 255u;
        if (__builtin_expect(ex < 24u || ex == 255u, 0)) { adv = 0; continue; }  // no integer image (zero, tiny, inf, nan): serial stretch
        const uint32_t sign = bits & 0x80000000u;
        const float scale = __uint_as_float(((277u - ex) << 23) | sign);         // +-1/u
        const float ulp = __uint_as_float(((ex - 23u) << 23) | sign);            // +-u
        const FoldQuant Q = fold_quantise<true>(p, scale, lin0, pos);            // entries below pos are already in acc
        const FoldStep st = fold_group_finish<true>(p, Q, (bits & 0x7fffffu) | 0x800000u, scale, ulp, pos);
        acc = st.acc;
        if (st.done) return acc;
        pos = st.pos; adv = st.adv;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fold_groups_spec<G>: G consecutive 256-entry groups (lane L holds entries 4L .. 4L+3 of every group) folded into acc.
//
// The quotients r_i = rndne(p_i / u) and their wave-wide prefix sums depend only on the binade of the running sum, not on
// its value, so for a run of groups in one binade everything per entry (multiply, round, remainder, convert, lane prefix,
// one DPP scan per group) is independent work the wave issues back to back; the only chain from group to group is
// M <- v_readlane(lane 63's running value) and the range / tie test of the next group against it.  The running value stays
// an integer in an SGPR across the groups.  Measured (tools/microbench/fold_ablate.hip, fold_pass.hip): the float <->
// integer round trip per group costs as much as the scan and the arithmetic together; clean groups cost 480 / 376 / 325
// clock ticks each for G = 1 / 2 / 4 against 1244 for 256 dependent additions.
// The first group that holds a tie or leaves the binade is finished from its quantisation at hand (fold_group_finish,
// then fold_group_loop for what follows the real addition); the groups after it take fold_group_loop from their start.
// ---------------------------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ float fold_groups_spec(float acc, const float (&p)[G][4], int n = G * kFoldGroupEntries) {
    constexpr uint32_t kLo = 0x800101u, kSpan = 0x7ffdffu;
    const int lin0 = 4 * lane_id();
    int g = 0, pos = 0, adv = kFoldGroupEntries;     // next entry: `pos` of group g
    {
        const uint32_t bits = static_cast<uint32_t>(readfirst_i(__float_as_int(acc)));
        const uint32_t ex = (bits >> 23) & 255u;
        if (__builtin_expect(ex >= 24u && ex != 255u, 1)) {
            const uint32_t sign = bits & 0x80000000u;
            const float scale = __uint_as_float(((277u - ex) << 23) | sign);
            const float ulp = __uint_as_float(((ex - 23u) << 23) | sign);
            FoldQuant Q[G];
#pragma unroll
            for (int gg = 0; gg < G; ++gg) Q[gg] = fold_quantise<false>(p[gg], scale, lin0, 0);
            uint32_t M[G + 1];
            M[0] = (bits & 0x7fffffu) | 0x800000u;
            int first_bad = G;
#pragma unroll
            for (int gg = 0; gg < G; ++gg) {
                const uint32_t baseB = Q[gg].incl - Q[gg].a3 + (M[gg] - kLo);
                const uint32_t U3 = baseB + Q[gg].a3;
                const uint32_t bmax = max(max(max(baseB + Q[gg].q0, baseB + Q[gg].a1), baseB + Q[gg].a2), U3);
                const bool bad = (__ballot(Q[gg].odd || bmax >= kSpan) != 0ull);
                if (bad && first_bad == G) first_bad = gg;       // scalar select, no branch
                M[gg + 1] = readlane_u(U3, 63) + kLo;            // meaningful while no earlier group was bad
            }
            if (__builtin_expect(first_bad == G, 1)) return __fmul_rn(static_cast<float>(static_cast<int>(M[G])), ulp);
            // the groups before first_bad are absorbed; first_bad is finished from its quantisation at hand
            g = first_bad;
#pragma unroll
            for (int gg = 0; gg < G; ++gg) {
                if (gg != first_bad) continue;               // (not `g`: a ties-only group advances g inside this loop)
                const FoldStep st = fold_group_finish<false>(p[gg], Q[gg], M[gg], scale, ulp, 0);
                acc = st.acc;
                if (st.done) { g = gg + 1; } else { pos = st.pos; adv = st.adv; }
            }
        } else {
            adv = 0;                                             // no integer image: the first group starts serially
        }
    }
#pragma unroll
    for (int gg = 0; gg < G; ++gg) {
        if (gg < g) continue;
        const int n_g = n - gg * kFoldGroupEntries;
        if (n_g <= 0) break;
        acc = fold_group_loop(acc, p[gg], min(n_g, kFoldGroupEntries), gg == g ? pos : 0, gg == g ? adv : kFoldGroupEntries);
    }
    return acc;
}

// acc <- (...((acc + e_0) + e_1) ... + e_255) in float32, e_{4L+k} = p<k> of lane L.  All 64 lanes must be active;
// `acc` is uniform and so is the result.  Entries past the end of the data must be +0.0 (a sum that started at +0
// cannot become -0, so they never change it).  `n` (uniform, <= 256): entries that may be non-zero.
__device__ __forceinline__ float fold256_spec(float acc, float p0, float p1, float p2, float p3, int n = kFoldGroupEntries) {
    const float p[1][4] = {{p0, p1, p2, p3}};
    return fold_groups_spec<1>(acc, p, n);
}

}  // namespace rtrec
