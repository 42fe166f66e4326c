/*
 * oracle/fold_model.c -- CPU model of the device's binade-speculative ordered fold (TEST INFRASTRUCTURE).
 *
 * What it checks.  The reference accumulates a dot product strictly left to right in float32
 * (scikit-learn 1.7.2 sklearn/linear_model/_cd_fast.pyx:464-466 `tmp += R[X_indices[jj]] * X_data[jj]`;
 * scipy csr_matvec likewise).  The HIP kernels (rtrec_amd/csrc/fold_spec.hip.h) reproduce that sum without
 * the chain of dependent float additions:
 *
 *   while the running sum stays inside one binade, acc = +-M u with u = 2^(e-23) and M an integer in
 *   [2^23, 2^24), and  RN(acc + p) = +-u RNI(M + p/u) = +-u (M + RNI(p/u))  unless p/u lies exactly half
 *   way between two integers -- so a run of sequentially rounded float additions is an INTEGER prefix sum
 *   of q_i = RNI(p_i / u), which is associative.  The first entry whose running value leaves (2^23, 2^24),
 *   or whose quotient is an exact tie (its rounding depends on the parity of the prefix), ends the run: it
 *   is added with one real float addition and the run restarts in the new binade.
 *
 * fold_model_group() below is the same control flow as the device function, entry for entry (256-entry
 * groups, the same pass / serial-stretch policy, wrapping 32-bit integer prefixes, the saturating float->int
 * conversion of v_cvt_i32_f32); fold_model_fuzz() compares it with the plain sequential loop on generated
 * sums.  Nothing here is part of the product path.  Build: see oracle/Makefile (-ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* v_cvt_i32_f32 of an integer-valued (or non-finite) float: saturates, NaN -> 0 */
static inline int32_t cvt_i32_sat(float q)
{
    if (q != q) return 0;
    if (q >= 2147483648.0f) return INT32_MAX;
    if (q <= -2147483648.0f) return INT32_MIN;
    return (int32_t)q;
}

long long g_dbg[8]; int g_dbg_on = 0;
typedef struct {
    int64_t entries;        /* entries folded */
    int64_t spec_entries;   /* entries absorbed by integer prefix sums */
    int64_t serial_entries; /* entries added with a real float addition */
    int64_t passes;         /* speculative passes */
} fold_model_stats;

enum { kGroup = 256 };
int kMinAdvance = 32, kSerialLead = 16;   /* policy constants (the device's kFoldMinAdvance / kFoldSerialLead); variables only so that tools can explore */
/* a serial stretch from pos runs to the end of the 64-entry row that holds entry pos + kSerialLead (the device adds whole
 * rows with its DPP chain) */
static inline int stretch_end(int pos, int n) { const int to = ((pos + kSerialLead) | 63) + 1; return to < n ? to : n; }
long long g_fail_passes = 0;

static float serial_add(float acc, const float *p, int from, int to, fold_model_stats *st)
{
    for (int i = from; i < to; i++) acc = acc + p[i];
    if (st) st->serial_entries += to - from;
    return acc;
}

/* One group of up to 256 products (the device pads a short group with +0.0 products; +0.0 never changes a
 * sum that started at +0, and the model simply stops at n).
 *
 * One speculative pass over entries [pos, n) from acc = +-M0 u:
 *   s_i = p_i / (+-u) (exact), r_i = rndne(s_i), d_i = s_i - r_i, A_i = M0 + sum_{k<=i} r_k  (plain integer prefixes).
 *   Non-tie entries add r_i.  A tie (|d_i| == 1/2: the exact sum lies half way between two grid points) rounds to
 *   the EVEN grid point: M -> M + r_i if M is even (r_i is the even neighbour of s_i), M + r_i + 2 d_i if M is odd;
 *   either way the value after a tie is even.  So the true running value is M_i = A_i + C_i with C_i the sum of the
 *   tie corrections c_t = 2 d_t ((A_{t-1} + C_{t-1}) & 1) of the ties t <= i: a walk over the TIE entries only.
 *   Range: every running value must stay strictly inside (2^23, 2^24) (same binade, grid u).  |C_i| <= the number of
 *   ties <= 256, so the test is made on A_i with a margin of 256 on both sides (conservative: an entry that fails
 *   it is simply added with a real float addition).  Entries whose quotient is not finite fail it too.
 *   The first failing entry j ends the pass: acc = +-(A_{j-1} + C_{j-1}) u, then acc += p_j in float. */
float fold_model_group(float acc, const float *p, int n, fold_model_stats *st)
{
    int pos = 0;
    while (pos < n) {
        const uint32_t bits = f2u(acc);
        const uint32_t ex = (bits >> 23) & 255u;
        if (ex < 24u || ex == 255u) {           /* zero, subnormal, tiny, inf, nan: no integer image */
            const int to = stretch_end(pos, n);
            acc = serial_add(acc, p, pos, to, st);
            pos = to;
            continue;
        }
        const uint32_t sign = bits & 0x80000000u;
        const float scale = u2f(((277u - ex) << 23) | sign);      /* +-2^(150-ex) = +-1/u */
        const float ulp = u2f(((ex - 23u) << 23) | sign);          /* +-u */
        const uint32_t M0 = (bits & 0x7fffffu) | 0x800000u;
        uint32_t A = M0;            /* plain prefix (wrapping, like the device's integer scan) */
        int32_t C = 0;              /* tie corrections so far */
        int j = -1;
        if (st) st->passes++;
        for (int i = pos; i < n; i++) {
            const float s = p[i] * scale;                          /* exact (power of two) unless it overflows */
            const float r = rintf(s);                              /* v_rndne_f32 */
            const float d = s - r;
            const int odd = !(fabsf(d) < 0.5f);                    /* exact tie, or inf / nan */
            const int tie = odd && (fabsf(d) == 0.5f);
            const uint32_t An = A + (uint32_t)cvt_i32_sat(r);
            const int ok = (uint32_t)(An - 0x800101u) < 0x7ffdffu; /* 2^23 + 256 < An < 2^24 - 256 */
            if (!ok || (odd && !tie)) { j = i; if (g_dbg_on) { g_dbg[An - 0x800101u >= 0x80000000u ? 0 : (An <= 0x800101u ? 1 : 2)]++; int a = i - pos; g_dbg[3 + (a < 4 ? 0 : a < 16 ? 1 : a < 64 ? 2 : 3)]++; g_dbg[7] += (i < 256 && pos == 0); } break; }
            if (tie && ((A + (uint32_t)C) & 1u)) C += d > 0.0f ? 1 : -1;
            A = An;
        }
        acc = (float)(int32_t)(A + (uint32_t)C) * ulp;            /* exact: 2^23 < A + C < 2^24 */
        if (j < 0) { if (st) st->spec_entries += n - pos; break; }
        if (st) { st->spec_entries += j - pos; st->serial_entries += 1; }
        acc = acc + p[j];                                          /* the one real addition */
        g_fail_passes++;
        const int adv = j - pos;
        pos = j + 1;
        if ((adv < kMinAdvance || n - pos <= 24) && pos < n) {
            const int to = stretch_end(pos, n);
            acc = serial_add(acc, p, pos, to, st);
            pos = to;
        }
    }
    if (st) st->entries += n;
    return acc;
}

long long g_pos_hist[8]; long long g_fold_count;
float fold_model_fold(float acc, const float *p, int64_t n, fold_model_stats *st)
{
    g_fold_count++;
    for (int64_t o = 0; o < n; o += kGroup) {
        const long long f0 = g_fail_passes;
        const int m = (int)(n - o < kGroup ? n - o : kGroup);
        acc = fold_model_group(acc, p + o, m, st);
        g_pos_hist[o < 256 ? 0 : o < 1024 ? 1 : o < 4096 ? 2 : o < 16384 ? 3 : 4] += g_fail_passes - f0;
    }
    return acc;
}

float fold_model_sequential(float acc, const float *p, int64_t n)
{
    for (int64_t i = 0; i < n; i++) acc = acc + p[i];
    return acc;
}

/* ---- generated sums ------------------------------------------------------------------------- */
static inline uint32_t xs32(uint32_t *s)
{
    uint32_t x = *s;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    return *s = x;
}
static inline float unif(uint32_t *s) { return (float)(xs32(s) >> 8) * (1.0f / 16777216.0f); }

/* kind: 0 positive drift (ratings x residual-like), 1 zero mean, 2 integers, 3 half-integer tie stress on a
 * large running sum, 4 wide dynamic range (random exponents, both signs), 5 cancellation (pairs +x, -x and
 * near pairs), 6 multiples of the running sum's half ulp (ties on most entries), 7 rare inf / huge values. */
static void gen(float *p, int n, int kind, uint32_t *s)
{
    switch (kind) {
    case 0: { const float a = 0.01f + 5.0f * unif(s); for (int i = 0; i < n; i++) p[i] = a * unif(s) * (1.0f + (float)(xs32(s) % 5u)); break; }
    case 1: { const float a = 0.01f + 5.0f * unif(s); for (int i = 0; i < n; i++) p[i] = a * (unif(s) - 0.5f) * (1.0f + (float)(xs32(s) % 5u)); break; }
    case 2: for (int i = 0; i < n; i++) p[i] = (float)((int)(xs32(s) % 51u) - ((xs32(s) & 3u) ? 0 : 25)); break;
    case 3: { p[0] = 8388608.0f * (1.0f + unif(s)) * (float)(1u << (xs32(s) % 6u));
              for (int i = 1; i < n; i++) p[i] = 0.5f * (float)((int)(xs32(s) % 9u) - 2) * (float)(1u << (xs32(s) % 4u));
              break; }
    case 4: for (int i = 0; i < n; i++) { const uint32_t e = 100u + xs32(s) % 60u; p[i] = u2f((xs32(s) & 0x807fffffu) | (e << 23)); } break;
    case 5: for (int i = 0; i + 1 < n; i += 2) { const float x = 10.0f * unif(s); p[i] = x; p[i + 1] = (xs32(s) & 1u) ? -x : -x * (1.0f + 1e-6f * unif(s)); }
            if (n & 1) p[n - 1] = unif(s);
            break;
    case 6: { float run = 1000.0f * (1.0f + unif(s)); p[0] = run;
              for (int i = 1; i < n; i++) { const uint32_t ex = (f2u(run) >> 23) & 255u; const float hu = u2f((ex - 24u) << 23);
                  p[i] = hu * (float)((int)(xs32(s) % 7u) - 2); run = run + p[i]; } break; }
    default: for (int i = 0; i < n; i++) { p[i] = unif(s) * 1e30f; if ((xs32(s) & 1023u) == 0u) p[i] = (xs32(s) & 1u) ? INFINITY : 3e38f; } break;
    }
}

/* Runs n_folds generated sums of up to max_len entries; returns the number whose speculative fold differs from
 * the sequential float loop in any bit (NaN results compare equal to NaN results).  stats accumulates. */
int64_t fold_model_fuzz(uint32_t seed, int64_t n_folds, int32_t max_len, int32_t kind, fold_model_stats *st)
{
    uint32_t s = seed ? seed : 1u;
    float *p = (float *)malloc(sizeof(float) * (size_t)(max_len > 0 ? max_len : 1));
    int64_t bad = 0;
    for (int64_t f = 0; f < n_folds; f++) {
        const int n = 1 + (int)(xs32(&s) % (uint32_t)max_len);
        const int k = kind >= 0 ? kind : (int)(xs32(&s) % 8u);
        gen(p, n, k, &s);
        float acc0 = 0.0f;
        if ((xs32(&s) & 7u) == 0u) acc0 = (unif(&s) - 0.3f) * 1000.0f;     /* a fold continuing an earlier one */
        const float a = fold_model_fold(acc0, p, n, st);
        const float b = fold_model_sequential(acc0, p, n);
        if (f2u(a) != f2u(b) && !(a != a && b != b)) bad++;
    }
    free(p);
    return bad;
}
