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
 * group_finish / group_loop / fold_model_window below are the device's fold_group_finish / fold_group_loop /
 * fold_groups_spec<G>, entry for entry (256-entry groups, windows of G groups, the same pass / serial-stretch policy,
 * wrapping 32-bit integer prefixes, the saturating float->int conversion of v_cvt_i32_f32); fold_model_fuzz() compares
 * them with the plain sequential loop on generated sums.  Nothing here is part of the product path.  Build: see oracle/Makefile (-ffp-contract=off).
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

typedef struct {
    int64_t entries;        /* entries folded */
    int64_t spec_entries;   /* entries absorbed by integer prefix sums */
    int64_t serial_entries; /* entries added with a real float addition */
    int64_t passes;         /* speculative passes (quantisations of a group) */
} fold_model_stats;

enum { kGroup = 256 };
int kMinAdvance = 32, kSerialLead = 16;   /* the device's kFoldMinAdvance / kFoldSerialLead (variables so that tools can explore) */
/* a serial stretch from pos runs to the end of the 64-entry row that holds entry pos + kSerialLead (the device adds whole
 * rows with its DPP chain) */
static inline int stretch_end(int pos, int n) { const int to = ((pos + kSerialLead) | 63) + 1; return to < n ? to : n; }

static float serial_add(float acc, const float *p, int from, int to, fold_model_stats *st)
{
    for (int i = from; i < to; i++) acc = acc + p[i];
    if (st) st->serial_entries += to - from;
    return acc;
}

/* fold_group_finish of the device: one group (entries [pos, n) of p; the device pads to 256 with +0.0) against the
 * running value M u.  Returns 1 when the group is absorbed (*acc = the new sum); otherwise *acc is the sum through the
 * first failing entry j (added with a real float addition), *pos = j + 1, *adv = entries absorbed before j.
 *   s_i = p_i / (+-u) (exact), r_i = rndne(s_i), d_i = s_i - r_i, A_i = M + sum_{k<=i} r_k  (plain, wrapping prefixes).
 *   A tie (|d_i| == 1/2) rounds to the EVEN grid point: the running value takes r_i when it is even, r_i + 2 d_i when it is
 *   odd; with C_i the sum of those corrections the true running value is A_i + C_i.  Range: A_i strictly inside
 *   (2^23 + 256, 2^24 - 256) -- |C_i| <= 256 -- ; an entry that fails it, or whose quotient is not finite, ends the pass. */
static int group_finish(float *acc, uint32_t bits, const float *p, int n, int *pos, int *adv, fold_model_stats *st)
{
    const uint32_t ex = (bits >> 23) & 255u, sign = bits & 0x80000000u;
    const float scale = u2f(((277u - ex) << 23) | sign);      /* +-2^(150-ex) = +-1/u */
    const float ulp = u2f(((ex - 23u) << 23) | sign);          /* +-u */
    uint32_t A = (bits & 0x7fffffu) | 0x800000u;
    int32_t C = 0;
    int j = -1;
    if (st) st->passes++;
    for (int i = *pos; i < n; i++) {
        const float s = p[i] * scale;                          /* exact (power of two) unless it overflows */
        const float r = rintf(s);                              /* v_rndne_f32 */
        const float d = s - r;
        const int odd = !(fabsf(d) < 0.5f);                    /* exact tie, or inf / nan */
        const int tie = odd && (fabsf(d) == 0.5f);
        const uint32_t An = A + (uint32_t)cvt_i32_sat(r);
        const int ok = (uint32_t)(An - 0x800101u) < 0x7ffdffu; /* 2^23 + 256 < An < 2^24 - 256 */
        if (!ok || (odd && !tie)) { j = i; break; }
        if (tie && ((A + (uint32_t)C) & 1u)) C += d > 0.0f ? 1 : -1;
        A = An;
    }
    *acc = (float)(int32_t)(A + (uint32_t)C) * ulp;           /* exact: 2^23 < A + C < 2^24 */
    if (j < 0) { if (st) st->spec_entries += n - *pos; return 1; }
    if (st) { st->spec_entries += j - *pos; st->serial_entries += 1; }
    *acc = *acc + p[j];                                        /* the one real addition */
    *adv = j - *pos;
    *pos = j + 1;
    return 0;
}

/* fold_group_loop of the device: one group from entry pos on, a speculative pass per turn */
static float group_loop(float acc, const float *p, int n, int pos, int adv, fold_model_stats *st)
{
    for (;;) {
        if ((adv < kMinAdvance || n - pos <= 24) && pos < n) {
            const int to = stretch_end(pos, n);
            acc = serial_add(acc, p, pos, to, st);
            pos = to;
        }
        if (pos >= n) return acc;
        const uint32_t bits = f2u(acc);
        const uint32_t ex = (bits >> 23) & 255u;
        if (ex < 24u || ex == 255u) { adv = 0; continue; }      /* zero, subnormal, tiny, inf, nan: no integer image */
        if (group_finish(&acc, bits, p, n, &pos, &adv, st)) return acc;
    }
}

/* fold_groups_spec<G> of the device: a window of up to G groups.  While the groups are clean in the binade of the sum the
 * device keeps the running value an integer from group to group; the first group that is not is finished from its
 * quantisation at hand, the groups after it go through the loop from their start. */
float fold_model_window(float acc, const float *p, int n, int G, fold_model_stats *st)
{
    int g = 0, pos = 0, adv = kGroup;
    const uint32_t bits0 = f2u(acc);
    const uint32_t ex0 = (bits0 >> 23) & 255u;
    if (ex0 >= 24u && ex0 != 255u) {
        for (; g < G && g * kGroup < n; g++) {
            const int n_g = n - g * kGroup < kGroup ? n - g * kGroup : kGroup;
            int ps = 0, ad = kGroup;
            const uint32_t bits = f2u(acc);                      /* same binade as bits0 while the groups are clean */
            if (((bits >> 23) & 255u) != ex0 || ((bits ^ bits0) & 0x80000000u)) break;   /* (cannot happen after a clean group) */
            if (!group_finish(&acc, bits, p + g * kGroup, n_g, &ps, &ad, st)) { pos = ps; adv = ad; break; }
        }
    } else {
        adv = 0;
    }
    for (; g < G && g * kGroup < n; g++) {
        const int n_g = n - g * kGroup < kGroup ? n - g * kGroup : kGroup;
        acc = group_loop(acc, p + g * kGroup, n_g, pos, adv, st);
        pos = 0; adv = kGroup;
    }
    if (st) st->entries += n;
    return acc;
}

int fold_model_groups_per_window = 4;     /* the multi-wave kernel's consumer; the single-wave kernel and rtrec_slim_ordered_sums mode 0 use 1 */

float fold_model_fold(float acc, const float *p, int64_t n, fold_model_stats *st)
{
    const int G = fold_model_groups_per_window, W = G * kGroup;
    for (int64_t o = 0; o < n; o += W) {
        const int m = (int)(n - o < W ? n - o : W);
        acc = fold_model_window(acc, p + o, m, G, st);
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
