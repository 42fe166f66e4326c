// rtrec_amd/csrc/store_host.hip -- HOST-side helper of the interaction store (no device code).
//
// Replaces, for the columnar store of rtrec_amd/utils/interactions.py, what the reference does with one
// Python dict update per interaction (/root/reference/rtrec/utils/interactions.py:81-119): the store keeps
// sorted (key, value, timestamp) blocks (key = user << 32 | item) and every write merges a sorted block
// into a larger one.  In numpy that merge is a binary search plus masked scatters at ~50 ns per element
// and array; here it is a two-pointer merge at memory speed, split over threads by key range.
#include "../../include/rtrec_amd.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

namespace {

struct Cols {
    const int64_t *k; const double *v; const double *t;
};

// Merge a[a0, a1) and b[b0, b1) (sorted, each with distinct keys); on equal keys the entry of b wins.
// WRITE = false only counts the output.
template <bool WRITE>
int64_t merge_range(const Cols &a, int64_t a0, int64_t a1, const Cols &b, int64_t b0, int64_t b1,
                    int64_t *ko, double *vo, double *to) {
    int64_t i = a0, j = b0, o = 0;
    while (i < a1 && j < b1) {
        const int64_t ka = a.k[i], kb = b.k[j];
        if (ka < kb) {
            if (WRITE) { ko[o] = ka; vo[o] = a.v[i]; to[o] = a.t[i]; }
            ++i;
        } else {
            if (WRITE) { ko[o] = kb; vo[o] = b.v[j]; to[o] = b.t[j]; }
            ++j;
            i += (ka == kb);
        }
        ++o;
    }
    if (WRITE) {
        for (; i < a1; ++i, ++o) { ko[o] = a.k[i]; vo[o] = a.v[i]; to[o] = a.t[i]; }
        for (; j < b1; ++j, ++o) { ko[o] = b.k[j]; vo[o] = b.v[j]; to[o] = b.t[j]; }
    } else {
        o += (a1 - i) + (b1 - j);
    }
    return o;
}

}  // namespace

extern "C" int64_t rtrec_store_merge_sorted(const int64_t *a_key, const double *a_val, const double *a_ts, int64_t n_a,
                                            const int64_t *b_key, const double *b_val, const double *b_ts, int64_t n_b,
                                            int64_t *out_key, double *out_val, double *out_ts, int32_t n_threads) {
    if (n_a < 0 || n_b < 0 || (n_a > 0 && !a_key) || (n_b > 0 && !b_key)) return -1;
    const Cols a{a_key, a_val, a_ts}, b{b_key, b_val, b_ts};
    int T = n_threads > 0 ? n_threads : static_cast<int>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
    if (n_a + n_b < (1 << 16)) T = 1;
    if (T == 1) return merge_range<true>(a, 0, n_a, b, 0, n_b, out_key, out_val, out_ts);
    // partition by key: thread p takes a[sa[p], sa[p+1]) and the entries of b below a's next split key
    std::vector<int64_t> sa(T + 1), sb(T + 1), cnt(T + 1, 0);
    for (int p = 0; p <= T; ++p) sa[p] = n_a * p / T;
    sb[0] = 0; sb[T] = n_b;
    for (int p = 1; p < T; ++p) sb[p] = std::lower_bound(b_key, b_key + n_b, a_key[sa[p]]) - b_key;
    std::vector<std::thread> th;
    for (int p = 0; p < T; ++p)
        th.emplace_back([&, p] { cnt[p + 1] = merge_range<false>(a, sa[p], sa[p + 1], b, sb[p], sb[p + 1], nullptr, nullptr, nullptr); });
    for (auto &x : th) x.join();
    for (int p = 0; p < T; ++p) cnt[p + 1] += cnt[p];
    th.clear();
    for (int p = 0; p < T; ++p)
        th.emplace_back([&, p] {
            merge_range<true>(a, sa[p], sa[p + 1], b, sb[p], sb[p + 1], out_key + cnt[p], out_val + cnt[p], out_ts + cnt[p]);
        });
    for (auto &x : th) x.join();
    return cnt[T];
}

// Positions of ascending `needles` in the sorted `hay`: pos[i] = lower bound of needles[i] (clamped to
// n - 1), found[i] = 1 when hay[pos[i]] == needles[i].  Galloping search from the previous hit: O(m log(n/m))
// and sequential in memory, where numpy's searchsorted pays a cache-missing binary search per needle.
extern "C" int rtrec_store_find_sorted(const int64_t *hay, int64_t n, const int64_t *needles, int64_t m,
                                       int64_t *pos, uint8_t *found, int32_t n_threads) {
    if (n < 0 || m < 0 || (n > 0 && !hay) || (m > 0 && (!needles || !pos || !found))) return -1;
    if (n == 0) {
        for (int64_t i = 0; i < m; ++i) { pos[i] = 0; found[i] = 0; }
        return 0;
    }
    int T = n_threads > 0 ? n_threads : static_cast<int>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
    if (m < (1 << 14)) T = 1;
    auto work = [&](int64_t i0, int64_t i1) {
        int64_t lo = std::lower_bound(hay, hay + n, needles[i0]) - hay;
        for (int64_t i = i0; i < i1; ++i) {
            const int64_t key = needles[i];
            int64_t step = 1, hi = lo;                   // gallop: hay[lo - 1] < key by the ascending order
            while (hi < n && hay[hi] < key) { lo = hi + 1; hi += step; step <<= 1; }
            lo = std::lower_bound(hay + lo, hay + std::min(hi, n), key) - hay;
            const int64_t p = std::min(lo, n - 1);
            pos[i] = p;
            found[i] = hay[p] == key;
        }
    };
    if (T == 1) { if (m) work(0, m); return 0; }
    std::vector<std::thread> th;
    for (int p = 0; p < T; ++p) {
        const int64_t i0 = m * p / T, i1 = m * (p + 1) / T;
        if (i1 > i0) th.emplace_back(work, i0, i1);
    }
    for (auto &x : th) x.join();
    return 0;
}

// Hot-item bookkeeping (rtrec/utils/lru.py:11-60, LRUFreqSet.add): replay `values` in order on a
// capacity-bounded recency list with hit counts -- an existing key is bumped and becomes most recent, a new
// key first evicts the least recent one when the list is full (its count restarts at 1 if it returns).
// The list comes in and goes out oldest-first as (key, count) arrays; keys are ids in [0, id_bound).
// An intrusive doubly linked list over the ids: ~10 ns per value where the Python loop needs ~1.5 us.
extern "C" int64_t rtrec_lru_replay(const int64_t *state_keys, const int64_t *state_counts, int64_t n_state,
                                    const int64_t *values, int64_t n, int64_t capacity, int64_t id_bound,
                                    int64_t *out_keys, int64_t *out_counts) {
    if (capacity <= 0 || id_bound <= 0 || n_state < 0 || n < 0 || n_state > capacity) return -1;
    std::vector<int32_t> prev(id_bound, -1), next(id_bound, -1);
    std::vector<int64_t> cnt(id_bound, 0);
    int64_t head = -1, tail = -1, size = 0;
    auto append = [&](int64_t v) {
        prev[v] = static_cast<int32_t>(tail); next[v] = -1;
        if (tail >= 0) next[tail] = static_cast<int32_t>(v); else head = v;
        tail = v;
    };
    auto unlink = [&](int64_t v) {
        const int64_t p = prev[v], q = next[v];
        if (p >= 0) next[p] = static_cast<int32_t>(q); else head = q;
        if (q >= 0) prev[q] = static_cast<int32_t>(p); else tail = p;
    };
    for (int64_t i = 0; i < n_state; ++i) {
        const int64_t k = state_keys[i];
        if (k < 0 || k >= id_bound || cnt[k] != 0 || state_counts[i] <= 0) return -1;
        cnt[k] = state_counts[i];
        append(k);
        ++size;
    }
    for (int64_t i = 0; i < n; ++i) {
        const int64_t v = values[i];
        if (v < 0 || v >= id_bound) return -1;
        if (cnt[v] > 0) {
            unlink(v);
            ++cnt[v];
        } else {
            if (size >= capacity) { const int64_t h = head; unlink(h); cnt[h] = 0; --size; }
            cnt[v] = 1;
            ++size;
        }
        append(v);
    }
    int64_t o = 0;
    for (int64_t k = head; k >= 0; k = next[k], ++o) { out_keys[o] = k; out_counts[o] = cnt[k]; }
    return o;
}

// One round of a batch of interactions on DISTINCT pairs without time decay (interactions.py:81-119):
// interaction order[k] lands on the k-th key of the round; its value is clip(old[k] + delta, lo, hi) -- or
// delta itself when `old` is NULL (upsert) -- and its timestamp the interaction's.  Two random gathers
// and a clip, spread over threads.
extern "C" int rtrec_store_apply_round(const int64_t *order, int64_t n, const double *delta, const double *tstamp,
                                       const double *old, double lo, double hi, double *out_val, double *out_ts,
                                       int32_t n_threads) {
    if (n < 0 || (n > 0 && (!order || !delta || !tstamp || !out_val || !out_ts))) return -1;
    int T = n_threads > 0 ? n_threads : static_cast<int>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
    if (n < (1 << 16)) T = 1;
    auto work = [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) {
            const int64_t i = order[k];
            double v = delta[i];
            if (old) { v += old[k]; v = hi < v ? hi : v; v = v > lo ? v : lo; }      // Python's max(lo, min(v, hi)): a NaN ends as lo
            out_val[k] = v;
            out_ts[k] = tstamp[i];
        }
    };
    if (T == 1) { work(0, n); return 0; }
    std::vector<std::thread> th;
    for (int p = 0; p < T; ++p) {
        const int64_t k0 = n * p / T, k1 = n * (p + 1) / T;
        if (k1 > k0) th.emplace_back(work, k0, k1);
    }
    for (auto &x : th) x.join();
    return 0;
}

// Time decay of stored values (rtrec/utils/interactions.py:62-79): out[k] = val[k] * rate ** ((now - ts[k]) / 86400)
// in float64 with libm's pow() -- the function CPython's float ** float calls, so the result is the
// reference's bit for bit (numpy's vectorised pow differs by an ulp now and then) -- `now` per entry
// (now_arr) or one value.  out64 and/or out32 (the float32 the exports carry) may be NULL.
extern "C" int rtrec_store_decay(const double *val, const double *ts, int64_t n, double rate, const double *now_arr,
                                 double now, double *out64, float *out32, int32_t n_threads) {
    if (n < 0 || (n > 0 && (!val || !ts))) return -1;
    int T = n_threads > 0 ? n_threads : static_cast<int>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
    if (n < (1 << 14)) T = 1;
    auto work = [&](int64_t k0, int64_t k1) {
        for (int64_t k = k0; k < k1; ++k) {
            const double elapsed_days = ((now_arr ? now_arr[k] : now) - ts[k]) / 86400.0;
            const double v = val[k] * std::pow(rate, elapsed_days);
            if (out64) out64[k] = v;
            if (out32) out32[k] = static_cast<float>(v);
        }
    };
    if (T == 1) { work(0, n); return 0; }
    std::vector<std::thread> th;
    for (int p = 0; p < T; ++p) {
        const int64_t k0 = n * p / T, k1 = n * (p + 1) / T;
        if (k1 > k0) th.emplace_back(work, k0, k1);
    }
    for (auto &x : th) x.join();
    return 0;
}
