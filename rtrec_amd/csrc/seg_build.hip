// rtrec_amd/csrc/seg_build.hip -- the segment layout of W (rtrec_amd/seg_layout.py) built on the device in two calls.
//
// After every mini-batch (SLIM.fit, /root/reference/rtrec/models/slim.py:29-43) W has changed and the form the scoring
// kernels read (csrc/score_seg.hip.h) has to be rebuilt before the next recommend.  The tensor-op builder
// (seg_layout.build_seg_layout_device) is ~45 small launches and 5 host round trips: 1.7-1.8 ms for the ML-20M shapes,
// all of it launch and synchronisation latency (W is 360k entries -- 3 MB).  Here the same arrays come out of a dozen
// kernels plus rocPRIM sorts / scans (hipcub front end) with ONE round trip (plan: how many columns and rows, hence
// the tile width and the size of every output).
//
// Specification: seg_layout.build_seg_layout (numpy); tests/test_gpu_seg.py compares every array.
//   plan : flags of the shard's columns / rows -> n_cols, R; layout order of the columns = stable sort by cluster label
//   fill : entries keyed (row, layout column) and sorted; segment (row, tile) = [lower_bound(s T), lower_bound((s + 1) T));
//          records (sparse: {column in tile, weight bits}, padded to even with {T, 0}; 256-column tiles with more than 64
//          entries: 256 floats), begin pointers (bit 31 = dense), bfloat16 bounds rounded up, the tile-side segment lists.
#include "common.hip.h"
#include "../../include/rtrec_amd.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>

namespace rtrec {
namespace {

constexpr int kSbDenseMin = 64;        // = seg_layout.SG_DENSE_MIN
constexpr int kSbBlock = 256;

typedef unsigned long long u64;

inline size_t sb_align(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }
inline unsigned sb_grid(long long n) {
    const long long b = (n + kSbBlock - 1) / kSbBlock;
    return static_cast<unsigned>(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
#define SB_LOOP(i, n) \
    for (long long i = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; i < (n); i += static_cast<long long>(gridDim.x) * blockDim.x)

// ---------------------------------------------------------------------------------------------------------- plan
__global__ void sb_mark(const long long *__restrict__ rows, const long long *__restrict__ cols, long long nnz, int lo, int hi,
                        int *__restrict__ colflag, int *__restrict__ rowflag) {
    SB_LOOP(e, nnz) {
        const long long c = cols[e];
        if (c >= lo && c < hi) { colflag[c] = 1; rowflag[rows[e]] = 1; }
    }
}
// order key of item i: (cluster label, item) for a column of the shard, beyond every label for the others
__global__ void sb_order_keys(const int *__restrict__ colflag, const long long *__restrict__ labels, int n_items, u64 *__restrict__ key) {
    SB_LOOP(i, n_items) {
        const u64 lab = colflag[i] ? static_cast<u64>(labels[i]) : static_cast<u64>(n_items);
        key[i] = (lab << 32) | static_cast<u64>(i);
    }
}
__global__ void sb_totals(const int *__restrict__ colflag, const int *__restrict__ cscan, const int *__restrict__ rowflag,
                          const int *__restrict__ rscan, int n_items, int *__restrict__ hdr) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr[0] = cscan[n_items - 1] + colflag[n_items - 1];
        hdr[1] = rscan[n_items - 1] + rowflag[n_items - 1];
    }
}

// ---------------------------------------------------------------------------------------------------------- fill
__global__ void sb_clear_pos(int n_items, int *__restrict__ pos) { SB_LOOP(i, n_items) pos[i] = -1; }
__global__ void sb_pos(const u64 *__restrict__ okey, int n_cols, int *__restrict__ pos, int *__restrict__ col_ids) {
    SB_LOOP(p, n_cols) {
        const int item = static_cast<int>(okey[p] & 0xffffffffull);
        col_ids[p] = item;
        pos[item] = static_cast<int>(p);
    }
}
__global__ void sb_info(const int *__restrict__ rowflag, const int *__restrict__ rscan, const int *__restrict__ pos, int n_items,
                        int2 *__restrict__ info, int *__restrict__ row_item) {
    SB_LOOP(i, n_items) {
        const int r = rowflag[i] ? rscan[i] : -1;
        info[i] = make_int2(r, pos[i]);
        if (r >= 0) row_item[r] = static_cast<int>(i);
    }
}
__global__ void sb_entry_keys(const long long *__restrict__ rows, const long long *__restrict__ cols, long long nnz, int lo, int hi,
                              const int2 *__restrict__ info, u64 row_span, u64 key_end, u64 *__restrict__ key, int *__restrict__ idx) {
    SB_LOOP(e, nnz) {
        const long long c = cols[e];
        u64 k = key_end;                              // entries of other shards sort behind every segment
        if (c >= lo && c < hi) k = static_cast<u64>(info[rows[e]].x) * row_span + static_cast<u64>(info[c].y);
        key[e] = k;
        idx[e] = static_cast<int>(e);
    }
}
// seg_begin[s] = first sorted entry with key >= s * T, s = row * n_tiles + tile (0 .. n_seg)
__global__ void sb_seg_begin(const u64 *__restrict__ key, int nnz, long long n_seg, int T, int *__restrict__ seg_begin) {
    SB_LOOP(s, n_seg + 1) {
        const u64 want = static_cast<u64>(s) * static_cast<u64>(T);
        int a = 0, b = nnz;
        while (a < b) {
            const int m = (a + b) >> 1;
            if (key[m] < want) a = m + 1; else b = m;
        }
        seg_begin[s] = a;
    }
}
__global__ void sb_alloc(const int *__restrict__ seg_begin, long long n_seg, int R, int n_tiles, int T, int *__restrict__ alloc,
                         int *__restrict__ flag_t) {
    SB_LOOP(s, n_seg + 1) {
        int a = 0;
        if (s < n_seg) {
            const int len = seg_begin[s + 1] - seg_begin[s];
            a = (T == 256 && len > kSbDenseMin) ? T / 2 : len + (len & 1);
            const long long r = s / n_tiles, t = s % n_tiles;
            flag_t[t * R + r] = len > 0 ? 1 : 0;
        } else {
            flag_t[n_seg] = 0;
        }
        alloc[s] = a;
    }
}
__global__ void sb_init_ent(long long n_rec, int T, int2 *__restrict__ ent) { SB_LOOP(i, n_rec) ent[i] = make_int2(T, 0); }
// dense blocks are all weights: clear their pad markers (one wave per 64 segments, a dense one zeroed by all lanes)
__global__ void sb_dense_zero(const int *__restrict__ seg_begin, const int *__restrict__ start, long long n_seg, int T, int *__restrict__ ent_words) {
    const int lane = threadIdx.x & 63;
    const long long wave = (blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x) >> 6;
    const long long n_waves = (static_cast<long long>(gridDim.x) * blockDim.x) >> 6;
    for (long long base = wave * 64; base < n_seg; base += n_waves * 64) {
        const long long s = base + lane;
        const bool dense = s < n_seg && T == 256 && (seg_begin[s + 1] - seg_begin[s]) > kSbDenseMin;
        unsigned long long m = __ballot(dense);
        while (m) {
            const int q = __builtin_ctzll(m);
            m &= m - 1;
            const long long w0 = static_cast<long long>(start[base + q]) * 2;
            for (int c = lane; c < 256; c += 64) ent_words[w0 + c] = 0;
        }
    }
}
__global__ void sb_scatter(const u64 *__restrict__ key, const int *__restrict__ idx, int nnz, u64 key_end, int T,
                           const int *__restrict__ seg_begin, const int *__restrict__ start, const float *__restrict__ vals,
                           int *__restrict__ ent_words) {
    SB_LOOP(i, nnz) {
        const u64 k = key[i];
        if (k >= key_end) continue;
        const long long s = static_cast<long long>(k / static_cast<u64>(T));
        const int col_in = static_cast<int>(k % static_cast<u64>(T));
        const int b = seg_begin[s];
        const bool dense = T == 256 && (seg_begin[s + 1] - b) > kSbDenseMin;
        const int vbits = __float_as_int(vals[idx[i]]);
        const long long w0 = static_cast<long long>(start[s]) * 2;
        if (dense) {
            ent_words[w0 + col_in] = vbits;
        } else {
            const long long w = w0 + 2ll * (static_cast<int>(i) - b);
            ent_words[w] = col_in;
            ent_words[w + 1] = vbits;
        }
    }
}
__global__ void sb_ptr(const int *__restrict__ seg_begin, const int *__restrict__ start, int R, int n_tiles, int T, int *__restrict__ seg_ptr) {
    const long long n = static_cast<long long>(R) * (n_tiles + 1);
    SB_LOOP(q, n) {
        const long long r = q / (n_tiles + 1), t = q % (n_tiles + 1);
        const long long s = r * n_tiles + t;                  // t == n_tiles: the next row's first segment = this row's end
        unsigned v = static_cast<unsigned>(start[s]);
        if (t < n_tiles && T == 256 && (seg_begin[s + 1] - seg_begin[s]) > kSbDenseMin) v |= 0x80000000u;
        seg_ptr[q] = static_cast<int>(v);
    }
}
// bound[r][l] = bf16-rounded-up max |w| of tiles 2 l (low half) and 2 l + 1 (high half)
__global__ void sb_bound(const int *__restrict__ seg_begin, const int *__restrict__ idx, const float *__restrict__ vals, int R, int n_tiles,
                         uint32_t *__restrict__ bound) {
    const long long n = static_cast<long long>(R) * 64;
    SB_LOOP(q, n) {
        const long long r = q >> 6;
        const int l = static_cast<int>(q & 63);
        uint32_t word = 0;
        for (int h = 0; h < 2; ++h) {
            const int t = 2 * l + h;
            if (t >= n_tiles) break;
            const long long s = r * n_tiles + t;
            float mx = 0.0f;
            for (int i = seg_begin[s]; i < seg_begin[s + 1]; ++i) mx = fmaxf(mx, fabsf(vals[idx[i]]));
            const uint32_t up = (__float_as_uint(mx) + 0xFFFFu) >> 16;
            word |= up << (16 * h);
        }
        bound[q] = word;
    }
}
__global__ void sb_trow(const int *__restrict__ seg_begin, const int *__restrict__ start, const int *__restrict__ seg_ptr,
                        const int *__restrict__ tpos, const int *__restrict__ row_item, int R, int n_tiles, int4 *__restrict__ trow,
                        int *__restrict__ trow_ptr) {
    const long long n_seg = static_cast<long long>(R) * n_tiles;
    SB_LOOP(s, n_seg + n_tiles + 1) {
        if (s >= n_seg) {                                      // the tile pointers
            const long long t = s - n_seg;
            trow_ptr[t] = tpos[t * R];
            continue;
        }
        if (seg_begin[s + 1] == seg_begin[s]) continue;
        const long long r = s / n_tiles, t = s % n_tiles;
        trow[tpos[t * R + r]] = make_int4(row_item[r], seg_ptr[r * (n_tiles + 1) + t], start[s + 1], 0);
    }
}

int sb_tile_cols(int n_cols) {          // = seg_layout.seg_tile_cols
    int T = 256;
    while ((std::max(n_cols, 1) + T - 1) / T > 128) T *= 2;
    return T <= 4096 ? T : 0;
}

struct PlanWs {
    int *colflag, *rowflag, *cscan, *rscan, *hdr;
    u64 *okey_in, *okey_out;
    void *cub; size_t cub_bytes, total;
};
PlanWs plan_ws(unsigned char *base, int n_items) {
    PlanWs w{};
    size_t scan_b = 0, sort_b = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_b, static_cast<int *>(nullptr), static_cast<int *>(nullptr), n_items);
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_b, static_cast<u64 *>(nullptr), static_cast<u64 *>(nullptr), n_items);
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *p = base ? base + off : nullptr; off += sb_align(bytes); return p; };
    // colflag and rowflag are adjacent: one memset clears both
    w.colflag = reinterpret_cast<int *>(take(static_cast<size_t>(n_items) * 4));
    w.rowflag = reinterpret_cast<int *>(take(static_cast<size_t>(n_items) * 4));
    w.cscan = reinterpret_cast<int *>(take(static_cast<size_t>(n_items) * 4));
    w.rscan = reinterpret_cast<int *>(take(static_cast<size_t>(n_items) * 4));
    w.hdr = reinterpret_cast<int *>(take(64));
    w.okey_in = reinterpret_cast<u64 *>(take(static_cast<size_t>(n_items) * 8));
    w.okey_out = reinterpret_cast<u64 *>(take(static_cast<size_t>(n_items) * 8));
    w.cub_bytes = std::max(scan_b, sort_b);
    w.cub = take(w.cub_bytes);
    w.total = off;
    return w;
}

struct FillWs {
    int *pos, *row_item, *eidx_in, *eidx_out, *seg_begin, *alloc, *start, *flag_t, *tpos;
    u64 *ekey_in, *ekey_out;
    void *cub; size_t cub_bytes, total;
};
FillWs fill_ws(unsigned char *base, int n_items, long long nnz, long long n_seg) {
    FillWs w{};
    size_t scan_b = 0, sort_b = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_b, static_cast<int *>(nullptr), static_cast<int *>(nullptr), static_cast<int>(n_seg + 1));
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_b, static_cast<u64 *>(nullptr), static_cast<u64 *>(nullptr), static_cast<int *>(nullptr),
                                       static_cast<int *>(nullptr), static_cast<int>(nnz));
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *p = base ? base + off : nullptr; off += sb_align(bytes); return p; };
    w.pos = reinterpret_cast<int *>(take(static_cast<size_t>(n_items) * 4));
    w.row_item = reinterpret_cast<int *>(take(static_cast<size_t>(n_items) * 4));
    w.ekey_in = reinterpret_cast<u64 *>(take(static_cast<size_t>(nnz) * 8));
    w.ekey_out = reinterpret_cast<u64 *>(take(static_cast<size_t>(nnz) * 8));
    w.eidx_in = reinterpret_cast<int *>(take(static_cast<size_t>(nnz) * 4));
    w.eidx_out = reinterpret_cast<int *>(take(static_cast<size_t>(nnz) * 4));
    w.seg_begin = reinterpret_cast<int *>(take(static_cast<size_t>(n_seg + 1) * 4));
    w.alloc = reinterpret_cast<int *>(take(static_cast<size_t>(n_seg + 1) * 4));
    w.start = reinterpret_cast<int *>(take(static_cast<size_t>(n_seg + 1) * 4));
    w.flag_t = reinterpret_cast<int *>(take(static_cast<size_t>(n_seg + 1) * 4));
    w.tpos = reinterpret_cast<int *>(take(static_cast<size_t>(n_seg + 1) * 4));
    w.cub_bytes = std::max(scan_b, sort_b);
    w.cub = take(w.cub_bytes);
    w.total = off;
    return w;
}

bool sb_sizes_ok(int n_items, long long nnz) { return n_items > 0 && nnz > 0 && nnz < (1ll << 27); }   // 2 nnz records < SG_MAX_RECORDS

}  // namespace
}  // namespace rtrec

using namespace rtrec;

extern "C" size_t rtrec_slim_seg_plan_workspace_bytes(int32_t n_items) {
    return n_items > 0 ? plan_ws(nullptr, n_items).total : 0;
}

extern "C" size_t rtrec_slim_seg_fill_workspace_bytes(int32_t n_items, int64_t nnz, int32_t n_rows, int32_t n_tiles) {
    if (!sb_sizes_ok(n_items, nnz) || n_rows <= 0 || n_tiles <= 0 || static_cast<long long>(n_rows) * n_tiles >= (1ll << 31) - 1) return 0;
    return fill_ws(nullptr, n_items, nnz, static_cast<long long>(n_rows) * n_tiles).total;
}

extern "C" int rtrec_slim_seg_plan(int32_t n_items, int64_t nnz, const int64_t *d_rows, const int64_t *d_cols,
                                   int32_t col_lo, int32_t col_hi, const int64_t *d_labels, void *d_workspace, size_t workspace_bytes,
                                   int32_t *h_out, void *stream) {
    if (!sb_sizes_ok(n_items, nnz) || !d_rows || !d_cols || !d_labels || !d_workspace || !h_out || col_lo < 0 || col_hi > n_items ||
        col_lo >= col_hi)
        return RTREC_ERR_INVALID_ARG;
    PlanWs w = plan_ws(static_cast<unsigned char *>(d_workspace), n_items);
    if (workspace_bytes < w.total) return RTREC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    const size_t flags_bytes = reinterpret_cast<unsigned char *>(w.cscan) - reinterpret_cast<unsigned char *>(w.colflag);
    if (hipMemsetAsync(w.colflag, 0, flags_bytes, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    hipLaunchKernelGGL(sb_mark, dim3(sb_grid(nnz)), dim3(kSbBlock), 0, st, reinterpret_cast<const long long *>(d_rows),
                       reinterpret_cast<const long long *>(d_cols), static_cast<long long>(nnz), col_lo, col_hi, w.colflag, w.rowflag);
    size_t cb = w.cub_bytes;
    (void)hipcub::DeviceScan::ExclusiveSum(w.cub, cb, w.colflag, w.cscan, n_items, st);
    cb = w.cub_bytes;
    (void)hipcub::DeviceScan::ExclusiveSum(w.cub, cb, w.rowflag, w.rscan, n_items, st);
    hipLaunchKernelGGL(sb_totals, dim3(1), dim3(64), 0, st, w.colflag, w.cscan, w.rowflag, w.rscan, n_items, w.hdr);
    hipLaunchKernelGGL(sb_order_keys, dim3(sb_grid(n_items)), dim3(kSbBlock), 0, st, w.colflag, reinterpret_cast<const long long *>(d_labels),
                       n_items, w.okey_in);
    int bits = 33;                                       // labels (and the sentinel n_items) need bits(n_items) above the item's 32
    while ((1ll << (bits - 32)) <= n_items) ++bits;
    cb = w.cub_bytes;
    (void)hipcub::DeviceRadixSort::SortKeys(w.cub, cb, w.okey_in, w.okey_out, n_items, 0, bits, st);
    int hdr[2] = {0, 0};
    if (hipMemcpyAsync(hdr, w.hdr, sizeof(hdr), hipMemcpyDeviceToHost, st) != hipSuccess) return RTREC_ERR_LAUNCH;
    if (hipStreamSynchronize(st) != hipSuccess) return RTREC_ERR_LAUNCH;
    const int T = sb_tile_cols(hdr[0]);
    h_out[0] = hdr[0]; h_out[1] = hdr[1]; h_out[2] = T; h_out[3] = T ? (hdr[0] + T - 1) / T : 0;
    return launch_status();
}

extern "C" int rtrec_slim_seg_fill(int32_t n_items, int64_t nnz, const int64_t *d_rows, const int64_t *d_cols, const float *d_vals,
                                   int32_t col_lo, int32_t col_hi, const void *d_plan_workspace, int32_t n_cols, int32_t n_rows,
                                   int32_t tile_cols, int32_t n_tiles, void *d_workspace, size_t workspace_bytes,
                                   int32_t *d_info, int32_t *d_seg_ptr, int32_t *d_ent, int64_t ent_capacity, uint32_t *d_bound,
                                   int32_t *d_col_ids, int32_t *d_trow_ptr, int32_t *d_trow, int64_t trow_capacity, void *stream) {
    if (!sb_sizes_ok(n_items, nnz) || !d_rows || !d_cols || !d_vals || !d_plan_workspace || !d_workspace || !d_info || !d_seg_ptr ||
        !d_ent || !d_bound || !d_col_ids || !d_trow_ptr || !d_trow)
        return RTREC_ERR_INVALID_ARG;
    const int T = tile_cols;
    const long long n_seg = static_cast<long long>(n_rows) * n_tiles;
    if (n_cols <= 0 || n_rows <= 0 || T < 256 || T > 4096 || (T & (T - 1)) || n_tiles != (n_cols + T - 1) / T || n_tiles > 128 ||
        n_seg >= (1ll << 31) - 1 || ent_capacity < 2 * nnz || trow_capacity < std::min<long long>(nnz, n_seg) ||
        (reinterpret_cast<uintptr_t>(d_info) & 7u) || (reinterpret_cast<uintptr_t>(d_ent) & 7u) || (reinterpret_cast<uintptr_t>(d_trow) & 15u))
        return RTREC_ERR_INVALID_ARG;
    const PlanWs p = plan_ws(const_cast<unsigned char *>(static_cast<const unsigned char *>(d_plan_workspace)), n_items);
    FillWs w = fill_ws(static_cast<unsigned char *>(d_workspace), n_items, nnz, n_seg);
    if (workspace_bytes < w.total) return RTREC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipGetLastError();
    const int m = static_cast<int>(nnz);
    const u64 row_span = static_cast<u64>(n_tiles) * static_cast<u64>(T);
    const u64 key_end = static_cast<u64>(n_rows) * row_span;
    int2 *info = reinterpret_cast<int2 *>(d_info);
    int *ent_words = d_ent;

    hipLaunchKernelGGL(sb_clear_pos, dim3(sb_grid(n_items)), dim3(kSbBlock), 0, st, n_items, w.pos);
    hipLaunchKernelGGL(sb_pos, dim3(sb_grid(n_cols)), dim3(kSbBlock), 0, st, p.okey_out, n_cols, w.pos, d_col_ids);
    hipLaunchKernelGGL(sb_info, dim3(sb_grid(n_items)), dim3(kSbBlock), 0, st, p.rowflag, p.rscan, w.pos, n_items, info, w.row_item);
    hipLaunchKernelGGL(sb_entry_keys, dim3(sb_grid(nnz)), dim3(kSbBlock), 0, st, reinterpret_cast<const long long *>(d_rows),
                       reinterpret_cast<const long long *>(d_cols), static_cast<long long>(nnz), col_lo, col_hi, info, row_span, key_end,
                       w.ekey_in, w.eidx_in);
    int bits = 1;
    while (bits < 64 && (key_end >> bits) != 0) ++bits;
    size_t cb = w.cub_bytes;
    (void)hipcub::DeviceRadixSort::SortPairs(w.cub, cb, w.ekey_in, w.ekey_out, w.eidx_in, w.eidx_out, m, 0, bits, st);
    hipLaunchKernelGGL(sb_seg_begin, dim3(sb_grid(n_seg + 1)), dim3(kSbBlock), 0, st, w.ekey_out, m, n_seg, T, w.seg_begin);
    hipLaunchKernelGGL(sb_alloc, dim3(sb_grid(n_seg + 1)), dim3(kSbBlock), 0, st, w.seg_begin, n_seg, n_rows, n_tiles, T, w.alloc, w.flag_t);
    cb = w.cub_bytes;
    (void)hipcub::DeviceScan::ExclusiveSum(w.cub, cb, w.alloc, w.start, static_cast<int>(n_seg + 1), st);
    cb = w.cub_bytes;
    (void)hipcub::DeviceScan::ExclusiveSum(w.cub, cb, w.flag_t, w.tpos, static_cast<int>(n_seg + 1), st);
    hipLaunchKernelGGL(sb_init_ent, dim3(sb_grid(ent_capacity)), dim3(kSbBlock), 0, st, static_cast<long long>(ent_capacity), T,
                       reinterpret_cast<int2 *>(d_ent));
    if (T == 256)
        hipLaunchKernelGGL(sb_dense_zero, dim3(sb_grid(n_seg)), dim3(kSbBlock), 0, st, w.seg_begin, w.start, n_seg, T, ent_words);
    hipLaunchKernelGGL(sb_scatter, dim3(sb_grid(nnz)), dim3(kSbBlock), 0, st, w.ekey_out, w.eidx_out, m, key_end, T, w.seg_begin, w.start,
                       d_vals, ent_words);
    hipLaunchKernelGGL(sb_ptr, dim3(sb_grid(static_cast<long long>(n_rows) * (n_tiles + 1))), dim3(kSbBlock), 0, st, w.seg_begin, w.start,
                       n_rows, n_tiles, T, d_seg_ptr);
    hipLaunchKernelGGL(sb_bound, dim3(sb_grid(static_cast<long long>(n_rows) * 64)), dim3(kSbBlock), 0, st, w.seg_begin, w.eidx_out, d_vals,
                       n_rows, n_tiles, d_bound);
    hipLaunchKernelGGL(sb_trow, dim3(sb_grid(n_seg + n_tiles + 1)), dim3(kSbBlock), 0, st, w.seg_begin, w.start, d_seg_ptr, w.tpos, w.row_item,
                       n_rows, n_tiles, reinterpret_cast<int4 *>(d_trow), d_trow_ptr);
    return launch_status();
}
