"""Score layouts of W -- host (numpy) specifications and their device (tensor-op) builders.

The tiled CSR form (all modes; include/rtrec_amd.h) and the feature-row form (SPARSE mode, W with at most 128 non-empty
rows: DESIGN.md section 2).  The numpy builders are the executable specification; the device builders produce identical
arrays (tests/test_host_logic.py::test_device_layout_builders_equal_the_host_builders).  The segment form lives in
seg_layout.py.  (Split out of engine.py in round 4: VERDICT round 3, repo hygiene.)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

@dataclass
class TiledW:
    """Host-side description of one shard of W in the kernel's tiled layout."""
    n_items: int
    col_lo: int
    n_cols: int           # layout columns (shard width, or number of active columns when compacted)
    tile_cols: int
    n_tiles: int
    tile_ptr: np.ndarray  # int32 [n_tiles * (n_items + 1)]
    w_col: np.ndarray     # uint16 [nnz]
    w_val: np.ndarray     # float32 [nnz]
    col_ids: Optional[np.ndarray] = None   # int32 [n_cols]: layout column -> global item id
    col_map: Optional[np.ndarray] = None   # int32 [n_items]: global item id -> layout column or -1
    dense_idx: Optional[np.ndarray] = None  # int32 [n_tiles * n_items]: dense block of (tile, row) or -1
    dense_val: Optional[np.ndarray] = None  # float32 [n_dense * tile_cols]: zero-padded dense rows


def build_tiled_w(W_csc: sp.csc_matrix, col_lo: int, col_hi: int, tile_cols: int, compact: bool = False,
                  dense_fill: Optional[float] = None) -> TiledW:
    """Cut columns [col_lo, col_hi) of W (CSC, I x I) into tiles; per tile a CSR over all rows.
    compact=True keeps only the columns that store at least one weight (the only ones that can
    be recommended in SPARSE mode), in ascending id order."""
    n_items = W_csc.shape[0]
    indptr = np.asarray(W_csc.indptr, dtype=np.int64)
    s, e = int(indptr[col_lo]), int(indptr[col_hi])
    rows = np.asarray(W_csc.indices[s:e], dtype=np.int64)
    vals = np.asarray(W_csc.data[s:e], dtype=np.float32)
    counts = np.diff(indptr[col_lo:col_hi + 1])
    col_ids = col_map = None
    if compact:
        active = np.flatnonzero(counts > 0)
        n_cols = int(active.shape[0])
        col_ids = (active + col_lo).astype(np.int32)
        col_map = np.full(n_items, -1, dtype=np.int32)
        col_map[col_ids] = np.arange(n_cols, dtype=np.int32)
        kloc = np.repeat(np.arange(n_cols, dtype=np.int64), counts[active])
    else:
        n_cols = col_hi - col_lo
        kloc = np.repeat(np.arange(n_cols, dtype=np.int64), counts)
    tile_cols = max(256, min(int(tile_cols), -(-max(n_cols, 1) // 256) * 256))
    n_tiles = max(1, -(-n_cols // tile_cols))
    tile = kloc // tile_cols
    order = np.lexsort((kloc, rows, tile))
    key = (tile * n_items + rows)[order]
    cnt = np.bincount(key, minlength=n_tiles * n_items)
    # (tile, row) segments that fill at least `dense_fill` of the tile are stored as zero-padded
    # dense vectors: the kernel then updates 4 accumulators per lane and instruction
    dense_idx = dense_val = None
    kl, vl = kloc[order], vals[order]
    if dense_fill is not None:
        dense_keys = np.flatnonzero(cnt >= max(64, int(dense_fill * tile_cols)))
        if dense_keys.size:
            dense_idx = np.full(n_tiles * n_items, -1, dtype=np.int32)
            dense_idx[dense_keys] = np.arange(dense_keys.size, dtype=np.int32)
            is_dense = dense_idx[key] >= 0
            dense_val = np.zeros(dense_keys.size * tile_cols, dtype=np.float32)
            dense_val[dense_idx[key[is_dense]].astype(np.int64) * tile_cols + kl[is_dense] % tile_cols] = vl[is_dense]
            key, kl, vl = key[~is_dense], kl[~is_dense], vl[~is_dense]
            cnt = np.bincount(key, minlength=n_tiles * n_items)
    starts = np.zeros(n_tiles * n_items + 1, dtype=np.int64)
    np.cumsum(cnt, out=starts[1:])
    if starts[-1] >= 2 ** 31:
        raise ValueError("W shard has more than 2**31 stored weights")
    tile_ptr = np.empty(n_tiles * (n_items + 1), dtype=np.int32)
    for t in range(n_tiles):
        tile_ptr[t * (n_items + 1):(t + 1) * (n_items + 1)] = starts[t * n_items:t * n_items + n_items + 1]
    w_col = (kl % tile_cols).astype(np.uint16)
    return TiledW(n_items, col_lo, n_cols, tile_cols, n_tiles, tile_ptr, w_col, np.ascontiguousarray(vl),
                  col_ids, col_map, dense_idx, dense_val)


def row_header_table(T: TiledW) -> np.ndarray:
    """[n_tiles, n_items, 4] int32 records {ptr begin, ptr end, dense block or -1, tile-local layout column
    of the item or -1}: everything the kernel looks up per (tile, user item), in one 16-byte gather."""
    n_items, S = T.n_items, T.tile_cols
    tp = T.tile_ptr.reshape(T.n_tiles, n_items + 1)
    hdr = np.empty((T.n_tiles, n_items, 4), dtype=np.int32)
    hdr[:, :, 0] = tp[:, :-1]
    hdr[:, :, 1] = tp[:, 1:]
    hdr[:, :, 2] = T.dense_idx.reshape(T.n_tiles, n_items) if T.dense_idx is not None else -1
    loc = T.col_map.astype(np.int64) if T.col_map is not None else np.arange(n_items, dtype=np.int64) - T.col_lo
    loc = np.where((loc >= 0) & (loc < T.n_cols), loc, -1)
    for t in range(T.n_tiles):
        l = loc - t * S
        hdr[t, :, 3] = np.where((loc >= 0) & (l >= 0) & (l < S), l, -1)
    return hdr


FR_MAX_ROWS = 128           # kFrMaxRows of csrc/score.hip
FR_MIN_FILL = 1.0 / 64.0    # the dense R x n_cols form pays when at least this share of it is stored weights


def build_feature_rows(W_csc: sp.csc_matrix, col_lo: int, col_hi: int, col_ids: np.ndarray, col_map: np.ndarray,
                       tile_cols: int = 256) -> Optional[Dict[str, Any]]:
    """"Feature-row" form of columns [col_lo, col_hi) of W for score_frows_kernel (include/rtrec_amd.h,
    rtrec_score_opts): only items that some column selected with a non-zero weight have a row in W; when
    those rows are few the shard is the small dense matrix of those rows over the compacted columns, cut
    into tiles of 256 (<= 66 rows) or 128 columns so that two slices fit LDS.  None when W does not have
    that shape (many rows, or hardly filled): the tiled-CSR kernel serves it then."""
    n_items = W_csc.shape[0]
    indptr = np.asarray(W_csc.indptr, dtype=np.int64)
    s, e = int(indptr[col_lo]), int(indptr[col_hi])
    rows = np.asarray(W_csc.indices[s:e], dtype=np.int64)
    vals = np.asarray(W_csc.data[s:e], dtype=np.float32)
    cols = np.repeat(np.arange(col_lo, col_hi, dtype=np.int64), np.diff(indptr[col_lo:col_hi + 1]))
    F = np.unique(rows)
    R, n_cols = int(len(F)), int(len(col_ids))
    if R == 0 or R > FR_MAX_ROWS or n_cols == 0 or len(vals) < FR_MIN_FILL * R * n_cols:
        return None
    tc = 128 if int(tile_cols) == 128 else 256      # 256: four sums per lane and instruction; slices are cut into fragments, so
                                                    # the tallest tile (R rows) no longer has to fit one LDS buffer
    n_tiles = -(-n_cols // tc)
    if n_tiles * (tc // 64) > 416:
        return None
    fmap = np.full(n_items, -1, dtype=np.int32)
    fmap[F] = np.arange(R, dtype=np.int32)
    # Column order: the kernel skips (row, tile) blocks without a weight, so columns that use the same RARE rows
    # should share tiles.  Sort the columns lexicographically by their row pattern, rarest row first (ML-20M
    # shape: 35 % of the blocks a pass visits are non-empty instead of 94 % in item-id order).  Any order gives
    # the same scores: a skipped block only ever added +-0.
    f_of, c_of = fmap[rows].astype(np.int64), col_map[cols].astype(np.int64)
    pattern = np.zeros((R, n_cols), dtype=bool)
    pattern[f_of, c_of] = True
    by_rarity = np.argsort(pattern.sum(axis=1), kind="stable")          # rarest row = primary key = last lexsort key
    # ... descending, so that the columns with the most / rarest rows -- the high scorers -- come FIRST: the
    # kernel's running top-k then settles within the first tiles (ascending, nearly every column displaces one)
    order = np.lexsort(tuple(pattern[r] for r in by_rarity[::-1]))[::-1]     # layout position -> compacted column
    # ... and the TILES this order forms are visited heaviest first (sum of |w|): the columns that end up in a user's
    # top-k are overwhelmingly in the heavy tiles, so the running k-th best score is near its final value after the
    # first tiles and almost nothing enters the lists later (ML-20M shape: 113 -> 14 list candidates per user).
    order = _heavy_tiles_first(order, np.bincount(c_of, weights=np.abs(vals).astype(np.float64), minlength=n_cols), tc)
    fr_col_ids = np.asarray(col_ids, dtype=np.int32)[order]
    fr_col_map = np.full(n_items, -1, dtype=np.int32)
    fr_col_map[fr_col_ids] = np.arange(n_cols, dtype=np.int32)
    lc = fr_col_map[cols].astype(np.int64)
    t_of = lc // tc
    present = np.zeros((n_tiles, R), dtype=bool)            # (tile, row) blocks that hold a weight
    present[t_of, f_of] = True
    n_rows_t = present.sum(axis=1)
    local = np.cumsum(present, axis=1) - 1                  # row -> index inside the tile's compact slice
    P = _pack_fragments(n_rows_t.astype(np.int64), tc)           # slices -> fragments -> super-tiles staged in LDS
    k_of = local[t_of, f_of]                                     # rank of the weight's row among its tile's stored rows
    g_of = P["frag_of"][t_of, k_of]                              # ... and the fragment that holds it
    wd = np.zeros(max(int(P["super_kb"][-1]) * 256, 256), dtype=np.float32)
    base = P["super_kb"][P["frag_super"][g_of]] * 256 + P["frag_off"][g_of] // 4
    wd[base + (k_of - P["frag_k0"][g_of]) * tc + lc % tc] = vals
    # tile headers: max |w| per row -- the kernel skips a tile for a wave when sum_f |x_f| max|w_f| cannot beat any of
    # its users' current (k+1)-th best scores
    hbase = P["super_kb"][P["frag_super"][P["first_frag"]]] * 256 + (P["frag_off"][P["first_frag"]] - FR_TILE_HEADER_BYTES) // 4
    np.maximum.at(wd, hbase[t_of] + f_of, np.abs(vals))
    frag_rows = np.zeros((P["n_frags"], 2), dtype=np.uint64)
    np.bitwise_or.at(frag_rows, (g_of, f_of // 64), np.uint64(1) << (f_of % 64).astype(np.uint64))
    tile_rows = np.zeros((n_tiles, 2), dtype=np.uint64)
    np.bitwise_or.at(tile_rows, (t_of, f_of // 64), np.uint64(1) << (f_of % 64).astype(np.uint64))
    return dict(fr_map=fmap, fr_col_ids=fr_col_ids, fr_col_map=fr_col_map, fr_w=wd, fr_tile_rows=frag_rows.view(np.int64),
                fr_tile_off=P["frag_off"].astype(np.int32), fr_frag_tile=P["frag_flags"].astype(np.int32),
                fr_super_kb=P["super_kb"].astype(np.int32), fr_super_tile=P["super_frag"].astype(np.int32), fr_rows=R,
                fr_tile_cols=tc, fr_n_tiles=n_tiles, fr_n_frags=P["n_frags"], fr_n_super=P["n_super"], fr_buf_bytes=P["buf_bytes"],
                fr_rows_of_tile=tile_rows.view(np.int64))



def _heavy_tiles_first(order: np.ndarray, col_mass: np.ndarray, tc: int) -> np.ndarray:
    """Permute the full tiles of a column order by descending weight mass (float32-rounded, ties: earlier tile first);
    a partial last tile stays last."""
    n_full = len(order) // tc
    if n_full < 2:
        return order
    head = order[:n_full * tc].reshape(n_full, tc)
    tmass = col_mass[head].sum(axis=1).astype(np.float32)
    return np.concatenate([head[np.argsort(-tmass, kind="stable")].ravel(), order[n_full * tc:]])


FR_TILE_HEADER_BYTES = 512          # per tile, in front of its first fragment: max |w| of each of the (<= 128) rows in the tile
FR_STREAM_BUF_BYTES = 36 * 1024     # slice buffer of the streaming layout: two 8-wave workgroups (2 buffers each) share a CU's LDS


def _pack_fragments(n_rows_t: np.ndarray, tc: int) -> Dict[str, Any]:
    """Lay the tiles' slices (n_rows_t[t] rows of tc floats each, rows ascending) out for LDS staging -- shared by the
    host and device builders of the feature-row layout.

    RESIDENT form: everything fits one CU's LDS next to the per-wave setup scratch -> one super-tile, one fragment
    per tile; the kernel loads it once per workgroup.  STREAMING form: super-tiles of at most FR_STREAM_BUF_BYTES, filled
    greedily with FRAGMENTS -- a tile's slice may continue in the next super-tile (its accumulators stay in registers
    across the hand-over), so the buffer size is independent of the tallest tile.
    Returns per fragment: tile, k0, k1 (ranks of the tile's stored rows it holds), first / last flags, byte offset inside
    its super-tile; per super-tile: first fragment, KiB offset in the weight array; buf_bytes; resident."""
    row_bytes = tc * 4
    n_tiles = len(n_rows_t)
    total = -(-(int(n_rows_t.sum()) * row_bytes + n_tiles * FR_TILE_HEADER_BYTES) // 1024) * 1024
    setup = 16 * (-(-(n_tiles * (tc // 64) * 8 + 768) // 256) * 256)
    resident = n_tiles <= 64 and total + setup + 16 * 512 + 1024 + 16 <= 160 * 1024
    cap = max(total, 1024) if resident else FR_STREAM_BUF_BYTES
    f_tile, f_k0, f_k1, f_off, f_super, st_frag, used = [], [], [], [], [], [0], 0
    for t in range(n_tiles):
        k, n = 0, int(n_rows_t[t])
        while k < n:
            hdr = FR_TILE_HEADER_BYTES if k == 0 else 0      # the tile's header sits in front of its first fragment
            room = (cap - used - hdr) // row_bytes
            # close the super-tile when it is full, holds 64 fragments (a lane per fragment), or the rest of it would
            # take less than 8 rows of a slice that needs more
            if used > 0 and (room < min(n - k, 8) or len(f_tile) - st_frag[-1] >= 64):
                st_frag.append(len(f_tile))
                used = 0
                continue
            take = min(n - k, room)
            f_tile.append(t); f_k0.append(k); f_k1.append(k + take); f_off.append(used + hdr); f_super.append(len(st_frag) - 1)
            used += hdr + take * row_bytes
            k += take
    st_frag.append(len(f_tile))
    f_tile, f_k0, f_k1 = (np.asarray(a, dtype=np.int64) for a in (f_tile, f_k0, f_k1))
    f_super, st_frag = np.asarray(f_super, dtype=np.int64), np.asarray(st_frag, dtype=np.int64)
    n_super = len(st_frag) - 1
    bytes_s = np.zeros(n_super, dtype=np.int64)
    np.add.at(bytes_s, f_super, (f_k1 - f_k0) * row_bytes + np.where(f_k0 == 0, FR_TILE_HEADER_BYTES, 0))
    super_kb = np.zeros(n_super + 1, dtype=np.int64)
    super_kb[1:] = np.cumsum(-(-bytes_s // 1024))
    flags = f_tile | ((f_k0 == 0).astype(np.int64) << 24) | ((f_k1 == n_rows_t[f_tile]).astype(np.int64) << 25)
    # (tile, rank of a stored row) -> fragment
    frag_of = np.zeros((n_tiles, max(int(n_rows_t.max()), 1)), dtype=np.int64)
    for i in range(len(f_tile)):
        frag_of[f_tile[i], f_k0[i]:f_k1[i]] = i
    buf_bytes = int(cap) if resident else FR_STREAM_BUF_BYTES
    first_frag = np.zeros(n_tiles, dtype=np.int64)
    first_frag[f_tile[f_k0 == 0]] = np.flatnonzero(f_k0 == 0)
    return dict(first_frag=first_frag, frag_tile=f_tile, frag_k0=f_k0, frag_flags=flags, frag_off=np.asarray(f_off, dtype=np.int64), frag_super=f_super,
                super_frag=st_frag, super_kb=super_kb, frag_of=frag_of, buf_bytes=buf_bytes, resident=bool(resident),
                n_frags=len(f_tile), n_super=n_super)


def build_feature_rows_device(torch, rows, cols, vals, n_items: int, col_lo: int, col_hi: int,
                              tile_cols: int = 256) -> Optional[Dict[str, Any]]:
    """build_feature_rows for a W that is resident on the device as COO triples (int64 rows / cols sorted by (col, row),
    float32 vals): the same layout, built with tensor ops -- only the per-tile row counts (a few dozen integers) visit
    the host for the super-tile packing.  Returns device tensors (plus the scalars and, for bench.py, small host copies)."""
    sel = (cols >= col_lo) & (cols < col_hi)
    r, c, v = rows[sel], cols[sel], vals[sel]
    if r.numel() == 0:
        return None
    F = torch.unique(r)
    col_ids_sorted = torch.unique(c)
    R, n_cols = int(F.numel()), int(col_ids_sorted.numel())
    if R > FR_MAX_ROWS or r.numel() < FR_MIN_FILL * R * n_cols:
        return None
    tc = 128 if int(tile_cols) == 128 else 256      # 256: four sums per lane and instruction; slices are cut into fragments, so
                                                    # the tallest tile (R rows) no longer has to fit one LDS buffer
    n_tiles = -(-n_cols // tc)
    if n_tiles * (tc // 64) > 416:
        return None
    dev = r.device
    i64 = torch.int64
    fmap = torch.full((n_items,), -1, dtype=torch.int32, device=dev)
    fmap[F] = torch.arange(R, dtype=torch.int32, device=dev)
    f_of = fmap[r].to(i64)
    c_of = torch.searchsorted(col_ids_sorted, c)
    # column order: lexicographic by row pattern, rarest row most significant, descending (see build_feature_rows)
    counts = torch.bincount(f_of, minlength=R)
    by_rarity = torch.argsort(counts, stable=True)
    rank = torch.empty(R, dtype=i64, device=dev)
    rank[by_rarity] = torch.arange(R, dtype=i64, device=dev)
    sig = (R - 1) - rank[f_of]                                  # bit significance of the entry's row: rarest = highest
    n_words = -(-R // 60)
    order = torch.arange(n_cols, dtype=i64, device=dev).flip(0)  # ties: descending column position, like the host builder
    for w in range(n_words):                                     # least significant word first, stable sorts
        inw = (sig // 60) == w
        word = torch.zeros(n_cols, dtype=i64, device=dev)
        word.index_add_(0, c_of[inw], torch.ones_like(sig[inw]) << (sig[inw] % 60))
        order = order[torch.argsort(word[order], descending=True, stable=True)]
    n_full = n_cols // tc                                        # tiles heaviest first (see build_feature_rows)
    if n_full > 1:
        col_mass = torch.zeros(n_cols, dtype=torch.float64, device=dev).index_add_(0, c_of, v.abs().double())
        head = order[:n_full * tc].view(n_full, tc)
        tmass = col_mass[head].sum(dim=1).float()
        order = torch.cat([head[torch.argsort(tmass, descending=True, stable=True)].reshape(-1), order[n_full * tc:]])
    fr_col_ids = col_ids_sorted[order].to(torch.int32)
    fr_col_map = torch.full((n_items,), -1, dtype=torch.int32, device=dev)
    fr_col_map[fr_col_ids.to(i64)] = torch.arange(n_cols, dtype=torch.int32, device=dev)
    lc = fr_col_map[c].to(i64)
    t_of = lc // tc
    present = torch.zeros((n_tiles, R), dtype=torch.bool, device=dev)
    present[t_of, f_of] = True
    n_rows_t = present.sum(dim=1).cpu().numpy().astype(np.int64)
    P = _pack_fragments(n_rows_t, tc)
    local = torch.cumsum(present.to(i64), dim=1) - 1
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    k_of = local[t_of, f_of]
    g_of = dv(P["frag_of"])[t_of, k_of]
    wd = torch.zeros(max(int(P["super_kb"][-1]) * 256, 256), dtype=torch.float32, device=dev)
    base = dv(P["super_kb"])[dv(P["frag_super"])[g_of]] * 256 + dv(P["frag_off"])[g_of] // 4
    wd[base + (k_of - dv(P["frag_k0"])[g_of]) * tc + lc % tc] = v
    hbase = dv(P["super_kb"][P["frag_super"][P["first_frag"]]] * 256 + (P["frag_off"][P["first_frag"]] - FR_TILE_HEADER_BYTES) // 4)
    wd.scatter_reduce_(0, hbase[t_of] + f_of, v.abs(), reduce="amax")             # tile headers: max |w| per row
    # one bit per (row, fragment) / (row, tile) block that holds a weight
    blk_key = torch.unique(g_of * 128 + f_of)
    bg, bf = blk_key // 128, blk_key % 128
    frag_rows = torch.zeros((P["n_frags"], 2), dtype=i64, device=dev)
    frag_rows.view(-1).index_add_(0, bg * 2 + bf // 64, torch.ones_like(bf) << (bf % 64))
    blk = torch.zeros((n_tiles, 2), dtype=i64, device=dev)
    pt, pf = torch.nonzero(present, as_tuple=True)
    blk.view(-1).index_add_(0, pt * 2 + pf // 64, torch.ones_like(pf) << (pf % 64))
    fmap_host = fmap.cpu().numpy()
    host = dict(fr_map=fmap_host, fr_rows_of_tile=blk.cpu().numpy(), fr_super_kb=P["super_kb"].astype(np.int32), fr_rows=R,
                fr_tile_cols=tc, fr_resident=P["resident"])
    return dict(fr_map=fmap, fr_col_ids=fr_col_ids, fr_col_map=fr_col_map, fr_w=wd, fr_tile_rows=frag_rows,
                fr_tile_off=dv(P["frag_off"].astype(np.int32)), fr_frag_tile=dv(P["frag_flags"].astype(np.int32)),
                fr_super_kb=dv(P["super_kb"].astype(np.int32)), fr_super_tile=dv(P["super_frag"].astype(np.int32)),
                fr_rows=R, fr_tile_cols=tc, fr_n_tiles=n_tiles, fr_n_frags=P["n_frags"], fr_n_super=P["n_super"],
                fr_buf_bytes=P["buf_bytes"], fr_host=host, col_ids_sorted=col_ids_sorted)


def build_tiled_w_device(torch, rows, cols, vals, n_items: int, col_lo: int, col_hi: int, tile_cols: int,
                         compact: bool = False, dense_fill: Optional[float] = None) -> Optional[Dict[str, Any]]:
    """build_tiled_w + row_header_table for a W that is resident on the device (COO triples sorted by (col, row)):
    the same arrays, as device tensors, built with tensor ops."""
    sel = (cols >= col_lo) & (cols < col_hi)
    r, c, v = rows[sel], cols[sel], vals[sel]
    dev, i64 = rows.device, torch.int64
    col_ids = col_map = None
    if compact:
        col_ids = torch.unique(c)
        n_cols = int(col_ids.numel())
        kloc = torch.searchsorted(col_ids, c)
        col_map = torch.full((n_items,), -1, dtype=torch.int32, device=dev)
        col_map[col_ids] = torch.arange(n_cols, dtype=torch.int32, device=dev)
    else:
        n_cols = col_hi - col_lo
        kloc = c - col_lo
    if n_cols <= 0:
        return None
    S = max(256, min(int(tile_cols), -(-max(n_cols, 1) // 256) * 256))
    n_tiles = max(1, -(-n_cols // S))
    seg = (kloc // S) * n_items + r                                # (tile, row) segment of every weight
    order = torch.argsort(seg * S + kloc % S)                      # by tile, row, column
    seg, kl, vl = seg[order], kloc[order] % S, v[order]
    cnt = torch.bincount(seg, minlength=n_tiles * n_items)
    dense_idx = dense_val = None
    if dense_fill is not None:
        dense_keys = torch.nonzero(cnt >= max(64, int(dense_fill * S))).view(-1)
        if dense_keys.numel():
            dense_idx = torch.full((n_tiles * n_items,), -1, dtype=torch.int32, device=dev)
            dense_idx[dense_keys] = torch.arange(dense_keys.numel(), dtype=torch.int32, device=dev)
            is_dense = dense_idx[seg] >= 0
            dense_val = torch.zeros(int(dense_keys.numel()) * S, dtype=torch.float32, device=dev)
            dense_val[dense_idx[seg[is_dense]].to(i64) * S + kl[is_dense]] = vl[is_dense]
            seg, kl, vl = seg[~is_dense], kl[~is_dense], vl[~is_dense]
            cnt = torch.bincount(seg, minlength=n_tiles * n_items)
    starts = torch.zeros(n_tiles * n_items + 1, dtype=i64, device=dev)
    torch.cumsum(cnt, 0, out=starts[1:])
    if int(starts[-1]) >= 2 ** 31:
        raise ValueError("W shard has more than 2**31 stored weights")
    tile_ptr = torch.empty((n_tiles, n_items + 1), dtype=torch.int32, device=dev)
    tile_ptr[:, :-1] = starts[:-1].view(n_tiles, n_items).to(torch.int32)
    tile_ptr[:, -1] = starts[torch.arange(1, n_tiles + 1, device=dev) * n_items].to(torch.int32)
    hdr = torch.empty((n_tiles, n_items, 4), dtype=torch.int32, device=dev)
    hdr[:, :, 0] = tile_ptr[:, :-1]
    hdr[:, :, 1] = tile_ptr[:, 1:]
    hdr[:, :, 2] = dense_idx.view(n_tiles, n_items) if dense_idx is not None else -1
    loc = col_map.to(i64) if col_map is not None else torch.arange(n_items, dtype=i64, device=dev) - col_lo
    loc = torch.where((loc >= 0) & (loc < n_cols), loc, torch.full_like(loc, -1))
    for t in range(n_tiles):
        l = loc - t * S
        hdr[t, :, 3] = torch.where((loc >= 0) & (l >= 0) & (l < S), l, torch.full_like(l, -1)).to(torch.int32)
    return dict(n_cols=n_cols, tile_cols=S, n_tiles=n_tiles, nnz=int(vl.numel()), dense_idx=dense_idx, dense_val=dense_val,
                n_dense=0 if dense_idx is None else int(dense_val.numel() // S),
                tile_ptr=tile_ptr.view(-1), w_col=kl.to(torch.int16), w_val=vl.contiguous(),
                col_ids=None if col_ids is None else col_ids.to(torch.int32), col_map=col_map, row_hdr=hdr)
