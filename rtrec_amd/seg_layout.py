"""Segment layout of W for score_seg_kernel (csrc/score_seg.hip.h): the SPARSE-mode scoring path for a GENERAL W.

A W fitted on data with item-item structure has thousands of non-empty rows (every cluster's popular items are the
features of that cluster's columns) and its weight sits in blocks: the columns that select a row are that row's
neighbours.  The layout makes those blocks contiguous and prices them:

  * columns: the columns that hold a weight, ORDERED BY CLUSTER -- a few rounds of label propagation over the graph
    of W (`cluster_labels*`) put columns with common features next to each other; any order gives the same scores
    (a column's sum runs over the user's items in ascending item order whatever position the column has);
  * tiles of T columns (T = 256 unless that makes more than 128 tiles; at most 4096) -- a wave accumulates one
    (user, tile) at a time in T floats of LDS;
  * rows: only items that hold a weight have one (`info[item] = (row, layout column)`); a row's entries are sorted by
    layout column, so its SEGMENT in tile t is the range seg_ptr[row][t] .. seg_ptr[row][t + 1] of the 8-byte records
    ent = {column inside the tile, float32 bits of the weight};
  * bounds: bound[row][t] = max |w| of the segment, rounded UP to bfloat16 -- sum_i |x_ui| bound[i][t] bounds every
    score user u can have in tile t, so a tile that cannot beat the user's current (k+1)-th best is never opened.

`build_seg_layout` (numpy) is the executable specification; `build_seg_layout_device` builds the same arrays with
tensor ops from the device-resident W (tests/test_host_logic.py requires identical arrays).
Reference path served: slim_elastic.py:707-708 (X[users] @ W) + :782-818 (_sparse_topk_indicies).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import scipy.sparse as sp

SG_MIN_TILE = 256
SG_MAX_TILE = 4096
SG_MAX_TILES = 128           # two tiles per lane of the bound registers
SG_LPA_ITERS = 8
SG_DENSE_MIN = 64            # 256-column tiles: a segment with more entries than this is stored as 256 floats
SG_MAX_RECORDS = 1 << 28     # 8-byte records the kernel's buffer view addresses
_HASH_MUL = 2654435761


def seg_tile_cols(n_cols: int) -> Optional[int]:
    """Tile width for n_cols active columns: the smallest power of two >= 256 that needs <= 128 tiles; None when even
    4096-column tiles are too many (> 524,288 active columns in one shard: the tiled-CSR kernel serves that)."""
    T = SG_MIN_TILE
    while -(-max(n_cols, 1) // T) > SG_MAX_TILES:
        T *= 2
    return T if T <= SG_MAX_TILE else None


def _edge_weights_int(absw: np.ndarray) -> np.ndarray:
    """|w| as integers in 1 .. 2**20 + 1 (relative to the largest |w|): label sums are then exact and independent of
    the order of summation, so the host and the device builder agree bit for bit."""
    top = float(absw.max()) if absw.size else 1.0
    top = top if top > 0 else 1.0
    return (absw.astype(np.float64) / top * float(1 << 20)).astype(np.int64) + 1


def cluster_labels(rows: np.ndarray, cols: np.ndarray, vals: np.ndarray, n_items: int, iters: int = SG_LPA_ITERS) -> np.ndarray:
    """Label propagation over the undirected graph of W (edge i -- j weighted |W[i, j]|): every node repeatedly takes the
    label that carries the most weight among its neighbours (ties: the larger label); half of the nodes (a hash parity
    that alternates per round) update per round, which keeps two-cycles from oscillating.  Returns int64 labels[n_items]
    (isolated nodes keep their own id).  Deterministic; cluster_labels_device computes the same labels."""
    rows, cols = np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)
    wq = _edge_weights_int(np.abs(np.asarray(vals)))
    src, dst, w = np.concatenate([rows, cols]), np.concatenate([cols, rows]), np.concatenate([wq, wq])
    lab = np.arange(n_items, dtype=np.int64)
    h = (np.arange(n_items, dtype=np.int64) * _HASH_MUL) >> 15
    bits = max(1, int(n_items - 1).bit_length())
    for it in range(iters):
        key = dst * n_items + lab[src]
        o = np.argsort(key, kind="stable")
        k, ww = key[o], w[o]
        first = np.concatenate([[True], k[1:] != k[:-1]]) if len(k) else np.zeros(0, bool)
        start = np.flatnonzero(first)
        sums = np.add.reduceat(ww, start) if len(start) else np.zeros(0, np.int64)
        gk = k[first]
        node, l = gk // n_items, gk % n_items
        best = np.zeros(n_items, dtype=np.int64)
        np.maximum.at(best, node, (sums << bits) | l)
        new = np.where(best > 0, best & ((1 << bits) - 1), lab)
        upd = ((h + it) & 1) == 0
        lab = np.where(upd, new, lab)
    return lab


def cluster_labels_device(torch, rows, cols, vals, n_items: int, iters: int = SG_LPA_ITERS):
    """cluster_labels with tensor ops (one sort of the 2 nnz edge keys per round)."""
    dev, i64 = rows.device, torch.int64
    absw = vals.abs().double()
    top = float(absw.max()) if absw.numel() else 1.0
    top = top if top > 0 else 1.0
    wq = (absw / top * float(1 << 20)).to(i64) + 1
    src, dst, w = torch.cat([rows, cols]).to(i64), torch.cat([cols, rows]).to(i64), torch.cat([wq, wq])
    lab = torch.arange(n_items, dtype=i64, device=dev)
    h = (torch.arange(n_items, dtype=i64, device=dev) * _HASH_MUL) >> 15
    bits = max(1, int(n_items - 1).bit_length())
    for it in range(iters):
        key = dst * n_items + lab[src]
        k, o = torch.sort(key, stable=True)
        ww = w[o]
        gk, inv = torch.unique_consecutive(k, return_inverse=True)
        sums = torch.zeros(gk.numel(), dtype=i64, device=dev).index_add_(0, inv, ww)       # integers: exact in any order
        node, l = gk // n_items, gk % n_items
        best = torch.zeros(n_items, dtype=i64, device=dev).scatter_reduce_(0, node, (sums << bits) | l, reduce="amax")
        new = torch.where(best > 0, best & ((1 << bits) - 1), lab)
        upd = ((h + it) & 1) == 0
        lab = torch.where(upd, new, lab)
    return lab


def build_seg_layout(W_csc: sp.csc_matrix, col_lo: int, col_hi: int, labels: Optional[np.ndarray] = None,
                     iters: int = SG_LPA_ITERS) -> Optional[Dict[str, Any]]:
    """Segment layout of columns [col_lo, col_hi) of W (CSC, I x I, float32 values).  labels: cluster label per item
    (None: label propagation over the shard's own graph; np.arange(I) keeps the item-id order)."""
    n_items = W_csc.shape[0]
    indptr = np.asarray(W_csc.indptr, dtype=np.int64)
    s, e = int(indptr[col_lo]), int(indptr[col_hi])
    rows = np.asarray(W_csc.indices[s:e], dtype=np.int64)
    vals = np.asarray(W_csc.data[s:e], dtype=np.float32)
    cols = np.repeat(np.arange(col_lo, col_hi, dtype=np.int64), np.diff(indptr[col_lo:col_hi + 1]))
    if rows.size == 0:
        return None
    active = np.unique(cols)
    n_cols = int(active.size)
    T = seg_tile_cols(n_cols)
    if T is None:
        return None
    n_tiles = -(-n_cols // T)
    if labels is None:
        labels = cluster_labels(rows, cols, vals, n_items, iters)
    order = active[np.argsort(np.asarray(labels)[active], kind="stable")]       # layout position -> item id
    pos = np.full(n_items, -1, dtype=np.int64)
    pos[order] = np.arange(n_cols)
    F = np.unique(rows)
    R = int(F.size)
    rmap = np.full(n_items, -1, dtype=np.int64)
    rmap[F] = np.arange(R)
    rr, pc = rmap[rows], pos[cols]
    key = rr * (n_tiles * T) + pc
    o = np.argsort(key, kind="stable")
    key = key[o]
    if key.size >= 2 ** 31:
        raise ValueError("W shard has more than 2**31 stored weights")
    bnd = (np.arange(R, dtype=np.int64)[:, None] * (n_tiles * T) + np.arange(n_tiles + 1, dtype=np.int64)[None, :] * T).ravel()
    seg_ptr = np.searchsorted(key, bnd).astype(np.int32).reshape(R, n_tiles + 1)
    mx = np.zeros(R * SG_MAX_TILES, dtype=np.float32)
    np.maximum.at(mx, rr * SG_MAX_TILES + pc // T, np.abs(vals))
    up = (mx.view(np.uint32).astype(np.int64) + 0xFFFF) >> 16                    # bfloat16, rounded up
    up = up.reshape(R, SG_MAX_TILES // 2, 2)
    bound = (up[:, :, 0] | (up[:, :, 1] << 16)).astype(np.uint32).view(np.int32)
    info = np.stack([rmap, pos], axis=1).astype(np.int32)
    # Records.  A segment is stored either SPARSE -- its {column, weight} records, padded with one {T, +0.0} record (the junk slot behind the
    # tile's accumulators) to an even count -- or, in 256-column tiles when it has more than 64 entries, DENSE: 256 floats (128 record slots, 0 where
    # W[row, column] is not stored; adding x * 0 never changes a sum), flagged by bit 31 of its begin pointer.  A sparse
    # segment of a 256-column tile therefore never exceeds one 64-lane step.
    seg_len = np.diff(seg_ptr.astype(np.int64), axis=1)                      # [R, n_tiles]
    dense = (seg_len > SG_DENSE_MIN) if T == SG_MIN_TILE else np.zeros_like(seg_len, dtype=bool)
    alloc = np.where(dense, T // 2, seg_len + (seg_len & 1)).ravel()
    start = np.zeros(alloc.size + 1, dtype=np.int64)
    np.cumsum(alloc, out=start[1:])
    n_rec = int(start[-1])
    if n_rec >= SG_MAX_RECORDS:
        return None
    ent = np.zeros((n_rec, 2), dtype=np.int32)
    ent[:, 0] = T                                                            # pads update the junk slot behind the tile
    seg_of = key // T                                                        # = row * n_tiles + tile
    rank = np.arange(key.size, dtype=np.int64) - seg_ptr[:, :-1].astype(np.int64).ravel()[seg_of]
    col_in = (pc[o] % T).astype(np.int64)
    vbits = vals[o].view(np.int32)
    is_d = dense.ravel()[seg_of]
    dst = start[seg_of[~is_d]] + rank[~is_d]
    ent[dst, 0] = col_in[~is_d]
    ent[dst, 1] = vbits[~is_d]
    dbase = np.flatnonzero(dense.ravel())
    if dbase.size:                                                           # dense blocks are all weights: clear the pad columns
        ent.reshape(-1)[(start[dbase] * 2)[:, None] + np.arange(T)[None, :]] = 0
    ent.reshape(-1)[start[seg_of[is_d]] * 2 + col_in[is_d]] = vbits[is_d]
    ptr2 = np.empty((R, n_tiles + 1), dtype=np.int64)
    ptr2[:, :-1] = start[:-1].reshape(R, n_tiles)
    ptr2[:, -1] = start[n_tiles::n_tiles]
    flagged = ptr2.copy()
    flagged[:, :-1] |= dense.astype(np.int64) << 31
    seg_ptr2 = flagged.astype(np.uint32).view(np.int32)
    # the same segments from the tile's side (heavy pass): per tile its non-empty segments, ascending item
    sr, st = np.nonzero(seg_len)
    o2 = np.lexsort((sr, st))
    sr, st = sr[o2], st[o2]
    trow = np.stack([F[sr].astype(np.int32), seg_ptr2[sr, st], ptr2[sr, st + 1].astype(np.int32), np.zeros(len(sr), np.int32)], axis=1)
    trow_ptr = np.searchsorted(st, np.arange(n_tiles + 1)).astype(np.int32)
    return dict(sg_trow=np.ascontiguousarray(trow).reshape(-1, 4), sg_trow_ptr=trow_ptr, sg_T=T, sg_n_tiles=n_tiles, sg_rows=R,
                sg_n_cols=n_cols, sg_info=np.ascontiguousarray(info), sg_ptr=np.ascontiguousarray(seg_ptr2), sg_ent=ent,
                sg_bound=np.ascontiguousarray(bound), sg_col_ids=order.astype(np.int32), sg_nnz=int(key.size),
                sg_segments=int(np.count_nonzero(seg_len)), sg_dense_segments=int(dense.sum()))


def build_seg_layout_device(torch, rows, cols, vals, n_items: int, col_lo: int, col_hi: int, labels=None,
                            iters: int = SG_LPA_ITERS) -> Optional[Dict[str, Any]]:
    """build_seg_layout for a W resident on the device as COO triples (int64 rows / cols sorted by (col, row), float32
    vals): the same arrays as device tensors.  `labels` (an int64 device tensor, e.g. the previous layout's) skips the
    label propagation."""
    sel = (cols >= col_lo) & (cols < col_hi)
    r, c, v = rows[sel], cols[sel], vals[sel]
    if r.numel() == 0:
        return None
    dev, i64 = r.device, torch.int64
    active = torch.unique(c)
    n_cols = int(active.numel())
    T = seg_tile_cols(n_cols)
    if T is None:
        return None
    n_tiles = -(-n_cols // T)
    if labels is None:
        labels = cluster_labels_device(torch, r, c, v, n_items, iters)
    order = active[torch.argsort(labels[active], stable=True)]
    pos = torch.full((n_items,), -1, dtype=i64, device=dev)
    pos[order] = torch.arange(n_cols, dtype=i64, device=dev)
    F = torch.unique(r)
    R = int(F.numel())
    rmap = torch.full((n_items,), -1, dtype=i64, device=dev)
    rmap[F] = torch.arange(R, dtype=i64, device=dev)
    rr, pc = rmap[r], pos[c]
    key = rr * (n_tiles * T) + pc
    key, o = torch.sort(key, stable=True)
    if int(key.numel()) >= 2 ** 31:
        raise ValueError("W shard has more than 2**31 stored weights")
    bnd = (torch.arange(R, dtype=i64, device=dev)[:, None] * (n_tiles * T)
           + torch.arange(n_tiles + 1, dtype=i64, device=dev)[None, :] * T).reshape(-1)
    seg_ptr = torch.searchsorted(key, bnd).to(torch.int32).view(R, n_tiles + 1)
    mx = torch.zeros(R * SG_MAX_TILES, dtype=torch.float32, device=dev)
    mx.scatter_reduce_(0, rr * SG_MAX_TILES + pc // T, v.abs(), reduce="amax")
    up = (mx.view(torch.int32).to(i64) + 0xFFFF) >> 16
    up = up.view(R, SG_MAX_TILES // 2, 2)
    bound = (up[:, :, 0] | (up[:, :, 1] << 16)).to(torch.int32).contiguous()        # sign bit clear: |w| is positive
    info = torch.stack([rmap, pos], dim=1).to(torch.int32).contiguous()
    # records: sparse segments padded to an even count, dense 256-float blocks (see build_seg_layout)
    sp64 = seg_ptr.to(i64)
    seg_len = sp64[:, 1:] - sp64[:, :-1]
    dense = (seg_len > SG_DENSE_MIN) if T == SG_MIN_TILE else torch.zeros_like(seg_len, dtype=torch.bool)
    alloc = torch.where(dense, torch.full_like(seg_len, T // 2), seg_len + (seg_len & 1)).reshape(-1)
    start = torch.zeros(alloc.numel() + 1, dtype=i64, device=dev)
    torch.cumsum(alloc, 0, out=start[1:])
    n_rec = int(start[-1])
    if n_rec >= SG_MAX_RECORDS:
        return None
    ent = torch.zeros((n_rec, 2), dtype=torch.int32, device=dev)
    ent[:, 0] = T
    seg_of = key // T
    rank = torch.arange(key.numel(), dtype=i64, device=dev) - sp64[:, :-1].reshape(-1)[seg_of]
    col_in = pc[o] % T
    vbits = v[o].contiguous().view(torch.int32)
    is_d = dense.reshape(-1)[seg_of]
    dst = start[seg_of[~is_d]] + rank[~is_d]
    ent[dst, 0] = col_in[~is_d].to(torch.int32)
    ent[dst, 1] = vbits[~is_d]
    dbase = torch.nonzero(dense.reshape(-1)).view(-1)
    if dbase.numel():
        ent.view(-1)[(start[dbase] * 2)[:, None] + torch.arange(T, device=dev)[None, :]] = 0
    ent.view(-1)[start[seg_of[is_d]] * 2 + col_in[is_d]] = vbits[is_d]
    ptr2 = torch.empty((R, n_tiles + 1), dtype=i64, device=dev)
    ptr2[:, :-1] = start[:-1].view(R, n_tiles)
    ptr2[:, -1] = start[n_tiles::n_tiles]
    flagged = ptr2.clone()
    flagged[:, :-1] |= dense.to(i64) << 31
    seg_ptr2 = torch.where(flagged >= 2 ** 31, flagged - 2 ** 32, flagged).to(torch.int32).contiguous()
    sr, st = torch.nonzero(seg_len, as_tuple=True)
    o2 = torch.argsort(st * R + sr)                          # by tile, then row (= item) ascending
    sr, st = sr[o2], st[o2]
    trow = torch.stack([F[sr].to(torch.int32), seg_ptr2[sr, st], ptr2[sr, st + 1].to(torch.int32),
                        torch.zeros_like(sr, dtype=torch.int32)], dim=1).contiguous()
    trow_ptr = torch.searchsorted(st, torch.arange(n_tiles + 1, dtype=i64, device=dev)).to(torch.int32)
    return dict(sg_trow=trow.view(-1, 4), sg_trow_ptr=trow_ptr, sg_T=T, sg_n_tiles=n_tiles, sg_rows=R, sg_n_cols=n_cols, sg_info=info,
                sg_ptr=seg_ptr2, sg_ent=ent, sg_bound=bound, sg_col_ids=order.to(torch.int32).contiguous(), sg_nnz=int(key.numel()),
                sg_labels=labels)


def build_seg_layout_native(be, rows, cols, vals, n_items: int, col_lo: int, col_hi: int, labels) -> Optional[Dict[str, Any]]:
    """build_seg_layout_device through the library's own builder (csrc/seg_build.hip: rtrec_slim_seg_plan + _fill): a dozen
    kernels and one host round trip instead of ~45 tensor ops and five -- what a recommend right after a mini-batch waits
    for.  `labels` must be given (int64 device tensor with values in [0, n_items): the cached cluster labels, or arange).
    Returns None when the shard is empty or too wide for the layout; raises when the library refuses the arguments.
    sg_ent / sg_trow are allocated for the worst case (2 nnz records / min(nnz, segments) list entries): the entries in
    use are the first sg_ptr[-1, -1] / sg_trow_ptr[-1]; the rest is padding the kernels never address."""
    import ctypes as C
    from . import _native
    torch, lib = be.torch, be.lib
    nnz = int(rows.numel())
    if nnz == 0 or nnz >= (1 << 27):
        return None
    rows, cols, vals = rows.contiguous(), cols.contiguous(), vals.contiguous()
    labels = labels.contiguous()
    p = be.ptr
    pws = be.empty((int(lib.rtrec_slim_seg_plan_workspace_bytes(n_items)),), torch.uint8)
    n_cols, R, T, n_tiles = (int(x) for x in be.ops.seg_plan(rows, cols, n_items, int(col_lo), int(col_hi), labels, pws))
    if n_cols == 0 or R == 0 or T == 0:
        return None
    i32 = torch.int32
    fws = be.empty((int(lib.rtrec_slim_seg_fill_workspace_bytes(n_items, nnz, R, n_tiles)),), torch.uint8)
    info, seg_ptr = be.empty((n_items, 2), i32), be.empty((R, n_tiles + 1), i32)
    ent, bound = be.empty((2 * nnz, 2), i32), be.empty((R, 64), i32)
    col_ids, trow_ptr = be.empty((n_cols,), i32), be.empty((n_tiles + 1,), i32)
    trow = be.empty((min(nnz, R * n_tiles), 4), i32)
    be.ops.seg_fill(rows, cols, vals, n_items, int(col_lo), int(col_hi), pws, n_cols, R, T, n_tiles, fws, info, seg_ptr, ent, bound,
                    col_ids, trow_ptr, trow)
    return dict(sg_trow=trow, sg_trow_ptr=trow_ptr, sg_T=T, sg_n_tiles=n_tiles, sg_rows=R, sg_n_cols=n_cols, sg_info=info,
                sg_ptr=seg_ptr, sg_ent=ent, sg_bound=bound, sg_col_ids=col_ids, sg_nnz=nnz, sg_labels=labels,
                _workspaces=(pws, fws))        # the fill is only enqueued: its workspaces live as long as the layout
