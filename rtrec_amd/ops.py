"""PyTorch-ROCm custom ops over the C-ABI (`torch.ops.rtrec_amd.*`).

One op per entry point of include/rtrec_amd.h, in out-variant style: every buffer is a CUDA (ROCm)
tensor owned by the caller, outputs and scratch are listed in `mutates_args`, scalars are plain
ints / floats / bools, and the work is enqueued on the current stream of the tensors' device.  The
ops only marshal pointers -- all computation is in librtrec_amd.so -- so they compose with torch
streams, the caching allocator and torch.distributed without copies.  rtrec_amd.engine.HipBackend
is written on top of them; they are also the interface for callers that already hold device tensors.

    torch.ops.rtrec_amd.column_sqnorms      norm_cols_X of the coordinate descent
    torch.ops.rtrec_amd.fit_workspace_init  scratch invariants of the fit kernel
    torch.ops.rtrec_amd.gram_matrix         Gram matrix of the popular items (Gram tracking)
    torch.ops.rtrec_amd.fit_columns         X^T y feature selection + elastic-net CD per item column
    torch.ops.rtrec_amd.score_topk          user rows x W shard, interacted filter, top-k
    torch.ops.rtrec_amd.score_rows          score vectors (predict*)
    torch.ops.rtrec_amd.merge_topk          per-tile / per-GPU top-k lists -> top-k
    torch.ops.rtrec_amd.similar_topk        similar_items
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
from torch import Tensor
from torch.library import custom_op

from . import _native


def _p(t: Optional[Tensor]) -> C.c_void_p:
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else C.c_void_p(0)


def _stream(t: Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


@custom_op("rtrec_amd::column_sqnorms", mutates_args=("out",), device_types="cuda")
def column_sqnorms(cptr: Tensor, cval: Tensor, out: Tensor) -> None:
    lib = _native.load()
    _native.check(lib.rtrec_slim_column_sqnorms(int(out.shape[0]), _p(cptr), _p(cval), _p(out), _stream(out)),
                  "rtrec_slim_column_sqnorms")


@custom_op("rtrec_amd::fit_workspace_init", mutates_args=("ws",), device_types="cuda")
def fit_workspace_init(ws: Tensor, n_users: int, n_items: int, n_slots: int, top_features: int) -> None:
    lib = _native.load()
    _native.check(lib.rtrec_slim_fit_workspace_init(_p(ws), ws.numel(), n_users, n_items, n_slots, top_features,
                                                    _stream(ws)), "rtrec_slim_fit_workspace_init")


@custom_op("rtrec_amd::gram_matrix", mutates_args=("ws", "gram"), device_types="cuda")
def gram_matrix(cptr: Tensor, crow: Tensor, cval: Tensor, top_items: Tensor, ws: Tensor, gram: Tensor,
                n_users: int, n_items: int) -> None:
    lib = _native.load()
    _native.check(lib.rtrec_slim_gram_matrix(n_users, n_items, _p(cptr), _p(crow), _p(cval), _p(top_items),
                                             int(top_items.shape[0]), _p(ws), ws.numel(), _p(gram), _stream(gram)),
                  "rtrec_slim_gram_matrix")


@custom_op("rtrec_amd::fit_columns",
           mutates_args=("out_items", "out_coef", "out_count", "out_n_iter", "ws", "queue", "trace", "xty_ws"), device_types="cuda")
def fit_columns(cptr: Tensor, crow: Tensor, cval: Tensor, rptr: Tensor, rcol: Tensor, rval: Tensor, sqn: Tensor,
                targets: Tensor, n_users: int, n_items: int,
                l1_reg: float, l2_reg: float, tol: float, max_iter: int, seed: int, positive: bool, top_features: int,
                out_items: Tensor, out_coef: Tensor, out_count: Tensor, out_n_iter: Tensor, cap: int,
                ws: Tensor, n_slots: int, queue: Tensor, trace: Optional[Tensor],
                gram: Optional[Tensor], gram_index: Optional[Tensor], gram_n: int, gram_rel_err: float,
                fast: int, kernel: int, colwalk_min_rows: int, screen_min: int, lane_max: int,
                xty_ws: Optional[Tensor], col_order: Optional[Tensor]) -> None:
    lib = _native.load()
    cfg = _native.FitCfg(l1_reg, l2_reg, tol, max_iter, seed, int(positive), top_features)
    opts = _native.FitOpts(_p(trace), _p(gram), _p(gram_index), gram_n if gram is not None else 0, gram_rel_err,
                           int(fast), kernel, colwalk_min_rows, screen_min, lane_max,
                           _p(xty_ws), xty_ws.numel() if xty_ws is not None else 0, int(rcol.shape[0]), _p(col_order))
    _native.check(lib.rtrec_slim_fit_columns_opt(
        n_users, n_items, _p(cptr), _p(crow), _p(cval), _p(rptr), _p(rcol), _p(rval), _p(sqn),
        _p(targets), int(targets.shape[0]), C.byref(cfg), _p(out_items), _p(out_coef), _p(out_count), _p(out_n_iter),
        cap, _p(ws), ws.numel(), n_slots, _p(queue), _stream(out_items), C.byref(opts)), "rtrec_slim_fit_columns_opt")


@custom_op("rtrec_amd::score_topk", mutates_args=("ids", "scores", "scores64", "aux", "count", "ws", "fr_scratch", "rescored", "sg_scratch", "flagged"),
           device_types="cuda")
def score_topk(row_ids: Optional[Tensor], xb_ptr: Tensor, xb_col: Tensor, xb_val: Tensor, n_rows: int,
               n_items: int, n_cols: int, col_offset: int, col_ids: Optional[Tensor], col_map: Optional[Tensor],
               tile_cols: int, n_tiles: int, tile_ptr: Optional[Tensor], w_col: Optional[Tensor], w_val: Optional[Tensor],
               dense_idx: Optional[Tensor], dense_val: Optional[Tensor], row_hdr: Optional[Tensor],
               col_rank: Optional[Tensor], top_k: int, filter_interacted: bool, mode: int, acc_f64: bool,
               ids: Tensor, scores: Tensor, scores64: Optional[Tensor], aux: Tensor, count: Tensor, ws: Tensor,
               fr_map: Optional[Tensor], fr_col_ids: Optional[Tensor], fr_col_map: Optional[Tensor], fr_w: Optional[Tensor],
               fr_tile_rows: Optional[Tensor], fr_tile_off: Optional[Tensor], fr_super_kb: Optional[Tensor],
               fr_super_tile: Optional[Tensor], fr_frag_tile: Optional[Tensor], fr_rows: int, fr_tile_cols: int, fr_n_tiles: int,
               fr_n_frags: int, fr_n_super: int,
               fr_buf_bytes: int, fr_scratch: Optional[Tensor], row_order: Optional[Tensor], timer: int,
               diagnostics: int, rescored: Optional[Tensor], row_order_grouped: int,
               sg_info: Optional[Tensor], sg_ptr: Optional[Tensor], sg_ent: Optional[Tensor],
               sg_bound: Optional[Tensor], sg_col_ids: Optional[Tensor], sg_tile_cols: int, sg_n_tiles: int, sg_rows: int,
               sg_n_cols: int, sg_trow_ptr: Optional[Tensor], sg_trow: Optional[Tensor], sg_scratch: Optional[Tensor],
               row_order_longest_first: int, flagged: Optional[Tensor]) -> None:
    """rtrec_slim_score_topk_opt.  n_x_rows is taken from xb_ptr; fr_* is the optional feature-row form of the
    shard, sg_* its optional segment form; `timer` is an rtrec_timer handle (0 = none)."""
    lib = _native.load()
    opts = _native.ScoreOpts(int(xb_ptr.shape[0]) - 1, _p(fr_map), _p(fr_col_ids), _p(fr_col_map), _p(fr_w),
                             _p(fr_tile_rows), _p(fr_tile_off), _p(fr_super_kb), _p(fr_super_tile), _p(fr_frag_tile), fr_rows,
                             fr_tile_cols, fr_n_tiles, fr_n_frags, fr_n_super, fr_buf_bytes, _p(fr_scratch), fr_scratch.numel() if fr_scratch is not None else 0,
                             _p(row_order), C.c_void_p(timer or None), diagnostics, _p(rescored), row_order_grouped,
                             _p(sg_info), _p(sg_ptr), _p(sg_ent), int(sg_ent.shape[0]) if sg_ent is not None else 0, _p(sg_bound), _p(sg_col_ids),
                             sg_tile_cols, sg_n_tiles, sg_rows, sg_n_cols, _p(sg_trow_ptr), _p(sg_trow), _p(sg_scratch),
                             sg_scratch.numel() if sg_scratch is not None else 0, row_order_longest_first, _p(flagged))
    _native.check(lib.rtrec_slim_score_topk_opt(
        n_rows, _p(row_ids), _p(xb_ptr), _p(xb_col), _p(xb_val), n_items, n_cols, col_offset, _p(col_ids), _p(col_map),
        tile_cols, n_tiles, _p(tile_ptr), _p(w_col), _p(w_val), _p(dense_idx), _p(dense_val), _p(row_hdr), _p(col_rank),
        top_k, int(filter_interacted), mode, int(acc_f64), _p(ids), _p(scores), _p(scores64), _p(aux), _p(count),
        _p(ws), ws.numel(), _stream(ids), C.byref(opts)), "rtrec_slim_score_topk_opt")


@custom_op("rtrec_amd::score_rows", mutates_args=("out",), device_types="cuda")
def score_rows(row_ids: Optional[Tensor], xb_ptr: Tensor, xb_col: Tensor, xb_val: Tensor, n_rows: int,
               n_items: int, n_cols: int, col_offset: int, tile_cols: int, n_tiles: int,
               tile_ptr: Tensor, w_col: Tensor, w_val: Tensor, acc_f64: bool, out: Tensor) -> None:
    lib = _native.load()
    _native.check(lib.rtrec_slim_score_rows(
        n_rows, _p(row_ids), _p(xb_ptr), _p(xb_col), _p(xb_val), n_items, n_cols, col_offset, tile_cols, n_tiles,
        _p(tile_ptr), _p(w_col), _p(w_val), int(acc_f64), _p(out), int(out.stride(0)), _stream(out)),
        "rtrec_slim_score_rows")


@custom_op("rtrec_amd::merge_topk", mutates_args=("out_ids", "out_scores", "out_count"), device_types="cuda")
def merge_topk(in_ids: Tensor, in_scores: Tensor, in_scores64: Optional[Tensor], in_aux: Tensor, in_count: Tensor,
               top_k: int, out_ids: Tensor, out_scores: Tensor, out_count: Tensor) -> None:
    """in_* are [n_lists, n_rows, top_k] (in_count [n_lists, n_rows]); strided views into one packed
    all-gather buffer are fine as long as the last dimension is contiguous."""
    lib = _native.load()
    n_lists, n_rows = int(in_ids.shape[0]), int(in_ids.shape[1])
    assert in_ids.stride(2) == 1 and in_ids.stride() == in_scores.stride() == in_aux.stride()
    s64 = in_scores64.stride() if in_scores64 is not None else (0, 0, 1)
    assert s64[2] == 1
    _native.check(lib.rtrec_slim_merge_topk_strided(
        n_rows, n_lists, top_k, _p(in_ids), _p(in_scores), _p(in_scores64), _p(in_aux), _p(in_count),
        in_ids.stride(0), in_ids.stride(1), s64[0], s64[1], in_count.stride(0), in_count.stride(1),
        _p(out_ids), _p(out_scores), _p(out_count), _stream(out_ids)), "rtrec_slim_merge_topk_strided")


@custom_op("rtrec_amd::similar_topk", mutates_args=("ids", "scores", "count"), device_types="cuda")
def similar_topk(queries: Tensor, wc_ptr: Tensor, wc_row: Tensor, wc_val: Tensor, top_k: int,
                 ids: Tensor, scores: Tensor, count: Tensor) -> None:
    lib = _native.load()
    _native.check(lib.rtrec_slim_similar_topk(int(queries.shape[0]), _p(queries), _p(wc_ptr), _p(wc_row), _p(wc_val),
                                              top_k, _p(ids), _p(scores), _p(count), _stream(ids)),
                  "rtrec_slim_similar_topk")


OPS = ["column_sqnorms", "fit_workspace_init", "gram_matrix", "fit_columns", "score_topk", "score_rows", "merge_topk",
       "similar_topk"]
