"""ctypes binding of the C-ABI in include/rtrec_amd.h (librtrec_amd.so).

There is no CPU fallback: if the HIP library is missing or does not load, every product
entry point raises.  Device buffers are torch tensors (plumbing only); the functions take
their raw device pointers and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build
from . import settings

_lib: Optional[C.CDLL] = None

RTREC_OK = 0
TOPK_SPARSE, TOPK_DENSE, TOPK_CANDIDATES = 0, 1, 2
_STATUS = {0: "ok", -1: "invalid argument", -2: "unsupported parameter", -3: "workspace too small",
           -4: "kernel launch failed"}

EXPORTS = [
    "rtrec_amd_version", "rtrec_amd_last_error", "rtrec_timer_create", "rtrec_timer_read", "rtrec_timer_destroy",
    "rtrec_slim_score_fr_scratch_bytes", "rtrec_slim_score_sg_scratch_bytes", "rtrec_slim_score_topk_opt", "rtrec_slim_column_sqnorms", "rtrec_slim_fit_workspace_bytes",
    "rtrec_slim_fit_workspace_init", "rtrec_slim_fit_columns", "rtrec_slim_fit_columns_opt", "rtrec_slim_gram_workspace_bytes", "rtrec_slim_xty_workspace_bytes", "rtrec_slim_gram_matrix", "rtrec_slim_score_workspace_bytes",
    "rtrec_slim_score_topk", "rtrec_slim_score_rows", "rtrec_slim_merge_topk", "rtrec_slim_merge_topk_strided", "rtrec_slim_similar_topk",
    "rtrec_store_merge_sorted", "rtrec_store_find_sorted", "rtrec_lru_replay", "rtrec_store_apply_round", "rtrec_store_decay",
    "rtrec_store_decay_device",
    "rtrec_store_fold_device",
    "rtrec_slim_seg_plan_workspace_bytes",
    "rtrec_slim_seg_fill_workspace_bytes",
    "rtrec_slim_seg_plan",
    "rtrec_slim_seg_fill",
    "rtrec_slim_refine_topk_f64",
    "rtrec_slim_score_candidates",
    "rtrec_slim_sgd_schedule",
    "rtrec_slim_fit_sgd_epochs",
    "rtrec_slim_dense_fill",
    "rtrec_slim_first_touch_aux",
    "rtrec_slim_ordered_sums",
]


class FitCfg(C.Structure):
    _fields_ = [("l1_reg", C.c_float), ("l2_reg", C.c_float), ("tol", C.c_float), ("max_iter", C.c_int32),
                ("seed", C.c_uint32), ("positive", C.c_int32), ("top_features", C.c_int32)]


class FitOpts(C.Structure):
    _fields_ = [("d_trace", C.c_void_p), ("d_gram", C.c_void_p), ("d_gram_index", C.c_void_p),
                ("gram_n", C.c_int32), ("gram_rel_err", C.c_double), ("fast", C.c_int32), ("kernel", C.c_int32),
                ("colwalk_min_rows", C.c_int32), ("screen_min", C.c_int32), ("lane_max", C.c_int32),
                ("d_xty_ws", C.c_void_p), ("xty_ws_bytes", C.c_size_t), ("nnz", C.c_int64), ("d_col_order", C.c_void_p),
                ("fold", C.c_int32)]


class ScoreOpts(C.Structure):
    _fields_ = [("n_x_rows", C.c_int32), ("d_fr_map", C.c_void_p), ("d_fr_col_ids", C.c_void_p),
                ("d_fr_col_map", C.c_void_p), ("d_fr_w", C.c_void_p), ("d_fr_tile_rows", C.c_void_p),
                ("d_fr_tile_off", C.c_void_p), ("d_fr_super_kb", C.c_void_p), ("d_fr_super_tile", C.c_void_p),
                ("d_fr_frag_tile", C.c_void_p),
                ("fr_rows", C.c_int32), ("fr_tile_cols", C.c_int32), ("fr_n_tiles", C.c_int32), ("fr_n_frags", C.c_int32),
                ("fr_n_super", C.c_int32), ("fr_buf_bytes", C.c_int32), ("d_fr_scratch", C.c_void_p), ("fr_scratch_bytes", C.c_size_t),
                ("d_row_order", C.c_void_p), ("timer", C.c_void_p), ("diagnostics", C.c_int32), ("d_rescored", C.c_void_p), ("row_order_grouped", C.c_int32),
                ("d_sg_info", C.c_void_p), ("d_sg_ptr", C.c_void_p), ("d_sg_ent", C.c_void_p), ("sg_nnz", C.c_int64),
                ("d_sg_bound", C.c_void_p), ("d_sg_col_ids", C.c_void_p),
                ("sg_tile_cols", C.c_int32), ("sg_n_tiles", C.c_int32), ("sg_rows", C.c_int32), ("sg_n_cols", C.c_int32),
                ("d_sg_trow_ptr", C.c_void_p), ("d_sg_trow", C.c_void_p), ("d_sg_scratch", C.c_void_p), ("sg_scratch_bytes", C.c_size_t),
                ("row_order_longest_first", C.c_int32), ("d_flagged", C.c_void_p), ("aux_stream", C.c_void_p)]


class NativeLibraryError(RuntimeError):
    pass


def lib_path() -> str:
    # RTREC_AMD_LIB: load another build of the same ABI (A/B timing of kernel variants, tools/ab_build.sh)
    return settings.raw("RTREC_AMD_LIB") or _build.LIB_PATH


def load() -> C.CDLL:
    """Load librtrec_amd.so (never builds implicitly on the hot path; see rtrec_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: librtrec_amd.so must bind to the HIP runtime torch ships and initialises
    # (device pointers and streams come from it); loading ours first pulls in a second runtime.
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} is missing: build it with `python -m rtrec_amd.build` (hipcc, gfx950). "
            "rtrec_amd has no CPU fallback.")
    try:
        L = C.CDLL(path)
    except OSError as e:  # pragma: no cover - depends on the host
        raise NativeLibraryError(f"cannot load {path}: {e}") from e
    vp, i32, u64 = C.c_void_p, C.c_int32, C.c_size_t
    L.rtrec_amd_version.restype = C.c_char_p
    L.rtrec_amd_version.argtypes = []
    L.rtrec_amd_last_error.restype = C.c_char_p
    L.rtrec_amd_last_error.argtypes = []
    L.rtrec_timer_create.restype = C.c_int
    L.rtrec_timer_create.argtypes = [C.POINTER(C.c_void_p)]
    L.rtrec_timer_read.restype = C.c_int
    L.rtrec_timer_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), i32]
    L.rtrec_timer_destroy.restype = None
    L.rtrec_timer_destroy.argtypes = [vp]
    L.rtrec_slim_score_fr_scratch_bytes.restype = u64
    L.rtrec_slim_score_fr_scratch_bytes.argtypes = [i32, i32]
    L.rtrec_slim_score_sg_scratch_bytes.restype = u64
    L.rtrec_slim_score_sg_scratch_bytes.argtypes = [i32, i32, i32]
    L.rtrec_slim_column_sqnorms.restype = C.c_int
    L.rtrec_slim_column_sqnorms.argtypes = [i32, vp, vp, vp, vp]
    L.rtrec_slim_fit_workspace_bytes.restype = u64
    L.rtrec_slim_fit_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.rtrec_slim_fit_workspace_init.restype = C.c_int
    L.rtrec_slim_fit_workspace_init.argtypes = [vp, u64, i32, i32, i32, i32, vp]
    L.rtrec_slim_fit_columns.restype = C.c_int
    L.rtrec_slim_fit_columns.argtypes = [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, C.POINTER(FitCfg),
                                         vp, vp, vp, vp, i32, vp, u64, i32, vp, vp]
    L.rtrec_slim_fit_columns_opt.restype = C.c_int
    L.rtrec_slim_fit_columns_opt.argtypes = L.rtrec_slim_fit_columns.argtypes + [C.POINTER(FitOpts)]
    L.rtrec_slim_gram_workspace_bytes.restype = u64
    L.rtrec_slim_gram_workspace_bytes.argtypes = [i32, i32]
    L.rtrec_slim_xty_workspace_bytes.restype = u64
    L.rtrec_slim_xty_workspace_bytes.argtypes = [i32, i32, C.c_int64, i32]
    L.rtrec_slim_gram_matrix.restype = C.c_int
    L.rtrec_slim_gram_matrix.argtypes = [i32, i32, vp, vp, vp, vp, i32, vp, u64, vp, vp]
    L.rtrec_slim_score_workspace_bytes.restype = u64
    L.rtrec_slim_score_workspace_bytes.argtypes = [i32, i32, i32]
    L.rtrec_slim_score_topk.restype = C.c_int
    L.rtrec_slim_score_topk.argtypes = ([i32] + [vp] * 4 + [i32] * 3 + [vp] * 2 + [i32] * 2 + [vp] * 7
                                        + [i32] * 4 + [vp] * 5 + [vp, u64, vp])
    L.rtrec_slim_score_topk_opt.restype = C.c_int
    L.rtrec_slim_score_topk_opt.argtypes = L.rtrec_slim_score_topk.argtypes + [C.POINTER(ScoreOpts)]
    L.rtrec_slim_score_rows.restype = C.c_int
    L.rtrec_slim_score_rows.argtypes = [i32] + [vp] * 4 + [i32] * 5 + [vp] * 3 + [i32, vp, C.c_int64, vp]
    L.rtrec_slim_merge_topk.restype = C.c_int
    L.rtrec_slim_merge_topk.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rtrec_slim_merge_topk_strided.restype = C.c_int
    L.rtrec_slim_merge_topk_strided.argtypes = [i32, i32, i32, vp, vp, vp, vp, vp] + [C.c_int64] * 6 + [vp, vp, vp, vp]
    L.rtrec_slim_similar_topk.restype = C.c_int
    L.rtrec_slim_similar_topk.argtypes = [i32, vp, vp, vp, vp, i32, vp, vp, vp, vp]
    L.rtrec_store_merge_sorted.restype = C.c_int64
    L.rtrec_store_merge_sorted.argtypes = [vp, vp, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, vp, i32]
    L.rtrec_store_find_sorted.restype = C.c_int
    L.rtrec_store_find_sorted.argtypes = [vp, C.c_int64, vp, C.c_int64, vp, vp, i32]
    L.rtrec_lru_replay.restype = C.c_int64
    L.rtrec_lru_replay.argtypes = [vp, vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int64, vp, vp]
    L.rtrec_store_apply_round.restype = C.c_int
    L.rtrec_store_apply_round.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_double, C.c_double, vp, vp, i32]
    L.rtrec_store_decay.restype = C.c_int
    L.rtrec_store_decay.argtypes = [vp, vp, C.c_int64, C.c_double, vp, C.c_double, vp, vp, i32]
    L.rtrec_store_decay_device.restype = C.c_int
    L.rtrec_store_decay_device.argtypes = [vp, vp, C.c_int64, C.c_double, C.c_double, vp, vp, vp, i32, vp]
    L.rtrec_slim_seg_plan_workspace_bytes.restype = C.c_size_t
    L.rtrec_slim_seg_plan_workspace_bytes.argtypes = [i32]
    L.rtrec_slim_seg_fill_workspace_bytes.restype = C.c_size_t
    L.rtrec_slim_seg_fill_workspace_bytes.argtypes = [i32, C.c_int64, i32, i32]
    L.rtrec_slim_seg_plan.restype = C.c_int
    L.rtrec_slim_seg_plan.argtypes = [i32, C.c_int64, vp, vp, i32, i32, vp, vp, C.c_size_t, vp, vp]
    L.rtrec_slim_seg_fill.restype = C.c_int
    L.rtrec_slim_seg_fill.argtypes = [i32, C.c_int64, vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, vp, C.c_size_t,
                                      vp, vp, vp, C.c_int64, vp, vp, vp, vp, C.c_int64, vp]
    L.rtrec_slim_refine_topk_f64.restype = C.c_int
    L.rtrec_slim_refine_topk_f64.argtypes = [i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32, vp, vp, vp, C.c_double, vp,
                                             vp, vp, vp, vp, vp, vp]
    L.rtrec_slim_first_touch_aux.restype = C.c_int
    L.rtrec_slim_first_touch_aux.argtypes = [i32, vp, vp, vp, i32, i32, vp, vp, i32, vp, vp, vp, vp]
    L.rtrec_slim_dense_fill.restype = C.c_int
    L.rtrec_slim_dense_fill.argtypes = [i32, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    L.rtrec_slim_score_candidates.restype = C.c_int
    L.rtrec_slim_score_candidates.argtypes = [i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp]
    L.rtrec_slim_sgd_schedule.restype = C.c_int
    L.rtrec_slim_sgd_schedule.argtypes = [i32, i32, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp,
                                          vp, vp, vp, vp, vp, vp, vp, i32, vp]
    L.rtrec_slim_fit_sgd_epochs.restype = C.c_int
    L.rtrec_slim_fit_sgd_epochs.argtypes = [i32, i32, vp, vp, vp, C.c_int64, vp, i32, vp, vp, i32, i32, i32, i32, C.c_double,
                                            vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.rtrec_store_fold_device.restype = C.c_int
    L.rtrec_store_fold_device.argtypes = [vp, vp, C.c_int64, vp, vp, vp, C.c_double, C.c_double, i32, vp, vp, vp, vp]
    L.rtrec_slim_ordered_sums.restype = C.c_int
    L.rtrec_slim_ordered_sums.argtypes = [vp, vp, i32, i32, vp, vp]
    _lib = L
    return L


def check(status: int, what: str) -> None:
    if status != RTREC_OK:
        detail = load().rtrec_amd_last_error().decode() if status == -4 else ""
        raise NativeLibraryError(f"{what} failed: {_STATUS.get(status, status)} ({status}) {detail}")


def version() -> str:
    return load().rtrec_amd_version().decode()
