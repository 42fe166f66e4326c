"""PyTorch-ROCm custom ops over the C-ABI (`torch.ops.rtrec_amd.*`), registered from C++.

One op per entry point of include/rtrec_amd.h, in out-variant style: every buffer is a ROCm tensor owned by the caller,
outputs and scratch are marked mutable in the schema, scalars are plain ints / floats / bools, and the work is enqueued on
the current stream of the tensors' device.  The ops only marshal pointers -- all computation is in librtrec_amd.so -- so
they compose with torch streams, the caching allocator and torch.distributed without copies.  rtrec_amd.engine.HipBackend
is written on top of them; they are also the interface for callers that already hold device tensors.

    torch.ops.rtrec_amd.column_sqnorms      norm_cols_X of the coordinate descent
    torch.ops.rtrec_amd.fit_workspace_init  scratch invariants of the fit kernel
    torch.ops.rtrec_amd.gram_matrix         Gram matrix of the popular items (Gram tracking)
    torch.ops.rtrec_amd.fit_columns         X^T y feature selection + elastic-net CD per item column
    torch.ops.rtrec_amd.score_topk          user rows x W shard, interacted filter, top-k
    torch.ops.rtrec_amd.score_rows          score vectors (predict*)
    torch.ops.rtrec_amd.merge_topk          per-tile / per-GPU top-k lists -> top-k
    torch.ops.rtrec_amd.similar_topk        similar_items
    torch.ops.rtrec_amd.store_decay_device  time decay of a resident store      .store_fold_device   bulk ingest: fold per (user, item)
    torch.ops.rtrec_amd.fit_sgd_epochs      optim="sgd" epochs                  .first_touch_aux     cross-shard tie key
    torch.ops.rtrec_amd.dense_fill          DENSE lists completed in place      .refine_topk_f64     float64 refine of a fast pass
    torch.ops.rtrec_amd.score_candidates    request-sized CANDIDATES calls      .seg_plan / .seg_fill  segment layout of a general W
    torch.ops.rtrec_amd.ordered_sums        left-to-right float32 sums (the fit kernels' fold, for tests and tools)

The registration lives in csrc/torch_ops.cpp (TORCH_LIBRARY / TORCH_LIBRARY_IMPL: librtrec_amd_ops.so, built by
rtrec_amd/build.py with the host compiler); importing this module loads it and binds it to the C-ABI library
(`rtrec_ops_bind`: the build named by RTREC_AMD_LIB, or the in-tree one).  Round 2 registered the same schemas as Python
`torch.library.custom_op` bodies that called ctypes: a call now goes dispatcher -> C++ -> C-ABI without re-entering Python.
There is no fallback: without the ops library the import fails.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _native
from . import build as _build

OPS = ["column_sqnorms", "fit_workspace_init", "gram_matrix", "fit_columns", "score_topk", "score_rows", "merge_topk",
       "similar_topk", "store_decay_device", "store_fold_device", "fit_sgd_epochs", "first_touch_aux", "dense_fill",
       "refine_topk_f64", "score_candidates", "seg_plan", "seg_fill", "ordered_sums"]

# C-ABI export each op launches (tests: every kernel-launching export of include/rtrec_amd.h is behind an op)
EXPORT_OF = {"column_sqnorms": "rtrec_slim_column_sqnorms", "fit_workspace_init": "rtrec_slim_fit_workspace_init",
             "gram_matrix": "rtrec_slim_gram_matrix", "fit_columns": "rtrec_slim_fit_columns_opt",
             "score_topk": "rtrec_slim_score_topk_opt", "score_rows": "rtrec_slim_score_rows",
             "merge_topk": "rtrec_slim_merge_topk_strided", "similar_topk": "rtrec_slim_similar_topk",
             "store_decay_device": "rtrec_store_decay_device", "store_fold_device": "rtrec_store_fold_device",
             "fit_sgd_epochs": "rtrec_slim_fit_sgd_epochs", "first_touch_aux": "rtrec_slim_first_touch_aux",
             "dense_fill": "rtrec_slim_dense_fill", "refine_topk_f64": "rtrec_slim_refine_topk_f64",
             "score_candidates": "rtrec_slim_score_candidates", "seg_plan": "rtrec_slim_seg_plan", "seg_fill": "rtrec_slim_seg_fill",
             "ordered_sums": "rtrec_slim_ordered_sums"}


def _load() -> None:
    path = _build.OPS_PATH
    if not os.path.exists(path):
        raise _native.NativeLibraryError(
            f"{path} is missing: build it with `python -m rtrec_amd.build` (host compiler + torch headers). "
            "rtrec_amd has no Python fallback for its custom ops.")
    _native.load()                       # torch's HIP runtime first, then the C-ABI library
    torch.ops.load_library(path)
    lib = C.CDLL(path)
    lib.rtrec_ops_bind.restype = C.c_int
    lib.rtrec_ops_bind.argtypes = [C.c_char_p]
    rc = lib.rtrec_ops_bind(_native.lib_path().encode())
    if rc != 0:
        raise _native.NativeLibraryError(f"rtrec_ops_bind({_native.lib_path()}) failed ({rc}): the C-ABI library does not load "
                                         "or misses an entry point")


_load()
