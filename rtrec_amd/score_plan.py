"""Which kernels serve one local scoring call -- the decision alone, as pure functions.

`SlimEngine._local_topk_impl` used to derive its path from a cascade of a dozen interacting booleans (dense_fast, dense_fill,
f64_fast, f64_signed, use_fr, use_sg, lazy_tiled, ...); round 3's high-severity bug lived in that cascade.  The decision is
now two pure steps with the engine's side effects (building a layout, launching kernels) between and after them:

    req  = plan_fast_layout(facts)                  # does the call want a fast (feature-row / segment) form, and which
    lay  = <engine builds / fetches that layout>
    plan = choose_path(facts, req, has_fr, has_sg, tiled_exists)

`facts` holds everything the decision reads.  The four data-dependent facts (are W's weights / X's ratings positive ...)
cost a device reduction when asked for the first time, so they are passed as zero-argument callables and only called where
the old short-circuit evaluation called them.  tests/test_host_logic.py checks both functions over the full cross product
of their inputs against an explicit table and a set of invariants -- on the CPU, without a GPU.

Semantics being served: rtrec/models/internal/slim_elastic.py:782-818 (SPARSE: int ids, only stored products compete),
:744-779 (DENSE: string ids, every column competes), :661-672 / :722-739 (CANDIDATES), float64 scores when W came from the
serial fit (:252).
"""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum
from typing import Callable

TOPK_SPARSE, TOPK_DENSE, TOPK_CANDIDATES = 0, 1, 2          # rtrec_topk_mode of include/rtrec_amd.h


class Path(Enum):
    FAST = "fast"            # float32 fast pass (feature rows / segments) + the tiled kernel for the rows it flags
    FAST_F64 = "fast+f64"    # float32 fast pass for top_k + 1 columns + rtrec_slim_refine_topk_f64 + flagged rows
    TILED = "tiled"          # the tiled-CSR kernel (float32 or float64 accumulators) for every row


@dataclass(frozen=True)
class Limits:
    fr_small_batch: int = 513     # SlimEngine.FR_SMALL_BATCH
    fr_min_rows: int = 1          # SlimEngine.FR_MIN_ROWS
    fr_max_top_k: int = 15        # kFrMaxKk - 1 of csrc/score.hip
    sg_max_top_k: int = 63        # kSgMaxKk - 1 of csrc/score_seg.hip.h
    dense_fill_max_top_k: int = 63


@dataclass(frozen=True)
class ScoreFacts:
    mode: int
    hip: bool                     # the HIP backend (the CPU stand-in of the tests has the tiled form only)
    acc_f64: bool                 # W is float64 on the host: scores accumulate in float64
    top_k: int
    n_rows: int
    full_range: bool              # this rank holds every column of W
    nonempty_shard: bool          # col_hi > col_lo
    # engine switches (settings.py)
    dense_fast_on: bool
    dense_fill_on: bool
    lazy_tiled: bool
    feature_rows_on: bool
    seg_layout_on: bool
    seg_supported: bool
    # data-dependent facts, evaluated on demand
    dense_fill_ok: Callable[[], bool]      # weights and ratings all positive normal float32 products (rtrec_slim_dense_fill)
    f64_w_ok: Callable[[], bool]           # float64 W: float32-valued, positive weights
    f64_x_ok: Callable[[], bool]           # ratings positive
    f64_refine_mode: Callable[[], int]     # 0: no refine step, 1: positive W, 2: signed W (absolute slack)
    limits: Limits = Limits()


@dataclass(frozen=True)
class FastRequest:
    want: bool          # ask the engine for a fast layout at all
    small: bool         # ... the request-sized (segment) form rather than the bulk one
    k_need: int         # columns the fast pass must deliver (top_k + 1 ahead of a float64 refine step)
    dense_fast: bool    # DENSE mode through the fast pass
    dense_fill: bool    # ... with short lists completed in place (what lets a column shard take it)


@dataclass(frozen=True)
class Plan:
    path: Path
    kernel: str         # "feature_rows" | "segments" | "tiled": the kernel of the first (or only) pass
    use_fr: bool
    use_sg: bool
    k_fast: int         # list length of the fast pass
    f64_signed: bool    # FAST_F64 with the absolute per-user slack (signed W or ratings)
    fill: bool          # FAST, DENSE mode: rtrec_slim_dense_fill after the fast pass
    lazy: bool          # FAST: the tiled layout is built only if a row is flagged


def plan_fast_layout(f: ScoreFacts) -> FastRequest:
    L = f.limits
    sparse, dense = f.mode == TOPK_SPARSE, f.mode == TOPK_DENSE
    dense_fill = bool(dense and f.hip and f.dense_fast_on and f.dense_fill_on and f.lazy_tiled and not f.acc_f64
                      and f.top_k <= L.dense_fill_max_top_k and f.dense_fill_ok())
    dense_fast = bool(dense and f.hip and f.dense_fast_on and f.lazy_tiled and (not f.acc_f64 or f.f64_w_ok())
                      and (f.full_range or dense_fill))
    want = bool((sparse or dense_fast) and f.hip and f.nonempty_shard)
    k_need, small = f.top_k, False
    if want:
        if f.acc_f64 and (f.f64_w_ok() or (sparse and f.f64_refine_mode() == 2)):
            k_need = f.top_k + 1            # a float64 W asks its fast pass for one column more (the refine step's margin)
        small = bool((f.n_rows < L.fr_small_batch or k_need > L.fr_max_top_k) and f.seg_layout_on and k_need <= L.sg_max_top_k
                     and f.seg_supported)
    return FastRequest(want, small, k_need, dense_fast, dense_fill)


def choose_path(f: ScoreFacts, req: FastRequest, has_fr: bool, has_sg: bool, tiled_exists: bool) -> Plan:
    """has_fr / has_sg: the fast layout the engine obtained for `req` holds a feature-row / a segment form (both False when
    it obtained none); tiled_exists: the tiled layout of this (mode, tile width) is already built."""
    L = f.limits
    sparse = f.mode == TOPK_SPARSE
    have_fast = req.want and (has_fr or has_sg)
    # float64 W: fast pass + refine when W and X are positive; signed weights or ratings (SPARSE only): the same with an
    # absolute per-user slack; otherwise the float64 tiled kernel
    f64_fast = bool((sparse or req.dense_fast) and f.hip and f.acc_f64 and have_fast and f.lazy_tiled and f.f64_w_ok() and f.f64_x_ok())
    f64_signed = bool(not f64_fast and sparse and f.hip and f.acc_f64 and have_fast and f.lazy_tiled and f.f64_refine_mode() != 0)
    f64_fast = f64_fast or f64_signed
    if f.acc_f64 and not f64_fast:
        have_fast = False
    k_fast = f.top_k + 1 if f64_fast else f.top_k
    use_fr = bool(have_fast and f.feature_rows_on and has_fr and f.n_rows >= L.fr_min_rows and k_fast <= L.fr_max_top_k)
    use_sg = bool(have_fast and not use_fr and f.seg_layout_on and has_sg and k_fast <= L.sg_max_top_k)
    kernel = "feature_rows" if use_fr else ("segments" if use_sg else "tiled")
    if f64_fast and (use_fr or use_sg):
        return Plan(Path.FAST_F64, kernel, use_fr, use_sg, k_fast, f64_signed, False, True)
    if f.hip and (use_fr or use_sg) and f.lazy_tiled and (req.dense_fast or not tiled_exists):
        return Plan(Path.FAST, kernel, use_fr, use_sg, k_fast, False, bool(req.dense_fast and req.dense_fill), True)
    # the tiled layout exists (or laziness is off): one launch -- the fast kernel with the tiled layout beside it for its
    # exact-tie pass, or the tiled kernel alone
    if not (use_fr or use_sg):
        return Plan(Path.TILED, "tiled", False, False, f.top_k, False, False, False)
    return Plan(Path.FAST, kernel, use_fr, use_sg, k_fast, False, False, False)
