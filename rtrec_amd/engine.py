"""Device engine: owns the HBM-resident matrices and drives the HIP kernels through the C-ABI.

Data layout in HBM (per GPU)
  X   : the U x I interaction matrix, replicated on every GPU, in BOTH orientations
        (CSC for the coordinate-descent column sweeps, CSR for X^T y and for scoring);
        int32 indices, float32 values -- 16 B per interaction in total.
  W   : this GPU's column shard [col_lo, col_hi) of the I x I item-item matrix, in the
        column-tiled CSR layout described in include/rtrec_amd.h (6 B per stored weight).
Nothing here computes on the host: numpy is used only to marshal index arrays.

Multi-GPU (one process per GPU, torch.distributed / RCCL): fit has no collective (each rank
fits the target columns it owns), scoring ends with one all-gather of the per-shard top-k
followed by the merge kernel.
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from . import _native
from . import settings
# (round 4: the layout builders and the backend live in modules of their own; the names stay importable from here)
from .backend import DeviceWeights, HipBackend, FIT_MW_MAX_TARGETS, XTY_SCRATCH_MAX_BYTES  # noqa: F401
from . import score_plan
from .layouts import (TiledW, build_tiled_w, row_header_table, build_feature_rows, build_feature_rows_device,  # noqa: F401
                      build_tiled_w_device, _heavy_tiles_first, _pack_fragments, FR_MAX_ROWS, FR_MIN_FILL,
                      FR_TILE_HEADER_BYTES, FR_STREAM_BUF_BYTES)

DEFAULT_TILE_COLS = 4096   # 16 KB of float accumulators: 8 persistent waves per CU (2 per SIMD) hide each other's latency
DENSE_ROW_FILL = 1.0 / 3.0   # W row segments at least this full are stored dense (sparse layout)
MAX_SLOTS = 4096   # 4 single-wave workgroups per SIMD; with Gram tracking the sweet spot moved down from 5120 (C3: 2.16 s vs 2.3 s)
GATHER_CHUNK_ROWS = 131072  # rows per exchange chunk of a column-sharded scoring call: a launch of the score kernel needs this
                            # many users to fill the chip a few times over (32k-row chunks: 6.4 ms per ML-20M pass against 2.6)
MAX_GATHER_CHUNKS = 8
ROW_CHUNK_ROWS = 49152     # least slots per exchange chunk of a row-sharded scoring call (each rank's launch per chunk): C4 at 8
                           # ranks (125k slots per rank) runs as two chunks, the gather of the first beside the kernel of the second
ALLF_OUTPUT_CAP = 2048      # coefficients per target the K=None output block holds before a refit with cap = I
GRAM_ITEMS = 512            # most popular items whose pairwise dot products the fit kernel may look up
FIT_SCRATCH_GIB = 16.0      # total per-slot scratch of a bulk fit is kept near this (see fit_columns)
XTY_MIN_WALK_ENTRIES = 1e9         # ... and the per-target column walks it replaces would visit at least this many entries
FIT_HEAVY_TARGETS = 256     # head of a bulk call sent to the multi-wave kernel (one workgroup per CU)
FIT_HEAVY_SLOTS = 256
FIT_HEAVY_MIN_ROWS = 2048   # ... as long as the target has at least this many users


def sklearn_seed(random_state: Optional[int]) -> int:
    """The xorshift seed sklearn draws per ElasticNet.fit: check_random_state(rs).randint(0, 2**31-1)
    (sklearn/linear_model/_cd_fast.pyx:367).  random_state=43 -> 494155588."""
    return int(np.random.RandomState(random_state).randint(0, 2147483647))


def shard_bounds(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Item-column shard [lo, hi) owned by `rank` (contiguous blocks of ceil(I / G) columns)."""
    per = -(-n_items // world_size) if n_items > 0 else 0
    lo = min(n_items, rank * per)
    hi = min(n_items, lo + per)
    return lo, hi


def spread_giant_rows(torch, order, lens, max_giants: int, giant_len: int):
    """The pattern-grouped work order with its giant rows spread out (SlimEngine._row_order).  A wave of the feature-row kernel
    takes CONSECUTIVE positions of that order (2, 4 or 8) and sets its users up one after the other; the few giant rows (tens of
    thousands of entries: they rate every feature item, so their patterns are neighbours at the head) must not share a wave --
    four of them in one wave made one of eight C4 row shards 0.81 ms instead of 0.59 (tools/row_slice_probe.py).  Up to
    `max_giants` rows of more than `giant_len` entries (`lens`: entries per row, indexed like the values of `order`) go to
    positions 0, 8, 16, ... of the head, longest first; the gaps are filled from the tail of the order (its lightest patterns);
    everything else keeps its place.  Returns a permutation of the same rows."""
    n_rows = int(order.numel())
    n_g = min(int(max_giants), n_rows // 64)
    if n_g <= 0:
        return order
    g_len, g_idx = torch.topk(lens, n_g)
    g_idx = g_idx[g_len > giant_len]
    n_g = int(g_idx.numel())
    if n_g == 0:
        return order
    is_g = torch.zeros(n_rows, dtype=torch.bool, device=order.device)
    is_g[g_idx] = True
    rest = order[~is_g[order]]
    n_fill = 7 * n_g
    head = torch.empty(8 * n_g, dtype=order.dtype, device=order.device)
    gap = torch.ones(8 * n_g, dtype=torch.bool, device=order.device)
    gap[0::8] = False
    head[0::8] = g_idx.to(order.dtype)
    head[gap] = rest[rest.numel() - n_fill:]
    return torch.cat([head, rest[:rest.numel() - n_fill]])


class SlimEngine:
    """Fit / score / similar-items on one GPU (one shard of W)."""

    _process_warm: set = set()          # devices whose code objects this process has loaded (see _process_warm_up)

    def __init__(self, device: Any = None, rank: int = 0, world_size: int = 1, process_group: Any = None,
                 tile_cols: Optional[int] = None, backend: Any = None, score_shard: Optional[str] = None,
                 warm_up: bool = True, shard_w: Optional[bool] = None):
        self.rank, self.world_size, self.group = rank, world_size, process_group
        # How a multi-GPU scoring pass is divided.  "columns" (default; BASELINE.json's configuration): every
        # rank scores all users against its item-column shard of W, lists are exchanged and merged.
        # "rows": W is replicated (it is assembled on every rank after a fit anyway), every rank scores its
        # 1/G slice of the users against all of W and only the final lists are all-gathered -- for
        # catalogues whose W is so small that a pass is bound by reading the user rows (C4: 3.2k active
        # columns), which column sharding does not divide.
        self.score_shard = (score_shard or settings.raw("RTREC_AMD_SCORE_SHARD", "columns")).lower()
        if self.score_shard not in ("columns", "rows"):
            raise ValueError(f"score_shard must be 'columns' or 'rows': {self.score_shard}")
        # shard_w (round 4; BASELINE.json: "W shards by item column across the 8 GPUs"): every rank FITS the item columns of its
        # own scoring shard [lo, hi) and keeps only those -- the fit output is the score input and W never moves.
        # Without it (the default) the targets are dealt out by column length and the coefficient triples are all-gathered, so
        # that every rank holds all of W: right while W is a few MB, not a design for a W that is not degenerate (500k
        # items x K=50 = 200 MB per copy and 8x the merge work).  Column-sharded scoring only; `item_similarity`,
        # `similar_items` and pickling gather on demand (collective calls: every rank must make them).
        if shard_w is None:
            shard_w = settings.raw("RTREC_AMD_SHARD_W", "0") == "1"
        self.shard_w = bool(shard_w) and world_size > 1 and self.score_shard == "columns"
        # testing aid: run the multi-GPU exchange (collectives, strided merge) even with a single rank, so that
        # the RCCL code path can be exercised on a one-GPU box through a 1-rank process group
        self.force_exchange = settings.raw("RTREC_AMD_FORCE_EXCHANGE") == "1"
        self.tile_cols = int(tile_cols or DEFAULT_TILE_COLS)
        self.be = backend if backend is not None else HipBackend(device)
        self.n_users = 0
        self.n_items = 0
        self._X: Dict[str, Any] = {}
        self._W: Dict[str, Any] = {}
        self._fit_ws: Dict[Tuple[int, int, int, int], Any] = {}
        self._score_ws = None
        self.gather_chunk_rows = GATHER_CHUNK_ROWS
        self.row_chunk_rows = ROW_CHUNK_ROWS
        self.last_fit_stats: Dict[str, Any] = {}
        self.score_timer = 0          # rtrec_timer handle (HipBackend.timer_create) bracketing the dominant score kernel
        self.use_feature_rows = settings.raw("RTREC_AMD_FEATURE_ROWS", "1") != "0"     # A/B switch of the score kernel
        # ablation switches of tools/score_ablate.sh: only a diagnostic build of the library looks at them
        self.diagnostics = int(settings.raw("RTREC_AMD_ABLATE", "0")) & 0xff
        self.fr_users_per_wave = int(settings.raw("RTREC_AMD_FR_USERS", "0"))      # 8 / 4 / 2: force the feature-row kernel's form
        self.native_seg_builder = settings.raw("RTREC_AMD_NATIVE_SEG_BUILD", "1") != "0"   # csrc/seg_build.hip (else tensor ops)
        self.FR_SMALL_BATCH = int(settings.raw("RTREC_AMD_FR_SMALL_BATCH", self.FR_SMALL_BATCH))     # A/B: segments for larger passes
        self.f64_refine = settings.raw("RTREC_AMD_F64_REFINE", "1") != "0"      # float64 W: float32 fast pass + float64 refine
        self.cands_direct = settings.raw("RTREC_AMD_CANDS_DIRECT", "1") != "0"   # request-sized CANDIDATES calls: csrc/score_cands.hip
        self.dense_fast = settings.raw("RTREC_AMD_DENSE_FAST", "1") != "0"    # DENSE mode through the fast SPARSE-style pass
        self.dense_fill = settings.raw("RTREC_AMD_DENSE_FILL", "1") != "0"    # ... its short lists completed in place (and so column shards)
        self._sg_scratch = None           # zeroed scratch of the segment path's workgroup-per-user kernel
        self._order_grouped = False       # the work order _row_order handed out last is the pattern-grouped one
        self.sg_heavy_min = int(settings.raw("RTREC_AMD_SG_HEAVY_MIN", "0"))   # v > 0: segment path, users of more than v - 1 items get a workgroup
        self.use_seg_layout = settings.raw("RTREC_AMD_SEG_LAYOUT", "1") != "0"        # A/B switch of the general-W score kernel
        self.seg_cluster = settings.raw("RTREC_AMD_SEG_CLUSTER", "1") != "0"          # ... and of its column clustering
        self.use_seg_heavy = settings.raw("RTREC_AMD_SEG_HEAVY", "1") != "0"          # ... and of its workgroup-per-long-user pass
        self.lazy_tiled = settings.raw("RTREC_AMD_LAZY_TILED", "1") != "0"           # tiled layout only when a call flags exact ties
        self.last_score_path = ""     # which kernel family served the last _local_topk call (tests, bench.py)
        self._sg_labels = None        # (cluster labels of the last segment layout, n_items, nnz of W when they were computed)
        if warm_up and isinstance(self.be, HipBackend) and str(self.be.device) not in SlimEngine._process_warm:
            SlimEngine._process_warm.add(str(self.be.device))
            self._process_warm_up()

    def _process_warm_up(self) -> None:
        """Once per process and device: a toy model through the whole path (fit, write-back, every layout builder, every
        scoring mode) on a throw-away engine.  The first use of a kernel -- ours or one of the tensor ops of the builders --
        loads its code object (tens of milliseconds each: the first W write-back of a process took 120-320 ms, the first
        layouts 60-90 ms); that belongs next to the creation of the HIP context, not inside the first fit or recommend."""
        rng = np.random.default_rng(0)
        U, I = 96, 48
        X = sp.random(U, I, density=0.3, random_state=rng, format="csr", dtype=np.float32)
        X.data[:] = rng.integers(1, 6, X.nnz).astype(np.float32)
        Xc = X.tocsc()
        Xc.sort_indices()
        X.sort_indices()
        toy = SlimEngine(backend=self.be, rank=0, world_size=1, warm_up=False)
        try:
            toy.set_interactions(Xc, X)
            for mode in ("exact", "gram"):
                out = toy.fit_columns(np.arange(I), nn_feature_selection=8, device_out=True, mode=mode)
            dw = toy.merge_fit(None, I, False, *out[:4])
            dw = toy.merge_fit(dw, I, False, *out[:4])                      # the merge into an existing W as well
            rows = np.arange(U)
            for f64 in (False, True):
                toy.set_weights(dw, acc_f64=f64)
                for md in (_native.TOPK_SPARSE, _native.TOPK_DENSE):
                    toy.recommend_rows(rows, top_k=5, mode=md)              # fast layouts, flagged rows -> tiled layouts
                    toy.recommend_rows(rows[:3], top_k=5, mode=md)          # the request-sized path
            toy.set_weights(dw, acc_f64=False)
            toy._layout(True, 5)
            toy._layout(False, 5)
            # the tensor ops of the write-back and the builders pick other kernel configurations for real sizes (radix sort,
            # scans, compaction): once more on a quarter of a million elements
            torch = self.be.torch
            big = torch.arange(1 << 18, dtype=torch.int64, device=self.be.device).flip(0)
            srt, order = torch.sort(big)
            _ = (torch.argsort(big), torch.unique(big % 1000), big[big % 2 == 0], torch.isin(big, big[:100]), torch.cumsum(big, 0),
                 torch.searchsorted(srt, big[:1000]), torch.nonzero(big % 3 == 0), torch.repeat_interleave(big[:1000] % 7),
                 torch.bincount(big % 1000), big.to(torch.int32).to(torch.float32).abs().max(), torch.cat([big, big]),
                 torch.sort(big.to(torch.float32), descending=True), torch.argsort(big % 977, stable=True))
            self.be.synchronize()
        except Exception as e:          # a warm-up must never keep an engine from being constructed -- but it must not hide a
            # broken library either: say so, and surface / clear pending device errors now
            logging.warning(f"rtrec_amd warm-up failed (the first real call will hit the same problem): {e!r}")
            try:
                self.be.synchronize()
            except Exception as e2:
                logging.warning(f"rtrec_amd: device error after the failed warm-up: {e2!r}")

    # ------------------------------------------------------------------------------ X
    def set_interactions(self, X_csc: sp.csc_matrix, X_csr: Optional[sp.csr_matrix] = None,
                         need_csc: bool = True) -> None:
        """Upload X (U x I).  Both orientations must have sorted indices (scipy canonical form)."""
        be = self.be
        if X_csr is None and need_csc and getattr(be, "supports_device_store", False):
            # only the CSC orientation comes from the host: the CSR one is a sort of the keys on the device
            # (utils/device_store.py) instead of a scipy tocsr() here and a second upload
            from .utils.device_store import DeviceInteractions
            if not X_csc.has_sorted_indices:
                X_csc = X_csc.sorted_indices()
            if X_csc.nnz >= 2 ** 31:
                raise ValueError("more than 2**31 interactions per GPU are not supported")
            d = DeviceInteractions(be.torch, be.device)
            d.load_csc(X_csc.indptr, X_csc.indices, X_csc.data, X_csc.shape[0], X_csc.shape[1], 0)
            self.set_interactions_device(d.full(), X_csc.shape[0], X_csc.shape[1])
            return
        if X_csr is None:
            X_csr = X_csc.tocsr()
        if not X_csr.has_sorted_indices:
            X_csr = X_csr.sorted_indices()
        self.n_users, self.n_items = X_csr.shape
        if X_csr.nnz >= 2 ** 31:
            raise ValueError("more than 2**31 interactions per GPU are not supported")
        X: Dict[str, Any] = {}
        X["rptr"] = be.to_dev(np.asarray(X_csr.indptr, dtype=np.int32))
        X["rcol"] = be.to_dev(np.asarray(X_csr.indices, dtype=np.int32))
        X["rval"] = be.to_dev(np.asarray(X_csr.data, dtype=np.float32))
        if need_csc:
            if X_csc is None:
                X_csc = X_csr.tocsc()
            if not X_csc.has_sorted_indices:
                X_csc = X_csc.sorted_indices()
            X["cptr"] = be.to_dev(np.asarray(X_csc.indptr, dtype=np.int32))
            X["crow"] = be.to_dev(np.asarray(X_csc.indices, dtype=np.int32))
            X["cval"] = be.to_dev(np.asarray(X_csc.data, dtype=np.float32))
            X["col_nnz"] = np.diff(np.asarray(X_csc.indptr, dtype=np.int64))
            X["nonneg"] = bool(X_csc.nnz == 0 or float(X_csc.data.min()) >= 0.0)
        self._X = X

    def set_interactions_device(self, X: Dict[str, Any], n_users: int, n_items: int) -> None:
        """Use an X whose arrays are already resident on this device (utils/device_store.py): the same
        keys as set_interactions builds (rptr/rcol/rval, cptr/crow/cval, col_nnz, nonneg)."""
        if int(X["rcol"].shape[0]) >= 2 ** 31:
            raise ValueError("more than 2**31 interactions per GPU are not supported")
        self.n_users, self.n_items = int(n_users), int(n_items)
        self._X = dict(X)          # per-matrix caches (column norms, Gram matrix) attach to this copy

    # ------------------------------------------------------------------------------ fit
    def owned_columns(self, columns: np.ndarray) -> np.ndarray:
        """The part of `columns` (the same list on every rank) THIS rank fits.  Fitting has no
        collective and a popular item costs far more than a rare one, so the targets are dealt out
        by column length: sorted by nnz(X[:, j]) descending and dealt in snake order (ranks 0..G-1,
        then G-1..0, ...) -- every rank gets the same mix of heavy and light targets.  Which
        rank fitted a column is independent of which rank scores it: W is assembled on every rank
        after the fit and re-sharded by contiguous column block for scoring (shard_bounds)."""
        columns = np.asarray(columns, dtype=np.int64)
        if self.world_size == 1:
            return columns
        if self.shard_w:        # the rank that scores a column fits it: W never moves
            lo, hi = shard_bounds(self.n_items, self.world_size, self.rank)
            return np.sort(columns[(columns >= lo) & (columns < hi)])
        nnz = self._X["col_nnz"][columns] if "col_nnz" in self._X else np.zeros(len(columns), dtype=np.int64)
        order = np.lexsort((columns, -nnz))            # nnz descending, ties by id: identical on all ranks
        pos = np.arange(len(columns))
        blk, off = pos // self.world_size, pos % self.world_size
        owner = np.where(blk % 2 == 0, off, self.world_size - 1 - off)
        return np.sort(columns[order[owner == self.rank]])

    def fit_columns(self, targets: Sequence[int], alpha: float = 0.1, l1_ratio: float = 0.1,
                    positive: bool = True, max_iter: int = 100, tol: float = 1e-4, random_state: Optional[int] = 43,
                    nn_feature_selection: Optional[int] = None, n_slots: Optional[int] = None,
                    trace: bool = False, exact: bool = True, mode: Optional[str] = None, device_out: bool = False
                    ) -> Tuple[Any, Any, Any, Any, np.ndarray]:
        """Fit the given target columns on this GPU.  mode: "exact" (default; bit-identical to scikit-learn),
        "gram" (Gram-form coordinate descent where all features are in the Gram matrix, tree-reduced dots elsewhere;
        a few 1e-5 relative; what exact=False selects) or "shuffle" (tree-reduced dots only) -- rtrec_fit_opts.fast.

        Returns (targets_in_processing_order, items[n, cap], coef[n, cap], count[n], n_iter[n]);
        row t describes model.sparse_coef_ of target t (see rtrec_slim_fit_columns).  device_out=True (feature
        selection only) leaves the first four on the device -- int32 targets, the kernels' output blocks -- for
        merge_fit(); only n_iter (one int per target) is downloaded.
        """
        be, X = self.be, self._X
        if "cptr" not in X:
            raise RuntimeError("set_interactions() with the CSC orientation must be called before fit_columns()")
        U, I = self.n_users, self.n_items
        targets = np.asarray(targets, dtype=np.int64)
        # the one-pass X^T y of small calls maps a target to ONE slot of the call (xty_tmap_kernel): a repeated target would lose
        # its candidates there, so such a call takes the per-target walks (same results)
        distinct_targets = len(np.unique(targets)) == len(targets)
        # longest columns first: the device work queue then ends on short jobs
        order = np.argsort(-X["col_nnz"][targets], kind="stable")
        targets = targets[order]
        K = int(nn_feature_selection) if nn_feature_selection is not None else 0
        if nn_feature_selection is not None and K <= 0:
            raise AssertionError(f"n_neighbors must be a positive integer: {K}")
        # K = None: a column's solution is sparse, so the output block is sized for ALLF_OUTPUT_CAP
        # coefficients per target and the rare target that has more is refitted with room for all I
        cap = min(K, I) if K > 0 else min(I, int(settings.raw("RTREC_AMD_ALLF_CAP", ALLF_OUTPUT_CAP)))
        cfg = _native.FitCfg(np.float32(alpha * l1_ratio * U), np.float32(alpha * (1.0 - l1_ratio) * U),
                             np.float32(tol), int(max_iter), sklearn_seed(random_state), int(bool(positive)), K)
        torch = be.torch
        if "sqn" not in X:
            X["sqn"] = be.empty((I,), torch.float32)
            be.column_sqnorms(I, X["cptr"], X["cval"], X["sqn"])
        n = len(targets)
        slots = int(n_slots or min(int(settings.raw("RTREC_AMD_FIT_SLOTS", self._fit_slots_for(targets, cfg, cap, K))), max(1, n)))
        # Per-slot scratch is R (U floats) + s/touched/candidates (I each).  The X^T y step is a random
        # read-modify-write over s, i.e. bound by cache lines moved, and measured faster with FEWER
        # targets in flight once a slot is several MB (C4: 1024 slots 7.3 s, 5120 slots 9.2 s): keep
        # the total near 16 GiB but never below 1024 slots.
        per_slot = 4 * (U + (5 if K <= 0 else 4) * I)
        scratch_gib = float(settings.raw("RTREC_AMD_FIT_SCRATCH_GIB", FIT_SCRATCH_GIB))
        slots = max(1, min(slots, max(1024, int(scratch_gib * (1 << 30)) // max(per_slot, 1))))

        # Bulk calls end on their heaviest targets: a popular item's X^T y is a walk over tens of
        # thousands of user rows and one wave does it strictly row after row.  The head of the
        # (length-sorted) list therefore goes to the multi-wave latency kernel on a side stream while
        # the single-wave throughput kernel works through the rest; the C-ABI picks the kernel by
        # call size (<= FIT_MW_MAX_TARGETS targets -> multi-wave), so this is two plain calls.
        n_heavy = 0
        heavy_slots = FIT_HEAVY_SLOTS
        if K > 0 and n > FIT_MW_MAX_TARGETS:
            # A call of a few thousand targets (one rank's share of a sharded fit, a large incremental fit)
            # ends with its slowest target: more of it goes to the multi-wave kernel, two workgroups per CU
            # (C3, an eighth of the targets: 0.77 -> 0.59 s; tools/fit_shard_model.py).
            small_call = n <= 2 * FIT_MW_MAX_TARGETS
            want = int(settings.raw("RTREC_AMD_FIT_HEAVY", 4 * FIT_HEAVY_TARGETS if small_call else FIT_HEAVY_TARGETS))
            heavy_slots = 2 * FIT_HEAVY_SLOTS if small_call else FIT_HEAVY_SLOTS
            heavy_min_rows = FIT_HEAVY_MIN_ROWS // 8 if small_call else FIT_HEAVY_MIN_ROWS
            nnz_sorted = X["col_nnz"][targets]
            n_heavy = int(min(want, n - FIT_MW_MAX_TARGETS - 1, np.searchsorted(-nnz_sorted, -int(settings.raw("RTREC_AMD_FIT_HEAVY_MIN_ROWS", heavy_min_rows)),
                                                                                 side="right")))
            n_heavy = max(n_heavy, 0)

        # Gram tracking (csrc/fit.hip): bulk calls on a non-negative X get the Gram matrix of the most
        # popular items, which lets the kernel decide most zero coordinates without a pass over memory.
        gram = None
        gmode = settings.raw("RTREC_AMD_GRAM", "auto")
        mode_ = mode or ("exact" if exact else "gram")
        if mode_ not in ("exact", "shuffle", "gram"):
            raise ValueError(f"fit mode must be 'exact', 'shuffle' or 'gram': {mode_}")
        fast = {"exact": 0, "shuffle": 1, "gram": 2}[mode_] if K > 0 else 0
        if fast == 1:
            n_heavy = 0            # tree-reduced dots: single-wave kernel only, one launch
        # exact mode: Gram TRACKING needs a non-negative X and pays on bulk calls; tolerance mode: Gram-form CD, any X
        if (K > 0 and min(K, I) <= 64 and getattr(be, "supports_gram", False) and gmode != "0"
                and (fast == 2 or (fast == 0 and X.get("nonneg") and (n > FIT_MW_MAX_TARGETS or gmode == "force")))):
            n_top = self._gram_items(targets, cfg, cap)
            if X.get("gram_n") != n_top:
                X["gram"], X["gram_n"] = be.gram_matrix(X, U, I, n_top), n_top
            gram = X["gram"]

        def workspace(n_slots_: int, role: str = "main"):
            """(scratch, queue, slots of the scratch layout); `role` keeps the side-stream launch of the
            heavy targets on a scratch of its own (the two kernels run concurrently).  A cached scratch of the same problem shape
            with at least n_slots_ slots is reused as it is -- the kernel launches min(slots, targets)
            workgroups, so extra slots are simply idle -- which keeps mini-batches of varying size
            (online partial_fit) from re-allocating and re-initialising it every call."""
            kk = K if K > 0 else 0
            fits = [k for k in self._fit_ws
                    if k[0] == U and k[1] == I and k[3] == kk and k[2] >= n_slots_ and k[4] == role]
            if fits:
                key = min(fits, key=lambda k: k[2])
            else:
                want = 1 << max(0, int(n_slots_ - 1).bit_length())          # next power of two
                key = (U, I, int(min(max(want, n_slots_), max(slots, n_slots_))), kk, role)
                if len(self._fit_ws) >= 4:
                    self._fit_ws.clear()
                self._fit_ws[key] = be.fit_workspace(U, I, key[2], K)
            ws_, queue_ = self._fit_ws[key]
            return ws_, queue_, key[2]

        # chunk so that the output block stays below ~1 GiB (matters for K=None, cap = I)
        chunk = max(1, min(n, int((1 << 30) // max(cap * 8, 1)))) if n else 1
        if device_out and K <= 0:
            raise ValueError("device_out needs nn_feature_selection (the all-features output block is host-sized)")
        kept: List[Dict[str, Any]] = []
        items_out = np.empty((0 if device_out else n, cap), dtype=np.int32)
        coef_out = np.empty((0 if device_out else n, cap), dtype=np.float32)
        count_out = np.empty((n,), dtype=np.int32)
        niter_out = np.empty((n,), dtype=np.int32)
        trace_out = np.zeros((n, 8), dtype=np.int64) if trace else None

        def launch(lo_: int, hi_: int, n_slots_: int, role: str = "main"):
            tg = targets[lo_:hi_]
            m = len(tg)
            ws, queue, ws_slots = workspace(n_slots_, role)
            d = dict(lo=lo_, hi=hi_, t=be.to_dev(tg.astype(np.int32)), items=be.empty((m, cap), torch.int32),
                     coef=be.empty((m, cap), torch.float32), count=be.empty((m,), torch.int32),
                     niter=be.empty((m,), torch.int32), trace=be.zeros((m, 8), torch.int64) if trace else None,
                     ws=(ws, queue))   # keeps the scratch alive while the kernel runs
            if isinstance(be, HipBackend):
                be.fit_columns(U, I, X, d["t"], cfg, d["items"], d["coef"], d["count"], d["niter"], cap, ws, queue,
                               ws_slots, d["trace"], gram, fast=fast,
                               # the heavy head of a bulk call starts its long ordered folds at once: its targets' own
                               # multi-wave walks overlap them, a one-pass X^T y up front would only delay the chain
                               one_pass_xty=(role != "heavy" and distinct_targets and self._one_pass_xty_pays(tg)))
            else:
                be.fit_columns(U, I, X, d["t"], cfg, d["items"], d["coef"], d["count"], d["niter"], cap, ws, queue,
                               ws_slots, d["trace"], gram)
            return d

        def collect(d):
            lo_, hi_ = d["lo"], d["hi"]
            if device_out:
                kept.append(d)
                if trace:
                    trace_out[lo_:hi_] = d["trace"].cpu().numpy()
                return
            items_out[lo_:hi_] = d["items"].cpu().numpy()
            coef_out[lo_:hi_] = d["coef"].cpu().numpy()
            count_out[lo_:hi_] = d["count"].cpu().numpy()
            niter_out[lo_:hi_] = d["niter"].cpu().numpy()
            if trace:
                trace_out[lo_:hi_] = d["trace"].cpu().numpy()

        heavy = None
        if n_heavy > 0:
            main = torch.cuda.current_stream(be.device)
            side = self._side_stream = getattr(self, "_side_stream", None) or torch.cuda.Stream(be.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                heavy = launch(0, n_heavy, min(n_heavy, int(settings.raw("RTREC_AMD_FIT_HEAVY_SLOTS", heavy_slots))), role="heavy")
        for s in range(n_heavy, n, chunk):
            collect(launch(s, min(n, s + chunk), min(slots, max(1, min(n, s + chunk) - s))))
        if heavy is not None:
            torch.cuda.current_stream(be.device).wait_stream(side)
            collect(heavy)
        if K <= 0 and cap < I:
            over = np.flatnonzero(count_out > cap)
            if over.size:      # refit these with room for every item, then widen the block
                full = self._fit_overflow(targets[over], cfg, U, I, slots)
                wide = int(max(int(full[2].max()), cap))
                items_w = np.zeros((n, wide), dtype=np.int32); items_w[:, :cap] = items_out
                coef_w = np.zeros((n, wide), dtype=np.float32); coef_w[:, :cap] = coef_out
                items_w[over] = full[0][:, :wide]; coef_w[over] = full[1][:, :wide]
                niter_out[over] = full[3]
                items_out, coef_out = items_w, coef_w
        self.last_fit_stats = {"n_targets": n, "slots": slots, "cap": cap, "trace": trace_out, "n_heavy": n_heavy}
        self.last_fit_targets = targets
        if device_out:
            kept.sort(key=lambda d: d["lo"])
            cat = (lambda k, shape, dt: torch.cat([d[k] for d in kept]) if kept else be.empty(shape, dt))
            d_niter = cat("niter", (0,), torch.int32)
            return (cat("t", (0,), torch.int32), cat("items", (0, cap), torch.int32), cat("coef", (0, cap), torch.float32),
                    cat("count", (0,), torch.int32), d_niter.cpu().numpy())
        return targets, items_out, coef_out, count_out, niter_out

    SGD_EPOCH_BLOCK_BYTES = 2 << 30   # time-sorted copies of X held at once by fit_columns_sgd (8 B per stored entry and epoch)

    def fit_columns_sgd(self, targets: Sequence[int], alpha: float = 0.1, l1_ratio: float = 0.1, eta0: float = 0.001,
                        max_iter: int = 100, tol: Optional[float] = 1e-4, random_state: Optional[int] = 43,
                        nn_feature_selection: Optional[int] = None):
        """optim="sgd" (slim_elastic.py:209-222): scikit-learn's SGDRegressor behind FeatureSelectionWrapper, one model per
        target column, bit-identical coef_ and n_iter_ (csrc/fit_sgd.hip; oracle: slim_oracle_sgd).  Returns device tensors
        (targets, items [n, K] in selection order, coef [n, K], count [n]) and n_iter (numpy) like
        fit_columns(device_out=True).  Steps: (1) the X^T y / top-K selection of the coordinate-descent kernel (one sweep,
        coefficients dropped); (2) per block of epochs the host schedule (shuffled sample order, learning rates, weight
        scale, cumulative L1 penalty: target-independent) and one device sort of X's entries by (column, time); (3) the
        solver kernel, one wave per target, until every target has stopped (n_iter_no_change = 5) or max_iter is reached.
        The solver is strictly sequential in the samples: U x epochs dependent steps per target (the reference's cost too)."""
        if nn_feature_selection is None:
            # the reference itself fails here (slim_elastic.py:273: SGDRegressor has no sparse_coef_)
            raise AttributeError("'SGDRegressor' object has no attribute 'sparse_coef_'")
        be, X = self.be, self._X
        U, I = self.n_users, self.n_items
        K = min(int(nn_feature_selection), I)
        if K <= 0:
            raise AssertionError(f"n_neighbors must be a positive integer: {K}")
        targets = np.asarray(targets, dtype=np.int64)
        if not isinstance(be, HipBackend):
            return be.fit_columns_sgd(X, U, I, targets, alpha, l1_ratio, eta0, max_iter, tol, random_state, K)
        if K > self.SGD_MAX_FEATURES:
            raise NotImplementedError(f"optim='sgd' on the GPU serves nn_feature_selection <= {self.SGD_MAX_FEATURES} "
                                      "(up to four features per wave lane)")
        torch = be.torch
        # (1) feature selection: the coordinate-descent kernel's own X^T y + top-K (its item lists come in selection order)
        d_t, d_sel, _coef, d_cnt, _ = self.fit_columns(targets, alpha=alpha, l1_ratio=l1_ratio, positive=True, max_iter=1,
                                                      tol=1e-4, random_state=random_state, nn_feature_selection=K,
                                                      device_out=True, mode="exact")
        n = int(d_t.numel())
        cap = int(d_sel.shape[1]) if n else K
        if n == 0:
            return d_t, d_sel, be.empty((0, cap), torch.float32), d_cnt, np.empty(0, np.int32)
        d_sel, d_cnt = d_sel.contiguous(), d_cnt.contiguous()
        nnz = int(X["crow"].numel())
        tol_ = float("-inf") if tol is None else float(tol)
        seed = sklearn_seed(random_state)
        col_of = torch.repeat_interleave(torch.arange(I, device=X["crow"].device, dtype=torch.int64),
                                         (X["cptr"][1:] - X["cptr"][:-1]).long())
        crow64 = X["crow"].long()
        block = int(max(1, min(int(max_iter), self.SGD_EPOCH_BLOCK_BYTES // max(8 * nnz, 1), 16)))
        d_w, d_q = be.zeros((n, cap), torch.float32), be.zeros((n, cap), torch.float32)
        d_best = torch.full((n,), float("inf"), dtype=torch.float64, device=be.device)
        d_noimp, d_niter = be.zeros((n,), torch.int32), be.zeros((n,), torch.int32)
        d_unf = be.zeros((1,), torch.int32)
        order = np.arange(U, dtype=np.int32)
        state = np.array([1.0, 0.0, 1.0], dtype=np.float64)
        p, hp = be.ptr, (lambda a: C.c_void_p(a.ctypes.data))
        first = 0
        while first < max_iter:
            ne = min(block, max_iter - first)
            steps = ne * U
            time_of = np.empty((ne, U), dtype=np.int32)
            eta, wsb, wsa, ua = (np.empty(steps, dtype=np.float64) for _ in range(4))
            rcnt = np.empty(steps + 1, dtype=np.int32)
            rmult = np.empty(steps + 1, dtype=np.float32)
            n_res = C.c_int32(0)
            _native.check(be.lib.rtrec_slim_sgd_schedule(U, ne, seed, float(alpha), float(l1_ratio), float(eta0), 0.25, hp(order),
                                                         hp(state), hp(time_of), hp(eta), hp(wsb), hp(wsa), hp(ua), hp(rcnt),
                                                         hp(rmult), steps + 1, C.byref(n_res)), "rtrec_slim_sgd_schedule")
            d_time_of = be.to_dev(time_of)
            d_tt = be.empty((ne, nnz), torch.int32)
            d_tv = be.empty((ne, nnz), torch.float32)
            for e in range(ne):        # every column's entries by their time in this epoch (stable in the column: keys are distinct)
                key = col_of * U + d_time_of[e].long()[crow64]
                skey, perm = torch.sort(key)
                d_tt[e] = (skey % U).to(torch.int32)
                d_tv[e] = X["cval"][perm]
            d_eta, d_wsb, d_wsa, d_ua = (be.to_dev(a) for a in (eta, wsb, wsa, ua))
            d_rcnt, d_rmult = be.to_dev(rcnt), be.to_dev(rmult[:max(int(n_res.value), 1)].copy())
            be.ops.fit_sgd_epochs(X["cptr"], d_tt, d_tv, d_t, d_sel, d_cnt, U, I, nnz, cap, first, ne, int(max_iter), tol_, d_eta, d_wsb,
                                  d_wsa, d_ua, d_rcnt, d_rmult, d_w, d_q, d_best, d_noimp, d_niter, d_unf)
            first += ne
            if int(d_unf.item()) == 0:
                break
        n_iter = d_niter.cpu().numpy()
        if (n_iter < 0).any():
            raise ValueError("Floating-point under-/overflow occurred during the SGD fit of column "
                             f"{int(targets[np.flatnonzero(n_iter < 0)[0]])}. Scaling input data with StandardScaler or "
                             "MinMaxScaler might help.")
        return d_t, d_sel, d_w, d_cnt, n_iter

    SGD_MAX_FEATURES = 256           # csrc/fit_sgd.hip: kSgdMaxFeatures

    FIT_SLOTS_SPARSE = 1024          # targets in flight when the folded columns are sparse (see _fit_slots_for)
    FIT_SPARSE_FEATURES = 0.08       # mean density of the selected feature columns up to which FIT_SLOTS_SPARSE are used (c3s: 0.058)
    FIT_DENSE_FEATURES = 0.25        # ... and from which the full MAX_SLOTS are (C3: 0.39; C2: 0.13 -> 2048, flat there)

    def _fit_slots_for(self, targets: np.ndarray, cfg, cap: int, K: int) -> int:
        """Targets in flight (work-queue slots) of a bulk fit with feature selection.  Every target owns a residual of U floats
        and its waves gather from it at the rows of the columns they fold.  Where those columns are DENSE (popularity-only
        data: the features of every target are the ~100 most popular items, 30-70 % of all users each) a fold streams its
        residual nearly sequentially, the kernel is bound by the dependent-add chains and the fabric's streaming rate, and more
        targets in flight are better up to 4 per SIMD (C3: 4096 slots 2.05 s, 2048 2.3 s, 1024 3.6 s).  Where they are SPARSE
        (item clusters: ~8k-entry columns, 6 % of the users) every gather is a 64-byte sector of its own, the memory system
        serves a fixed number of such sectors per second whatever the number of waves asking (Little's law: a wave's step
        took ~100 us with 4096 waves in flight), and what helps is residuals that stay in L2 + Infinity Cache: c3s 4096 slots
        1.43-1.52 s, 2048 1.30 s, 1024 0.91 s, 768 0.97-1.03 s, 512 1.03 s (tools/fit_sweep.py, profiles/r04_fit_sweep_*.jsonl;
        W bits and sweep counts identical for every slot count -- the slots only decide who runs when).  The density comes
        from the Gram pilot's feature selection (_gram_items); in between the count is interpolated (C2, 0.13: flat)."""
        if K <= 0 or len(targets) <= FIT_MW_MAX_TARGETS:
            return MAX_SLOTS
        if "pilot_feature_density" not in self._X:
            self._gram_items(targets, cfg, cap)            # runs the pilot when the call is large enough for one
        d = self._X.get("pilot_feature_density")
        if d is None:
            return MAX_SLOTS
        f = min(1.0, max(0.0, (d - self.FIT_SPARSE_FEATURES) / (self.FIT_DENSE_FEATURES - self.FIT_SPARSE_FEATURES)))
        lo = self.FIT_SLOTS_SPARSE
        return int(-(-(lo + f * (MAX_SLOTS - lo)) // 256) * 256) if f < 1.0 else MAX_SLOTS

    GRAM_ITEMS_MAX = 4096       # rtrec_slim_gram_matrix's limit
    GRAM_PILOT_TARGETS = 64
    GRAM_PILOT_MIN_NNZ = 8_000_000      # smaller fits take a fraction of a second: the pilot (and a larger G) would not pay

    def _gram_items(self, targets: np.ndarray, cfg, cap: int) -> int:
        """How many of the most popular item columns the shared Gram matrix covers (RTREC_AMD_GRAM_ITEMS=<n> fixes it).
        Gram tracking (exact mode) decides a zero coordinate without touching memory only while every UPDATED feature of the
        target has a row in G, and the Gram-form coordinate descent (tolerance mode) needs ALL of a target's features there.
        Which items are features depends on the co-occurrence structure, not on the popularity law: on popularity-only data
        the features of every target are the ~100 most popular items, with item clusters they are each cluster's own
        popular items and reach down to popularity rank ~5,000 (ML-20M shape, 80 clusters: top-512 holds 23 % of the
        selected features, top-4096 97 % and every feature that ends up with a non-zero weight).  So the size is read off a
        PILOT: the feature selection (one sweep) of 64 of the call's targets from the middle of the length-sorted
        list -- the 90th percentile of the deepest popularity rank a target selects, rounded up to a power of two in
        [512, 4096].  A matrix with few non-empty columns (a mini-batch's partial matrix, slim.py:33-36) gets them all.
        The answer is cached with X."""
        be, X = self.be, self._X
        env = settings.raw("RTREC_AMD_GRAM_ITEMS", "auto")
        if env != "auto":
            return min(self.n_items, int(env))
        if "gram_auto" in X:
            return X["gram_auto"]
        col_nnz = X["col_nnz"]
        nonempty = int(np.count_nonzero(col_nnz))
        n_top = GRAM_ITEMS
        if nonempty <= 1024:
            n_top = max(64, min(self.n_items, nonempty))
        elif (len(targets) > FIT_MW_MAX_TARGETS and isinstance(be, HipBackend) and int(cfg.top_features) > 0
              and int(col_nnz.sum()) >= self.GRAM_PILOT_MIN_NNZ):
            torch = be.torch
            live = targets[col_nnz[targets] > 0]
            # (not the head of the length-sorted list: a popular target's X^T y walks 100k user rows, and what it selects is no
            # different -- the pilot costs a few ms this way instead of ~100)
            pick = (live[np.linspace(0.05 * (len(live) - 1), 0.6 * (len(live) - 1), min(self.GRAM_PILOT_TARGETS, len(live))).astype(np.int64)]
                    if len(live) else live)
            if len(pick):
                m = len(pick)
                pcfg = _native.FitCfg(cfg.l1_reg, cfg.l2_reg, cfg.tol, 1, cfg.seed, cfg.positive, cfg.top_features)
                key = (self.n_users, self.n_items, m, int(cfg.top_features), "pilot")
                if key not in self._fit_ws:
                    self._fit_ws[key] = be.fit_workspace(self.n_users, self.n_items, m, int(cfg.top_features))
                ws, queue = self._fit_ws[key]
                items = be.empty((m, cap), torch.int32)
                coef = be.empty((m, cap), torch.float32)
                count, niter = be.empty((m,), torch.int32), be.empty((m,), torch.int32)
                be.fit_columns(self.n_users, self.n_items, X, be.to_dev(pick.astype(np.int32)), pcfg, items, coef, count, niter, cap,
                               ws, queue, m, None, None, fast=0, one_pass_xty=False)
                rank = np.empty(self.n_items, dtype=np.int64)
                rank[np.argsort(-col_nnz, kind="stable")] = np.arange(self.n_items)
                it, cn = items.cpu().numpy(), count.cpu().numpy()
                deepest = np.array([rank[it[k, :cn[k]]].max() if cn[k] else 0 for k in range(m)])
                need = int(np.percentile(deepest, 90)) + 1
                sel = np.concatenate([it[k, :cn[k]] for k in range(m)]) if cn.sum() else np.empty(0, np.int64)
                if len(sel):          # how dense the columns a target folds are (_fit_slots_for)
                    X["pilot_feature_density"] = float(col_nnz[sel].mean()) / max(self.n_users, 1)
                n_top = GRAM_ITEMS if need <= GRAM_ITEMS else min(self.GRAM_ITEMS_MAX, 1 << int(need - 1).bit_length())
                self._fit_ws.pop(key, None)
        X["gram_auto"] = min(self.n_items, n_top)
        return X["gram_auto"]

    def _one_pass_xty_pays(self, targets: np.ndarray) -> bool:
        """The one-pass X^T y of a small call (csrc/fit.hip, xty_batch_kernel) replaces one walk per target by one pass
        over X plus fixed costs (row compaction, per-column scans of the target sums, the latency chain of the longest
        column).  Targets with >= 1024 users walk all of X each (kColWalkMinRows), so it pays when that traffic is
        large (targets x entries of the matrix being fitted): measured between C2 (175 such targets x 1.1 M entries: 16 -> 23 ms with
        it) and C3 (450 x 5.0 M: 58 -> 44 ms)."""
        if settings.raw("RTREC_AMD_XTY_BATCH") == "force":      # parity tests: small matrices through this path
            return True
        col_nnz = self._X["col_nnz"]
        big = int(np.count_nonzero(col_nnz[targets] >= 1024))
        walks = big * float(col_nnz.sum())
        # ... and its own fixed cost is one scan of the target sums per item column: a wide catalogue (C4: 500k columns,
        # 879 -> 957 ms with it) pays more for that than the walks cost
        scans = float(self.n_items) * -(-len(targets) // 64)
        if settings.raw("RTREC_AMD_DEBUG_XTY"):
            print(f"[xty] targets={len(targets)} big={big} nnz={int(col_nnz.sum())} walks={walks:.3g} scans={scans:.3g}", flush=True)
        if walks < 2000.0 * scans:
            return False
        return walks >= XTY_MIN_WALK_ENTRIES

    def _fit_overflow(self, targets: np.ndarray, cfg, U: int, I: int, slots: int):
        """K = None targets whose solution did not fit the output block: fit them again with cap = I."""
        be, X, torch = self.be, self._X, self.be.torch
        m = len(targets)
        n_slots = max(1, min(slots, m))
        key = (U, I, n_slots, 0, "overflow")
        if key not in self._fit_ws:
            self._fit_ws[key] = be.fit_workspace(U, I, n_slots, 0)
        ws, queue = self._fit_ws[key]
        d_t = be.to_dev(targets.astype(np.int32))
        items, coef = be.empty((m, I), torch.int32), be.empty((m, I), torch.float32)
        count, niter = be.empty((m,), torch.int32), be.empty((m,), torch.int32)
        be.fit_columns(U, I, X, d_t, cfg, items, coef, count, niter, I, ws, queue, n_slots, None, None)
        return items.cpu().numpy(), coef.cpu().numpy(), count.cpu().numpy(), niter.cpu().numpy()

    # ------------------------------------------------------------------------------ W
    def upload_weights(self, W_csc: sp.csc_matrix, acc_f64: Optional[bool] = None) -> DeviceWeights:
        """Host W (I x I, CSC) -> DeviceWeights (weights are float32 on the device, like every layout built from them)."""
        be, torch = self.be, self.be.torch
        W_csc = W_csc if W_csc.has_sorted_indices else W_csc.sorted_indices()
        n_items = W_csc.shape[1]
        data = np.asarray(W_csc.data)
        nz = data != 0
        cols = np.repeat(np.arange(n_items, dtype=np.int64), np.diff(W_csc.indptr))
        f64 = bool(W_csc.dtype == np.float64) if acc_f64 is None else bool(acc_f64)
        d32 = data[nz].astype(np.float32)
        lossy = bool(data.dtype != np.float32 and not np.array_equal(d32.astype(data.dtype), data[nz]))
        return DeviceWeights(be.to_dev(np.asarray(W_csc.indices, dtype=np.int64)[nz]), be.to_dev(cols[nz]), be.to_dev(d32),
                             n_items, f64, host=W_csc if nz.all() else None, lossy=lossy)

    def set_weights(self, W: Any, acc_f64: Optional[bool] = None) -> None:
        """Register W: a host scipy CSC matrix (uploaded) or a DeviceWeights (what merge_fit returns: nothing moves).
        The layouts of this rank's column shard are built ON THE DEVICE on first use: a compacted one (columns with
        at least one weight) plus the feature-row form for SPARSE mode, a plain one for DENSE / CANDIDATES mode."""
        dw = W if isinstance(W, DeviceWeights) else self.upload_weights(W, acc_f64)
        if acc_f64 is not None:
            dw.f64 = bool(acc_f64)
        n_items = dw.n_items
        lo, hi = (0, n_items) if self.score_shard == "rows" else shard_bounds(n_items, self.world_size, self.rank)
        self._W = {"n_items": n_items, "col_lo": lo, "col_hi": hi, "acc_f64": dw.f64, "dw": dw, "layouts": {}}
        self._X.pop("_orders", None)           # work orders are per layout

    @property
    def weights(self) -> Optional[DeviceWeights]:
        return self._W.get("dw") if self._W else None

    def merge_fit(self, old: Optional[DeviceWeights], n_items: int, f64: bool, d_targets, d_items, d_coef, d_count
                  ) -> DeviceWeights:
        """The LIL write-back of slim_elastic.py:271-274 / 371-374 / 554-557 on the device: start from `old`, then for
        every fitted (i, j) a non-zero value overwrites W[i, j] and an explicit zero deletes it; entries of column j
        that the new solution does not mention survive.  The arguments are this rank's
        fit_columns(device_out=True) tensors; with several ranks the triples are exchanged with all_gather_into_tensor
        (int64 keys + float32 values, padded to the longest part) -- nothing is pickled and nothing visits the host."""
        torch = self.be.torch
        n, cap = d_items.shape
        mask = torch.arange(cap, device=d_items.device)[None, :] < d_count[:, None]
        key = (d_targets.to(torch.int64)[:, None] * n_items + d_items.to(torch.int64))[mask]
        val = d_coef[mask]
        if self.world_size > 1 and not self.shard_w:
            import torch.distributed as dist
            sizes = torch.zeros(self.world_size, dtype=torch.int64, device=key.device)
            dist.all_gather_into_tensor(sizes, torch.tensor([key.numel()], dtype=torch.int64, device=key.device), group=self.group)
            sizes = sizes.cpu().tolist()
            m = max(max(sizes), 1)
            kp = torch.zeros(m, dtype=torch.int64, device=key.device); kp[:key.numel()] = key
            vp = torch.zeros(m, dtype=torch.float32, device=key.device); vp[:val.numel()] = val
            ka = torch.empty(m * self.world_size, dtype=torch.int64, device=key.device)
            va = torch.empty(m * self.world_size, dtype=torch.float32, device=key.device)
            dist.all_gather_into_tensor(ka, kp, group=self.group)
            dist.all_gather_into_tensor(va, vp, group=self.group)
            key = torch.cat([ka[r * m:r * m + s] for r, s in enumerate(sizes)])
            val = torch.cat([va[r * m:r * m + s] for r, s in enumerate(sizes)])
        if self.shard_w and old is not None and (getattr(old, "shard", None) is None or old.n_items != n_items):
            # the catalogue grew (the block boundaries moved) or the old W was not sharded: every rank keeps the old entries
            # of ITS columns under the new boundaries
            old = self.gather_weights(old)
            lo, hi = shard_bounds(n_items, self.world_size, self.rank)
            sel = (old.cols >= lo) & (old.cols < hi)
            old = DeviceWeights(old.rows[sel], old.cols[sel], old.vals[sel], old.n_items, old.f64)
        if old is not None and old.nnz:
            inside = (old.rows < n_items) & (old.cols < n_items)
            o_key = (old.cols * n_items + old.rows)[inside]
            keep = ~torch.isin(o_key, key)
            key = torch.cat([o_key[keep], key])
            val = torch.cat([old.vals[inside][keep], val])
        nz = val != 0
        key, val = key[nz], val[nz]
        order = torch.argsort(key)
        key = key[order]
        dw = DeviceWeights(key % n_items, key // n_items, val[order].contiguous(), n_items, f64)
        if self.shard_w:
            dw.shard = (self.rank, self.world_size)
        return dw

    def gather_weights(self, dw: "DeviceWeights") -> "DeviceWeights":
        """All of W from its column shards (shard_w): the ranks' triples concatenated in rank order = (column, row) order,
        because the shards are contiguous column blocks.  A COLLECTIVE: every rank calls it (reading `item_similarity`,
        pickling).  A W that is not sharded is returned as it is."""
        if getattr(dw, "shard", None) is None or self.world_size == 1:
            return dw
        import torch.distributed as dist
        torch = self.be.torch
        dev = dw.vals.device
        sizes = torch.zeros(self.world_size, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(sizes, torch.tensor([dw.nnz], dtype=torch.int64, device=dev), group=self.group)
        sizes = sizes.cpu().tolist()
        m = max(max(sizes), 1)
        key = dw.cols * dw.n_items + dw.rows
        kp = torch.zeros(m, dtype=torch.int64, device=dev); kp[:dw.nnz] = key
        vp = torch.zeros(m, dtype=torch.float32, device=dev); vp[:dw.nnz] = dw.vals
        ka = torch.empty(m * self.world_size, dtype=torch.int64, device=dev)
        va = torch.empty(m * self.world_size, dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(ka, kp, group=self.group)
        dist.all_gather_into_tensor(va, vp, group=self.group)
        key = torch.cat([ka[r * m:r * m + n] for r, n in enumerate(sizes)])
        val = torch.cat([va[r * m:r * m + n] for r, n in enumerate(sizes)])
        return DeviceWeights(key % dw.n_items, key // dw.n_items, val.contiguous(), dw.n_items, dw.f64)

    def _fast_layout(self) -> Optional[Dict[str, Any]]:
        """The fast SPARSE-mode form of this rank's shard (float32 W): feature rows when W has at most 128 non-empty rows
        (popularity-only data), else segments (seg_layout.py).  Built once per W, on the device; None when neither applies
        (float64 W, a backend without the kernels, an empty shard).  Independent of the tiled layout, which SPARSE-mode
        scoring then needs only for the rows whose lists hold an exact score tie (see _local_topk)."""
        W, be = self._W, self.be
        if "fast" in W:
            return W["fast"]
        fast = None
        dw: DeviceWeights = W["dw"]
        torch = be.torch
        if W["col_hi"] > W["col_lo"] and self._fast_dtype_ok() and dw.nnz > 0:
            if getattr(be, "supports_feature_rows", False):
                fr = build_feature_rows_device(torch, dw.rows, dw.cols, dw.vals, W["n_items"], W["col_lo"], W["col_hi"],
                                               tile_cols=self.FR_TILE_COLS)
                if fr is not None:
                    nb = int(be.lib.rtrec_slim_score_fr_scratch_bytes(fr["fr_n_tiles"], fr["fr_tile_cols"]))
                    fr["n_cols"] = int(fr.pop("col_ids_sorted").numel())
                    fr["fr_scratch"] = be.empty((nb,), torch.uint8)      # fr_host: small host copies bench.py prices the kernel's work from
                    fast = fr
            # a W with many rows (item-item structure in the data): the segment layout; the cluster labels that order its
            # columns are kept while W changes little (a mini-batch refits ~3 % of the columns)
            if fast is None:
                fast = W.get("fast_small") or self._seg_form()       # (a small batch may have built it already)
        if fast is not None:
            W["n_active"] = fast["n_cols"]
        W["fast"] = fast
        return fast

    def _fast_dtype_ok(self) -> bool:
        """The fast layouts hold float32 weights and their kernels accumulate float32: right for a float32 W, and usable for a
        float64 W whose values are float32 numbers and all positive (the serial fit's output) through the refine step of
        _local_topk (rtrec_slim_refine_topk_f64)."""
        W = self._W
        if not W["acc_f64"]:
            return True
        return self._f64_refine_mode() != 0

    def _f64_refine_w_ok(self) -> bool:
        """W serves the float64 refine step with the SIGN argument (all weights >= F64_REFINE_MIN_VALUE)."""
        return self._f64_refine_mode() == 1

    def _f64_refine_mode(self) -> int:
        """0: no refine step for this W (lossy upload, switched off, another backend); 1: every weight positive -- relative
        margin (csrc/score_refine.hip); 2: signed weights (positive_only=False, slim_elastic.py:187) -- the absolute slack
        2 (n_u + 2) 2^-24 sum_i |x_ui| max_c |w_ic| per user (_f64_abs_slack), SPARSE mode only."""
        W = self._W
        if "f64_refine" not in W:
            dw: DeviceWeights = W["dw"]
            mode = 0
            if self.f64_refine and isinstance(self.be, HipBackend) and not dw.lossy and dw.nnz > 0:
                mode = 1 if float(dw.vals.min()) >= self.F64_REFINE_MIN_VALUE else 2
                ptr = dw.csc_arrays(self.be.torch)[0]
                W["col_nnz_max"] = int((ptr[1:] - ptr[:-1]).max())
            W["f64_refine"] = mode
        return W["f64_refine"]

    def _f64_abs_slack(self, xb):
        """Signed refine: per row of the batch matrix the bound 2 (n_u + 2) 2^-24 B_u (1 + 1e-6) + 1e-30 with B_u = sum_i
        |x_ui| max_c |w_ic| >= sum |x w| of every column (float64 tensor; cached for the resident X and this W)."""
        torch, W = self.be.torch, self._W
        ptr, col, val = xb
        resident = self._X.get("rptr") is ptr
        # keyed by the identity of the resident row-pointer tensor: set_interactions() replaces X without touching W, and a
        # slack computed for the previous X is too short (new users) or too small (changed ratings) -- ADVICE round 4
        ent = W.get("_f64_slack")
        if resident and ent is not None and ent[0] is ptr:
            return ent[1]
        dw: DeviceWeights = W["dw"]
        if "_row_absmax" not in W:
            rmax = torch.zeros(max(W["n_items"], 1), dtype=torch.float64, device=dw.vals.device)
            rmax.scatter_reduce_(0, dw.rows, dw.vals.abs().double(), "amax", include_self=True)
            W["_row_absmax"] = rmax
        rmax = W["_row_absmax"]
        c64 = col.long()
        inside = c64 < rmax.shape[0]
        contrib = val.abs().double() * torch.where(inside, rmax[c64.clamp(max=rmax.shape[0] - 1)], torch.zeros((), dtype=torch.float64, device=val.device))
        p64 = ptr.long()
        lengths = p64[1:] - p64[:-1]
        n_u = lengths.double()
        try:                # per-row float64 sums (error n_u 2^-53 B_u: far inside the 1e-6 below)
            lo, hi = int(p64[0]), int(p64[-1])
            B = torch.segment_reduce(contrib[lo:hi], "sum", lengths=lengths, unsafe=True)
            err_b = 0.0
        except Exception:   # cumulative sums instead: a difference of two prefixes carries up to 2 N 2^-53 of the TOTAL
            cs = torch.cat([torch.zeros(1, dtype=torch.float64, device=val.device), torch.cumsum(contrib, 0)])
            B = cs[p64[1:]] - cs[p64[:-1]]
            err_b = 2.0 * float(contrib.numel()) * 2.0 ** -53 * float(cs[-1])
        slack = 2.0 * (n_u + 2.0) * 2.0 ** -24 * (B * (1.0 + 1e-6) + err_b) + 1e-30
        if resident:
            W["_f64_slack"] = (ptr, slack)
        return slack

    def _f64_refine_x_ok(self, xb) -> bool:
        """Ratings all >= F64_REFINE_MIN_VALUE (no negative addend, no float32 underflow of a product); cached for the
        resident X."""
        val = xb[2]
        if val.numel() == 0:
            return True
        if self._X.get("rval") is val:
            if "rval_min_ok" not in self._X:
                self._X["rval_min_ok"] = bool(float(val.min()) >= self.F64_REFINE_MIN_VALUE)
            return self._X["rval_min_ok"]
        return bool(float(val.min()) >= self.F64_REFINE_MIN_VALUE)

    def _dense_fill_ok(self, xb) -> bool:
        """rtrec_slim_dense_fill's precondition: weights and ratings all >= F64_REFINE_MIN_VALUE -- every product is a positive
        normal float32 number, so a column the user's row touches has a positive score and every other one exactly +0.0."""
        W = self._W
        if "w_min_ok" not in W:
            dw: DeviceWeights = W["dw"]
            W["w_min_ok"] = bool(not dw.lossy and (dw.nnz == 0 or float(dw.vals.min()) >= self.F64_REFINE_MIN_VALUE))
        return W["w_min_ok"] and self._f64_refine_x_ok(xb)

    def _seg_form(self) -> Optional[Dict[str, Any]]:
        """The segment layout of this rank's shard as a layout dict ({"sg": ..., "n_cols": ...}), or None."""
        W, be = self._W, self.be
        dw: DeviceWeights = W["dw"]
        torch = be.torch
        if not (self.use_seg_layout and getattr(be, "supports_seg_layout", False)):
            return None
        from .seg_layout import build_seg_layout_device
        labels = None
        kept = self._sg_labels
        if kept is not None and kept[1] == W["n_items"] and abs(dw.nnz - kept[2]) <= self.SG_RELABEL_FRACTION * max(kept[2], 1):
            labels = kept[0]
        if not self.seg_cluster:
            labels = torch.arange(W["n_items"], dtype=torch.int64, device=dw.rows.device)
        if self.native_seg_builder and hasattr(be, "lib") and dw.nnz < (1 << 27):
            from .seg_layout import build_seg_layout_native, cluster_labels_device
            fresh = labels is None
            if fresh:       # first layout of a model (or W has changed a lot): label propagation over the shard's graph
                sel = (dw.cols >= W["col_lo"]) & (dw.cols < W["col_hi"])
                labels = cluster_labels_device(torch, dw.rows[sel], dw.cols[sel], dw.vals[sel], W["n_items"])
            sg = build_seg_layout_native(be, dw.rows, dw.cols, dw.vals, W["n_items"], W["col_lo"], W["col_hi"], labels)
            labels = None if fresh else labels
        else:
            sg = build_seg_layout_device(torch, dw.rows, dw.cols, dw.vals, W["n_items"], W["col_lo"], W["col_hi"], labels=labels)
        if sg is None:
            return None
        if labels is None:
            self._sg_labels = (sg["sg_labels"], W["n_items"], dw.nnz)
        # zeroed scratch of the heavy pass (long users, one workgroup each); the kernel leaves it zero
        # (kept across layouts of the same size: 130 MB for the ML-20M shape)
        nb = int(be.lib.rtrec_slim_score_sg_scratch_bytes(W["n_items"], sg["sg_n_tiles"], sg["sg_T"]))
        if self._sg_scratch is None or self._sg_scratch.numel() != nb:
            self._sg_scratch = be.zeros((nb,), torch.uint8)
        sg["sg_scratch"] = self._sg_scratch
        return {"sg": sg, "n_cols": int(sg["sg_n_cols"])}

    def _small_batch_layout(self) -> Optional[Dict[str, Any]]:
        """The layout a request-sized batch is scored with: the segment form, whatever the shape of W.  The feature-row kernel
        gives a user to ONE wave that sweeps every tile of W from LDS: 150 us per launch on the ML-20M shape however few
        the users (it is a throughput kernel -- a full pass of 138k users is 1.4 ms).  The segment kernels touch only the
        user's own rows of W, prune tiles by their bounds and give a long user a workgroup: 10-25 us for one user.  So the
        segment form is what a small batch builds (once per W; after a mini-batch the first recommend builds this one
        layout), and the feature-row form waits for the first large pass."""
        W, be = self._W, self.be
        if "fast" in W and (W["fast"] is None or "sg" in W["fast"]):
            return W["fast"]
        if "fast_small" not in W:
            ok = W["col_hi"] > W["col_lo"] and self._fast_dtype_ok() and W["dw"].nnz > 0
            W["fast_small"] = self._seg_form() if ok else None
            if W["fast_small"] is not None:
                W.setdefault("n_active", W["fast_small"]["n_cols"])
        return W["fast_small"] or self._fast_layout()

    def _tile_width(self, compact: bool, top_k: int) -> int:
        """Tile width of the tiled layout: self.tile_cols unless the merge of the per-tile lists (n_tiles * (top_k + 1) <= 1024
        candidates per user) needs wider tiles; capped by what the exact-tie pass fits in LDS (accumulator + first-touch
        word per column)."""
        W = self._W
        acc = 8 if W["acc_f64"] else 4
        max_tile = 256
        while max_tile * 2 * (acc + 4) + 4096 + 16 * (top_k + 64) <= 160 * 1024 and max_tile < 32768:
            max_tile *= 2
        tile = min(self.tile_cols, max_tile)
        width = W["col_hi"] - W["col_lo"]
        if compact and "n_active" in W:
            width = W["n_active"]
        while tile < max_tile and -(-max(width, 1) // tile) * (top_k + 1) > 1024:
            tile *= 2
        return tile

    def _layout(self, compact: bool, top_k: int = 10) -> Optional[Dict[str, Any]]:
        """Tiled layout of this rank's shard (built on the device on first use), with the fast SPARSE-mode form of the
        shard (_fast_layout) merged in when compact."""
        W, be = self._W, self.be
        if compact:
            self._fast_layout()               # (sets n_active, which the tile width of a compacted layout depends on)
        tile = self._tile_width(compact, top_k)
        key = (compact, tile)
        if key not in W["layouts"]:
            lay = None
            dw: DeviceWeights = W["dw"]
            torch = be.torch
            if W["col_hi"] > W["col_lo"]:
                lay = build_tiled_w_device(torch, dw.rows, dw.cols, dw.vals, W["n_items"], W["col_lo"], W["col_hi"], tile,
                                           compact=compact, dense_fill=DENSE_ROW_FILL if compact else None)
                if lay is not None and compact:
                    W["n_active"] = lay["n_cols"]
                    fast = self._fast_layout()
                    if fast is not None:
                        lay.update({k: v for k, v in fast.items() if k != "n_cols"})
            W["layouts"][key] = lay
        return W["layouts"][key]

    # ------------------------------------------------------------------------------ score
    def _local_topk(self, d_row_ids, n_rows: int, xb, top_k: int, filter_interacted: bool, mode: int,
                    d_col_rank, pad_rows: int = 0, host: bool = False):
        """This rank's lists (ids, scores, float64 scores or None, tie keys, counts) for its columns of W.  A rank that holds
        only PART of the columns (column shards) completes the SPARSE-mode tie key of every entry before the lists travel: two
        columns of different shards can have equal scores without either shard seeing a tie (csrc/score_first_touch.hip)."""
        out = self._local_topk_impl(d_row_ids, n_rows, xb, top_k, filter_interacted, mode, d_col_rank, pad_rows, host)
        be, W = self.be, self._W
        lo, hi = W.get("col_lo", 0), W.get("col_hi", 0)
        if (mode == _native.TOPK_SPARSE and n_rows > 0 and isinstance(be, HipBackend) and hi > lo
                and not (lo == 0 and hi == W["n_items"])):
            ids, _sc, _sc64, aux, cnt = out
            wc_ptr, wc_row, _ = W["dw"].csc_arrays(be.torch)
            be.ops.first_touch_aux(d_row_ids, xb[0], xb[1], n_rows, W["n_items"], wc_ptr, wc_row, top_k, ids, cnt, aux)
        return out

    def _local_topk_impl(self, d_row_ids, n_rows: int, xb, top_k: int, filter_interacted: bool, mode: int,
                         d_col_rank, pad_rows: int = 0, host: bool = False):
        be, W = self.be, self._W
        torch = be.torch
        # ids | scores | counts are views of ONE buffer: a caller that wants them on the host downloads it in one copy
        # (_download: a single-user recommend is three device-to-host round trips otherwise), and the row-sharded path
        # all-gathers it as it is (pad_rows: the buffer is laid out for that many rows, so that every rank's has one size)
        # (host: the caller downloads the lists; the lazy path below then reads its flag counter out of the same copy)
        cap = max(n_rows, int(pad_rows))
        nk = cap * top_k
        flag_words = n_rows + 1 if (host and not pad_rows) else 0
        pack = be.empty((2 * nk + cap + flag_words,), torch.int32)
        ids = pack[:nk].view(cap, top_k)[:n_rows]
        sc = pack[nk:2 * nk].view(torch.float32).view(cap, top_k)[:n_rows]
        cnt = pack[2 * nk:2 * nk + n_rows]
        if cap == n_rows:
            ids._rtrec_pack = pack
        aux = be.empty((n_rows, top_k), torch.int32)
        sc64 = be.empty((n_rows, top_k), torch.float64) if W["acc_f64"] else None
        sparse = mode == _native.TOPK_SPARSE
        hip = isinstance(be, HipBackend)
        # Which kernels serve this call is decided by two pure functions (rtrec_amd/score_plan.py; table-tested on the CPU):
        #   SPARSE mode, float32 W: the fast form of the shard (feature rows / segments).  The tiled layout is then needed only
        #   for the rows whose fast-pass list holds an exact score tie (the exact-tie pass orders those like the reference): with
        #   lazy_tiled it is built when a call first flags such a row -- after a mini-batch the first recommend builds one
        #   layout, not two -- at the price of reading one counter back per call while it does not exist.
        #   DENSE mode (string ids: every column competes, slim_elastic.py:745-778) takes the same fast pass when this rank holds
        #   all of W's columns: a row whose leading top_k scores are all positive is final (positives outrank every zero-score
        #   column, zeros outrank negatives); the kernel flags the others, and every tie (DENSE orders ties by item id), for the
        #   tiled DENSE kernel below.  (A COLUMN SHARD flags nearly every row -- a user's positive scores sit in a few shards --
        #   so it takes the fast pass only when the flagged rows can be completed in place: rtrec_slim_dense_fill.)
        #   float64 W (serial fit): the float32 fast pass for top_k + 1 columns, then the candidates' float64 scores
        #   (rtrec_slim_refine_topk_f64) -- W and X all positive, or (SPARSE) signed with an absolute per-user slack; otherwise
        #   the float64 tiled kernel.
        facts = score_plan.ScoreFacts(
            mode=mode, hip=hip, acc_f64=bool(W["acc_f64"]), top_k=top_k, n_rows=n_rows,
            full_range=(W.get("col_lo", 0) == 0 and W.get("col_hi", 0) == W["n_items"]),
            nonempty_shard=(W.get("col_hi", 0) > W.get("col_lo", 0)),
            dense_fast_on=bool(self.dense_fast), dense_fill_on=bool(self.dense_fill), lazy_tiled=bool(self.lazy_tiled),
            feature_rows_on=bool(self.use_feature_rows), seg_layout_on=bool(self.use_seg_layout),
            seg_supported=bool(getattr(be, "supports_seg_layout", False)),
            dense_fill_ok=lambda: self._dense_fill_ok(xb), f64_w_ok=self._f64_refine_w_ok,
            f64_x_ok=lambda: self._f64_refine_x_ok(xb), f64_refine_mode=self._f64_refine_mode,
            limits=score_plan.Limits(self.FR_SMALL_BATCH, self.FR_MIN_ROWS, self.FR_MAX_TOP_K, self.SG_MAX_TOP_K))
        req = score_plan.plan_fast_layout(facts)
        fast = None
        if req.want:
            # the segment form: request-sized batches, and any batch whose top_k the feature-row kernel's lists do not hold
            fast = self._small_batch_layout() if req.small else self._fast_layout()
        tiled_key = (sparse, self._tile_width(sparse, top_k))
        plan = score_plan.choose_path(facts, req, has_fr=bool(fast is not None and fast.get("fr_w") is not None),
                                      has_sg=bool(fast is not None and fast.get("sg") is not None),
                                      tiled_exists=tiled_key in W["layouts"])
        use_fr, use_sg, dense_fast = plan.use_fr, plan.use_sg, req.dense_fast
        if plan.path is score_plan.Path.FAST_F64:
            return self._local_topk_f64(d_row_ids, n_rows, xb, top_k, filter_interacted, d_col_rank, fast, use_fr, use_sg,
                                        ids, sc, sc64, aux, cnt, mode, signed=plan.f64_signed)
        if plan.path is score_plan.Path.FAST and plan.lazy:
            need = be.score_workspace_bytes(n_rows, 1, top_k)
            if self._score_ws is None or self._score_ws.numel() < need:
                self._score_ws = be.empty((need,), torch.uint8)
            order = self._row_order(d_row_ids, n_rows, xb, fast, allow_grouped=use_fr)
            self.last_score_path = "feature_rows" if use_fr else "segments"
            flagged = pack[2 * nk + cap:] if flag_words else be.empty((n_rows + 1,), torch.int32)
            fill = plan.fill
            flagged_fast = be.empty((n_rows + 1,), torch.int32) if fill else flagged
            be.score_topk(n_rows, d_row_ids, xb, W["n_items"], W["col_lo"], fast, d_col_rank, top_k, filter_interacted,
                          mode, False, ids, sc, None, aux, cnt, self._score_ws, timer=self.score_timer,
                          diagnostics=self.diagnostics | ((self.fr_users_per_wave & 0xf) << 8) | ((self.sg_heavy_min & 0xfff) << 12), use_fr=use_fr, row_order=order,
                          rescored=None, row_order_grouped=(order is not None and use_fr and self._order_grouped),
                          use_sg=use_sg, use_sg_heavy=self.use_seg_heavy, flagged=flagged_fast)
            if fill:          # DENSE mode: short all-positive lists are completed with the zero-score columns, the rest stays flagged
                be.ops.dense_fill(d_row_ids, xb[0], xb[1], n_rows, int(W["col_lo"]), int(W["col_hi"]), top_k, bool(filter_interacted),
                                  ids, sc, aux, cnt, flagged_fast, flagged)
            if flag_words:                # one download: the lists and the counter
                h = pack.cpu().numpy()
                n_flag = int(h[2 * nk + cap])
                if n_flag == 0:
                    ids._rtrec_host = h
            else:
                n_flag = int(flagged[0].item())
            if self.rescored is not None:
                self.rescored.fill_(n_flag)
            if n_flag:
                # exact ties: those rows again, against the tiled layout (built now) -- its exact-tie pass orders them
                lay = self._layout(compact=not dense_fast, top_k=top_k)
                rows_f = flagged[1:1 + n_flag].long()
                sub_ids = d_row_ids[rows_f].contiguous() if d_row_ids is not None else rows_f.to(torch.int32)
                t_ids, t_sc, t_aux = (be.empty((n_flag, top_k), dt) for dt in (torch.int32, torch.float32, torch.int32))
                t_cnt = be.empty((n_flag,), torch.int32)
                need = be.score_workspace_bytes(n_flag, lay["n_tiles"], top_k)
                if self._score_ws.numel() < need:
                    self._score_ws = be.empty((need,), torch.uint8)
                be.score_topk(n_flag, sub_ids, xb, W["n_items"], W["col_lo"], lay, d_col_rank, top_k, filter_interacted,
                              mode, False, t_ids, t_sc, None, t_aux, t_cnt, self._score_ws, use_fr=False, use_sg=False)
                ids[rows_f] = t_ids; sc[rows_f] = t_sc; aux[rows_f] = t_aux; cnt[rows_f] = t_cnt
            return ids, sc, sc64, aux, cnt
        lay = self._layout(compact=sparse, top_k=top_k)
        if lay is None:
            ids.fill_(-1); sc.fill_(float("-inf")); aux.zero_(); cnt.zero_()
            if sc64 is not None:
                sc64.fill_(float("-inf"))
            return ids, sc, sc64, aux, cnt
        need = be.score_workspace_bytes(n_rows, lay["n_tiles"], top_k)
        if self._score_ws is None or self._score_ws.numel() < need:
            self._score_ws = be.empty((need,), torch.uint8)
        if hip:
            if use_sg and lay.get("sg") is None:          # a small batch against a feature-row W: its segment form
                lay = dict(lay, sg=fast["sg"])
            order = self._row_order(d_row_ids, n_rows, xb, lay, allow_grouped=use_fr) if (use_fr or use_sg) else None
            self.last_score_path = "feature_rows" if use_fr else ("segments" if use_sg else "tiled")
            be.score_topk(n_rows, d_row_ids, xb, W["n_items"], W["col_lo"], lay, d_col_rank, top_k, filter_interacted,
                          mode, W["acc_f64"], ids, sc, sc64, aux, cnt, self._score_ws, timer=self.score_timer,
                          diagnostics=self.diagnostics | ((self.fr_users_per_wave & 0xf) << 8) | ((self.sg_heavy_min & 0xfff) << 12), use_fr=use_fr, row_order=order,
                          rescored=self.rescored,
                          row_order_grouped=(order is not None and use_fr and self._order_grouped), use_sg=use_sg,
                          use_sg_heavy=self.use_seg_heavy)
        else:
            be.score_topk(n_rows, d_row_ids, xb, W["n_items"], W["col_lo"], lay, d_col_rank, top_k, filter_interacted,
                          mode, W["acc_f64"], ids, sc, sc64, aux, cnt, self._score_ws)
        return ids, sc, sc64, aux, cnt

    F64_REFINE_MIN_VALUE = 1e-18     # ratings and weights at least this large: a product is a normal float32 number

    def _local_topk_f64(self, d_row_ids, n_rows: int, xb, top_k: int, filter_interacted: bool, d_col_rank, fast, use_fr: bool,
                        use_sg: bool, ids, sc, sc64, aux, cnt, mode: int = _native.TOPK_SPARSE, signed: bool = False):
        """SPARSE mode, float64 W with positive float32-valued weights, positive ratings: the float32 fast pass for top_k + 1
        columns, the float64 scores of those candidates (csrc/score_refine.hip), and the float64 tiled kernel for the rows
        either step flags (float32 ties; a top_k-th float64 score too close to what a column outside the list could reach;
        float64 ties).  DENSE mode likewise: its fast pass also flags the rows with fewer than top_k + 1 positive scores (every
        non-zero score is positive here), which the DENSE float64 tiled kernel then scores."""
        dense = mode == _native.TOPK_DENSE
        be, W = self.be, self._W
        torch = be.torch
        k1 = top_k + 1
        ids1, sc1 = be.empty((n_rows, k1), torch.int32), be.empty((n_rows, k1), torch.float32)
        aux1, cnt1 = be.empty((n_rows, k1), torch.int32), be.empty((n_rows,), torch.int32)
        flagged = be.empty((2 * n_rows + 2,), torch.int32)
        need = be.score_workspace_bytes(n_rows, 1, k1)
        if self._score_ws is None or self._score_ws.numel() < need:
            self._score_ws = be.empty((need,), torch.uint8)
        order = self._row_order(d_row_ids, n_rows, xb, fast, allow_grouped=use_fr)
        self.last_score_path = ("feature_rows" if use_fr else "segments") + "+f64"
        be.score_topk(n_rows, d_row_ids, xb, W["n_items"], W["col_lo"], fast, d_col_rank, k1, filter_interacted,
                      mode, False, ids1, sc1, None, aux1, cnt1, self._score_ws, timer=self.score_timer,
                      diagnostics=self.diagnostics | ((self.fr_users_per_wave & 0xf) << 8) | ((self.sg_heavy_min & 0xfff) << 12),
                      use_fr=use_fr, row_order=order, rescored=None,
                      row_order_grouped=(order is not None and use_fr and self._order_grouped),
                      use_sg=use_sg, use_sg_heavy=self.use_seg_heavy, flagged=flagged)
        wc_ptr, wc_row, wc_val = W["dw"].csc_arrays(torch)
        margin = 2.0 * (W["col_nnz_max"] + 2) * 2.0 ** -24
        be.ops.refine_topk_f64(d_row_ids, xb[0], xb[1], xb[2], n_rows, W["n_items"], wc_ptr, wc_row, wc_val, top_k, ids1, sc1, cnt1,
                               float(margin), self._f64_abs_slack(xb) if signed else None, ids, sc, sc64, cnt, flagged)
        aux.zero_()
        n_flag = int(flagged[0].item())
        if self.rescored is not None:
            self.rescored.fill_(n_flag)
        if n_flag:
            rows_f = torch.unique(flagged[1:1 + n_flag].long())
            n_f = int(rows_f.numel())
            lay = self._layout(compact=not dense, top_k=top_k)
            sub_ids = d_row_ids[rows_f].contiguous() if d_row_ids is not None else rows_f.to(torch.int32)
            t_ids, t_sc, t_aux = (be.empty((n_f, top_k), dt) for dt in (torch.int32, torch.float32, torch.int32))
            t_sc64, t_cnt = be.empty((n_f, top_k), torch.float64), be.empty((n_f,), torch.int32)
            need = be.score_workspace_bytes(n_f, lay["n_tiles"], top_k)
            if self._score_ws.numel() < need:
                self._score_ws = be.empty((need,), torch.uint8)
            be.score_topk(n_f, sub_ids, xb, W["n_items"], W["col_lo"], lay, d_col_rank, top_k, filter_interacted,
                          mode, True, t_ids, t_sc, t_sc64, t_aux, t_cnt, self._score_ws, use_fr=False, use_sg=False)
            ids[rows_f] = t_ids; sc[rows_f] = t_sc; sc64[rows_f] = t_sc64; aux[rows_f] = t_aux; cnt[rows_f] = t_cnt
        return ids, sc, sc64, aux, cnt

    ROW_ORDER_MIN = 2048        # batches below this are one or two waves of jobs: nothing to level
    GROUPED_ORDER_MIN = 32768   # the pattern-grouped order (~1 ms to compute, kept per row set) pays for bulk passes only
    # Batches below this many rows go to the tiled-CSR kernel.  Since the feature-row kernel has its 4- and 2-users-per-wave
    # forms (chosen from the batch size inside rtrec_slim_score_topk) it is ahead at every batch size, one user included
    # (tools/score_batch_sweep.py: DESIGN.md section 3.1); the threshold is kept for A/B runs and the tests.
    FR_MIN_ROWS = 1
    SG_RELABEL_FRACTION = 0.2        # the cluster labels that order the segment layout's columns are kept until W's nnz has
                                     # moved by this much (they decide locality only, never a result; label propagation is 3.4 ms)
    FR_SMALL_BATCH = 513              # batches below this many rows are scored from the segment form (_small_batch_layout)
    FR_MAX_TOP_K = 15           # kFrMaxKk - 1 of csrc/score.hip: a list of top_k + 1 entries fits one 16-lane DPP row
    SG_MAX_TOP_K = 63           # kSgMaxKk - 1 of csrc/score_seg.hip.h: the list of top_k + 1 entries is one register across the lanes
    FR_TILE_COLS = 256          # columns per tile of the feature-row layout (128: the narrow kernels, kept for A/B and tests)
    pattern_order = settings.raw("RTREC_AMD_PATTERN_ORDER", "1") != "0"
    rescored = None             # optional int32[1] device tensor: rows the exact-tie pass re-scored in the last call

    def _grouped_order(self, lay, n_rows: Optional[int] = None) -> bool:
        """Pattern-sorted work order with eight consecutive rows per wave: for the STREAMING feature-row layout (C3:
        2.37 -> 2.26 ms).  The resident layout keeps the longest-first order dealt out in strides: its waves claim jobs
        on their own, and evenly mixed jobs matter more there than small unions (C2: 0.41 ms against 0.57 ms grouped)."""
        host = (lay or {}).get("fr_host") or {}
        return bool(self.pattern_order and host and not host.get("fr_resident")
                    and (n_rows is None or n_rows >= self.GROUPED_ORDER_MIN))

    ROW_ORDER_GIANTS = 32            # pattern-grouped order: at most this many giant rows are spread over the head ...
    ROW_ORDER_GIANT_LEN = 4096       # ... rows with more entries than this

    def _row_order(self, d_row_ids, n_rows: int, xb, lay=None, allow_grouped: bool = True):
        """Work order for the feature-row kernel (rtrec_score_opts.d_row_order).  Default: the batch's rows by descending
        length.  Streaming layout (_grouped_order): a wave sweeps, per tile, the UNION of the rows of W its eight users
        rate, so users with the same rated feature items should share a wave -- the rows are sorted by their feature-row
        pattern as one big integer, the row of W that holds a weight in the most tiles most significant, descending
        (heavy patterns first: the tail of the launch is light): 13 % fewer swept rows on C3.  A function of X, the row set and
        the layout only: kept with the resident X and reused while the same row-id tensor is scored against the same
        layout (bulk scoring, bench.py); small batches go in the order given.
        `allow_grouped=False` (the segment kernels): always the length order -- they are told "longest first" and stop at the
        first short user (a pattern-grouped order there left long users unscored); such a call neither reads
        nor writes the feature-row entries of the cache (its entries carry no layout)."""
        if n_rows < self.ROW_ORDER_MIN:
            self._order_grouped = False
            return None
        torch = self.be.torch
        resident = self._X.get("rptr") is xb[0]
        fr_host = (lay or {}).get("fr_host") if allow_grouped else None
        # One entry per (row-id tensor, layout), matched by IDENTITY: the entry keeps the tensor and the layout alive, so a
        # recycled address can never stand for another row set; a temporary row tensor simply misses.
        # The pattern-grouped order costs milliseconds to compute: a row set gets it the SECOND time it is scored (a caller
        # that keeps its row tensor -- bulk scoring, bench.py); a row set seen once (an API batch) runs in the length order.
        cache = self._X.setdefault("_orders", []) if resident else None
        want_grouped = bool(fr_host is not None and lay.get("fr_map") is not None and self._grouped_order(lay, n_rows))
        seen = None
        self._order_grouped = False
        if cache is not None:
            for ent in cache:
                if (ent[0] is d_row_ids and ent[1] == (None if d_row_ids is None else d_row_ids._version) and ent[2] == n_rows
                        and ent[3] is fr_host):
                    if ent[5] or not want_grouped:
                        self._order_grouped = ent[5]
                        return ent[4]
                    seen = ent
                    break
        grouped = want_grouped and (seen is not None or d_row_ids is None)
        ptr, col = xb[0], xb[1]
        if d_row_ids is None:
            rows = None
            lens = ptr[1:n_rows + 1] - ptr[:n_rows]
        else:
            rows = d_row_ids[:n_rows].long().clamp_(0, ptr.shape[0] - 2)
            lens = ptr[rows + 1] - ptr[rows]
        if not grouped:
            order = torch.argsort(lens, descending=True, stable=True).to(torch.int32)
        else:
            R = int(fr_host["fr_rows"])
            tr = fr_host["fr_rows_of_tile"].view(np.uint64).reshape(-1, 2)
            tiles_of_row = np.array([sum((int(tr[t, f // 64]) >> (f % 64)) & 1 for t in range(tr.shape[0])) for f in range(R)])
            rank = np.empty(R, dtype=np.int64)
            rank[np.argsort(-tiles_of_row, kind="stable")] = np.arange(R)          # 0 = the row in the most tiles
            bit_of_row = torch.from_numpy((R - 1) - rank).to(col.device)              # its bit in the pattern integer
            n_words = -(-R // 60)
            lens64 = lens.long()
            ent_row = torch.repeat_interleave(torch.arange(n_rows, device=col.device), lens64)
            if rows is None:
                ent_col = col[int(ptr[0]):int(ptr[n_rows])].long()
            else:
                within = torch.arange(ent_row.shape[0], device=col.device) - torch.repeat_interleave(
                    torch.cumsum(lens64, 0) - lens64, lens64)
                ent_col = col[ptr[rows].long()[ent_row] + within].long()
            ok = ent_col < lay["fr_map"].shape[0]
            f = torch.full_like(ent_col, -1)
            f[ok] = lay["fr_map"][ent_col[ok]].long()
            keep = f >= 0
            b = bit_of_row[f[keep]]
            words = torch.zeros(n_rows * n_words, dtype=torch.int64, device=col.device)
            words.index_add_(0, ent_row[keep] * n_words + b // 60, torch.ones_like(b) << (b % 60))   # a row's items are distinct
            words = words.view(n_rows, n_words)
            order = torch.arange(n_rows, device=col.device)
            for w in range(n_words):                                  # least significant word first, stable sorts
                order = order[torch.argsort(words[order, w], descending=True, stable=True)]
            order = spread_giant_rows(torch, order, lens64, self.ROW_ORDER_GIANTS, self.ROW_ORDER_GIANT_LEN)
            order = order.to(torch.int32)
        if cache is not None:
            if seen is not None:
                cache[:] = [ent for ent in cache if ent is not seen]       # (by identity: the entries hold tensors)
            if len(cache) >= 32:
                cache.clear()
            cache.append((d_row_ids, None if d_row_ids is None else d_row_ids._version, n_rows, fr_host, order, grouped))
            self._X["_order"] = order          # the most recent one (bench.py's bounds model reads it)
        self._order_grouped = grouped
        return order

    def score_topk_device(self, row_ids: Optional[np.ndarray], n_rows: int, top_k: int, filter_interacted: bool,
                          mode: int, col_rank: Optional[np.ndarray] = None, xb=None, d_rows=None, host: bool = False,
                          candidates: Optional[np.ndarray] = None, with_scores: bool = True):
        """Device tensors (ids, scores, counts) of the GLOBAL top-k for the given rows of X
        (or of the CSR batch `xb` = (ptr, col, val) device tensors).  `d_rows` may pass the row ids
        as a device tensor that is already resident (bench.py reuses it across steps).

        Multi-GPU: every rank scores the rows against its own column shard and packs its per-shard lists
        into ONE int32 record per user -- [scores | ids | first-touch aux | count] (float64 scores in
        front when W is float64).  The exchange is sized for xGMI's point-to-point mesh rather than for
        a switch: an ALL-TO-ALL hands rank r the records of ITS slice of the users from every shard
        (B/G * 124 B from each peer instead of all B * 124 B of an all-gather), merge_topk_kernel merges
        that slice in place through strides, and only the final k-lists (84 B per user) are all-gathered.
        Rows are processed in chunks whose exchanges (RCCL, asynchronous on its own stream) overlap the
        scoring kernel of the next chunk.
        with_scores=False (row shards): only ids and counts travel -- what `recommend_batch` returns (the reference hands out
        item ids, base.py:188-269) -- 44 instead of 84 bytes per user at k = 10; the score tensor is then None."""
        be = self.be
        if not self._W:
            raise RuntimeError("Model must be fitted before calling batch_recommend.")
        if xb is None:
            xb = (self._X["rptr"], self._X["rcol"], self._X["rval"])
        up = getattr(be, "to_dev_small", be.to_dev)
        if d_rows is None and row_ids is not None:
            d_rows = up(np.asarray(row_ids, dtype=np.int32))
        if mode == _native.TOPK_CANDIDATES and candidates is not None:
            direct = self._candidates_direct(d_rows, n_rows, xb, top_k, candidates)
            if direct is not None:
                return direct
            if col_rank is None:          # the bulk form: a rank per column of W (the later of two equal candidates wins)
                col_rank = np.full(self._W["n_items"], -1, dtype=np.int32)
                col_rank[np.asarray(candidates, dtype=np.int64)] = np.arange(len(candidates), dtype=np.int32)
        d_rank = up(np.asarray(col_rank, dtype=np.int32)) if col_rank is not None else None
        if self.world_size == 1 and not self.force_exchange:
            ids, sc, sc64, aux, cnt = self._local_topk(d_rows, n_rows, xb, top_k, filter_interacted, mode, d_rank, host=host)
            return ids, sc, cnt
        import torch.distributed as dist
        torch = be.torch
        G, k = self.world_size, top_k
        f64 = bool(self._W["acc_f64"])
        if d_rows is None:
            d_rows = torch.arange(n_rows, dtype=torch.int32, device=xb[0].device)
        if self.score_shard == "rows":
            return self._score_row_sharded(d_rows, n_rows, xb, k, filter_interacted, mode, d_rank, with_scores)
        # chunks exist so that one chunk's exchange overlaps the next chunk's kernel; a chunk must still be large enough to
        # fill the chip (GATHER_CHUNK_ROWS), so passes of up to ~200k rows are one chunk
        per = max(1, int(self.gather_chunk_rows))
        n_chunks = max(1, min(MAX_GATHER_CHUNKS, (n_rows + per // 2) // per))
        per = -(-n_rows // n_chunks)
        # exchanged record (int32 words): [2k float64 scores]? | k scores | k ids | k aux | count | pad to even
        o_sc = 2 * k if f64 else 0
        width = o_sc + 3 * k + 1
        width += width & 1
        fwidth = 2 * k + 1          # final record: k scores | k ids | count | pad to even
        fwidth += fwidth & 1
        o_ids = be.empty((n_rows, k), torch.int32)
        o_scs = be.empty((n_rows, k), torch.float32)
        o_cnt = be.empty((n_rows,), torch.int32)
        pending = []
        for c in range(n_chunks):
            a, b = c * per, min(n_rows, (c + 1) * per)
            m = b - a
            if m <= 0:
                break
            ids, sc, sc64, aux, cnt = self._local_topk(d_rows[a:b], m, xb, k, filter_interacted, mode, d_rank)
            parts = ([sc64.view(torch.int32)] if f64 else []) + [sc.view(torch.int32), ids, aux.view(torch.int32),
                                                                 cnt.view(m, 1)]
            if (o_sc + 3 * k + 1) & 1:
                parts.append(torch.zeros((m, 1), dtype=torch.int32, device=ids.device))
            q = -(-m // G)                      # rows of this chunk each rank merges
            if q * G == m:
                packed = torch.cat(parts, dim=1)
            else:                               # pad to G equal slices; padded rows carry count 0
                packed = torch.zeros((q * G, width), dtype=torch.int32, device=ids.device)
                packed[:m] = torch.cat(parts, dim=1)
            recv = be.empty((G * q, width), torch.int32)
            work = dist.all_to_all_single(recv, packed, group=self.group, async_op=True)
            pending.append((a, b, q, recv, packed, work))
        finals = []
        for a, b, q, recv, packed, work in pending:
            work.wait()
            g3 = recv.view(G, q, width)         # [source shard, row of my slice, record]
            g_sc = g3[:, :, o_sc:o_sc + k].view(torch.float32)
            g_ids = g3[:, :, o_sc + k:o_sc + 2 * k]
            g_aux = g3[:, :, o_sc + 2 * k:o_sc + 3 * k]
            g_cnt = g3[:, :, o_sc + 3 * k]
            g_sc64 = recv.view(torch.float64).view(G, q, width // 2)[:, :, :k] if f64 else None
            s_ids = be.empty((q, k), torch.int32)
            s_sc = be.empty((q, k), torch.float32)
            s_cnt = be.empty((q,), torch.int32)
            be.merge_topk(q, G, k, g_ids, g_sc, g_sc64, g_aux, g_cnt, s_ids, s_sc, s_cnt)
            fparts = [s_sc.view(torch.int32), s_ids, s_cnt.view(q, 1)]
            if (2 * k + 1) & 1:
                fparts.append(torch.zeros((q, 1), dtype=torch.int32, device=s_ids.device))
            fin = torch.cat(fparts, dim=1)
            out = be.empty((G * q, fwidth), torch.int32)
            work2 = dist.all_gather_into_tensor(out, fin, group=self.group, async_op=True)
            finals.append((a, b, out, fin, work2))
        for a, b, out, fin, work2 in finals:
            work2.wait()
            m = b - a
            o_scs[a:b] = out[:m, :k].view(torch.float32)
            o_ids[a:b] = out[:m, k:2 * k]
            o_cnt[a:b] = out[:m, 2 * k]
        return o_ids, o_scs, o_cnt

    def _score_row_sharded(self, d_rows, n_rows: int, xb, k: int, filter_interacted: bool, mode: int, d_rank,
                           with_scores: bool = True):
        """score_shard == "rows": this rank scores rows r, r+G, r+2G, ... of the batch (strided, so that
        every rank gets the same mix of heavy and light users whatever their order) against the whole W
        -- no merge step: its lists are final -- and the final records [k scores | k ids | count] of all
        ranks are all-gathered, 84 B per user at k = 10."""
        import torch.distributed as dist
        be = self.be
        torch = be.torch
        G = self.world_size
        q = -(-n_rows // G)
        # this rank's slice of the batch, kept while the same row tensor is scored again (its work order is cached by it)
        # (matched by the IDENTITY of the row tensor, which the entry keeps alive: an address recycled by the allocator for
        # another call's row ids must never hit -- ADVICE round 2; a temporary tensor simply misses and is sliced again)
        slices = self._X.setdefault("_row_slices", [])
        mine = None
        for ent in slices:
            if ent[0] is d_rows and ent[1] == d_rows._version and ent[2] == (n_rows, self.rank, G):
                mine = ent[3]
        if mine is None:
            mine = d_rows[self.rank::G].contiguous()
            if len(slices) >= 8:
                slices.clear()
            slices.append((d_rows, d_rows._version, (n_rows, self.rank, G), mine))
        m = int(mine.shape[0])
        # Slots (rows of a rank's slice) are cut into chunks so that the all-gather of chunk c (RCCL, asynchronous) runs
        # beside the kernel of chunk c + 1 -- the same overlap the column path has; one blocking gather after the whole
        # local pass used to stand behind the kernel (C4 at 8 ranks: 73.5 MB received per rank after a 0.6 ms kernel,
        # VERDICT round 4).  A chunk keeps at least row_chunk_rows slots: a launch must still fill the chip.
        per = max(1, int(self.row_chunk_rows))
        n_chunks = max(1, min(MAX_GATHER_CHUNKS, q // per))
        bounds = [q * c // n_chunks for c in range(n_chunks + 1)]
        chunk_rows = None                    # this rank's row tensors per chunk: kept with the slice (their work orders are cached by identity)
        for ent in slices:
            if ent[3] is mine and len(ent) > 4 and ent[4][0] == bounds:
                chunk_rows = ent[4][1]
        if chunk_rows is None:
            chunk_rows = [mine[min(bounds[c], m):min(bounds[c + 1], m)] if n_chunks > 1 else mine for c in range(n_chunks)]
            for i_, ent in enumerate(slices):
                if ent[3] is mine:
                    slices[i_] = ent[:4] + ((bounds, chunk_rows),)
        pending = []
        for c in range(n_chunks):
            qc = bounds[c + 1] - bounds[c]
            rows_c = chunk_rows[c]
            mc = int(rows_c.shape[0])
            # every rank's (ids | scores | counts) buffer, laid out for qc slots, is gathered as it is; slots a short slice does
            # not have (global row index >= n_rows) are cut off below, so they may hold anything
            width = (2 * k + 1) * qc
            if mc > 0:
                ids, sc, sc64, aux, cnt = self._local_topk(rows_c, mc, xb, k, filter_interacted, mode, d_rank, pad_rows=qc)
                fin = ids._base if ids._base is not None else ids          # the flat buffer the three are views of
                fin = fin.reshape(-1)
            else:
                fin = be.empty((width,), torch.int32)
            if with_scores:
                out = be.empty((G * width,), torch.int32)
                work = dist.all_gather_into_tensor(out, fin, group=self.group, async_op=(n_chunks > 1))
                pending.append((qc, out, None, fin, (work,)))
            else:           # ids and counts only: two contiguous pieces of the same buffer, two collectives
                out_i, out_c = be.empty((G * qc * k,), torch.int32), be.empty((G * qc,), torch.int32)
                w1 = dist.all_gather_into_tensor(out_i, fin[:qc * k], group=self.group, async_op=(n_chunks > 1))
                w2 = dist.all_gather_into_tensor(out_c, fin[2 * qc * k:2 * qc * k + qc], group=self.group, async_op=(n_chunks > 1))
                pending.append((qc, out_i, out_c, fin, (w1, w2)))
        o_ids_p, o_sc_p, o_cnt_p = [], [], []
        for qc, out, out_c, fin, works in pending:
            for work in works:
                if work is not None:
                    work.wait()
            # slot i of rank p <- row i*G + p of the batch: [slot, rank] order is the batch order
            if with_scores:
                out = out.view(G, (2 * k + 1) * qc)
                o_ids_p.append(out[:, :qc * k].view(G, qc, k).transpose(0, 1).reshape(G * qc, k))
                o_sc_p.append(out[:, qc * k:2 * qc * k].view(G, qc, k).transpose(0, 1).reshape(G * qc, k))
                o_cnt_p.append(out[:, 2 * qc * k:].t().reshape(G * qc))
            else:
                o_ids_p.append(out.view(G, qc, k).transpose(0, 1).reshape(G * qc, k))
                o_cnt_p.append(out_c.view(G, qc).t().reshape(G * qc))
        o_ids = (o_ids_p[0] if n_chunks == 1 else torch.cat(o_ids_p))[:n_rows]
        o_cnt = (o_cnt_p[0] if n_chunks == 1 else torch.cat(o_cnt_p))[:n_rows]
        o_sc = None
        if with_scores:
            o_sc = (o_sc_p[0] if n_chunks == 1 else torch.cat(o_sc_p))[:n_rows].view(torch.float32)
        return o_ids, o_sc, o_cnt

    MAX_TOP_K = 1023            # kMaxTopK of csrc/score.hip
    MAX_MERGE_CANDIDATES = 1024  # per-tile lists of one row the merge kernel takes: n_tiles * (top_k + 1)

    def topk_supported(self, top_k: int, mode: int) -> bool:
        """Whether the fused score + top-k kernels serve this request (rtrec_slim_score_topk's limits);
        callers fall back to score rows from the device + a host selection otherwise."""
        if top_k > self.MAX_TOP_K:
            return False
        compact = mode == _native.TOPK_SPARSE
        W = self._W
        if compact and "n_active" not in W:
            # a shard as wide as ALL its columns is an upper bound: when even that fits the merge, no count is needed
            full = W["col_hi"] - W["col_lo"]
            if -(-max(full, 1) // self._tile_width(False, top_k)) * (top_k + 1) <= self.MAX_MERGE_CANDIDATES:
                return True
        if compact and "n_active" not in W and not W["acc_f64"] and W["dw"].nnz > 0 and W["col_hi"] > W["col_lo"]:
            # the number of columns that hold a weight (what every compacted layout is as wide as): counted from the sorted
            # COO -- no layout is built just to answer this question (the feature-row form alone is 2 ms)
            cols = W["dw"].cols
            inside = cols[(cols >= W["col_lo"]) & (cols < W["col_hi"])]
            W["n_active"] = int((inside[1:] != inside[:-1]).sum()) + 1 if inside.numel() else 0
        if compact and "n_active" not in self._W:
            lay = self._layout(compact=True, top_k=top_k)
            return lay is None or lay["n_tiles"] * (top_k + 1) <= self.MAX_MERGE_CANDIDATES
        width = self._W["n_active"] if compact else self._W["col_hi"] - self._W["col_lo"]
        return -(-max(width, 1) // self._tile_width(compact, top_k)) * (top_k + 1) <= self.MAX_MERGE_CANDIDATES

    CANDS_DIRECT_MAX = 4096             # candidates of a request ranked by rtrec_slim_score_candidates ...
    CANDS_DIRECT_MAX_PAIRS = 1 << 21    # ... while rows x candidates stays a request, not a bulk pass ...
    CANDS_DIRECT_BULK = 1024            # ... and BULK calls with up to this many candidates (round 4): the direct kernel costs
                                        # per (row, candidate column entry), the tiled kernel a pass over all of W whatever the
                                        # list -- all 138,493 c3s users: 20 candidates 2.3 against 18.8 ms, 200: 4.6 / 21.1, 1,000:
                                        # 15.8 / 20.2, 4,096: 84.9 / 20.1, same ids (tools/cands_bench.py)

    def _candidates_direct(self, d_rows, n_rows: int, xb, top_k: int, candidates: np.ndarray):
        """CANDIDATES mode for a request-sized call (csrc/score_cands.hip): the candidates' scores straight from W's CSC
        columns, no pass over all columns and no n_items-sized rank array.  None: not applicable (bulk call, a backend or a
        W it does not serve, several column shards) -- the caller takes the tiled kernel."""
        be, W = self.be, self._W
        n_c = int(len(candidates))
        dw: DeviceWeights = W["dw"]
        if (not isinstance(be, HipBackend) or not self.cands_direct or n_c == 0 or n_c > self.CANDS_DIRECT_MAX
                or (n_rows * n_c > self.CANDS_DIRECT_MAX_PAIRS and n_c > self.CANDS_DIRECT_BULK)
                or (self.world_size > 1 and self.score_shard != "rows")
                or self.force_exchange or dw.lossy or dw.nnz == 0 or top_k > n_c):
            return None
        cands = np.asarray(candidates)
        if int(cands.min()) < 0 or int(cands.max()) >= W["n_items"]:
            return None                       # (the rank-array path raises like the reference's W[:, candidates])
        torch = be.torch
        f64 = bool(W["acc_f64"])
        d_c = be.to_dev_small(cands.astype(np.int32))
        wc_ptr, wc_row, wc_val = dw.csc_arrays(torch)
        nk = n_rows * top_k
        pack = be.empty((2 * nk + n_rows,), torch.int32)
        ids = pack[:nk].view(n_rows, top_k)
        sc = pack[nk:2 * nk].view(torch.float32).view(n_rows, top_k)
        cnt = pack[2 * nk:]
        ids._rtrec_pack = pack
        sc64 = be.empty((n_rows, top_k), torch.float64) if f64 else None
        be.ops.score_candidates(d_rows, xb[0], xb[1], xb[2], n_rows, W["n_items"], wc_ptr, wc_row, wc_val, d_c, top_k, bool(f64),
                                ids, sc, sc64, cnt)
        self.last_score_path = "candidates_direct"
        return ids, sc, cnt

    def _check_rows(self, row_ids: np.ndarray) -> None:
        """Row ids index the resident X: anything outside [0, n_users) would be an out-of-bounds device read
        (the reference's scipy indexing raises IndexError for it)."""
        if len(row_ids) and (int(row_ids.min()) < 0 or int(row_ids.max()) >= self.n_users):
            bad = row_ids[(row_ids < 0) | (row_ids >= self.n_users)]
            raise IndexError(f"row index ({int(bad[0])}) out of range for the resident interaction matrix "
                             f"with {self.n_users} rows")

    def rows_csr(self, row_ids: Sequence[int]) -> sp.csr_matrix:
        """Rows of the resident X as a host CSR (download of just those rows)."""
        row_ids = np.asarray(row_ids, dtype=np.int64)
        self._check_rows(row_ids)
        torch = self.be.torch
        X = self._X
        d_rows = self.be.to_dev(row_ids)
        beg, end = X["rptr"][d_rows].long(), X["rptr"][d_rows + 1].long()
        cnt = end - beg
        indptr = np.zeros(len(row_ids) + 1, dtype=np.int64)
        indptr[1:] = np.cumsum(cnt.cpu().numpy())
        total = int(indptr[-1])
        if total:
            off = torch.arange(total, device=d_rows.device) - torch.repeat_interleave(
                torch.as_tensor(indptr[:-1], device=d_rows.device), cnt) + torch.repeat_interleave(beg, cnt)
            cols, vals = X["rcol"][off].cpu().numpy(), X["rval"][off].cpu().numpy()
        else:
            cols, vals = np.empty(0, np.int32), np.empty(0, np.float32)
        return sp.csr_matrix((vals, cols, indptr), shape=(len(row_ids), self.n_items))

    def recommend_rows(self, row_ids: Sequence[int], top_k: int = 10, filter_interacted: bool = True,
                       mode: int = _native.TOPK_SPARSE, col_rank: Optional[np.ndarray] = None,
                       candidates: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Top-k for rows of the resident X.  Returns numpy (ids[B,k], scores[B,k], counts[B]).  CANDIDATES mode: either the
        rank array `col_rank` (n_items entries, -1 = not a candidate) or the list `candidates` itself."""
        row_ids = np.asarray(row_ids, dtype=np.int64)
        if len(row_ids) == 0:
            return (np.empty((0, top_k), np.int32), np.empty((0, top_k), np.float32), np.empty((0,), np.int32))
        self._check_rows(row_ids)
        row_ids = row_ids.astype(np.int32)
        return self._download(*self.score_topk_device(row_ids, len(row_ids), top_k, filter_interacted, mode, col_rank, host=True,
                                                      candidates=candidates))

    @staticmethod
    def _download(ids, sc, cnt) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(ids, scores, counts) device tensors -> numpy; one copy when they are the views _local_topk hands out."""
        pack = getattr(ids, "_rtrec_pack", None)
        if pack is None:
            return ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()
        n, k = ids.shape
        h = getattr(ids, "_rtrec_host", None)
        if h is None:
            h = pack.cpu().numpy()
        return h[:n * k].reshape(n, k), h[n * k:2 * n * k].view(np.float32).reshape(n, k), h[2 * n * k:2 * n * k + n]

    def recommend_csr(self, Xb: sp.csr_matrix, top_k: int = 10, filter_interacted: bool = True,
                      mode: int = _native.TOPK_SPARSE, col_rank: Optional[np.ndarray] = None,
                      candidates: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """Top-k for the rows of a host CSR batch (the SLIMElastic.recommend_batch boundary)."""
        be = self.be
        B = Xb.shape[0]
        if B == 0:
            return (np.empty((0, top_k), np.int32), np.empty((0, top_k), np.float32), np.empty((0,), np.int32))
        if not Xb.has_sorted_indices:
            Xb = Xb.sorted_indices()
        xb = (be.to_dev(np.asarray(Xb.indptr, dtype=np.int32)), be.to_dev(np.asarray(Xb.indices, dtype=np.int32)),
              be.to_dev(np.asarray(Xb.data, dtype=np.float32)))
        return self._download(*self.score_topk_device(None, B, top_k, filter_interacted, mode, col_rank, xb=xb, candidates=candidates))

    # ------------------------------------------------------------------------------ score vectors
    def predict_csr(self, Xb: sp.csr_matrix) -> np.ndarray:
        """Dense score rows Xb . W (float32, or float64 for a float64 W) for a host CSR batch:
        the SLIMElastic.predict* boundary.  [B, n_items]; with several ranks each computes its own
        column block and the blocks are concatenated on the host."""
        be, W = self.be, self._W
        if not W:
            raise RuntimeError("Model must be fitted before calling predict.")
        torch = be.torch
        B = Xb.shape[0]
        if not Xb.has_sorted_indices:
            Xb = Xb.sorted_indices()
        xb = (be.to_dev(np.asarray(Xb.indptr, dtype=np.int32)), be.to_dev(np.asarray(Xb.indices, dtype=np.int32)),
              be.to_dev(np.asarray(Xb.data, dtype=np.float32)))
        lay = self._layout(compact=False)
        dt = torch.float64 if W["acc_f64"] else torch.float32
        n_local = W["col_hi"] - W["col_lo"]
        out = be.zeros((B, max(n_local, 1)), dt)
        if lay is not None and B > 0:
            be.score_rows(B, None, xb, W["n_items"], W["col_lo"], lay, W["acc_f64"], out)
        block = out.cpu().numpy()[:, :n_local]
        if self.world_size == 1 or self.score_shard == "rows":      # rows mode: W is not sharded
            return block
        import torch.distributed as dist
        parts: List[Any] = [None] * self.world_size
        dist.all_gather_object(parts, block, group=self.group)
        return np.concatenate(parts, axis=1)

    # ------------------------------------------------------------------------------ similar
    def similar_items(self, queries: Sequence[int], top_k: int = 10) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        be, W = self.be, self._W
        if not W:
            raise RuntimeError("Model must be fitted before calling similar_items.")
        torch = be.torch
        q = np.asarray(queries, dtype=np.int32)
        n = len(q)
        if n == 0:
            return (np.empty((0, top_k), np.int32), np.empty((0, top_k), np.float32), np.empty((0,), np.int32))
        d_q = be.to_dev(q)
        ids = be.empty((n, top_k), torch.int32)
        sc = be.empty((n, top_k), torch.float32)
        cnt = be.empty((n,), torch.int32)
        cptr, crow, cval = W["dw"].csc_arrays(torch)
        be.similar_topk(d_q, {"cptr": cptr, "crow": crow, "cval": cval}, top_k, ids, sc, cnt)
        if getattr(W["dw"], "shard", None) is not None and self.world_size > 1:
            # W is column-sharded: column j is held by its owner alone ("similar_items(j) is served by the owner of
            # column j"); every rank answers its own queries and one all-gather of [n, 2k + 1] words brings them together
            import torch.distributed as dist
            pack = torch.cat([ids, sc.view(torch.int32), cnt.view(n, 1)], dim=1).contiguous()
            allp = be.empty((self.world_size * n, 2 * top_k + 1), torch.int32)
            dist.all_gather_into_tensor(allp, pack, group=self.group)
            bounds = np.array([shard_bounds(W["n_items"], self.world_size, r)[1] for r in range(self.world_size)])
            owner = np.searchsorted(bounds, np.clip(q.astype(np.int64), 0, W["n_items"] - 1), side="right")
            pick = be.to_dev(owner.astype(np.int64) * n + np.arange(n, dtype=np.int64))
            mine = allp[pick]
            ids, sc, cnt = mine[:, :top_k], mine[:, top_k:2 * top_k].view(torch.float32), mine[:, 2 * top_k]
        return ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()


def coefficients_to_updates(targets: np.ndarray, items: np.ndarray, coef: np.ndarray, count: np.ndarray
                            ) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Flatten fit_columns output to COO triples (row item i, column target j, value), explicit
    zeros included, ready for merge_coefficients()."""
    n, cap = items.shape
    mask = np.arange(cap)[None, :] < count[:, None]
    rows = items[mask].astype(np.int64)
    cols = np.repeat(targets.astype(np.int64), count)
    return rows, cols, coef[mask]


def merge_coefficients(W_old: Optional[sp.csc_matrix], n_items: int, rows: np.ndarray, cols: np.ndarray,
                       vals: np.ndarray, dtype=np.float32) -> sp.csc_matrix:
    """The LIL write-back of slim_elastic.py:271-274 / 371-374 / 554-557 as one vectorised merge:
    start from the old matrix (resized to n_items), then for every fitted (i, j): a non-zero value
    overwrites W[i, j], an explicit zero deletes it; entries of column j that the new solution does
    not mention survive."""
    if W_old is not None and W_old.nnz:
        Wo = W_old.tocoo()
        o_rows, o_cols, o_vals = Wo.row.astype(np.int64), Wo.col.astype(np.int64), Wo.data.astype(dtype)
        o_key = o_cols * n_items + o_rows
        n_key = cols * n_items + rows
        keep = ~np.isin(o_key, n_key)
        rows_all = np.concatenate([o_rows[keep], rows])
        cols_all = np.concatenate([o_cols[keep], cols])
        vals_all = np.concatenate([o_vals[keep], vals.astype(dtype)])
    else:
        rows_all, cols_all, vals_all = rows, cols, vals.astype(dtype)
    nz = vals_all != 0
    W = sp.csc_matrix((vals_all[nz], (rows_all[nz], cols_all[nz])), shape=(n_items, n_items), dtype=dtype)
    W.sort_indices()
    return W
