"""Every RTREC_AMD_* environment switch of the Python host layer, in ONE table.

The C library reads no environment at all; the engine, the model layer and the serving shell read theirs only through
`settings.raw(name, default)`, which refuses a name that is not documented here.  All switches are A/B, test or tuning aids:
defaults are what the tests and benchmarks run, and -- except for the fit modes, which say so -- results do not depend on them.
`python -m rtrec_amd.settings` prints the table with the values in effect.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

# name -> (default as the code spells it, what it does)
TABLE: Dict[str, tuple] = {
    # library / build
    "RTREC_AMD_LIB": ("rtrec_amd/lib/librtrec_amd.so", "load another build of the same C-ABI (tools/ab_build.sh; diagnostic builds)"),
    # interaction store
    "RTREC_AMD_DEVICE_STORE": ("1", "0: never keep X resident on the GPU, every call exports from the host store"),
    "RTREC_AMD_DEVICE_INGEST": ("1", "0: bulk batches are reduced to distinct pairs by the host store, not on the device"),
    # scoring
    "RTREC_AMD_SCORE_SHARD": ("columns", "multi-GPU scoring division: item-column shards of W, or `rows` (user rows, W replicated)"),
    "RTREC_AMD_SHARD_W": ("0", "1: multi-GPU, column-sharded scoring: every rank fits and keeps only its own column block of W (no all-gather of the coefficients)"),
    "RTREC_AMD_FORCE_EXCHANGE": (None, "run the multi-GPU exchange with one rank (test aid)"),
    "RTREC_AMD_FEATURE_ROWS": ("1", "0: no feature-row kernel"),
    "RTREC_AMD_FR_USERS": ("0", "8 / 4 / 2: force the users-per-wave form of the feature-row kernel (0: from the batch size)"),
    "RTREC_AMD_FR_SMALL_BATCH": ("513", "batches below this many rows are scored from the segment form even when W has a feature-row form"),
    "RTREC_AMD_PATTERN_ORDER": ("1", "0: no pattern-grouped work order for the streaming feature-row layout"),
    "RTREC_AMD_SEG_LAYOUT": ("1", "0: no segment kernel for a general W (tiled-CSR kernel instead)"),
    "RTREC_AMD_SEG_CLUSTER": ("1", "0: segment layout in item-id column order instead of the clustered one"),
    "RTREC_AMD_SEG_HEAVY": ("1", "0: no workgroup-per-long-user pass"),
    "RTREC_AMD_SG_HEAVY_MIN": ("0", "v > 0: users of more than v - 1 items get a workgroup instead of a wave (0: from the pass size)"),
    "RTREC_AMD_SG_FORK": ("1", "0: the workgroup-per-long-user pass runs on the main stream, not beside the main kernel"),
    "RTREC_AMD_NATIVE_SEG_BUILD": ("1", "0: build the segment layout with tensor ops instead of csrc/seg_build.hip"),
    "RTREC_AMD_F64_REFINE": ("1", "0: a float64 W is scored by the float64 tiled kernel only, not by the float32 fast pass + float64 refine step"),
    "RTREC_AMD_CANDS_DIRECT": ("1", "0: request-sized CANDIDATES calls go through the tiled kernel like bulk ones"),
    "RTREC_AMD_DENSE_FILL": ("1", "0: DENSE mode's short fast-pass lists go to the tiled kernel instead of being completed in place (and column shards keep the tiled kernel)"),
    "RTREC_AMD_DENSE_FAST": ("1", "0: DENSE mode (string ids) is scored by the tiled kernel only, not by the fast pass + flagged rows"),
    "RTREC_AMD_LAZY_TILED": ("1", "0: build the tiled layout with every W instead of only when a call flags exact score ties"),
    "RTREC_AMD_ABLATE": ("0", "ablation bits forwarded as rtrec_score_opts.diagnostics (diagnostic builds only)"),
    # fit
    "RTREC_AMD_FIT_MODE": ("", "exact | gram | ...: force the fit mode (CHANGES results within the documented tolerance when not exact)"),
    "RTREC_AMD_GRAM": ("auto", "0 / force: Gram tracking in the fit kernel off / also for small calls"),
    "RTREC_AMD_GRAM_ITEMS": ("auto", "<n>: fix the size of the shared Gram matrix (auto: from a pilot feature selection)"),
    "RTREC_AMD_FIT_SLOTS": ("MAX_SLOTS", "work-queue slots of the single-wave fit kernel"),
    "RTREC_AMD_FIT_SCRATCH_GIB": ("FIT_SCRATCH_GIB", "scratch budget of the fit kernel"),
    "RTREC_AMD_FIT_HEAVY": ("FIT_HEAVY_TARGETS", "how many of the longest targets of a bulk call go to the multi-wave kernel"),
    "RTREC_AMD_FIT_HEAVY_SLOTS": ("per call", "work-queue slots of the multi-wave fit kernel"),
    "RTREC_AMD_FIT_HEAVY_MIN_ROWS": ("per call", "targets with at least this many users count as heavy"),
    "RTREC_AMD_ALLF_CAP": ("ALLF_OUTPUT_CAP", "output capacity per target of the all-features fit (K=None)"),
    "RTREC_AMD_LANE_MAX": (None, "fit kernel: longest column handled one entry per lane"),
    "RTREC_AMD_COLWALK_MIN": ("0", "fit kernel: column-walk threshold"),
    "RTREC_AMD_SCREEN_MIN": ("0", "fit kernel: screening threshold"),
    "RTREC_AMD_FOLD": ("", "chain | spec | spec-all: ordered dot products by the literal add chain (default), or by the binade-speculative fold for columns >= 512 / >= 64 entries (bit-identical results, A/B and tests)"),
    "RTREC_AMD_XTY_BATCH": ("1", "0: no one-pass X^T y for small fit calls"),
    "RTREC_AMD_DEBUG_XTY": (None, "print the one-pass X^T y decision"),
    # serving
    "RTREC_AMD_COALESCE_MS": ("1", "bounded wait of the /recommend request coalescer (0: only what queued behind the lock; < 0: off)"),
}


def raw(name: str, default: Any = None) -> Optional[str]:
    """The variable's string value, or `default` (the call site's own default, which TABLE documents)."""
    if name not in TABLE:
        raise KeyError(f"{name} is not a documented rtrec_amd setting (rtrec_amd/settings.py)")
    return os.environ.get(name, default)


def describe() -> str:
    rows = []
    for name, (default, doc) in TABLE.items():
        cur = os.environ.get(name)
        rows.append(f"{name:32s} default {str(default):22s} {'= ' + cur if cur is not None else '':14s} {doc}")
    return "\n".join(rows)


if __name__ == "__main__":
    print(describe())
