#!/usr/bin/env python3
"""Per-target timeline of the fit kernel (rtrec_slim_fit_columns_opt).

Runs the bulk fit of a bench workload on cuda:0 with the device-side trace enabled and prints
where the time goes: X^T y / feature selection vs coordinate descent, the longest targets
(critical path), and how many work-queue slots are busy over the kernel's lifetime.

    python tools/fit_trace.py --workload c3 [--out gpurun_out/fit_trace_c3.json]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--out", default=None)
    ap.add_argument("--shards", type=int, default=1, help="fit only shard 0 of this many column shards")
    ap.add_argument("--mode", default="exact", choices=["exact", "gram", "shuffle"])
    args = ap.parse_args()
    import torch
    from rtrec_amd.engine import SlimEngine, shard_bounds
    from rtrec_amd.synth import workload_matrix

    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    lo, hi = shard_bounds(I, args.shards, 0)
    eng.fit_columns(np.arange(lo, hi), nn_feature_selection=K, mode=args.mode)   # warm-up (allocations, sqnorms, Gram matrix)
    torch.cuda.synchronize()
    t0 = time.time()
    tg, items, coef, count, n_iter = eng.fit_columns(np.arange(lo, hi), nn_feature_selection=K, trace=True, mode=args.mode)
    torch.cuda.synchronize()
    wall = time.time() - t0
    tr = eng.last_fit_stats["trace"].astype(np.float64)
    tick = 1e-8   # 100 MHz
    start, prep_end, end, folded = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
    t_min = start.min()
    dur = (end - start) * tick
    prep = (prep_end - start) * tick
    cd = (end - prep_end) * tick
    span = (end.max() - t_min) * tick
    col_nnz = np.diff(Xc.indptr)[tg]
    nz = (coef != 0).sum(axis=1)
    order = np.argsort(-dur)
    # slot occupancy over time
    grid = np.linspace(0, span, 21)
    busy = [(int(((start - t_min) * tick <= g) & ((end - t_min) * tick > g)).sum()) if False else
            int((((start - t_min) * tick <= g) & ((end - t_min) * tick > g)).sum()) for g in grid]
    rep = {
        "workload": args.workload, "targets": int(len(tg)), "slots": int(eng.last_fit_stats["slots"]),
        "wall_s": wall, "kernel_span_s": span,
        "sum_target_s": float(dur.sum()), "sum_prep_s": float(prep.sum()), "sum_cd_s": float(cd.sum()),
        "folded_entries": float(folded.sum()),
        "fold_ns_per_entry": float(cd.sum() / max(folded.sum(), 1) * 1e9),
        "targets_with_cd": int((folded > 0).sum()),
        "mean_sweeps": float(n_iter.mean()),
        "top": [dict(item=int(tg[i]), nnz=int(col_nnz[i]), dur_s=float(dur[i]), prep_s=float(prep[i]), cd_s=float(cd[i]),
                     start_s=float((start[i] - t_min) * tick), folded=float(folded[i]), sweeps=int(n_iter[i]),
                     nonzero=int(nz[i])) for i in order[:12]],
        "busy_slots_at_5pct_steps": busy,
        "dur_percentiles_s": {str(p): float(np.percentile(dur, p)) for p in (50, 90, 99, 99.9, 100)},
    }
    if tr.shape[1] >= 8 and tr[:, 4:8].any():
        # phase clocks (a library built with -DRTREC_FIT_PHASES: tools/ab_build.sh with AB_FLAGS): the single-wave kernel's
        # targets are those after the heavy head (engine.last_fit_stats["n_heavy"]; the head runs fit_columns_mw_kernel, whose
        # slots 4..7 mean fold / update / gap / cycles)
        nh = int(eng.last_fit_stats.get("n_heavy", 0))
        sw = slice(nh, None)
        ph = tr[sw, 4:8].sum(axis=0) * tick
        cd_sw = float(cd[sw].sum())
        rep["single_wave_phases_wave_seconds"] = {
            "targets": int(len(tg) - nh), "prep (X^T y + selection)": float(prep[sw].sum()), "coordinate descent": cd_sw,
            "ordered folds (dot_pass)": float(ph[0]), "residual updates (update_pass)": float(ph[1]),
            "screening passes (screen_pass)": float(ph[2]), "duality gaps": float(ph[3]),
            "rest of the descent (draws, Gram-tracked screens, bookkeeping)": cd_sw - float(ph.sum()),
            "folded_entries": float(folded[sw].sum()),
            "fold_ns_per_entry_in_dot_pass": float(ph[0] / max(folded[sw].sum(), 1) * 1e9)}
    print(json.dumps(rep, indent=1))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rep, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
