#!/usr/bin/env python3
"""Where one streaming mini-batch (SLIM.fit(batch) == Recommender.partial_fit) spends its time.

Same set-up as tools/stream_bench.py; the phases of a few mini-batches are timed separately
(ingest, CSC export, upload, fit kernel, coefficient merge) and the device-side per-target trace of
the latency-mode fit kernel is summarised (longest targets = the critical path of the call).

    python tools/stream_profile.py --workload c3 [--out gpurun_out/stream_profile_c3.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.stream_bench import SHAPES  # noqa: E402


class Clock:
    def __init__(self):
        self.t = {}

    def wrap(self, obj, name, key, sync=None):
        fn = getattr(obj, name)

        def timed(*a, **k):
            t0 = time.perf_counter()
            out = fn(*a, **k)
            if sync is not None:
                sync()
            self.t[key] = self.t.get(key, 0.0) + time.perf_counter() - t0
            return out
        setattr(obj, name, timed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=sorted(SHAPES))
    ap.add_argument("--batches", type=int, default=6)
    ap.add_argument("--batch-size", type=int, default=1000)
    ap.add_argument("--out", default=None)
    ap.add_argument("--fit-mode", default="exact")
    args = ap.parse_args()
    import torch
    from rtrec_amd import SLIM
    from rtrec_amd import engine as engine_mod
    from rtrec_amd.models.internal import slim_elastic as se
    from tools.stream_bench import workload_pairs

    rng = np.random.default_rng(5)
    U, I, u, i = workload_pairs(args.workload)
    n = len(u)
    order = rng.permutation(n)
    u, i = u[order], i[order]
    r = (rng.integers(1, 6, n) * np.exp(-rng.random(n) * 0.7)).astype(np.float64)
    ts = 1.7e9 + np.arange(n, dtype=np.float64)
    n_bulk = n - args.batches * args.batch_size

    model = SLIM(min_value=0, max_value=15, nn_feature_selection=50, fit_mode=args.fit_mode)
    for a in range(0, n_bulk, 4_000_000):
        b = min(a + 4_000_000, n_bulk)
        model.interactions.add_interactions_batch(model.user_ids.identify_many(u[a:b].astype(np.int64)),
                                                  model.item_ids.identify_many(i[a:b].astype(np.int64)), ts[a:b], r[a:b])
    model.bulk_fit(parallel=True, progress_bar=False)
    torch.cuda.synchronize()
    eng = model.model.engine
    sync = torch.cuda.synchronize

    reports = []
    for k in range(args.batches):
        a = n_bulk + k * args.batch_size
        b = a + args.batch_size
        batch = list(zip(u[a:b].tolist(), i[a:b].tolist(), ts[a:b].tolist(), r[a:b].tolist()))
        ck = Clock()
        ck.wrap(model, "_ingest", "ingest")
        ck.wrap(model.interactions, "to_csc", "to_csc")
        ck.wrap(eng, "set_interactions", "upload", sync)
        orig_fit = eng.fit_columns

        def traced(*aa, **kk):
            kk["trace"] = True
            return orig_fit(*aa, **kk)
        eng.fit_columns = traced
        ck.wrap(eng, "fit_columns", "fit_kernel", sync)
        orig_merge = se.merge_coefficients
        ck.wrap(se, "merge_coefficients", "merge")
        t0 = time.perf_counter()
        model.fit(batch, progress_bar=False)
        sync()
        total = time.perf_counter() - t0
        se.merge_coefficients = orig_merge
        for nm in ("_ingest",):
            delattr(model, nm)
        delattr(model.interactions, "to_csc")
        delattr(eng, "set_interactions")
        delattr(eng, "fit_columns")
        tr = eng.last_fit_stats["trace"].astype(np.float64)
        tick = 1e-8
        dur = (tr[:, 2] - tr[:, 0]) * tick
        prep = (tr[:, 1] - tr[:, 0]) * tick
        span = (tr[:, 2].max() - tr[:, 0].min()) * tick
        top = np.argsort(-dur)[:5]
        X = eng._X
        ny = X["col_nnz"][eng.last_fit_targets] if hasattr(eng, "last_fit_targets") else None
        rep = {"batch": k, "total_ms": total * 1e3, **{f"{kk}_ms": v * 1e3 for kk, v in ck.t.items()},
               "targets": int(len(dur)), "kernel_span_ms": span * 1e3, "sum_target_s": float(dur.sum()),
               "dur_ms_quantiles": [float(np.quantile(dur, q) * 1e3) for q in (0.5, 0.9, 0.99, 1.0)],
               "prep_ms_quantiles": [float(np.quantile(prep, q) * 1e3) for q in (0.5, 0.9, 0.99, 1.0)], "sum_prep_s": float(prep.sum()),
               "top": [dict(ny=(int(ny[j]) if ny is not None else None), dur_ms=float(dur[j] * 1e3), prep_ms=float(prep[j] * 1e3), folded=float(tr[j, 3]), fold_ms=float(tr[j, 4] * 1e-5), upd_ms=float(tr[j, 5] * 1e-5), gap_ms=float(tr[j, 6] * 1e-5), fold_cycles_per_entry=float(tr[j, 7] / max(tr[j, 3], 1)), wait_ms=float(tr[j, 7] * 1e-5), folded_entries=float(tr[j, 3])) for j in top[:3]]}
        reports.append(rep)
        print(json.dumps(rep), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(reports, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
