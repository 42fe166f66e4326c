#!/usr/bin/env python3
"""Phases of the first recommend_batch(100 users) after a mini-batch refit (synchronised between phases):
    python tools/after_update_phases.py --workload c3s"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3s")
    ap.add_argument("--batches", type=int, default=12)
    args = ap.parse_args()
    import torch
    from tools.stream_bench import workload_pairs
    from rtrec_amd import SLIM
    rng = np.random.default_rng(5)
    U, I, u, i = workload_pairs(args.workload)
    n = len(u)
    order = rng.permutation(n)
    u, i = u[order].astype(np.int64), i[order].astype(np.int64)
    r = (rng.integers(1, 6, n) * np.exp(-rng.random(n) * 0.7)).astype(np.float64)
    ts = 1.7e9 + np.arange(n, dtype=np.float64)
    n_bulk = n - (args.batches + 3) * 1000
    model = SLIM(min_value=0, max_value=15, nn_feature_selection=50, fit_mode="gram")
    model.add_interactions_columns(u[:n_bulk], i[:n_bulk], ts[:n_bulk], r[:n_bulk])
    model.bulk_fit(parallel=True, progress_bar=False)
    eng = model.model.engine
    acc = {}

    def wrap(obj, name, key):
        fn = getattr(obj, name)

        def timed(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*a, **k)
            torch.cuda.synchronize()
            acc.setdefault(key, []).append((time.perf_counter() - t0) * 1e3)
            return out
        setattr(obj, name, timed)
    wrap(model, "_sync_interactions", "sync_interactions")
    wrap(eng, "topk_supported", "topk_supported")
    wrap(eng, "_small_batch_layout", "small_batch_layout")
    wrap(eng.be, "score_topk", "score_kernel_call")
    wrap(eng, "_download", "download")
    wrap(eng, "recommend_rows", "recommend_rows_total")
    wrap(model.model, "_format", "format")
    total = []
    for k in range(args.batches + 3):
        a = n_bulk + k * 1000
        model.fit(list(zip(u[a:a + 1000].tolist(), i[a:a + 1000].tolist(), ts[a:a + 1000].tolist(), r[a:a + 1000].tolist())), progress_bar=False)
        torch.cuda.synchronize()
        users = rng.integers(0, U, 100).tolist()
        t0 = time.perf_counter()
        model.recommend_batch(users, top_k=10)
        total.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"workload": args.workload, "total_ms_p50_with_syncs": float(np.median(total[3:])),
                      "phases_ms_p50": {k: float(np.median(v[3:])) for k, v in acc.items()}}))


if __name__ == "__main__":
    main()
