#!/usr/bin/env python3
"""DENSE mode (string item ids) over COLUMN SHARDS, shards scored one after another on one GPU: the fast pass with its short
lists completed in place (rtrec_slim_dense_fill) against the tiled DENSE kernel a shard kept before.  Per world size the slowest
shard's local pass and the rows it still hands to the tiled kernel.   python tools/dense_shard_bench.py --workload c3s"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3s", choices=sorted(WORKLOADS))
    ap.add_argument("--worlds", default="1,2,8")
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    import torch
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine, coefficients_to_updates, merge_coefficients
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc(); Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    tg, items, coef, count, _ = eng.fit_columns(np.arange(I), nn_feature_selection=K, mode="gram")
    W = merge_coefficients(None, I, *coefficients_to_updates(tg, items, coef, count))
    d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    for N in [int(x) for x in args.worlds.split(",")]:
        rec = {"workload": args.workload, "world": N}
        for name, fill in (("fast_pass_with_fill", True), ("before", False)):
            per_rank, resc, paths = [], [], set()
            for r in range(N):
                e = SlimEngine(device="cuda:0", rank=r, world_size=N)
                e._X = eng._X
                e.n_users, e.n_items = U, I
                e.set_weights(W)
                e.dense_fill = fill
                e.rescored = torch.zeros(1, dtype=torch.int32, device="cuda:0")
                for _ in range(3):
                    e._local_topk(d_rows, U, xb, 10, True, _native.TOPK_DENSE, None)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    e._local_topk(d_rows, U, xb, 10, True, _native.TOPK_DENSE, None)
                torch.cuda.synchronize()
                per_rank.append((time.perf_counter() - t0) / args.steps * 1e3)
                resc.append(int(e.rescored.item()))
                paths.add(e.last_score_path)
            rec[name] = {"local_ms_max": max(per_rank), "local_ms_min": min(per_rank), "rows_to_tiled_max": max(resc),
                         "paths": sorted(paths)}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
