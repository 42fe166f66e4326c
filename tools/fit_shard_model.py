#!/usr/bin/env python3
"""Single-GPU model of the per-rank FIT time under multi-GPU sharding.

Fitting has no collective: rank r fits SlimEngine.owned_columns() of the targets, so its time can be
measured on one GPU by constructing the engine with (rank=0, world_size=G) and no process group.
    python tools/fit_shard_model.py --workload c3 --shards 1 2 4 8 [--heavy 256 --heavy-slots 256 --heavy-min-rows 2048]
"""
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
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--reps", type=int, default=2)
    args = ap.parse_args()
    import torch
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    for G in args.shards:
        eng = SlimEngine(device="cuda:0", rank=0, world_size=G)
        eng.set_interactions(Xc, X)
        mine = eng.owned_columns(np.arange(I))
        eng.fit_columns(mine[:64], nn_feature_selection=K)
        best = None
        for _ in range(args.reps):
            torch.cuda.synchronize()
            t0 = time.time()
            eng.fit_columns(mine, nn_feature_selection=K)
            torch.cuda.synchronize()
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
        print(json.dumps({"workload": args.workload, "shards": G, "targets": int(len(mine)), "fit_s": round(best, 3),
                          "n_heavy": int(eng.last_fit_stats["n_heavy"]),
                          "env": {k: v for k, v in os.environ.items() if k.startswith("RTREC_AMD_FIT")}}), flush=True)


if __name__ == "__main__":
    main()
