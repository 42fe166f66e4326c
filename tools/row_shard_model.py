#!/usr/bin/env python3
"""Single-GPU model of the row-sharded scoring pass: rank 0 of N scores users 0, N, 2N, ... against the whole W (what
SlimEngine._score_row_sharded launches); the all-gather of the final lists (84 B per user) comes on top.
    python tools/row_shard_model.py --workload c3 --worlds 1,2,4,8
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
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc(); Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    d = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True)
    eng.set_weights(eng.merge_fit(None, I, False, *d[:4]))
    for N in [int(x) for x in args.worlds.split(",")]:
        d_rows = eng.be.to_dev(np.arange(0, U, N, dtype=np.int32))
        n = int(d_rows.shape[0])
        for _ in range(2):
            eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows)
        eng.score_timer = eng.be.timer_create()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        kern_ms, launches = eng.be.timer_read(eng.score_timer)
        eng.be.timer_destroy(eng.score_timer)
        eng.score_timer = 0
        print(json.dumps({"workload": args.workload, "world": N, "users_per_rank": n, "step_ms": ms,
                          "kernel_ms": kern_ms / max(launches, 1), "users_per_s_if_all_ranks": U / (ms * 1e-3)}), flush=True)


if __name__ == "__main__":
    main()
