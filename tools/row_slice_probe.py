#!/usr/bin/env python3
"""Why is a STRIDED 1/8 row shard of C4 slower than a contiguous slice of the same size?  Times SlimEngine._local_topk for
different 125k-user row sets of one fitted model (contiguous head / middle / tail, strided, random), with the work order the
engine picks and with none.   python tools/row_slice_probe.py --workload c4"""
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
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--giants", type=int, default=-1, help="SlimEngine.ROW_ORDER_GIANTS override (0: giant rows are not spread)")
    args = ap.parse_args()
    import torch
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc(); Xc.sort_indices()
    eng = SlimEngine(device="cuda:0", score_shard="rows")
    if args.giants >= 0:
        eng.ROW_ORDER_GIANTS = args.giants
    eng.set_interactions(Xc, X)
    out = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True, mode="gram")
    eng.set_weights(eng.merge_fit(None, I, False, *out[:4]))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    lens = np.diff(X.indptr)
    n = U // args.parts
    rng = np.random.default_rng(1)
    sets = {"contiguous_head": np.arange(n), "contiguous_middle": np.arange(U // 2, U // 2 + n), "contiguous_tail": np.arange(U - n, U),
            "strided": np.arange(0, U, args.parts)[:n], "random_sorted": np.sort(rng.choice(U, n, replace=False)),
            "random_unsorted": rng.choice(U, n, replace=False)}
    for r in range(args.parts):                      # every strided shard of an N-rank pass (tools/shard_model.py takes their maximum)
        sets[f"strided_{r}"] = np.arange(r, U, args.parts)
    for name, rows in sets.items():
        d = eng.be.to_dev(rows.astype(np.int32))
        m = len(rows)
        top = np.sort(lens[rows])[::-1][:4]
        rec = {"set": name, "rows": m, "items_mean": float(lens[rows].mean()), "items_max": int(lens[rows].max()), "longest_4": top.tolist()}
        for _ in range(3):
            eng._local_topk(d, m, xb, 10, True, _native.TOPK_SPARSE, None)
        torch.cuda.synchronize()
        eng.score_timer = eng.be.timer_create()               # the kernel's own duration (HIP events on its stream) ...
        t0 = time.perf_counter()
        for _ in range(5):
            eng._local_topk(d, m, xb, 10, True, _native.TOPK_SPARSE, None)
        t_host = (time.perf_counter() - t0) / 5 * 1e3           # ... the host's share of a call (launches are asynchronous) ...
        torch.cuda.synchronize()
        rec["ms"] = (time.perf_counter() - t0) / 5 * 1e3        # ... and the call as a whole
        kms, kn = eng.be.timer_read(eng.score_timer)
        eng.be.timer_destroy(eng.score_timer)
        eng.score_timer = 0
        rec["kernel_ms"] = kms / max(kn, 1)
        rec["host_ms_per_call"] = t_host
        rec["path"] = eng.last_score_path
        rec["grouped"] = bool(getattr(eng, "_order_grouped", False))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
