#!/usr/bin/env python3
"""DENSE mode (string item ids: every column competes) through the fast pass + flagged rows vs the tiled kernel alone:
all users of a workload in one engine call, and one user per call.   python tools/dense_bench.py --workload c3s"""
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
    args = ap.parse_args()
    import torch
    from rtrec_amd import _native
    from rtrec_amd import engine as E
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    X = workload_matrix(wl)
    Xc = X.tocsc(); Xc.sort_indices()
    U, I, K = wl["U"], wl["I"], wl["K"]
    eng = E.SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    out = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True, mode="gram")
    eng.set_weights(eng.merge_fit(None, I, False, *out[:4]))
    d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
    rep = {"workload": args.workload, "n_users": U}
    res = {}
    for name, flag in (("fast_pass", True), ("tiled_only", False)):
        eng.dense_fast = flag
        step = lambda: eng.score_topk_device(None, U, 10, True, _native.TOPK_DENSE, d_rows=d_rows)
        o = step(); torch.cuda.synchronize()
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); o = step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        res[name] = tuple(t.cpu().numpy() for t in o)
        lat = []
        for u in np.random.default_rng(1).integers(0, U, 330).tolist():
            t0 = time.perf_counter(); eng.recommend_rows(np.array([u]), top_k=10, mode=_native.TOPK_DENSE); lat.append((time.perf_counter() - t0) * 1e3)
        rep[name] = {"all_users_ms": float(np.median(ts)), "users_per_sec": U / (float(np.median(ts)) * 1e-3), "path": eng.last_score_path,
                     "rows_rescored": int(eng.rescored.item()) if eng.rescored is not None else None,
                     "single_user_p50_ms": float(np.quantile(lat[30:], .5))}
    # float64 W (the reference's serial fit): float32 fast pass + float64 refine step vs the float64 tiled kernel alone
    dw = eng.weights
    f64 = {}
    for name, flag in (("refine", True), ("tiled_only", False)):
        eng.f64_refine = flag
        eng.set_weights(dw, acc_f64=True)
        step = lambda: eng.score_topk_device(None, U, 10, True, _native.TOPK_SPARSE, d_rows=d_rows)
        eng.rescored = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        o = step(); torch.cuda.synchronize()
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); o = step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        f64[name] = tuple(t.cpu().numpy() for t in o)
        rep["f64_" + name] = {"all_users_ms": float(np.median(ts)), "users_per_sec": U / (float(np.median(ts)) * 1e-3),
                              "path": eng.last_score_path, "rows_to_tiled_kernel": int(eng.rescored.item())}
        eng.rescored = None
    rep["f64_identical"] = bool(all(np.array_equal(x.view(np.int32), y.view(np.int32)) for x, y in zip(f64["refine"], f64["tiled_only"])))
    a, b = res["fast_pass"], res["tiled_only"]
    rep["identical"] = bool(all(np.array_equal(x.view(np.int32) if x.dtype == np.float32 else x, y.view(np.int32) if y.dtype == np.float32 else y)
                                for x, y in zip(a, b)))
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
