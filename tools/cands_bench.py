#!/usr/bin/env python3
"""CANDIDATES mode over all users (slim_elastic.py:722-739: X[users] @ W[:, candidates], argsort top-k, no interacted filter):
the direct kernel (csrc/score_cands.hip: W's CSC columns of the candidates, no pass over all columns) against the tiled
kernel with a rank array, for several candidate-list sizes.   python tools/cands_bench.py --workload c3s"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3s")
    args = ap.parse_args()
    import torch
    from bench import WORKLOADS
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl)
    Xc = X.tocsc(); Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    d = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True, mode="gram")
    eng.set_weights(eng.merge_fit(None, I, False, *d[:4]))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
    rng = np.random.default_rng(2)
    pop = np.argsort(-np.diff(Xc.indptr))
    for n_c in (20, 200, 1000, 4096):
        cands = np.sort(rng.choice(pop[:8000], n_c, replace=False)).astype(np.int64)
        res = {}
        for label, direct in (("direct", True), ("tiled", False)):
            eng.cands_direct = direct
            eng.CANDS_DIRECT_MAX_PAIRS, eng.CANDS_DIRECT_BULK = 1 << 40, 1 << 30       # (force the direct kernel at every size)
            f = lambda: eng.score_topk_device(None, U, 10, False, _native.TOPK_CANDIDATES, d_rows=d_rows, xb=xb, candidates=cands)
            out = f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = f()
            torch.cuda.synchronize()
            res[label] = {"ms": round((time.perf_counter() - t0) / 3 * 1e3, 3), "path": eng.last_score_path,
                          "ids": out[0].cpu().numpy(), "cnt": out[2].cpu().numpy()}
        same = bool(np.array_equal(res["direct"]["ids"], res["tiled"]["ids"]) and np.array_equal(res["direct"]["cnt"], res["tiled"]["cnt"]))
        print(json.dumps({"workload": args.workload, "users": U, "candidates": n_c, "direct_ms": res["direct"]["ms"],
                          "tiled_ms": res["tiled"]["ms"], "paths": [res["direct"]["path"], res["tiled"]["path"]], "same_ids": same}), flush=True)


if __name__ == "__main__":
    main()
