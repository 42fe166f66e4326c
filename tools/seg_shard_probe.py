#!/usr/bin/env python3
"""General-W scoring at row-shard sizes: the strided user slices of 2 / 4 / 8 / 16 ranks on c3s (users r, r + N, ...), for the
workgroup-per-user threshold given by RTREC_AMD_SG_HEAVY_MIN (0 = the library's rule).   python tools/seg_shard_probe.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from bench import WORKLOADS
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS["c3s"]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl)
    Xc = X.tocsc(); Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    d = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True, mode="gram")
    eng.set_weights(eng.merge_fit(None, I, False, *d[:4]))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    out = {"heavy_min_env": os.environ.get("RTREC_AMD_SG_HEAVY_MIN", "0")}
    for N in (1, 2, 4, 8, 16, 32):
        rows = np.arange(0, U, N).astype(np.int32)
        d_rows = eng.be.to_dev(rows)
        n = len(rows)
        for _ in range(3):
            eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            o = eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
        torch.cuda.synchronize()
        out[f"ranks_{N}"] = {"users": n, "ms": round((time.perf_counter() - t0) / 10 * 1e3, 3)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
