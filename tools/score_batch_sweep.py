#!/usr/bin/env python3
"""Score-kernel latency against batch size for the two SPARSE-mode kernels (feature-row kernel vs tiled-CSR kernel):
where the engine's FR_MIN_ROWS threshold comes from.   python tools/score_batch_sweep.py --workload c3
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--reps", type=int, default=30)
    args = ap.parse_args()
    import torch
    from bench import WORKLOADS
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc(); Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    out = eng.fit_columns(eng.owned_columns(np.arange(I)), nn_feature_selection=K, device_out=True, mode="gram")
    eng.set_weights(eng.merge_fit(None, I, False, *out[:4]))
    rng = np.random.default_rng(1)
    res = []
    for n in (1, 2, 4, 8, 16, 32, 64, 256, 1024, 4096, 16384, 65536, U):
        rows = np.sort(rng.choice(U, n, replace=False)).astype(np.int32)
        d_rows = eng.be.to_dev(rows)
        rec = {"rows": n}
        for name, fr, order_min in (("tiled", False, None), ("feature_rows", True, None)):
            eng.use_feature_rows = fr
            eng.FR_MIN_ROWS = 0
            for _ in range(3):
                eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.reps):
                eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows)
            torch.cuda.synchronize()
            rec[name + "_ms"] = (time.perf_counter() - t0) / args.reps * 1e3
        res.append(rec)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
