#!/usr/bin/env python3
"""optim="sgd" on the device at the ML-1M shape (6,040 x 3,706, ~0.63 M interactions, K=50): wall time of
SLIMElastic({"optim": "sgd", ...}).fit_in_parallel and a sample of columns checked against the C oracle.
    python tools/sgd_bench.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ml1m", help="ml1m (6,040 x 3,706) or a bench.py workload name (c3: ML-20M shape)")
    args = ap.parse_args()
    import torch
    from oracle import slim_oracle as so
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.models.internal.slim_elastic import SLIMElastic
    from rtrec_amd.synth import interaction_matrix
    if args.workload == "ml1m":
        X = interaction_matrix(6040, 3706, 1_000_000, seed=20251003).tocsc()
    else:
        from bench import WORKLOADS
        from rtrec_amd.synth import workload_matrix
        X = workload_matrix(WORKLOADS[args.workload]).tocsc()
    X.sort_indices()
    m = SLIMElastic({"optim": "sgd", "nn_feature_selection": 50}, engine=SlimEngine(device="cuda:0"))
    m.fit_in_parallel(X.copy(), item_ids=np.arange(64))           # warm-up (code objects, allocations)
    m.item_similarity = None
    torch.cuda.synchronize()
    t0 = time.time()
    m.fit_in_parallel(X.copy())
    torch.cuda.synchronize()
    dt = time.time() - t0
    W = m.item_similarity.tocsc()
    tg = m.engine.last_fit_targets
    nit = np.empty(X.shape[1], np.int64); nit[tg] = m.n_iter_
    cols = np.random.default_rng(1).choice(X.shape[1], 40 if args.workload == "ml1m" else 8, replace=False)
    ptr, idx, val, o_nit = so.fit_columns_sgd(X, cols, nn_feature_selection=50)
    ok = True
    for t, j in enumerate(cols):
        oi, ov = idx[ptr[t]:ptr[t + 1]], val[ptr[t]:ptr[t + 1]]
        nz = ov != 0
        col = W[:, j].tocoo()
        o = np.argsort(col.row)
        ok &= bool(np.array_equal(col.row[o], oi[nz]) and np.array_equal(col.data[o].astype(np.float32).view(np.uint32), ov[nz].view(np.uint32))
                   and int(nit[j]) == int(o_nit[t]))
    print(json.dumps({"shape": list(X.shape), "nnz": int(X.nnz), "fit_s": round(dt, 3), "interactions_per_s": round(X.nnz / dt),
                      "epochs_min_mean_max": [int(nit.min()), float(nit.mean()), int(nit.max())], "W_nnz": int(W.nnz),
                      "oracle_sample_columns": int(len(cols)), "bit_equal_to_oracle": ok}))


if __name__ == "__main__":
    main()
