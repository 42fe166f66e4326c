#!/usr/bin/env python3
"""Time the nn_feature_selection=None fit (every item is a feature) on a bench workload."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="small")
    ap.add_argument("--ml1m", action="store_true")
    args = ap.parse_args()
    import torch
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import interaction_matrix
    if args.ml1m:
        U, I, draws = 6040, 3706, 1_000_000
    else:
        wl = WORKLOADS[args.workload]
        U, I, draws = wl["U"], wl["I"], wl["draws"]
    X = interaction_matrix(U, I, draws, seed=20251003, float_ratings=True)
    Xc = X.tocsc(); Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    eng.fit_columns(np.arange(8), nn_feature_selection=None)
    torch.cuda.synchronize()
    t0 = time.time()
    tg, items, coef, count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=None, trace=True)
    torch.cuda.synchronize()
    dt = time.time() - t0
    tr = eng.last_fit_stats["trace"].astype(np.float64)
    print(json.dumps({"U": U, "I": I, "nnz": int(X.nnz), "K": None, "fit_s": dt, "interactions_per_s": X.nnz / dt,
                      "W_nnz": int(count.sum()), "mean_sweeps": float(n_iter.mean()),
                      "sum_target_s": float(((tr[:, 2] - tr[:, 0]) * 1e-8).sum()),
                      "max_target_s": float(((tr[:, 2] - tr[:, 0]) * 1e-8).max()),
                      "prep_share": float(((tr[:, 1] - tr[:, 0]).sum()) / max((tr[:, 2] - tr[:, 0]).sum(), 1))}))
    dur = (tr[:, 2] - tr[:, 0]) * 1e-8
    col_nnz = np.diff(Xc.indptr)[tg]
    for i in np.argsort(-dur)[:6]:
        print(json.dumps({"item": int(tg[i]), "nnz": int(col_nnz[i]), "dur_s": float(dur[i]), "prep_s": float((tr[i, 1] - tr[i, 0]) * 1e-8),
                          "sweeps": int(n_iter[i]), "nonzero": int(count[i]), "folded": float(tr[i, 3]),
                          "start_s": float((tr[i, 0] - tr[:, 0].min()) * 1e-8),
                          "rng_s": float(tr[i, 4] * 1e-8), "dots_s": float(tr[i, 5] * 1e-8), "loop_s": float(tr[i, 6] * 1e-8),
                          "gap_s": float(tr[i, 7] * 1e-8)}))


if __name__ == "__main__":
    main()
