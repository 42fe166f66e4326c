#!/usr/bin/env python3
"""Does the segment kernel care which users run together?  ("users of one cluster on one XCD".)

Scores two row sets of equal size against the same W of the structured workload (c3s): a random eighth of the users, and the
users whose HOME cluster (the generator's own label: an upper bound for any clustering of the users) lies in one eighth of the
clusters.  With the second set every XCD's L2 only has to hold that eighth of W's rows, bound rows and segment pointers -- the
best case an XCD-aware work order could reach for its share.  Prints kernel ms per pass and per user for both.

    python tools/seg_locality_probe.py [--steps 10]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def home_clusters(wl, seed=20251003):
    """home cluster per user id, by replaying rtrec_amd.synth.clustered_pairs' draws"""
    U, I, n_draws, C_ = wl["U"], wl["I"], wl["draws"], wl.get("clusters", 80)
    rng = np.random.default_rng(seed)
    perm_u, _perm_i = rng.permutation(U), rng.permutation(I)
    rng.random(n_draws)                                   # the user ranks of the draws
    home = rng.integers(0, C_, size=U)                    # by user RANK
    out = np.empty(U, dtype=np.int64)
    out[perm_u] = home
    return out, C_


def main():
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS["c3s"]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc(); Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    d_tg, d_items, d_coef, d_count, _ = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True)
    eng.set_weights(eng.merge_fit(None, I, False, d_tg, d_items, d_coef, d_count))
    home, C_ = home_clusters(wl)
    rng = np.random.default_rng(1)
    group = np.flatnonzero(home < C_ // 8)
    rand = np.sort(rng.choice(U, len(group), replace=False))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    res = {}
    for name, rows in (("random eighth", rand), ("one eighth of the clusters", np.sort(group)), ("all users", np.arange(U))):
        d_rows = eng.be.to_dev(rows.astype(np.int32))
        step = lambda: eng.score_topk_device(None, len(rows), 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        eng.score_timer = eng.be.timer_create()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        kms, kn = eng.be.timer_read(eng.score_timer)
        eng.be.timer_destroy(eng.score_timer); eng.score_timer = 0
        res[name] = {"users": int(len(rows)), "items_per_user": float(np.diff(X.indptr)[rows].mean()), "ms_per_pass": dt * 1e3,
                     "kernel_ms": kms / max(kn, 1), "ns_per_user": dt / len(rows) * 1e9, "path": eng.last_score_path}
        print(json.dumps({name: res[name]}), flush=True)


if __name__ == "__main__":
    main()
