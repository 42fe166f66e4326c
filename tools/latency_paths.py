#!/usr/bin/env python3
"""Request-sized batches through each scoring path (feature rows / segments / tiled): wall p50 of SLIM.recommend and of
recommend_batch(B users), per path.     python tools/latency_paths.py --workload c3s
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
    ap.add_argument("--requests", type=int, default=600)
    ap.add_argument("--batches", default="1,32,100,256,512,1000,2000,4000,8000")
    args = ap.parse_args()
    import torch
    from tools.stream_bench import workload_pairs
    from rtrec_amd import SLIM
    rng = np.random.default_rng(5)
    U, I, u, i = workload_pairs(args.workload)
    n = len(u)
    r = (rng.integers(1, 6, n) * np.exp(-rng.random(n) * 0.7)).astype(np.float64)
    model = SLIM(min_value=0, max_value=15, nn_feature_selection=50, fit_mode="gram")
    model.add_interactions_columns(u.astype(np.int64), i.astype(np.int64), 1.7e9 + np.arange(n, dtype=np.float64), r)
    model.bulk_fit(parallel=True, progress_bar=False)
    eng = model.model.engine
    users = rng.integers(0, U, 140000).tolist()
    args.requests = min(args.requests, 200)
    out = {"workload": args.workload, "paths": {}}
    paths = {"default": {}, "fast_only": {"FR_SMALL_BATCH": 0}, "seg_small": {"FR_SMALL_BATCH": 1 << 20},
             "seg_wave_only": {"FR_SMALL_BATCH": 1 << 20, "sg_heavy_min": 513},
             "seg_wg32": {"FR_SMALL_BATCH": 1 << 20, "sg_heavy_min": 33}}
    for name, flags in paths.items():
        saved = {k: getattr(eng, k) for k in flags}
        for k, v in flags.items():
            setattr(eng, k, v)
        res = {}
        for B in [int(x) for x in args.batches.split(",")]:
            lat = []
            for q in range(args.requests + 30):
                chunk = users[(q * 7) % 90:(q * 7) % 90 + B]
                t = time.perf_counter()
                if B == 1:
                    model.recommend(user=chunk[0], top_k=10)
                elif B <= 128:
                    model.recommend_batch(chunk, top_k=10)
                else:           # larger batches: the engine call (the API's list-of-lists formatting would dominate)
                    eng.recommend_rows(np.asarray(chunk), top_k=10)
                lat.append((time.perf_counter() - t) * 1e3)
            lat = np.asarray(lat[30:])
            res[str(B)] = {"p50_ms": float(np.quantile(lat, .5)), "p99_ms": float(np.quantile(lat, .99)), "path": eng.last_score_path}
        out["paths"][name] = res
        for k, v in saved.items():
            setattr(eng, k, v)
    # CANDIDATES mode: one user, 200 candidates (a re-ranking request): direct kernel against the tiled kernel
    cands = rng.permutation(I)[:200].tolist()
    cres = {}
    for name, flag in (("direct", True), ("tiled", False)):
        eng.cands_direct = flag
        lat = []
        for q in range(330):
            t = time.perf_counter()
            model.recommend(user=users[q], candidate_items=cands, top_k=10)
            lat.append((time.perf_counter() - t) * 1e3)
        cres[name] = {"p50_ms": float(np.quantile(lat[30:], .5)), "p99_ms": float(np.quantile(lat[30:], .99)), "path": eng.last_score_path}
    eng.cands_direct = True
    out["candidates_200_single_user"] = cres
    print(json.dumps(out))


if __name__ == "__main__":
    main()
