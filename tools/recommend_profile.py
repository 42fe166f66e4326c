#!/usr/bin/env python3
"""cProfile + wall percentiles of single-user SLIM.recommend (the /recommend boundary, rtrec/serving/app.py:77-93).
    python tools/recommend_profile.py --workload c3"""
import argparse, cProfile, io, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
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
    probe = rng.integers(0, U, 2300).tolist()
    for x in probe[:300]:
        model.recommend(x, top_k=10)
    lat = []
    for x in probe[300:1300]:
        t0 = time.perf_counter(); model.recommend(x, top_k=10); lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.asarray(lat)
    print(f"recommend(user): p50 {np.quantile(lat, .5):.4f} ms  p90 {np.quantile(lat, .9):.4f}  p99 {np.quantile(lat, .99):.4f}")
    pr = cProfile.Profile()
    pr.enable()
    for x in probe[1300:]:
        model.recommend(x, top_k=10)
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue())


if __name__ == "__main__":
    main()
