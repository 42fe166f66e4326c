#!/usr/bin/env python3
"""cProfile of Recommender.recommend_batch(all users) (the api leg of bench.py).  python tools/api_profile.py --workload c3s"""
import argparse, cProfile, io, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3s")
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
    users = list(range(U))
    model.recommend_batch(users[:1000], top_k=10)
    model.recommend_batch(users, top_k=10)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    out = model.recommend_batch(users, top_k=10)
    pr.disable()
    dt = time.perf_counter() - t0
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25)
    print(f"recommend_batch({U} users): {dt * 1e3:.1f} ms = {U / dt:,.0f} users/s\n" + s.getvalue())


if __name__ == "__main__":
    main()
