#!/usr/bin/env python3
"""Host-side profile (cProfile) of the two latency paths of the drop-in API on a fitted model:
single-user SLIM.recommend and a 1,000-interaction SLIM.fit mini-batch.   python tools/host_profile.py --workload c3 --fit-mode gram
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def top(pr, n=45):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(n)
    return s.getvalue()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--fit-mode", default="gram")
    ap.add_argument("--batches", type=int, default=6)
    args = ap.parse_args()
    import torch
    from rtrec_amd import SLIM
    from stream_bench import workload_pairs
    rng = np.random.default_rng(5)
    U, I, u, i = workload_pairs(args.workload)
    n = len(u)
    order = rng.permutation(n)
    u, i = u[order], i[order]
    r = (rng.integers(1, 6, n) * np.exp(-rng.random(n) * 0.7)).astype(np.float64)
    ts = 1.7e9 + np.arange(n, dtype=np.float64)
    n_bulk = n - (args.batches + 3) * 1000
    model = SLIM(min_value=0, max_value=15, nn_feature_selection=50, fit_mode=args.fit_mode)
    torch.zeros(1, device="cuda")
    from rtrec_amd.engine import HipBackend
    HipBackend()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for a in range(0, n_bulk, 4_000_000):
        b = min(a + 4_000_000, n_bulk)
        model.interactions.add_interactions_batch(model.user_ids.identify_many(u[a:b].astype(np.int64)),
                                                  model.item_ids.identify_many(i[a:b].astype(np.int64)), ts[a:b], r[a:b])
    t_ing = time.perf_counter() - t0
    model.bulk_fit(parallel=True, progress_bar=False)
    torch.cuda.synchronize()
    pr.disable()
    print(f"=== bulk ingest {t_ing:.2f}s + bulk_fit {time.perf_counter() - t0 - t_ing:.2f}s\n" + top(pr, 40))
    users = rng.integers(0, U, 300).tolist()
    model.recommend_batch(users[:64], top_k=10)
    for x in users[:20]:
        model.recommend(user=x, top_k=10)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for x in users:
        model.recommend(user=x, top_k=10)
    pr.disable()
    print("=== 300 x SLIM.recommend(user)\n" + top(pr))

    def batch(k):
        a = n_bulk + k * 1000
        return list(zip(u[a:a + 1000].tolist(), i[a:a + 1000].tolist(), ts[a:a + 1000].tolist(), r[a:a + 1000].tolist()))
    for k in range(3):
        model.fit(batch(k), progress_bar=False)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for k in range(3, 3 + args.batches):
        model.fit(batch(k), progress_bar=False)
        torch.cuda.synchronize()
    pr.disable()
    print(f"=== {args.batches} x SLIM.fit(1000 interactions), fit_mode={args.fit_mode}: {(time.perf_counter() - t0) / args.batches * 1e3:.1f} ms each\n" + top(pr, 60))

    pr = cProfile.Profile()
    t_rec = 0.0
    for k in range(3 + args.batches, 3 + 2 * args.batches):
        model.fit(batch(k), progress_bar=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pr.enable()
        model.recommend_batch(users[:100], top_k=10)
        pr.disable()
        t_rec += time.perf_counter() - t0
    print(f"=== {args.batches} x recommend_batch(100 users) right after a fit: {t_rec / args.batches * 1e3:.2f} ms each\n" + top(pr, 45))


if __name__ == "__main__":
    main()
