#!/usr/bin/env python3
"""Streaming update benchmark (BASELINE.json config 4 pattern at single-GPU scale).

Bulk-loads a synthetic catalogue, fits W once, then feeds mini-batches of new interactions through
the drop-in API exactly as the reference's Kinesis consumer / FastAPI app do (SLIM.fit(batch) ==
Recommender.partial_fit) and interleaves small recommend_batch calls.  Reports per-mini-batch
latency and the sustained interactions/s.   python tools/stream_bench.py --workload c3
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {"small": (20_000, 5_000, 500_000), "c2": (100_000, 50_000, 5_000_000), "c3": (138_493, 26_744, 26_000_000),
          "c4": (1_000_000, 500_000, 100_000_000)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2", choices=sorted(SHAPES))
    ap.add_argument("--batches", type=int, default=50)
    ap.add_argument("--batch-size", type=int, default=1000)
    ap.add_argument("--score-users", type=int, default=100)
    ap.add_argument("--qps", type=float, default=0.0,
                    help="fixed arrival rate in interactions/s (0 = back to back): mini-batch k is released at "
                         "k * batch_size / qps; reports how far completion lags behind release")
    ap.add_argument("--fit-mode", default="exact", choices=["exact", "gram", "shuffle"],
                    help="SLIMElastic fit_mode: exact (bit-identical to scikit-learn) or a tolerance mode (DESIGN.md 3.4)")
    ap.add_argument("--bulk-chunk", type=int, default=4_000_000, help="rows per vectorised bulk-ingest call")
    args = ap.parse_args()
    import torch
    from rtrec_amd import SLIM
    from rtrec_amd.synth import zipf_pairs

    U, I, draws = SHAPES[args.workload]
    rng = np.random.default_rng(5)
    u, i = zipf_pairs(U, I, draws, seed=20251003)
    n = len(u)
    order = rng.permutation(n)
    u, i = u[order], i[order]
    r = (rng.integers(1, 6, n) * np.exp(-rng.random(n) * 0.7)).astype(np.float64)
    ts = 1.7e9 + np.arange(n, dtype=np.float64)
    n_stream = args.batches * args.batch_size
    n_bulk = n - n_stream

    model = SLIM(min_value=0, max_value=15, nn_feature_selection=50, fit_mode=args.fit_mode)
    t0 = time.time()
    for a in range(0, n_bulk, args.bulk_chunk):    # vectorised bulk ingest
        b = min(a + args.bulk_chunk, n_bulk)
        model.interactions.add_interactions_batch(model.user_ids.identify_many(u[a:b].astype(np.int64)),
                                                  model.item_ids.identify_many(i[a:b].astype(np.int64)), ts[a:b], r[a:b])
    t_ingest = time.time() - t0
    t0 = time.time()
    model.bulk_fit(parallel=True, progress_bar=False)
    torch.cuda.synchronize()
    t_fit = time.time() - t0
    model.recommend_batch(list(range(args.score_users)), top_k=10)          # warm the scoring path
    print(f"bulk: {n_bulk} interactions ingested in {t_ingest:.2f}s, fitted in {t_fit:.2f}s, "
          f"W nnz={model.model._w_dev.nnz if model.model._w_dev is not None else model.model.item_similarity.nnz}", file=sys.stderr)

    fit_ms, rec_ms, touched, lag_ms = [], [], [], []
    period = args.batch_size / args.qps if args.qps > 0 else 0.0
    t_start = time.perf_counter()
    for k in range(args.batches):
        a = n_bulk + k * args.batch_size
        b = a + args.batch_size
        batch = list(zip(u[a:b].tolist(), i[a:b].tolist(), ts[a:b].tolist(), r[a:b].tolist()))
        release = t_start + k * period
        if period > 0.0:
            while time.perf_counter() < release:      # the batch has not arrived yet
                time.sleep(min(0.001, max(0.0, release - time.perf_counter())))
        t0 = time.perf_counter()
        model.fit(batch, progress_bar=False)
        torch.cuda.synchronize()
        fit_ms.append((time.perf_counter() - t0) * 1e3)
        lag_ms.append((time.perf_counter() - release) * 1e3)          # completion behind release (queueing + service)
        touched.append(len(set(i[a:b].tolist())))
        users = rng.integers(0, U, args.score_users).tolist()
        t0 = time.perf_counter()
        model.recommend_batch(users, top_k=10)
        rec_ms.append((time.perf_counter() - t0) * 1e3)
    fit_ms, rec_ms = np.array(fit_ms[2:]), np.array(rec_ms[2:])      # first two include allocation / layout warm-up
    out = {"workload": args.workload, "fit_mode": args.fit_mode, "w_host_copies": int(model.model._item_similarity is not None), "n_users": U, "n_items": I, "bulk_interactions": int(n_bulk),
           "batch_size": args.batch_size, "batches": args.batches, "items_touched_per_batch": float(np.mean(touched)),
           "partial_fit_ms": {"p50": float(np.median(fit_ms)), "p95": float(np.quantile(fit_ms, 0.95)), "max": float(fit_ms.max())},
           "partial_fit_interactions_per_sec": float(args.batch_size / (np.mean(fit_ms) * 1e-3)),
           "recommend_ms_per_call": {"users": args.score_users, "p50": float(np.median(rec_ms)), "p95": float(np.quantile(rec_ms, 0.95))},
           "bulk_ingest_interactions_per_sec": float(n_bulk / t_ingest), "bulk_fit_seconds": t_fit}
    if period > 0.0:
        lag = np.array(lag_ms[2:])
        out["fixed_qps"] = {"interactions_per_sec": args.qps, "batch_period_ms": period * 1e3,
                            "completion_lag_ms": {"p50": float(np.median(lag)), "p95": float(np.quantile(lag, 0.95)),
                                                  "max": float(lag.max())},
                            "keeps_up": bool(lag[-1] <= lag[0] + period * 1e3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
