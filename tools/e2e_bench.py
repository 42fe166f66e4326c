#!/usr/bin/env python3
"""End-to-end drop-in timing through the DataFrame API (the reference's own definition of
"samples/sec": len(train) / (ingest + fit wall time), recommender.py:81,126).

Same calls and the same ML-1M-shaped synthetic DataFrame as tests/calibration/time_reference.py runs against
the real rtrec + scikit-learn (59,623 interactions/s bulk_fit, 520 users/s
recommend_batch in the build container), here through rtrec_amd on the GPU:
    Recommender(SLIM(min_value=0, max_value=15, nn_feature_selection=50)).bulk_fit(df)
    model.recommend_batch(users, top_k=10)          # one call, and 100-user calls like evaluate()
    Recommender.partial_fit(1000 interactions)
    python tools/e2e_bench.py --shape ml1m|c2|c3
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = {"ml1m": (6040, 3706, 1_000_000), "c2": (100_000, 50_000, 5_000_000), "c3": (138_493, 26_744, 26_000_000)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml1m", choices=sorted(SHAPES) + ["c3s"])
    ap.add_argument("--stream", type=int, default=1000, help="interactions held back for the partial_fit step")
    ap.add_argument("--profile", action="store_true", help="cProfile the bulk_fit call (host-side breakdown to stderr)")
    ap.add_argument("--decay-days", type=int, default=0, help="SLIM(decay_in_days=...): BASELINE config 5 (0 = no time decay)")
    args = ap.parse_args()
    import torch
    from rtrec_amd import SLIM, Recommender
    from rtrec_amd.synth import interaction_matrix

    K = 50
    if args.shape == "c3s":                 # the structured C3 workload of bench.py (clustered items)
        from bench import WORKLOADS
        from rtrec_amd.synth import workload_matrix
        X = workload_matrix(WORKLOADS["c3s"])
        U, I = X.shape
    else:
        U, I, draws = SHAPES[args.shape]
        X = interaction_matrix(U, I, draws, seed=20251003, float_ratings=True)
    coo = X.tocoo()
    rng = np.random.default_rng(0)
    order = rng.permutation(coo.nnz)
    df = pd.DataFrame({"user": coo.row[order].astype(int), "item": coo.col[order].astype(int),
                       "tstamp": 1.7e9 + np.arange(coo.nnz, dtype=float), "rating": coo.data[order].astype(float)})
    train, stream = df.iloc[:-args.stream], df.iloc[-args.stream:]
    # one-time process start-up (HIP context, loading librtrec_amd.so's code objects, registering the
    # torch custom ops) is not part of the measurement -- like importing scikit-learn for the reference
    torch.zeros(1, device="cuda")
    from rtrec_amd.engine import HipBackend
    HipBackend()

    kw = {"decay_in_days": args.decay_days} if args.decay_days > 0 else {}
    rec = Recommender(SLIM(min_value=0, max_value=15, nn_feature_selection=K, **kw))
    sink = io.StringIO()
    import cProfile
    import pstats
    pr = cProfile.Profile() if args.profile else None
    t = time.time()
    if pr:
        pr.enable()
    with contextlib.redirect_stdout(sink):
        rec.bulk_fit(train, parallel=True)
    torch.cuda.synchronize()
    t_fit = time.time() - t
    if pr:
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(30)
    model = rec.get_model()

    users = list(range(0, U, 3))
    model.recommend_batch(users[:100], top_k=10)                       # uploads X once
    t = time.time(); recs = model.recommend_batch(users, top_k=10); t_rec = time.time() - t
    t = time.time()
    for s in range(0, len(users), 100):
        model.recommend_batch(users[s:s + 100], top_k=10)
    t_rec100 = time.time() - t
    q = list(range(0, I, 7))
    t = time.time(); rec.similar_items(q, top_k=10); t_sim = time.time() - t

    batch = list(stream.itertuples(index=False, name=None))
    half = len(batch) // 2
    with contextlib.redirect_stdout(sink):
        rec.partial_fit(batch[:half])                 # first update: allocations, lazy code-object loads
    model.recommend_batch(users[:100], top_k=10)
    torch.cuda.synchronize()
    batch = batch[half:]
    pr = cProfile.Profile() if args.profile else None
    if pr:
        pr.enable()
    t = time.time()
    with contextlib.redirect_stdout(sink):
        rec.partial_fit(batch)
    torch.cuda.synchronize()
    t_pf = time.time() - t
    t = time.time(); model.recommend_batch(users[:100], top_k=10); t_after = time.time() - t
    if pr:
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(45)

    print(json.dumps({
        "shape": args.shape, "n_users": U, "n_items": I, "train_interactions": int(len(train)), "K": K,
        "bulk_fit_s": t_fit, "bulk_fit_samples_per_sec_incl_ingest": len(train) / t_fit,
        "recommend_batch_users": len(users), "recommend_batch_s": t_rec, "recommend_users_per_sec": len(users) / t_rec,
        "recommend_100user_calls_users_per_sec": len(users) / t_rec100,
        "similar_items_queries": len(q), "similar_items_per_sec": len(q) / t_sim,
        "partial_fit_interactions": len(batch), "partial_fit_s": t_pf, "partial_fit_interactions_per_sec": len(batch) / t_pf,
        "recommend_100_after_update_ms": t_after * 1e3,
        "W_nnz": int(model.model.item_similarity.nnz), "mean_rec_len": float(np.mean([len(r) for r in recs])),
        "device_ingest": os.environ.get("RTREC_AMD_DEVICE_INGEST", "1") != "0", "host_cores": os.cpu_count()}))


if __name__ == "__main__":
    main()
