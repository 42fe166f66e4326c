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
          "c4": (1_000_000, 500_000, 100_000_000),
          # structured variants (rtrec_amd.synth.clustered_pairs): (users, items, draws, item clusters)
          "smalls": (20_000, 5_000, 900_000, 25), "c3s": (138_493, 26_744, 46_000_000, 80)}


def workload_pairs(workload):
    """(U, I, users, items) of the workload's distinct (user, item) pairs."""
    from rtrec_amd.synth import clustered_pairs, zipf_pairs
    shape = SHAPES[workload]
    U, I, draws = shape[:3]
    if len(shape) > 3:
        u, i = clustered_pairs(U, I, draws, seed=20251003, n_clusters=shape[3])
    else:
        u, i = zipf_pairs(U, I, draws, seed=20251003)
    return U, I, u, i


def run_stream(workload="c2", batches=50, batch_size=1000, score_users=100, qps=0.0, fit_modes=("exact",),
               bulk_chunk=4_000_000, log=lambda msg: print(msg, file=sys.stderr)):
    """Bulk-load + fit, then `batches` mini-batches per fit mode through SLIM.fit / recommend_batch.  Returns the report
    (one entry per fit mode under "modes"; the first mode's figures are also at the top level)."""
    import torch
    from rtrec_amd import SLIM

    rng = np.random.default_rng(5)
    U, I, u, i = workload_pairs(workload)
    n = len(u)
    order = rng.permutation(n)
    u, i = u[order], i[order]
    r = (rng.integers(1, 6, n) * np.exp(-rng.random(n) * 0.7)).astype(np.float64)
    ts = 1.7e9 + np.arange(n, dtype=np.float64)
    n_stream = batches * batch_size * len(fit_modes)
    n_bulk = n - n_stream

    model = SLIM(min_value=0, max_value=15, nn_feature_selection=50, fit_mode=fit_modes[0])
    t0 = time.time()
    for a in range(0, n_bulk, bulk_chunk):    # vectorised bulk ingest
        b = min(a + bulk_chunk, n_bulk)
        model.interactions.add_interactions_batch(model.user_ids.identify_many(u[a:b].astype(np.int64)),
                                                  model.item_ids.identify_many(i[a:b].astype(np.int64)), ts[a:b], r[a:b])
    t_ingest = time.time() - t0
    t0 = time.time()
    model.bulk_fit(parallel=True, progress_bar=False)
    torch.cuda.synchronize()
    t_fit = time.time() - t0
    model.recommend_batch(list(range(score_users)), top_k=10)          # warm the scoring path
    log(f"[stream] bulk: {n_bulk} interactions ingested in {t_ingest:.2f}s, fitted in {t_fit:.2f}s, W nnz={model.model._w_dev.nnz}")
    out = {"workload": workload, "n_users": U, "n_items": I, "bulk_interactions": int(n_bulk), "batch_size": batch_size,
           "batches": batches, "bulk_ingest_interactions_per_sec": float(n_bulk / t_ingest), "bulk_fit_seconds": t_fit, "modes": {}}
    period = batch_size / qps if qps > 0 else 0.0
    for mi, mode in enumerate(fit_modes):
        model.model.fit_mode = mode
        fit_ms, rec_ms, touched, lag_ms = [], [], [], []
        t_start = time.perf_counter()
        for k in range(batches):
            a = n_bulk + (mi * batches + k) * batch_size
            b = a + batch_size
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
            users = rng.integers(0, U, score_users).tolist()
            t0 = time.perf_counter()
            model.recommend_batch(users, top_k=10)
            rec_ms.append((time.perf_counter() - t0) * 1e3)
        skip = 2 if batches > 4 else 0                                    # the first two include allocation / layout warm-up
        fit_a, rec_a = np.array(fit_ms[skip:]), np.array(rec_ms[skip:])
        res = {"fit_mode": mode, "items_touched_per_batch": float(np.mean(touched)),
               "partial_fit_ms": {"p50": float(np.median(fit_a)), "p95": float(np.quantile(fit_a, 0.95)), "max": float(fit_a.max())},
               "partial_fit_interactions_per_sec": float(batch_size / (np.mean(fit_a) * 1e-3)),
               "recommend_after_update_ms": {"users": score_users, "p50": float(np.median(rec_a)), "p95": float(np.quantile(rec_a, 0.95)),
                                             "note": "first recommend_batch after the refit: includes the device-side rebuild "
                                                     "of the score layouts from the resident W"},
               "turnaround_ms_p50": float(np.median(fit_a + rec_a)),
               "w_host_copies": int(model.model._item_similarity is not None)}
        if period > 0.0:
            lag = np.array(lag_ms[skip:])
            res["fixed_qps"] = {"interactions_per_sec": qps, "batch_period_ms": period * 1e3,
                                "completion_lag_ms": {"p50": float(np.median(lag)), "p95": float(np.quantile(lag, 0.95)),
                                                      "max": float(lag.max())},
                                "keeps_up": bool(lag[-1] <= lag[0] + period * 1e3)}
        out["modes"][mode] = res
        log(f"[stream] {mode}: partial_fit p50 {res['partial_fit_ms']['p50']:.1f} ms ({res['partial_fit_interactions_per_sec']:,.0f} "
            f"interactions/s), recommend({score_users}) after update p50 {res['recommend_after_update_ms']['p50']:.2f} ms")
    return out


def run_open_loop(workload="c3", rates=(1000.0, 5000.0, 20000.0), duration_s=3.0, fit_modes=("exact", "gram"), max_batch=10_000,
                  score_users=100, bulk_chunk=4_000_000, model=None, held=None, log=lambda msg: print(msg, file=sys.stderr)):
    """Fixed-QPS streaming (BASELINE.json config 4's pattern; the caller is the reference's Kinesis consumer,
    examples/kinesis/kinesis_consumer.py:87-99: poll, SLIM.fit(whatever arrived), repeat; rtrec/serving/app.py:56-91 serves
    recommend between updates).  OPEN LOOP: interactions arrive as a Poisson process at `rate` per second whether or not the
    consumer keeps up.  The consumer takes everything that has arrived (at most max_batch), SLIM.fit()s it, and a
    recommend_batch(score_users) follows -- the update is VISIBLE when that returns (it rebuilds the score layouts from the
    refitted W).  Per interaction: update-to-visible latency = visible time - arrival time.  Reports, per rate and fit mode,
    the sustained rate, batch sizes, p50 / p99 latency and whether the backlog stayed bounded."""
    import torch
    from rtrec_amd import SLIM
    rng = np.random.default_rng(9)
    if model is None:
        U, I, u, i = workload_pairs(workload)
        n = len(u)
        order = rng.permutation(n)
        u, i = u[order], i[order]
        r = (rng.integers(1, 6, n) * np.exp(-rng.random(n) * 0.7)).astype(np.float64)
        ts = 1.7e9 + np.arange(n, dtype=np.float64)
        n_hold = int(min(n // 4, sum(rates) * duration_s * len(fit_modes) * 1.3 + 20_000))
        n_bulk = n - n_hold
        model = SLIM(min_value=0, max_value=15, nn_feature_selection=50, fit_mode=fit_modes[0])
        for a in range(0, n_bulk, bulk_chunk):
            b = min(a + bulk_chunk, n_bulk)
            model.interactions.add_interactions_batch(model.user_ids.identify_many(u[a:b].astype(np.int64)),
                                                      model.item_ids.identify_many(i[a:b].astype(np.int64)), ts[a:b], r[a:b])
        model.bulk_fit(parallel=True, progress_bar=False)
        torch.cuda.synchronize()
        held = (u[n_bulk:], i[n_bulk:], ts[n_bulk:], r[n_bulk:], U)
    hu, hi, hts, hr, U = held
    model.recommend_batch(list(range(score_users)), top_k=10)
    out = {"workload": workload, "max_batch": max_batch, "duration_s": duration_s, "score_users": score_users, "runs": []}
    cursor = 0
    for mode in fit_modes:
        model.model.fit_mode = mode
        for rate in rates:
            n_arr = int(rate * duration_s)
            if cursor + n_arr > len(hu):
                break
            arrive = np.cumsum(rng.exponential(1.0 / rate, n_arr))
            a0 = cursor
            cursor += n_arr
            done, lat, sizes, fit_ms = 0, [], [], []
            t0 = time.perf_counter()
            while done < n_arr:
                now = time.perf_counter() - t0
                avail = int(np.searchsorted(arrive, now, side="right")) - done
                if avail <= 0:
                    time.sleep(max(0.0, min(0.002, arrive[done] - now)))
                    continue
                take = min(avail, max_batch)
                sl = slice(a0 + done, a0 + done + take)
                batch = list(zip(hu[sl].tolist(), hi[sl].tolist(), hts[sl].tolist(), hr[sl].tolist()))
                t1 = time.perf_counter()
                model.fit(batch, progress_bar=False)
                torch.cuda.synchronize()
                fit_ms.append((time.perf_counter() - t1) * 1e3)
                model.recommend_batch(rng.integers(0, U, score_users).tolist(), top_k=10)
                visible = time.perf_counter() - t0
                lat.append(visible - arrive[done:done + take])
                sizes.append(take)
                done += take
            wall = time.perf_counter() - t0
            lat = np.concatenate(lat) * 1e3
            third = max(1, len(sizes) // 3)
            res = {"fit_mode": mode, "arrival_rate_per_sec": rate, "interactions": n_arr, "wall_s": wall,
                   "sustained_interactions_per_sec": n_arr / wall, "batches": len(sizes),
                   "batch_size": {"mean": float(np.mean(sizes)), "max": int(np.max(sizes))},
                   "fit_ms": {"p50": float(np.median(fit_ms)), "max": float(np.max(fit_ms))},
                   "update_to_visible_ms": {"p50": float(np.median(lat)), "p99": float(np.quantile(lat, 0.99)), "max": float(lat.max())},
                   # the backlog is bounded when the last batches are no larger than the first ones
                   "keeps_up": bool(np.mean(sizes[-third:]) <= 2.0 * max(np.mean(sizes[:third]), 1.0) and wall <= duration_s * 1.25 + 1.0)}
            out["runs"].append(res)
            log(f"[stream] open loop {mode} @ {rate:,.0f}/s: sustained {res['sustained_interactions_per_sec']:,.0f}/s, batch mean "
                f"{res['batch_size']['mean']:.0f}, update-to-visible p50 {res['update_to_visible_ms']['p50']:.0f} ms p99 "
                f"{res['update_to_visible_ms']['p99']:.0f} ms, keeps up: {res['keeps_up']}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2", choices=sorted(SHAPES))
    ap.add_argument("--batches", type=int, default=50)
    ap.add_argument("--batch-size", type=int, default=1000)
    ap.add_argument("--score-users", type=int, default=100)
    ap.add_argument("--qps", type=float, default=0.0,
                    help="fixed arrival rate in interactions/s (0 = back to back): mini-batch k is released at "
                         "k * batch_size / qps; reports how far completion lags behind release")
    ap.add_argument("--fit-mode", default="exact",
                    help="comma-separated SLIMElastic fit modes to stream with: exact (bit-identical to scikit-learn), gram, shuffle")
    ap.add_argument("--bulk-chunk", type=int, default=4_000_000, help="rows per vectorised bulk-ingest call")
    ap.add_argument("--open-loop", default="", help="comma-separated arrival rates (interactions/s): run the fixed-QPS open-loop "
                                                    "harness instead (Poisson arrivals, consumer takes what has arrived)")
    ap.add_argument("--duration", type=float, default=3.0, help="seconds of arrivals per open-loop run")
    args = ap.parse_args()
    if args.open_loop:
        print(json.dumps(run_open_loop(args.workload, tuple(float(v) for v in args.open_loop.split(",")), args.duration,
                                       tuple(args.fit_mode.split(",")), score_users=args.score_users)))
        return
    print(json.dumps(run_stream(args.workload, args.batches, args.batch_size, args.score_users, args.qps,
                                tuple(args.fit_mode.split(",")), args.bulk_chunk)))


if __name__ == "__main__":
    main()
