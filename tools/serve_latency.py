#!/usr/bin/env python3
"""Single-user serving latency (the /recommend boundary, rtrec/serving/app.py:77-93) on a fitted model.

Bulk-loads a synthetic catalogue through the drop-in API, fits W, then measures
  * one caller: model.recommend(user) p50 / p99 (one launch per request);
  * C concurrent callers through serving.app.ModelGate.recommend with and without request coalescing:
    per-request p50 / p99 and requests/s (coalesced requests share one recommend_batch launch).
    python tools/serve_latency.py --workload c3 --clients 32
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.stream_bench import SHAPES  # noqa: E402


def pct(a, q):
    return float(np.quantile(np.asarray(a), q))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=sorted(SHAPES))
    ap.add_argument("--requests", type=int, default=2000)
    ap.add_argument("--clients", type=int, default=32)
    ap.add_argument("--coalesce-ms", type=float, default=1.0)
    ap.add_argument("--single-only", action="store_true", help="only the one-caller loop (for tracing)")
    args = ap.parse_args()
    import torch
    from rtrec_amd import SLIM
    from rtrec_amd.serving.app import ModelGate
    from tools.stream_bench import workload_pairs

    rng = np.random.default_rng(5)
    U, I, u, i = workload_pairs(args.workload)
    n = len(u)
    r = (rng.integers(1, 6, n) * np.exp(-rng.random(n) * 0.7)).astype(np.float64)
    ts = 1.7e9 + np.arange(n, dtype=np.float64)
    model = SLIM(min_value=0, max_value=15, nn_feature_selection=50)
    for a in range(0, n, 4_000_000):
        b = min(a + 4_000_000, n)
        model.interactions.add_interactions_batch(model.user_ids.identify_many(u[a:b].astype(np.int64)),
                                                  model.item_ids.identify_many(i[a:b].astype(np.int64)), ts[a:b], r[a:b])
    model.bulk_fit(parallel=True, progress_bar=False)
    torch.cuda.synchronize()
    users = rng.integers(0, U, args.requests).tolist()
    model.recommend_batch(users[:64], top_k=10)
    for x in users[:50]:
        model.recommend(user=x, top_k=10)

    lat = []
    t0 = time.perf_counter()
    for x in users:
        t = time.perf_counter()
        model.recommend(user=x, top_k=10)
        lat.append((time.perf_counter() - t) * 1e3)
    single = {"requests": len(users), "p50_ms": pct(lat, .5), "p99_ms": pct(lat, .99), "requests_per_sec": len(users) / (time.perf_counter() - t0)}

    if args.single_only:
        print(json.dumps({"workload": args.workload, "one_caller": single}))
        return

    def concurrent(gate):
        lat_c, lock = [], threading.Lock()
        chunks = [users[c::args.clients] for c in range(args.clients)]

        def client(mine):
            out = []
            for x in mine:
                t = time.perf_counter()
                gate.recommend(x, 10, True)
                out.append((time.perf_counter() - t) * 1e3)
            with lock:
                lat_c.extend(out)

        th = [threading.Thread(target=client, args=(c,)) for c in chunks]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        res = {"clients": args.clients, "requests": len(lat_c), "p50_ms": pct(lat_c, .5), "p99_ms": pct(lat_c, .99),
               "requests_per_sec": len(lat_c) / wall}
        if gate.coalescer is not None:
            res.update(launches=gate.coalescer.rounds, mean_batch=len(lat_c) / max(gate.coalescer.rounds, 1))
        return res

    out = {"workload": args.workload, "n_users": U, "n_items": I, "one_caller": single,
           "concurrent_uncoalesced": concurrent(ModelGate(model, coalesce_ms=-1)),
           "concurrent_coalesced": dict(concurrent(ModelGate(model, coalesce_ms=args.coalesce_ms)), coalesce_ms=args.coalesce_ms),
           "concurrent_coalesced_no_wait": concurrent(ModelGate(model, coalesce_ms=0))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
