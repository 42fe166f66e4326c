#!/usr/bin/env python3
"""Bulk ingest of a workload's interactions, host path vs device path:
Recommender-style columnar ingest of the whole frame, timed end to end and by phase.

    python tools/ingest_bench.py --workload c3 [--chunk 0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--chunk", type=int, default=0, help="rows per add_interactions_columns call (0 = the model's own choice)")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from bench import WORKLOADS
    from tools.stream_bench import workload_pairs
    from rtrec_amd import SLIM
    from rtrec_amd.engine import HipBackend
    from rtrec_amd.utils.device_store import DeviceInteractions
    wl = WORKLOADS[args.workload]
    rng = np.random.default_rng(5)
    _, _, u, i = workload_pairs(args.workload)
    rep = rng.integers(0, len(u), len(u) // 10)            # a tenth of the pairs come a second time
    u, i = np.concatenate([u, u[rep]]), np.concatenate([i, i[rep]])
    order = rng.permutation(len(u))
    u, i = u[order].astype(np.int64), i[order].astype(np.int64)
    n = len(u)
    r = (rng.integers(1, 6, n) * np.exp(-rng.random(n) * 0.7)).astype(np.float64)
    ts = 1.7e9 + np.arange(n, dtype=np.float64)
    be = HipBackend()
    out = {"workload": args.workload, "interactions": int(n)}
    for device in (False, True):
        os.environ["RTREC_AMD_DEVICE_INGEST"] = "1" if device else "0"
        best = None
        for _ in range(args.reps):
            m = SLIM(min_value=0, max_value=15, nn_feature_selection=50)
            chunk = args.chunk or m.bulk_chunk_rows or (1 << 22)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for a in range(0, n, chunk):
                m.add_interactions_columns(u[a:a + chunk], i[a:a + chunk], ts[a:a + chunk], r[a:a + chunk])
            t1 = time.perf_counter()
            X = m._device_matrix(None)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            cur = {"ingest_s": t1 - t0, "matrix_on_device_s": t2 - t1, "total_s": t2 - t0, "chunk_rows": int(chunk),
                   "nnz": int(X["rval"].numel())}
            if best is None or cur["total_s"] < best["total_s"]:
                best = cur
        out["device" if device else "host"] = best
    # phases of the device path (one more pass, synchronised between steps)
    mir = DeviceInteractions(torch, be.device)
    ph = {}
    def tick(name, t):
        torch.cuda.synchronize(); ph[name] = time.perf_counter() - t; return time.perf_counter()
    t = time.perf_counter()
    du, di, dt, dd = (mir._dev(a) for a in (u, i, ts, r)); t = tick("upload_4_columns", t)
    res = mir.ingest(du, di, dt, dd, False, 0.0, 15.0, None, be.fold_pairs); t = tick("sort_fold_counts_download", t)
    out["device_phases_s"] = ph
    out["pairs_per_s_device"] = n / out["device"]["ingest_s"]
    out["pairs_per_s_host"] = n / out["host"]["ingest_s"]
    line = json.dumps(out)
    print(line)
    if args.out:
        open(args.out, "w").write(line + "\n")


if __name__ == "__main__":
    main()
