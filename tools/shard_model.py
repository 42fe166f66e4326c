#!/usr/bin/env python3
"""Single-GPU model of the sharded scoring pass (what each rank of an N-GPU job runs), for both ways
of dividing it: item-column shards of W (default) and user-row shards with W replicated.

For N in --worlds: build rank r's shard of W (r = 0..N-1) on cuda:0, time its local
score_topk launch over ALL users, and report max-over-ranks (the compute part of one bench step
at N GPUs; the RCCL all-gather of [B, k] ids/scores/aux/counts and the merge kernel come on top:
their byte counts are printed).  W comes from a real fit of the workload.

    python tools/shard_model.py --workload c3 --worlds 1,2,4,8
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


def main() -> None:
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--top-k", type=int, default=10)
    args = ap.parse_args()
    import torch
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine, coefficients_to_updates, merge_coefficients
    from rtrec_amd.synth import workload_matrix

    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    t0 = time.time()
    tg, items, coef, count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K)
    torch.cuda.synchronize()
    fit_s = time.time() - t0
    W = merge_coefficients(None, I, *coefficients_to_updates(tg, items, coef, count))
    print(json.dumps({"workload": args.workload, "fit_s": fit_s, "W_nnz": int(W.nnz)}), flush=True)
    d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    k = args.top_k
    base = None
    for N in [int(x) for x in args.worlds.split(",")]:
        per_rank = []
        for r in range(N):
            e = SlimEngine(device="cuda:0", rank=r, world_size=N)
            e._X = eng._X
            e.n_users, e.n_items = U, I
            e.set_weights(W)
            for _ in range(3):          # builds the layout; the feature-row kernel's pattern-grouped work order is computed the SECOND
                e._local_topk(d_rows, U, xb, k, True, _native.TOPK_SPARSE, None)     # time a row set is scored (host work: keep it out of the clock)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                e._local_topk(d_rows, U, xb, k, True, _native.TOPK_SPARSE, None)
            torch.cuda.synchronize()
            per_rank.append((time.perf_counter() - t0) / args.steps * 1e3)
            lay = e._layout(True)
        worst = max(per_rank)
        base = base or worst
        rec = k * 12 + 4                          # scores + ids + aux per entry, count per row
        # column shards: all-to-all of the per-shard lists (this rank receives its U/N slice from N-1 peers),
        # then an all-gather of the final lists (k * 8 + 4 bytes per user)
        exch_bytes = (N - 1) * (-(-U // N)) * rec + (N - 1) * (-(-U // N)) * (k * 8 + 4)
        # row shards: W replicated, this rank scores U/N users; only the final lists travel
        q = -(-U // N)
        e = SlimEngine(device="cuda:0", score_shard="rows")
        e._X = eng._X
        e.n_users, e.n_items = U, I
        e.set_weights(W)
        rows_ms = []
        for r in range(N):
            d_slice = d_rows[r::N].contiguous()           # strided: every rank the same mix of heavy and light users
            m = int(d_slice.shape[0])
            if m == 0:
                continue
            for _ in range(3):
                e._local_topk(d_slice, m, xb, k, True, _native.TOPK_SPARSE, None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                e._local_topk(d_slice, m, xb, k, True, _native.TOPK_SPARSE, None)
            torch.cuda.synchronize()
            rows_ms.append((time.perf_counter() - t0) / args.steps * 1e3)
        # The exchange, priced for a stated range of achieved all-gather / all-to-all rates (bytes RECEIVED per rank per second
        # over xGMI: 7 links x ~153 GB/s peak per GPU; RCCL's ring forms are per-link bound, so 100-400 GB/s brackets what
        # messages of this size reach) plus a fixed cost per collective.  `serial`: the whole exchange after the whole local pass
        # (round 4's row path); `overlapped`: the pass cut into C chunks whose exchanges run beside the next chunk's kernel
        # (both paths now): max(local, exchange) + min(local, exchange) / C.
        from rtrec_amd.engine import GATHER_CHUNK_ROWS, MAX_GATHER_CHUNKS, ROW_CHUNK_ROWS
        row_bytes = (N - 1) * q * (k * 8 + 4)
        c_rows = max(1, min(MAX_GATHER_CHUNKS, q // ROW_CHUNK_ROWS))
        c_cols = max(1, min(MAX_GATHER_CHUNKS, (U + GATHER_CHUNK_ROWS // 2) // GATHER_CHUNK_ROWS))
        lat_ms = 0.04                                  # launch + handshake of one collective

        def priced(local_ms, nbytes, n_coll, chunks):
            out = {}
            for bw in (100.0, 200.0, 400.0):
                ex = 0.0 if N == 1 else nbytes / (bw * 1e9) * 1e3 + lat_ms * n_coll * chunks
                serial = local_ms + ex
                over = max(local_ms, ex) + min(local_ms, ex) / chunks
                out[f"{int(bw)}GBps"] = {"exchange_ms": ex, "step_ms_serial": serial, "speedup_serial": base / serial,
                                         "step_ms_overlapped": over, "speedup_overlapped": base / over}
            return out
        print(json.dumps({"world": N, "column_shards": {"local_ms_max": worst, "local_ms_min": min(per_rank),
                                                         "speedup_vs_1": base / worst, "bytes_received_per_rank": exch_bytes,
                                                         "chunks": c_cols, "with_exchange": priced(worst, exch_bytes, 2, c_cols)},
                          "row_shards": {"local_ms_max": max(rows_ms), "speedup_vs_1": base / max(rows_ms),
                                         "bytes_received_per_rank": row_bytes, "chunks": c_rows,
                                         "with_exchange": priced(max(rows_ms), row_bytes, 1, c_rows)},
                          "users_per_s_compute_only": {"columns": U / (worst * 1e-3), "rows": U / (max(rows_ms) * 1e-3)},
                          "note": "speedup_vs_1 is compute only (slowest rank's kernel); with_exchange adds the collective at an ASSUMED "
                                  "rate -- no multi-GPU hardware run exists for this repository"}),
              flush=True)

if __name__ == "__main__":
    main()
