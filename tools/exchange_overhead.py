#!/usr/bin/env python3
"""Cost of the multi-GPU exchange machinery of the scoring path on ONE GPU: a 1-rank NCCL (RCCL) group, the sharded code
path forced on (SlimEngine.force_exchange) -- packing, all_to_all_single, strided merge, all_gather_into_tensor -- against
the plain single-GPU launch on the same workload.  What an N-GPU step pays on top of its kernels, minus the wire.
    python tools/exchange_overhead.py --workload c3
"""
import argparse
import json
import os
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--chunk-rows", default="", help="comma-separated SlimEngine.gather_chunk_rows values to try (column shards)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine, coefficients_to_updates, merge_coefficients
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc(); Xc.sort_indices()
    out = {"workload": args.workload}
    ref = None
    runs = [("single", "columns", False, 0), ("exchange_columns", "columns", True, 0), ("exchange_rows", "rows", True, 0)]
    runs += [(f"exchange_columns_chunk{c}", "columns", True, int(c)) for c in filter(None, args.chunk_rows.split(","))]
    for name, mode, force, chunk in runs:
        eng = SlimEngine(device="cuda:0", rank=0, world_size=1, score_shard=mode)
        eng.force_exchange = force
        if chunk:
            eng.gather_chunk_rows = chunk
        eng.set_interactions(Xc, X)
        tg, items, coef, count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K)
        eng.set_weights(merge_coefficients(None, I, *coefficients_to_updates(tg, items, coef, count)))
        d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
        for _ in range(2):
            ids, sc, cnt = eng.score_topk_device(None, U, top_k=10, filter_interacted=True, mode=_native.TOPK_SPARSE, d_rows=d_rows)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ids, sc, cnt = eng.score_topk_device(None, U, top_k=10, filter_interacted=True, mode=_native.TOPK_SPARSE, d_rows=d_rows)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        h = ids.cpu().numpy()
        if ref is None:
            ref = h
        out[name] = {"ms_per_step": ms, "same_ids": bool(np.array_equal(ref, h))}
        print(f"[exchange] {name}: {ms:.3f} ms per step", file=sys.stderr, flush=True)
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
