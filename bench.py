#!/usr/bin/env python3
"""bench.py -- SLIM hot path on MI355X: fit W, then score every user (top-10) K times.

Contract (one JSON line on rank 0):
  metric  users-scored/sec at top-10 (BASELINE.json's score metric); the fit half of the metric
          (interactions/sec) is reported in the `fit` object of the same line.
  step    one pass of the fused SpMV + interacted filter + top-k path over ALL users of the
          workload, inputs (X in CSR, W tiles) already resident in HBM.
  N > 1   W is sharded by item column over the ranks (each rank also fits only its own columns);
          every rank scores all users against its shard, the per-shard top-k lists are
          all-gathered over RCCL and merged -> total work is fixed: "scaling": "strong".
  roofline  algorithmic bytes of score_tiles_kernel (user row ids+vals,
          gathered W row ids+vals, top-k output; 8 B per entry) / its mean launch time measured
          with HIP events on the launch stream, against the 8 TB/s HBM peak.
  cpu_baseline  the C oracle (bit-checked restatement of the reference's scipy/sklearn path)
          timed single-threaded on this host on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1]: synthetic 100k users x 50k items, 5M interactions
    "c2": dict(U=100_000, I=50_000, draws=5_000_000, K=50,
               desc="synthetic 100k users x 50k items, 5M interaction draws (Zipf users 0.6 / items 0.85), "
                    "SLIM nn_feature_selection=50, fit + recommend top-10 for all users"),
    # BASELINE.json configs[2] shape (MovieLens-20M): 138,493 x 26,744
    "c3": dict(U=138_493, I=26_744, draws=26_000_000, K=50,
               desc="MovieLens-20M-shaped synthetic 138,493 users x 26,744 items, ~20M interactions, K=50"),
    # BASELINE.json configs[3] shape: 1M users x 500k items, 100M interactions (bulk part)
    "c4": dict(U=1_000_000, I=500_000, draws=100_000_000, K=50,
               desc="synthetic 1M users x 500k items, 100M interaction draws, K=50 (bulk fit + score; 8-GPU shard config)"),
    "small": dict(U=20_000, I=5_000, draws=500_000, K=50, desc="small plumbing workload 20k x 5k, 500k draws, K=50"),
}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--tile-cols", type=int, default=4096)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--score-shard", default="columns", choices=["columns", "rows"],
                    help="multi-GPU scoring: item-column shards of W + list exchange (BASELINE.json's configuration), or "
                         "user-row shards with W replicated (for catalogues whose W is tiny, e.g. c4)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget per cpu_baseline leg")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the
    # driver's own torch.distributed.run command line) as a CHILD, before this process has touched the GPU,
    # and leave with its exit code -- a bare `--gpus 8` must never silently measure one GPU.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}")
        sys.exit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        log(f"error: WORLD_SIZE={world} but --gpus {args.gpus}: the launcher and the bench disagree about the number of ranks")
        sys.exit(2)
    # RTREC_BENCH_SAME_GPU=1 (functional test only): every rank drives cuda:0 and the collectives
    # run over gloo, so the sharded path can be exercised on a single-GPU box.
    same_gpu = os.environ.get("RTREC_BENCH_SAME_GPU") == "1"
    if same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    ranks_seen, rank_devices, backend = 1, [f"cuda:{local_rank}"], None
    if world > 1:
        if same_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        backend = dist.get_backend()
        devs = [None] * world
        dist.all_gather_object(devs, f"cuda:{local_rank}")
        rank_devices = [d for d in devs if d is not None]
        ranks_seen = len(rank_devices)
        if ranks_seen != args.gpus:
            log(f"error: {ranks_seen} ranks joined, --gpus {args.gpus}")
            sys.exit(3)

    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine, coefficients_to_updates, merge_coefficients, shard_bounds
    from rtrec_amd.synth import interaction_matrix

    wl = WORKLOADS[args.workload]
    U, I, K, top_k = wl["U"], wl["I"], wl["K"], args.top_k
    t0 = time.time()
    X = interaction_matrix(U, I, wl["draws"], seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    nnz = int(X.nnz)
    if rank == 0:
        log(f"[bench] workload {args.workload}: {U} x {I}, nnz={nnz} generated in {time.time() - t0:.1f}s")

    eng = SlimEngine(device=f"cuda:{local_rank}", rank=rank, world_size=world, tile_cols=args.tile_cols,
                     score_shard=args.score_shard)
    eng.set_interactions(Xc, X)

    # ------------------------------------------------------------------ fit (each rank: its own columns)
    lo, hi = shard_bounds(I, world, rank)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    mine = eng.owned_columns(np.arange(I))       # this rank's fit targets, balanced by column length
    tg, items, coef, count, n_iter = eng.fit_columns(mine, nn_feature_selection=K)
    torch.cuda.synchronize()
    fit_local = time.time() - t0
    rows, cols, vals = coefficients_to_updates(tg, items, coef, count)
    # algorithmic bytes of this rank's fit (every datum once per target column):
    # y (8 nnz_j) + the co-occurring user rows that form X^T y (8 |I_u| per u in U_j) + the K selected
    # feature columns (8 nnz(c)) + the written coefficients (8 |S_j|)
    col_nnz_all = np.diff(Xc.indptr).astype(np.float64)
    row_nnz_all = np.diff(X.indptr).astype(np.float64)
    owned = np.zeros(I, dtype=bool)
    owned[mine] = True
    # sum over owned targets j of sum_{u in U_j} |I_u|  ==  sum over interactions (u, j owned) of |I_u|
    cooc = float(np.repeat(row_nnz_all, np.diff(X.indptr))[owned[X.indices]].sum())
    sel_mask = np.arange(items.shape[1])[None, :] < count[:, None]
    feat = float(col_nnz_all[items[sel_mask]].sum())
    fit_algo_bytes = 8.0 * float(col_nnz_all[mine].sum()) + 8.0 * cooc + 8.0 * feat + 8.0 * float(count.sum())
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (rows, cols, vals))
        rows = np.concatenate([p[0] for p in parts])
        cols = np.concatenate([p[1] for p in parts])
        vals = np.concatenate([p[2] for p in parts])
        t = torch.tensor([fit_local], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        fit_s = float(t.item())
    else:
        fit_s = fit_local
    W = merge_coefficients(None, I, rows, cols, vals)
    eng.set_weights(W)
    if rank == 0:
        log(f"[bench] fit: {fit_s:.2f}s ({nnz / fit_s:,.0f} interactions/s), W nnz={W.nnz}, "
            f"sweeps mean={n_iter.mean():.1f} max={n_iter.max()}")

    # ------------------------------------------------------------------ score: K timed steps
    row_ids = np.arange(U, dtype=np.int32)
    d_rows = eng.be.to_dev(row_ids)
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])

    def step():
        # world == 1: one fused launch.  world > 1: the engine's sharded path (SlimEngine.score_topk_device)
        # -- local top-k per column shard in row chunks, an all-to-all of the per-shard lists per chunk
        # overlapped with the next chunk's kernel, strided merge of this rank's slice, all-gather of the
        # final lists; or, with --score-shard rows, this rank's slice of the users against all of W.
        return eng.score_topk_device(None, U, top_k, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)

    for _ in range(args.warmup):
        out = step()
    lib = eng.be.lib
    torch.cuda.synchronize()
    prof_fn = getattr(lib, "rtrec_amd_score_profile", None) if os.environ.get("RTREC_AMD_LIB") else None
    if prof_fn is not None:          # diagnostic build (-DSCORE_PROFILE): per-phase clocks of the sparse kernel
        prof_fn(None, 1)
    eng.score_timer = eng.be.timer_create()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_total_ms, kern_launches = eng.be.timer_read(eng.score_timer)
    tot_ms, n_launch = C.c_double(kern_total_ms), C.c_int64(kern_launches)
    eng.be.timer_destroy(eng.score_timer)
    eng.score_timer = 0
    if prof_fn is not None and rank == 0:
        buf = (C.c_uint64 * 16)()
        prof_fn(buf, 0)
        names = ["jobs", "rowptr", "hdr", "group", "dense", "sparse", "select", "emit", "reset", "queue",
                 "n_dense", "n_sparse_rows", "n_sparse_chunks", "n_overflow", "total"]
        v = dict(zip(names, [int(x) for x in buf]))
        tot = max(v["total"], 1)
        log("[score profile] " + json.dumps({k: (v[k] if k in ("jobs", "total") or k.startswith("n_") else round(v[k] / tot, 4))
                                             for k in names}))
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = U * args.steps / dt

    # host-buffer boundary (SLIMElastic.recommend_batch hands over a scipy CSR): upload the user
    # rows over PCIe, score, download ids + scores.  Reported beside `value`, never as `value`.
    pcie_users_per_s = None
    if world == 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.recommend_csr(X, top_k=top_k, filter_interacted=True, mode=_native.TOPK_SPARSE)
        pcie_users_per_s = U / (time.perf_counter() - t1)

    # ------------------------------------------------------------------ roofline of score_tiles_kernel (this rank)
    Wr = W.tocsr()
    if world > 1 and args.score_shard == "rows":      # this rank: its slice of the users against all of W
        row_nnz_w = np.diff(Wr.indptr).astype(np.float64)
        Xs = X[rank::world]
        gathered_entries = float(row_nnz_w[Xs.indices].sum())
        algo_bytes = 8.0 * Xs.nnz + 8.0 * gathered_entries + 8.0 * top_k * Xs.shape[0] + 4.0 * (Xs.shape[0] + 1)
    else:
        shard_row_nnz = np.diff(Wr[:, lo:hi].tocsr().indptr).astype(np.float64) if hi > lo else np.zeros(I)
        users_per_item = np.diff(Xc.indptr).astype(np.float64)
        gathered_entries = float((users_per_item * shard_row_nnz).sum())
        algo_bytes = 8.0 * nnz + 8.0 * gathered_entries + 8.0 * top_k * U + 4.0 * (U + 1)
    kern_ms = tot_ms.value / max(n_launch.value, 1)
    achieved = algo_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0

    lay = eng._layout(True) or {"tile_cols": None, "n_tiles": 0, "n_cols": 0}
    import zlib
    # same value for every --gpus N: the sharded path returns the unsharded answer
    topk_crc = zlib.crc32(out[0].cpu().numpy().tobytes()) if rank == 0 else 0
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # HBM traffic of the dominant kernel comes from a separate rocprofv3 --pmc run (FETCH_SIZE /
    # WRITE_SIZE cannot be read from inside the process); the committed summary is attached when
    # it belongs to this workload.
    traffic, traffic_src = None, None
    import glob
    tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{args.workload}_pmc_traffic.json")))
    tpath = tpaths[-1] if tpaths else ""          # the most recent round's summary
    if world == 1 and tpath:
        try:
            traffic = json.load(open(tpath))["hbm_bytes_per_launch_corrected"]
            traffic_src = os.path.relpath(tpath, ROOT)
        except Exception:
            traffic = None

    line = {
        "metric": "users-scored/sec top-10 (SLIM recommend, int ids, filter_interacted) + fit interactions/sec in `fit`",
        "value": value, "unit": "users/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['desc']}", "n_users": U, "n_items": I, "nnz": nnz,
                   "nn_feature_selection": K, "top_k": top_k, "tile_cols": lay["tile_cols"], "n_tiles": lay["n_tiles"],
                   "active_columns": lay["n_cols"],
                   "parallelism": ("single GPU" if world == 1 else f"item-column shard x{world}" if args.score_shard == "columns"
                                   else f"user-row shard x{world}, W replicated")},
        "ranks_seen": ranks_seen, "rank_devices": rank_devices, "backend": backend,
        "pcie_inclusive_users_per_sec": pcie_users_per_s, "topk_ids_crc32": topk_crc,
        "fit": {"seconds": fit_s, "interactions_per_sec": nnz / fit_s, "columns_per_sec": I / fit_s,
                "W_nnz": int(W.nnz), "mean_sweeps": float(n_iter.mean()),
                "roofline": {"kernel": "fit_columns_kernel<false> (+ fit_columns_mw_kernel for the heaviest targets)",
                             "bound": "hbm", "achieved": fit_algo_bytes / fit_local / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": fit_algo_bytes / fit_local / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes": fit_algo_bytes, "seconds": fit_local,
                             "note": "rank 0's columns; the coordinate-descent sweeps re-read the K feature columns "
                                     "and the residual (not counted): measured fabric traffic is ~10 TB on c3 "
                                     "(profiles/r01_c3_fit_pmc.json), i.e. the kernel is bound by random 64-B sector "
                                     "traffic at ~3 TB/s, not by its compulsory bytes"}},
        "roofline": {"kernel": "score_sparse_kernel<float,false>", "bound": "hbm", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": algo_bytes,
                     "kernel_ms_avg": kern_ms, "launches": int(n_launch.value)},
    }

    # ------------------------------------------------------------------ cpu_baseline (oracle, 1 thread, bounded sample)
    if world == 1 and not args.no_cpu_baseline:
        from oracle import slim_oracle as so
        so.lib()
        rng = np.random.default_rng(7)
        sample = rng.permutation(U)
        n, done, spent = 64, 0, 0.0
        ok = True
        ids_gpu = out[0].cpu().numpy()
        while spent < args.cpu_seconds and done < U:
            rows_s = np.sort(sample[done:done + n])
            t1 = time.perf_counter()
            o_ids, _, _ = so.recommend_batch(X[rows_s], Wr, top_k=top_k)
            spent += time.perf_counter() - t1
            ok = ok and np.array_equal(o_ids, ids_gpu[rows_s])
            done += len(rows_s)
            n = min(n * 2, 8192)
        line["cpu_baseline"] = {"value": done / spent, "unit": "users/s", "cores": 1, "kind": "port",
                                "sample": f"{done} random users of the same workload scored with the C oracle in "
                                          f"{spent:.1f}s (top-k ids identical to the GPU: {ok})"}
        # fit leg: random columns until the budget is spent
        perm = rng.permutation(I)
        spent_f, nnz_f, ncol_f = 0.0, 0, 0
        col_nnz = np.diff(Xc.indptr)
        while spent_f < args.cpu_seconds and ncol_f < I:
            c = perm[ncol_f:ncol_f + 1]
            t1 = time.perf_counter()
            so.fit_columns(Xc, c, nn_feature_selection=K)
            spent_f += time.perf_counter() - t1
            nnz_f += int(col_nnz[c[0]])
            ncol_f += 1
        line["fit"]["cpu_baseline"] = {"value": nnz_f / spent_f, "unit": "interactions/s", "cores": 1, "kind": "port",
                                       "sample": f"{ncol_f} random item columns ({nnz_f} interactions) fitted with the "
                                                 f"C oracle in {spent_f:.1f}s"}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
