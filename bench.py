#!/usr/bin/env python3
"""bench.py -- SLIM hot path on MI355X: fit W, then score every user (top-10) K times.

Contract (one JSON line on rank 0):
  metric  users-scored/sec at top-10 (BASELINE.json's score metric); the fit half of the metric
          (interactions/sec) is reported in the `fit` object of the same line.
  step    one pass of the fused SpMV + interacted filter + top-k path over ALL users of the
          workload, inputs (X in CSR, W tiles) already resident in HBM.
  N > 1   the fit is sharded by item column (each rank fits its own columns, the coefficients are all-gathered
          on the device, every rank holds W).  Scoring is divided over the ranks either by USER ROWS (default:
          W is ~1 MB and replicated, a rank scores every N-th user, only the final lists are all-gathered)
          or by ITEM COLUMNS as BASELINE.json words it (--score-shard columns: every rank scores all users
          against its shard, the per-shard lists go through an all-to-all, a merge and an all-gather);
          the division not chosen is timed right after the pass of record and reported as `alt_sharding`.
          Total work is fixed either way: "scaling": "strong".
  roofline  the dominant score kernel against the limits that can bind it -- vector-ALU issue (the sums
          are unfused float32 multiply + add), LDS reads, L2 -> LDS staging -- and its compulsory HBM
          bytes; mean launch time from HIP events on the launch stream (a caller-owned rtrec_timer).
          SURVEY 8d's "algorithmic bytes" (8 B per gathered W entry) are kept as `algorithmic`: W never
          leaves L2 / LDS, so they are not a fraction of any hardware limit.
  cpu_baseline  the C oracle (bit-checked restatement of the reference's scipy/sklearn path) timed on this
          host with one thread and with all cores, on bounded samples of the same workload.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

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
    # the C3 shape with item-item structure: 80 item clusters, 85 % of a user's draws inside its home
    # cluster, same Zipf marginals -> W has thousands of non-empty rows (the general-W scoring path)
    "c3s": dict(U=138_493, I=26_744, draws=46_000_000, K=50, gen="clustered", clusters=80, p_in=0.85,
                desc="MovieLens-20M-shaped STRUCTURED synthetic 138,493 users x 26,744 items, ~20M interactions in 80 item "
                     "clusters (85 % of a user's draws inside its home cluster, Zipf marginals kept), K=50"),
    "smalls": dict(U=20_000, I=5_000, draws=900_000, K=50, gen="clustered", clusters=25, p_in=0.85,
                   desc="small structured plumbing workload 20k x 5k in 25 item clusters, K=50"),
    # the 500k-item shape WITH item-item structure.  With the reference's default alpha = 0.1 the L1 threshold (alpha x
    # l1_ratio x n_users) leaves a degenerate W at 1M users whatever the generator (round 3: 1,920 weights in 51 rows); a
    # smaller regularisation strength is a legitimate hyper-parameter choice for a catalogue this sparse and gives the
    # wide-tile (T > 256) segment layout its full-size workload.  Not a BASELINE config: tools/c3s_probe.py --workload c4s.
    "c4s": dict(U=1_000_000, I=500_000, draws=100_000_000, K=50, gen="clustered", clusters=1500, p_in=0.85, alpha=0.005,
                desc="STRUCTURED synthetic 1M users x 500k items, ~84M interactions in 1,500 item clusters, K=50, alpha=0.005 "
                     "(a hyper-parameter choice: the default 0.1 leaves W degenerate at this scale)"),
}
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def attach_profile(path, kernel_name, fingerprint, root=None):
    """A committed rocprofv3 summary (profiles/r*_<workload>_*.json) may ride on the bench line only when it describes (a) the
    kernel the line names as dominant and (b) THE BUILD BEING TIMED: its "build" stamp names the librtrec_amd.so this process
    loaded (equal sha256 of the library, or of the kernel sources + compiler flags it was built from -- a rebuild of the
    same tree need not be byte-identical; rtrec_amd.build.same_build) (a summary of another round's kernels used to be attached by name
    alone).  Returns (summary or None, relative path of a summary that was refused as stale or None)."""
    root = root or ROOT
    if not path:
        return None, None
    rel = os.path.relpath(path, root)
    try:
        j = json.load(open(path))
    except Exception:
        return None, None
    if kernel_name is not None and j.get("kernel", "").split("<")[0] != kernel_name.split("<")[0]:
        return None, None                      # another kernel's counters: not this line's business
    from rtrec_amd import build as _b
    if not _b.same_build(j.get("build"), fingerprint):      # equal library bytes, or equal kernel sources + compiler flags
        return None, rel
    return dict(j, source=rel), None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def api_leg(X, K, top_k):
    """SURVEY 8d's API-level figures (N = 1): the reference's own `len(train) / wall(bulk_fit)` including ingest
    (recommender.py:81,126) and `B / wall(recommend_batch)` (recommender.py:141-151) through the drop-in DataFrame API --
    Python lists in, Python lists out, one call for all users."""
    import contextlib
    import io
    import pandas as pd
    import torch
    from rtrec_amd import SLIM, Recommender
    U = X.shape[0]
    coo = X.tocoo()
    order = np.random.default_rng(0).permutation(coo.nnz)
    df = pd.DataFrame({"user": coo.row[order].astype(int), "item": coo.col[order].astype(int),
                       "tstamp": 1.7e9 + np.arange(coo.nnz, dtype=float), "rating": coo.data[order].astype(float)})
    rec = Recommender(SLIM(min_value=0, max_value=15, nn_feature_selection=K))
    sink = io.StringIO()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(sink):
        rec.bulk_fit(df, parallel=True)
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    users = list(range(U))
    rec.recommend_batch(users[:128], top_k=top_k)
    t0 = time.perf_counter()
    recs = rec.recommend_batch(users, top_k=top_k)          # first full-size call of the model: builds the large-pass layout
    t_first = time.perf_counter() - t0
    t_rec = []
    for _ in range(3):
        t0 = time.perf_counter()
        recs = rec.recommend_batch(users, top_k=top_k)
        t_rec.append(time.perf_counter() - t0)
    t_rec = float(np.median(t_rec))
    # the array-returning form of the same call (round 4: rtrec_amd extension, BaseModel.recommend_batch(as_arrays=True)):
    # an id array in, (ids[B, k], counts[B]) out -- no Python object per user or item on either side
    users_arr = np.arange(U, dtype=np.int64)
    rec.recommend_batch(users_arr, top_k=top_k, as_arrays=True)
    t_arr = []
    for _ in range(3):
        t0 = time.perf_counter()
        a_ids, a_cnt = rec.recommend_batch(users_arr, top_k=top_k, as_arrays=True)
        t_arr.append(time.perf_counter() - t0)
    t_arr = float(np.median(t_arr))
    arrays_equal_lists = bool(all(a_ids[b, :a_cnt[b]].tolist() == recs[b] for b in range(0, U, max(1, U // 4000))))
    t0 = time.perf_counter()
    for s0 in range(0, 20000, 100):            # the reference's evaluate() asks in 100-user batches (recommender.py:163-200)
        rec.recommend_batch(users[s0:s0 + 100], top_k=top_k)
    t_100 = time.perf_counter() - t0
    lat = []
    probe = np.random.default_rng(3).integers(0, U, 330).tolist()
    for x in probe:                                # the /recommend boundary (rtrec/serving/app.py:77-93): one user per call
        t0 = time.perf_counter()
        rec.recommend(x, top_k=top_k)
        lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.asarray(lat[30:])
    out = {"single_user_recommend_ms": {"p50": float(np.quantile(lat, .5)), "p99": float(np.quantile(lat, .99)), "requests": int(lat.size)},
           "bulk_fit_seconds": t_fit, "bulk_fit_samples_per_sec_incl_ingest": len(df) / t_fit,
           "recommend_batch_users": U, "recommend_batch_seconds": t_rec, "api_users_per_sec": U / t_rec,
           "recommend_batch_first_call_seconds": t_first,
           "recommend_batch_as_arrays_seconds": t_arr, "api_users_per_sec_arrays": U / t_arr,
           "arrays_equal_lists_on_sample": arrays_equal_lists,
           "recommend_batch_100_user_calls_users_per_sec": 20000 / t_100,
           "mean_list_length": float(np.mean([len(r) for r in recs[:5000]])),
           "note": "Recommender(SLIM(min_value=0, max_value=15, nn_feature_selection=K)).bulk_fit(DataFrame) and one "
                   "recommend_batch(list of all users, top_k) call: id mapping, upload of the row ids, kernels, download and "
                   "conversion to Python lists all inside the clock"}
    del rec, df
    torch.cuda.empty_cache()
    return out


def structured_leg(args, top_k):
    """The C3 shape with item-item structure (workload c3s) on one GPU: exact fit, layouts, all-users scoring through the
    general-W (segment) kernel, a sample of the answers checked against the C oracle.  Part of the default line so that the
    driver-visible record does not rest on the popularity-only generator alone."""
    import torch
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS["c3s"]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng = SlimEngine(device="cuda:0", tile_cols=args.tile_cols)
    eng.set_interactions(Xc, X)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d_tg, d_items, d_coef, d_count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True)
    torch.cuda.synchronize()
    fit_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng.set_weights(eng.merge_fit(None, I, False, d_tg, d_items, d_coef, d_count))
    lay = eng._layout(compact=True, top_k=top_k)
    torch.cuda.synchronize()
    to_score_s = time.perf_counter() - t0
    d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    step = lambda: eng.score_topk_device(None, U, top_k, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = step()
    torch.cuda.synchronize()
    cold_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
    eng.score_timer = eng.be.timer_create()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms, kn = eng.be.timer_read(eng.score_timer)
    eng.be.timer_destroy(eng.score_timer)
    eng.score_timer = 0
    W = eng.weights.to_csc(torch)
    Wr = W.tocsr()
    row_nnz_w = np.diff(Wr.indptr).astype(np.float64)
    gathered = float(row_nnz_w[X.indices].sum())
    algo_bytes = 8.0 * X.nnz + 8.0 * gathered + 8.0 * top_k * U + 4.0 * (U + 1)
    res = {"workload": f"c3s: {wl['desc']}", "n_users": U, "n_items": I, "nnz": int(X.nnz),
           "fit_seconds": fit_s, "fit_interactions_per_sec": X.nnz / fit_s, "mean_sweeps": float(n_iter.mean()),
           "W_nnz": int(W.nnz), "w_rows": int(np.count_nonzero(row_nnz_w)), "active_columns": int(lay["n_cols"]),
           "to_score_ms": to_score_s * 1e3, "score_path": eng.last_score_path,
           "ms_per_step": dt / args.steps * 1e3, "users_per_sec": U * args.steps / dt, "cold_step_ms": cold_ms,
           "kernel_ms_avg": kms / max(kn, 1),
           "algorithmic_bytes_per_launch": algo_bytes, "algorithmic_GBps": algo_bytes / (kms / max(kn, 1) * 1e-3) / 1e9,
           "algorithmic_frac_of_hbm_peak": algo_bytes / (kms / max(kn, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "topk_ids_crc32": zlib.crc32(out[0].cpu().numpy().tobytes())}
    # roofline of the general-W (segment) kernel in the driver's own line: kernel time by HIP
    # events over the timed steps (main kernel + the workgroup-per-long-user pass it forks), the bytes that MUST cross HBM
    # (user rows in, lists out, W once), and the counter traffic of the same kernel of the SAME build when a profile exists
    import glob
    from rtrec_amd import build as _build
    kern_s = kms / max(kn, 1) * 1e-3
    compulsory = 8.0 * X.nnz + 4.0 * (U + 1) + 12.0 * top_k * U + 4.0 * U + 8.0 * float(W.nnz)
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3s_pmc_traffic.json")))
    tj, stale = attach_profile(paths[-1] if paths else None, eng.last_score_path and "score_seg_kernel", _build.fingerprint())
    traffic = tj["hbm_bytes_per_launch_corrected"] if tj else None
    res["roofline"] = {"kernel": "score_seg_kernel (+ score_seg_heavy_kernel)", "bound": "hbm", "kernel_ms_avg": kern_s * 1e3,
                       "achieved": compulsory / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": compulsory / kern_s / 1e9 / HBM_PEAK_GBS, "compulsory_bytes_per_launch": compulsory,
                       "traffic": traffic, "traffic_vs_compulsory": (traffic / compulsory if traffic else None),
                       "traffic_source": (tj or {}).get("source"), "traffic_stale": stale, "traffic_measured_in_run": False,
                       "l2_hit_rate": (tj or {}).get("l2_hit_rate"),
                       "note": "W (2.9 MB of records) is L2-resident: the SURVEY 8d figure (algorithmic_*) prices gathered W entries "
                               "that never reach DRAM; `frac` is on the compulsory bytes, `traffic` is what the fabric counters saw"}
    # the other forms the reference scores in: DENSE mode (string item ids: every column competes, slim_elastic.py:745-778)
    # and a float64 W (its serial fit, :252) -- both through the fast pass plus the rows it flags
    other = {}

    def timed_mode(name, mode):
        stepm = lambda: eng.score_topk_device(None, U, top_k, True, mode, d_rows=d_rows, xb=xb)
        o = stepm()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            o = stepm()
        torch.cuda.synchronize()
        other[name] = {"ms_per_step": (time.perf_counter() - t1) / 3 * 1e3, "score_path": eng.last_score_path}
        return o
    timed_mode("dense_mode", _native.TOPK_DENSE)
    eng.set_weights(eng.weights, acc_f64=True)
    o64 = timed_mode("float64_w", _native.TOPK_SPARSE)
    other["float64_w"]["same_ids_as_float32"] = bool(np.array_equal(o64[0].cpu().numpy(), out[0].cpu().numpy()))
    eng.set_weights(eng.weights, acc_f64=False)
    res["other_modes"] = other
    if not args.no_cpu_baseline:
        from oracle import slim_oracle as so
        rows_s = np.sort(np.random.default_rng(7).choice(U, 1024, replace=False))
        o_ids, o_sc, o_cnt = so.recommend_batch(X[rows_s], Wr, top_k=top_k, n_threads=max(1, min(16, len(os.sched_getaffinity(0)))))
        res["oracle_sample_ids_identical"] = bool(np.array_equal(o_ids, out[0].cpu().numpy()[rows_s])
                                                  and np.array_equal(o_sc.view(np.uint32), out[1].cpu().numpy()[rows_s].view(np.uint32)))
    del eng
    torch.cuda.empty_cache()
    return res


def scale_leg(name, args, rank, world, local_rank, shard_auto):
    """The same measurement as the main line on another BASELINE config, on ALL ranks: sharded exact fit, W merged on the
    device, K timed all-users scoring steps between barriers (MAX over ranks).  bench.py --gpus N carries it for c4 -- the
    1 M x 500k config `north_star` quotes its >= 6x at 8 GPUs on -- so that a scaling run records that shape next to the
    ML-20M-shape headline."""
    import torch
    import torch.distributed as dist
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[name]
    U, I, K, top_k = wl["U"], wl["I"], wl["K"], args.top_k
    t0 = time.time()
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    gen_s = time.time() - t0
    eng = SlimEngine(device=f"cuda:{local_rank}", rank=rank, world_size=world, tile_cols=args.tile_cols,
                     score_shard=args.score_shard, shard_w=args.shard_w)
    eng.set_interactions(Xc, X)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    mine = eng.owned_columns(np.arange(I))
    d_tg, d_items, d_coef, d_count, n_iter = eng.fit_columns(mine, nn_feature_selection=K, device_out=True)
    torch.cuda.synchronize()
    fit_s = time.time() - t0
    dw = eng.merge_fit(None, I, False, d_tg, d_items, d_coef, d_count)
    if shard_auto:
        eng.score_shard = "columns" if dw.nnz > (1 << 28) else "rows"
    eng.set_weights(dw)
    d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    step = lambda: eng.score_topk_device(None, U, top_k, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb,
                                         with_scores=(world == 1 or args.exchange_scores))
    out = step()
    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()
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
    if world > 1:
        t = torch.tensor([dt, fit_s], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, fit_s = float(t[0].item()), float(t[1].item())
    res = {"workload": f"{name}: {wl['desc']}", "n_users": U, "n_items": I, "nnz": int(X.nnz), "value": U * args.steps / dt, "unit": "users/s",
           "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "fit_seconds": fit_s, "fit_interactions_per_sec": X.nnz / fit_s,
           "score_shard": eng.score_shard if world > 1 else "single GPU", "score_path": eng.last_score_path, "W_nnz": int(dw.nnz),
           "topk_ids_crc32": zlib.crc32(out[0].cpu().numpy().tobytes()), "generate_seconds": gen_s,
           "note": "same protocol as the main line: barrier + synchronize on both sides of the K steps, MAX over ranks; ids CRC is the "
                   "same for every --gpus N (the sharded pass returns the unsharded answer)"}
    del eng
    torch.cuda.empty_cache()
    return res


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--tile-cols", type=int, default=4096)
    ap.add_argument("--top-k", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast-fit", action="store_true", help="skip the second (tolerance-mode) fit")
    ap.add_argument("--stream-batches", type=int, default=8,
                    help="streaming leg (N = 1 only): this many 1,000-interaction SLIM.fit mini-batches per fit mode, each "
                         "followed by a 100-user recommend_batch, on a model bulk-fitted at the same workload shape (0 = skip)")
    ap.add_argument("--score-shard", default="auto", choices=["auto", "columns", "rows"],
                    help="multi-GPU scoring: item-column shards of W + list exchange (BASELINE.json's configuration), or "
                         "user-row shards with W replicated.  auto = rows unless W is too large to replicate (the single-GPU shard "
                         "models of round 4 favour rows for the feature-row AND the segment kernel: DESIGN.md section 6); the other "
                         "division is timed right after and reported as `alt_sharding`, and the line says which was chosen and why "
                         "(`score_shard_choice`)")
    ap.add_argument("--shard-w", action="store_true",
                    help="multi-GPU, column shards: every rank fits and KEEPS only its own column block of W (no all-gather of the "
                         "coefficients; SlimEngine.shard_w).  Implies --score-shard columns and no alt_sharding leg")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU time budget per cpu_baseline leg (4 legs)")
    ap.add_argument("--no-structured", action="store_true",
                    help="skip the `structured` leg of the default (c3) line: the same shape with item-item structure (c3s), "
                         "fit + all-users scoring through the general-W kernel")
    ap.add_argument("--no-api", action="store_true", help="skip the `api` leg (Recommender.bulk_fit / recommend_batch through the DataFrame API)")
    ap.add_argument("--exchange-scores", action="store_true",
                    help="several ranks, user-row shards: all-gather the float32 scores with the ids (84 instead of 44 bytes per user)")
    ap.add_argument("--no-c4", action="store_true",
                    help="skip the `c4` leg of the default (c3) line: BASELINE config 4's shape (1 M x 500k, 100 M interactions), "
                         "sharded fit + all-users scoring at this --gpus N (about 45 s)")
    args = ap.parse_args()
    shard_auto = args.score_shard == "auto" and not args.shard_w
    if args.shard_w:
        args.score_shard = "columns"
    if args.score_shard == "auto":
        args.score_shard = "columns"        # provisional: decided from the fitted W below (the engine re-shards in place)

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
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
        for out_line in proc.stdout:            # stdout carries exactly the JSON line; library chatter goes to stderr
            if out_line.lstrip().startswith("{"):
                sys.stdout.write(out_line)
                sys.stdout.flush()
            else:
                sys.stderr.write(out_line)
        sys.exit(proc.wait())

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
        # several PROCESSES time-slice one GPU here: the fork / join of the segment path's second stream then waits a
        # scheduling slice per event (measured: 6 ms -> 307 ms per c3s step at two ranks).  One rank per GPU is unaffected.
        os.environ.setdefault("RTREC_AMD_SG_FORK", "0")
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
    from rtrec_amd.synth import workload_matrix

    wl = WORKLOADS[args.workload]
    U, I, K, top_k = wl["U"], wl["I"], wl["K"], args.top_k
    t0 = time.time()
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    nnz = int(X.nnz)
    if rank == 0:
        log(f"[bench] workload {args.workload}: {U} x {I}, nnz={nnz} generated in {time.time() - t0:.1f}s")

    eng = SlimEngine(device=f"cuda:{local_rank}", rank=rank, world_size=world, tile_cols=args.tile_cols,
                     score_shard=args.score_shard, shard_w=args.shard_w)
    eng.set_interactions(Xc, X)

    # ------------------------------------------------------------------ fit (each rank: its own columns)
    lo, hi = shard_bounds(I, world, rank)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    mine = eng.owned_columns(np.arange(I))       # this rank's fit targets, balanced by column length
    d_tg, d_items, d_coef, d_count, n_iter = eng.fit_columns(mine, nn_feature_selection=K, device_out=True)
    torch.cuda.synchronize()
    fit_local = time.time() - t0
    # W stays on the device: the write-back (and, with several ranks, the exchange of the triples) and the score
    # layouts are tensor ops there; the host copies below only feed this script's bookkeeping
    t1 = time.time()
    dw_fit = eng.merge_fit(None, I, False, d_tg, d_items, d_coef, d_count)
    # --score-shard auto: by what one rank's share of the pass costs (tools/shard_model.py on one GPU, round 4,
    # profiles/r04_shard_model_*.jsonl: local kernel time of the slowest rank at 1 / 2 / 4 / 8 ranks, before the exchange):
    #   C3  (feature rows)  rows 1.79x / 2.94x / 4.00x   columns 1.21x / 1.69x / 2.10x
    #   c3s (segments)      rows 1.83x / 2.57x / 3.53x   columns 1.17x / 1.14x / 1.23x
    #   C4  (feature rows)  rows 1.89x / 3.50x / 4.08x (6.4x at 8 with the long-row setup and the giant rows spread over the waves)   columns 1.19x / 1.12x / 1.11x
    # A pass costs per USER (row setup, bound rows, the tiles every user opens) with either kernel, so dividing the users
    # divides the work and dividing the columns mostly repeats it on every rank; W is 0.2-3 MB here, so replicating it is
    # free.  The item-column division of BASELINE.json is the one to take when W is too large to replicate (auto: more than
    # 2^28 stored weights, 3 GB of layouts per rank) -- then with --shard-w semantics (SlimEngine.shard_w).
    w_rows = int(torch.unique(dw_fit.rows).numel()) if dw_fit.nnz else 0
    shard_choice = {"mode": args.score_shard, "chosen_by": "flag", "w_rows": w_rows, "w_nnz": int(dw_fit.nnz)}
    if shard_auto:
        big_w = dw_fit.nnz > (1 << 28)
        args.score_shard = "columns" if big_w else "rows"
        eng.score_shard = args.score_shard
        shard_choice = {"mode": args.score_shard, "chosen_by": "auto", "w_rows": w_rows, "w_nnz": int(dw_fit.nnz),
                        "layout": "feature rows" if w_rows <= 128 else "segments",
                        "why": ("W too large to replicate: item-column shards" if big_w else
                                "a pass costs per user with either kernel and W is small enough to replicate: user-row shards "
                                "(one rank's share at 8 ranks, single-GPU model: C3 4.0x rows / 2.1x columns, c3s 3.5x / 1.2x, "
                                "C4 6.4x / 1.1x; profiles/r04_shard_model_*.jsonl)")}
    eng.set_weights(dw_fit)
    torch.cuda.synchronize()
    merge_s = time.time() - t1
    t1 = time.time()
    eng._layout(compact=True, top_k=top_k)
    torch.cuda.synchronize()
    layout_s = time.time() - t1
    tg, items, coef, count = (t.cpu().numpy() for t in (d_tg, d_items, d_coef, d_count))
    tg = tg.astype(np.int64)
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
        t = torch.tensor([fit_local], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        fit_s = float(t.item())
    else:
        fit_s = fit_local
    W = eng.gather_weights(eng.weights).to_csc(torch)      # (a column-sharded W is gathered for the bookkeeping below: a collective)
    if rank == 0:
        log(f"[bench] W write-back on the device {merge_s * 1e3:.1f} ms, score layouts built on the device {layout_s * 1e3:.1f} ms "
            f"(W nnz={W.nnz})")
    # the same fit in the tolerance mode (exact=False: Gram-form CD / tree-reduced dots), timed beside the exact one;
    # the scored W is the exact one
    fit_fast = None
    if not args.no_fast_fit:
        We = merge_coefficients(None, I, *coefficients_to_updates(tg, items, coef, count))
        fit_fast = {"note": "rank 0's columns, same call as the exact fit with mode=... (rtrec_fit_opts.fast): same features, "
                            "coordinate sequence and stopping rules, dot products not in scikit-learn's left-to-right order"}
        for mode_name in ("shuffle", "gram"):
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.time()
            tg_f, items_f, coef_f, count_f, n_iter_f = eng.fit_columns(mine, nn_feature_selection=K, mode=mode_name)
            torch.cuda.synchronize()
            fast_s = time.time() - t0
            if world > 1:
                t = torch.tensor([fast_s], device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                fast_s = float(t.item())
            Wf = merge_coefficients(None, I, *coefficients_to_updates(tg_f, items_f, coef_f, count_f))
            diff = abs(Wf - We)
            same = n_iter_f[np.argsort(tg_f)] == n_iter[np.argsort(tg)]
            col_same = np.zeros(I, bool)
            col_same[np.sort(tg)[same]] = True
            dcoo = diff.tocoo()
            d_same = float(dcoo.data[col_same[dcoo.col]].max()) if dcoo.nnz and col_same[dcoo.col].any() else 0.0
            fit_fast[mode_name] = {"seconds": fast_s, "interactions_per_sec": nnz / fast_s, "speedup_vs_exact": fit_s / fast_s,
                                   "max_coef": float(abs(We).max()) if We.nnz else 0.0,
                                   "max_abs_coef_diff_vs_exact": float(diff.max()) if diff.nnz else 0.0,
                                   "max_abs_coef_diff_where_sweeps_agree": d_same,
                                   "targets_with_other_sweep_count": int((~same).sum()), "targets": int(len(tg))}
            if rank == 0:
                log(f"[bench] fit ({mode_name}): {fast_s:.2f}s ({nnz / fast_s:,.0f} interactions/s), max |dW| = "
                    f"{fit_fast[mode_name]['max_abs_coef_diff_vs_exact']:.3g} ({d_same:.3g} where the sweep counts agree: "
                    f"{int(same.sum())} of {len(tg)}) of {fit_fast[mode_name]['max_coef']:.3g}")

    # ------------------------------------------------------------------ score: K timed steps
    row_ids = np.arange(U, dtype=np.int32)
    d_rows = eng.be.to_dev(row_ids)
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])

    def step():
        # world == 1: one fused launch.  world > 1: the engine's sharded path (SlimEngine.score_topk_device)
        # -- local top-k per column shard in row chunks, an all-to-all of the per-shard lists per chunk
        # overlapped with the next chunk's kernel, strided merge of this rank's slice, all-gather of the
        # final lists; or, with --score-shard rows, this rank's slice of the users against all of W.
        # (several ranks, user-row shards: ids + counts travel, 44 B per user -- recommend_batch hands out item ids; with
        # --exchange-scores the float32 scores travel too, 84 B per user)
        return eng.score_topk_device(None, U, top_k, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb,
                                     with_scores=(world == 1 or args.exchange_scores))

    # Honest step accounting: the first pass after a new X / W pays for the work order of the rows
    # (SlimEngine._row_order: an index of X for the layout in use, cached afterwards) -- timed here on its own and as
    # part of the first ("cold") step; the K timed steps below reuse the cached order.
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = step()
    torch.cuda.synchronize()
    cold_step_ms = (time.perf_counter() - t1) * 1e3
    row_order_ms = None
    if world == 1:
        lay0 = eng._layout(True, top_k)
        eng._X.pop("_orders", None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng._row_order(d_rows, U, xb, lay0)
        torch.cuda.synchronize()
        row_order_first_ms = (time.perf_counter() - t1) * 1e3
        # a row set scored a second time gets the pattern-grouped order where the layout wants one (SlimEngine._row_order):
        # that is the order the timed steps run in -- its build is timed here, outside their clock
        t1 = time.perf_counter()
        eng._row_order(d_rows, U, xb, lay0)
        torch.cuda.synchronize()
        row_order_ms = row_order_first_ms + (time.perf_counter() - t1) * 1e3
    for _ in range(args.warmup):
        out = step()
    lib = eng.be.lib
    torch.cuda.synchronize()
    prof_fn = getattr(lib, "rtrec_amd_score_profile", None) if os.environ.get("RTREC_AMD_LIB") else None
    if prof_fn is not None:          # diagnostic build (-DSCORE_PROFILE): per-phase clocks of the sparse kernel
        prof_fn(None, 1)
    eng.score_timer = eng.be.timer_create()
    eng.rescored = torch.zeros(1, dtype=torch.int32, device=f"cuda:{local_rank}")
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
    n_rescored = int(eng.rescored.item())
    eng.rescored = None
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
    crc_main = zlib.crc32(out[0].cpu().numpy().tobytes())
    if out[1] is None:          # the timed exchange carried ids + counts only: the bookkeeping below reads the scores too
        out = eng.score_topk_device(None, U, top_k, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb, with_scores=True)

    # the other way of dividing the scoring pass over the ranks, timed the same way (K steps, barrier + synchronize on both
    # sides, max over ranks) right after the pass of record; the answers must agree
    alt_sharding = None
    if world > 1 and not args.shard_w:         # (a column-sharded W has no replicated copy to divide the users against)
        other = "columns" if args.score_shard == "rows" else "rows"
        dw = eng.weights
        eng.score_shard = other
        eng.set_weights(dw)                    # this rank's layouts for the other division (built on the device)
        for _ in range(max(args.warmup, 1)):
            out_alt = step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            out_alt = step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t1], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_alt = float(t.item())
        alt_sharding = {"score_shard": other, "ms_per_step": dt_alt / args.steps * 1e3, "users_per_sec": U * args.steps / dt_alt,
                        "same_topk_ids": bool(zlib.crc32(out_alt[0].cpu().numpy().tobytes()) == crc_main)}
        eng.score_shard = args.score_shard
        eng.set_weights(dw)
        step()                                 # the layouts of the division of record again (the models below read them)
        torch.cuda.synchronize()

    # host-buffer boundary (SLIMElastic.recommend_batch hands over a scipy CSR): upload the user
    # rows over PCIe, score, download ids + scores.  Reported beside `value`, never as `value`.
    pcie_users_per_s = None
    if world == 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        eng.recommend_csr(X, top_k=top_k, filter_interacted=True, mode=_native.TOPK_SPARSE)
        pcie_users_per_s = U / (time.perf_counter() - t1)

    # ------------------------------------------------------------------ what bounds the dominant score kernel (this rank)
    # SURVEY 8d prices the path in "algorithmic bytes" (8 B per gathered W entry), but W is ~1 MB and never
    # leaves L2 / LDS: that figure is kept as `algorithmic`, and the kernel is priced against the bounds that
    # can bind -- vector-ALU issue, LDS reads, L2 -> LDS staging -- plus its compulsory HBM bytes.
    Wr = W.tocsr()
    if world > 1 and args.score_shard == "rows":      # this rank: its slice of the users against all of W
        row_nnz_w = np.diff(Wr.indptr).astype(np.float64)
        Xs = X[rank::world]
        gathered_entries = float(row_nnz_w[Xs.indices].sum())
        algo_bytes = 8.0 * Xs.nnz + 8.0 * gathered_entries + 8.0 * top_k * Xs.shape[0] + 4.0 * (Xs.shape[0] + 1)
        users_per_item = np.bincount(Xs.indices, minlength=I).astype(np.float64)
        n_scored = Xs.shape[0]
    else:
        shard_row_nnz = np.diff(Wr[:, lo:hi].tocsr().indptr).astype(np.float64) if hi > lo else np.zeros(I)
        users_per_item = np.diff(Xc.indptr).astype(np.float64)
        gathered_entries = float((users_per_item * shard_row_nnz).sum())
        algo_bytes = 8.0 * nnz + 8.0 * gathered_entries + 8.0 * top_k * U + 4.0 * (U + 1)
        n_scored = U
    kern_ms = tot_ms.value / max(n_launch.value, 1)
    kern_s = kern_ms * 1e-3
    lay = eng._layout(True) or {"tile_cols": None, "n_tiles": 0, "n_cols": 0}
    VALU_PEAK_TFLOPS = 78.6        # 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz, one flop per lane-op: the sums are NOT fused
                                   # (x * w is rounded before it is added, like scipy's) so FMA's factor 2 does not apply
    LDS_READ_PEAK_GBS = 150_000.0  # MI355X_MICROARCH.md: ~150 TB/s aggregate for ds_read_b64/b128
    L2_PEAK_GBS = 34_500.0         # MI355X_MICROARCH.md: ~34.5 TB/s
    bounds, kernel_name, bound = {}, "score_sparse_kernel<float,false>", "l2"
    compulsory_hbm = 8.0 * (nnz if n_scored == U else int(Xs.nnz)) + 4.0 * (n_scored + 1) + (8.0 * top_k + 4.0) * n_scored
    fr = lay.get("fr_host")

    def tiled_model():
            l2_bytes = 6.0 * gathered_entries + 24.0 * (nnz if n_scored == U else int(Xs.nnz)) * max(lay["n_tiles"], 1)
            lds_bytes = 8.0 * gathered_entries
            bounds = {"l2": {"achieved": l2_bytes / kern_s / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                             "frac": l2_bytes / kern_s / 1e9 / L2_PEAK_GBS, "bytes_per_launch": l2_bytes},
                      "lds": {"achieved": lds_bytes / kern_s / 1e9, "peak": 44_000.0, "unit": "GB/s",
                              "frac": lds_bytes / kern_s / 1e9 / 44_000.0, "bytes_per_launch": lds_bytes,
                              "note": "read-modify-write of the LDS accumulators, priced at the LDS write rate"}}
            return bounds, "score_sparse_kernel<float,false>"

    def feature_row_model():
            tc, R = fr["fr_tile_cols"], fr["fr_rows"]
            # users per wave: the library's choice for this launch (csrc/score.hip: streaming layout 8 from ~393k rows, resident
            # layout from ~98k; 4 from ~25k, else 2)
            uw = 8 if n_scored >= (24 if fr.get("fr_resident") else 96) * 4096 else (4 if n_scored >= 6 * 4096 else 2)
            kernel_name = f"score_frows_kernel<{tc // 64},{2 if R > 64 else 1},{uw}>"
            tr = fr["fr_rows_of_tile"].view(np.uint64).reshape(-1, 2)
            feat_items = np.flatnonzero(fr["fr_map"] >= 0)
            tiles_of_row = np.array([sum(((int(tr[t, f // 64]) >> (f % 64)) & 1) for t in range(tr.shape[0])) for f in range(R)],
                                    dtype=np.float64)
            blocks = float((users_per_item[feat_items] * tiles_of_row).sum())      # (user, row, tile) blocks that hold a weight
            flops = 2.0 * blocks * tc                                               # one rounded multiply + one rounded add per column
            # what the kernel executes: a wave sweeps, per tile, the rows that hold a weight there and that ANY of its 8 users
            # rates (one LDS read per row, applied to all 8): the union over the wave's users, in the order the engine hands
            # the rows over (position p of a 128-user job -> wave p % 16)
            Xsc = (X if n_scored == U else Xs)[:, feat_items].tocsr()
            own = np.zeros((n_scored + 1, R), dtype=bool)                          # last row: padding (no ratings)
            own[np.repeat(np.arange(n_scored), np.diff(Xsc.indptr)), Xsc.indices] = Xsc.data != 0
            order = eng._X.get("_order") if n_scored == U else None
            order = order.cpu().numpy().astype(np.int64) if order is not None else np.arange(n_scored, dtype=np.int64)
            if len(order) != n_scored:         # the sharded path scores in row chunks: the cached order is one chunk's
                order = np.arange(n_scored, dtype=np.int64)
            if fr.get("fr_resident"):          # every wave claims uw users: positions j, j + n_wj, ... (score_frows_kernel)
                n_wj = -(-n_scored // uw)
                pos = np.concatenate([order, np.full(uw * n_wj - n_scored, n_scored, dtype=np.int64)])
                own_or = own[pos.reshape(uw, n_wj)].any(axis=0)                     # [wave jobs, R]
            else:                              # jobs of 8 x uw users: a wave takes uw consecutive positions of the pattern-sorted
                pos = np.concatenate([order, np.full((-n_scored) % (8 * uw), n_scored, dtype=np.int64)])      # order, else p % 8
                own_or = own[pos.reshape(-1, 8, uw) if getattr(eng, "_order_grouped", False) else pos.reshape(-1, uw, 8)
                             ].any(axis=2 if getattr(eng, "_order_grouped", False) else 1)      # [jobs, 8 waves, R]
            nz_rt = np.array([[(int(tr[t, f // 64]) >> (f % 64)) & 1 for f in range(R)] for t in range(tr.shape[0])], dtype=np.int32)
            swept_rows = float((own_or.reshape(-1, R).astype(np.int32) @ nz_rt.T).sum())      # (wave, tile, row) reads of 1 slice row
            lds_bytes = swept_rows * tc * 4.0                 # upper bound: tiles pruned by the score bound are not read
            executed_flops = 2.0 * swept_rows * float(uw) * tc
            n_jobs = 256 if fr.get("fr_resident") else -(-n_scored // (8 * uw))     # resident: W is loaded once per workgroup
            l2_bytes = float(n_jobs) * float(fr["fr_super_kb"][-1]) * 1024.0 + 8.0 * (nnz if n_scored == U else int(Xs.nnz))
            bounds = {"valu": {"achieved": flops / kern_s / 1e12, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": flops / kern_s / 1e12 / VALU_PEAK_TFLOPS, "flops_per_launch": flops,
                               "useful_flops_per_launch": 2.0 * gathered_entries, "unpruned_lockstep_flops_per_launch": executed_flops,
                               "note": "algorithmic work of the feature-row formulation: unfused float32 multiply + add over every "
                                       "(user, rated row of W, tile) block that holds a weight (zeros inside a block included); "
                                       "`unpruned_lockstep` is what the sweep would execute without pruning (a wave applies each swept "
                                       "row to all 8 of its users); the kernel skips every tile whose score bound sum|x|max|w| cannot "
                                       "beat its users' current (k+1)-th best, so it executes far less than either figure; "
                                       "peak = vector f32 lane-ops/s without FMA"},
                      "lds": {"achieved": lds_bytes / kern_s / 1e9, "peak": LDS_READ_PEAK_GBS, "unit": "GB/s",
                              "frac": lds_bytes / kern_s / 1e9 / LDS_READ_PEAK_GBS, "bytes_per_launch": lds_bytes,
                              "note": "slice rows read from LDS: one ds_read_b128 per lane and swept (wave, tile, row), shared by the wave's 8 users"},
                      "l2": {"achieved": l2_bytes / kern_s / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                             "frac": l2_bytes / kern_s / 1e9 / L2_PEAK_GBS, "bytes_per_launch": l2_bytes,
                             "note": "W slices staged into LDS once per 64-user job (once per workgroup when resident) + the user rows"}}
            return bounds, kernel_name

    def segment_model():
            """score_seg_kernel (general W): what it executes is estimated on a sample of users from the layout itself -- the
            tiles a user must open (score bound >= its final (k+1)-th best score: a lower bound of what the kernel opens, which
            learns that threshold as it goes) and the 8-byte records of the user's segments in them."""
            sg = lay["sg"]
            T, nt = int(sg["sg_T"]), int(sg["sg_n_tiles"])
            info = sg["sg_info"].cpu().numpy()
            ptr = sg["sg_ptr"].cpu().numpy().view(np.uint32).astype(np.int64) & 0x7fffffff
            b = sg["sg_bound"].cpu().numpy().view(np.uint32)
            Bm = np.empty((b.shape[0], 128), np.float32)
            Bm[:, 0::2] = (b << 16).view(np.float32)
            Bm[:, 1::2] = (b & 0xffff0000).view(np.float32)
            rec = np.diff(ptr, axis=1).astype(np.float64)                           # records per (row, tile)
            rs = np.random.default_rng(5)
            Xall = X if n_scored == U else Xs
            us = np.sort(rs.choice(n_scored, min(3000, n_scored), replace=False))
            Xu = Xall[us]
            sc_u = out[1].cpu().numpy()[us if n_scored == U else np.arange(rank, U, world)[us]]
            theta = np.where(np.isfinite(sc_u[:, -1]), sc_u[:, -1], -np.inf)        # k-th best score (the (k+1)-th is <= it)
            rows_u = info[:, 0][Xu.indices]
            ok = rows_u >= 0
            ui = np.repeat(np.arange(Xu.shape[0]), np.diff(Xu.indptr))
            Bu = np.zeros((Xu.shape[0], 128))
            np.add.at(Bu, ui[ok], np.abs(Xu.data[ok])[:, None] * Bm[rows_u[ok]])
            need = (Bu[:, :nt] >= theta[:, None]) & (Bu[:, :nt] > 0)
            Ru = np.zeros((Xu.shape[0], nt))
            np.add.at(Ru, ui[ok], rec[rows_u[ok]])
            scale = n_scored / Xu.shape[0]
            rec_bytes = 8.0 * float((Ru * need).sum()) * scale
            bound_bytes = 256.0 * float(ok.sum()) * scale
            nnz_s = nnz if n_scored == U else int(Xs.nnz)
            l2_bytes = rec_bytes + bound_bytes + (8.0 + 8.0) * nnz_s + 8.0 * float(ok.sum()) * scale * float(need.sum(1).mean())
            name = ("score_seg_kernel<8, unsigned short, true> (+ score_seg_heavy_kernel for users with more than 512 items, on a second "
                    "stream beside it: the event bracket spans both)")
            bounds = {"hbm_algorithmic": {"achieved": algo_bytes / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": algo_bytes / kern_s / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": algo_bytes,
                                          "note": "SURVEY 8d: 8 B per (user item, stored weight of its row of W) + user rows + outputs, over the "
                                                  "kernels' time.  W (a few MB) is served by L2 and the kernel skips every tile whose score "
                                                  "bound cannot reach the user's list, so fewer bytes than this move: a rate of the path's "
                                                  "algorithmic work against the HBM peak, not DRAM traffic"},
                      "l2": {"achieved": l2_bytes / kern_s / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                             "frac": l2_bytes / kern_s / 1e9 / L2_PEAK_GBS, "bytes_per_launch": l2_bytes,
                             "tiles_opened_per_user_at_least": float(need.sum(1).mean()), "tiles": nt, "tile_cols": T,
                             "note": "estimate from a 3,000-user sample: records of the segments in the tiles a user must open, its bound "
                                     "rows, segment pointers and its own row"}}
            return bounds, name

    score_path = eng.last_score_path
    if score_path == "segments" and lay.get("sg") is not None:
        try:
            bounds, kernel_name = segment_model()
        except Exception as exc:
            log(f"[bench] segment bounds model failed ({exc!r}); falling back to the tiled-CSR model")
            bounds, kernel_name = tiled_model()
            kernel_name = "score_seg_kernel"
    elif fr is not None and eng.use_feature_rows:
        try:
            bounds, kernel_name = feature_row_model()
        except Exception as exc:      # the model is bookkeeping: it must never cost the measurement
            log(f"[bench] feature-row bounds model failed ({exc!r}); falling back to the tiled-CSR model")
            bounds, kernel_name = tiled_model()
            kernel_name = "score_frows_kernel"
    else:
        bounds, kernel_name = tiled_model()
    bounds["hbm"] = {"achieved": compulsory_hbm / kern_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": compulsory_hbm / kern_s / 1e9 / HBM_PEAK_GBS, "compulsory_bytes_per_launch": compulsory_hbm,
                     "note": "user rows + outputs: the only bytes that must come from / go to HBM"}
    if "valu" in bounds:        # feature-row kernel: priced on USEFUL flops (2 per gathered W entry), not on the zero padding inside blocks
        useful = bounds["valu"]["useful_flops_per_launch"] / kern_s / 1e12
        bounds["valu"].update(block_flops_frac=bounds["valu"]["frac"], achieved=useful, frac=useful / VALU_PEAK_TFLOPS)
    bound = max(bounds, key=lambda k: bounds[k]["frac"])        # the bound the kernel is closest to
    algorithmic = {"bytes_per_launch": algo_bytes, "GBps": algo_bytes / kern_s / 1e9,
                   "frac_of_hbm_peak": algo_bytes / kern_s / 1e9 / HBM_PEAK_GBS,
                   "exceeds_hbm_peak": bool(algo_bytes / kern_s / 1e9 > HBM_PEAK_GBS),
                   "note": "SURVEY 8d figure (8 B per gathered W entry); it prices W entries that never reach DRAM (W stays in L2 / LDS "
                           "and most of it is pruned), so where it exceeds the peak it is NOT a physical fraction of any hardware limit"}

    # the config the multi-GPU target is quoted on, at this N, on every rank (before the ranks other than 0 leave)
    c4_leg = None
    if args.workload == "c3" and not args.no_c4:
        try:
            c4_leg = scale_leg("c4", args, rank, world, local_rank, shard_auto)
            if rank == 0:
                log(f"[bench] c4 leg: fit {c4_leg['fit_seconds']:.2f}s, score {c4_leg['ms_per_step']:.2f} ms/step "
                    f"({c4_leg['value']:,.0f} users/s) over {c4_leg['score_shard']}")
        except Exception as exc:          # (every rank fails or none: the leg has no rank-dependent branch before its first collective)
            log(f"[bench] c4 leg failed on rank {rank}: {exc!r}")
            c4_leg = {"error": repr(exc)}

    # same value for every --gpus N: the sharded path returns the unsharded answer
    topk_crc = zlib.crc32(out[0].cpu().numpy().tobytes()) if rank == 0 else 0
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # Counter-backed figures come from separate rocprofv3 --pmc runs of this same command (they cannot be read
    # from inside the process); the committed summaries of the most recent round are attached.
    import glob

    def latest(pattern):
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        return paths[-1] if paths else None

    from rtrec_amd import build as _build
    fp = _build.fingerprint()
    traffic, traffic_src, counters, fit_traffic, stale = None, None, None, None, []
    if world == 1:
        tj, st_ = attach_profile(latest(f"r*_{args.workload}_pmc_traffic.json"), kernel_name, fp)
        if tj:
            traffic, traffic_src = tj["hbm_bytes_per_launch_corrected"], tj["source"]
        stale += [st_] if st_ else []
        counters, st_ = attach_profile(latest(f"r*_{args.workload}_score_counters.json"), kernel_name, fp)
        stale += [st_] if st_ else []
        fit_traffic, st_ = attach_profile(latest(f"r*_{args.workload}_fit_pmc.json"), None, fp)
        stale += [st_] if st_ else []
        for st_ in stale:
            log(f"[bench] {st_} was measured on another build of librtrec_amd.so: not attached")

    # What the counters say physically: the share of the launch during which a SIMD's vector ALU is issuing.
    # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the 1,024 SIMDs (MI355X_MICROARCH.md); the kernel time is the one this
    # run measured by events, the counters are those of the same build (attach_profile) -- so the figure mixes two runs of one build.
    valu_issue_busy = None
    if counters and (counters.get("per_launch") or {}).get("SQ_ACTIVE_INST_VALU"):
        simd_cycles = kern_s * 2.4e9
        valu_issue_busy = {"frac": counters["per_launch"]["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / simd_cycles,
                           "how": "SQ_ACTIVE_INST_VALU (quad-cycles, all SIMDs) x 4 / 1024 SIMDs / (kernel_ms_avg x 2.4 GHz)",
                           "wait_any_frac_of_wave_cycles": (counters.get("derived") or {}).get("wait_any_frac")}
    best = bounds[bound]
    line = {
        "metric": "users-scored/sec top-10 (SLIM recommend, int ids, filter_interacted) + fit interactions/sec in `fit`",
        "value": value, "unit": "users/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {wl['desc']}", "n_users": U, "n_items": I, "nnz": nnz,
                   "nn_feature_selection": K, "top_k": top_k, "tile_cols": lay["tile_cols"], "n_tiles": lay["n_tiles"],
                   "active_columns": lay["n_cols"],
                   "w_rows": (fr["fr_rows"] if fr is not None else (int(lay["sg"]["sg_rows"]) if lay.get("sg") else None)),
                   "score_layout": ("feature rows" if fr is not None else "segments" if lay.get("sg") else "tiled CSR"),
                   "parallelism": ("single GPU" if world == 1 else f"item-column shard x{world}" if args.score_shard == "columns"
                                   else f"user-row shard x{world}, W replicated"),
                   "exchange": (None if world == 1 else "per-shard records, all-to-all + all-gather of the final lists" if args.score_shard == "columns"
                                else ("ids + scores + counts, 84 B per user" if args.exchange_scores else "ids + counts, 44 B per user") +
                                f", all-gather in chunks of >= {eng.row_chunk_rows} slots per rank overlapped with the next chunk's kernel")},
        "ranks_seen": ranks_seen, "rank_devices": rank_devices, "backend": backend, "alt_sharding": alt_sharding,
        "score_shard_choice": dict(shard_choice, w_column_sharded=bool(args.shard_w)),
        "pcie_inclusive_users_per_sec": pcie_users_per_s, "topk_ids_crc32": topk_crc, "rows_rescored_by_exact_tie_pass": n_rescored,
        "fit": {"seconds": fit_s, "interactions_per_sec": nnz / fit_s, "columns_per_sec": I / fit_s,
                "W_nnz": int(W.nnz), "mean_sweeps": float(n_iter.mean()), "mode": "exact", "tolerance_modes": fit_fast,
                "to_score": {"write_back_ms": merge_s * 1e3, "layouts_ms": layout_s * 1e3,
                             "note": "fit output -> W merge -> score layouts, all on the device (engine.merge_fit, "
                                     "build_tiled_w_device / build_feature_rows_device); no host copy of W is on this path"},
                "roofline": {"kernel": "fit_columns_kernel<false> (+ fit_columns_mw_kernel for the heaviest targets)",
                             "bound": "hbm", "achieved": fit_algo_bytes / fit_local / 1e9, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": fit_algo_bytes / fit_local / 1e9 / HBM_PEAK_GBS,
                             "algorithmic_bytes": fit_algo_bytes, "seconds": fit_local, "traffic": fit_traffic,
                             "note": "rank 0's columns; compulsory bytes only (SURVEY 8d): the coordinate-descent sweeps "
                                     "re-read the K feature columns and stream the per-target residual, which is what "
                                     "`traffic` (PMC, profiles/) measures"}},
        "roofline": {"kernel": kernel_name, "bound": ("hbm" if bound.startswith("hbm") else bound), "achieved": best["achieved"], "peak": best["peak"],
                     "unit": best["unit"], "frac": best["frac"], "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_measured_in_run": False, "traffic_stale": (stale or None), "build": fp,
                     "traffic_vs_compulsory": (traffic / compulsory_hbm if traffic else None),
                     "kernel_ms_avg": kern_ms, "launches": int(n_launch.value), "bounds": bounds,
                     "algorithmic": algorithmic, "counters": counters, "valu_issue_busy": valu_issue_busy, "score_path": score_path},
        "c4": c4_leg,
        "step_accounting": {"cold_step_ms": cold_step_ms, "row_order_ms": row_order_ms, "warm_ms_per_step": ms_per_step,
                            "note": "cold = the first pass after a new X / W: it builds the work order of the rows (an index of X for the "
                                    "layout in use: argsort by row length, or by feature-row pattern for the feature-row kernel), which the "
                                    "timed steps reuse; row_order_ms is that build alone (the length order of the first pass plus, where the layout wants one, the pattern-grouped order a row set gets when it is scored a second time).  The layouts themselves are fit.to_score.layouts_ms"},
    }

    # ------------------------------------------------------------------ streaming leg (SURVEY 8 row S1 / BASELINE config 4 pattern)
    if world == 1 and args.stream_batches > 0:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        from stream_bench import SHAPES, run_open_loop, run_stream
        if args.workload in ("c2", "c3", "c3s", "small", "smalls") and args.workload in SHAPES:      # the c4 bulk load alone takes minutes
            line["streaming"] = run_stream(args.workload, batches=args.stream_batches, fit_modes=("exact", "gram"), log=log)
            # fixed-QPS open loop (BASELINE config 4's pattern): Poisson arrivals at 1k / 5k / 20k interactions per second,
            # the consumer fits whatever has arrived and a recommend makes it visible
            line["streaming"]["open_loop"] = run_open_loop(args.workload, rates=(1000.0, 5000.0, 20000.0), duration_s=3.0,
                                                           fit_modes=("exact", "gram"), log=log)

    # ------------------------------------------------------------------ API-level figures and the structured workload
    if world == 1 and not args.no_api and args.workload != "c4":          # (c4: 93 M DataFrame rows take minutes to ingest)
        try:
            line["api"] = api_leg(X, K, top_k)
            log(f"[bench] api: bulk_fit {line['api']['bulk_fit_seconds']:.2f}s ({line['api']['bulk_fit_samples_per_sec_incl_ingest']:,.0f} samples/s "
                f"incl. ingest), recommend_batch(all users) {line['api']['api_users_per_sec']:,.0f} users/s as lists, "
                f"{line['api']['api_users_per_sec_arrays']:,.0f} users/s as arrays")
        except Exception as exc:
            log(f"[bench] api leg failed: {exc!r}")
            line["api"] = {"error": repr(exc)}
    if world == 1 and args.workload == "c3" and not args.no_structured:
        try:
            line["structured"] = structured_leg(args, top_k)
            log(f"[bench] structured (c3s): fit {line['structured']['fit_seconds']:.2f}s, score {line['structured']['ms_per_step']:.2f} ms/step "
                f"({line['structured']['users_per_sec']:,.0f} users/s) through {line['structured']['score_path']}")
        except Exception as exc:
            log(f"[bench] structured leg failed: {exc!r}")
            line["structured"] = {"error": repr(exc)}
    if world == 1 and args.workload in ("c2", "small", "smalls") and not args.no_fast_fit:
        # the reference's DEFAULT model has no feature selection (nn_feature_selection=None, slim_elastic.py:182-190): every
        # item is a feature of every target (reported for C1-C2)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tg_a, items_a, coef_a, count_a, n_iter_a = eng.fit_columns(np.arange(I), nn_feature_selection=None)
        torch.cuda.synchronize()
        allf_s = time.perf_counter() - t1
        line["fit"]["all_features"] = {"seconds": allf_s, "interactions_per_sec": nnz / allf_s, "columns_per_sec": I / allf_s,
                                       "W_nnz": int(count_a.sum()), "mean_sweeps": float(n_iter_a.mean()),
                                       "note": "nn_feature_selection=None: fit_columns_kernel<true>, exact mode"}
        log(f"[bench] fit (K=None): {allf_s:.2f}s ({nnz / allf_s:,.0f} interactions/s)")

    # ------------------------------------------------------------------ cpu_baseline: the C oracle on this host,
    # one thread and all cores (POSIX threads over users / item columns: the reference's own parallel axis,
    # slim_elastic.py:296-301,358-366), bounded samples of the same workload
    if world == 1 and not args.no_cpu_baseline:
        from oracle import slim_oracle as so
        so.lib()
        # threads of the multi-core legs: the host's share for one GPU is 16 cores (the affinity mask of a GPU box
        # shows the whole machine); `cores` reports the threads actually used
        n_cores = max(1, min(16, len(os.sched_getaffinity(0))))
        rng = np.random.default_rng(7)
        ids_gpu = out[0].cpu().numpy()
        sample = rng.permutation(U)

        def score_leg(threads, budget, start):
            n, done, spent, ok = 64 * threads, 0, 0.0, True
            while spent < budget and start + done < U:
                rows_s = np.sort(sample[start + done:start + done + n])
                t1 = time.perf_counter()
                o_ids, _, _ = so.recommend_batch(X[rows_s], Wr, top_k=top_k, n_threads=threads)
                spent += time.perf_counter() - t1
                ok = ok and np.array_equal(o_ids, ids_gpu[rows_s])
                done += len(rows_s)
                n = min(n * 2, 8192 * threads)
            return {"value": done / max(spent, 1e-9), "unit": "users/s", "cores": threads, "kind": "port",
                    "sample": f"{done} random users of the same workload scored with the C oracle in {spent:.1f}s "
                              f"(top-k ids identical to the GPU: {ok})"}, done

        leg1, used = score_leg(1, args.cpu_seconds / 2, 0)
        # (a small workload can be finished by the first leg: the second then re-scores the sample from its start)
        legn, _ = score_leg(n_cores, args.cpu_seconds / 2, used if used + 64 * n_cores <= U else 0)
        line["cpu_baseline"] = dict(leg1, all_cores=legn)

        # fit: a STRATIFIED sample of target columns (by column length: the top 1 %, the next 19 %, the tail), each
        # stratum's mean time scaled to its size -> an estimate of the whole catalogue's fit time on this host,
        # comparable with the GPU's whole-catalogue interactions/s (400 random columns are nearly all tail)
        col_nnz = np.diff(Xc.indptr)
        by_len = np.argsort(-col_nnz, kind="stable")
        strata = [("top 1 %", by_len[:max(1, I // 100)]), ("next 19 %", by_len[max(1, I // 100):I // 5]), ("tail 80 %", by_len[I // 5:])]

        def fit_leg(threads, budget):
            est, parts = 0.0, []
            for (name, cols_s), share in zip(strata, (0.5, 0.3, 0.2)):
                pick = rng.permutation(cols_s)
                done, spent, n = 0, 0.0, max(2, threads)
                while spent < budget * share and done < len(pick):
                    c = np.sort(pick[done:done + n]).astype(np.int32)
                    t1 = time.perf_counter()
                    so.fit_columns(Xc, c, nn_feature_selection=K, n_threads=threads)
                    spent += time.perf_counter() - t1
                    done += len(c)
                    n = min(n * 2, 64 * threads)
                est += spent / max(done, 1) * len(cols_s)
                parts.append(f"{done} of {len(cols_s)} columns of the {name} in {spent:.1f}s")
            return {"value": nnz / est, "unit": "interactions/s", "cores": threads, "kind": "port",
                    "estimated_full_fit_seconds": est,
                    "sample": "stratified by column length, each stratum's mean time scaled to its size: " + "; ".join(parts)}

        f1 = fit_leg(1, args.cpu_seconds)
        fn = fit_leg(n_cores, args.cpu_seconds)
        line["fit"]["cpu_baseline"] = dict(f1, all_cores=fn)
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
