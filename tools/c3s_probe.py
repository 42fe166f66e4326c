#!/usr/bin/env python3
"""Round-3 probe: fit a structured workload on the GPU, describe the W it yields (rows, row lengths, score support),
time the scoring paths that exist for it, and save W (npz) under gpurun_out/ for layout studies on the CPU.

    python tools/c3s_probe.py --workload c3s [--save gpurun_out/c3s_W.npz]
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
    ap.add_argument("--workload", default="c3s", choices=sorted(WORKLOADS))
    ap.add_argument("--save", default="")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--mode", default="exact")
    ap.add_argument("--only-all", action="store_true", help="time the all-users pass only (profiling runs)")
    ap.add_argument("--check-tiled", action="store_true", help="compare the all-users pass with the tiled-CSR kernel on every row")
    args = ap.parse_args()
    import torch
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix

    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    t0 = time.time()
    X = workload_matrix(wl)
    Xc = X.tocsc()
    Xc.sort_indices()
    print(f"[probe] {args.workload}: {U} x {I}, nnz={X.nnz}, generated in {time.time() - t0:.1f}s", flush=True)
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    torch.cuda.synchronize()
    t0 = time.time()
    d_tg, d_items, d_coef, d_count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True, mode=args.mode,
                                                             alpha=wl.get("alpha", 0.1))
    torch.cuda.synchronize()
    fit_s = time.time() - t0
    print(f"[probe] fit ({args.mode}) {fit_s:.2f}s = {X.nnz / fit_s:,.0f} interactions/s, mean sweeps {n_iter.mean():.1f}, max {n_iter.max()}",
          flush=True)
    eng.set_weights(eng.merge_fit(None, I, False, d_tg, d_items, d_coef, d_count))
    W = eng.weights.to_csc(torch)
    Wr = W.tocsr()
    rl = np.diff(Wr.indptr)
    cl = np.diff(W.indptr)
    out = {"workload": args.workload, "nnz_X": int(X.nnz), "fit_s": fit_s, "W_nnz": int(W.nnz), "W_rows_nonempty": int((rl > 0).sum()),
           "W_cols_nonempty": int((cl > 0).sum()),
           "row_len_pct_50_90_99_100": [float(v) for v in np.percentile(rl, [50, 90, 99, 100])],
           "gathered_entries_per_pass": float(rl[X.indices].sum())}
    print("[probe] " + json.dumps(out), flush=True)
    if args.save:
        os.makedirs(os.path.dirname(args.save) or ".", exist_ok=True)
        np.savez_compressed(args.save, indptr=W.indptr, indices=W.indices, data=W.data, n_iter=n_iter)
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    lens = np.diff(X.indptr)
    sets = [("all users", np.arange(U, dtype=np.int32)), ("16k users", np.arange(16384, dtype=np.int32))]
    for cap in (2048, 512, 256):
        sets.append((f"users with <= {cap} items", np.flatnonzero(lens <= cap).astype(np.int32)))
    if args.only_all:
        sets = sets[:1]
    for label, rows in sets:
        d_rows = eng.be.to_dev(rows)
        nrows = len(rows)
        o = eng.score_topk_device(None, nrows, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
        torch.cuda.synchronize()
        eng.score_timer = eng.be.timer_create()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            o = eng.score_topk_device(None, nrows, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        prof = getattr(eng.be.lib, "rtrec_amd_seg_profile", None) if os.environ.get("RTREC_AMD_LIB") else None
        if prof is not None:          # diagnostic build (-DSCORE_PROFILE): per-phase clocks of score_seg_kernel, one more pass
            import ctypes as C
            prof(None, 1)
            eng.score_topk_device(None, nrows, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
            torch.cuda.synchronize()
            buf = (C.c_uint64 * 16)()
            prof(buf, 0)
            names = ["jobs", "claim", "items", "bounds", "next", "filter", "segptr", "acc", "scan", "emit", "n_tiles", "n_segs", "total"]
            v = dict(zip(names, [int(x) for x in buf]))
            tot = max(v["total"], 1)
            print("[seg profile] " + json.dumps({k: (v[k] if k in ("jobs", "total") or k.startswith("n_") else round(v[k] / tot, 4))
                                                 for k in names}), flush=True)
        kms, kn = eng.be.timer_read(eng.score_timer)
        eng.be.timer_destroy(eng.score_timer)
        eng.score_timer = 0
        path_used = eng.last_score_path
        if args.check_tiled and label == "all users":
            # every row against the tiled-CSR kernel (ids, score bits, counts)
            fast = [t.cpu().numpy() for t in o]
            eng.use_seg_layout = eng.use_feature_rows = False
            t0 = time.perf_counter()
            ot = eng.score_topk_device(None, nrows, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
            torch.cuda.synchronize()
            t_tiled = (time.perf_counter() - t0) * 1e3
            tiled = [t.cpu().numpy() for t in ot]
            eng.use_seg_layout = eng.use_feature_rows = True
            m = np.arange(10)[None, :] < fast[2][:, None]
            same = bool(np.array_equal(fast[2], tiled[2]) and np.array_equal(fast[0][m], tiled[0][m])
                        and np.array_equal(fast[1].view(np.uint32)[m], tiled[1].view(np.uint32)[m]))
            sg = (eng._fast_layout() or {}).get("sg") or {}
            print("[probe] " + json.dumps({"all_rows_equal_tiled_kernel": same, "tiled_pass_ms_cold": round(t_tiled, 2),
                                           "seg_T": int(sg.get("sg_T", 0)), "seg_tiles": int(sg.get("sg_n_tiles", 0)),
                                           "seg_rows": int(sg.get("sg_rows", 0)), "seg_cols": int(sg.get("sg_n_cols", 0))}), flush=True)
        lay = eng._layout(True, 10)
        print(f"[probe] score {label} ({nrows}): {ms:.2f} ms/pass, kernel {kms / max(kn, 1):.3f} ms ({nrows / ms * 1e3:,.0f} users/s), "
              f"path={path_used}, tiled layout {lay['n_tiles']} x {lay['tile_cols']}, active cols={lay['n_cols']}", flush=True)


if __name__ == "__main__":
    main()
