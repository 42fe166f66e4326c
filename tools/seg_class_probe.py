#!/usr/bin/env python3
"""Where the general-W scoring pass spends its time by user length class: passes over chosen subsets of the c3s users
(all <= 256-item users plus a fraction of the 257..512 class; the mid class alone; the long users alone).  Linear growth
with the fraction = throughput-bound work; a jump at the first few = a latency tail.   python tools/seg_class_probe.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from bench import WORKLOADS
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS["c3s"]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl)
    Xc = X.tocsc(); Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    d = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True, mode="gram")
    eng.set_weights(eng.merge_fit(None, I, False, *d[:4]))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
    lens = np.diff(X.indptr)
    short = np.flatnonzero(lens <= 256)
    mid = np.flatnonzero((lens > 256) & (lens <= 512))
    long_ = np.flatnonzero(lens > 512)
    rng = np.random.default_rng(1)
    rng.shuffle(mid)
    sets = [("<=256", short)]
    for f in (0.1, 0.25, 0.5, 1.0):
        sets.append((f"<=256 + {f:.2f} of 257..512", np.concatenate([short, mid[:int(f * len(mid))]])))
    sets += [("257..512 alone", mid), ("129..256 alone", np.flatnonzero((lens > 128) & (lens <= 256))), ("> 512 alone", long_),
             ("<= 512 + > 512 (all)", np.arange(U))]
    for label, rows in sets:
        rows = np.sort(rows).astype(np.int32)
        d_rows = eng.be.to_dev(rows)
        n = len(rows)
        for _ in range(2):
            eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
        torch.cuda.synchronize()
        rec = {"set": label, "users": n, "ms_per_pass": round((time.perf_counter() - t0) / 5 * 1e3, 3), "path": eng.last_score_path}
        prof = getattr(eng.be.lib, "rtrec_amd_seg_heavy_profile", None) if os.environ.get("RTREC_AMD_LIB") else None
        if prof is not None:          # diagnostic build (-DSCORE_PROFILE): phase clocks of the workgroup-per-user kernel, one more pass
            import ctypes as C
            prof(None, 1)
            eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
            torch.cuda.synchronize()
            buf = (C.c_uint64 * 16)()
            prof(buf, 0)
            names = ["users", "setup", "sync1", "next", "trow", "acc", "scan", "sync2", "merge", "clean", "tiles", "total"]
            v = dict(zip(names, [int(x) for x in buf]))
            tot = max(v["total"], 1)
            rec["heavy_profile"] = {k: (v[k] if k in ("users", "tiles", "total") else round(v[k] / tot, 4)) for k in names}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
