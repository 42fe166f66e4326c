#!/usr/bin/env python3
"""A/B sweep of the fit kernel's launch configuration on one workload (slots x kernel mode).

    python tools/fit_sweep.py --workload c3 --configs sw:5120,sw:2560,mw:512
Prints one JSON line per configuration: wall seconds of the bulk fit and a checksum of W so that
every configuration can be seen to produce the same coefficients.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--configs", default="sw:5120,sw:2560,sw:1280,mw:1024,mw:512,mw:256")
    ap.add_argument("--env", default="", help="extra KEY=VALUE pairs, comma separated, applied to every run")
    args = ap.parse_args()
    import torch
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix

    wl = WORKLOADS[args.workload]
    U, I, K = wl["U"], wl["I"], wl["K"]
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    for kv in filter(None, args.env.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
    for cfg in args.configs.split(","):
        mode, slots, heavy = (cfg.split(":") + ["0"])[:3]     # mode:slots[:heavy], mode auto = the library's choice
        if mode == "auto":
            os.environ.pop("RTREC_AMD_FIT_MODE", None)
        else:
            os.environ["RTREC_AMD_FIT_MODE"] = mode
        os.environ["RTREC_AMD_FIT_SLOTS"] = slots
        os.environ["RTREC_AMD_FIT_HEAVY"] = heavy
        eng._fit_ws.clear()           # (a cached larger scratch would be reused as it is: the kernel launches min(its slots, targets))
        eng.fit_columns(np.arange(64), nn_feature_selection=K)      # workspace + warm-up
        torch.cuda.synchronize()
        t0 = time.time()
        tg, items, coef, count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K, trace=True)
        torch.cuda.synchronize()
        wall = time.time() - t0
        o = np.argsort(tg)
        crc = zlib.crc32(items[o].tobytes()) ^ zlib.crc32(coef[o].tobytes()) ^ zlib.crc32(n_iter[o].tobytes())
        tr = eng.last_fit_stats["trace"].astype(np.float64)
        dur = (tr[:, 2] - tr[:, 0]) * 1e-8
        print(json.dumps({"workload": args.workload, "mode": mode, "slots": int(eng.last_fit_stats["slots"]),
                          "heavy": int(eng.last_fit_stats["n_heavy"]),
                          "heavy_span_s": float((tr[:max(eng.last_fit_stats["n_heavy"], 1), 2].max() - tr[:, 0].min()) * 1e-8),
                          "fit_s": wall, "interactions_per_s": X.nnz / wall, "sum_target_s": float(dur.sum()),
                          "max_target_s": float(dur.max()), "crc": crc}), flush=True)


if __name__ == "__main__":
    main()
