#!/usr/bin/env python3
"""Exact bulk fit against the size of the shared Gram matrix (Gram tracking, csrc/fit.hip): build time of G and the fit
with G already built.    python tools/gram_sweep.py --workload c3s --items 512,2048,4096,8192"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3s", choices=sorted(WORKLOADS))
    ap.add_argument("--items", default="512,2048,4096")
    args = ap.parse_args()
    import torch
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    X = workload_matrix(wl)
    Xc = X.tocsc()
    Xc.sort_indices()
    I, K = wl["I"], wl["K"]
    crc0 = None
    for n in [int(v) for v in args.items.split(",")]:
        os.environ["RTREC_AMD_GRAM_ITEMS"] = str(n)
        eng = SlimEngine(device="cuda:0")
        eng.set_interactions(Xc, X)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        t0 = time.perf_counter()
        d_tg, d_items, d_coef, d_count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True)
        torch.cuda.synchronize()
        second = time.perf_counter() - t0
        import zlib
        o = torch.argsort(d_tg)
        crc = zlib.crc32(d_coef[o].cpu().numpy().tobytes()) ^ zlib.crc32(np.ascontiguousarray(n_iter[o.cpu().numpy()]).tobytes())
        crc0 = crc if crc0 is None else crc0
        print(json.dumps({"workload": args.workload, "gram_items": n, "first_fit_s": first, "fit_with_gram_built_s": second,
                          "gram_build_s": first - second, "same_coefficients_and_sweeps": crc == crc0}), flush=True)
        del eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
