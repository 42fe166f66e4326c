#!/usr/bin/env python3
"""Feature-row kernel, users per wave (8 / 4 / 2) against pass size: the row-shard sizes of 2 / 4 / 8 ranks on the C3 shape.
    python tools/fr_uw_sweep.py --workload c3"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    args = ap.parse_args()
    import torch
    from rtrec_amd import _native
    from rtrec_amd import engine as E
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    X = workload_matrix(wl)
    Xc = X.tocsc(); Xc.sort_indices()
    U, I, K = wl["U"], wl["I"], wl["K"]
    eng = E.SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    out = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True, mode="gram")
    eng.set_weights(eng.merge_fit(None, I, False, *out[:4]))
    rep = {}
    for n in (U // 16, U // 8, U // 4, U // 2, U):
        d_rows = eng.be.to_dev(np.arange(n, dtype=np.int32) + (U - n) // 2)
        row = {}
        for uw in (0, 2, 4, 8):
            eng.fr_users_per_wave = uw
            step = lambda: eng.score_topk_device(None, n, 10, True, _native.TOPK_SPARSE, d_rows=d_rows)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            row["auto" if uw == 0 else str(uw)] = round((time.perf_counter() - t0) / 8 * 1e3, 4)
        row["path"] = eng.last_score_path
        rep[str(n)] = row
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
