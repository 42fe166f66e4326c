#!/usr/bin/env python3
"""Feature-row kernel: 256- against 128-column tiles on a workload's full pass.   python tools/fr_tile_ab.py --workload c3"""
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
    dw = eng.merge_fit(None, I, False, *out[:4])
    d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
    rep = {}
    for tc in (256, 128):
        eng.FR_TILE_COLS = tc
        eng.set_weights(dw)
        step = lambda: eng.score_topk_device(None, U, 10, True, _native.TOPK_SPARSE, d_rows=d_rows)
        for _ in range(4):
            o = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            o = step()
        torch.cuda.synchronize()
        fast = eng._W.get("fast") or {}
        rep[str(tc)] = {"ms": round((time.perf_counter() - t0) / 10 * 1e3, 4), "path": eng.last_score_path,
                        "resident": bool((fast.get("fr_host") or {}).get("fr_resident")), "n_tiles": fast.get("fr_n_tiles"),
                        "crc": int(np.bitwise_xor.reduce(o[0].cpu().numpy().astype(np.int64).ravel() * 2654435761 % (1 << 31)))}
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
