#!/usr/bin/env python3
"""Where the rebuild of the score layouts goes (after a mini-batch the resident W changed and every layout is rebuilt
on the device): per builder, warm, median of 5.   python tools/layout_profile.py --workload c3s"""
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
    args = ap.parse_args()
    import torch
    from rtrec_amd import engine as E
    from rtrec_amd import seg_layout as S
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    X = workload_matrix(wl)
    Xc = X.tocsc()
    Xc.sort_indices()
    I, K = wl["I"], wl["K"]
    eng = E.SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    out = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True)
    dw = eng.merge_fit(None, I, False, *out[:4])
    eng.set_weights(dw)

    def timed(fn, n=5):
        ts = []
        for _ in range(n + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts[1:])), r

    rep = {"workload": args.workload, "W_nnz": dw.nnz}
    rep["tiled_compact_ms"], lay = timed(lambda: E.build_tiled_w_device(torch, dw.rows, dw.cols, dw.vals, I, 0, I, 4096, compact=True,
                                                                        dense_fill=E.DENSE_ROW_FILL))
    rep["feature_rows_ms"], fr = timed(lambda: E.build_feature_rows_device(torch, dw.rows, dw.cols, dw.vals, I, 0, I, tile_cols=256))
    rep["feature_rows_built"] = fr is not None
    rep["cluster_labels_ms"], lab = timed(lambda: S.cluster_labels_device(torch, dw.rows, dw.cols, dw.vals, I))
    rep["segments_with_labels_ms"], sg = timed(lambda: S.build_seg_layout_device(torch, dw.rows, dw.cols, dw.vals, I, 0, I, labels=lab))
    rep["segments_native_ms"], _ = timed(lambda: S.build_seg_layout_native(eng.be, dw.rows, dw.cols, dw.vals, I, 0, I, lab), n=9)
    nb = int(eng.be.lib.rtrec_slim_score_sg_scratch_bytes(I, sg["sg_n_tiles"], sg["sg_T"])) if sg else 0
    rep["heavy_scratch_bytes"] = nb
    rep["heavy_scratch_zero_ms"], _ = timed(lambda: eng.be.zeros((nb,), torch.uint8))

    def whole():
        eng.set_weights(dw)
        return eng._layout(True, 10)
    rep["engine_layout_total_ms"], _ = timed(whole)
    rows = np.arange(0, 100 * 97, 97)

    def after_update():
        eng.set_weights(dw)
        return eng.recommend_rows(rows, top_k=10)
    rep["recommend_100_after_set_weights_ms"], _ = timed(after_update, n=9)
    rep["recommend_100_path"] = eng.last_score_path
    rep["recommend_100_warm_ms"], _ = timed(lambda: eng.recommend_rows(rows, top_k=10), n=9)
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
