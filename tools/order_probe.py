#!/usr/bin/env python3
"""Does a cluster-aware work order help the segment kernel?  Times the all-users pass with the default (length) order and with
rows ordered by (length class, dominant cluster label of the user's items, length).   python tools/order_probe.py"""
import os
import sys
import time

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
    X = workload_matrix(wl)
    Xc = X.tocsc(); Xc.sort_indices()
    U, I, K = wl["U"], wl["I"], wl["K"]
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    out = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True)
    eng.set_weights(eng.merge_fit(None, I, False, *out[:4]))
    d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
    xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])

    def run(label):
        for _ in range(2):
            eng.score_topk_device(None, U, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            o = eng.score_topk_device(None, U, 10, True, _native.TOPK_SPARSE, d_rows=d_rows, xb=xb)
        torch.cuda.synchronize()
        print(f"[order] {label}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms/pass", flush=True)
        return o[0].cpu().numpy()

    a = run("length order")
    fast = eng._fast_layout()
    lab = fast["sg"]["sg_labels"].cpu().numpy()
    lens = np.diff(X.indptr)
    # dominant label per user (weighted by nothing: item count)
    li = lab[X.indices]
    ui = np.repeat(np.arange(U), lens)
    key = ui.astype(np.int64) * I + li
    k, c = np.unique(key, return_counts=True)
    best = np.zeros(U, np.int64)
    np.maximum.at(best, k // I, c.astype(np.int64) * I + k % I)
    home = best % I
    cls = np.where(lens > 512, 0, np.where(lens > 256, 1, np.where(lens > 128, 2, 3)))
    order = np.lexsort((-lens, home, cls)).astype(np.int32)
    d_order = eng.be.to_dev(order)
    orig = eng._row_order
    eng._row_order = lambda *args, **kw: d_order
    b = run("cluster order (class, home label, length)")
    order2 = np.lexsort((-lens, home)).astype(np.int32)
    d_order2 = eng.be.to_dev(order2)
    eng._row_order = lambda *args, **kw: d_order2
    c2 = run("cluster order (home label, length)")
    print("[order] same ids:", np.array_equal(a, b) and np.array_equal(a, c2))


if __name__ == "__main__":
    main()
