#!/usr/bin/env python3
"""How many tiles a user must open under the per-row bound (sum_i |x_i| max|w| of row i in the tile: what the kernel uses) and
under a column-norm bound (max_i |x_i| times the tile's largest column L1 norm), by user length.   python tools/bound_probe.py"""
import argparse, json, os, sys
import numpy as np
import scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bench import WORKLOADS
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3s")
    args = ap.parse_args()
    import torch
    from rtrec_amd import engine as E
    from rtrec_amd.seg_layout import build_seg_layout, cluster_labels
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS[args.workload]
    X = workload_matrix(wl)
    Xc = X.tocsc(); Xc.sort_indices()
    U, I, K = wl["U"], wl["I"], wl["K"]
    eng = E.SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    out = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True, mode="gram")
    dw = eng.merge_fit(None, I, False, *out[:4])
    W = dw.to_csc(torch).astype(np.float32)
    coo = W.tocoo()
    o = np.lexsort((coo.row, coo.col))
    labels = cluster_labels(coo.row[o].astype(np.int64), coo.col[o].astype(np.int64), coo.data[o], I)
    lay = build_seg_layout(W, 0, I, labels=labels)
    T, n_tiles = lay["sg_T"], lay["sg_n_tiles"]
    pos = lay["sg_info"][:, 1]                         # item -> layout column
    Wr = W.tocsr()
    absW = abs(W)
    col_l1 = np.asarray(absW.sum(axis=0)).ravel()
    tile_of_col = np.where(pos >= 0, pos // T, -1)
    l1max = np.zeros(n_tiles)
    np.maximum.at(l1max, tile_of_col[tile_of_col >= 0], col_l1[tile_of_col >= 0])
    # per (row of W, tile) max |w|
    Wc = absW.tocoo()
    rt_max = sp.coo_matrix((Wc.data, (Wc.row, tile_of_col[Wc.col])), shape=(I, n_tiles)).tocsr()
    rt_max.sum_duplicates()
    rowtile = np.zeros((I, n_tiles), dtype=np.float32)
    np.maximum.at(rowtile, (Wc.row, tile_of_col[Wc.col]), Wc.data)
    # the same bound at a quarter and a sixteenth of a tile: a tile's bound = the largest of its sub-blocks' bounds
    sub = {}
    for div in (4, 16):
        Ts = T // div
        sub_of_col = np.where(pos >= 0, pos // Ts, -1)
        m = np.zeros((I, n_tiles * div), dtype=np.float32)
        np.maximum.at(m, (Wc.row, sub_of_col[Wc.col]), Wc.data)
        sub[div] = m
    lens = np.diff(X.indptr)
    rng = np.random.default_rng(3)
    rep = {}
    for name, lo, hi in (("<=128", 1, 128), ("128-256", 129, 256), ("256-512", 257, 512), (">512", 513, 10 ** 9)):
        users = np.flatnonzero((lens >= lo) & (lens <= hi))
        users = rng.choice(users, min(200, len(users)), replace=False)
        opened_sum, opened_l1, opened_need = [], [], []
        extra = {}
        for u in users:
            xs, xv = X.indices[X.indptr[u]:X.indptr[u + 1]], X.data[X.indptr[u]:X.indptr[u + 1]]
            s = np.asarray((sp.csr_matrix((xv, (np.zeros(len(xs), int), xs)), shape=(1, I)) @ Wr).todense()).ravel()
            s[xs] = -np.inf
            nz = np.flatnonzero((s != 0) & np.isfinite(s))
            if len(nz) <= 11:
                continue
            theta = np.sort(s[nz])[-11]
            B_sum = np.abs(xv) @ rowtile[xs]
            B_l1 = np.abs(xv).max() * l1max
            tmax = np.full(n_tiles, -np.inf)
            np.maximum.at(tmax, tile_of_col[nz], s[nz])
            opened_need.append(int((tmax >= theta).sum()))
            opened_sum.append(int((B_sum >= theta).sum()))
            opened_l1.append(int((np.minimum(B_sum, B_l1) >= theta).sum()))
            for div in (4, 16):
                Bs = (np.abs(xv) @ sub[div][xs]).reshape(n_tiles, div).max(axis=1)
                extra.setdefault(div, []).append(int((Bs >= theta).sum()))
        rep[name] = {"users": len(opened_sum), "tiles_holding_a_top11_score": float(np.mean(opened_need)),
                     "tiles_with_row_bound_above_theta": float(np.mean(opened_sum)),
                     "with_min_of_row_and_column_norm_bound": float(np.mean(opened_l1)),
                     "with_quarter_tile_bounds": float(np.mean(extra[4])), "with_sixteenth_tile_bounds": float(np.mean(extra[16]))}
    print(json.dumps({"workload": args.workload, "n_tiles": int(n_tiles), "by_length": rep}))


if __name__ == "__main__":
    main()
