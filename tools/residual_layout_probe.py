#!/usr/bin/env python3
"""How many 64-byte sectors of the per-target residual R the exact fit's column visits touch, under different storage
orders of R (a permutation of the user ADDRESSES only: iteration order and every sum stay as they are).

For a sample of target columns (stratified by length) the oracle fits the column (features, coefficients, sweeps); the visits
of the coordinate descent are modelled as sweeps x the columns of the features that end non-zero (ordered fold + residual
update).  Per visited column the distinct 16-row sectors under: identity | users by activity | (home cluster, activity) with
the generator's own home cluster (an upper bound for any clustering) | (activity tier, cluster).
    python tools/residual_layout_probe.py --workload c3s --targets 200
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sectors(rows, inv, sec=16):
    return np.unique(inv[rows] // sec).size


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3s")
    ap.add_argument("--targets", type=int, default=200)
    args = ap.parse_args()
    from bench import WORKLOADS
    from rtrec_amd.synth import workload_matrix
    from oracle import slim_oracle as so
    wl = WORKLOADS[args.workload]
    t0 = time.time()
    X = workload_matrix(wl, seed=20251003, float_ratings=True)
    Xc = X.tocsc(); Xc.sort_indices()
    U, I = X.shape
    print(f"X {U}x{I} nnz {X.nnz} in {time.time() - t0:.1f}s", flush=True)
    lens = np.diff(Xc.indptr)
    act = np.diff(X.indptr)
    rng = np.random.default_rng(3)
    order = np.argsort(-lens)
    # stratified: the fit's time is in the long targets
    tg = np.unique(np.concatenate([order[:args.targets // 4], rng.choice(order[:4000], args.targets // 2, replace=False),
                                   rng.choice(order[4000:], args.targets // 4, replace=False)]))
    t0 = time.time()
    ptr, idx, val, nit = so.fit_columns(Xc, tg, nn_feature_selection=wl["K"], n_threads=8)
    print(f"oracle fit of {len(tg)} targets {time.time() - t0:.1f}s; mean sweeps {nit.mean():.1f}", flush=True)
    perms = {"identity": np.arange(U)}
    perms["activity"] = np.argsort(-act, kind="stable")
    if wl.get("gen") == "clustered":
        # the generator's own home cluster (rtrec_amd/synth.py: clustered_pairs): an upper bound for any user clustering
        g = np.random.default_rng(20251003)
        C_ = wl["clusters"]
        perm_u, perm_i = g.permutation(U), g.permutation(I)
        g.random(wl["draws"])
        home_rank = g.integers(0, C_, size=U)
        home = np.empty(U, np.int64); home[perm_u] = home_rank
        perms["home,activity"] = np.lexsort((-act, home))
        tier = np.searchsorted(np.quantile(act, [0.5, 0.8, 0.95]), act)       # 0 = light ... 3 = heaviest 5 %
        perms["tier,home"] = np.lexsort((home, -tier))
        perms["home,tier-desc"] = np.lexsort((-tier, home))
    inv = {}
    for k, p in perms.items():
        a = np.empty(U, np.int64); a[p] = np.arange(U); inv[k] = a
    tot = {k: 0.0 for k in perms}
    ent = 0.0
    per_len = []
    for t, j in enumerate(tg):
        f, w = idx[ptr[t]:ptr[t + 1]], val[ptr[t]:ptr[t + 1]]
        nz = f[w != 0]
        if nz.size == 0:
            continue
        sweeps = int(nit[t])
        for c in nz:
            rows = Xc.indices[Xc.indptr[c]:Xc.indptr[c + 1]]
            ent += sweeps * rows.size
            for k in perms:
                tot[k] += sweeps * sectors(rows, inv[k])
    print(f"modelled visits: {ent:.3e} column entries (sweeps x non-zero features' columns)")
    for k in perms:
        print(f"  {k:18s} sectors/entry {tot[k] / ent:.3f}  = {64 * tot[k] / ent:5.1f} B of residual per entry   ({tot['identity'] / tot[k]:.2f}x fewer than identity)")


if __name__ == "__main__":
    main()
