#!/usr/bin/env python3
"""Randomised parity run of the exact fit path against the C oracle: random matrix shapes / densities / K / sign
constraint / rating signs, through the throughput kernel, the latency kernel (own X^T y walks) and the latency kernel with
the one-pass X^T y of all targets.   python tools/fuzz_fit.py --iters 60 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(iters: int, seed: int, log=print) -> int:
    from oracle import slim_oracle as so
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import interaction_matrix
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for it in range(iters):
        U = int(rng.choice([150, 600, 2500]))
        I = int(rng.choice([20, 90, 300, 700]))
        draws = int(U * rng.choice([3, 12, 30]))
        K = [None, 5, 20, 50][int(rng.integers(0, 4))] if I <= 300 else [5, 20, 50][int(rng.integers(0, 3))]
        positive = bool(rng.integers(0, 2))
        X = interaction_matrix(U, I, draws, seed=int(rng.integers(1, 10 ** 6)), float_ratings=bool(rng.integers(0, 2)))
        if rng.random() < 0.3:
            X.data = (X.data * np.where(rng.random(X.nnz) < 0.25, -1.0, 1.0)).astype(np.float32)
        Xc = X.tocsc(); Xc.sort_indices()
        mode = ["sw", "mw", "mw-xty"][int(rng.integers(0, 3))]
        os.environ["RTREC_AMD_FIT_MODE"] = mode[:2]
        os.environ["RTREC_AMD_XTY_BATCH"] = "force" if mode == "mw-xty" else "0"
        os.environ["RTREC_AMD_SCREEN_MIN"] = str(rng.choice([1, 64, 100000]))
        eng = SlimEngine(device="cuda:0")
        eng.set_interactions(Xc, X)
        n_t = int(rng.integers(1, I + 1))
        cols = np.sort(rng.choice(I, n_t, replace=False))
        tg, items, coef, count, n_iter = eng.fit_columns(cols, positive=positive, nn_feature_selection=K)
        ptr, idx, val, nit = so.fit_columns(Xc, tg, positive=positive, nn_feature_selection=K)
        ok = np.array_equal(n_iter, nit) and np.array_equal(count, np.diff(ptr))
        if ok:
            for t in range(len(tg)):
                c = count[t]
                gi, gv = items[t, :c], coef[t, :c]
                o = np.argsort(gi, kind="stable")
                if not (np.array_equal(gi[o], idx[ptr[t]:ptr[t + 1]]) and
                        np.array_equal(gv[o].view(np.uint32), val[ptr[t]:ptr[t + 1]].view(np.uint32))):
                    ok = False
                    break
        if not ok:
            bad += 1
            log(f"MISMATCH it={it} U={U} I={I} draws={draws} K={K} positive={positive} mode={mode} targets={n_t}")
        if it % 20 == 19:
            log(f"[fuzz-fit] {it + 1} configurations, {bad} mismatches, {time.time() - t0:.0f}s")
    for k in ("RTREC_AMD_FIT_MODE", "RTREC_AMD_XTY_BATCH", "RTREC_AMD_SCREEN_MIN"):
        os.environ.pop(k, None)
    return bad


def run_sgd(iters: int, seed: int, log=print) -> int:
    """optim="sgd": csrc/fit_sgd.hip against slim_oracle_sgd -- random shapes, K 1..256 (one, two, four features per lane), penalties, learning rates (large ones
    drive the weight scale through its resets), epochs, tol (None: no early stop), seeds, signed ratings."""
    from oracle import slim_oracle as so
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import interaction_matrix
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for it in range(iters):
        U = int(rng.choice([40, 150, 600, 2500]))
        I = int(rng.choice([8, 20, 90, 300]))
        draws = int(U * rng.choice([2, 6, 20]))
        K = int(rng.choice([1, 2, 5, 20, 50, 64, 65, 100, 128, 129, 200, 256]))
        alpha = float(rng.choice([1e-4, 0.01, 0.1, 1.0]))
        l1_ratio = float(rng.choice([0.0, 0.1, 0.5, 1.0]))
        eta0 = float(rng.choice([1e-3, 0.01, 0.05, 0.3]))
        max_iter = int(rng.choice([1, 3, 12, 40]))
        tol = [None, 1e-4, 1e-2][int(rng.integers(0, 3))]
        rs = int(rng.integers(0, 2 ** 31 - 1))
        X = interaction_matrix(U, I, draws, seed=int(rng.integers(1, 10 ** 6)), float_ratings=bool(rng.integers(0, 2)))
        if rng.random() < 0.3:
            X.data = (X.data * np.where(rng.random(X.nnz) < 0.25, -1.0, 1.0)).astype(np.float32)
        Xc = X.tocsc(); Xc.sort_indices()
        eng = SlimEngine(device="cuda:0")
        eng.set_interactions(Xc, X)
        n_t = int(rng.integers(1, min(I, 40) + 1))
        cols = np.sort(rng.choice(I, n_t, replace=False))
        cfg = dict(alpha=alpha, l1_ratio=l1_ratio, eta0=eta0, max_iter=max_iter, random_state=rs, nn_feature_selection=K)
        try:
            tg, items, coef, count, n_iter = eng.fit_columns_sgd(cols, tol=tol, **cfg)
            got_err = None
        except ValueError as e:                       # non-finite weights: the oracle must have diverged as well
            got_err = e
        if got_err is not None:
            ptr, idx, val, nit = so.fit_columns_sgd(Xc, cols, tol=(-np.inf if tol is None else tol), **cfg)
            ok = not np.all(np.isfinite(val))
            why = "error without a non-finite oracle weight"
        else:
            tg, items, coef, count = (np.asarray(a.cpu()) if hasattr(a, "cpu") else np.asarray(a) for a in (tg, items, coef, count))
            ptr, idx, val, nit = so.fit_columns_sgd(Xc, tg, tol=(-np.inf if tol is None else tol), **cfg)     # the engine's target order
            ok, why = True, ""
            if not np.array_equal(np.sort(tg), cols):
                ok, why = False, f"targets {tg.tolist()} != {cols.tolist()}"
            elif not np.array_equal(np.asarray(n_iter), nit):
                ok, why = False, f"n_iter {np.asarray(n_iter).tolist()} != {nit.tolist()}"
            for t in range(len(tg)):
                if not ok:
                    break
                c = int(count[t])
                gi, gv = items[t, :c], coef[t, :c]
                o = np.argsort(gi, kind="stable")
                ri, rv = idx[ptr[t]:ptr[t + 1]], val[ptr[t]:ptr[t + 1]]
                if c != len(ri):                       # fewer features than K on the device: the missing ones must be zero in the oracle
                    keep = np.isin(ri, gi)
                    if np.any(rv[~keep] != 0):
                        ok, why = False, f"target {tg[t]}: count {c} != {len(ri)} and a dropped feature has a weight"
                        break
                    ri, rv = ri[keep], rv[keep]
                if not (np.array_equal(gi[o], ri) and np.array_equal(gv[o].view(np.uint32), rv.view(np.uint32))):
                    ok, why = False, f"target {tg[t]}: items {gi[o].tolist()[:8]} vs {ri.tolist()[:8]}, coef {gv[o].tolist()[:6]} vs {rv.tolist()[:6]}"
        if not ok:
            bad += 1
            log(f"MISMATCH it={it} U={U} I={I} draws={draws} tol={tol} targets={n_t} cfg={cfg} err={got_err} why={why}")
        if it % 20 == 19:
            log(f"[fuzz-sgd] {it + 1} configurations, {bad} mismatches, {time.time() - t0:.0f}s")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sgd", action="store_true", help="fuzz optim='sgd' (csrc/fit_sgd.hip) instead of coordinate descent")
    args = ap.parse_args()
    bad = (run_sgd if args.sgd else run)(args.iters, args.seed, log=lambda m: print(m, flush=True))
    print(f"fuzz-{'sgd' if args.sgd else 'fit'} done: {args.iters} configurations, mismatches: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
