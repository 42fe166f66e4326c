#!/usr/bin/env python3
"""Randomised end-to-end parity of the drop-in API: the same random interaction stream (bulk fit, mini-batches with repeats
and upserts, optional time decay, int or str ids) goes through rtrec_amd.SLIM on the GPU engine and through the same class on
the CPU oracle backend (tests/cpu_backend.py, itself pinned to the reference's goldens); W (bits), recommend_batch and
similar_items must agree after every step.   python tools/fuzz_api.py --iters 50 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def same_w(A, B):
    A, B = A.tocsc(), B.tocsc()
    A.sort_indices(); B.sort_indices()
    return (A.shape == B.shape and A.dtype == B.dtype and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
            and np.array_equal(A.data.view(np.uint32 if A.dtype == np.float32 else np.uint64),
                               B.data.view(np.uint32 if B.dtype == np.float32 else np.uint64)))


def outcome(call):
    """The call's result, or the type of the exception it raised."""
    try:
        return call()
    except Exception as e:       # noqa: BLE001 -- the comparison is the point
        return ("raised", type(e).__name__)


def run(iters: int, seed: int, log=print) -> int:
    from rtrec_amd import SLIM
    from rtrec_amd.engine import SlimEngine
    from tests.cpu_backend import OracleBackend
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for it in range(iters):
        U, I = int(rng.choice([40, 150, 400])), int(rng.choice([25, 80, 250, 1200]))
        kw = dict(min_value=0, max_value=15)
        K = [None, 5, 20][int(rng.integers(0, 3))]
        if K is not None:
            kw["nn_feature_selection"] = K
        if rng.random() < 0.4:
            kw["decay_in_days"] = int(rng.choice([7, 30, 365]))
        if rng.random() < 0.3:
            kw["min_value"], kw["max_value"] = -5, 10
        if K is not None and rng.random() < 0.15:        # scikit-learn's SGDRegressor behind the feature selection (csrc/fit_sgd.hip)
            kw["optim"] = "sgd"
            kw["max_iter"] = int(rng.choice([3, 12, 30]))
            if rng.random() < 0.5:
                kw["eta0"] = float(rng.choice([1e-4, 1e-3, 1e-2]))
        strings = rng.random() < 0.3
        uid = (lambda x: f"u{int(x)}") if strings else int
        iid = (lambda x: f"i{int(x)}") if strings else int
        g, c = SLIM(**kw), SLIM(**kw)
        c.model._engine = SlimEngine(backend=OracleBackend())
        ts = 1.7e9
        ok = True
        for step in range(int(rng.integers(2, 6))):
            n = int(rng.choice([30, 300, 2000])) if step == 0 else int(rng.choice([5, 60, 300]))
            us = rng.zipf(1.6, n) % U
            its = rng.zipf(1.4, n) % I
            rs = (rng.integers(1, 6, n) * rng.choice([1.0, 0.5, -1.0], n, p=[0.7, 0.2, 0.1])).astype(float)
            tss = ts + np.cumsum(rng.integers(0, 86400 * 3, n)).astype(float)
            ts = float(tss[-1])
            batch = [(uid(a), iid(b), float(t), float(r)) for a, b, t, r in zip(us, its, tss, rs)]
            upsert = bool(rng.integers(0, 2))
            fg = outcome(lambda: g.fit(batch, update_interaction=upsert, progress_bar=False))
            fc = outcome(lambda: c.fit(batch, update_interaction=upsert, progress_bar=False))
            if isinstance(fg, tuple) or isinstance(fc, tuple):      # a fit that raises (a diverging SGD fit: ValueError) must raise on both sides
                if fg != fc:
                    ok = False
                    log(f"MISMATCH it={it} step={step} kw={kw}: fit outcomes {fg} vs {fc}")
                break
            users = [uid(x) for x in rng.integers(0, U + 5, 25)]
            k = int(rng.integers(1, 13)) if rng.random() < 0.85 else int(rng.choice([40, 64, 90]))
            filt = bool(rng.integers(0, 2))
            cands = None
            if rng.random() < 0.25:          # a candidate list (known and unknown items, random order)
                cands = [iid(x) for x in rng.permutation(I + 3)[:int(rng.integers(1, min(I, 40) + 1))]]
            if rng.random() < 0.3:           # a large batch: the feature-row kernel where W has few non-empty rows
                users = [uid(x) for x in rng.integers(0, U + 5, 300)]
            # (an item that only a similar_items query has registered -- the reference's identify() there -- has an id beyond
            # W: a candidate list holding it raises IndexError, in the reference inside scipy; both sides must raise alike)
            rg = outcome(lambda: g.recommend_batch(users, candidate_items=cands, top_k=k, filter_interacted=filt))
            rc = outcome(lambda: c.recommend_batch(users, candidate_items=cands, top_k=k, filter_interacted=filt))
            if not isinstance(rg, tuple) and rng.random() < 0.4:      # the array form of the same call: ids [n, k] (-1 padded) + counts
                ag = outcome(lambda: g.recommend_batch(users, candidate_items=cands, top_k=k, filter_interacted=filt, as_arrays=True))
                if isinstance(ag, tuple) and len(ag) == 2 and isinstance(ag[0], str):
                    rg = ag
                else:
                    a_ids, a_cnt = ag
                    back = [a_ids[n, :a_cnt[n]].tolist() for n in range(len(users))]
                    if back != [list(x) for x in rg] or not all(x is None or x == -1 for n in range(len(users)) for x in a_ids[n, a_cnt[n]:].tolist()):
                        rg = ("arrays differ from lists",)
            one = users[0]
            og = outcome(lambda: g.recommend(one, candidate_items=cands, top_k=k, filter_interacted=filt))
            oc = outcome(lambda: c.recommend(one, candidate_items=cands, top_k=k, filter_interacted=filt))
            q = iid(int(rng.integers(0, I)))
            sg, sc_ = g.similar_items(q, top_k=5), c.similar_items(q, top_k=5)
            if rg != rc or og != oc or sg != sc_ or not same_w(g.model.item_similarity, c.model.item_similarity):
                ok = False
                log(f"MISMATCH it={it} step={step} kw={kw} strings={strings} upsert={upsert} cands={cands is not None} k={k} "
                    f"rec_equal={rg == rc} one_equal={og == oc} sim_equal={sg == sc_}")
                break
        bad += 0 if ok else 1
        if it % 10 == 9:
            log(f"[fuzz-api] {it + 1} scenarios, {bad} mismatches, {time.time() - t0:.0f}s")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    bad = run(args.iters, args.seed, log=lambda m: print(m, flush=True))
    print(f"fuzz-api done: {args.iters} scenarios, mismatches: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
