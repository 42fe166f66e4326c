#!/usr/bin/env python3
"""Randomised parity run of the score path (feature-row kernel: resident and streaming layouts, fragments, tile test,
exact-tie pass; segment kernels: wave per user and workgroup per user; SPARSE and DENSE mode; request-sized and full
batches) against the C oracle: random numbers of rows of W, columns, densities, users, top_k, filter, integer / float /
negative values, 256- and 128-column tiles.   python tools/fuzz_score.py --iters 60 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(iters: int, seed: int, log=print, shards: bool = True) -> int:
    """Number of (configuration, call) pairs whose ids, score bits or counts differ from the oracle's."""
    from oracle import slim_oracle as so
    from rtrec_amd import _native
    from rtrec_amd.engine import SlimEngine
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    for it in range(iters):
        R = int(rng.choice([1, 3, 17, 40, 64, 65, 66, 90, 128, 129, 300, 1500]))
        n_cols = int(rng.choice([8, 100, 700, 3000, 9000, 20000]))
        I = max(n_cols + int(rng.integers(50, 3000)), R + 10, 2 * R)
        U = int(rng.choice([97, 700, 3000, 9000]))
        per_col = int(rng.integers(1, max(2, min(R, 40)) + 1))
        integer = bool(rng.integers(0, 2))
        feat = np.sort(rng.choice(I, R, replace=False))
        cols = np.sort(rng.choice(I, n_cols, replace=False))
        pop = 1.0 / np.arange(1, R + 1) ** float(rng.uniform(0.3, 1.5))
        rr, cc = [], []
        for c in cols:
            k = min(R, max(1, int(rng.integers(1, per_col + 1))))
            rr.append(rng.choice(feat, k, replace=False, p=pop / pop.sum()))
            cc.append(np.full(k, c))
        rr, cc = np.concatenate(rr), np.concatenate(cc)
        vals = (rng.random(len(rr)).astype(np.float32) + 0.05) * np.where(rng.random(len(rr)) < rng.choice([0.0, 0.1, 0.5]), -1, 1).astype(np.float32)
        if integer:
            vals = np.round(vals * 4).astype(np.float32)
            vals[vals == 0] = 1.0
        W = sp.csc_matrix((vals, (rr, cc)), shape=(I, I), dtype=np.float32)
        W.sum_duplicates(); W.eliminate_zeros(); W.sort_indices()
        n_it = rng.integers(0, int(rng.choice([5, 60, 400, 400, 3000])), U)      # (3000: rows of 640+ entries take the long setup rounds)
        ur = np.repeat(np.arange(U), n_it)
        ui = np.where(rng.random(len(ur)) < rng.uniform(0.1, 0.9), rng.choice(feat, len(ur)), rng.integers(0, I, len(ur)))
        xv = rng.integers(1, 6, len(ur)).astype(np.float32) if integer else (rng.random(len(ur)).astype(np.float32) * 5 - rng.choice([0.0, 0.5]))
        X = sp.csr_matrix((xv, (ur, ui)), shape=(U, I), dtype=np.float32)
        X.sum_duplicates(); X.eliminate_zeros(); X.sort_indices()
        eng = SlimEngine(device="cuda:0")
        eng.FR_TILE_COLS = int(rng.choice([256, 256, 128]))
        eng.FR_MIN_ROWS = int(rng.choice([0, 32]))
        eng.set_interactions(None, X, need_csc=False)
        f64 = bool(rng.integers(0, 4) == 0)              # a float64 W (float32-valued): float64 accumulation
        eng.set_weights(W.astype(np.float64) if f64 else W, acc_f64=f64)
        lay = eng._layout(True)
        Wr = W.tocsr()
        for _ in range(3):
            top_k = int(rng.integers(1, 16)) if rng.integers(0, 3) else int(rng.integers(16, 40))
            filt = bool(rng.integers(0, 2))
            dense = bool(rng.integers(0, 3) == 0)
            pick = int(rng.integers(0, 3))
            rows = (np.arange(U) if pick == 0 else rng.permutation(U)[:int(rng.integers(1, U + 1))] if pick == 1
                    else rng.permutation(U)[:int(rng.integers(1, 60))])
            eng.sg_heavy_min = int(rng.choice([0, 0, 1, 9, 300]))
            ids, sc, cnt = eng.recommend_rows(rows, top_k=top_k, filter_interacted=filt,
                                              mode=_native.TOPK_DENSE if dense else _native.TOPK_SPARSE)
            o_ids, o_sc, o_cnt = so.recommend_batch(X[rows], Wr, top_k=top_k, filter_interacted=filt, dense=dense, use_f64=f64)
            ok = np.array_equal(cnt, o_cnt) and np.array_equal(ids, o_ids) and np.array_equal(sc.view(np.uint32), o_sc.view(np.uint32))
            if not ok:
                bad += 1
                wrong = np.flatnonzero((ids != o_ids).any(axis=1) | (cnt != o_cnt))
                log(f"MISMATCH it={it} R={R} n_cols={n_cols} U={U} top_k={top_k} filt={filt} dense={dense} f64={f64} n_rows={len(rows)} path={eng.last_score_path} "
                    f"heavy_min={eng.sg_heavy_min} integer={integer} tc={eng.FR_TILE_COLS} "
                    f"fr={lay.get('fr_w') is not None} rows_wrong={len(wrong)} first={wrong[:5].tolist()}")
        if shards and it % 3 == 0:
            # the same W as 2 .. 4 COLUMN SHARDS on this GPU (the engines of N ranks, scored one after another), their lists merged
            # like the exchange merges them: SPARSE and DENSE mode against the oracle on the full W
            import torch
            N = int(rng.integers(2, 5))
            top_k = int(rng.integers(1, 16))
            filt = bool(rng.integers(0, 2))
            dense = bool(rng.integers(0, 2))
            mode = _native.TOPK_DENSE if dense else _native.TOPK_SPARSE
            rows = np.sort(rng.permutation(U)[:int(rng.integers(1, U + 1))]).astype(np.int32)
            parts, paths = [], []
            for r in range(N):
                e = SlimEngine(device="cuda:0", rank=r, world_size=N)
                e.world_size_for_merge = N
                e.FR_TILE_COLS, e.FR_MIN_ROWS = eng.FR_TILE_COLS, eng.FR_MIN_ROWS
                e.set_interactions(None, X, need_csc=False)
                e.set_weights(W.astype(np.float64) if f64 else W, acc_f64=f64)
                xb = (e._X["rptr"], e._X["rcol"], e._X["rval"])
                parts.append(e._local_topk(e.be.to_dev(rows), len(rows), xb, top_k, filt, mode, None))
                paths.append(e.last_score_path)
            be = eng.be
            g = [torch.stack([p[j] for p in parts]).contiguous() for j in (0, 1, 3, 4)]
            g64 = torch.stack([p[2] for p in parts]).contiguous() if f64 else None
            m_ids, m_sc = be.empty((len(rows), top_k), torch.int32), be.empty((len(rows), top_k), torch.float32)
            m_cnt = be.empty((len(rows),), torch.int32)
            _native.check(be.lib.rtrec_slim_merge_topk(len(rows), N, top_k, be.ptr(g[0]), be.ptr(g[1]), be.ptr(g64), be.ptr(g[2]), be.ptr(g[3]),
                                                       be.ptr(m_ids), be.ptr(m_sc), be.ptr(m_cnt), be.stream()), "merge")
            o_ids, o_sc, o_cnt = so.recommend_batch(X[rows], Wr, top_k=top_k, filter_interacted=filt, dense=dense, use_f64=f64)
            ids, sc, cnt = m_ids.cpu().numpy(), m_sc.cpu().numpy(), m_cnt.cpu().numpy()
            if not (np.array_equal(cnt, o_cnt) and np.array_equal(ids, o_ids) and np.array_equal(sc.view(np.uint32), o_sc.view(np.uint32))):
                bad += 1
                wrong = np.flatnonzero((ids != o_ids).any(axis=1) | (cnt != o_cnt))
                log(f"MISMATCH (column shards) it={it} f64={f64} N={N} R={R} n_cols={n_cols} U={U} top_k={top_k} filt={filt} dense={dense} "
                    f"n_rows={len(rows)} paths={paths} integer={integer} rows_wrong={len(wrong)} first={wrong[:5].tolist()}")
        if it % 50 == 49:
            log(f"[fuzz] {it + 1} configurations, {bad} mismatches, {time.time() - t0:.0f}s")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-shards", action="store_true", help="skip the column-shard draws")
    args = ap.parse_args()
    bad = run(args.iters, args.seed, log=lambda m: print(m, flush=True), shards=not args.no_shards)
    print(f"fuzz done: {args.iters} configurations x 3 calls, mismatches: {bad}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
