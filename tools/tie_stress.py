#!/usr/bin/env python3
"""Exact-tie stress of the scoring path at sizes that select the 4- and 8-users-per-wave forms of the feature-row kernel:
integer ratings and a W whose columns are copies of 50 base columns (ties inside every list and at its threshold), 30k / 60k /
120k users, top-10 and top-3, filter on and off -- ids, score bits and counts against the C oracle.   python tools/tie_stress.py
"""
import sys, os, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtrec_amd import _native
from rtrec_amd.engine import SlimEngine
from rtrec_amd.synth import interaction_matrix
from oracle import slim_oracle as oracle
bits = lambda a: a.view(np.uint32)
bad = 0
for U, draws, seed in ((30000, 400000, 1), (120000, 1500000, 2), (60000, 700000, 3)):
    I = 700
    X = interaction_matrix(U, I, draws, seed=seed)                 # integer ratings: ties everywhere
    rng = np.random.default_rng(seed)
    base = sp.random(I, 50, density=0.3, random_state=seed, format="csc", dtype=np.float32)
    keep = np.isin(base.indices, np.arange(0, I, 13)); base.data[~keep] = 0; base.eliminate_zeros()
    W = sp.csc_matrix(base[:, rng.integers(0, 50, size=I)]); W.sort_indices()
    eng = SlimEngine(device="cuda:0", tile_cols=256)
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    rows = np.arange(U)
    for filt in (True, False):
        for k in (10, 3):
            t0 = time.time()
            ids, sc, cnt = eng.recommend_rows(rows, top_k=k, filter_interacted=filt, mode=_native.TOPK_SPARSE)
            o_ids, o_sc, o_cnt = oracle.recommend_batch(X, W.tocsr(), top_k=k, filter_interacted=filt, n_threads=8)
            ok = np.array_equal(cnt, o_cnt) and np.array_equal(ids, o_ids) and np.array_equal(bits(sc), bits(o_sc))
            bad += 0 if ok else 1
            print(U, filt, k, "ok" if ok else "MISMATCH", "rows differing:", int((ids != o_ids).any(axis=1).sum()), flush=True)
print("tie stress done, mismatching calls:", bad)
