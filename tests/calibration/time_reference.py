#!/usr/bin/env python3
"""Reference-of-record timings in the build container.

Times the REAL rtrec + scikit-learn path (imported from /root/reference) and the C oracle on the
same ML-1M-shaped synthetic input, so the oracle's CPU numbers reported by bench.py on the GPU host
can be related to the true reference.  Runs only here; prints a markdown table.
"""
import os
import sys
import time

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
from ref_import import import_reference  # noqa: E402

import_reference()
from rtrec.models import SLIM as RefSLIM  # noqa: E402
from rtrec.recommender import Recommender as RefRecommender  # noqa: E402

from oracle import slim_oracle as so  # noqa: E402
from rtrec_amd.synth import interaction_matrix  # noqa: E402


def main():
    U, I, draws, K = 6040, 3706, 1_000_000, 50
    X = interaction_matrix(U, I, draws, seed=20251003, float_ratings=True)
    coo = X.tocoo()
    rng = np.random.default_rng(0)
    order = rng.permutation(coo.nnz)
    df = pd.DataFrame({"user": coo.row[order].astype(int), "item": coo.col[order].astype(int),
                       "tstamp": 1.7e9 + np.arange(coo.nnz, dtype=float), "rating": coo.data[order].astype(float)})
    print(f"ML-1M-shaped synthetic: {U} x {I}, {coo.nnz} interactions, float ratings, K={K}, {os.cpu_count()} cores")

    rec = RefRecommender(RefSLIM(min_value=0, max_value=15, nn_feature_selection=K))
    t = time.time(); rec.bulk_fit(df, parallel=True); t_ref_fit = time.time() - t
    model = rec.get_model()
    users = list(range(0, U, 3))
    t = time.time(); ref_recs = model.recommend_batch(users, top_k=10); t_ref_rec = time.time() - t

    Xc = X.tocsc(); Xc.sort_indices()
    t = time.time(); ptr, idx, val, nit = so.fit_columns(Xc, np.arange(I), nn_feature_selection=K); t_or_fit = time.time() - t
    W = model.model.item_similarity.tocsr()
    t = time.time(); ids, sc, cnt = so.recommend_batch(X[users], W, top_k=10); t_or_rec = time.time() - t
    same = all(ids[r, :cnt[r]].tolist() == ref_recs[r] for r in range(len(users)))

    print("| measurement | real reference (rtrec + sklearn) | C oracle, 1 thread | oracle / reference |")
    print("|---|---|---|---|")
    print(f"| bulk_fit K={K} (reference: ingest + 5-process pool; oracle: fit only) | {t_ref_fit:.2f} s = {coo.nnz / t_ref_fit:,.0f} interactions/s | "
          f"{t_or_fit:.2f} s = {coo.nnz / t_or_fit:,.0f} interactions/s | {t_ref_fit / t_or_fit:.2f}x faster |")
    print(f"| recommend_batch top-10, {len(users)} users | {t_ref_rec:.2f} s = {len(users) / t_ref_rec:,.0f} users/s | "
          f"{t_or_rec:.3f} s = {len(users) / t_or_rec:,.0f} users/s | {t_ref_rec / t_or_rec:.1f}x faster |")
    print(f"top-k ids identical between reference and oracle: {same}")


if __name__ == "__main__":
    main()
