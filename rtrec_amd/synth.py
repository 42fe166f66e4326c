"""Synthetic MovieLens-shaped interaction generator.

users ~ Zipf(s=0.6), items ~ Zipf(s=0.85) over a random permutation, duplicate (u, i) pairs
dropped; ratings either integers 1..5 or decayed floats.  Used by tests/ and bench.py.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import scipy.sparse as sp


def zipf_pairs(n_users: int, n_items: int, n_draws: int, seed: int, user_s: float = 0.6, item_s: float = 0.85
               ) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(seed)
    pu = 1.0 / np.arange(1, n_users + 1, dtype=np.float64) ** user_s
    pi = 1.0 / np.arange(1, n_items + 1, dtype=np.float64) ** item_s
    cu, ci = np.cumsum(pu / pu.sum()), np.cumsum(pi / pi.sum())
    perm_u, perm_i = rng.permutation(n_users), rng.permutation(n_items)
    u = perm_u[np.minimum(np.searchsorted(cu, rng.random(n_draws)), n_users - 1)]
    i = perm_i[np.minimum(np.searchsorted(ci, rng.random(n_draws)), n_items - 1)]
    key = np.unique(u.astype(np.int64) * n_items + i)
    return (key // n_items).astype(np.int32), (key % n_items).astype(np.int32)


def interaction_matrix(n_users: int, n_items: int, n_draws: int, seed: int, float_ratings: bool = True
                       ) -> sp.csr_matrix:
    """U x I CSR float32.  float_ratings=True gives tie-free 'decayed' values in (0.5, 5]."""
    u, i = zipf_pairs(n_users, n_items, n_draws, seed)
    rng = np.random.default_rng(seed + 1)
    if float_ratings:
        v = (rng.integers(1, 6, size=len(u)) * np.exp(-rng.random(len(u)) * 0.7)).astype(np.float32)
    else:
        v = rng.integers(1, 6, size=len(u)).astype(np.float32)
    X = sp.csr_matrix((v, (u, i)), shape=(n_users, n_items), dtype=np.float32)
    X.sort_indices()
    return X


def clustered_pairs(n_users: int, n_items: int, n_draws: int, seed: int, n_clusters: int = 80, p_in: float = 0.85,
                    user_s: float = 0.6, item_s: float = 0.85) -> Tuple[np.ndarray, np.ndarray]:
    """Structured variant of zipf_pairs: the same Zipf marginals, but the items form
    `n_clusters` clusters and every user has a home cluster; a draw takes its item from the user's home cluster with
    probability p_in (by the cluster's share of the global Zipf law) and from the whole catalogue otherwise.  The item
    of global popularity rank r belongs to cluster r % n_clusters, so every cluster is itself Zipf-shaped and holds
    1 / n_clusters of the popularity mass: the item marginal stays the global law.  With item-item structure a SLIM fit
    selects cluster-mates as features, W gets thousands of non-empty rows (nnz close to K x I) instead of the few
    dozen "popular item" rows the independent draws of zipf_pairs yield."""
    rng = np.random.default_rng(seed)
    C_ = max(1, min(int(n_clusters), n_items))
    pu = 1.0 / np.arange(1, n_users + 1, dtype=np.float64) ** user_s
    pi = 1.0 / np.arange(1, n_items + 1, dtype=np.float64) ** item_s
    cu, ci = np.cumsum(pu / pu.sum()), np.cumsum(pi / pi.sum())
    perm_u, perm_i = rng.permutation(n_users), rng.permutation(n_items)
    ur = np.minimum(np.searchsorted(cu, rng.random(n_draws)), n_users - 1)          # user popularity rank per draw
    home = rng.integers(0, C_, size=n_users)                                       # home cluster of the user of rank r
    inside = rng.random(n_draws) < p_in
    ir = np.minimum(np.searchsorted(ci, rng.random(n_draws)), n_items - 1)          # global draw (item rank)
    # in-cluster draw: cluster c holds ranks c, c + C, c + 2C, ...; inverse-CDF over its own slice of the law
    h = home[ur[inside]]
    rnd = rng.random(h.shape[0])
    order = np.argsort(h, kind="stable")
    ir_in = np.empty(h.shape[0], dtype=np.int64)
    bounds = np.searchsorted(h[order], np.arange(C_ + 1))
    for c in range(C_):
        sel = order[bounds[c]:bounds[c + 1]]
        if sel.size == 0:
            continue
        ranks = np.arange(c, n_items, C_)
        cc = np.cumsum(pi[ranks])
        ir_in[sel] = ranks[np.minimum(np.searchsorted(cc / cc[-1], rnd[sel]), len(ranks) - 1)]
    ir[inside] = ir_in
    u, i = perm_u[ur], perm_i[ir]
    key = np.unique(u.astype(np.int64) * n_items + i)
    return (key // n_items).astype(np.int32), (key % n_items).astype(np.int32)


def structured_matrix(n_users: int, n_items: int, n_draws: int, seed: int, float_ratings: bool = True,
                      n_clusters: int = 80, p_in: float = 0.85) -> sp.csr_matrix:
    """interaction_matrix over clustered_pairs: U x I CSR float32 with item-item structure."""
    u, i = clustered_pairs(n_users, n_items, n_draws, seed, n_clusters=n_clusters, p_in=p_in)
    rng = np.random.default_rng(seed + 1)
    if float_ratings:
        v = (rng.integers(1, 6, size=len(u)) * np.exp(-rng.random(len(u)) * 0.7)).astype(np.float32)
    else:
        v = rng.integers(1, 6, size=len(u)).astype(np.float32)
    X = sp.csr_matrix((v, (u, i)), shape=(n_users, n_items), dtype=np.float32)
    X.sort_indices()
    return X


def workload_matrix(wl: dict, seed: int = 20251003, float_ratings: bool = True) -> sp.csr_matrix:
    """The interaction matrix of a bench.py WORKLOADS entry ({"U", "I", "draws", optional "gen": "zipf" | "clustered",
    "clusters", "p_in"}): one definition for bench.py, the tools and the full-size GPU tests."""
    if wl.get("gen", "zipf") == "clustered":
        return structured_matrix(wl["U"], wl["I"], wl["draws"], seed, float_ratings=float_ratings,
                                 n_clusters=wl.get("clusters", 80), p_in=wl.get("p_in", 0.85))
    return interaction_matrix(wl["U"], wl["I"], wl["draws"], seed, float_ratings=float_ratings)
