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
