"""GPU tests at BASELINE.json's full single-GPU size (C2: 100k users x 50k items, 5M draws, K=50).

At this size the oracle cannot check everything in seconds, so the whole output is checked through
size-independent properties (sortedness, uniqueness, filter respected, shard merge == unsharded,
idempotence) and random samples are compared bit-for-bit with the oracle.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from rtrec_amd import _native
from rtrec_amd.engine import SlimEngine, coefficients_to_updates, merge_coefficients
from rtrec_amd.synth import interaction_matrix

pytestmark = pytest.mark.gpu
U, I, DRAWS, K = 100_000, 50_000, 5_000_000, 50


@pytest.fixture(scope="module")
def c2():
    X = interaction_matrix(U, I, DRAWS, seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    tg, items, coef, count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K)
    W = merge_coefficients(None, I, *coefficients_to_updates(tg, items, coef, count))
    eng.set_weights(W)
    return dict(X=X, Xc=Xc, eng=eng, W=W, tg=tg, items=items, coef=coef, count=count, n_iter=n_iter)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_fit_sampled_columns_match_oracle(c2, oracle):
    rng = np.random.default_rng(1)
    nnz = np.diff(c2["Xc"].indptr)
    heavy = np.argsort(-nnz)[:12]                      # the columns with real CD work
    active = np.flatnonzero(np.diff(c2["W"].indptr) > 0)
    sample = np.unique(np.concatenate([heavy, rng.choice(active, 40, replace=False), rng.choice(I, 60, replace=False)]))
    pos = {int(t): k for k, t in enumerate(c2["tg"])}
    ptr, idx, val, nit = oracle.fit_columns(c2["Xc"], sample, nn_feature_selection=K)
    for n, j in enumerate(sample):
        k = pos[int(j)]
        c = c2["count"][k]
        o = np.argsort(c2["items"][k, :c], kind="stable")
        assert c2["n_iter"][k] == nit[n], f"column {j}"
        assert np.array_equal(c2["items"][k, :c][o], idx[ptr[n]:ptr[n + 1]]), f"column {j}"
        assert np.array_equal(bits(c2["coef"][k, :c][o]), bits(val[ptr[n]:ptr[n + 1]])), f"column {j}"


def test_fit_is_idempotent(c2):
    sample = np.arange(0, I, 97)
    tg, items, coef, count, n_iter = c2["eng"].fit_columns(sample, nn_feature_selection=K)
    pos = {int(t): k for k, t in enumerate(c2["tg"])}
    for n, j in enumerate(tg):
        k = pos[int(j)]
        assert np.array_equal(items[n], c2["items"][k]) and np.array_equal(bits(coef[n]), bits(c2["coef"][k]))


def test_recommend_all_users_properties_and_samples(c2, oracle):
    eng, X, W = c2["eng"], c2["X"], c2["W"]
    rows = np.arange(U)
    ids, sc, cnt = eng.recommend_rows(rows, top_k=10, filter_interacted=True, mode=_native.TOPK_SPARSE)
    k = np.arange(10)[None, :]
    valid = k < cnt[:, None]
    assert np.all(ids[valid] >= 0) and np.all(ids[~valid] == -1) and np.all(np.isneginf(sc[~valid]))
    # descending scores
    with np.errstate(invalid="ignore"):
        d = sc[:, 1:] - sc[:, :-1]
    assert np.all((d <= 0) | ~valid[:, 1:])
    # ids unique per row, never an interacted item, always a column that stores weights
    srt = np.sort(np.where(valid, ids, -np.arange(1, 11)[None, :]), axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1])
    seen = X[np.repeat(rows, 10)[valid.ravel()], ids[valid]]
    assert seen.nnz == 0 if sp.issparse(seen) else not np.any(seen)
    active = np.zeros(I, bool)
    active[np.flatnonzero(np.diff(W.indptr) > 0)] = True
    assert np.all(active[ids[valid]])
    # idempotence
    ids2, sc2, cnt2 = eng.recommend_rows(rows, top_k=10)
    assert np.array_equal(ids, ids2) and np.array_equal(bits(sc), bits(sc2)) and np.array_equal(cnt, cnt2)
    # random sample vs oracle, bit for bit
    sample = np.sort(np.random.default_rng(3).choice(U, 3000, replace=False))
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X[sample], W.tocsr(), top_k=10)
    assert np.array_equal(ids[sample], o_ids) and np.array_equal(cnt[sample], o_cnt)
    assert np.array_equal(bits(sc[sample]), bits(o_sc))
    # filter_interacted=False returns a superset ranking: its head may contain seen items
    ids3, _, cnt3 = eng.recommend_rows(sample[:500], top_k=10, filter_interacted=False)
    o3, _, c3 = oracle.recommend_batch(X[sample[:500]], W.tocsr(), top_k=10, filter_interacted=False)
    assert np.array_equal(ids3, o3) and np.array_equal(cnt3, c3)


def test_column_shards_merge_to_unsharded(c2):
    import torch
    X, W = c2["X"], c2["W"]
    rows = np.arange(0, U, 50, dtype=np.int32)
    ids, sc, cnt = c2["eng"].recommend_rows(rows, top_k=10)
    parts = []
    G = 4
    for r in range(G):
        e = SlimEngine(device="cuda:0", rank=r, world_size=G)
        e._X = c2["eng"]._X
        e.n_users, e.n_items = U, I
        e.set_weights(W)
        d_rows = e.be.to_dev(rows)
        xb = (e._X["rptr"], e._X["rcol"], e._X["rval"])
        parts.append(e._local_topk(d_rows, len(rows), xb, 10, True, _native.TOPK_SPARSE, None))
    be = c2["eng"].be
    g = [torch.stack([p[k] for p in parts]).contiguous() for k in (0, 1, 3, 4)]
    o_ids = be.empty((len(rows), 10), torch.int32)
    o_sc = be.empty((len(rows), 10), torch.float32)
    o_cnt = be.empty((len(rows),), torch.int32)
    be.merge_topk(len(rows), G, 10, g[0], g[1], None, g[2], g[3], o_ids, o_sc, o_cnt)
    assert np.array_equal(o_ids.cpu().numpy(), ids)
    assert np.array_equal(bits(o_sc.cpu().numpy()), bits(sc))
    assert np.array_equal(o_cnt.cpu().numpy(), cnt)


def test_dense_row_blocks_equal_sparse_rows(oracle, monkeypatch):
    """The zero-padded dense W-row blocks must not change a single bit (incl. the exact-tie pass)."""
    import rtrec_amd.engine as E
    X = interaction_matrix(1500, 700, 40000, seed=3)
    Xc = X.tocsc()
    Xc.sort_indices()
    ptr, idx, val, _ = oracle.fit_columns(Xc, np.arange(700), nn_feature_selection=20)
    W = merge_coefficients(None, 700, idx.astype(np.int64), np.repeat(np.arange(700), np.diff(ptr)), val)
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X, W.tocsr(), top_k=10)
    for fill in (None, 0.02, 0.5):
        monkeypatch.setattr(E, "DENSE_ROW_FILL", fill if fill is not None else 2.0)
        for tile in (256, 8192):
            eng = SlimEngine(device="cuda:0", tile_cols=tile)
            eng.set_interactions(None, X, need_csc=False)
            eng.set_weights(W)
            ids, sc, cnt = eng.recommend_rows(np.arange(1500), top_k=10)
            lay = eng._layout(True)
            if fill == 0.02:
                assert lay["n_dense"] > 0
            assert np.array_equal(ids, o_ids) and np.array_equal(cnt, o_cnt) and np.array_equal(bits(sc), bits(o_sc))
