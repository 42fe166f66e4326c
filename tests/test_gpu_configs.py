"""GPU tests at the FULL size of BASELINE.json's other configurations (C2 is tests/test_gpu_fullsize.py):

  C3  MovieLens-20M shape, 138,493 x 26,744, ~21 M interactions, K = 50 (the bench workload)
  C4  1 M users x 500 k items, ~93 M interactions: bulk fit + score, then streamed 1,000-interaction
      SLIM.fit mini-batches on the device-resident store against a host-export run
  C5  C3 shape, SLIM(min_value=0, max_value=15, decay_in_days=180, nn_feature_selection=50) (README :89),
      similar_items for ALL items and recommend_batch on the decayed matrix

At these sizes the oracle checks SAMPLES bit for bit (the heaviest target columns -- the ones with real
coordinate-descent work -- plus random ones; random users) and the whole output is checked through
size-independent properties: sorted scores, unique ids, the interacted filter, only columns that store
weights, idempotence, and the two score kernels (feature-row and tiled-CSR) agreeing on every row.
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from rtrec_amd import _native
from rtrec_amd.engine import SlimEngine, coefficients_to_updates, merge_coefficients
from rtrec_amd.synth import interaction_matrix, zipf_pairs

pytestmark = pytest.mark.gpu
CPU_THREADS = max(1, min(16, len(os.sched_getaffinity(0))))       # the host share of one GPU
T0 = 1_700_000_000.0


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def check_topk_properties(ids, sc, cnt, X, W, rows, top_k):
    """Everything a correct top-k list satisfies whatever the size of the problem."""
    k = np.arange(top_k)[None, :]
    valid = k < cnt[:, None]
    assert np.all(ids[valid] >= 0) and np.all(ids[~valid] == -1) and np.all(np.isneginf(sc[~valid]))
    with np.errstate(invalid="ignore"):
        d = sc[:, 1:] - sc[:, :-1]
    assert np.all((d <= 0) | ~valid[:, 1:]), "scores must be descending"
    srt = np.sort(np.where(valid, ids, -np.arange(1, top_k + 1)[None, :]), axis=1)
    assert np.all(srt[:, 1:] != srt[:, :-1]), "ids must be unique per row"
    seen = X[np.repeat(rows, top_k)[valid.ravel()], ids[valid]]
    assert (seen.nnz == 0) if sp.issparse(seen) else (not np.any(seen)), "an interacted item was recommended"
    active = np.zeros(W.shape[0], bool)
    active[np.flatnonzero(np.diff(W.indptr) > 0)] = True
    assert np.all(active[ids[valid]]), "only columns that store weights can score"


def check_fit_sample(oracle, Xc, sample, tg, items, coef, count, n_iter, K):
    pos = {int(t): k for k, t in enumerate(tg)}
    ptr, idx, val, nit = oracle.fit_columns(Xc, sample, nn_feature_selection=K, n_threads=CPU_THREADS)
    for n, j in enumerate(sample):
        k = pos[int(j)]
        c = count[k]
        o = np.argsort(items[k, :c], kind="stable")
        assert n_iter[k] == nit[n], f"column {j}: sweeps"
        assert np.array_equal(items[k, :c][o], idx[ptr[n]:ptr[n + 1]]), f"column {j}: features"
        assert np.array_equal(bits(coef[k, :c][o]), bits(val[ptr[n]:ptr[n + 1]])), f"column {j}: coefficients"


# ------------------------------------------------------------------------------------------ C3
C3 = dict(U=138_493, I=26_744, draws=26_000_000, K=50)


@pytest.fixture(scope="module")
def c3():
    U, I, K = C3["U"], C3["I"], C3["K"]
    X = interaction_matrix(U, I, C3["draws"], seed=20251003, float_ratings=True)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    tg, items, coef, count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K)
    W = merge_coefficients(None, I, *coefficients_to_updates(tg, items, coef, count))
    eng.set_weights(W)
    yield dict(X=X, Xc=Xc, eng=eng, W=W, tg=tg, items=items, coef=coef, count=count, n_iter=n_iter)
    import torch
    del eng
    torch.cuda.empty_cache()


def test_c3_fit_all_columns_match_oracle(c3, oracle):
    """The exact fit against the oracle over the WHOLE catalogue: feature sets, coefficient bits and sweep counts of all 26,744
    columns (the 16-thread oracle takes about a minute; rounds 1-3 compared ~110 sampled columns, round 4 every third by default).
    RTREC_AMD_FULL_PARITY=0 goes back to every third column plus the targets with the most work and the longest columns."""
    I = C3["I"]
    if os.environ.get("RTREC_AMD_FULL_PARITY", "1") != "0":
        cols = np.arange(I)
    else:
        nnz = np.diff(c3["Xc"].indptr)
        cols = np.unique(np.concatenate([np.arange(0, I, 3), c3["tg"][np.argsort(-c3["n_iter"] * 1.0)[:8]], np.argsort(-nnz)[:8]]))
    check_fit_sample(oracle, c3["Xc"], cols, c3["tg"], c3["items"], c3["coef"], c3["count"], c3["n_iter"], C3["K"])


def test_c3_fit_is_idempotent(c3):
    sample = np.arange(0, C3["I"], 53)
    tg, items, coef, count, n_iter = c3["eng"].fit_columns(sample, nn_feature_selection=C3["K"])
    pos = {int(t): k for k, t in enumerate(c3["tg"])}
    for n, j in enumerate(tg):
        k = pos[int(j)]
        assert np.array_equal(items[n], c3["items"][k]) and np.array_equal(bits(coef[n]), bits(c3["coef"][k]))
        assert n_iter[n] == c3["n_iter"][k]


def test_c3_recommend_all_users_properties_samples_and_both_kernels(c3, oracle):
    eng, X, W, U = c3["eng"], c3["X"], c3["W"], C3["U"]
    rows = np.arange(U)
    ids, sc, cnt = eng.recommend_rows(rows, top_k=10, filter_interacted=True, mode=_native.TOPK_SPARSE)
    check_topk_properties(ids, sc, cnt, X, W, rows, 10)
    ids2, sc2, cnt2 = eng.recommend_rows(rows, top_k=10)
    assert np.array_equal(ids, ids2) and np.array_equal(bits(sc), bits(sc2)) and np.array_equal(cnt, cnt2), "idempotence"
    # the feature-row kernel and the tiled-CSR kernel are two implementations of the same sums
    lay = eng._layout(True)
    assert lay.get("fr_w") is not None, "the ML-20M-shape W (66 long rows) must take the feature-row kernel"
    eng.use_feature_rows = False
    try:
        ids3, sc3, cnt3 = eng.recommend_rows(rows, top_k=10)
    finally:
        eng.use_feature_rows = True
    assert np.array_equal(ids, ids3) and np.array_equal(bits(sc), bits(sc3)) and np.array_equal(cnt, cnt3)
    # the oracle over ALL 138,493 users (round 4: not a sample -- a defect common to both kernels would show here)
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X, W.tocsr(), top_k=10, n_threads=CPU_THREADS)
    bad = np.flatnonzero((ids != o_ids).any(axis=1) | (cnt != o_cnt) | (bits(sc) != bits(o_sc)).any(axis=1))
    assert bad.size == 0, f"{bad.size} users differ from the oracle, first {bad[:5]}"
    sample = np.sort(np.random.default_rng(3).choice(U, 4000, replace=False))
    ids4, _, cnt4 = eng.recommend_rows(sample[:500], top_k=10, filter_interacted=False)
    o4, _, c4 = oracle.recommend_batch(X[sample[:500]], W.tocsr(), top_k=10, filter_interacted=False)
    assert np.array_equal(ids4, o4) and np.array_equal(cnt4, c4)


# ------------------------------------------------------------------------------------------ C4
C4 = dict(U=1_000_000, I=500_000, draws=100_000_000, K=50, n_stream=3, batch=1000)


def _c4_stream():
    u, i = zipf_pairs(C4["U"], C4["I"], C4["draws"], seed=20251003)
    rng = np.random.default_rng(5)
    order = rng.permutation(len(u))
    u, i = u[order].astype(np.int64), i[order].astype(np.int64)
    r = (rng.integers(1, 6, len(u)) * np.exp(-rng.random(len(u)) * 0.7)).astype(np.float64)
    ts = T0 + np.arange(len(u), dtype=np.float64)
    return u, i, ts, r


def _bulk_ingest(model, u, i, ts, r, n_bulk, chunk=4_000_000):
    for a in range(0, n_bulk, chunk):
        b = min(a + chunk, n_bulk)
        model.add_interactions_columns(u[a:b], i[a:b], ts[a:b], r[a:b])


def test_c3_row_sets_scored_again_get_the_grouped_order(c3):
    """A row tensor scored a second time gets the pattern-grouped work order (SlimEngine._row_order keeps one entry per row
    tensor: several live at once here); the answers do not depend on the order."""
    eng = c3["eng"]
    U = eng.n_users
    sets = [eng.be.to_dev(np.arange(a, a + 40_000, dtype=np.int32)) for a in (0, 30_000, U - 40_000)]
    first = []
    for d in sets:                                     # first sight: the length order
        o = eng.score_topk_device(None, 40_000, 10, True, _native.TOPK_SPARSE, d_rows=d)
        first.append([t.cpu().numpy() for t in o])
    for rep in range(2):                               # second sight builds the grouped order, third reuses it
        for d, f in zip(sets, first):
            o = eng.score_topk_device(None, 40_000, 10, True, _native.TOPK_SPARSE, d_rows=d)
            assert eng.last_score_path == "feature_rows"
            assert all(np.array_equal(a.cpu().numpy().view(np.int32), b.view(np.int32)) for a, b in zip(o, f))
    # the grouped order is a permutation, and its giant rows (more than ROW_ORDER_GIANT_LEN entries) sit at every 8th position
    # of the head, longest first: a wave takes consecutive positions, and giants must not share one (round 4)
    X = c3["X"]
    lens = np.diff(X.indptr)
    for d in sets:
        rows = d.cpu().numpy()
        ent = next(e for e in eng._X["_orders"] if e[0] is d)
        assert ent[5], "the cached order of a row set scored three times is the grouped one"
        order = ent[4].cpu().numpy()
        assert np.array_equal(np.sort(order), np.arange(len(rows)))
        giants = np.flatnonzero(lens[rows] > eng.ROW_ORDER_GIANT_LEN)
        giants = giants[np.argsort(-lens[rows][giants], kind="stable")][:eng.ROW_ORDER_GIANTS]
        assert len(giants) > 0, "the ML-20M shape has users of more than 4,096 items in every 40k-user slice"
        assert np.array_equal(lens[rows][order[0:8 * len(giants):8]], lens[rows][giants])
        assert (lens[rows][order[1:8]] <= eng.ROW_ORDER_GIANT_LEN).all()


@pytest.fixture(scope="module")
def c4():
    from rtrec_amd import SLIM
    u, i, ts, r = _c4_stream()
    n_bulk = len(u) - C4["n_stream"] * C4["batch"]
    model = SLIM(min_value=0, max_value=15, nn_feature_selection=C4["K"])
    _bulk_ingest(model, u, i, ts, r, n_bulk)
    model.bulk_fit(parallel=True, progress_bar=False)
    yield dict(model=model, u=u, i=i, ts=ts, r=r, n_bulk=n_bulk)
    import torch
    del model
    torch.cuda.empty_cache()


def test_c4_bulk_fit_and_score_at_full_size(c4, oracle):
    m = c4["model"]
    W = m.model.item_similarity
    U, I = m.interactions.shape
    assert U > 990_000 and I > 490_000 and m.interactions.nnz > 90_000_000
    X = m.interactions.to_csr()
    rng = np.random.default_rng(7)
    users = np.sort(rng.choice(U, 200_000, replace=False))
    recs = m.recommend_batch(users.tolist(), top_k=10)
    ids = np.full((len(users), 10), -1, np.int32)
    cnt = np.array([len(x) for x in recs], np.int32)
    for n, row in enumerate(recs):
        ids[n, :len(row)] = row
    # scores are not part of the list API: recompute them for the property check from the engine
    e_ids, e_sc, e_cnt = m.model.engine.recommend_rows(users, top_k=10)
    assert np.array_equal(e_ids, ids) and np.array_equal(e_cnt, cnt)
    check_topk_properties(e_ids, e_sc, e_cnt, X, W, users, 10)
    # the oracle over ALL 200,000 scored users (round 4: not a sample of 600)
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X[users], W.tocsr(), top_k=10, n_threads=CPU_THREADS)
    bad = np.flatnonzero((e_ids != o_ids).any(axis=1) | (e_cnt != o_cnt) | (bits(e_sc) != bits(o_sc)).any(axis=1))
    assert bad.size == 0, f"{bad.size} users differ from the oracle, first {users[bad[:5]]}"
    # fitted columns: the heaviest targets and random ones against the oracle on the exported matrix
    Xc = m.interactions.to_csc()
    nnz = np.diff(Xc.indptr)
    sample_c = np.unique(np.concatenate([np.argsort(-nnz)[:6], rng.choice(np.flatnonzero(np.diff(W.indptr) > 0), 12, replace=False),
                                         rng.choice(I, 14, replace=False)]))
    ptr, idx, val, _ = oracle.fit_columns(Xc, sample_c, nn_feature_selection=C4["K"], n_threads=CPU_THREADS)
    for n, j in enumerate(sample_c):
        col = W[:, int(j)].tocoo()
        nzo = val[ptr[n]:ptr[n + 1]] != 0          # the W matrix keeps non-zero coefficients only
        assert np.array_equal(np.sort(col.row), idx[ptr[n]:ptr[n + 1]][nzo]), f"column {j}"
        assert np.array_equal(bits(col.data[np.argsort(col.row)]), bits(val[ptr[n]:ptr[n + 1]][nzo])), f"column {j}"


def test_c4_streamed_mini_batches_on_the_device_store_equal_a_host_export_run(c4, monkeypatch):
    """SLIM.fit of 1,000-interaction mini-batches: the device-resident store (X merged in HBM, touched columns
    gathered there) against the same batches through host exports (to_csc(select_items), slim.py:33-36)."""
    from rtrec_amd import SLIM
    m, u, i, ts, r, n_bulk = (c4[k] for k in ("model", "u", "i", "ts", "r", "n_bulk"))
    assert m._dev_x is not None and m._dev_x.version == m._store_tag(), "the bulk fit must have left X resident"
    monkeypatch.setenv("RTREC_AMD_DEVICE_STORE", "0")
    host = SLIM(min_value=0, max_value=15, nn_feature_selection=C4["K"])
    _bulk_ingest(host, u, i, ts, r, n_bulk)
    host.model.item_similarity = m.model.item_similarity.copy()
    monkeypatch.setenv("RTREC_AMD_DEVICE_STORE", "1")
    for k in range(C4["n_stream"]):
        a = n_bulk + k * C4["batch"]
        b = a + C4["batch"]
        batch = list(zip(u[a:b].tolist(), i[a:b].tolist(), ts[a:b].tolist(), r[a:b].tolist()))
        monkeypatch.setenv("RTREC_AMD_DEVICE_STORE", "1")
        m.fit(batch, progress_bar=False)
        assert m._dev_x is not None and m._dev_x.version == m._store_tag(), "the mini-batch must advance the resident X"
        monkeypatch.setenv("RTREC_AMD_DEVICE_STORE", "0")
        host.fit(batch, progress_bar=False)
        A, B = m.model.item_similarity, host.model.item_similarity
        A.sort_indices(); B.sort_indices()
        assert A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices), f"batch {k}"
        assert np.array_equal(bits(A.data), bits(B.data)), f"batch {k}"
    monkeypatch.setenv("RTREC_AMD_DEVICE_STORE", "1")
    users = np.unique(u[n_bulk:])[:2000].tolist()
    assert m.recommend_batch(users, top_k=10) == host.recommend_batch(users, top_k=10)


# ------------------------------------------------------------------------------------------ C5
@pytest.fixture(scope="module")
def c5():
    from rtrec_amd import SLIM
    u, i = zipf_pairs(C3["U"], C3["I"], C3["draws"], seed=20251003)
    rng = np.random.default_rng(9)
    order = rng.permutation(len(u))
    u, i = u[order].astype(np.int64), i[order].astype(np.int64)
    r = rng.integers(1, 6, len(u)).astype(np.float64)
    ts = T0 + np.sort(rng.random(len(u))) * 300 * 86400.0          # 300 days, ascending
    model = SLIM(min_value=0, max_value=15, decay_in_days=180, nn_feature_selection=50)
    for a in range(0, len(u), 4_000_000):
        model.add_interactions_columns(u[a:a + 4_000_000], i[a:a + 4_000_000], ts[a:a + 4_000_000], r[a:a + 4_000_000])
    model.bulk_fit(parallel=True, progress_bar=False)
    yield model
    import torch
    del model
    torch.cuda.empty_cache()


def test_c5_decayed_fit_similar_items_for_all_items_and_recommend(c5, oracle):
    m = c5
    W = m.model.item_similarity
    U, I = m.interactions.shape
    assert m.interactions.decay_rate is not None and W.dtype == np.float32
    # fit on the DECAYED matrix: heaviest and random columns against the oracle
    Xc = m.interactions.to_csc()
    assert float(Xc.data.max()) < 5.0 + 1e-6 and np.unique(Xc.data).size > 1000, "values must carry the time decay"
    rng = np.random.default_rng(13)
    nnz = np.diff(Xc.indptr)
    sample_c = np.unique(np.concatenate([np.argsort(-nnz)[:8], rng.choice(np.flatnonzero(np.diff(W.indptr) > 0), 30, replace=False),
                                         rng.choice(I, 40, replace=False)]))
    ptr, idx, val, _ = oracle.fit_columns(Xc, sample_c, nn_feature_selection=50, n_threads=CPU_THREADS)
    for n, j in enumerate(sample_c):
        col = W[:, int(j)].tocoo()
        nzo = val[ptr[n]:ptr[n + 1]] != 0
        assert np.array_equal(np.sort(col.row), idx[ptr[n]:ptr[n + 1]][nzo]), f"column {j}"
        assert np.array_equal(bits(col.data[np.argsort(col.row)]), bits(val[ptr[n]:ptr[n + 1]][nzo])), f"column {j}"
    # similar_items for ALL items (one launch) against the oracle's per-item answer
    got = m.similar_items_batch(list(range(I)), top_k=10, ret_scores=True)
    Wc = W.tocsc()
    n_nonempty = 0
    for j in range(I):
        oi, ov = oracle.similar_items(Wc, j, top_k=10)
        assert [a for a, _ in got[j]] == oi.tolist(), f"item {j}"
        assert np.array_equal(bits([b for _, b in got[j]]), bits(ov)), f"item {j}"
        n_nonempty += len(oi) > 0
    assert n_nonempty > 1000
    # recommend_batch on the decayed matrix
    X = m.interactions.to_csr()
    users = np.sort(rng.choice(U, 60_000, replace=False))
    ids, sc, cnt = m.model.engine.recommend_rows(users, top_k=10)
    assert m.recommend_batch(users[:3000].tolist(), top_k=10) == [row[:c].tolist() for row, c in zip(ids[:3000], cnt[:3000])]
    check_topk_properties(ids, sc, cnt, X, W, users, 10)
    pos = rng.choice(len(users), 3000, replace=False)
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X[users[pos]], W.tocsr(), top_k=10, n_threads=CPU_THREADS)
    assert np.array_equal(ids[pos], o_ids) and np.array_equal(cnt[pos], o_cnt) and np.array_equal(bits(sc[pos]), bits(o_sc))


# ------------------------------------------------------------------------------------------ C3S
# The C3 shape with item-item structure (rtrec_amd.synth.clustered_pairs: 80 item clusters, 85 % of a user's draws inside
# its home cluster): W has thousands of non-empty rows -- the general-W scoring path (csrc/score_seg.hip.h).
C3S = dict(U=138_493, I=26_744, draws=46_000_000, K=50, gen="clustered", clusters=80, p_in=0.85)


@pytest.fixture(scope="module")
def c3s():
    from rtrec_amd.synth import workload_matrix
    I, K = C3S["I"], C3S["K"]
    X = workload_matrix(C3S)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    tg, items, coef, count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K)
    W = merge_coefficients(None, I, *coefficients_to_updates(tg, items, coef, count))
    eng.set_weights(W)
    yield dict(X=X, Xc=Xc, eng=eng, W=W, tg=tg, items=items, coef=coef, count=count, n_iter=n_iter)
    import torch
    del eng
    torch.cuda.empty_cache()


def test_c3s_fit_all_columns_match_oracle(c3s, oracle):
    """Like test_c3_fit_all_columns_match_oracle, on the structured workload: every column."""
    check_fit_sample(oracle, c3s["Xc"], np.arange(C3S["I"]), c3s["tg"], c3s["items"], c3s["coef"], c3s["count"], c3s["n_iter"],
                     C3S["K"])


def test_c3s_recommend_all_users_segment_kernel(c3s, oracle):
    eng, X, W, U = c3s["eng"], c3s["X"], c3s["W"], C3S["U"]
    assert np.count_nonzero(np.diff(W.tocsr().indptr)) > 1000, "the structured workload must give W thousands of rows"
    rows = np.arange(U)
    ids, sc, cnt = eng.recommend_rows(rows, top_k=10, filter_interacted=True, mode=_native.TOPK_SPARSE)
    assert eng.last_score_path == "segments"
    check_topk_properties(ids, sc, cnt, X, W, rows, 10)
    ids2, sc2, cnt2 = eng.recommend_rows(rows, top_k=10)
    assert np.array_equal(ids, ids2) and np.array_equal(bits(sc), bits(sc2)) and np.array_equal(cnt, cnt2), "idempotence"
    # the segment kernel and the tiled-CSR kernel are two implementations of the same sums: ALL rows must agree
    eng.use_seg_layout = False
    try:
        ids3, sc3, cnt3 = eng.recommend_rows(rows, top_k=10)
        assert eng.last_score_path == "tiled"
    finally:
        eng.use_seg_layout = True
    assert np.array_equal(ids, ids3) and np.array_equal(bits(sc), bits(sc3)) and np.array_equal(cnt, cnt3)
    # oracle: random users plus the longest ones (the heavy pass) and the lists just below its threshold
    lens = np.diff(X.indptr)
    by_len = np.argsort(-lens)
    sample = np.unique(np.concatenate([np.random.default_rng(3).choice(U, 3000, replace=False), by_len[:300],
                                       np.flatnonzero((lens > 400) & (lens <= 512))[:300]]))
    # the oracle over ALL users at top-10 (round 4: not a sample), then the sample at other k / filter settings
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X, W.tocsr(), top_k=10, n_threads=CPU_THREADS)
    bad = np.flatnonzero((ids != o_ids).any(axis=1) | (cnt != o_cnt) | (bits(sc) != bits(o_sc)).any(axis=1))
    assert bad.size == 0, f"{bad.size} users differ from the oracle, first {bad[:5]}"
    for k, filt in ((25, True), (10, False), (1, True)):
        ids4, sc4, cnt4 = eng.recommend_rows(sample[:600], top_k=k, filter_interacted=filt)
        assert eng.last_score_path == "segments"
        o4, s4, c4 = oracle.recommend_batch(X[sample[:600]], W.tocsr(), top_k=k, filter_interacted=filt, n_threads=CPU_THREADS)
        assert np.array_equal(ids4, o4) and np.array_equal(cnt4, c4) and np.array_equal(bits(sc4), bits(s4))


def test_c3s_dense_mode_and_float64_w_at_full_size(c3s, oracle):
    """The other two forms the reference scores in, at full size: DENSE mode (string item ids, slim_elastic.py:745-778) and a
    float64 W (the serial fit, :252).  Both run the fast pass plus the rows it flags; ALL rows must equal the tiled kernel
    of the same mode and the oracle's (all 138,493 users)."""
    eng, X, W, U = c3s["eng"], c3s["X"], c3s["W"], C3S["U"]
    rows = np.arange(U)
    Wr = W.tocsr()
    # DENSE
    d = eng.recommend_rows(rows, top_k=10, mode=_native.TOPK_DENSE)
    assert eng.last_score_path == "segments"
    eng.dense_fast = False
    try:
        t = eng.recommend_rows(rows, top_k=10, mode=_native.TOPK_DENSE)
        assert eng.last_score_path == "tiled"
    finally:
        eng.dense_fast = True
    assert all(np.array_equal(a.view(np.int32), b.view(np.int32)) for a, b in zip(d, t))
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X, Wr, top_k=10, dense=True, n_threads=CPU_THREADS)          # ALL users (round 4)
    assert np.array_equal(d[0], o_ids) and np.array_equal(bits(d[1]), bits(o_sc)) and np.array_equal(d[2], o_cnt)
    # float64 W
    try:
        eng.set_weights(W.astype(np.float64), acc_f64=True)
        f = eng.recommend_rows(rows, top_k=10)
        assert eng.last_score_path == "segments+f64"
        eng.f64_refine = False
        eng.set_weights(W.astype(np.float64), acc_f64=True)
        t = eng.recommend_rows(rows, top_k=10)
        assert eng.last_score_path == "tiled"
        assert all(np.array_equal(a.view(np.int32), b.view(np.int32)) for a, b in zip(f, t))
        o_ids, o_sc, o_cnt = oracle.recommend_batch(X, Wr, top_k=10, use_f64=True, n_threads=CPU_THREADS)    # ALL users (round 4)
        assert np.array_equal(f[0], o_ids) and np.array_equal(bits(f[1]), bits(o_sc)) and np.array_equal(f[2], o_cnt)
    finally:
        eng.f64_refine = True
        eng.set_weights(W)


# ------------------------------------------------------------------------------------------ mid-size reference fixtures
def test_midsize_reference_models_on_gpu(engine):
    """tests/golden/midsize.json (real SLIMElastic at the ML-1M shape and on a structured 3000 x 1500 matrix): the HIP fit's W
    has the reference's CSC checksums and scikit-learn's n_iter_ for every column."""
    from tests.test_oracle_golden import check_model_crc, midsize, midsize_matrix
    for name in ("ml1m", "s3000"):
        X = midsize_matrix(name)
        I = X.shape[1]
        engine.set_interactions(X, X.tocsr())
        tg, items, coef, count, n_iter = engine.fit_columns(np.arange(I), nn_feature_selection=50)
        W = merge_coefficients(None, I, *coefficients_to_updates(tg, items, coef, count))
        order = np.argsort(tg)
        check_model_crc(midsize()[name], W, n_iter[order])


def test_midsize_long_columns_on_gpu(c3):
    """80 long target columns (12k .. 128k entries) of the ML-20M shape: features, coefficient bits and sweep counts of the
    full-size HIP fit equal scikit-learn's (fixture from the real reference)."""
    from tests.test_oracle_golden import midsize
    ref = midsize()["long_columns"]
    pos = {int(t): k for k, t in enumerate(c3["tg"])}
    for n, j in enumerate(ref["targets"]):
        k = pos[int(j)]
        c = c3["count"][k]
        o = np.argsort(c3["items"][k, :c], kind="stable")
        assert c3["n_iter"][k] == ref["n_iter"][n], f"column {j}: sweeps"
        assert np.array_equal(c3["items"][k, :c][o], ref["features"][n]), f"column {j}: features"
        assert np.array_equal(bits(c3["coef"][k, :c][o]), np.asarray(ref["coef_bits"][n], dtype=np.uint32)), f"column {j}: coefficients"


# ------------------------------------------------------------------------------------------ C4S: wide tiles at full size
def test_c4s_wide_tile_segment_path_at_full_size(oracle):
    """The 500k-item shape WITH item clusters and alpha = 0.005 (bench.py WORKLOADS["c4s"]: with the default alpha the L1
    threshold leaves a degenerate W at a million users): W gets tens of thousands of rows and ~100k active columns, so the
    segment layout takes tiles wider than 256 columns and the GENERIC accumulate loop -- the path that had no full-size
    workload.  All 1M users: segment kernels == tiled-CSR kernel == oracle on every row; the
    exact fit of sampled columns equals the oracle's."""
    import torch
    from bench import WORKLOADS
    from rtrec_amd.synth import workload_matrix
    wl = WORKLOADS["c4s"]
    U, I, K, alpha = wl["U"], wl["I"], wl["K"], wl["alpha"]
    X = workload_matrix(wl)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(Xc, X)
    d_t, d_items, d_coef, d_count, n_iter = eng.fit_columns(np.arange(I), nn_feature_selection=K, device_out=True, alpha=alpha)
    eng.set_weights(eng.merge_fit(None, I, False, d_t, d_items, d_coef, d_count))
    W = eng.weights.to_csc(torch)
    assert np.count_nonzero(np.diff(W.tocsr().indptr)) > 10_000, "the workload must give W tens of thousands of rows"
    # the exact fit of sampled columns (popular, middle, tail) against the oracle with the same alpha
    tg = d_t.cpu().numpy().astype(np.int64)
    lens = np.diff(Xc.indptr)
    sample = np.unique(np.concatenate([np.argsort(-lens)[:4], np.random.default_rng(5).choice(np.flatnonzero(lens > 50), 40, replace=False)]))
    pos = {int(t): k for k, t in enumerate(tg)}
    ptr, idx, val, nit = oracle.fit_columns(Xc, sample, nn_feature_selection=K, alpha=alpha, n_threads=CPU_THREADS)
    items, coef, count = d_items.cpu().numpy(), d_coef.cpu().numpy(), d_count.cpu().numpy()
    for n, j in enumerate(sample):
        k = pos[int(j)]
        c = count[k]
        o = np.argsort(items[k, :c], kind="stable")
        assert n_iter[k] == nit[n] and np.array_equal(items[k, :c][o], idx[ptr[n]:ptr[n + 1]])
        assert np.array_equal(bits(coef[k, :c][o]), bits(val[ptr[n]:ptr[n + 1]])), f"column {j}"
    rows = np.arange(U)
    ids, sc, cnt = eng.recommend_rows(rows, top_k=10)
    assert eng.last_score_path == "segments"
    sg = eng._fast_layout()["sg"]
    assert int(sg["sg_T"]) > 256, "the catalogue must need tiles wider than 256 columns"
    eng.use_seg_layout = False
    try:
        ids3, sc3, cnt3 = eng.recommend_rows(rows, top_k=10)
        assert eng.last_score_path == "tiled"
    finally:
        eng.use_seg_layout = True
    assert np.array_equal(cnt, cnt3)
    m = np.arange(10)[None, :] < cnt[:, None]
    assert np.array_equal(ids[m], ids3[m]) and np.array_equal(bits(sc)[m], bits(sc3)[m])
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X, W.tocsr(), top_k=10, n_threads=CPU_THREADS)          # ALL 1M users (round 4)
    bad = np.flatnonzero((ids != o_ids).any(axis=1) | (cnt != o_cnt) | (bits(sc) != bits(o_sc)).any(axis=1))
    assert bad.size == 0, f"{bad.size} users differ from the oracle, first {bad[:5]}"
    del eng
    torch.cuda.empty_cache()
