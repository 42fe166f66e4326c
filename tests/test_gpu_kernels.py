"""GPU parity: HIP kernels (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Bar: bit-exact for everything -- coefficient bits, sweep counts, selected features, score bits
and top-k ids (the kernels reproduce the reference's float32 operation order).
"""
import numpy as np
import pytest
import scipy.sparse as sp

from rtrec_amd import _native
from rtrec_amd.engine import SlimEngine, coefficients_to_updates, merge_coefficients
from rtrec_amd.synth import interaction_matrix

pytestmark = pytest.mark.gpu


def oracle_w(oracle, X_csc, cols, **kw):
    ptr, idx, val, nit = oracle.fit_columns(X_csc, cols, **kw)
    rows = idx.astype(np.int64)
    cc = np.repeat(np.asarray(cols, dtype=np.int64), np.diff(ptr))
    return merge_coefficients(None, X_csc.shape[1], rows, cc, val), (ptr, idx, val, nit)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("U,I,draws,K,positive,float_ratings", [
    (600, 200, 12000, None, True, True),
    (600, 200, 12000, 8, True, True),
    (600, 200, 12000, 8, False, True),
    (3000, 800, 90000, 50, True, True),
    (3000, 800, 90000, 50, True, False),     # integer ratings: ties everywhere, same tie rule both sides
    (400, 30, 3000, 50, True, True),         # K > I
])
# single-wave (throughput) kernel, multi-wave (latency) kernel, the latter with the column-walk X^T y forced, and with the
# one-pass X^T y of all the call's targets (xty_batch_kernel: LDS float atomics in the reference's fold order) forced
@pytest.mark.parametrize("fit_mode", ["sw", "mw", "mw-colwalk", "mw-xty"])
# screening (order-free pass + error bound before an ordered fold) from its default length, and forced on
# every column so that these small matrices exercise it
@pytest.mark.parametrize("screen_min", [None, "1"])
def test_fit_columns_bit_exact(engine, oracle, U, I, draws, K, positive, float_ratings, fit_mode, screen_min,
                               monkeypatch):
    monkeypatch.setenv("RTREC_AMD_FIT_MODE", fit_mode[:2])
    if screen_min is not None:
        monkeypatch.setenv("RTREC_AMD_SCREEN_MIN", screen_min)
    if fit_mode == "mw-colwalk":
        monkeypatch.setenv("RTREC_AMD_COLWALK_MIN", "1")
    monkeypatch.setenv("RTREC_AMD_XTY_BATCH", "force" if fit_mode == "mw-xty" else "0")
    X = interaction_matrix(U, I, draws, seed=11, float_ratings=float_ratings)
    Xc = X.tocsc()
    Xc.sort_indices()
    engine.set_interactions(Xc, X)
    cols = np.arange(I)
    tg, items, coef, count, n_iter = engine.fit_columns(cols, positive=positive, nn_feature_selection=K)
    ptr, idx, val, nit = oracle.fit_columns(Xc, tg, positive=positive, nn_feature_selection=K)
    assert np.array_equal(n_iter, nit), f"n_iter differs on {np.flatnonzero(n_iter != nit)[:10]}"
    assert np.array_equal(count, np.diff(ptr))
    for t in range(len(tg)):
        c = count[t]
        got_i, got_v = items[t, :c], coef[t, :c]
        if K is not None:   # kernel emits selection order, oracle ascending ids
            o = np.argsort(got_i, kind="stable")
            got_i, got_v = got_i[o], got_v[o]
        assert np.array_equal(got_i, idx[ptr[t]:ptr[t + 1]]), f"feature set differs for column {tg[t]}"
        assert np.array_equal(bits(got_v), bits(val[ptr[t]:ptr[t + 1]])), f"coefficient bits differ for column {tg[t]}"


def test_fit_subset_of_columns_and_empty_column(engine, oracle):
    X = interaction_matrix(500, 120, 6000, seed=5).tolil()
    X[:, 7] = 0          # an item nobody interacted with
    X = X.tocsr()
    X.eliminate_zeros()
    Xc = X.tocsc()
    Xc.sort_indices()
    engine.set_interactions(Xc, X)
    cols = np.array([7, 3, 119, 0, 64])
    for K in (None, 10):
        tg, items, coef, count, n_iter = engine.fit_columns(cols, nn_feature_selection=K)
        ptr, idx, val, nit = oracle.fit_columns(Xc, tg, nn_feature_selection=K)
        assert np.array_equal(n_iter, nit)
        assert n_iter[list(tg).index(7)] == 100      # y == 0 never meets gap < tol: all max_iter sweeps
        rows, cc, vals = coefficients_to_updates(tg, items, coef, count)
        W = merge_coefficients(None, 120, rows, cc, vals)
        Wo = merge_coefficients(None, 120, idx.astype(np.int64), np.repeat(tg, np.diff(ptr)), val)
        assert (W != Wo).nnz == 0 and np.array_equal(bits(W.data), bits(Wo.data))


def make_model(oracle, U=1500, I=700, draws=40000, K=20, seed=3):
    X = interaction_matrix(U, I, draws, seed=seed)
    Xc = X.tocsc()
    Xc.sort_indices()
    W, _ = oracle_w(oracle, Xc, np.arange(I), nn_feature_selection=K)
    return X, W


@pytest.mark.parametrize("feature_rows", [True, False])     # False: the tiled-CSR kernel also where the feature-row one applies
@pytest.mark.parametrize("tile_cols", [256, 8192])
@pytest.mark.parametrize("mode,filt", [("sparse", True), ("sparse", False), ("dense", True), ("dense", False)])
@pytest.mark.parametrize("f64", [False, True])
def test_score_topk_bit_exact(oracle, mode, filt, f64, tile_cols, feature_rows):
    if not feature_rows and (mode == "dense" or f64):
        pytest.skip("the feature-row kernel serves SPARSE mode with float32 accumulation only")
    X, W = make_model(oracle)
    eng = SlimEngine(device="cuda:0", tile_cols=tile_cols)
    eng.use_feature_rows = feature_rows
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W.astype(np.float64) if f64 else W, acc_f64=f64)
    rows = np.arange(0, X.shape[0], 3)
    m = _native.TOPK_DENSE if mode == "dense" else _native.TOPK_SPARSE
    ids, sc, cnt = eng.recommend_rows(rows, top_k=10, filter_interacted=filt, mode=m)
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows], W.tocsr(), top_k=10, filter_interacted=filt,
                                                dense=(mode == "dense"), use_f64=f64)
    assert np.array_equal(cnt, o_cnt)
    assert np.array_equal(ids, o_ids)
    assert np.array_equal(bits(sc), bits(o_sc))


def test_score_exact_ties_follow_reference_order(oracle):
    """Duplicate W columns give exactly equal scores; the sparse path must order them like
    Python's stable sorted() over scipy's reverse-first-touch product order."""
    rng = np.random.default_rng(0)
    I = 600
    base = sp.random(I, 40, density=0.08, random_state=1, format="csc", dtype=np.float32)
    pick = rng.integers(0, 40, size=I)
    W = sp.csc_matrix(base[:, pick])          # every column is a copy of one of 40 -> many exact ties
    W.sort_indices()
    X = interaction_matrix(300, I, 6000, seed=9)
    for tile, feature_rows in ((256, True), (1024, True), (256, False), (1024, False)):
        eng = SlimEngine(device="cuda:0", tile_cols=tile)
        eng.use_feature_rows = feature_rows
        eng.set_interactions(None, X, need_csc=False)
        eng.set_weights(W)
        rows = np.arange(X.shape[0])
        for filt in (True, False):
            ids, sc, cnt = eng.recommend_rows(rows, top_k=10, filter_interacted=filt, mode=_native.TOPK_SPARSE)
            o_ids, o_sc, o_cnt = oracle.recommend_batch(X, W.tocsr(), top_k=10, filter_interacted=filt)
            assert np.array_equal(cnt, o_cnt)
            assert np.array_equal(ids, o_ids)
            assert np.array_equal(bits(sc), bits(o_sc))


@pytest.mark.parametrize("f64", [False, True])
def test_score_signed_zero_and_denormal_ratings(oracle, f64):
    """Time decay turns old ratings into float32 denormals and negative ones into -0.0 (rtrec stores both).  A product
    of -0.0 -- a stored -0.0, or a negative denormal times a small coefficient -- must still count as the column's first
    contribution: the tiled kernel's accumulators start from a -0.0 marker, and (-0) + (-0) = -0 used to leave the column
    "untouched", so that its next contribution listed it a second time (duplicate / uninitialised ids in the answer;
    found by tools/fuzz_api.py)."""
    rng = np.random.default_rng(12)
    U, I = 120, 90
    rows_, cols_, vals_ = [], [], []
    specials = np.array([-0.0, 0.0, -1.4e-40, 1.4e-40, -3e-39, 2.5, -1.5, 1e-38, -1e-38], np.float32)
    for u in range(U):
        its = np.sort(rng.choice(I, size=int(rng.integers(1, 9)), replace=False))
        rows_ += [u] * len(its)
        cols_ += its.tolist()
        vals_ += rng.choice(specials, size=len(its)).tolist()
    X = sp.csr_matrix((np.array(vals_, np.float32), (rows_, cols_)), shape=(U, I))
    X.sort_indices()
    assert (X.data == 0).any() and np.signbit(X.data[X.data == 0]).any()      # the explicit zeros are still stored
    Wd = sp.random(I, I, density=0.12, random_state=3, format="csc", dtype=np.float32)
    Wd.data = (Wd.data * rng.choice(np.array([1.0, -1.0, 1e-6, -1e-6, 1e-3], np.float32), size=Wd.nnz)).astype(np.float32)
    Wd.setdiag(0)
    Wd.eliminate_zeros()
    Wd.sort_indices()
    Wn = Wd.copy()                                   # a second W with few non-empty rows: the feature-row kernel's shape
    keep = np.isin(Wn.indices, np.arange(0, I, 9))
    Wn.data[~keep] = 0
    Wn.eliminate_zeros()
    Wt = sp.csc_matrix(Wn[:, rng.integers(0, 12, size=I)])      # every column a copy of one of 12: exact ties everywhere, which
    Wt.sort_indices()                                            # the feature-row kernel orders itself unless a stored zero rating
    for W in (Wd, Wn, Wt):                                       # of the user may have touched a column first
        for feature_rows, rows in ((False, np.arange(U)), (False, np.arange(0, U, 7)), (True, np.arange(U))):
            if f64 and feature_rows:
                continue
            eng = SlimEngine(device="cuda:0", tile_cols=256)
            eng.use_feature_rows = feature_rows
            eng.set_interactions(None, X, need_csc=False)
            eng.set_weights(W.astype(np.float64) if f64 else W, acc_f64=f64)
            for filt in (False, True):
                ids, sc, cnt = eng.recommend_rows(rows, top_k=6, filter_interacted=filt, mode=_native.TOPK_SPARSE)
                o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows], W.tocsr(), top_k=6, filter_interacted=filt, use_f64=f64)
                assert np.array_equal(cnt, o_cnt)
                assert np.array_equal(ids, o_ids)
                assert np.array_equal(bits(sc), bits(o_sc))


@pytest.mark.parametrize("users_per_wave", [2, 4, 8])
def test_feature_row_kernel_users_per_wave_forms(oracle, users_per_wave):
    """score_frows_kernel<REGS, XR, UW>: the library picks 8, 4 or 2 users per wave from the batch size (small passes get
    more, shorter jobs); every form must give the same answer at any size -- forced here through rtrec_score_opts.diagnostics
    bits 8-11, on a float W, on a W of duplicated columns (ties inside the list and at its threshold) and for batches that
    leave waves partly or wholly without users."""
    X, W = make_model(oracle, U=900, I=500, draws=30000, K=12, seed=5)
    keep = np.isin(W.indices, np.arange(0, 500, 11))                  # few non-empty rows: the feature-row kernel's shape
    Wn = W.copy(); Wn.data[~keep] = 0; Wn.eliminate_zeros()
    rng = np.random.default_rng(2)
    Wt = sp.csc_matrix(Wn[:, rng.integers(0, 30, size=500)]); Wt.sort_indices()
    for Wx in (Wn, Wt):
        eng = SlimEngine(device="cuda:0", tile_cols=256)
        eng.fr_users_per_wave = users_per_wave
        eng.set_interactions(None, X, need_csc=False)
        eng.set_weights(Wx)
        for rows in (np.arange(X.shape[0]), np.arange(0, X.shape[0], 7), np.array([3]), np.arange(5, 22)):
            for filt in (True, False):
                ids, sc, cnt = eng.recommend_rows(rows, top_k=10, filter_interacted=filt, mode=_native.TOPK_SPARSE)
                o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows], Wx.tocsr(), top_k=10, filter_interacted=filt)
                assert np.array_equal(cnt, o_cnt)
                assert np.array_equal(ids, o_ids)
                assert np.array_equal(bits(sc), bits(o_sc))


def test_candidate_mode(oracle):
    X, W = make_model(oracle, U=400, I=300, draws=8000, K=10)
    eng = SlimEngine(device="cuda:0", tile_cols=256)
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    cands = [5, 17, 250, 3, 99, 100, 101, 42]
    rank = np.full(300, -1, np.int32)
    rank[cands] = np.arange(len(cands))
    rows = np.arange(50)
    ids, sc, cnt = eng.recommend_rows(rows, top_k=5, filter_interacted=True, mode=_native.TOPK_CANDIDATES, col_rank=rank)
    S = (X[rows] @ W.tocsr()[:, cands]).toarray().astype(np.float32)
    for r in range(len(rows)):
        order = sorted(range(len(cands)), key=lambda c: (-S[r, c], -c))[:5]     # stable-argsort tie rule
        assert ids[r].tolist() == [cands[c] for c in order]
        assert cnt[r] == 5


@pytest.mark.parametrize("f64", [False, True])
def test_candidate_mode_request_sized_calls_take_the_direct_kernel(oracle, f64):
    """A request (few rows x a candidate list) is ranked by rtrec_slim_score_candidates -- scores from W's CSC columns in the
    reference's summation order, no pass over all columns: ids, score bits and counts equal the scipy product ranked by the
    stable-argsort rule (later candidate first on ties, zeros compete, duplicates compete), and the tiled kernel's answer."""
    X, W = make_model(oracle, U=400, I=300, draws=8000, K=10)
    X = X.tolil()
    X[7, :] = 0                                            # an empty row: every score 0, candidates by position descending
    X = X.tocsr().astype(np.float32)
    X.eliminate_zeros(); X.sort_indices()
    eng = SlimEngine(device="cuda:0", tile_cols=256)
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W.astype(np.float64) if f64 else W, acc_f64=f64)
    rng = np.random.default_rng(2)
    Wr = W.tocsr()
    for cands, k, rows in (([5, 17, 250, 3, 99, 100, 101, 42], 5, np.arange(50)), (rng.permutation(300)[:200].tolist(), 10, np.array([7, 3, 11])),
                           ([9, 9, 40, 9], 4, np.arange(20)), (list(range(300)), 300, np.array([0])), ([123], 3, np.arange(5))):
        ids, sc, cnt = eng.recommend_rows(rows, top_k=min(k, len(cands)), mode=_native.TOPK_CANDIDATES, candidates=np.asarray(cands))
        assert eng.last_score_path == "candidates_direct"
        kk = min(k, len(cands))
        if f64:
            S = (X[rows].astype(np.float64) @ Wr.astype(np.float64)[:, cands]).toarray()
        else:
            S = (X[rows] @ Wr[:, cands]).toarray().astype(np.float32)
        for r in range(len(rows)):
            order = sorted(range(len(cands)), key=lambda c: (-S[r, c], -c))[:kk]
            assert ids[r].tolist() == [cands[c] for c in order], (cands[:8], r)
            assert np.array_equal(bits(sc[r]), bits(S[r, order].astype(np.float32)))
            assert cnt[r] == kk
        if len(set(cands)) == len(cands):                  # (the rank-array form keeps one position per item)
            eng.cands_direct = False
            t = eng.recommend_rows(rows, top_k=kk, mode=_native.TOPK_CANDIDATES, candidates=np.asarray(cands))
            eng.cands_direct = True
            assert eng.last_score_path != "candidates_direct"
            assert np.array_equal(t[0], ids) and np.array_equal(bits(t[1]), bits(sc)) and np.array_equal(t[2], cnt)
    got = eng.recommend_csr(X[[3, 7, 9]], top_k=3, mode=_native.TOPK_CANDIDATES, candidates=np.array([5, 17, 250, 3]))
    ref = eng.recommend_rows(np.array([3, 7, 9]), top_k=3, mode=_native.TOPK_CANDIDATES, candidates=np.array([5, 17, 250, 3]))
    assert all(np.array_equal(a.view(np.int32), b.view(np.int32)) for a, b in zip(got, ref))
    with pytest.raises(IndexError):
        eng.recommend_rows(np.array([1]), top_k=2, mode=_native.TOPK_CANDIDATES, candidates=np.array([5, 300]))


def test_candidate_mode_bulk_calls_take_the_direct_kernel_too(oracle):
    """A BULK call (rows x candidates far beyond a request: slim_elastic.py:722-739 over all users) with up to
    SlimEngine.CANDS_DIRECT_BULK candidates is ranked by the direct kernel as well; a longer list keeps the tiled kernel.  Same
    ids / score bits / counts as the scipy product ranked by the stable-argsort rule and as the tiled kernel."""
    X, W = make_model(oracle, U=9000, I=1500, draws=260000, K=20)
    eng = SlimEngine(device="cuda:0", tile_cols=256)
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    rng = np.random.default_rng(4)
    rows = np.arange(X.shape[0])
    cands = np.sort(rng.choice(1500, 400, replace=False))
    assert len(rows) * len(cands) > eng.CANDS_DIRECT_MAX_PAIRS and len(cands) <= eng.CANDS_DIRECT_BULK
    ids, sc, cnt = eng.recommend_rows(rows, top_k=10, mode=_native.TOPK_CANDIDATES, candidates=cands)
    assert eng.last_score_path == "candidates_direct"
    S = (X @ W.tocsr()[:, cands]).toarray().astype(np.float32)
    for r in rng.choice(len(rows), 300, replace=False):
        order = sorted(range(len(cands)), key=lambda c: (-S[r, c], -c))[:10]
        assert ids[r].tolist() == cands[order].tolist() and cnt[r] == 10
        assert np.array_equal(bits(sc[r]), bits(S[r, order]))
    eng.cands_direct = False
    t = eng.recommend_rows(rows, top_k=10, mode=_native.TOPK_CANDIDATES, candidates=cands)
    eng.cands_direct = True
    assert eng.last_score_path != "candidates_direct"
    assert np.array_equal(t[0], ids) and np.array_equal(bits(t[1]), bits(sc)) and np.array_equal(t[2], cnt)
    long_list = np.sort(rng.choice(1500, eng.CANDS_DIRECT_BULK + 100, replace=False))
    eng.recommend_rows(rows, top_k=10, mode=_native.TOPK_CANDIDATES, candidates=long_list)
    assert eng.last_score_path != "candidates_direct"


def test_similar_items(engine, oracle):
    X, W = make_model(oracle, U=800, I=300, draws=15000, K=30)
    engine.set_weights(W)
    q = np.arange(300)
    ids, sc, cnt = engine.similar_items(q, top_k=7)
    for j in q:
        oi, ov = oracle.similar_items(W, int(j), top_k=7)
        assert cnt[j] == len(oi)
        assert np.array_equal(ids[j, :cnt[j]], oi)
        assert np.array_equal(bits(sc[j, :cnt[j]]), bits(ov))


def test_merge_topk_equals_unsharded(oracle):
    """Column-sharded scoring + merge kernel == single-shard result (the multi-GPU data path,
    run here as two engines on one GPU)."""
    import torch
    X, W = make_model(oracle, U=500, I=640, draws=12000, K=15)
    rows = np.arange(0, 500, 2)
    full = SlimEngine(device="cuda:0", tile_cols=256)
    full.set_interactions(None, X, need_csc=False)
    full.set_weights(W)
    ids, sc, cnt = full.recommend_rows(rows, top_k=10)
    parts = []
    for r in range(2):
        e = SlimEngine(device="cuda:0", rank=r, world_size=2, tile_cols=256)
        e.world_size_for_merge = 2
        e.set_interactions(None, X, need_csc=False)
        e.set_weights(W)
        d_rows = e.be.to_dev(rows.astype(np.int32))
        xb = (e._X["rptr"], e._X["rcol"], e._X["rval"])
        parts.append(e._local_topk(d_rows, len(rows), xb, 10, True, _native.TOPK_SPARSE, None))
    be = full.be
    g = [torch.stack([p[k] for p in parts]).contiguous() for k in (0, 1, 3, 4)]
    o_ids = be.empty((len(rows), 10), torch.int32)
    o_sc = be.empty((len(rows), 10), torch.float32)
    o_cnt = be.empty((len(rows),), torch.int32)
    _native.check(be.lib.rtrec_slim_merge_topk(len(rows), 2, 10, be.ptr(g[0]), be.ptr(g[1]), None, be.ptr(g[2]),
                                               be.ptr(g[3]), be.ptr(o_ids), be.ptr(o_sc), be.ptr(o_cnt), be.stream()),
                  "merge")
    assert np.array_equal(o_ids.cpu().numpy(), ids)
    assert np.array_equal(bits(o_sc.cpu().numpy()), bits(sc))
    assert np.array_equal(o_cnt.cpu().numpy(), cnt)


@pytest.mark.parametrize("shape", ["feature_rows", "general"])
def test_column_shards_order_cross_shard_ties_like_the_reference(oracle, shape):
    """Columns of DIFFERENT shards with bit-equal scores (here: every column of the second half of W is a copy of one in the
    first half, integer ratings): neither shard sees a tie, so neither would compute the reference's tie key (position of the
    first touching item in the user's row, slim_elastic.py:782-818 over scipy's product order) -- a rank that holds part of
    the columns completes it for every entry (rtrec_slim_first_touch_aux) and the merged lists equal the oracle's."""
    import torch
    rng = np.random.default_rng(12)
    I, U, half = 800, 900, 400
    R = 60 if shape == "feature_rows" else 300
    rows_w = np.sort(rng.choice(I, R, replace=False))
    nnz = 4000 if shape == "feature_rows" else 2500
    r, c = rng.choice(rows_w, nnz), rng.integers(0, half, nnz)
    v = rng.integers(1, 4, nnz).astype(np.float32)
    A = sp.csc_matrix((v, (r, c)), shape=(I, half), dtype=np.float32)
    A.sum_duplicates()
    W = sp.hstack([A, A], format="csc").astype(np.float32)          # column j + 400 == column j
    W.sort_indices()
    ur = np.repeat(np.arange(U), rng.integers(1, 40, U))
    ui = np.where(rng.random(len(ur)) < 0.6, rng.choice(rows_w, len(ur)), rng.integers(0, I, len(ur)))
    X = sp.csr_matrix((rng.integers(1, 6, len(ur)).astype(np.float32), (ur, ui)), shape=(U, I), dtype=np.float32)
    X.sum_duplicates(); X.sort_indices()
    rows = np.arange(U, dtype=np.int32)
    for k, filt in ((10, True), (3, False)):
        o_ids, o_sc, o_cnt = oracle.recommend_batch(X, W.tocsr(), top_k=k, filter_interacted=filt)
        parts = []
        for rank in range(2):
            e = SlimEngine(device="cuda:0", rank=rank, world_size=2)
            e.set_interactions(None, X, need_csc=False)
            e.set_weights(W)
            xb = (e._X["rptr"], e._X["rcol"], e._X["rval"])
            parts.append(e._local_topk(e.be.to_dev(rows), U, xb, k, filt, _native.TOPK_SPARSE, None))
        be = e.be
        g = [torch.stack([p[j] for p in parts]).contiguous() for j in (0, 1, 3, 4)]
        m_ids, m_sc, m_cnt = be.empty((U, k), torch.int32), be.empty((U, k), torch.float32), be.empty((U,), torch.int32)
        _native.check(be.lib.rtrec_slim_merge_topk(U, 2, k, be.ptr(g[0]), be.ptr(g[1]), None, be.ptr(g[2]), be.ptr(g[3]),
                                                   be.ptr(m_ids), be.ptr(m_sc), be.ptr(m_cnt), be.stream()), "merge")
        assert np.array_equal(m_cnt.cpu().numpy(), o_cnt)
        bad = np.flatnonzero((m_ids.cpu().numpy() != o_ids).any(axis=1))
        assert bad.size == 0, f"{bad.size} rows differ, first {bad[:5]}: {m_ids.cpu().numpy()[bad[0]]} vs {o_ids[bad[0]]}"
        assert np.array_equal(bits(m_sc.cpu().numpy()), bits(o_sc))


@pytest.mark.parametrize("tile_cols", [256, 1024, 4096])     # 256: the engine widens the tiles for large k
@pytest.mark.parametrize("top_k", [1, 10, 50, 64, 200])    # > 63: the successive-scan selection
def test_score_wide_catalogue_heavy_users(oracle, tile_cols, top_k):
    """Users that touch more columns than the touched list holds (> 1024 per tile: the full-tile scan /
    reset path), several tiles, dense W row blocks next to sparse ones, k from 1 to 50."""
    rng = np.random.default_rng(4)
    I, U = 5000, 260
    # W: 40 "popular" rows that reach most columns (dense blocks), the rest sparse
    rows, cols, vals = [], [], []
    for i in rng.choice(I, 40, replace=False):
        c = rng.choice(I, 3500, replace=False)
        rows.append(np.full(len(c), i)); cols.append(c); vals.append(rng.random(len(c)).astype(np.float32) + 0.01)
    r2 = rng.integers(0, I, 30000); c2 = rng.integers(0, I, 30000)
    rows.append(r2); cols.append(c2); vals.append(rng.random(30000).astype(np.float32) + 0.01)
    W = sp.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(I, I), dtype=np.float32)
    W.sum_duplicates(); W.sort_indices()
    # users: light (5 items), medium (80), heavy (900 items incl. the popular rows)
    n_items = np.concatenate([np.full(100, 5), np.full(100, 80), np.full(60, 900)])
    ur = np.repeat(np.arange(U), n_items)
    uc = np.concatenate([rng.choice(I, n, replace=False) for n in n_items])
    X = sp.csr_matrix(((rng.integers(1, 6, len(ur)) * np.exp(-rng.random(len(ur)))).astype(np.float32), (ur, uc)),
                      shape=(U, I), dtype=np.float32)
    X.sort_indices()
    Wr = W.tocsr()
    for world, rank in ((1, 0), (3, 1)):
        eng = SlimEngine(device="cuda:0", tile_cols=tile_cols, rank=rank, world_size=world)
        eng.set_interactions(None, X, need_csc=False)
        eng.set_weights(W)
        for filt in (True, False):
            d_rows = eng.be.to_dev(np.arange(U, dtype=np.int32))
            xb = (eng._X["rptr"], eng._X["rcol"], eng._X["rval"])
            ids, sc, sc64, aux, cnt = eng._local_topk(d_rows, U, xb, top_k, filt, _native.TOPK_SPARSE, None)
            lo, hi = eng._W["col_lo"], eng._W["col_hi"]
            Wshard = sp.csr_matrix(Wr.shape, dtype=np.float32).tolil()
            Wshard = sp.hstack([sp.csr_matrix((I, lo), dtype=np.float32), Wr[:, lo:hi],
                                sp.csr_matrix((I, I - hi), dtype=np.float32)]).tocsr()
            o_ids, o_sc, o_cnt = oracle.recommend_batch(X, Wshard, top_k=top_k, filter_interacted=filt)
            assert np.array_equal(cnt.cpu().numpy(), o_cnt)
            assert np.array_equal(ids.cpu().numpy(), o_ids)
            assert np.array_equal(bits(sc.cpu().numpy()), bits(o_sc))


@pytest.mark.parametrize("U,I,draws,K,positive,float_ratings", [
    (600, 200, 12000, 8, True, True),
    (600, 200, 12000, 8, False, True),
    (3000, 800, 90000, 50, True, True),
    (3000, 800, 90000, 50, True, False),
    (5000, 300, 400000, 50, True, True),      # dense catalogue: every coordinate of the popular targets is non-zero
])
@pytest.mark.parametrize("gram_items", ["512", "64"])        # 64: most features are NOT in the Gram matrix
@pytest.mark.parametrize("screen_min", [None, "1000000000"])  # the latter: Gram tracking is the only screen
def test_fit_gram_tracking_bit_exact(engine, oracle, U, I, draws, K, positive, float_ratings, gram_items, screen_min,
                                     monkeypatch):
    """Gram tracking (zero coordinates decided from D_p -= dw G_pq with rounding bounds instead of a pass
    over memory) must leave coefficients and sweep counts bit-identical to the oracle."""
    monkeypatch.setenv("RTREC_AMD_FIT_MODE", "sw")
    monkeypatch.setenv("RTREC_AMD_GRAM", "force")
    monkeypatch.setenv("RTREC_AMD_GRAM_ITEMS", gram_items)
    if screen_min is not None:
        monkeypatch.setenv("RTREC_AMD_SCREEN_MIN", screen_min)
    X = interaction_matrix(U, I, draws, seed=23, float_ratings=float_ratings)
    Xc = X.tocsc()
    Xc.sort_indices()
    engine.set_interactions(Xc, X)
    tg, items, coef, count, n_iter = engine.fit_columns(np.arange(I), positive=positive, nn_feature_selection=K)
    assert engine._X.get("gram") is not None
    ptr, idx, val, nit = oracle.fit_columns(Xc, tg, positive=positive, nn_feature_selection=K)
    assert np.array_equal(n_iter, nit), f"n_iter differs on {np.flatnonzero(n_iter != nit)[:10]}"
    for t in range(len(tg)):
        c = count[t]
        o = np.argsort(items[t, :c], kind="stable")
        assert np.array_equal(items[t, :c][o], idx[ptr[t]:ptr[t + 1]]), f"feature set differs for column {tg[t]}"
        assert np.array_equal(bits(coef[t, :c][o]), bits(val[ptr[t]:ptr[t + 1]])), f"coefficient bits differ for column {tg[t]}"


def test_gram_matrix_kernel(engine):
    """rtrec_slim_gram_matrix vs numpy float64 (exact products, ~1e-16 relative sums)."""
    X = interaction_matrix(9000, 700, 300000, seed=31)
    Xc = X.tocsc()
    Xc.sort_indices()
    engine.set_interactions(Xc, X)
    for n_top in (700, 100, 64):
        g = engine.be.gram_matrix(engine._X, 9000, 700, n_top)
        P, p64 = len(g["items"]), g["n"]
        assert p64 % 64 == 0 and p64 >= P
        G = g["G"].cpu().numpy()
        XP = Xc[:, g["items"]].toarray().astype(np.float64)
        ref = XP.T @ XP
        assert np.allclose(G[:P, :P], ref, rtol=1e-12, atol=0)
        assert not G[P:, :].any() and not G[:, P:].any()
        idx = g["index"].cpu().numpy()
        assert np.array_equal(np.flatnonzero(idx >= 0), np.sort(g["items"])) and np.array_equal(idx[g["items"]], np.arange(P))


def test_torch_custom_ops_direct_use(oracle):
    """torch.ops.rtrec_amd.* (rtrec_amd/ops.py) called directly by a holder of device tensors:
    schema / mutation annotations (opcheck) and results."""
    import torch
    from rtrec_amd import ops  # noqa: F401
    X = interaction_matrix(300, 90, 4000, seed=8)
    Xc = X.tocsc()
    Xc.sort_indices()
    dev = torch.device("cuda:0")
    cptr = torch.from_numpy(Xc.indptr.astype(np.int32)).to(dev)
    cval = torch.from_numpy(Xc.data.astype(np.float32)).to(dev)
    out = torch.empty(90, dtype=torch.float32, device=dev)
    torch.ops.rtrec_amd.column_sqnorms(cptr, cval, out)
    ref = np.zeros(90, np.float32)
    for c in range(90):
        acc = np.float32(0)
        for v in Xc.data[Xc.indptr[c]:Xc.indptr[c + 1]]:
            acc = np.float32(acc + np.float32(v * v))
        ref[c] = acc
    assert np.array_equal(bits(out.cpu().numpy()), bits(ref))
    torch.library.opcheck(torch.ops.rtrec_amd.column_sqnorms.default, (cptr, cval, out), test_utils=("test_schema",))
    # similar_topk on a small W
    W, _ = oracle_w(oracle, Xc, np.arange(90), nn_feature_selection=10)
    wp = torch.from_numpy(W.indptr.astype(np.int32)).to(dev)
    wr = torch.from_numpy(W.indices.astype(np.int32)).to(dev)
    wv = torch.from_numpy(W.data.astype(np.float32)).to(dev)
    q = torch.arange(90, dtype=torch.int32, device=dev)
    ids = torch.empty((90, 5), dtype=torch.int32, device=dev)
    sc = torch.empty((90, 5), dtype=torch.float32, device=dev)
    cnt = torch.empty(90, dtype=torch.int32, device=dev)
    torch.ops.rtrec_amd.similar_topk(q, wp, wr, wv, 5, ids, sc, cnt)
    for j in range(90):
        oi, ov = oracle.similar_items(W, j, top_k=5)
        n = int(cnt[j])
        assert n == len(oi) and np.array_equal(ids[j, :n].cpu().numpy(), oi)


@pytest.mark.parametrize("fit_mode", ["sw", "mw"])
@pytest.mark.parametrize("positive", [True, False])
@pytest.mark.parametrize("K", [12, None])
def test_fit_with_negative_ratings_bit_exact(engine, oracle, fit_mode, positive, K, monkeypatch):
    """The reference's default store clips to [-5, 10]: X may hold negative values (dislikes).  Gram
    tracking must switch itself off, screening bounds use |products|, signs of zeros follow sklearn."""
    monkeypatch.setenv("RTREC_AMD_FIT_MODE", fit_mode)
    monkeypatch.setenv("RTREC_AMD_SCREEN_MIN", "1")
    monkeypatch.setenv("RTREC_AMD_GRAM", "force")
    monkeypatch.setenv("RTREC_AMD_XTY_BATCH", "force")      # latency kernel: one-pass X^T y with products of either sign
    X = interaction_matrix(1500, 300, 40000, seed=77)
    rng = np.random.default_rng(5)
    X.data = (X.data * np.where(rng.random(X.nnz) < 0.25, -1.0, 1.0)).astype(np.float32)
    Xc = X.tocsc()
    Xc.sort_indices()
    engine.set_interactions(Xc, X)
    assert engine._X["nonneg"] is False
    tg, items, coef, count, n_iter = engine.fit_columns(np.arange(300), positive=positive, nn_feature_selection=K)
    assert engine._X.get("gram") is None
    ptr, idx, val, nit = oracle.fit_columns(Xc, tg, positive=positive, nn_feature_selection=K)
    assert np.array_equal(n_iter, nit)
    assert np.array_equal(count, np.diff(ptr))
    for t in range(len(tg)):
        c = count[t]
        o = np.argsort(items[t, :c], kind="stable")
        assert np.array_equal(items[t, :c][o], idx[ptr[t]:ptr[t + 1]])
        assert np.array_equal(bits(coef[t, :c][o]), bits(val[ptr[t]:ptr[t + 1]]))


@pytest.mark.parametrize("lane_max", [None, "16", "0"])      # 16 / 0: most / all columns take the wave-wide path
@pytest.mark.parametrize("screen_min", [None, "1"])
@pytest.mark.parametrize("positive,float_ratings", [(True, True), (False, True), (True, False)])
def test_fit_every_item_path_bit_exact(engine, oracle, lane_max, screen_min, positive, float_ratings, monkeypatch):
    """nn_feature_selection=None (the reference's default): draws are taken 64 at a time and every lane
    folds its own column in order; committing them one by one must equal the sequential loop."""
    if lane_max is not None:
        monkeypatch.setenv("RTREC_AMD_LANE_MAX", lane_max)
    if screen_min is not None:
        monkeypatch.setenv("RTREC_AMD_SCREEN_MIN", screen_min)
    X = interaction_matrix(2500, 260, 70000, seed=41, float_ratings=float_ratings)
    Xc = X.tocsc()
    Xc.sort_indices()
    engine.set_interactions(Xc, X)
    tg, items, coef, count, n_iter = engine.fit_columns(np.arange(260), positive=positive, nn_feature_selection=None)
    ptr, idx, val, nit = oracle.fit_columns(Xc, tg, positive=positive, nn_feature_selection=None)
    assert np.array_equal(n_iter, nit), f"n_iter differs on {np.flatnonzero(n_iter != nit)[:10]}"
    assert np.array_equal(count, np.diff(ptr))
    for t in range(len(tg)):
        c = count[t]
        assert np.array_equal(items[t, :c], idx[ptr[t]:ptr[t + 1]]), f"support differs for column {tg[t]}"
        assert np.array_equal(bits(coef[t, :c]), bits(val[ptr[t]:ptr[t + 1]])), f"coefficient bits differ for column {tg[t]}"


def test_fit_every_item_output_overflow_is_refitted(engine, oracle, monkeypatch):
    """K=None output blocks hold ALLF_OUTPUT_CAP coefficients per target; a target with more is refitted
    with room for all items, so the result does not depend on the cap."""
    monkeypatch.setenv("RTREC_AMD_ALLF_CAP", "3")
    X = interaction_matrix(2500, 260, 70000, seed=41)
    Xc = X.tocsc()
    Xc.sort_indices()
    engine.set_interactions(Xc, X)
    tg, items, coef, count, n_iter = engine.fit_columns(np.arange(260), nn_feature_selection=None)
    ptr, idx, val, nit = oracle.fit_columns(Xc, tg, nn_feature_selection=None)
    assert count.max() > 3, "the scenario must overflow the 3-entry cap"
    assert np.array_equal(n_iter, nit) and np.array_equal(count, np.diff(ptr))
    for t in range(len(tg)):
        c = count[t]
        assert np.array_equal(items[t, :c], idx[ptr[t]:ptr[t + 1]])
        assert np.array_equal(bits(coef[t, :c]), bits(val[ptr[t]:ptr[t + 1]]))


@pytest.mark.parametrize("R,n_cols,per_col,tile_cols", [(5, 40, 3, 256), (64, 700, 12, 256), (66, 3000, 20, 256), (100, 900, 25, 256),
                                                         (128, 2600, 30, 256), (30, 9000, 6, 256), (128, 2600, 30, 128),
                                                         (40, 3000, 10, 128)])
@pytest.mark.parametrize("integer_ratings", [False, True])
def test_feature_row_kernel_shapes(oracle, R, n_cols, per_col, tile_cols, integer_ratings):
    """score_frows_kernel over the shapes it specialises on: <= 64 / 65-66 / <= 128 rows of W (one or two
    rating registers, 256- or 128-column tiles), one tile to dozens (several super-tiles), negative ratings,
    integer ratings (exact score ties -> the exact-tie pass), users without any feature item, empty users,
    row subsets in any order, top_k 1 .. 15, with and without the interacted filter -- against the oracle,
    and against the tiled-CSR kernel."""
    rng = np.random.default_rng(R * 1000 + n_cols)
    I, U = max(n_cols + 200, 400), 700
    feat = np.sort(rng.choice(I, R, replace=False))
    cols = np.sort(rng.choice(I, n_cols, replace=False))
    pop = 1.0 / np.arange(1, R + 1) ** 0.9
    rr, cc = [], []
    for c in cols:
        k = min(R, max(1, int(rng.integers(1, per_col + 1))))
        rr.append(rng.choice(feat, k, replace=False, p=pop / pop.sum()))
        cc.append(np.full(k, c))
    rr, cc = np.concatenate(rr), np.concatenate(cc)
    vals = (rng.random(len(rr)).astype(np.float32) + 0.05) * np.where(rng.random(len(rr)) < 0.1, -1, 1).astype(np.float32)
    if integer_ratings:
        vals = np.round(vals * 4).astype(np.float32)
        vals[vals == 0] = 1.0
    W = sp.csc_matrix((vals, (rr, cc)), shape=(I, I), dtype=np.float32)
    W.sum_duplicates(); W.eliminate_zeros(); W.sort_indices()
    n_items = rng.integers(0, 60, U)
    n_items[:20] = 0                                             # users without interactions
    ur = np.repeat(np.arange(U), n_items)
    # half of every user's items are feature items, the rest anything (incl. scored columns -> the filter matters)
    ui = np.where(rng.random(len(ur)) < 0.5, rng.choice(feat, len(ur)), rng.integers(0, I, len(ur)))
    xv = rng.integers(1, 6, len(ur)).astype(np.float32) if integer_ratings else (rng.random(len(ur)).astype(np.float32) * 5 - 0.5)
    X = sp.csr_matrix((xv, (ur, ui)), shape=(U, I), dtype=np.float32)
    X.sum_duplicates(); X.eliminate_zeros(); X.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.FR_TILE_COLS = tile_cols                                 # 128: the narrow-tile kernels
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    lay = eng._layout(True)
    assert lay.get("fr_w") is not None and lay["fr_rows"] == len(np.unique(W.tocoo().row)) and lay["fr_tile_cols"] == tile_cols
    Wr = W.tocsr()
    for rows in (np.arange(U), rng.permutation(U)[:333]):
        for top_k, filt in ((10, True), (1, True), (15, False), (7, False)):
            ids, sc, cnt = eng.recommend_rows(rows, top_k=top_k, filter_interacted=filt)
            o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows], Wr, top_k=top_k, filter_interacted=filt)
            assert np.array_equal(cnt, o_cnt) and np.array_equal(ids, o_ids) and np.array_equal(bits(sc), bits(o_sc)), \
                (len(rows), top_k, filt)
    eng.use_feature_rows = False
    ids2, sc2, cnt2 = eng.recommend_rows(np.arange(U), top_k=10)
    eng.use_feature_rows = True
    ids1, sc1, cnt1 = eng.recommend_rows(np.arange(U), top_k=10)
    assert np.array_equal(ids1, ids2) and np.array_equal(bits(sc1), bits(sc2)) and np.array_equal(cnt1, cnt2)
    # a batch large enough for the longest-first work order (ROW_ORDER_MIN) and several jobs per workgroup
    big = rng.integers(0, U, 5000)
    ids3, sc3, cnt3 = eng.recommend_rows(big, top_k=10)
    assert np.array_equal(ids3, ids1[big]) and np.array_equal(bits(sc3), bits(sc1[big])) and np.array_equal(cnt3, cnt1[big])


@pytest.mark.parametrize("mode,gram_items", [("shuffle", "512"), ("gram", "512"), ("gram", "16")])   # 16: most targets miss a feature in the Gram matrix
@pytest.mark.parametrize("U,I,draws,K,positive,float_ratings", [
    (5000, 300, 400000, 50, True, True),
    (3000, 800, 150000, 20, True, False),
    (2500, 400, 120000, 30, False, True),
])
def test_fit_tolerance_modes_track_the_exact_solution(engine, oracle, U, I, draws, K, positive, float_ratings, mode,
                                                      gram_items, monkeypatch):
    """mode="shuffle" / "gram" (rtrec_fit_opts.fast = 1 / 2): tree-reduced dots, Gram-form CD.  Same features as
    the exact mode on every target; where the sweep count agrees with scikit-learn's (nearly everywhere) the
    coefficients agree to 1e-5 (shuffle) / 1e-4 (gram: the float32 reference itself carries ~1e-5 of accumulated
    residual rounding) of the column's largest; a stopping test that flips by a rounding costs a target one sweep,
    i.e. a change of the order of the solver's own tolerance.  Scores and top-10 lists follow."""
    monkeypatch.setenv("RTREC_AMD_GRAM_ITEMS", gram_items)
    X = interaction_matrix(U, I, draws, seed=21, float_ratings=float_ratings)
    Xc = X.tocsc()
    Xc.sort_indices()
    engine.set_interactions(Xc, X)
    tg, items, coef, count, n_iter = engine.fit_columns(np.arange(I), positive=positive, nn_feature_selection=K, mode=mode)
    ptr, idx, val, nit = oracle.fit_columns(Xc, tg, positive=positive, nn_feature_selection=K)
    tol_same = 1e-5 if mode == "shuffle" else 1e-4
    same_sweeps = 0
    for n in range(len(tg)):
        c = count[n]
        o = np.argsort(items[n, :c], kind="stable")
        assert np.array_equal(items[n, :c][o], idx[ptr[n]:ptr[n + 1]]), f"column {tg[n]}: the selected features must be the exact ones"
        ref, got = val[ptr[n]:ptr[n + 1]].astype(np.float64), coef[n, :c][o].astype(np.float64)
        scale = max(np.abs(ref).max(), 1e-30) if len(ref) else 1.0
        if n_iter[n] == nit[n]:
            same_sweeps += 1
            assert np.abs(got - ref).max() <= tol_same * scale, f"column {tg[n]}"
        else:
            assert abs(int(n_iter[n]) - int(nit[n])) <= 2 and np.abs(got - ref).max() <= 5e-3 * scale, f"column {tg[n]}"
    assert same_sweeps >= 0.93 * len(tg)
    W_fast = merge_coefficients(None, I, *coefficients_to_updates(tg, items, coef, count))
    W_ref = merge_coefficients(None, I, idx.astype(np.int64), np.repeat(tg, np.diff(ptr)), val)
    S_fast, S_ref = (X @ W_fast).toarray().astype(np.float64), (X @ W_ref).toarray().astype(np.float64)
    agree = np.zeros(I, bool)
    agree[tg[n_iter == nit]] = True
    assert np.abs(S_fast - S_ref)[:, agree].max() <= tol_same * np.abs(S_ref).max()
    assert np.abs(S_fast - S_ref).max() <= 5e-3 * np.abs(S_ref).max()
    engine.set_weights(W_fast)
    ids, sc, cnt = engine.recommend_rows(np.arange(U), top_k=10)
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X, W_ref.tocsr(), top_k=11)
    # top-10 ids: identical wherever every gap between neighbouring scores of the reference's top-11 exceeds the
    # tolerance (5e-3 of the top score: the bound asserted on the scores above)
    with np.errstate(invalid="ignore"):
        clear = (o_cnt >= 11) & np.all((o_sc[:, :10] - o_sc[:, 1:11]) > 1e-2 * np.abs(o_sc[:, :1]), axis=1)
    assert clear.sum() >= 20
    assert np.array_equal(ids[clear], o_ids[clear, :10])
    # ... and nearly everywhere in practice
    full = o_cnt >= 10
    assert (ids[full] == o_ids[full, :10]).all(axis=1).mean() > 0.97


def test_score_path_randomised_parity():
    """tools/fuzz_score.py: random numbers of rows of W / columns / densities / users / top_k, filter on and off, integer,
    float and negative values, both tile widths, with and without the small-batch threshold -- ids, score bits and counts
    against the oracle (1,000 configurations were run clean when the feature-row kernel got its tile test; 25 run here)."""
    from tools.fuzz_score import run
    messages = []
    assert run(25, seed=4, log=messages.append) == 0, messages


def test_fit_path_randomised_parity():
    """tools/fuzz_fit.py: random matrix shapes, densities, K (incl. every item), sign constraint, negative ratings, screening
    thresholds, through the throughput kernel, the latency kernel and the latency kernel with the one-pass X^T y --
    coefficient bits, feature sets and sweep counts against the oracle (3,000 configurations were run clean; 40 run here)."""
    from tools.fuzz_fit import run
    messages = []
    assert run(40, seed=5, log=messages.append) == 0, messages


@pytest.mark.parametrize("case", range(7))
def test_optim_sgd_on_the_gpu_matches_the_reference(case):
    """optim="sgd" (slim_elastic.py:209-222: scikit-learn SGDRegressor behind FeatureSelectionWrapper) on the device
    (csrc/fit_sgd.hip): W of the serial fit and SGDRegressor.n_iter_ of every column equal the real reference's
    (tests/golden/sgd.json), K up to 130 (one, two and four features per wave lane), resets of the weight scale, max_iter
    reached, another seed / l1_ratio / tol."""
    import json
    import os
    from rtrec_amd.models.internal.slim_elastic import SLIMElastic
    from rtrec_amd.synth import interaction_matrix
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sgd.json")))
    c = g["cases"][case]
    X = interaction_matrix(c["U"], c["I"], c["draws"], seed=c["seed"]).tocsc()
    X.sort_indices()

    def golden(prefix):
        b = np.asarray(c[f"{prefix}_bits"], dtype=np.uint32)
        return sp.csc_matrix((b.view(np.float32), np.asarray(c[f"{prefix}_indices"]), np.asarray(c[f"{prefix}_indptr"])),
                             shape=(c["I"], c["I"]))
    m = SLIMElastic(dict(c["cfg"], optim="sgd"), engine=SlimEngine(device="cuda:0"))
    m.fit(X.copy())
    W, Wg = m.item_similarity.tocsc(), golden("W")
    W.sort_indices()
    assert W.dtype == np.float64
    targets = m.engine.last_fit_targets
    assert np.array_equal(np.sort(targets), np.arange(c["I"]))
    got_iter = np.empty(c["I"], dtype=np.int64)
    got_iter[targets] = m.n_iter_
    assert got_iter.tolist() == c["n_iter"], "epochs differ from SGDRegressor.n_iter_"
    assert np.array_equal(W.indptr, Wg.indptr) and np.array_equal(W.indices, Wg.indices)
    assert np.array_equal(W.data.astype(np.float32).view(np.uint32), Wg.data.view(np.uint32)), "coef_ bits differ"
    if "partial_items" in c:
        m.partial_fit_items(X.copy(), c["partial_items"])
        W2, W2g = m.item_similarity.tocsc(), golden("W2")
        W2.sort_indices()
        assert np.array_equal(W2.indptr, W2g.indptr) and np.array_equal(W2.indices, W2g.indices)
        assert np.array_equal(W2.data.astype(np.float32).view(np.uint32), W2g.data.view(np.uint32))
    if case == 0:
        bad = SLIMElastic({"optim": "sgd"}, engine=m.engine)
        with pytest.raises(AttributeError, match="sparse_coef_"):
            bad.fit(X.copy())



# ---- the ordered fold itself (csrc/fold_spec.hip.h): integer prefix sums inside a binade == the chain of float additions ----
def _fold_streams(rng):
    """Entry streams that exercise every branch of fold256_spec: drifting and zero-mean sums, integers, exact ties against a
    large running sum (half-integers of its ulp), cancellations to +0, wide dynamic range, infinities, tiny values, and short
    / ragged lengths around the 256-entry group."""
    out = []
    for n in (1, 3, 4, 63, 64, 65, 255, 256, 257, 511, 1000, 4099, 20011):
        out.append((rng.random(n, dtype=np.float32) * 3).astype(np.float32))
        out.append(((rng.random(n, dtype=np.float32) - 0.5) * 3).astype(np.float32))
        out.append(rng.integers(-10, 30, n).astype(np.float32))
        p = (0.5 * rng.integers(-2, 7, n)).astype(np.float32); p[0] = np.float32(2.0 ** 24 * (1 + rng.random())); out.append(p)
        p = (0.5 * rng.integers(-6, 3, n)).astype(np.float32); p[0] = np.float32(-2.0 ** 25 * (1 + rng.random())); out.append(p)
        x = (rng.random(n, dtype=np.float32) * 10).astype(np.float32); p = np.empty(n, np.float32); p[0::2] = x[0::2]; p[1::2] = -x[0::2][: len(p[1::2])]; out.append(p)
        e = rng.integers(100, 160, n).astype(np.uint32); m = rng.integers(0, 1 << 23, n).astype(np.uint32); sg = rng.integers(0, 2, n).astype(np.uint32)
        out.append(((sg << 31) | (e << 23) | m).view(np.float32))
        p = (rng.random(n, dtype=np.float32) * 1e30).astype(np.float32); p[n // 2] = np.inf; out.append(p)
        out.append((rng.random(n, dtype=np.float32) * 1e-40).astype(np.float32))              # subnormal sums
        p = (rng.random(n, dtype=np.float32)).astype(np.float32); p[: n // 3] = 0.0; out.append(p)   # leading zeros
        # multiples of the running sum's half ulp: ties on most entries
        run = np.float32(1000.0 * (1 + rng.random())); p = np.empty(n, np.float32); p[0] = run
        for i in range(1, n):
            hu = np.float32(2.0 ** (int(np.floor(np.log2(abs(float(run))))) - 24)) if run != 0 else np.float32(1.0)
            p[i] = hu * np.float32(rng.integers(-2, 5)); run = np.float32(run + p[i])
        out.append(p)
    return out


def test_ordered_sums_speculative_fold_equals_the_chain(oracle):
    import torch
    lib = _native.load()
    rng = np.random.default_rng(2025)
    streams = _fold_streams(rng)
    off = np.zeros(len(streams) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in streams])
    vals = torch.from_numpy(np.concatenate(streams)).cuda()
    d_off = torch.from_numpy(off).cuda()
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for mode in (0, 1, 2, 3):
        o = torch.empty(len(streams), dtype=torch.float32, device="cuda")
        _native.check(lib.rtrec_slim_ordered_sums(vals.data_ptr(), d_off.data_ptr(), len(streams), mode, o.data_ptr(), st), "ordered_sums")
        outs.append(o.cpu().numpy())
    ref = np.array([oracle.fold_sequential(s) for s in streams], dtype=np.float32)
    model = np.array([oracle.fold_speculative(s)[0] for s in streams], dtype=np.float32)
    nan = np.isnan(ref)
    for name, got in (("device chain", outs[1]), ("device speculative", outs[0]), ("device speculative, 2 groups per step", outs[2]),
                      ("device speculative, 4 groups per step", outs[3]), ("cpu model", model)):
        assert np.array_equal(np.isnan(got), nan), name
        bad = np.flatnonzero((bits(got) != bits(ref)) & ~nan)
        assert bad.size == 0, f"{name}: streams {bad[:10]} differ (lengths {[len(streams[b]) for b in bad[:10]]})"


@pytest.mark.parametrize("U,I,draws,K,positive,float_ratings", [
    (3000, 800, 90000, 50, True, True),
    (3000, 800, 90000, 50, True, False),
    (3000, 800, 90000, 50, False, True),
    (600, 200, 12000, None, True, True),
    (20000, 300, 700000, 20, True, True),      # long columns (2-10k entries): many 256-entry groups per fold, binade changes mid-column
])
@pytest.mark.parametrize("fit_mode", ["sw", "mw"])
@pytest.mark.parametrize("fold", ["spec-all", "chain"])
def test_fit_columns_fold_forms_bit_exact(engine, oracle, U, I, draws, K, positive, float_ratings, fit_mode, fold, monkeypatch):
    """rtrec_fit_opts.fold: the binade-speculative fold forced on every column of >= 64 entries, and the literal chain (the
    default of every other fit test), give the oracle's coefficient bits and sweep counts in both kernels."""
    monkeypatch.setenv("RTREC_AMD_FIT_MODE", fit_mode)
    monkeypatch.setenv("RTREC_AMD_FOLD", fold)
    monkeypatch.setenv("RTREC_AMD_XTY_BATCH", "0")
    X = interaction_matrix(U, I, draws, seed=23, float_ratings=float_ratings)
    Xc = X.tocsc()
    Xc.sort_indices()
    engine.set_interactions(Xc, X)
    cols = np.arange(I) if U < 20000 else np.argsort(-np.diff(Xc.indptr))[:48]
    tg, items, coef, count, n_iter = engine.fit_columns(cols, positive=positive, nn_feature_selection=K)
    ptr, idx, val, nit = oracle.fit_columns(Xc, tg, positive=positive, nn_feature_selection=K)
    assert np.array_equal(n_iter, nit), f"n_iter differs on {np.flatnonzero(n_iter != nit)[:10]}"
    assert np.array_equal(count, np.diff(ptr))
    for t in range(len(tg)):
        c = count[t]
        o_i, o_v = idx[ptr[t]:ptr[t + 1]], val[ptr[t]:ptr[t + 1]]
        if K is None:
            assert np.array_equal(items[t, :c], o_i) and np.array_equal(bits(coef[t, :c]), bits(o_v)), f"target {tg[t]}"
        else:
            a = np.argsort(items[t, :c], kind="stable"); b = np.argsort(o_i, kind="stable")
            assert np.array_equal(items[t, :c][a], o_i[b]) and np.array_equal(bits(coef[t, :c][a]), bits(o_v[b])), f"target {tg[t]}"
