"""GPU parity of the general-W scoring path (csrc/score_seg.hip.h, rtrec_amd/seg_layout.py) against the CPU oracle.

Bar: ids, score bits and counts identical to oracle.recommend_batch (the restatement of slim_elastic.py:674-741 +
:782-818) for every W -- many rows, negative weights, exact ties, users too long for the kernel's LDS lists, wide
catalogues (tiles wider than 256), any top_k the kernel serves, with and without the column clustering."""
import numpy as np
import pytest
import scipy.sparse as sp

from rtrec_amd import _native
from rtrec_amd.engine import SlimEngine, merge_coefficients
from rtrec_amd.synth import interaction_matrix, structured_matrix

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def random_w(I, density, seed, signed=False, n_blocks=0):
    """A general item-item matrix: uniformly random entries, optionally concentrated in diagonal blocks."""
    rng = np.random.default_rng(seed)
    nnz = int(I * I * density)
    r, c = rng.integers(0, I, nnz), rng.integers(0, I, nnz)
    if n_blocks:
        blk = I // n_blocks
        inside = rng.random(nnz) < 0.8
        c = np.where(inside, (r // blk) * blk + rng.integers(0, blk, nnz), c) % I
    v = (rng.random(nnz).astype(np.float32) + 0.01) * (np.where(rng.random(nnz) < 0.3, -1, 1) if signed else 1)
    W = sp.csc_matrix((v.astype(np.float32), (r, c)), shape=(I, I))
    W.sum_duplicates()
    W.setdiag(0)
    W.eliminate_zeros()
    W.sort_indices()
    return W


def check(eng, oracle, X, W, rows, top_k, filt, expect_path="segments"):
    ids, sc, cnt = eng.recommend_rows(rows, top_k=top_k, filter_interacted=filt, mode=_native.TOPK_SPARSE)
    if expect_path:
        assert eng.last_score_path == expect_path
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows], W.tocsr(), top_k=top_k, filter_interacted=filt)
    assert np.array_equal(cnt, o_cnt)
    bad = np.flatnonzero((ids != o_ids).any(axis=1))
    assert bad.size == 0, f"ids differ for rows {rows[bad][:8]}: {ids[bad[0]]} vs {o_ids[bad[0]]}"
    assert np.array_equal(bits(sc), bits(o_sc))


@pytest.mark.parametrize("signed", [False, True])
@pytest.mark.parametrize("top_k", [1, 10, 25, 63])
@pytest.mark.parametrize("filt", [True, False])
def test_seg_random_w_bit_exact(oracle, signed, top_k, filt):
    I = 1500
    W = random_w(I, 0.01, seed=5, signed=signed, n_blocks=12)
    X = interaction_matrix(2500, I, 90000, seed=17)
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    check(eng, oracle, X, W, np.arange(0, X.shape[0], 2), top_k, filt)


@pytest.mark.parametrize("cluster", [True, False])
def test_seg_fitted_structured_w(oracle, cluster):
    """W fitted by the oracle on clustered data (thousands of rows), all users, both column orders."""
    U, I, K = 6000, 1200, 50
    X = structured_matrix(U, I, 260000, seed=7, n_clusters=12)
    Xc = X.tocsc()
    Xc.sort_indices()
    ptr, idx, val, _ = oracle.fit_columns(Xc, np.arange(I), nn_feature_selection=K, n_threads=8)
    W = merge_coefficients(None, I, idx.astype(np.int64), np.repeat(np.arange(I, dtype=np.int64), np.diff(ptr)), val)
    assert np.count_nonzero(np.diff(W.tocsr().indptr)) > 128          # beyond the feature-row kernel
    eng = SlimEngine(device="cuda:0")
    eng.seg_cluster = cluster
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    check(eng, oracle, X, W, np.arange(U), 10, True)
    check(eng, oracle, X, W, np.arange(U)[::-1].copy(), 10, False)


@pytest.mark.parametrize("heavy_pass", [True, False])
def test_seg_long_users_and_foreign_rows(oracle, heavy_pass):
    """Users with more items than a wave's LDS lists hold (1024 items / 512 rows of W): the workgroup-per-user heavy pass
    (or, without its scratch, the per-tile re-read path); plus rows outside the matrix in the batch (IndexError) and
    items newer than W."""
    I = 2000
    W = random_w(I, 0.006, seed=9, n_blocks=8)
    X = interaction_matrix(900, I, 60000, seed=23).tolil()
    rng = np.random.default_rng(3)
    for u, n in ((5, 1900), (6, 1025), (7, 700), (8, 513), (9, 300), (10, 1024)):
        cols = rng.choice(I, n, replace=False)
        X[u, cols] = (rng.random(n) * 4 + 0.5).astype(np.float32)
    X = X.tocsr().astype(np.float32)
    X.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.use_seg_heavy = heavy_pass
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    check(eng, oracle, X, W, np.arange(X.shape[0]), 10, True)
    check(eng, oracle, X, W, np.arange(X.shape[0]), 10, True)          # again: the heavy pass left its scratch zeroed
    check(eng, oracle, X, W, np.array([5, 6, 7, 8, 9, 10, 5]), 20, False)
    check(eng, oracle, X, W, np.array([5, 6, 7, 8, 9, 10, 5]), 63, True)
    with pytest.raises(IndexError):
        eng.recommend_rows(np.array([0, 900]), top_k=5)
    # X has more item columns than W has rows: those items have no row and no column
    X2 = sp.hstack([X, sp.csr_matrix(np.ones((X.shape[0], 3), dtype=np.float32))]).tocsr()
    X2.sort_indices()
    eng2 = SlimEngine(device="cuda:0")
    eng2.set_interactions(None, X2, need_csc=False)
    eng2.set_weights(W)
    ids, sc, cnt = eng2.recommend_rows(np.arange(50), top_k=10)
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X[:50], W.tocsr(), top_k=10)
    assert eng2.last_score_path == "segments"
    assert np.array_equal(ids, o_ids) and np.array_equal(bits(sc), bits(o_sc)) and np.array_equal(cnt, o_cnt)


@pytest.mark.parametrize("knob", [0, 1, 17, 513])
def test_seg_request_sized_passes_give_users_a_workgroup(oracle, knob):
    """Passes of at most 256 rows hand every user of more than 32 items to the workgroup-per-user kernel (its latency is
    the longest user's); the knob (rtrec_score_opts.diagnostics bits 12-23) moves that threshold for any pass size.  Same
    answers whichever kernel scores a user: one user per call, request-sized batches, and a full pass with the knob."""
    I = 2000
    W = random_w(I, 0.006, seed=9, n_blocks=8, signed=True)
    X = interaction_matrix(900, I, 60000, seed=23).tolil()
    rng = np.random.default_rng(3)
    for u, n in ((5, 1900), (6, 1025), (7, 33), (8, 32), (9, 300), (10, 1)):
        X[u, :] = 0
        cols = rng.choice(I, n, replace=False)
        X[u, cols] = (rng.random(n) * 4 + 0.5).astype(np.float32)
    X[11, :] = 0                                  # an empty row
    X = X.tocsr().astype(np.float32)
    X.eliminate_zeros()
    X.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.sg_heavy_min = knob
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    for u in (5, 6, 7, 8, 9, 10, 11, 100):
        check(eng, oracle, X, W, np.array([u]), 10, True)
    check(eng, oracle, X, W, np.arange(0, 256), 10, True)
    check(eng, oracle, X, W, np.arange(0, 256), 10, True)           # again: the scratch was left zeroed
    check(eng, oracle, X, W, np.array([5, 11, 6, 7, 8, 9, 10, 5]), 63, False)
    check(eng, oracle, X, W, np.arange(X.shape[0]), 5, True)


def test_feature_row_w_scores_request_sized_batches_from_its_segment_form(oracle):
    """A W with at most 128 rows has the feature-row form (a throughput kernel: one wave sweeps all of W per user); batches
    below SlimEngine.FR_SMALL_BATCH rows are scored from the segment form of the same W instead -- same answers."""
    I = 3000
    rng = np.random.default_rng(12)
    rows = np.sort(rng.choice(I, 100, replace=False))
    nnz = 60_000
    W = sp.csc_matrix(((rng.random(nnz) + 0.01).astype(np.float32), (rng.choice(rows, nnz), rng.integers(0, I, nnz))), shape=(I, I))
    W.sum_duplicates(); W.setdiag(0); W.eliminate_zeros(); W.sort_indices()
    X = interaction_matrix(1200, I, 90000, seed=5).astype(np.float32)
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    check(eng, oracle, X, W, np.arange(X.shape[0]), 10, True, expect_path="feature_rows")
    for b in (1, 5, eng.FR_SMALL_BATCH - 1):
        check(eng, oracle, X, W, np.arange(7, 7 + b), 10, True, expect_path="segments")
    check(eng, oracle, X, W, np.arange(eng.FR_SMALL_BATCH), 10, False, expect_path="feature_rows")


@pytest.mark.parametrize("case", ["blocks", "wide", "shard", "dense_rows"])
def test_native_layout_builder_equals_the_numpy_specification(case):
    """csrc/seg_build.hip (rtrec_slim_seg_plan + rtrec_slim_seg_fill: what a recommend right after a mini-batch waits for)
    against seg_layout.build_seg_layout, array by array: cluster-ordered columns, tiles wider than 256, a column shard
    of a larger W, rows dense enough for 256-float blocks."""
    import torch
    from rtrec_amd.engine import HipBackend
    from rtrec_amd.seg_layout import build_seg_layout, build_seg_layout_native, cluster_labels
    if case == "wide":
        I, W, lo, hi = 70_000, random_w(70_000, 0.00004, seed=3), 0, 70_000
    elif case == "dense_rows":
        I, lo, hi = 3000, 0, 3000
        rng = np.random.default_rng(12)
        rows = np.sort(rng.choice(I, 60, replace=False))
        nnz = 90_000
        W = sp.csc_matrix(((rng.random(nnz) + 0.01).astype(np.float32), (rng.choice(rows, nnz), rng.integers(0, I, nnz))), shape=(I, I))
        W.sum_duplicates(); W.eliminate_zeros(); W.sort_indices()
    else:
        I, W = 2000, random_w(2000, 0.006, seed=9, n_blocks=8, signed=True)
        lo, hi = (0, I) if case == "blocks" else (700, 1500)
    coo = W.tocoo()
    order = np.lexsort((coo.row, coo.col))
    r, c, v = coo.row[order].astype(np.int64), coo.col[order].astype(np.int64), coo.data[order].astype(np.float32)
    for labels in (np.arange(I, dtype=np.int64), cluster_labels(r, c, v, I)):
        ref = build_seg_layout(W, lo, hi, labels=labels)
        be = HipBackend("cuda:0")
        got = build_seg_layout_native(be, be.to_dev(r), be.to_dev(c), be.to_dev(v), I, lo, hi, be.to_dev(labels))
        torch.cuda.synchronize()
        for k in ("sg_T", "sg_n_tiles", "sg_rows", "sg_n_cols"):
            assert got[k] == ref[k], k
        if case == "wide":
            assert got["sg_T"] > 256
        n_rec, n_list = int(ref["sg_ent"].shape[0]), int(ref["sg_trow"].shape[0])
        assert int(got["sg_ptr"][-1, -1]) == n_rec and int(got["sg_trow_ptr"][-1]) == n_list
        for k in ("sg_info", "sg_ptr", "sg_bound", "sg_col_ids", "sg_trow_ptr"):
            assert np.array_equal(got[k].cpu().numpy(), np.asarray(ref[k])), k
        assert np.array_equal(got["sg_ent"][:n_rec].cpu().numpy(), ref["sg_ent"])
        assert np.array_equal(got["sg_trow"][:n_list].cpu().numpy(), ref["sg_trow"])
        if case == "dense_rows":
            assert ref["sg_dense_segments"] > 0


@pytest.mark.parametrize("shape", ["general", "feature_rows"])
@pytest.mark.parametrize("filt", [True, False])
def test_dense_mode_through_the_fast_pass(oracle, shape, filt):
    """DENSE mode (string item ids: every column competes, zeros included; slim_elastic.py:745-778) runs the fast pass too:
    rows whose leading top_k scores are all positive are final, the kernel flags the rest (short users, negative scores,
    exact ties -- DENSE orders ties by item id) for the tiled DENSE kernel.  Against the oracle's dense mode, bit for bit."""
    I = 2500
    rng = np.random.default_rng(4)
    if shape == "general":
        W = random_w(I, 0.005, seed=21, n_blocks=10, signed=True)
    else:
        rows = np.sort(rng.choice(I, 90, replace=False))
        nnz = 50_000
        v = (rng.random(nnz) + 0.01).astype(np.float32) * np.where(rng.random(nnz) < 0.2, -1, 1)
        W = sp.csc_matrix((v.astype(np.float32), (rng.choice(rows, nnz), rng.integers(0, I, nnz))), shape=(I, I))
        W.sum_duplicates(); W.setdiag(0); W.eliminate_zeros(); W.sort_indices()
    X = interaction_matrix(1500, I, 80000, seed=6).tolil()
    X[3, :] = 0                                            # no items at all: the highest ids, all zeros
    X[4, :] = 0; X[4, 17] = 2.0                            # one item: a handful of non-zero scores, then zeros by id
    X[5, :] = 0; X[5, [100, 200]] = [1.0, 1.0]             # equal ratings: exact ties likely
    X = X.tocsr().astype(np.float32)
    X.eliminate_zeros(); X.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    for rows, k in ((np.arange(X.shape[0]), 10), (np.arange(0, 40), 5), (np.array([3, 4, 5, 700]), 15), (np.array([4]), 10)):
        ids, sc, cnt = eng.recommend_rows(rows, top_k=k, filter_interacted=filt, mode=_native.TOPK_DENSE)
        assert eng.last_score_path in ("segments", "feature_rows")
        o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows], W.tocsr(), top_k=k, filter_interacted=filt, dense=True)
        assert np.array_equal(cnt, o_cnt)
        bad = np.flatnonzero((ids != o_ids).any(axis=1))
        assert bad.size == 0, f"ids differ for rows {rows[bad][:8]}: {ids[bad[0]]} vs {o_ids[bad[0]]}"
        assert np.array_equal(bits(sc), bits(o_sc))
    # the same through the tiled kernel alone (the A/B switch)
    eng.dense_fast = False
    ids2, sc2, cnt2 = eng.recommend_rows(np.arange(X.shape[0]), top_k=10, filter_interacted=filt, mode=_native.TOPK_DENSE)
    assert eng.last_score_path == "tiled"
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X, W.tocsr(), top_k=10, filter_interacted=filt, dense=True)
    assert np.array_equal(ids2, o_ids) and np.array_equal(bits(sc2), bits(o_sc)) and np.array_equal(cnt2, o_cnt)


@pytest.mark.parametrize("filt", [True, False])
@pytest.mark.parametrize("shape", ["general", "feature_rows"])
def test_dense_mode_short_lists_are_completed_in_place_and_column_shards_take_the_fast_pass(oracle, shape, filt):
    """DENSE mode (slim_elastic.py:745-778), positive W and ratings: a user with fewer than top_k positive scores gets the
    zero-score columns behind them, the higher id first, interacted items excluded (rtrec_slim_dense_fill) -- without the tiled
    kernel -- and a COLUMN SHARD (where that is nearly every user) therefore takes the fast pass too.  Full W and three column
    shards merged, against the oracle's dense mode, bit for bit; users with no items, one item, equal ratings (exact ties:
    those rows stay flagged), and one who rated the whole top of the id range."""
    import torch
    I = 2500
    rng = np.random.default_rng(14)
    if shape == "general":
        W = random_w(I, 0.002, seed=23, n_blocks=10)
    else:
        rows = np.sort(rng.choice(I, 90, replace=False))
        nnz = 30_000
        W = sp.csc_matrix(((rng.random(nnz) + 0.01).astype(np.float32), (rng.choice(rows, nnz), rng.integers(0, I, nnz))), shape=(I, I))
        W.sum_duplicates(); W.setdiag(0); W.eliminate_zeros(); W.sort_indices()
    X = interaction_matrix(1500, I, 9000, seed=6).tolil()        # ~6 items per user: few positive scores, fewer per shard
    X[3, :] = 0
    X[4, :] = 0; X[4, 17] = 2.0
    X[5, :] = 0; X[5, [100, 200]] = [1.0, 1.0]
    X[6, :] = 0; X[6, np.arange(I - 40, I)] = 1.5            # rated the 40 highest ids: the zero fill has to skip them
    X[7, :] = 0; X[7, np.arange(0, I, 2)] = 0.5              # rated every other item
    X = X.tocsr().astype(np.float32)
    X.eliminate_zeros(); X.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    eng.rescored = eng.be.zeros((1,), torch.int32)
    for rows, k in ((np.arange(X.shape[0]), 10), (np.arange(0, 40), 5), (np.array([3, 4, 5, 6, 7, 700]), 15), (np.array([6]), 63)):
        ids, sc, cnt = eng.recommend_rows(rows, top_k=k, filter_interacted=filt, mode=_native.TOPK_DENSE)
        assert eng.last_score_path in ("segments", "feature_rows")
        o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows], W.tocsr(), top_k=k, filter_interacted=filt, dense=True)
        assert np.array_equal(cnt, o_cnt)
        bad = np.flatnonzero((ids != o_ids).any(axis=1))
        assert bad.size == 0, f"ids differ for rows {rows[bad][:8]}: {ids[bad[0]]} vs {o_ids[bad[0]]}"
        assert np.array_equal(bits(sc), bits(o_sc))
        if len(rows) == X.shape[0]:
            assert int(eng.rescored.item()) < len(rows) // 10, "most short lists should have been completed in place"
    # three column shards on one GPU, merged like the exchange does
    rows, k = np.arange(X.shape[0]), 10
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows], W.tocsr(), top_k=k, filter_interacted=filt, dense=True)
    parts = []
    for r in range(3):
        e = SlimEngine(device="cuda:0", rank=r, world_size=3)
        e.world_size_for_merge = 3
        e.set_interactions(None, X, need_csc=False)
        e.set_weights(W)
        e.rescored = e.be.zeros((1,), torch.int32)
        d_rows = e.be.to_dev(rows.astype(np.int32))
        xb = (e._X["rptr"], e._X["rcol"], e._X["rval"])
        parts.append(e._local_topk(d_rows, len(rows), xb, k, filt, _native.TOPK_DENSE, None))
        assert e.last_score_path in ("segments", "feature_rows"), e.last_score_path
        assert int(e.rescored.item()) < len(rows) // 10
    be = eng.be
    g = [torch.stack([p[j] for p in parts]).contiguous() for j in (0, 1, 3, 4)]
    m_ids, m_sc, m_cnt = be.empty((len(rows), k), torch.int32), be.empty((len(rows), k), torch.float32), be.empty((len(rows),), torch.int32)
    _native.check(be.lib.rtrec_slim_merge_topk(len(rows), 3, k, be.ptr(g[0]), be.ptr(g[1]), None, be.ptr(g[2]), be.ptr(g[3]),
                                               be.ptr(m_ids), be.ptr(m_sc), be.ptr(m_cnt), be.stream()), "merge")
    assert np.array_equal(m_cnt.cpu().numpy(), o_cnt)
    bad = np.flatnonzero((m_ids.cpu().numpy() != o_ids).any(axis=1))
    assert bad.size == 0, f"sharded ids differ for rows {bad[:8]}: {m_ids.cpu().numpy()[bad[0]]} vs {o_ids[bad[0]]}"
    assert np.array_equal(bits(m_sc.cpu().numpy()), bits(o_sc))
    # a catalogue smaller than top_k leaves after the filter: the fill runs out of columns and the count says so
    It = 14
    Wt = sp.csc_matrix((np.array([0.5, 0.25, 1.0], np.float32), ([1, 2, 3], [5, 6, 7])), shape=(It, It))
    Xt = sp.csr_matrix((np.ones(9, np.float32), ([0, 0, 0, 0, 0, 1, 1, 2, 2], [1, 2, 9, 12, 13, 3, 0, 4, 11])), shape=(4, It))
    Xt.sort_indices()
    et = SlimEngine(device="cuda:0")
    et.set_interactions(None, Xt, need_csc=False)
    et.set_weights(Wt)
    for kk in (10, 13, 3):
        t_ids, t_sc, t_cnt = et.recommend_rows(np.arange(4), top_k=kk, filter_interacted=filt, mode=_native.TOPK_DENSE)
        o_ids, o_sc, o_cnt = oracle.recommend_batch(Xt, Wt.tocsr(), top_k=kk, filter_interacted=filt, dense=True)
        assert np.array_equal(t_cnt, o_cnt) and np.array_equal(t_ids, o_ids) and np.array_equal(bits(t_sc), bits(o_sc)), (kk, t_ids, o_ids)
    # the switch: without the fill a shard keeps the tiled kernel
    e.dense_fill = False
    e._local_topk(d_rows, len(rows), xb, k, filt, _native.TOPK_DENSE, None)
    assert e.last_score_path == "tiled"


@pytest.mark.parametrize("shape", ["general", "feature_rows", "signed"])
def test_float64_w_through_the_fast_pass_and_the_refine_step(oracle, shape):
    """A W that is float64 on the host (the reference's serial fit, slim_elastic.py:252; float32-valued) accumulates float64
    scores.  With positive weights and ratings the float32 fast pass (top_k + 1 columns) + rtrec_slim_refine_topk_f64 gives
    the float64 answer: ids, float32 casts of the float64 scores and counts equal the oracle's use_f64 mode; rows the margin
    test or a tie flags go to the float64 tiled kernel.  A W with NEGATIVE weights (round 4) takes the same two steps in
    SPARSE mode with an absolute per-user slack instead of the sign argument; its DENSE mode keeps the tiled kernel."""
    I = 2500
    rng = np.random.default_rng(8)
    if shape == "feature_rows":
        rows = np.sort(rng.choice(I, 90, replace=False))
        nnz = 50_000
        W = sp.csc_matrix(((rng.random(nnz) + 0.01).astype(np.float32), (rng.choice(rows, nnz), rng.integers(0, I, nnz))), shape=(I, I))
        W.sum_duplicates(); W.setdiag(0); W.eliminate_zeros(); W.sort_indices()
    else:
        W = random_w(I, 0.005, seed=31, n_blocks=10, signed=(shape == "signed"))
    X = interaction_matrix(1500, I, 80000, seed=6).tolil()
    X[3, :] = 0
    X[4, :] = 0; X[4, 17] = 2.0
    X[5, :] = 0; X[5, [100, 200]] = [1.0, 1.0]
    X = X.tocsr().astype(np.float32)
    X.eliminate_zeros(); X.sort_indices()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W.astype(np.float64), acc_f64=True)
    Wr = W.tocsr()
    for rows_, k, filt in ((np.arange(X.shape[0]), 10, True), (np.arange(0, 40), 5, False), (np.array([3, 4, 5, 700]), 14, True),
                           (np.array([4]), 10, True), (np.arange(X.shape[0]), 30, True)):
        ids, sc, cnt = eng.recommend_rows(rows_, top_k=k, filter_interacted=filt, mode=_native.TOPK_SPARSE)
        assert eng.last_score_path.endswith("+f64"), eng.last_score_path
        o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows_], Wr, top_k=k, filter_interacted=filt, use_f64=True)
        assert np.array_equal(cnt, o_cnt)
        bad = np.flatnonzero((ids != o_ids).any(axis=1))
        assert bad.size == 0, f"ids differ for rows {rows_[bad][:8]}: {ids[bad[0]]} vs {o_ids[bad[0]]}"
        assert np.array_equal(bits(sc), bits(o_sc))
    for rows_, k in ((np.arange(X.shape[0]), 10), (np.array([3, 4, 5, 700]), 7)):         # DENSE mode with the float64 W
        ids, sc, cnt = eng.recommend_rows(rows_, top_k=k, filter_interacted=True, mode=_native.TOPK_DENSE)
        o_ids, o_sc, o_cnt = oracle.recommend_batch(X[rows_], Wr, top_k=k, filter_interacted=True, dense=True, use_f64=True)
        assert np.array_equal(cnt, o_cnt) and np.array_equal(ids, o_ids) and np.array_equal(bits(sc), bits(o_sc))
        assert (shape == "signed") == (eng.last_score_path == "tiled")
    if shape == "signed":       # most rows must be final after the refine step, or the path is pointless
        import torch
        eng.rescored = torch.zeros(1, dtype=torch.int32, device="cuda:0")
        eng.recommend_rows(np.arange(X.shape[0]), top_k=10)
        assert int(eng.rescored.item()) < X.shape[0] // 4, int(eng.rescored.item())
        eng.rescored = None
        # negative RATINGS on top (a store with min_value < 0): still the oracle's float64 answer
        Xn = X.copy()
        Xn.data[::7] *= -1.0
        eng.set_interactions(None, Xn, need_csc=False)
        ids, sc, cnt = eng.recommend_rows(np.arange(Xn.shape[0]), top_k=10, mode=_native.TOPK_SPARSE)
        assert eng.last_score_path.endswith("+f64")
        o_ids, o_sc, o_cnt = oracle.recommend_batch(Xn, Wr, top_k=10, filter_interacted=True, use_f64=True)
        assert np.array_equal(cnt, o_cnt) and np.array_equal(ids, o_ids) and np.array_equal(bits(sc), bits(o_sc))
        # X replaced (more users, larger ratings) while W -- and the per-user slack cached with it -- stays:
        # the slack must follow the new X (it used to be read past its end / be too small for the new ratings)
        Xg = sp.vstack([X * 3.0, X[:200] * 5.0]).tocsr().astype(np.float32)
        Xg.sort_indices()
        eng.set_interactions(None, Xg, need_csc=False)
        ids, sc, cnt = eng.recommend_rows(np.arange(Xg.shape[0]), top_k=10, mode=_native.TOPK_SPARSE)
        assert eng.last_score_path.endswith("+f64")
        o_ids, o_sc, o_cnt = oracle.recommend_batch(Xg, Wr, top_k=10, filter_interacted=True, use_f64=True)
        assert np.array_equal(cnt, o_cnt) and np.array_equal(ids, o_ids) and np.array_equal(bits(sc), bits(o_sc))
        eng.set_interactions(None, X, need_csc=False)
    if True:                    # A/B: the tiled float64 kernel alone gives the same arrays
        a = eng.recommend_rows(np.arange(X.shape[0]), top_k=10)
        eng.f64_refine = False
        eng.set_weights(W.astype(np.float64), acc_f64=True)
        b = eng.recommend_rows(np.arange(X.shape[0]), top_k=10)
        assert eng.last_score_path == "tiled"
        assert all(np.array_equal(x.view(np.int32), y.view(np.int32)) for x, y in zip(a, b))


def test_seg_exact_ties_go_through_the_exact_pass(oracle):
    """Integer ratings and duplicated columns of W: exact score ties inside and at the edge of the list; the flagged rows
    are re-scored by the first-touch kernel and come out in the reference's order."""
    rng = np.random.default_rng(0)
    I = 900
    base = sp.random(I, 60, density=0.05, random_state=1, format="csc", dtype=np.float32)
    W = sp.csc_matrix(base[:, rng.integers(0, 60, size=I)])
    W.sort_indices()
    assert np.count_nonzero(np.diff(W.tocsr().indptr)) > 128
    X = interaction_matrix(700, I, 20000, seed=9, float_ratings=False)
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    for filt in (True, False):
        check(eng, oracle, X, W, np.arange(X.shape[0]), 10, filt)


def test_seg_wide_catalogue_uses_wider_tiles(oracle):
    """More than 128 x 256 active columns: tiles of 512 columns."""
    I = 40000
    rng = np.random.default_rng(4)
    nnz = 400000
    r = rng.integers(0, 3000, nnz)                   # 3000 rows hold the weights
    c = rng.integers(0, I, nnz)
    W = sp.csc_matrix(((rng.random(nnz) + 0.05).astype(np.float32), (r, c)), shape=(I, I))
    W.sum_duplicates()
    W.sort_indices()
    X = interaction_matrix(1200, I, 150000, seed=31)
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    check(eng, oracle, X, W, np.arange(X.shape[0]), 10, True)
    lay = eng._layout(True, 10)
    assert lay["sg"]["sg_T"] == 512 and lay["sg"]["sg_n_tiles"] <= 128


def test_seg_small_and_empty_cases(oracle):
    I = 400
    W = random_w(I, 0.02, seed=2)
    X = interaction_matrix(300, I, 5000, seed=3).tolil()
    X[10, :] = 0                                         # a user without items
    X = X.tocsr().astype(np.float32)
    X.eliminate_zeros()
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    check(eng, oracle, X, W, np.array([10]), 10, True)              # empty row: count 0
    check(eng, oracle, X, W, np.array([3]), 10, True)               # one user
    check(eng, oracle, X, W, np.arange(300), 63, True)              # k larger than most users' candidate sets
    check(eng, oracle, X, W, np.arange(300), 64, True, expect_path="tiled")     # beyond the kernel's list: the tiled path


def test_seg_generic_path_at_scale_agrees_with_the_tiled_kernel(oracle):
    """A 100k-item catalogue with 2 M block-structured weights over ~95k active columns: tiles of 1024 columns, the generic
    accumulate loop (multi-step segments, no dense blocks), 16-bit LDS lists no longer possible (wide instantiation) -- all
    40,000 users against the tiled-CSR kernel, a sample against the oracle."""
    I = 100_000
    rng = np.random.default_rng(8)
    nnz = 2_000_000
    r = rng.integers(0, I, nnz)
    blk = 500
    c = np.where(rng.random(nnz) < 0.85, (r // blk) * blk + rng.integers(0, blk, nnz), rng.integers(0, I, nnz)) % I
    W = sp.csc_matrix(((rng.random(nnz) + 0.02).astype(np.float32), (r, c)), shape=(I, I))
    W.sum_duplicates()
    W.sort_indices()
    X = interaction_matrix(40_000, I, 3_000_000, seed=61)
    eng = SlimEngine(device="cuda:0")
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    rows = np.arange(X.shape[0])
    ids, sc, cnt = eng.recommend_rows(rows, top_k=10)
    assert eng.last_score_path == "segments"
    sg = eng._fast_layout()["sg"]
    assert sg["sg_T"] == 1024 and sg["sg_rows"] > 65_535
    eng.use_seg_layout = False
    ids2, sc2, cnt2 = eng.recommend_rows(rows, top_k=10)
    assert eng.last_score_path == "tiled"
    assert np.array_equal(ids, ids2) and np.array_equal(bits(sc), bits(sc2)) and np.array_equal(cnt, cnt2)
    sample = np.sort(rng.choice(X.shape[0], 1500, replace=False))
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X[sample], W.tocsr(), top_k=10, n_threads=8)
    assert np.array_equal(ids[sample], o_ids) and np.array_equal(bits(sc[sample]), bits(o_sc)) and np.array_equal(cnt[sample], o_cnt)


def test_segment_pass_never_gets_the_pattern_grouped_order(oracle):
    """a feature-row shaped STREAMING W, top_k beyond the feature-row kernel's lists (so the segment
    kernels score it), >= GROUPED_ORDER_MIN rows, the tiled layout already built, and the same resident row tensor scored
    three times: the cached work order must stay the length order (the segment kernels stop looking for long users at
    the first short one), never the feature-row kernel's pattern-grouped order.  Long users sit in the middle of the row
    set; every row is compared with the tiled kernel, the long ones with the oracle."""
    import torch
    I, U = 3000, 34000
    rng = np.random.default_rng(12)
    wrows = np.sort(rng.choice(I, 100, replace=False))
    nnz = 60_000
    W = sp.csc_matrix(((rng.random(nnz) + 0.01).astype(np.float32), (rng.choice(wrows, nnz), rng.integers(0, I, nnz))), shape=(I, I))
    W.sum_duplicates(); W.setdiag(0); W.eliminate_zeros(); W.sort_indices()
    X = interaction_matrix(U, I, 700_000, seed=5).astype(np.float32).tolil()
    long_users = np.sort(rng.choice(np.arange(5000, U - 5000), 40, replace=False))
    for u in long_users:
        items = rng.choice(I, int(rng.integers(600, 1400)), replace=False)
        X[u, items] = (rng.integers(1, 6, items.size)).astype(np.float32)
    X = X.tocsr().astype(np.float32)
    X.sort_indices()
    eng = SlimEngine(device="cuda:0")
    assert U >= eng.GROUPED_ORDER_MIN
    eng.set_interactions(None, X, need_csc=False)
    eng.set_weights(W)
    d_rows = torch.arange(U, dtype=torch.int32, device="cuda:0")
    S = _native.TOPK_SPARSE
    for _ in range(3):                                                  # the feature-row pass: its order becomes the grouped one
        eng.score_topk_device(None, U, 10, True, S, d_rows=d_rows)
    assert eng.last_score_path == "feature_rows"
    fast = eng._fast_layout()
    assert fast.get("fr_host") is not None and not fast["fr_host"].get("fr_resident"), "the test needs the streaming layout"
    assert eng._order_grouped, "the test needs the pattern-grouped order in the cache"
    eng._layout(compact=True, top_k=20)                                 # the tiled layout exists (as after an exact-tie call)
    outs = []
    for _ in range(3):
        ids, sc, cnt = eng.score_topk_device(None, U, 20, True, S, d_rows=d_rows)
        assert eng.last_score_path == "segments" and not eng._order_grouped
        outs.append((ids.cpu().numpy(), sc.cpu().numpy(), cnt.cpu().numpy()))
    eng.use_seg_layout = eng.use_feature_rows = False
    t_ids, t_sc, t_cnt = eng.score_topk_device(None, U, 20, True, S, d_rows=d_rows)
    assert eng.last_score_path == "tiled"
    t_ids, t_sc, t_cnt = t_ids.cpu().numpy(), t_sc.cpu().numpy(), t_cnt.cpu().numpy()
    o_ids, o_sc, o_cnt = oracle.recommend_batch(X[long_users], W.tocsr(), top_k=20, filter_interacted=True)
    for ids, sc, cnt in outs:
        assert np.array_equal(cnt, t_cnt)
        m = np.arange(20)[None, :] < cnt[:, None]
        assert np.array_equal(ids[m], t_ids[m]) and np.array_equal(bits(sc)[m], bits(t_sc)[m])
        assert np.array_equal(ids[long_users], o_ids) and np.array_equal(cnt[long_users], o_cnt)
        assert np.array_equal(bits(sc[long_users]), bits(o_sc))
