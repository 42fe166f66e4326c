"""Host-side mirror of the reference interface: store, ids, LRU, metrics, W write-back, layouts.

Expected values are either golden vectors produced by the real reference (tests/golden/, see
tools/gen_golden.py) or the expectations of the reference's own unit tests restated
(/root/reference/tests/utils/test_interactions.py, test_lru.py, test_metrics.py).
"""
import json
import os
import time

import numpy as np
import pytest
import scipy.sparse as sp

from rtrec_amd.engine import (DENSE_ROW_FILL, DeviceWeights, FR_STREAM_BUF_BYTES, FR_TILE_HEADER_BYTES, SlimEngine, _pack_fragments, build_feature_rows, build_feature_rows_device, build_tiled_w,
                              build_tiled_w_device, coefficients_to_updates, merge_coefficients, row_header_table, shard_bounds,
                              sklearn_seed)
from rtrec_amd.utils.identifiers import Identifier, IdentifierError
from rtrec_amd.utils.interactions import UserItemInteractions
from rtrec_amd.utils.lru import LRUFreqSet
from rtrec_amd.utils import metrics

G = os.path.join(os.path.dirname(__file__), "golden")


def load_csc(z, prefix):
    return sp.csc_matrix((z[f"{prefix}_data"], z[f"{prefix}_indices"], z[f"{prefix}_indptr"]),
                         shape=tuple(z[f"{prefix}_shape"]))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


# ------------------------------------------------------------------ interaction store
@pytest.mark.parametrize("name,kw,upsert", [("plain", {}, False), ("decay7", {"decay_in_days": 7}, False),
                                            ("decay7_upsert", {"decay_in_days": 7}, True),
                                            ("clip", {"min_value": -1, "max_value": 4}, False)])
@pytest.mark.parametrize("batch", [1, 7, 400])
def test_store_matches_reference(name, kw, upsert, batch):
    d = json.load(open(os.path.join(G, "store.json")))
    ev, ref = d["events"], d[name]
    s = UserItemInteractions(**kw)
    for a in range(0, len(ev["u"]), batch):   # batched ingest must equal one-by-one ingest
        s.add_interactions_batch(ev["u"][a:a + batch], ev["i"][a:a + batch], ev["ts"][a:a + batch],
                                 ev["r"][a:a + batch], upsert=upsert)
    assert list(s.shape) == ref["shape"]
    assert s.max_timestamp == ref["max_timestamp"]
    # float32 matrices: bit-identical -- decay goes through libm's pow (rtrec_store_decay), the function
    # behind the reference's `rate ** days`
    for got, exp in ((s.to_csr().toarray(), ref["csr"]), (s.to_csc([1, 3, 5, 24]).toarray(), ref["csc_sel"]),
                     (s.to_csr([0, 2, 29]).toarray(), ref["csr_sel"])):
        exp = np.asarray(exp, dtype=np.float32)
        assert got.dtype == np.float32 and got.shape == exp.shape
        assert np.array_equal(got != 0, exp != 0)
        assert np.array_equal(bits(got), bits(exp))
    assert s.get_hot_items(10, filter_interacted=False) == ref["hot"]
    assert s.get_user_item_rating(3, 4) == ref["rating_3_4"]
    assert sorted(s.get_user_items(5)) == ref["user_items_5"]


def test_decay_rates_match_reference():
    d = json.load(open(os.path.join(G, "store.json")))["decay_rate"]
    for days, rate in d.items():
        assert UserItemInteractions(decay_in_days=int(days)).decay_rate == rate
    assert UserItemInteractions(decay_in_days=7).decay_rate == 0.9009789742057221


def test_store_reference_unit_expectations():
    s = UserItemInteractions(min_value=-5, max_value=10)
    t = time.time()
    s.add_interaction(1, 10, t, 5.0)
    assert s.get_user_item_rating(1, 10) == 5.0
    s.add_interaction(1, 10, t, 3.0)
    assert s.get_user_item_rating(1, 10) == 8.0
    s.add_interaction(1, 10, t, 5.0)
    assert s.get_user_item_rating(1, 10) == 10.0          # clipped
    s.add_interaction(1, 20, t, -2.0)
    s.add_interaction(2, 15, t, 3.0)
    s.add_interaction(3, 9, t, 3.0)
    assert set(s.get_all_non_interacted_items(1)) == {15, 9}
    assert set(s.get_all_non_negative_items(1)) == {10, 15, 9}
    assert set(s.get_all_users()) == {1, 2, 3}
    assert set(s.get_user_items(1)) == {10, 20}
    assert s.get_user_items(99) == [] and s.get_user_item_rating(99, 10) == 0.0
    assert s.has_interaction(1, 20) and not s.has_interaction(1, 21)
    assert set(s.get_users_by_items([10, 9])) == {1, 3}
    s.add_interaction(1, 10, t, -10.0)
    assert s.get_user_item_rating(1, 10) == 0.0 and 10 in s.get_user_items(1)


def test_decay_half_life():
    s = UserItemInteractions(min_value=-5, max_value=10, decay_in_days=7)
    now = time.time()
    s.add_interaction(1, 10, now - 7 * 86400, 5.0)
    s.add_interaction(2, 10, now, 5.0)          # advances max_timestamp
    r = s.get_user_item_rating(1, 10)
    assert abs(r - 2.5) < 0.1
    assert abs(r - 5.0 * s.decay_rate ** ((s.max_timestamp - (now - 7 * 86400)) / 86400.0)) < 1e-9


def test_to_csc_exact_small():
    s = UserItemInteractions(min_value=-5, max_value=10, decay_in_days=None)
    s.add_interaction(0, 0, tstamp=12345, delta=5)
    s.add_interaction(0, 2, tstamp=12345, delta=3)
    s.add_interaction(1, 1, tstamp=12345, delta=2)
    s.add_interaction(2, 2, tstamp=12345, delta=4)
    assert np.array_equal(s.to_csc().toarray(), np.array([[5, 0, 3], [0, 2, 0], [0, 0, 4]], dtype=np.float32))
    assert np.array_equal(s.to_csc([2]).toarray(), np.array([[0, 0, 3], [0, 0, 0], [0, 0, 4]], dtype=np.float32))
    assert np.array_equal(s.to_coo(select_users=[0], select_items=[2]).toarray(),
                          np.array([[0, 0, 3], [0, 0, 0], [0, 0, 0]], dtype=np.float32))
    assert s.to_csr(include_weights=False).dtype == np.int32


# ------------------------------------------------------------------ identifiers / LRU
def test_identifier_semantics():
    ids = Identifier(name="user")
    assert ids.identify(7) == 7 and ids.pass_through is True and ids.get(7) == 7 and ids.get_id(9) == 9
    with pytest.raises(ValueError, match="Mixed types"):
        ids.identify("x")
    s = Identifier(name="item")
    assert [s.identify(x) for x in ("a", "b", "a")] == [0, 1, 0] and s.pass_through is False
    assert s.get(1) == "b" and s.get_id("b") == 1 and s.get_id("zz") is None and s[0] == "a"
    with pytest.raises(ValueError):
        s.identify(3)
    with pytest.raises(ValueError):
        s.get_id(3)
    with pytest.raises(IdentifierError):
        s.get(5)
    assert s.get_or_default(5, "d") == "d"
    f = Identifier(force_identify=True)
    assert f.identify(100) == 0 and f.identify(np.int64(5)) == 1 and f.get(0) == 100
    assert Identifier().identify_many(np.array([4, 2, 9])).tolist() == [4, 2, 9]
    assert Identifier().identify_many(["q", "r", "q"]).tolist() == [0, 1, 0]


def test_lru_freq_set():
    l = LRUFreqSet(capacity=3)
    for v in (1, 2, 3, 1, 4):        # 2 is evicted (least recently used), 1 was refreshed
        l.add(v)
    assert list(l) == [3, 1, 4] and 2 not in l and len(l) == 3
    assert list(l.get_freq_items()) == [1, 3, 4] and list(l.get_freq_items(1)) == [1]
    assert list(l.get_freq_items(2, exclude_items=[1])) == [3, 4]
    with pytest.raises(KeyError):
        l.discard(99)
    with pytest.raises(ValueError):
        LRUFreqSet(0)
    rng = np.random.default_rng(0)
    seq = rng.integers(0, 40, 500).tolist()
    for cap in (5, 64):
        a, b = LRUFreqSet(cap), LRUFreqSet(cap)
        for v in seq:
            a.add(v)
        for s in range(0, 500, 37):
            b.add_many(seq[s:s + 37])
        assert list(a.data.items()) == list(b.data.items())


# ------------------------------------------------------------------ metrics (reference formulas)
def test_metrics_known_values():
    ranked, truth = [1, 2, 3, 4, 5], [2, 5, 9]
    assert metrics.precision(ranked, truth, 5) == pytest.approx(0.4)
    assert metrics.recall(ranked, truth, 5) == pytest.approx(2 / 3)
    assert metrics.hit(ranked, truth, 5) == 1.0 and metrics.true_positives(ranked, truth, 5) == 2
    assert metrics.reciprocal_rank(ranked, truth, 5) == 0.5
    from math import log2
    dcg = 1 / log2(3) + 1 / log2(6)
    idcg = 1 / log2(2) + 1 / log2(3) + 1 / log2(4)
    assert metrics.ndcg(ranked, truth, 5) == pytest.approx(dcg / idcg)
    assert metrics.average_precision(ranked, truth, 5) == pytest.approx((1 / 2 + 2 / 5) / 3)
    assert metrics.auc(ranked, truth, 5) == pytest.approx((1 + 1 + 0) / 6)    # misses 3,4 after hit 2; none after 5
    assert metrics.precision([], [], 5) == 1.0 and metrics.f1_score([], [], 5) == 1.0 and metrics.auc([1], [], 5) == 0.0
    out = metrics.compute_scores([(ranked, truth), ([7], [7])], 5)
    assert out["tp"] == 3 and out["hit_rate"] == 1.0 and out["mrr"] == pytest.approx(0.75)
    assert metrics.compute_scores([], 5)["ndcg"] == 0.0


# ------------------------------------------------------------------ W write-back and layouts
def test_merge_coefficients_stale_entry_semantics():
    """non-zero overwrites, explicit zero deletes, unmentioned entries survive."""
    W0 = sp.csc_matrix(np.array([[0, .5, 0], [.25, 0, .75], [.125, 0, 0]], dtype=np.float32))
    rows = np.array([0, 1, 2]); cols = np.array([0, 0, 1]); vals = np.array([0.9, 0.0, 0.3], dtype=np.float32)
    W = merge_coefficients(W0, 3, rows, cols, vals).toarray()
    assert np.array_equal(W, np.array([[.9, .5, 0], [0, 0, .75], [.125, .3, 0]], dtype=np.float32))
    W4 = merge_coefficients(W0, 3, rows, cols, vals, dtype=np.float64)
    assert W4.dtype == np.float64 and W4.has_sorted_indices
    t, it, co, cnt = np.array([2, 0]), np.array([[1, 0, 9], [2, 9, 9]]), np.array([[.1, .2, 9], [.3, 9, 9]], np.float32), np.array([2, 1])
    r, c, v = coefficients_to_updates(t, it, co, cnt)
    assert r.tolist() == [1, 0, 2] and c.tolist() == [2, 2, 0] and np.allclose(v, [.1, .2, .3])


def _device_coo(W):
    import torch
    coo = W.tocoo()
    o = np.lexsort((coo.row, coo.col))
    return (torch, torch.from_numpy(coo.row[o].astype(np.int64)), torch.from_numpy(coo.col[o].astype(np.int64)),
            torch.from_numpy(coo.data[o].astype(np.float32)))


def _feature_row_w(n_items=700, n_feat=40, seed=5):
    """A W whose weights sit in a few rows (what top-K feature selection on a popularity-skewed catalogue gives)."""
    rng = np.random.default_rng(seed)
    feat = np.sort(rng.choice(n_items, n_feat, replace=False))
    M = np.zeros((n_items, n_items), dtype=np.float32)
    for j in rng.choice(n_items, int(n_items * 0.8), replace=False):
        rows = feat[rng.random(n_feat) < rng.uniform(0.05, 0.6)]
        M[rows, j] = rng.uniform(0.01, 1.0, len(rows)).astype(np.float32)
    np.fill_diagonal(M, 0)
    return sp.csc_matrix(M)


@pytest.mark.parametrize("shard", [(0, None), (1, 3)])
def test_device_layout_builders_equal_the_host_builders(shard):
    """The score layouts are built on the device from the resident W (engine.build_*_device); the numpy builders are
    their specification: every array must come out identical (tensor ops on CPU tensors here, the same code on the GPU)."""
    for W in (_feature_row_w(), sp.random(900, 900, density=0.03, random_state=2, format="csc", dtype=np.float32)):
        W.sort_indices()
        I = W.shape[0]
        lo, hi = (0, I) if shard[1] is None else shard_bounds(I, shard[1], shard[0])
        torch, r, c, v = _device_coo(W)
        for compact, tile in ((True, 256), (True, 4096), (False, 256)):
            T = build_tiled_w(W, lo, hi, tile, compact=compact, dense_fill=DENSE_ROW_FILL if compact else None)
            D = build_tiled_w_device(torch, r, c, v, I, lo, hi, tile, compact=compact, dense_fill=DENSE_ROW_FILL if compact else None)
            assert (D["n_cols"], D["tile_cols"], D["n_tiles"]) == (T.n_cols, T.tile_cols, T.n_tiles)
            assert np.array_equal(D["tile_ptr"].numpy(), T.tile_ptr) and np.array_equal(D["w_val"].numpy(), T.w_val)
            assert np.array_equal(D["w_col"].numpy().view(np.uint16), T.w_col)
            assert np.array_equal(D["row_hdr"].numpy(), row_header_table(T))
            assert (D["dense_idx"] is None) == (T.dense_idx is None)
            if T.dense_idx is not None:
                assert np.array_equal(D["dense_idx"].numpy(), T.dense_idx) and np.array_equal(D["dense_val"].numpy(), T.dense_val)
            if compact:
                assert np.array_equal(D["col_ids"].numpy(), T.col_ids) and np.array_equal(D["col_map"].numpy(), T.col_map)
                H = build_feature_rows(W, lo, hi, T.col_ids, T.col_map, tile_cols=128 if tile == 256 else 256)
                F = build_feature_rows_device(torch, r, c, v, I, lo, hi, tile_cols=128 if tile == 256 else 256)
                assert (H is None) == (F is None)
                if H is not None:
                    for k in ("fr_map", "fr_col_ids", "fr_col_map", "fr_w", "fr_tile_rows", "fr_tile_off", "fr_super_kb", "fr_super_tile",
                              "fr_frag_tile"):
                        assert np.array_equal(F[k].numpy().ravel(), np.asarray(H[k]).ravel()), k
                    for k in ("fr_rows", "fr_tile_cols", "fr_n_tiles", "fr_n_frags", "fr_n_super", "fr_buf_bytes"):
                        assert F[k] == H[k], k
    assert build_feature_rows_device(*_device_coo(_feature_row_w()), 700, 0, 700) is not None     # the dense form was exercised


@pytest.mark.parametrize("tc", [256, 128])
def test_fragment_packing_invariants(tc):
    """engine._pack_fragments: every stored row of every tile sits in exactly one fragment, fragments of a tile are
    consecutive and ascending, a tile's header fits in front of its first fragment, nothing overlaps or leaves its
    super-tile's buffer, at most 64 fragments per super-tile; small layouts come out resident (one super-tile)."""
    rng = np.random.default_rng(tc)
    row_bytes = tc * 4
    cases = [rng.integers(1, 128 if tc == 128 else 67, size=n) for n in (1, 5, 59, 200, 400)] + [np.full(300, 1), np.array([66] * 12)]
    for n_rows_t in cases:
        n_rows_t = np.asarray(n_rows_t, dtype=np.int64)
        P = _pack_fragments(n_rows_t, tc)
        F = P["n_frags"]
        assert P["super_frag"][0] == 0 and P["super_frag"][-1] == F and np.all(np.diff(P["super_frag"]) >= 1)
        assert np.all(np.diff(P["super_frag"]) <= 64)
        tile = P["frag_flags"] & 0xffffff
        first, last = (P["frag_flags"] >> 24) & 1, (P["frag_flags"] >> 25) & 1
        assert np.array_equal(tile, P["frag_tile"]) and np.all(np.diff(tile) >= 0)
        k1 = np.zeros(F, dtype=np.int64)
        for t in range(len(n_rows_t)):
            fr = np.flatnonzero(tile == t)
            assert len(fr) >= 1 and np.all(np.diff(fr) == 1) and first[fr[0]] == 1 and last[fr[-1]] == 1
            assert first[fr].sum() == 1 and last[fr].sum() == 1
            ends = list(P["frag_k0"][fr[1:]]) + [n_rows_t[t]]
            assert P["frag_k0"][fr[0]] == 0 and np.all(np.asarray(ends) > P["frag_k0"][fr])
            k1[fr] = ends
            for i, g in enumerate(fr):                      # the lookup table agrees
                assert np.all(P["frag_of"][t, P["frag_k0"][g]:k1[g]] == g)
        for s_ in range(P["n_super"]):
            lo, hi = P["super_frag"][s_], P["super_frag"][s_ + 1]
            cur = 0
            for g in range(lo, hi):
                start = P["frag_off"][g] - (FR_TILE_HEADER_BYTES if first[g] else 0)
                assert start == cur                            # packed back to back, header in front of a first fragment
                cur = P["frag_off"][g] + (k1[g] - P["frag_k0"][g]) * row_bytes
            assert cur <= P["buf_bytes"] and cur <= (P["super_kb"][s_ + 1] - P["super_kb"][s_]) * 1024
        total = int(n_rows_t.sum()) * row_bytes + len(n_rows_t) * FR_TILE_HEADER_BYTES
        if P["resident"]:
            assert P["n_super"] == 1 and len(n_rows_t) <= 64 and P["buf_bytes"] >= total
        else:
            assert P["buf_bytes"] == FR_STREAM_BUF_BYTES and 2 * (2 * P["buf_bytes"] + 8 * 512 + 1024 + 16) <= 160 * 1024


def test_device_merge_equals_the_host_write_back():
    """SlimEngine.merge_fit (device tensors) == merge_coefficients (the LIL write-back restated in numpy): overwrite,
    delete on explicit zero, keep unmentioned entries, grow with the catalogue; and DeviceWeights round-trips to the host."""
    import torch
    from tests.cpu_backend import OracleBackend
    eng = SlimEngine(backend=OracleBackend())
    rng = np.random.default_rng(11)
    W_host, dw, I = None, None, 50
    for step in range(4):
        I += 7 * step                                            # new items appear between fits
        tg = np.sort(rng.choice(I, 12, replace=False)).astype(np.int64)
        cap = 6
        items = np.stack([rng.choice(I, cap, replace=False) for _ in tg]).astype(np.int32)
        coef = rng.uniform(-1, 1, (len(tg), cap)).astype(np.float32)
        coef[rng.random(coef.shape) < 0.4] = 0.0                 # explicit zeros delete
        count = rng.integers(0, cap + 1, len(tg)).astype(np.int32)
        if W_host is not None and W_host.shape[0] != I:
            W_host = W_host.copy(); W_host.resize((I, I))
        W_host = merge_coefficients(W_host, I, *coefficients_to_updates(tg, items, coef, count))
        dw = eng.merge_fit(dw, I, False, torch.from_numpy(tg.astype(np.int32)), torch.from_numpy(items), torch.from_numpy(coef),
                           torch.from_numpy(count))
        got = dw.to_csc(torch)
        assert got.dtype == np.float32 and got.has_sorted_indices and (got != W_host).nnz == 0 and got.nnz == W_host.nnz
        assert np.array_equal(got.indptr, W_host.indptr) and np.array_equal(got.indices, W_host.indices)
    up = eng.upload_weights(W_host)
    assert not up.lossy and up.to_csc(torch) is W_host and np.array_equal(up.vals.numpy(), dw.vals.numpy())
    assert eng.upload_weights(sp.csc_matrix(W_host, dtype=np.float64) * (1.0 / 3.0)).lossy


@pytest.mark.parametrize("compact", [False, True])
def test_tiled_layout_roundtrip(compact):
    W = sp.random(300, 300, density=0.02, random_state=3, format="csc", dtype=np.float32)
    W.sort_indices()
    for lo, hi in ((0, 300), (150, 300)):
        T = build_tiled_w(W, lo, hi, 256, compact=compact)
        rows, cols, vals = [], [], []
        tp = T.tile_ptr.reshape(T.n_tiles, T.n_items + 1)
        for t in range(T.n_tiles):
            for i in range(T.n_items):
                seg = slice(tp[t, i], tp[t, i + 1])
                loc = T.w_col[seg].astype(np.int64) + t * T.tile_cols
                assert np.all(np.diff(loc) > 0)          # ascending columns inside a row
                rows += [i] * len(loc)
                cols += (T.col_ids[loc] if compact else loc + lo).tolist()
                vals += T.w_val[seg].tolist()
        R = sp.csc_matrix((vals, (rows, cols)), shape=(300, 300), dtype=np.float32)
        assert (R != W[:, lo:hi].tocsc().__class__(sp.hstack([sp.csc_matrix((300, lo)), W[:, lo:hi]]))).nnz == 0
        if compact:
            assert np.array_equal(np.flatnonzero(T.col_map >= 0), T.col_ids)
            assert T.n_cols == int(np.count_nonzero(np.diff(W.indptr[lo:hi + 1])))


def test_shard_bounds_and_seed():
    assert [shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_bounds(3, 8, r) for r in range(8)][:4] == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert sklearn_seed(43) == 494155588


def test_store_selected_exports_use_indexes_and_match_full_export():
    """Large store (item-index path), with a populated delta block on top of the base block."""
    rng = np.random.default_rng(4)
    n = 120_000
    u, i = rng.integers(0, 3000, n), rng.integers(0, 900, n)
    ts = 1.7e9 + np.sort(rng.random(n)) * 1e6
    r = rng.integers(1, 6, n).astype(float)
    s = UserItemInteractions(min_value=-5, max_value=50, decay_in_days=30)
    s.add_interactions_batch(u[:100_000], i[:100_000], ts[:100_000], r[:100_000])
    s._compact()
    for a in range(100_000, n, 1000):          # mini-batches land in the delta block
        s.add_interactions_batch(u[a:a + 1000], i[a:a + 1000], ts[a:a + 1000], r[a:a + 1000])
    assert len(s._delta) + len(s._l0) > 0 and len(s._base) > (1 << 15)
    sel_items = [5, 17, 123, 899, 400]
    sel_users = [0, 7, 2999, 1500]
    part_c = s.to_csc(sel_items)
    part_r = s.to_csr(sel_users)
    users_by = s.get_users_by_items(sel_items)
    items_7 = s.get_user_items(7)
    full_c, full_r = s.to_csc(), s.to_csr()          # compacts
    mask_c = np.zeros(full_c.shape[1], bool); mask_c[sel_items] = True
    exp_c = full_c.multiply(sp.csr_matrix(mask_c.astype(np.float32))).tocsc()
    assert (part_c != exp_c).nnz == 0
    mask_r = np.zeros(full_r.shape[0], bool); mask_r[sel_users] = True
    exp_r = sp.diags(mask_r.astype(np.float32)).dot(full_r).tocsr()
    assert (part_r != exp_r).nnz == 0
    assert sorted(users_by) == sorted(np.unique(full_c[:, sel_items].tocoo().row).tolist())
    assert sorted(items_7) == sorted(full_r[7].indices.tolist())


def test_lru_vectorised_add_many_equals_sequential_adds():
    """LRUFreqSet.add_many on an integer array (numpy counts + last-occurrence order) must leave
    exactly the state that one add() per value leaves, including when the capacity overflows."""
    rng = np.random.default_rng(3)
    for cap, n_keys in ((1000, 300), (200, 300), (64, 5000)):
        a, b = LRUFreqSet(cap), LRUFreqSet(cap)
        for _ in range(6):
            vals = rng.integers(0, n_keys, size=int(rng.integers(65, 900)))
            a.add_many(vals)
            for v in vals.tolist():
                b.add(v)
            assert list(a.data.items()) == list(b.data.items())


def test_columnar_ingest_equals_tuple_ingest():
    """Recommender's numeric-DataFrame fast path (columns handed over as arrays in big chunks) must
    leave the model in the state the reference's 1000-tuple mini-batch loop leaves it in."""
    import pandas as pd
    from rtrec_amd.models.slim import SLIM
    from rtrec_amd.recommender import Recommender
    rng = np.random.default_rng(11)
    n = 5000
    df = pd.DataFrame({"user": rng.integers(0, 300, n), "item": rng.integers(0, 120, n),
                       "tstamp": 1.7e9 + np.sort(rng.random(n) * 5e6), "rating": rng.integers(-2, 6, n).astype(float)})
    for kw in ({}, {"decay_in_days": 30}):
        fast = Recommender(SLIM(min_value=-3, max_value=12, **kw))
        fast._ingest_frame(df, 1000, False, True, True)
        slow = Recommender(SLIM(min_value=-3, max_value=12, **kw))
        for batch in Recommender.generate_batches(df[["user", "item", "tstamp", "rating"]], 1000):
            slow.model.add_interactions(batch, update_interaction=False, record_interactions=True)
        A, B = fast.model.interactions, slow.model.interactions
        assert A.shape == B.shape and A.max_timestamp == B.max_timestamp
        Xa, Xb = A.to_csr(), B.to_csr()
        assert np.array_equal(Xa.indptr, Xb.indptr) and np.array_equal(Xa.indices, Xb.indices)
        assert np.array_equal(Xa.data, Xb.data)
        assert list(A.hot_items.data.items()) == list(B.hot_items.data.items())
        assert fast.model.recorded_item_ids == slow.model.recorded_item_ids
        assert fast.model.item_ids.pass_through is True and fast.model.user_ids.pass_through is True
    # string ids and negative ids take the tuple path (same warn-and-skip semantics)
    m = SLIM()
    m.add_interactions_columns(np.array(["a", "b"], dtype=object), np.array(["x", "y"], dtype=object),
                               np.array([1.0, 2.0]), np.array([1.0, 1.0]))
    assert m.item_ids.pass_through is False and m.interactions.nnz == 2
    m2 = SLIM()
    m2.add_interactions_columns(np.array([1, -1]), np.array([2, 3]), np.array([1.0, 2.0]), np.array([1.0, 1.0]))
    assert m2.interactions.nnz >= 1


def test_csc_export_of_a_wide_catalogue_matches_scipy():
    """> 65536 items: the CSC export orders entries by a value sort of (item, position) composites."""
    rng = np.random.default_rng(2)
    n = 20000
    u, i = rng.integers(0, 500, n), rng.integers(0, 100_000, n)
    st = UserItemInteractions(min_value=-5, max_value=10)
    st.add_interactions_batch(u, i, 1.7e9 + np.arange(n, dtype=float), rng.integers(1, 6, n).astype(float))
    C, R = st.to_csc(), st.to_csr()
    ref = R.tocsc()
    ref.sort_indices()
    assert C.has_sorted_indices and np.array_equal(C.indptr, ref.indptr) and np.array_equal(C.indices, ref.indices)
    assert np.array_equal(C.data, ref.data)
    part = st.to_csc(select_items=[int(x) for x in np.unique(i)[:50]])
    assert part.nnz == int(np.isin(R.tocoo().col, np.unique(i)[:50]).sum())


def test_fit_targets_are_dealt_out_by_column_length():
    """owned_columns: disjoint, complete, identical on every rank, and balanced in work (nnz)."""
    from tests.cpu_backend import OracleBackend
    from rtrec_amd.engine import SlimEngine
    from rtrec_amd.synth import interaction_matrix
    X = interaction_matrix(3000, 500, 60000, seed=2)
    Xc = X.tocsc()
    cols = np.arange(500)
    parts, loads = [], []
    for r in range(4):
        eng = SlimEngine(backend=OracleBackend(), rank=r, world_size=4)
        eng.set_interactions(Xc, X)
        mine = eng.owned_columns(cols)
        parts.append(mine)
        loads.append(int(np.diff(Xc.indptr)[mine].sum()))
    assert np.array_equal(np.sort(np.concatenate(parts)), cols)
    assert max(loads) <= 1.1 * min(loads), loads
    sub = np.array([7, 3, 400, 12, 250])
    got = [SlimEngine(backend=OracleBackend(), rank=r, world_size=2) for r in range(2)]
    for g in got:
        g.set_interactions(Xc, X)
    assert np.array_equal(np.sort(np.concatenate([g.owned_columns(sub) for g in got])), np.sort(sub))


@pytest.mark.parametrize("decay", [None, 30])
def test_selected_column_export_through_the_item_major_mirror(decay, monkeypatch):
    """to_csc(select_items) on a large store takes the base block's item-major mirror plus a merge of
    the delta block; it must equal the generic export bit for bit -- through overwritten pairs, new
    pairs, new users and new items, with and without time decay."""
    rng = np.random.default_rng(0)
    st = UserItemInteractions(min_value=0, max_value=15, decay_in_days=decay)
    U, I, n = 3000, 700, 120_000
    u, i = rng.integers(0, U, n), rng.zipf(1.3, n) % I
    ts = 1.7e9 + np.arange(n) * 50.0
    st.add_interactions_batch(u, i, ts, rng.integers(1, 6, n).astype(float))
    st._compact()
    assert len(st._base) >= st._ITEM_MAJOR_MIN
    for step in range(5):
        m = 400 * (step + 1)
        u2, i2 = rng.integers(0, U + 50 * step, m), rng.integers(0, I + 20 * step, m)
        st.add_interactions_batch(u2, i2, ts[-1] + 1000.0 * (step + 1) + np.arange(m), rng.integers(1, 6, m).astype(float))
        items = np.unique(i2).tolist() + [10 ** 6]          # an id the store has never seen is ignored
        fast = st.to_csc(items)
        with monkeypatch.context() as mp:
            mp.setattr(UserItemInteractions, "_ITEM_MAJOR_MIN", 1 << 62)
            generic = st.to_csc(items)
        assert fast.shape == generic.shape and fast.data.dtype == np.float32 and fast.indices.dtype == np.int32
        assert np.array_equal(fast.indptr, generic.indptr)
        assert np.array_equal(fast.indices, generic.indices)
        assert np.array_equal(fast.data, generic.data)
    st._compact()                                            # in-place overwrites invalidate the value mirror
    items = list(range(0, 300, 7))
    fast = st.to_csc(items)
    with monkeypatch.context() as mp:
        mp.setattr(UserItemInteractions, "_ITEM_MAJOR_MIN", 1 << 62)
        generic = st.to_csc(items)
    assert np.array_equal(fast.indptr, generic.indptr) and np.array_equal(fast.indices, generic.indices)
    assert np.array_equal(fast.data, generic.data)


def _assert_same_matrix(dev, host, fmt):
    ptr, idx, val = ("rptr", "rcol", "rval") if fmt == "csr" else ("cptr", "crow", "cval")
    assert np.array_equal(dev[ptr].numpy(), host.indptr)
    assert np.array_equal(dev[idx].numpy(), host.indices)
    assert np.array_equal(dev[val].numpy(), host.data)
    assert dev[ptr].numpy().dtype == np.int32 and dev[idx].numpy().dtype == np.int32 and dev[val].numpy().dtype == np.float32


@pytest.mark.parametrize("upsert", [False, True])
def test_device_resident_store_tracks_the_host_store(upsert):
    """utils/device_store.py (run here on CPU tensors): after any sequence of mini-batches the
    resident sorted-COO copy yields exactly the host store's CSR / CSC exports, and partial(items)
    exactly to_csc(items) / the CSR restricted to those columns -- through overwritten pairs, new
    pairs, new users and items, duplicates inside a batch and a rebuild from a host export."""
    import torch
    from rtrec_amd.utils.device_store import DeviceInteractions
    rng = np.random.default_rng(3)
    st = UserItemInteractions(min_value=0, max_value=15)
    mir = DeviceInteractions(torch, torch.device("cpu"))
    U, I = 400, 90
    csr = st.to_csr()
    mir.load_csr(csr.indptr, csr.indices, csr.data, csr.shape[0], csr.shape[1], st.version)      # empty store
    t = 1.7e9
    for step in range(8):
        m = 300 + 100 * step
        u = rng.integers(0, U + 30 * step, m)
        i = rng.zipf(1.4, m) % (I + 10 * step)
        r = rng.integers(-2, 6, m).astype(float) if upsert else rng.integers(1, 6, m).astype(float)
        st.add_interactions_batch(u, i, t + np.arange(m), r, upsert=upsert)
        t += m
        keys = np.unique(st._keys(u, i))
        _, val, _ = st._lookup(keys)
        if step == 4:            # fell behind: rebuilt from a full export instead
            csr = st.to_csr()
            mir.load_csr(csr.indptr, csr.indices, csr.data, csr.shape[0], csr.shape[1], st.version)
        else:
            mir.apply(keys >> 32, keys & 0xFFFFFFFF, val.astype(np.float32), st.shape[0], st.shape[1], st.version)
        assert mir.version == st.version and (mir.n_users, mir.n_items) == st.shape and mir.nnz == st.nnz
        full = mir.full()
        _assert_same_matrix(full, st.to_csr(), "csr")
        _assert_same_matrix(full, st.to_csc(), "csc")
        assert np.array_equal(full["col_nnz"], np.diff(st.to_csc().indptr))
        assert full["nonneg"] == bool(st.to_csc().data.min() >= 0)
        items = np.unique(i).tolist() + [10 ** 6]
        part = mir.partial(np.asarray(items))
        host_csc = st.to_csc(items)
        _assert_same_matrix(part, host_csc, "csc")
        host_csr = host_csc.tocsr()
        host_csr.sort_indices()
        _assert_same_matrix(part, host_csr, "csr")
        assert np.array_equal(part["col_nnz"], np.diff(host_csc.indptr))
    # adopt(): the engine's uploaded arrays become the resident copy
    full = mir.full()
    other = DeviceInteractions(torch, torch.device("cpu"))
    other.adopt(full, mir.n_users, mir.n_items, st.version)
    _assert_same_matrix(other.full(), st.to_csr(), "csr")
    _assert_same_matrix(other.full(), st.to_csc(), "csc")
    # load_csc(): from a host CSC export (the engine's path for host-built matrices), CSR by a device sort
    c = st.to_csc()
    third = DeviceInteractions(torch, torch.device("cpu"))
    third.load_csc(c.indptr, c.indices, c.data, c.shape[0], c.shape[1], st.version)
    _assert_same_matrix(third.full(), st.to_csr(), "csr")
    _assert_same_matrix(third.full(), c, "csc")


def test_store_levels_mini_batches_equal_one_bulk_write(monkeypatch):
    """The store keeps three sorted levels (recent-writes l0 -> delta -> base).  Feeding the same
    interactions as hundreds of mini-batches (l0 spills into the delta block several times, the delta
    is compacted into the base block) must leave exactly the state of a few bulk writes: same exports,
    same point lookups while entries still sit in l0."""
    from rtrec_amd.utils import interactions as mod
    monkeypatch.setattr(mod, "_L0_MAX", 1 << 12)
    monkeypatch.setattr(mod, "_DELTA_MERGE_MIN", 1 << 13)
    rng = np.random.default_rng(8)
    n = 150_000
    u, i = rng.integers(0, 4000, n), rng.zipf(1.3, n) % 1200
    ts = 1.7e9 + np.arange(n, dtype=float)
    r = rng.integers(1, 6, n).astype(float)
    a_store = UserItemInteractions(min_value=0, max_value=12)
    b_store = UserItemInteractions(min_value=0, max_value=12)
    a_store.add_interactions_batch(u[:70_000], i[:70_000], ts[:70_000], r[:70_000])
    b_store.add_interactions_batch(u[:70_000], i[:70_000], ts[:70_000], r[:70_000])
    spills, compactions = 0, 0
    for s in range(70_000, n, 500):
        before, base_before = len(a_store._l0), len(a_store._base)
        a_store.add_interactions_batch(u[s:s + 500], i[s:s + 500], ts[s:s + 500], r[s:s + 500])
        spills += len(a_store._l0) < before
        compactions += len(a_store._base) > base_before
        if s % 20_000 == 0:                       # point lookups see l0 without flushing it
            l0_before = len(a_store._l0)
            for k in range(s, s + 500, 97):
                assert a_store.has_interaction(int(u[k]), int(i[k]))
            assert len(a_store._l0) == l0_before
    assert spills >= 5 and compactions >= 2 and len(a_store._l0) < mod._L0_MAX
    b_store.add_interactions_batch(u[70_000:], i[70_000:], ts[70_000:], r[70_000:])
    for fmt in ("to_csr", "to_csc"):
        A, B = getattr(a_store, fmt)(), getattr(b_store, fmt)()
        assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data)
    assert a_store.nnz == b_store.nnz and a_store.max_timestamp == b_store.max_timestamp
    for k in range(0, n, 7919):
        assert a_store.get_user_item_rating(int(u[k]), int(i[k])) == b_store.get_user_item_rating(int(u[k]), int(i[k]))
    assert sorted(a_store.get_user_items(int(u[-1]))) == sorted(b_store.get_user_items(int(u[-1])))


def test_native_block_merge_equals_numpy_merge(monkeypatch):
    """rtrec_store_merge_sorted (host routine of librtrec_amd.so, threaded two-pointer merge) against the
    numpy merge of the store: overlapping keys (the new block wins), disjoint keys, one-sided tails."""
    from rtrec_amd.utils import interactions as mod
    rng = np.random.default_rng(12)

    def block(keys):
        keys = np.unique(keys.astype(np.int64))
        return mod._Block(keys, rng.random(len(keys)), rng.random(len(keys)) * 1e9)

    cases = [(rng.integers(0, 1 << 40, 200_000), rng.integers(0, 1 << 40, 150_000)),          # almost disjoint
             (rng.integers(0, 300_000, 200_000), rng.integers(0, 300_000, 90_000)),            # heavy overlap
             (np.arange(100_000), np.arange(100_000, 180_000)),                                # b entirely above a
             (np.arange(50_000, 120_000), np.arange(0, 60_000))]                               # b below and into a
    for ka, kb in cases:
        a1, b1 = block(ka), block(kb)
        a2 = mod._Block(a1.key.copy(), a1.val.copy(), a1.ts.copy())
        native = mod._merge_blocks(a1, b1)
        assert mod._native_merge not in (None, False), "librtrec_amd.so must export rtrec_store_merge_sorted"
        with monkeypatch.context() as mp:
            mp.setattr(mod, "_NATIVE_MERGE_MIN", 1 << 62)
            plain = mod._merge_blocks(a2, b1)
        assert np.array_equal(native.key, plain.key) and np.array_equal(native.val, plain.val)
        assert np.array_equal(native.ts, plain.ts)
        assert np.all(np.diff(native.key) > 0)


def test_native_sorted_lookup_equals_searchsorted(monkeypatch):
    """rtrec_store_find_sorted (galloping search for ascending needles) against numpy's searchsorted path
    of _Block.find: hits, misses below / between / above the block, repeated needles."""
    from rtrec_amd.utils import interactions as mod
    rng = np.random.default_rng(13)
    hay = np.unique(rng.integers(1000, 1 << 34, 300_000).astype(np.int64))
    blk = mod._Block(hay, np.zeros(len(hay)), np.zeros(len(hay)))
    for needles in (np.sort(np.concatenate([rng.choice(hay, 40_000), rng.integers(0, 1 << 35, 40_000)])).astype(np.int64),
                    np.sort(rng.choice(hay[:5000], 30_000)).astype(np.int64),                       # dense repeats
                    np.concatenate([np.arange(20_000), np.arange(1 << 36, (1 << 36) + 20_000)]).astype(np.int64)):
        f1, p1 = blk.find(needles)
        with monkeypatch.context() as mp:
            mp.setattr(mod, "_NATIVE_FIND_MIN", 1 << 62)
            f2, p2 = blk.find(needles)
        assert mod._native_lib() is not None
        assert np.array_equal(f1, f2) and np.array_equal(p1, p2) and f1.dtype == np.bool_


def test_native_lru_replay_equals_sequential_adds():
    """Batches that overflow the capacity are replayed by rtrec_lru_replay (host routine of
    librtrec_amd.so); the end state -- keys, hit counts, recency order -- must be the one a loop of add()
    leaves, across several batches on a warm set."""
    rng = np.random.default_rng(21)
    a, b = LRUFreqSet(capacity=500), LRUFreqSet(capacity=500)
    for step in range(6):
        vals = (rng.zipf(1.2, 5000) % 3000).astype(np.int64) if step % 2 == 0 else rng.integers(0, 3000, 4000)
        for v in vals.tolist():
            a.add(v)
        b.add_many(vals)
        assert list(a.data.items()) == list(b.data.items())
        assert all(isinstance(k, int) for k in b.data)
    assert len(b) == 500
    assert list(a.get_freq_items(20)) == list(b.get_freq_items(20))


@pytest.mark.parametrize("upsert", [False, True])
def test_native_ingest_round_equals_numpy_round(upsert, monkeypatch):
    """Large batches apply their rounds through rtrec_store_apply_round (gather + add + clip in one
    threaded pass); the store must end up exactly as through the numpy expressions -- with repeated pairs
    inside the batch (several rounds), clipping at both ends and values already in the store."""
    from rtrec_amd.utils import interactions as mod
    rng = np.random.default_rng(31)
    n = 200_000
    u, i = rng.integers(0, 3000, n), rng.zipf(1.3, n) % 500          # many repeated pairs
    r = rng.integers(-4, 9, n).astype(float)
    ts = 1.7e9 + np.arange(n, dtype=float)[::-1].copy()               # not monotone either
    stores = []
    for native in (True, False):
        with monkeypatch.context() as mp:
            if not native:
                mp.setattr(mod, "_NATIVE_APPLY_MIN", 1 << 62)
            st = UserItemInteractions(min_value=-3, max_value=10)
            st.add_interactions_batch(u[:120_000], i[:120_000], ts[:120_000], r[:120_000], upsert=upsert)
            st.add_interactions_batch(u[120_000:], i[120_000:], ts[120_000:], r[120_000:], upsert=upsert)
            stores.append(st)
    a, b = stores
    A, B = a.to_csr(), b.to_csr()
    assert np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices) and np.array_equal(A.data, B.data)
    blk_a, blk_b = a._compact(), b._compact()
    assert np.array_equal(blk_a.val, blk_b.val) and np.array_equal(blk_a.ts, blk_b.ts)
    assert a.max_timestamp == b.max_timestamp
    if not upsert:          # an upsert stores the rating as it is
        assert A.data.min() >= -3 and A.data.max() <= 10


@pytest.mark.parametrize("kw,upsert", [({}, False), ({"decay_in_days": 7}, False), ({}, True)])
def test_heavily_repeated_pairs_fold_sequentially_and_stay_linear(kw, upsert):
    """A pair repeated far more often than the vectorised rounds cover (a bot / a replayed stream): the
    rest of its sequence is folded one interaction after the other, and the store must end up exactly as
    if every interaction had been added on its own (the reference's loop, interactions.py:81-119) --
    in time linear in the batch (the per-round masks made it O(n x repeats))."""
    import time as _time
    rng = np.random.default_rng(5)
    n = 3000
    u, i = rng.integers(0, 40, n), rng.integers(0, 30, n)
    hot = rng.random(n) < 0.4
    u[hot], i[hot] = 7, 3                                  # ~1200 occurrences of one pair
    u[rng.random(n) < 0.1], i[rng.random(n) < 0.1] = 8, 4
    r = rng.integers(-3, 6, n).astype(float)
    ts = 1.7e9 + np.sort(rng.random(n)) * 40 * 86400.0
    batch = UserItemInteractions(min_value=-2, max_value=9, **kw)
    batch.add_interactions_batch(u[:100], i[:100], ts[:100], r[:100], upsert=upsert)
    batch.add_interactions_batch(u[100:], i[100:], ts[100:], r[100:], upsert=upsert)
    single = UserItemInteractions(min_value=-2, max_value=9, **kw)
    for k in range(n):
        single.add_interaction(int(u[k]), int(i[k]), float(ts[k]), float(r[k]), upsert=upsert)
    a, b = batch._compact(), single._compact()
    assert np.array_equal(a.key, b.key) and np.array_equal(a.ts, b.ts)
    assert np.array_equal(bits64(a.val), bits64(b.val))
    assert batch.max_timestamp == single.max_timestamp
    # linear cost: one pair repeated 200,000 times in a 400,000-row batch
    m = 400_000
    uu, ii = rng.integers(0, 5000, m), rng.integers(0, 800, m)
    uu[::2], ii[::2] = 1, 1
    st = UserItemInteractions(min_value=-5, max_value=10)
    t0 = _time.perf_counter()
    st.add_interactions_batch(uu, ii, 1.7e9 + np.arange(m, dtype=float), np.ones(m))
    assert _time.perf_counter() - t0 < 5.0
    assert st.get_user_item_rating(1, 1) == 10.0


def numpy_fold(torch):
    """fold_fn for DeviceInteractions.ingest on CPU tensors: the per-pair loop of rtrec_store_fold_device in Python (test
    stand-in for HipBackend.fold_pairs; the HIP kernel itself is checked by tests/test_gpu_api.py)."""
    def fold(order, start, delta, tstamp, old, lo, hi, upsert):
        o, s, d, t = order.numpy(), start.numpy(), delta.numpy(), tstamp.numpy()
        g = len(s) - 1
        val, ts = np.zeros(g), np.zeros(g)
        for k in range(g):
            v = 0.0 if old is None else float(old[k])
            for q in range(s[k], s[k + 1]):
                v = float(d[o[q]]) if upsert else max(lo, min(v + float(d[o[q]]), hi))
            val[k], ts[k] = v, t[o[s[k + 1] - 1]]
        return torch.from_numpy(val), torch.from_numpy(ts), torch.from_numpy(val.astype(np.float32))
    return fold


def bulk_batches(seed=41, n=30_000, n_users=900, n_items=260):
    rng = np.random.default_rng(seed)
    u, i = rng.integers(0, n_users, n), rng.zipf(1.4, n) % n_items          # many repeated pairs
    r = rng.integers(-4, 9, n).astype(float)
    r[rng.random(n) < 0.01] = np.nan            # max(lo, min(nan, hi)) is lo in Python: the reference stores min_value
    ts = 1.7e9 + rng.permutation(n).astype(float)
    return u, i, ts, r


@pytest.mark.parametrize("upsert", [False, True])
def test_device_bulk_ingest_equals_sequential_adds(upsert, monkeypatch):
    """A bulk batch reduced on the device (sort by (user, item, arrival), runs, per-pair fold, hot-item counts:
    DeviceInteractions.ingest) leaves the store exactly as one add_interaction per row does -- into an empty store and on
    top of stored values, with NaN ratings, clipping at both ends and the composite sort key as well as its fallback."""
    import torch
    from rtrec_amd.utils import interactions as mod
    from rtrec_amd.utils.device_store import DeviceInteractions
    monkeypatch.setattr(mod, "_DEVICE_FOLD_MIN", 1000)
    u, i, ts, r = bulk_batches()
    cut = 18_000
    for wide_ids in (False, True):
        uu = u + (1 << 31) if wide_ids else u             # user ids too wide for the 63-bit composite: stable key sort instead
        mir = DeviceInteractions(torch, "cpu")
        fold = lambda *a: mir.ingest(*a, numpy_fold(torch))
        dev = UserItemInteractions(min_value=-3, max_value=10, n_recent_hot=200)
        dev.add_interactions_batch(uu[:cut], i[:cut], ts[:cut], r[:cut], upsert=upsert, device_fold=fold)
        assert mir.ingested is not None and len(mir.ingested["keys"]) == len(dev._compact())
        dev.add_interactions_batch(uu[cut:], i[cut:], ts[cut:], r[cut:], upsert=upsert, device_fold=fold)
        seq = UserItemInteractions(min_value=-3, max_value=10, n_recent_hot=200)
        step = 1 if not wide_ids else 997           # the per-row loop once; the host batch path (itself pinned to it) after that
        for a in range(0, len(u), step):
            seq.add_interactions_batch(uu[a:a + step], i[a:a + step], ts[a:a + step], r[a:a + step], upsert=upsert)
        a, b = dev._compact(), seq._compact()
        assert np.array_equal(a.key, b.key) and np.array_equal(a.ts, b.ts)
        assert np.array_equal(bits64(a.val), bits64(b.val))
        assert dev.max_timestamp == seq.max_timestamp and dev.shape == seq.shape
        assert dev.all_item_ids == seq.all_item_ids
        assert list(dev.hot_items.data.items()) == list(seq.hot_items.data.items())


def test_nan_rating_is_stored_as_min_value_like_the_reference():
    """interactions.py:106: max(self.min_value, min(new_value, self.max_value)) with Python's min / max -- a NaN sum
    compares false both times and min_value comes out (numpy.clip would keep the NaN)."""
    for n in (1, 100):          # the per-interaction path and a vectorised round
        st = UserItemInteractions(min_value=-2, max_value=5)
        st.add_interactions_batch(np.arange(n), np.zeros(n, np.int64), np.full(n, 1.7e9), np.full(n, np.nan))
        assert st.get_user_item_rating(0, 0) == -2.0
        st.add_interactions_batch(np.arange(n), np.zeros(n, np.int64), np.full(n, 1.7e9), np.full(n, 4.0))
        assert st.get_user_item_rating(0, 0) == 2.0


def bits64(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def test_bad_rows_in_an_otherwise_clean_batch_are_skipped_like_the_reference():
    """base.py:86-94 of the reference: an interaction that raises is logged and skipped, the rest of
    the batch is stored.  The vectorised batch path must not lose the batch: an unhashable
    id and a numeric STRING rating (which numpy would happily parse) each cost one row."""
    from rtrec_amd.models.slim import SLIM
    m = SLIM()
    rows = [("u1", "a", 1.7e9, 1.0), ("u2", ["x", "y"], 1.7e9, 2.0), ("u3", "b", 1.7e9, "3.5"), ("u1", "b", 1.7e9, 4.0)]
    m.add_interactions(rows)
    st = m.interactions
    assert st.nnz == 2
    assert m.user_ids.obj_to_id == {"u1": 0, "u2": 1, "u3": 2}        # registered in row order, like the reference
    assert m.item_ids.obj_to_id == {"a": 0, "b": 1}
    assert st.get_user_item_rating(0, 0) == 1.0 and st.get_user_item_rating(0, 1) == 4.0


def test_ranking_metrics_match_reference_goldens():
    """rtrec/utils/metrics.py:5-313 captured by tools/gen_golden.py (evaluate.json): every metric on 156
    (ranked, truth) pairs incl. the empty-list corner cases at sizes 1 / 5 / 10, and compute_scores' means."""
    g = json.load(open(os.path.join(G, "evaluate.json")))
    pairs = [(r, t) for r, t in g["pairs"]]
    for size in (1, 5, 10):
        for (r, t), want in zip(pairs, g["per_pair"][str(size)]):
            got = [float(getattr(metrics, nm)(r, t, size)) for nm in g["metric_names"]]
            assert got == pytest.approx(want, rel=1e-12, abs=1e-15), (r, t, size)
        agg = metrics.compute_scores(iter(pairs), size)
        ref = g["aggregate"][str(size)]
        assert set(agg) == set(ref) and agg["tp"] == ref["tp"]
        for k in ref:
            assert agg[k] == pytest.approx(ref[k], rel=1e-12), (k, size)
    assert metrics.mrr([r for r, _ in pairs], [t for _, t in pairs], 5) == pytest.approx(g["aggregate"]["mrr_5"], rel=1e-12)
    assert metrics.map_score([r for r, _ in pairs], [t for _, t in pairs], 5) == pytest.approx(g["aggregate"]["map_5"], rel=1e-12)


def test_device_resident_store_with_time_decay_tracks_the_host_store():
    """The resident store keeps raw values + timestamps for a store with time decay and re-values X whenever
    max_timestamp moves (here on CPU tensors, through the host's libm routine): after every mini-batch its CSR / CSC
    and the touched-columns matrix equal the host exports bit for bit -- also across a rebuild from the store block."""
    import torch
    from rtrec_amd.utils.device_store import DeviceInteractions
    rng = np.random.default_rng(8)
    st = UserItemInteractions(min_value=0, max_value=15, decay_in_days=30)
    mir = DeviceInteractions(torch, torch.device("cpu"))
    blk = st._compact()
    mir.load_store(blk.key, blk.val, blk.ts, st.shape[0], st.shape[1], (st.version, st.max_timestamp), rate=st.decay_rate,
                   now=st.max_timestamp)
    t = 1.7e9
    for step in range(7):
        m = 300 + 80 * step
        u = rng.integers(0, 300 + 30 * step, m)
        i = rng.zipf(1.4, m) % (80 + 10 * step)
        r = rng.integers(1, 6, m).astype(float)
        st.add_interactions_batch(u, i, t + np.sort(rng.random(m)) * 5 * 86400.0, r)
        t += 5 * 86400.0
        keys = np.unique(st._keys(u, i))
        _, val, ts = st._lookup(keys)
        tag = (st.version, st.max_timestamp)
        if step == 3:
            blk = st._compact()
            mir.load_store(blk.key, blk.val, blk.ts, st.shape[0], st.shape[1], tag, rate=st.decay_rate, now=st.max_timestamp)
        else:
            mir.apply(keys >> 32, keys & 0xFFFFFFFF, val, st.shape[0], st.shape[1], tag, tstamps=ts, now=st.max_timestamp)
        assert mir.version == tag and mir.nnz == st.nnz
        full = mir.full()
        _assert_same_matrix(full, st.to_csr(), "csr")
        _assert_same_matrix(full, st.to_csc(), "csc")
        items = np.unique(i).tolist()
        part = mir.partial(np.asarray(items))
        _assert_same_matrix(part, st.to_csc(items), "csc")


def test_recommend_coalescer_promotion_failure_and_isolated_requests():
    """RecommendCoalescer (serving/app.py): an isolated request does not pay the bounded wait; concurrent callers share
    launches and a follower woken without an answer leads the next round; a failing model call reaches every caller of that
    round and the coalescer keeps serving afterwards."""
    import threading
    import time
    from rtrec_amd.serving.app import RecommendCoalescer

    class FakeModel:
        def __init__(self):
            self.fail = False
            self.calls = []

        def recommend(self, user, top_k=10, filter_interacted=True):
            if self.fail:
                raise RuntimeError("boom")
            time.sleep(0.005)
            return [user, top_k]

        def recommend_batch(self, users, top_k=10, filter_interacted=True):
            self.calls.append(list(users))
            if self.fail:
                raise RuntimeError("boom")
            time.sleep(0.005)
            return [[u, top_k] for u in users]

    class Gate:
        def __init__(self, model):
            self.model = model
            self._lock = threading.RLock()

    m = FakeModel()
    co = RecommendCoalescer(Gate(m), max_wait_s=0.05, max_batch=4)
    co._known = lambda model, user: True
    t0 = time.perf_counter()
    assert co.submit(7, 5, True) == [7, 5]
    assert time.perf_counter() - t0 < 0.04, "an isolated request must not wait for company"
    out, errs = {}, []

    def call(u):
        try:
            out[u] = co.submit(u, 3, True)
        except Exception as e:      # noqa: BLE001
            errs.append(e)

    th = [threading.Thread(target=call, args=(u,)) for u in range(10)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=10)
    assert not errs and out == {u: [u, 3] for u in range(10)}
    assert all(len(c) <= 4 for c in m.calls) and sum(len(c) for c in m.calls if len(c) > 1) >= 4      # batches were shared, several rounds led
    m.fail = True
    th = [threading.Thread(target=call, args=(100 + u,)) for u in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=10)
    assert len(errs) == 3 and all(isinstance(e, RuntimeError) for e in errs)
    m.fail = False
    assert co.submit(9, 2, False) == [9, 2]


def test_every_environment_switch_is_in_the_settings_table():
    """rtrec_amd/settings.py is the one table of RTREC_AMD_* switches: no module of the package reads the environment for one
    directly, and every name the code asks for is documented there."""
    import glob
    import re
    from rtrec_amd import settings
    root = os.path.join(os.path.dirname(os.path.dirname(__file__)), "rtrec_amd")
    asked = set()
    for path in glob.glob(os.path.join(root, "**", "*.py"), recursive=True):
        src = open(path).read()
        if not path.endswith("settings.py"):
            assert not re.search(r'os\.(environ\.get|getenv|environ\[)\(?"RTREC_AMD_', src), path
        asked |= set(re.findall(r'settings\.raw\("(RTREC_AMD_[A-Z0-9_]+)"', src))
    assert asked and asked <= set(settings.TABLE)
    assert set(settings.TABLE) <= asked | {"RTREC_AMD_LIB"}
    with pytest.raises(KeyError):
        settings.raw("RTREC_AMD_NO_SUCH_SWITCH")
    assert "RTREC_AMD_SG_FORK" in settings.describe()


def test_grouped_work_order_spreads_giant_rows_over_the_head():
    """engine.spread_giant_rows (the feature-row kernel's pattern-grouped order): rows of more than `giant_len` entries, at most
    `max_giants` of them, longest first, at positions 0, 8, 16, ... -- a wave takes consecutive positions and sets its users up
    one after the other -- the gaps filled from the tail, everything else in place; always a permutation."""
    import torch
    from rtrec_amd.engine import spread_giant_rows
    rng = np.random.default_rng(1)
    n = 5000
    lens = torch.from_numpy(rng.integers(1, 300, n))
    giants = [7, 4000, 123, 4999]
    lens[giants] = torch.tensor([90_000, 50_000, 7_000, 4_097])
    order = torch.from_numpy(rng.permutation(n))
    out = spread_giant_rows(torch, order, lens, 32, 4096)
    assert sorted(out.tolist()) == list(range(n))
    assert out[0:32:8].tolist() == giants                                   # longest first, one per 8 positions
    rest = [x for x in order.tolist() if x not in giants]
    assert out[32:].tolist() == rest[:len(rest) - 28]                         # the body keeps its order ...
    assert sorted(out[:32].tolist()) == sorted(giants + rest[len(rest) - 28:])   # ... the gaps hold the tail
    # at most max_giants; none above the threshold: unchanged; tiny row sets: unchanged
    assert spread_giant_rows(torch, order, lens, 2, 4096)[0:16:8].tolist() == giants[:2]
    assert torch.equal(spread_giant_rows(torch, order, lens, 32, 100_000), order)
    assert torch.equal(spread_giant_rows(torch, order[:40], lens, 32, 4096), order[:40])
    assert torch.equal(spread_giant_rows(torch, order, lens, 0, 4096), order)


def test_bench_attaches_a_profile_summary_only_for_the_build_it_times(tmp_path):
    """a rocprofv3 summary rides on the bench line only when its "build".lib_sha256 equals the
    library being timed; a summary of another build is reported as stale, one of another kernel is ignored."""
    import json
    import bench
    fp = {"lib_sha256": "a" * 64, "git_head": "deadbeef"}
    prof = tmp_path / "profiles"
    prof.mkdir()
    fresh = prof / "r05_c3_pmc_traffic.json"
    fresh.write_text(json.dumps({"build": fp, "kernel": "score_frows_kernel<4,2,4>", "hbm_bytes_per_launch_corrected": 123}))
    stale = prof / "r04_c3_pmc_traffic.json"
    stale.write_text(json.dumps({"build": {"lib_sha256": "b" * 64}, "kernel": "score_frows_kernel<4,2,4>", "hbm_bytes_per_launch_corrected": 9}))
    unstamped = prof / "r03_c3_pmc_traffic.json"
    unstamped.write_text(json.dumps({"kernel": "score_frows_kernel<4,2,4>", "hbm_bytes_per_launch_corrected": 9}))
    other = prof / "r05_c3_score_counters.json"
    other.write_text(json.dumps({"build": fp, "kernel": "score_seg_kernel<8,unsigned short,true>", "per_launch": {}}))
    j, st = bench.attach_profile(str(fresh), "score_frows_kernel<4,2,4>", fp, root=str(tmp_path))
    assert j["hbm_bytes_per_launch_corrected"] == 123 and j["source"] == os.path.join("profiles", "r05_c3_pmc_traffic.json") and st is None
    for path in (stale, unstamped):
        j, st = bench.attach_profile(str(path), "score_frows_kernel<4,2,4>", fp, root=str(tmp_path))
        assert j is None and st == os.path.join("profiles", path.name)
    assert bench.attach_profile(str(other), "score_frows_kernel<4,2,4>", fp, root=str(tmp_path)) == (None, None)     # another kernel
    assert bench.attach_profile(None, "k", fp) == (None, None) and bench.attach_profile(str(prof / "missing.json"), "k", fp) == (None, None)
    # the running build's own fingerprint names the library file that is loaded
    from rtrec_amd import build
    f = build.fingerprint()
    assert f["lib_sha256"] == build._sha256(build.LIB_PATH) and f["src_sha256"] == build.source_sha256()
    # a rebuild of the same tree (other library bytes, same sources + flags) still matches; another tree does not
    assert build.same_build({"lib_sha256": "x", "src_sha256": f["src_sha256"]}, f)
    assert not build.same_build({"lib_sha256": "x", "src_sha256": "y"}, f) and not build.same_build({}, f) and not build.same_build(None, None)


# ---- the scoring dispatch as pure functions (rtrec_amd/score_plan.py): explicit table + invariants over the cross product ----
def _facts(**kw):
    from rtrec_amd import score_plan as sp_
    d = dict(mode=sp_.TOPK_SPARSE, hip=True, acc_f64=False, top_k=10, n_rows=100_000, full_range=True, nonempty_shard=True,
             dense_fast_on=True, dense_fill_on=True, lazy_tiled=True, feature_rows_on=True, seg_layout_on=True, seg_supported=True,
             dense_fill_ok=lambda: True, f64_w_ok=lambda: True, f64_x_ok=lambda: True, f64_refine_mode=lambda: 1)
    for k in ("dense_fill_ok", "f64_w_ok", "f64_x_ok", "f64_refine_mode"):
        if k in kw and not callable(kw[k]):
            v = kw[k]
            kw[k] = (lambda v=v: v)
    d.update(kw)
    return sp_.ScoreFacts(**d)


def _plan(has_fr=True, has_sg=False, tiled=False, **kw):
    from rtrec_amd import score_plan as sp_
    f = _facts(**kw)
    req = sp_.plan_fast_layout(f)
    got_fr, got_sg = (has_fr, has_sg) if req.want else (False, False)
    if req.want and req.small:          # the request-sized form is the segment form
        got_fr, got_sg = False, True
    return req, sp_.choose_path(f, req, got_fr, got_sg, tiled)


def test_score_plan_explicit_table():
    """One row per situation the engine documents: (inputs) -> (path, kernel, list length, flags)."""
    from rtrec_amd.score_plan import Path, TOPK_CANDIDATES as C, TOPK_DENSE as D, TOPK_SPARSE as S
    T = [
        # --- SPARSE, float32 W
        (dict(), dict(), (Path.FAST, "feature_rows", 10, True)),                                  # bulk pass, feature-row W, tiled layout not built
        (dict(), dict(tiled=True), (Path.FAST, "feature_rows", 10, False)),                        # ... tiled layout at hand: one launch, tie pass inside
        (dict(), dict(has_fr=False, has_sg=True), (Path.FAST, "segments", 10, True)),              # general W: segments
        (dict(n_rows=100), dict(), (Path.FAST, "segments", 10, True)),                             # request-sized batch: segment form even of a feature-row W
        (dict(top_k=20), dict(), (Path.FAST, "segments", 20, True)),                               # top_k beyond the feature-row lists
        (dict(top_k=100), dict(), (Path.TILED, "tiled", 100, False)),                              # beyond both fast kernels
        (dict(top_k=20, seg_layout_on=False), dict(), (Path.TILED, "tiled", 20, False)),
        (dict(feature_rows_on=False), dict(has_fr=True, has_sg=False), (Path.TILED, "tiled", 10, False)),
        (dict(hip=False), dict(), (Path.TILED, "tiled", 10, False)),                               # the CPU stand-in backend
        (dict(nonempty_shard=False), dict(), (Path.TILED, "tiled", 10, False)),                    # an empty column shard
        (dict(lazy_tiled=False), dict(), (Path.FAST, "feature_rows", 10, False)),                  # laziness off: tiled layout built with W
        (dict(full_range=False), dict(), (Path.FAST, "feature_rows", 10, True)),                   # SPARSE on a column shard is fine
        # --- SPARSE, float64 W
        (dict(acc_f64=True), dict(), (Path.FAST_F64, "feature_rows", 11, True)),                   # positive W and X: fast pass for k + 1, refine
        (dict(acc_f64=True, f64_x_ok=False, f64_refine_mode=1), dict(), (Path.FAST_F64, "feature_rows", 11, True)),   # signed ratings: slack form
        (dict(acc_f64=True, f64_w_ok=False, f64_refine_mode=2), dict(has_fr=False, has_sg=True), (Path.FAST_F64, "segments", 11, True)),
        (dict(acc_f64=True, f64_w_ok=False, f64_refine_mode=0), dict(), (Path.TILED, "tiled", 10, False)),            # lossy float64 W
        (dict(acc_f64=True, lazy_tiled=False), dict(), (Path.TILED, "tiled", 10, False)),
        (dict(acc_f64=True, top_k=15), dict(), (Path.FAST_F64, "segments", 16, True)),             # k + 1 no longer fits the feature-row lists
        (dict(acc_f64=True, top_k=63), dict(), (Path.TILED, "tiled", 63, False)),                  # k + 1 = 64 fits neither
        # --- DENSE (string ids)
        (dict(mode=D), dict(), (Path.FAST, "feature_rows", 10, True)),                             # all columns on this rank: fast pass + flagged rows
        (dict(mode=D), dict(tiled=True), (Path.FAST, "feature_rows", 10, True)),                   # ... dense_fast always takes the lazy form
        (dict(mode=D, full_range=False), dict(), (Path.FAST, "feature_rows", 10, True)),           # column shard: completed in place (dense_fill)
        (dict(mode=D, full_range=False, dense_fill_ok=False), dict(), (Path.TILED, "tiled", 10, False)),
        (dict(mode=D, full_range=False, dense_fill_on=False), dict(), (Path.TILED, "tiled", 10, False)),
        (dict(mode=D, dense_fast_on=False), dict(), (Path.TILED, "tiled", 10, False)),
        (dict(mode=D, acc_f64=True), dict(), (Path.FAST_F64, "feature_rows", 11, True)),
        (dict(mode=D, acc_f64=True, f64_w_ok=False, f64_refine_mode=2), dict(), (Path.TILED, "tiled", 10, False)),   # signed W: DENSE keeps the tiled kernel
        (dict(mode=D, acc_f64=True, full_range=False), dict(), (Path.TILED, "tiled", 10, False)),  # no dense_fill for a float64 W
        # --- CANDIDATES (bulk form): always the tiled kernel
        (dict(mode=C), dict(), (Path.TILED, "tiled", 10, False)),
        (dict(mode=C, acc_f64=True), dict(), (Path.TILED, "tiled", 10, False)),
    ]
    for kw, lay, want in T:
        req, plan = _plan(**lay, **kw)
        assert (plan.path, plan.kernel, plan.k_fast, plan.lazy) == want, (kw, lay, plan)
    # the dense_fill flag travels with the plan only where the fill kernel must run
    assert _plan(mode=D, full_range=False)[1].fill and _plan(mode=D)[1].fill                     # (fill also completes short lists on a full range)
    assert not _plan(mode=D, dense_fill_on=False)[1].fill and not _plan()[1].fill
    assert _plan(acc_f64=True, f64_x_ok=False, f64_refine_mode=1)[1].f64_signed and not _plan(acc_f64=True)[1].f64_signed


def test_score_plan_invariants_over_the_cross_product():
    """Every combination of the inputs: the plan is internally consistent and never asks for something the layouts, the
    backend or the kernels' list limits cannot deliver; data-dependent facts are only evaluated where the decision needs them."""
    import itertools
    from rtrec_amd.score_plan import Path, TOPK_DENSE as D, TOPK_SPARSE as S
    n = 0
    for (mode, hip, f64, top_k, n_rows, full, nonempty, dfast, dfill, lazy, fr_on, sg_on, sg_sup, fill_ok, w_ok, x_ok, rmode, has_fr, has_sg, tiled) in \
            itertools.product((0, 1, 2), (True, False), (False, True), (10, 15, 16, 63, 64), (100, 100_000), (True, False), (True, False),
                              (True, False), (True, False), (True, False), (True, False), (True, False), (True,), (True, False),
                              (True, False), (True, False), (0, 1, 2), (True, False), (True, False), (True, False)):
        if w_ok and rmode == 0:
            continue                    # (a positive float32-valued W always has a refine mode)
        calls = []
        fact = lambda name, v: (lambda: (calls.append(name), v)[1])
        f = _facts(mode=mode, hip=hip, acc_f64=f64, top_k=top_k, n_rows=n_rows, full_range=full, nonempty_shard=nonempty,
                   dense_fast_on=dfast, dense_fill_on=dfill, lazy_tiled=lazy, feature_rows_on=fr_on, seg_layout_on=sg_on, seg_supported=sg_sup,
                   dense_fill_ok=fact("fill", fill_ok), f64_w_ok=fact("w", w_ok), f64_x_ok=fact("x", x_ok), f64_refine_mode=fact("m", rmode))
        from rtrec_amd import score_plan as sp_
        req = sp_.plan_fast_layout(f)
        got_fr, got_sg = (has_fr, has_sg) if req.want else (False, False)
        plan = sp_.choose_path(f, req, got_fr, got_sg, tiled)
        n += 1
        ctx = (mode, hip, f64, top_k, n_rows, full, nonempty, dfast, dfill, lazy, fr_on, sg_on, fill_ok, w_ok, x_ok, rmode, has_fr, has_sg, tiled, plan)
        assert not (plan.use_fr and plan.use_sg), ctx
        assert plan.kernel == ("feature_rows" if plan.use_fr else "segments" if plan.use_sg else "tiled"), ctx
        assert (plan.path is Path.TILED) == (plan.kernel == "tiled"), ctx
        if plan.use_fr:
            assert hip and fr_on and got_fr and plan.k_fast <= 15, ctx
        if plan.use_sg:
            assert hip and sg_on and got_sg and plan.k_fast <= 63, ctx
        if not hip or not nonempty or mode == 2:
            assert plan.path is Path.TILED and not req.want, ctx
        if plan.path is Path.FAST_F64:
            assert f64 and lazy and plan.k_fast == top_k + 1 and (mode == S or (mode == D and w_ok and x_ok)), ctx
            assert plan.f64_signed == (not (w_ok and x_ok)), ctx
        else:
            assert plan.k_fast == top_k and not plan.f64_signed, ctx
        if f64 and plan.path is Path.FAST:
            assert False, ctx           # a float64 W never takes the float32 fast pass without the refine step
        if plan.fill:
            assert mode == D and plan.path is Path.FAST and plan.lazy and fill_ok and not f64 and top_k <= 63, ctx
        if mode == D and plan.path is not Path.TILED:
            assert dfast and lazy and (full or req.dense_fill), ctx
        if plan.lazy and plan.path is Path.FAST:
            assert lazy and (mode == D or not tiled), ctx
        if not f64:
            assert "w" not in calls and "x" not in calls and "m" not in calls, ctx      # float32 W: no float64 fact is ever computed
        if mode != D:
            assert "fill" not in calls, ctx
    assert n > 100_000
