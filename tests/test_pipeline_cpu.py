"""End-to-end host pipeline on CPU: SLIM / SLIMElastic / SlimEngine driven through the oracle
stand-in backend (tests/cpu_backend.py), checked against golden vectors of the real reference.

This exercises everything around the kernels -- ingest, partial CSC export,
coefficient write-back with stale entries (fact 6), dtype rules (fact 5), id mapping, cold-start
fallback, pickling -- and, under gloo with world_size 2, the column-sharded multi-process path.
The HIP kernels themselves are covered by the -m gpu tests.
"""
import io
import json
import os
import pickle
import time
import socket

import numpy as np
import pytest
import scipy.sparse as sp

from tests.cpu_backend import OracleBackend
from rtrec_amd.engine import SlimEngine, shard_bounds
from rtrec_amd.models.internal.slim_elastic import SLIMElastic
from rtrec_amd.models.slim import SLIM
from rtrec_amd.recommender import Recommender

G = os.path.join(os.path.dirname(__file__), "golden")


def load_csc(z, prefix):
    return sp.csc_matrix((z[f"{prefix}_data"], z[f"{prefix}_indices"], z[f"{prefix}_indptr"]),
                         shape=tuple(z[f"{prefix}_shape"]))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same_matrix(A, B):
    A, B = A.tocsc(), B.tocsc()
    A.sort_indices(); B.sort_indices()
    return (A.shape == B.shape and np.array_equal(A.indptr, B.indptr) and np.array_equal(A.indices, B.indices)
            and np.array_equal(bits(A.data), bits(B.data)))


def cpu_slim(**kw):
    m = SLIM(**kw)
    m.model._engine = SlimEngine(backend=OracleBackend())
    return m


@pytest.mark.parametrize("name,kw", [("k5", {"nn_feature_selection": 5}), ("all", {}),
                                     ("k5_decay", {"nn_feature_selection": 5, "decay_in_days": 30})])
def test_incremental_fit_sequence_matches_reference(name, kw):
    z = np.load(os.path.join(G, "partial_fit.npz"))
    u, i, v, ts = z["u"], z["i"], z["v"], z["ts"]
    m = cpu_slim(min_value=0, max_value=15, **kw)
    for label, key, upsert in (("A", "A", False), ("B", "B", False), ("C_add", "C", False), ("C_upsert", "C", True)):
        a, b = z[key]
        batch = [(int(x), int(y), float(t), float(r)) for x, y, t, r in zip(u[a:b], i[a:b], ts[a:b], v[a:b])]
        m.fit(batch, update_interaction=upsert, progress_bar=False)
        W_ref = load_csc(z, f"W_{name}_{label}")
        # time decay included: the store evaluates it with libm's pow like the reference (rtrec_store_decay)
        assert same_matrix(m.model.item_similarity, W_ref), f"{name} after {label}"
        assert m.model.item_similarity.dtype == np.float32
    users = z[f"rec_users_{name}"].tolist()
    recs = m.recommend_batch(users, top_k=5)
    ref = z[f"rec_{name}"]
    for r, row in enumerate(recs):
        assert row == [x for x in ref[r].tolist() if x >= 0]


def test_w_stays_on_the_device_between_fit_and_recommend():
    """A streaming fit -> recommend sequence never builds the host matrix: the fit output is merged into the resident W
    and the score layouts are built from it (engine.merge_fit / build_*_device); `item_similarity` is materialised only
    when somebody reads it, and then equals the reference's matrix."""
    z = np.load(os.path.join(G, "partial_fit.npz"))
    u, i, v, ts = z["u"], z["i"], z["v"], z["ts"]
    m = cpu_slim(min_value=0, max_value=15, nn_feature_selection=5)
    host = cpu_slim(min_value=0, max_value=15, nn_feature_selection=5)      # re-uploads a host matrix before every recommend
    for key in ("A", "B", "C"):
        a, b = z[key]
        batch = [(int(x), int(y), float(t), float(r)) for x, y, t, r in zip(u[a:b], i[a:b], ts[a:b], v[a:b])]
        m.fit(batch, progress_bar=False)
        host.fit(batch, progress_bar=False)
        host.model.item_similarity = host.model.item_similarity.copy()
        users = z["rec_users_k5"].tolist()
        assert m.recommend_batch(users, top_k=5) == host.recommend_batch(users, top_k=5)
        assert m.model._item_similarity is None and m.model.is_fitted and m.model._w_dev is m.model.engine.weights
    assert same_matrix(m.model.item_similarity, load_csc(z, "W_k5_C_add"))       # first read: one download
    assert m.model.item_similarity is m.model._item_similarity
    m2 = pickle.loads(pickle.dumps(m.model))
    assert same_matrix(m2.item_similarity, m.model.item_similarity) and m2._w_dev is None


def test_slimelastic_facade_dtypes_and_errors():
    z = np.load(os.path.join(G, "models.npz"))
    X = load_csc(z, "X")
    eng = SlimEngine(backend=OracleBackend())
    m = SLIMElastic({"nn_feature_selection": 8}, engine=eng).fit(X.copy())
    assert m.item_similarity.dtype == np.float64 and same_matrix(m.item_similarity, load_csc(z, "W_serial_k8"))
    m2 = SLIMElastic({"nn_feature_selection": 8}, engine=eng).fit(X.copy(), parallel=True)
    assert m2.item_similarity.dtype == np.float32 and same_matrix(m2.item_similarity, load_csc(z, "W_parallel_k8"))
    m3 = SLIMElastic({}, engine=eng).partial_fit_items(X.tocsr(), list(range(60)))      # CSR input is accepted
    assert same_matrix(m3.item_similarity, load_csc(z, "W_partial_all"))
    with pytest.raises(ValueError, match="scipy.sparse.csr_matrix or scipy.sparse.csc_matrix"):
        SLIMElastic({}, engine=eng).fit(X.toarray())
    with pytest.raises(ValueError, match="CSC format"):
        SLIMElastic({}, engine=eng).fit_in_parallel(X.tocsr())
    with pytest.raises(ValueError, match="Invalid Optimizer name"):
        SLIMElastic({"optim": "cg"}, engine=eng).fit(X.copy())
    with pytest.raises(RuntimeError, match="Model must be fitted"):
        SLIMElastic({}, engine=eng).recommend(0, X.tocsr())
    with pytest.raises(RuntimeError, match="Model must be fitted"):
        SLIMElastic({}, engine=eng).similar_items(0)
    # scoring through the facade == golden
    zs = np.load(os.path.join(G, "scoring.npz"))
    X2 = load_csc(z, "X2").tocsr()
    ms = SLIMElastic({"nn_feature_selection": 50}, engine=eng)
    ms.item_similarity = load_csc(z, "W2_k50")
    users = zs["users"].tolist()
    out = ms.recommend_batch(users, X2, top_k=10, filter_interacted=True, dense_output=False, ret_scores=True)
    for r, (ids, sc) in enumerate(out):
        ref = [x for x in zs["ids_f32_sparse_filter"][r].tolist() if x >= 0]
        assert ids == ref and np.array_equal(bits(sc), bits(zs["scores_f32_sparse_filter"][r, :len(ref)]))
    assert ms.recommend(users[3], X2, top_k=10, dense_output=False) == out[3][0]
    sim = ms.similar_items(7, top_k=6)
    assert [a for a, _ in sim] == [x for x in zs["similar_ids"][7].tolist() if x >= 0]


def test_api_scenarios_and_pickle_roundtrip():
    api = json.load(open(os.path.join(G, "api.json")))
    m = cpu_slim()
    m.fit([tuple(x) for x in api["int_ids"]["interactions"]], progress_bar=False)
    assert m.recommend_batch([1, 2, 3, 4], top_k=3) == api["int_ids"]["top3"]
    assert m.recommend_batch([1, 2, 3, 4], top_k=3, filter_interacted=False) == api["int_ids"]["nofilter"]
    assert m.recommend(99, top_k=3) == api["int_ids"]["cold_99"]              # unseen int id -> hot items
    assert m.similar_items(10, top_k=3) == api["int_ids"]["similar_10"]
    assert m.recommend_batch([], top_k=2) == []
    # pickle round trip gives identical recommendations (reference: test_serialization.py:152-180)
    buf = io.BytesIO()
    assert m.save(buf) > 0
    m2 = SLIM.loads(buf.getvalue())
    m2.model._engine = SlimEngine(backend=OracleBackend())
    assert m2.recommend_batch([1, 2, 3, 4], top_k=3) == api["int_ids"]["top3"]
    assert same_matrix(m2.model.item_similarity, m.model.item_similarity)
    # string ids (dense top-k path): untied scenarios of the reference's own tests
    s = cpu_slim()
    s.fit([tuple(x) for x in api["similar_items"]["interactions"]], progress_bar=False)
    assert s.similar_items("item_1", top_k=5) == api["similar_items"]["similar_item_1"] == ["item_4", "item_3"]
    got = s.similar_items("item_1", top_k=5, ret_scores=True)
    assert [a for a, _ in got] == ["item_4", "item_3"] and got[0][1] > got[1][1]
    assert s.get_users_by_items(["item_2"]) == ["user_2"] and s.get_users_by_items(["nope"]) == []
    assert SLIM().recommend("user_1", top_k=5) == []                       # empty model -> cold path
    assert s.register_user_feature("user_9", ["a", "b"]) == s.user_ids.identify("user_9")
    assert s.feature_store.build_user_features_matrix(user_ids=[s.user_ids.identify("user_9")]).nnz == 2


def test_bad_interactions_are_skipped_with_warning(caplog):
    m = cpu_slim()
    m.add_interactions([(1, 10, 1.7e9, 5.0), ("oops", 11, 1.7e9, 1.0), (2, 12, 1.7e9, 2.0)])
    assert m.interactions.get_user_item_rating(1, 10) == 5.0 and m.interactions.get_user_item_rating(2, 12) == 2.0
    assert "Error processing interaction" in caplog.text


def test_recommender_facade(capsys):
    import pandas as pd
    z = np.load(os.path.join(G, "partial_fit.npz"))
    a, b = z["A"]
    df = pd.DataFrame({"user": z["u"][a:b], "item": z["i"][a:b], "tstamp": z["ts"][a:b], "rating": z["v"][a:b]})
    rec = Recommender(cpu_slim(min_value=0, max_value=15, nn_feature_selection=5))
    rec.fit(df, batch_size=128)
    assert "Throughput:" in capsys.readouterr().out
    assert same_matrix(rec.get_model().model.item_similarity, load_csc(z, "W_k5_A"))
    rec2 = Recommender(cpu_slim(min_value=0, max_value=15, nn_feature_selection=5))
    rec2.bulk_fit(df, parallel=True)          # every column, float32
    assert rec2.get_model().model.item_similarity.dtype == np.float32
    scores = rec2.evaluate(df.iloc[:200], recommend_size=5, filter_interacted=False)
    assert set(scores) == {"precision", "recall", "f1", "ndcg", "hit_rate", "mrr", "map", "tp", "auc"}
    assert len(rec2.recommend_batch([0, 1], top_k=3)) == 2 and len(rec2.similar_items([1, 2], top_k=3)) == 2


# ------------------------------------------------------------------ world_size 2 over gloo
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, score_shard="columns"):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torch
        torch.set_num_threads(1)           # 8 ranks share this container's 8 cores
        z = np.load(os.path.join(G, "models.npz"))
        X = load_csc(z, "X2")
        shard_w = score_shard == "columns+w"        # W stays column-sharded: no all-gather of the coefficients
        score_shard = "columns" if shard_w else score_shard
        eng = SlimEngine(backend=OracleBackend(), rank=rank, world_size=world, score_shard=score_shard, shard_w=shard_w)
        eng.gather_chunk_rows = 7          # several chunks -> several asynchronous exchanges in flight (column path) ...
        eng.row_chunk_rows = 3             # ... and several overlapped all-gathers of a rank's row slice (row path), also at world 8
        m = SLIMElastic({"nn_feature_selection": 50}, engine=eng)
        m.partial_fit_items(X.copy(), list(range(400)))            # each rank fits its own column shard
        if shard_w:                        # this rank holds its own column block and nothing else ...
            lo, hi = shard_bounds(400, world, rank)
            own = m._w_dev
            assert own.shard == (rank, world) and (own.nnz == 0 or (int(own.cols.min()) >= lo and int(own.cols.max()) < hi))
            assert own.nnz < load_csc(z, "W2_k50").nnz
            sims = m.similar_items_batch([0, 7, 399, 123], top_k=6)      # ... and item-to-item queries go to the owner
        ok_w = same_matrix(m.item_similarity, load_csc(z, "W2_k50"))      # (sharded W: gathered on demand, a collective)
        if shard_w:
            solo = SLIMElastic({"nn_feature_selection": 50}, engine=SlimEngine(backend=OracleBackend()))
            solo.item_similarity = load_csc(z, "W2_k50")
            ok_w = ok_w and sims == solo.similar_items_batch([0, 7, 399, 123], top_k=6)
            # a second, partial fit merges into the sharded W (old entries of the refitted columns survive / are replaced)
            m.partial_fit_items(X.copy(), [3, 50, 51, 250, 399])
            m._item_similarity = None
            ok_w = ok_w and same_matrix(m.item_similarity, load_csc(z, "W2_k50"))
        zs = np.load(os.path.join(G, "scoring.npz"))
        users = zs["users"].tolist()
        out = m.recommend_batch(users, X.tocsr(), top_k=10, filter_interacted=True, dense_output=False)
        ref = [[x for x in row.tolist() if x >= 0] for row in zs["ids_f32_sparse_filter"]]
        ok = out == ref
        # dense (string-id) mode and a float64 W (float64 scores travel in front of the record)
        out = m.recommend_batch(users, X.tocsr(), top_k=10, filter_interacted=False, dense_output=True)
        ok = ok and out == [[x for x in row.tolist() if x >= 0] for row in zs["ids_f32_dense_nofilter"]]
        m.item_similarity = sp.csc_matrix(load_csc(z, "W2_k50"), dtype=np.float64)     # (an assigned W is replicated again)
        out = m.recommend_batch(users, X.tocsr(), top_k=10, filter_interacted=True, dense_output=False, ret_scores=True)
        ok = ok and [o[0] for o in out] == [[x for x in row.tolist() if x >= 0] for row in zs["ids_f64_sparse_filter"]]
        # resident X: two calls with DIFFERENT row sets of equal length must not share a cached slice / work order (ADVICE round
        # 2: a cache keyed on the row tensor's address returned the first call's users for the second)
        eng.set_weights(load_csc(z, "W2_k50"))
        eng.set_interactions(None, X.tocsr(), need_csc=False)
        solo = SlimEngine(backend=OracleBackend())
        solo.set_interactions(None, X.tocsr(), need_csc=False)
        solo.set_weights(load_csc(z, "W2_k50"))
        for rows in ([5, 17, 40, 41, 300], [9, 3, 77, 78, 1100], [5, 17, 40, 41, 300], [1, 2, 3, 4, 6]):
            got, ref = eng.recommend_rows(rows, top_k=5), solo.recommend_rows(rows, top_k=5)
            ok = ok and all(np.array_equal(a, b) for a, b in zip(got, ref))
        if score_shard == "rows":
            # ids and counts only (what recommend_batch hands out): two collectives per chunk, no score tensor
            from rtrec_amd import _native as nat
            rows_ = np.arange(0, 1200, 7, dtype=np.int32)
            got = eng.score_topk_device(rows_, len(rows_), 5, True, nat.TOPK_SPARSE, with_scores=False)
            ref = solo.recommend_rows(rows_, top_k=5)
            ok = ok and got[1] is None and np.array_equal(np.asarray(got[0]), ref[0]) and np.array_equal(np.asarray(got[2]), ref[2])
        if world <= 3 and score_shard == "columns":
            # exact score ties ACROSS column shards (the second half of W's columns are copies of the first, integer ratings):
            # neither shard sees a tie, the reference's tie key has to travel with every entry (round 4)
            rng = np.random.default_rng(3)
            half = 200
            rows_w = np.sort(rng.choice(400, 60, replace=False))
            A = sp.csc_matrix((rng.integers(1, 4, 1500).astype(np.float32), (rng.choice(rows_w, 1500), rng.integers(0, half, 1500))),
                              shape=(400, half), dtype=np.float32)
            A.sum_duplicates()
            Wt = sp.hstack([A, A], format="csc").astype(np.float32)
            Wt.sort_indices()
            ur = np.repeat(np.arange(150), rng.integers(1, 30, 150))
            Xt = sp.csr_matrix((rng.integers(1, 6, len(ur)).astype(np.float32), (ur, rng.choice(rows_w, len(ur)))), shape=(150, 400),
                               dtype=np.float32)
            Xt.sum_duplicates(); Xt.sort_indices()
            for e in (eng, solo):
                e.set_interactions(None, Xt, need_csc=False)
                e.set_weights(Wt)
            got, ref = eng.recommend_rows(np.arange(150), top_k=7), solo.recommend_rows(np.arange(150), top_k=7)
            ok = ok and all(np.array_equal(a, b) for a, b in zip(got, ref))
        if world == 2 and score_shard == "columns" and not shard_w:
            # optim="sgd" shards like the coordinate-descent fit: every rank fits its own targets, the triples are all-gathered
            sgd = SLIMElastic({"optim": "sgd", "nn_feature_selection": 6, "max_iter": 12},
                              engine=SlimEngine(backend=OracleBackend(), rank=rank, world_size=world))
            sgd.partial_fit_items(X.copy(), list(range(0, 400, 5)))
            ref = SLIMElastic({"optim": "sgd", "nn_feature_selection": 6, "max_iter": 12}, engine=SlimEngine(backend=OracleBackend()))
            ref.partial_fit_items(X.copy(), list(range(0, 400, 5)))
            ok_w = ok_w and same_matrix(sgd.item_similarity, ref.item_similarity)
        q.put((rank, ok_w, ok, eng._layout(True)["n_cols"]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,score_shard", [(2, "columns"), (3, "columns"), (2, "rows"), (3, "rows"), (8, "columns"), (8, "rows"),
                                               (2, "columns+w"), (3, "columns+w"), (8, "columns+w")])
def test_two_rank_column_sharding_gloo(world, score_shard):
    """Column-sharded fit + scoring over gloo: per-shard lists travel by all-to-all (every rank merges
    its slice of the users; 7-row chunks do not divide by 2 or 3, so padded slices are exercised) and the
    final lists by all-gather; "rows" mode replicates W and shards the users instead.  Either way the
    results must equal the single-process reference goldens."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, score_shard)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=420) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    assert all(r[1] for r in res), "merged W differs from the single-process reference W"
    assert all(r[2] for r in res), "sharded top-k + merge differs from the reference top-k"
    assert sum(r[3] for r in res) > 0


def test_predict_score_vectors_match_reference():
    z = np.load(os.path.join(G, "models.npz"))
    zs = np.load(os.path.join(G, "scoring.npz"))
    X2 = load_csc(z, "X2").tocsr()
    m = SLIMElastic({"nn_feature_selection": 50}, engine=SlimEngine(backend=OracleBackend()))
    with pytest.raises(RuntimeError, match="Model must be fitted before calling predict"):
        m.predict(0, X2)
    m.item_similarity = load_csc(z, "W2_k50")
    users, cands = zs["predict_users"].tolist(), zs["cands"].tolist()
    for r, u in enumerate(users):
        d = m.predict(u, X2, dense_output=True)
        assert d.shape == (1, 400) and d.dtype == np.float32
        assert np.array_equal(bits(d.ravel()), bits(zs["predict_dense"][r]))
        s_ = m.predict(u, X2, dense_output=False)
        assert sp.issparse(s_) and np.array_equal(bits(s_.toarray().ravel()), bits(zs["predict_sparse_as_dense"][r]))
        assert np.array_equal(bits(m.predict_selected(u, cands, X2).ravel()), bits(zs["predict_selected"][r]))
    assert np.array_equal(bits(m.predict_all(X2[:40])), bits(zs["predict_all_head"]))
    m.item_similarity = sp.csc_matrix(m.item_similarity, dtype=np.float64)
    d64 = m.predict(users[1], X2)
    assert d64.dtype == np.float64 and np.array_equal(d64.ravel(), zs["predict_dense_f64"][1])


def test_loads_model_files_written_by_the_reference():
    """tests/golden/ref_slim_*.pkl were saved by rtrec.models.SLIM.save (tools/gen_golden.py);
    rtrec is not importable here, the compat unpickler maps its classes."""
    import sys
    assert "rtrec" not in sys.modules
    exp = json.load(open(os.path.join(G, "ref_pickles.json")))
    m = SLIM.loads(open(os.path.join(G, "ref_slim_int.pkl"), "rb").read())
    m.model._engine = SlimEngine(backend=OracleBackend())
    assert type(m.interactions).__module__.startswith("rtrec_amd") and type(m.model).__module__.startswith("rtrec_amd")
    assert np.array_equal(m.interactions.to_csr().toarray(), np.asarray(exp["int"]["csr"], dtype=np.float32))
    assert m.interactions.get_hot_items(5, filter_interacted=False) == exp["int"]["hot"]
    assert m.recommend_batch(exp["int"]["users"], top_k=5) == exp["int"]["recs"]
    assert [list(x) for x in m.similar_items(3, top_k=4, ret_scores=False) and [[a] for a in m.similar_items(3, top_k=4)]] == [[a] for a in exp["int"]["similar_3"]]
    # the loaded model keeps learning
    m.fit([(0, 1, 1.8e9, 2.0), (1, 2, 1.8e9, 1.0)], progress_bar=False)
    buf = io.BytesIO(); m.save(buf)
    m2 = SLIM.loads(buf.getvalue())                    # and round-trips in rtrec_amd's own format
    assert same_matrix(m2.model.item_similarity, m.model.item_similarity)
    s = SLIM.loads(open(os.path.join(G, "ref_slim_str.pkl"), "rb").read())
    s.model._engine = SlimEngine(backend=OracleBackend())
    assert s.similar_items("item_1", top_k=5) == exp["str"]["similar_item_1"]
    assert s.recommend("user_2", top_k=5) == exp["str"]["rec_user_2"]
    assert s.feature_store.build_item_features_matrix(item_ids=[0]).nnz == exp["str"]["item_feature_nnz"]


@pytest.mark.parametrize("session", ["str", "int"])
def test_serving_shell_replays_the_reference_transcript(session):
    """tests/golden/serving.json is the request/response transcript of the reference's FastAPI app
    (rtrec/serving/app.py) for a fixed request sequence; rtrec_amd.serving.app must answer alike."""
    from fastapi.testclient import TestClient
    from rtrec_amd.serving.app import create_app
    steps = json.load(open(os.path.join(G, "serving.json")))[session]
    client = TestClient(create_app(lambda: cpu_slim(min_value=-5, max_value=10, decay_in_days=365)))
    for st in steps:
        r = client.get(st["path"]) if st["method"] == "GET" else client.post(st["path"], json=st["json"], headers=st["headers"])
        assert r.status_code == st["status"], st
        assert r.json() == st["response"], st
    # the batched route (an addition) answers like one /recommend per KNOWN user (for an unknown user
    # the reference's single-user cold path returns internal ids, base.py:162-167, its batch path raw ids)
    users = [s["json"]["user"] for s in steps if s["path"] == "/recommend" and s["status"] == 200][-4:-1]
    r = client.post("/recommend_batch", json={"users": users, "top_k": 3, "filter_interacted": False},
                    headers={"X-Token": "fake_secret_token"})
    assert r.status_code == 200
    assert r.json()["recommendations"] == [s["response"]["recommendations"] for s in steps
                                           if s["path"] == "/recommend" and s["status"] == 200][-4:-1]


@pytest.mark.parametrize("ids", ["int", "str"])
def test_concurrent_recommend_requests_are_coalesced_into_shared_launches(ids):
    """serving.app.RecommendCoalescer: concurrent /recommend requests share recommend_batch launches, every caller
    gets exactly the answer of its own model.recommend call (known, unknown and failing users alike), and a request
    arriving alone is answered without company."""
    import threading
    from rtrec_amd.serving.app import ModelGate
    z = np.load(os.path.join(G, "partial_fit.npz"))
    a, b = z["A"]
    conv = (lambda x: f"id{int(x)}") if ids == "str" else int
    m = cpu_slim(min_value=0, max_value=15, nn_feature_selection=5)
    m.fit([(conv(x), conv(y), float(t), float(r)) for x, y, t, r in zip(z["u"][a:b], z["i"][a:b], z["ts"][a:b], z["v"][a:b])],
          progress_bar=False)
    users = [conv(x) for x in sorted(set(z["u"][a:b].tolist()))[:40]] + [conv(10 ** 6), conv(10 ** 6 + 1)]     # two unknown
    want = {(u, k, f): m.recommend(user=u, top_k=k, filter_interacted=f) for u in users for k, f in ((5, True), (3, False))}
    calls = {"batch": 0, "single": 0}
    rb, r1 = m.recommend_batch, m.recommend
    m.recommend_batch = lambda *a_, **k_: (calls.__setitem__("batch", calls["batch"] + 1), rb(*a_, **k_))[1]
    m.recommend = lambda *a_, **k_: (calls.__setitem__("single", calls["single"] + 1), r1(*a_, **k_))[1]
    gate = ModelGate(m, coalesce_ms=50)
    got, errs = {}, []

    def client(u, k, f):
        try:
            got[(u, k, f)] = gate.recommend(u, k, f)
        except Exception as exc:        # pragma: no cover
            errs.append(exc)

    with gate._lock:                    # the model is busy (a /fit): the requests queue up behind it
        threads = [threading.Thread(target=client, args=key) for key in want]
        for t in threads:
            t.start()
        time.sleep(0.2)
    for t in threads:
        t.join(timeout=60)
    assert not errs and got == want
    assert gate.coalescer.requests == len(want) and not gate.coalescer._leading and not gate.coalescer._queue
    assert calls["batch"] <= 4 and calls["single"] <= 4 + 1, calls      # two option groups (+ a first round that may be small)
    # alone: answered by a plain recommend call after the bounded wait
    calls.update(batch=0, single=0)
    assert gate.recommend(users[0], 5, True) == want[(users[0], 5, True)] and calls == {"batch": 0, "single": 1}
    # a failing shared launch falls back to one call per request; a failing request is a 500 for that caller only
    m.recommend_batch = lambda *a_, **k_: (_ for _ in ()).throw(RuntimeError("boom"))
    m.recommend = lambda user, **k_: (_ for _ in ()).throw(RuntimeError("bad user")) if user == users[1] else r1(user=user, **k_)
    got.clear()
    with gate._lock:
        threads = [threading.Thread(target=client, args=(u, 5, True)) for u in users[:6]]
        for t in threads:
            t.start()
        time.sleep(0.2)
    for t in threads:
        t.join(timeout=60)
    assert len(errs) == 1 and getattr(errs[0], "status_code", None) == 500
    assert got == {(u, 5, True): want[(u, 5, True)] for u in users[:6] if u != users[1]}


@pytest.mark.parametrize("ids", ["int", "str"])
def test_batched_similar_items_equal_one_call_per_query(ids):
    """Recommender.similar_items(list) answers all queries with one kernel launch
    (BaseModel.similar_items_batch); it must return exactly what the reference's loop of
    model.similar_items(q) returns -- unknown items included (they are registered, base.py:329)."""
    rng = np.random.default_rng(2)
    n = 4000
    conv = (lambda x: int(x)) if ids == "int" else (lambda x: f"i{int(x)}")
    rows = [(int(u), conv(i), 1.7e9 + k, float(r)) for k, (u, i, r) in
            enumerate(zip(rng.integers(0, 300, n), rng.zipf(1.4, n) % 80, rng.integers(1, 6, n)))]
    from rtrec_amd import Recommender

    def build():
        m = SLIM(nn_feature_selection=8)
        m.model._engine = SlimEngine(backend=OracleBackend())
        m.fit(rows, progress_bar=False)
        return m
    queries = [conv(q) for q in (0, 3, 7, 79, 5, 3)] + [conv(10 ** 6)]      # a repeated and an unseen item
    a, b = build(), build()
    for ret_scores in (False, True):
        one_by_one = [a.similar_items(q, top_k=6, ret_scores=ret_scores) for q in queries]
        batched = Recommender(b).similar_items(queries, top_k=6, ret_scores=ret_scores)
        assert batched == one_by_one
    assert any(len(x) > 0 for x in one_by_one)
    assert a.item_ids.get_id(conv(10 ** 6)) == b.item_ids.get_id(conv(10 ** 6))


def test_odd_user_ids_and_large_top_k_on_the_host_paths():
    """Host logic of two round-2 additions, on the stand-in backend: (1) internal user ids outside [0, n_users)
    are resolved before anything is launched (IndexError / wrapped row), (2) top_k beyond the fused kernel's
    limit is served by score rows + a host selection that equals the oracle's order."""
    from oracle import slim_oracle as so
    rng = np.random.default_rng(9)
    n = 6000
    u, i = rng.integers(0, 120, n), (rng.zipf(1.25, n) - 1) % 1300
    r = rng.integers(1, 6, n).astype(float) + rng.random(n)
    m = cpu_slim(nn_feature_selection=10)
    m.add_interactions(list(zip(u.tolist(), i.tolist(), (1.7e9 + np.arange(n)).tolist(), r.tolist())))
    m.bulk_fit(progress_bar=False)
    n_users = m.interactions.shape[0]
    ok = m.recommend_batch([0, 1, 2], top_k=5)
    assert m._recommend_hot_batch([0, -1, 2], top_k=5) == [ok[0], [], ok[2]]
    last = m._recommend_hot_batch([n_users - 1], top_k=5)[0]
    assert m._recommend_hot_batch([-1, n_users - 1], top_k=5) == [last, last]
    with pytest.raises(IndexError):
        m._recommend_hot_batch([n_users], top_k=5)
    with pytest.raises(IndexError):
        m.model.engine.recommend_rows([-1], top_k=5)
    W, X = m.model.item_similarity, m.interactions.to_csr()
    users = list(range(0, 120, 5))
    for dense in (False, True):
        big = m.model.recommend_batch(users, X, top_k=1100, dense_output=dense, ret_scores=True)
        o_ids, o_sc, o_cnt = so.recommend_batch(X[users], W.tocsr(), top_k=1100, dense=dense, use_f64=(W.dtype == np.float64))
        for (ids, sc), oi, os_, oc in zip(big, o_ids, o_sc, o_cnt):
            assert ids == oi[:oc].tolist() and np.array_equal(bits(sc), bits(os_[:oc]))


def _evaluate_against_golden(make_model):
    """Recommender(SLIM).fit(train) -> evaluate(test) / recommend_batch vs the reference's own run
    (tests/golden/evaluate.json, tools/gen_golden.py: rtrec/recommender.py:39-82,163-200)."""
    import json
    import pandas as pd
    g = json.load(open(os.path.join(G, "evaluate.json")))["e2e"]
    cols = ("user", "item", "tstamp", "rating")
    train = pd.DataFrame(dict(zip(cols, g["train"])))
    test = pd.DataFrame(dict(zip(cols, g["test"])))
    rec = Recommender(make_model(**g["model_kwargs"]))
    rec.fit(train, batch_size=1000, parallel=False)
    assert rec.recommend_batch(g["users"], top_k=10) == g["recommend_top10"]
    for key, ref in g["evaluate"].items():
        size, fi = key.split("_")
        got = rec.evaluate(test, recommend_size=int(size), filter_interacted=bool(int(fi)))
        assert set(got) == set(ref) and got["tp"] == ref["tp"]
        for k in ref:
            assert got[k] == pytest.approx(ref[k], rel=1e-12), (key, k)


def test_evaluate_matches_the_reference_end_to_end():
    _evaluate_against_golden(cpu_slim)


def test_candidate_lists_follow_the_reference_ranking_on_the_oracle_backend():
    """Candidate mode of the CPU backend that tools/fuzz_api.py compares the GPU with (slim_elastic.py:723-735: dense scores
    of the candidate columns, argsort()[-k:][::-1]; ties by the stable-argsort rule of DESIGN D1, zeros included,
    filter_interacted ignored) -- against a direct numpy statement of those lines."""
    rng = np.random.default_rng(0)
    m = SLIM(min_value=0, max_value=15, nn_feature_selection=5)
    m.model._engine = SlimEngine(backend=OracleBackend())
    n, U, I = 300, 60, 25
    m.fit([(int(a), int(b), 1.7e9 + t, float(r)) for t, (a, b, r) in
           enumerate(zip(rng.zipf(1.6, n) % U, rng.zipf(1.4, n) % I, rng.integers(1, 6, n)))], progress_bar=False)
    cands = [13, 11, 0, 19, 3, 22, 1, 14, 5]
    X, W = m.interactions.to_csr(), m.model.item_similarity.tocsc()
    users = [u for u in range(U) if m.user_ids.get_id(u) is not None]
    for filt in (True, False):
        got = m.recommend_batch(users, candidate_items=cands, top_k=5, filter_interacted=filt)
        for u, g in zip(users, got):
            s = (X[m.user_ids.get_id(u)] @ W[:, cands]).toarray().ravel()
            assert g == [cands[i] for i in np.argsort(s, kind="stable")[-5:][::-1]]


def test_hybrid_slimfm_call_sequence_on_the_facade():
    """The calls HybridSlimFM makes on its SLIMElastic (hybrid.py:122 ... :477), in its order, with the real SLIMElastic's answers
    (hybrid itself needs lightfm / implicit, its SLIM half is what rtrec_amd replaces)."""
    from tests.hybrid_replay import replay
    replay(lambda cfg: SLIMElastic(cfg, engine=SlimEngine(backend=OracleBackend())))



@pytest.mark.parametrize("item_kind", ["int", "str"])
def test_vectorised_recommend_batch_equals_the_per_user_loop(item_kind, monkeypatch):
    """BaseModel.recommend_batch with integer users takes one array compare instead of get_id per user
    (rtrec/models/base.py:188-269 is the per-user loop): same lists for hot / cold mixes, candidates, numpy / range / list
    inputs, ids beyond and below the matrix; `as_arrays=True` carries the same answers as (ids, counts)."""
    rng = np.random.default_rng(4)
    n = 5000
    u, i = rng.integers(0, 150, n), (rng.zipf(1.3, n) - 1) % 400
    r = rng.integers(1, 6, n).astype(float) + rng.random(n)
    conv = (lambda x: int(x)) if item_kind == "int" else (lambda x: f"i{int(x)}")
    m = cpu_slim(nn_feature_selection=10)
    m.add_interactions([(int(a), conv(b), 1.7e9 + t, float(c)) for t, (a, b, c) in enumerate(zip(u, i, r))])
    m.bulk_fit(progress_bar=False)
    n_users = m.interactions.shape[0]
    cands = [conv(x) for x in (3, 1, 7, 399, 12, 5000)] + ([10 ** 7] if item_kind == "int" else ["never seen"])
    cases = [list(range(n_users)), [5, 900, 3, 901, 5], [900, 901], np.arange(0, n_users, 3), range(10, 40),
             np.array([2, 10 ** 6], dtype=np.int32), [7], (1, 2, 3)]

    def loop(users, **kw):
        with monkeypatch.context() as mp:
            mp.setattr(type(m), "_int_user_array", lambda self, users: None)
            return m.recommend_batch(list(users) if not isinstance(users, list) else users, **kw)

    for users in cases:
        for kw in (dict(top_k=5), dict(top_k=4, filter_interacted=False), dict(top_k=3, candidate_items=cands),
                   dict(top_k=50)):
            want = loop(users, **kw)
            assert m._int_user_array(users) is not None
            got = m.recommend_batch(users, **kw)
            assert got == want, (users, kw)
            ids, counts = m.recommend_batch(users, as_arrays=True, **kw)
            assert counts.tolist() == [len(x) for x in want]
            assert [ids[b, :c].tolist() for b, c in enumerate(counts.tolist())] == want
            a_ids, a_counts = loop(users, as_arrays=True, **kw)
            assert a_counts.tolist() == counts.tolist()
            assert [a_ids[b, :c].tolist() for b, c in enumerate(a_counts.tolist())] == want
    # batches that must keep the per-user loop: mixed kinds, floats, per-user tags, a negative (wrapping) id stays correct
    assert m._int_user_array([1, "x"]) is None and m._int_user_array([1.0, 2.0]) is None and m._int_user_array([]) is None
    assert m.recommend_batch([0, -1, 2], top_k=5) == loop([0, -1, 2], top_k=5)
    with pytest.raises(IndexError):
        m.recommend_batch([0, -n_users - 1], top_k=5)
    assert m.recommend_batch([], top_k=5) == [] and m.recommend_batch(np.empty(0, np.int64), top_k=5) == []
    e_ids, e_counts = m.recommend_batch(np.empty(0, np.int64), top_k=5, as_arrays=True)        # no (None, None)
    assert e_ids.shape == (0, 5) and e_counts.shape == (0,)
    cold_rows = m.recommend_batch([900, 901, 5], top_k=5)                                      # every cold user a list of its own
    assert cold_rows[0] == cold_rows[1] and cold_rows[0] is not cold_rows[1]
    rec = Recommender(m)
    assert rec.recommend_batch([1, 2], top_k=3) == m.recommend_batch([1, 2], top_k=3)
    assert rec.recommend_batch([1, 2], top_k=3, as_arrays=True)[1].tolist() == [3, 3]
    # string users never take the array route
    s = cpu_slim()
    s.fit([("a", 1, 1.7e9, 3.0), ("b", 2, 1.7e9, 2.0), ("a", 2, 1.7e9, 1.0)], progress_bar=False)
    assert s._int_user_array(["a", "b"]) is None and s._int_user_array([1, 2]) is None


def test_optim_sgd_on_the_facade_matches_the_reference():
    """SLIMElastic({"optim": "sgd", ...}) (slim_elastic.py:209-222) through the facade on the stand-in backend: the serial fit's
    float64 W and the incremental refit equal the reference goldens bit for bit; without nn_feature_selection the fit fails
    like the reference's (AttributeError at slim_elastic.py:273) unless there is nothing to fit; an unknown optimiser raises
    ValueError (slim_elastic.py:224)."""
    from rtrec_amd.synth import interaction_matrix
    g = json.load(open(os.path.join(G, "sgd.json")))
    c = next(x for x in g["cases"] if "partial_items" in x)
    X = interaction_matrix(c["U"], c["I"], c["draws"], seed=c["seed"]).tocsc()
    X.sort_indices()

    def golden(prefix):
        b = np.asarray(c[f"{prefix}_bits"], dtype=np.uint32)
        return sp.csc_matrix((b.view(np.float32), np.asarray(c[f"{prefix}_indices"]), np.asarray(c[f"{prefix}_indptr"])),
                             shape=(c["I"], c["I"]))
    m = SLIMElastic(dict(c["cfg"], optim="sgd"), engine=SlimEngine(backend=OracleBackend()))
    m.fit(X.copy())
    assert m.item_similarity.dtype == np.float64 == np.dtype(c["W_dtype"])
    assert same_matrix(m.item_similarity, golden("W")) and m.n_iter_.tolist() == c["n_iter"]
    m.partial_fit_items(X.copy(), c["partial_items"])
    assert m.item_similarity.dtype == np.dtype(c["W2_dtype"]) and same_matrix(m.item_similarity, golden("W2"))
    bad = SLIMElastic({"optim": "sgd"}, engine=SlimEngine(backend=OracleBackend()))
    with pytest.raises(AttributeError, match="'SGDRegressor' object has no attribute 'sparse_coef_'"):
        bad.fit(X.copy())
    with pytest.raises(AttributeError, match="sparse_coef_"):
        bad.partial_fit_items(X.copy(), [1, 2])
    bad.partial_fit_items(X.copy(), [])                      # nothing to fit: the reference does not fail either
    assert bad.item_similarity.nnz == 0
    with pytest.raises(ValueError, match="Invalid Optimizer name: bogus"):
        SLIMElastic({"optim": "bogus"}, engine=SlimEngine(backend=OracleBackend())).fit(X.copy())
