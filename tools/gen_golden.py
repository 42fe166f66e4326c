#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container: imports rtrec from /root/reference (read-only) plus the
scikit-learn 1.7.2 / scipy 1.15.3 / numpy 2.2.6 it drives, feeds it seeded synthetic inputs and
stores inputs + outputs as small .npz / .json fixtures.  Nothing here (nor the reference) ships
to the GPU box; the tests read only the fixtures.

    python tools/gen_golden.py

Inputs use float ("decayed") ratings so that X^T y has no exact ties: numpy's default argsort is
unstable, so the reference's own choice among tied features is platform-defined.
"""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from ref_import import import_reference  # noqa: E402

import_reference()
from rtrec.models import SLIM as RefSLIM  # noqa: E402
from rtrec.models.internal.slim_elastic import SLIMElastic as RefSLIMElastic  # noqa: E402
from rtrec.utils.interactions import UserItemInteractions as RefStore  # noqa: E402
from sklearn.exceptions import ConvergenceWarning  # noqa: E402
from sklearn.linear_model import ElasticNet  # noqa: E402

from rtrec_amd.synth import interaction_matrix  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
warnings.simplefilter("ignore", ConvergenceWarning)
T0 = 1_700_000_000.0


def csc_parts(M, prefix):
    M = M.tocsc()
    M.sort_indices()
    return {f"{prefix}_data": M.data, f"{prefix}_indices": M.indices.astype(np.int32),
            f"{prefix}_indptr": M.indptr.astype(np.int32), f"{prefix}_shape": np.array(M.shape),
            f"{prefix}_dtype": np.array(str(M.dtype))}


def pad_lists(lists, k, fill):
    out = np.full((len(lists), k), fill, dtype=np.float64 if isinstance(fill, float) else np.int64)
    for r, l in enumerate(lists):
        out[r, :len(l)] = l
    return out


# ------------------------------------------------------------------ G2: RNG + per-column CD known answers
def gen_rng():
    from sklearn.utils import check_random_state
    d = {}
    for rs in (43, 0, 1, 7, 12345):
        seed = int(check_random_state(rs).randint(0, 2147483647))
        st, draws = seed, []
        for _ in range(12):
            st ^= (st << 13) & 0xFFFFFFFF
            st ^= st >> 17
            st ^= (st << 5) & 0xFFFFFFFF
            draws.append(st % 2147483648)
        d[str(rs)] = {"seed": seed, "mod50": [x % 50 for x in draws], "mod3707": [x % 3707 for x in draws]}
    json.dump(d, open(os.path.join(OUT, "rng.json"), "w"), indent=1)


def gen_cd_columns():
    X = interaction_matrix(300, 40, 2600, seed=101).tocsc()
    X.sort_indices()
    out = csc_parts(X, "X")
    for name, kw in {"default": {}, "nonpositive": {"positive": False}, "loose": {"tol": 1e-2, "max_iter": 5},
                     "strong": {"alpha": 0.5, "l1_ratio": 0.5}}.items():
        coefs, iters = [], []
        for j in range(X.shape[1]):
            Xj = X.copy()
            y = Xj[:, j].toarray().ravel()
            Xj.data[Xj.indptr[j]:Xj.indptr[j + 1]] = 0
            m = ElasticNet(alpha=kw.get("alpha", 0.1), l1_ratio=kw.get("l1_ratio", 0.1), fit_intercept=False,
                           precompute=True, max_iter=kw.get("max_iter", 100), copy_X=False, tol=kw.get("tol", 1e-4),
                           positive=kw.get("positive", True), random_state=43, selection="random")
            m.fit(Xj, y)
            coefs.append(m.coef_.astype(np.float32))
            iters.append(m.n_iter_)
        out[f"{name}_coef"] = np.stack(coefs)
        out[f"{name}_n_iter"] = np.array(iters)
    np.savez_compressed(os.path.join(OUT, "cd_columns.npz"), **out)


# ------------------------------------------------------------------ G3: whole-model W through SLIMElastic
def gen_models():
    X = interaction_matrix(200, 60, 1800, seed=202).tocsc()
    X.sort_indices()
    out = csc_parts(X, "X")
    cases = {"serial_all": ({}, "fit"), "serial_k8": ({"nn_feature_selection": 8}, "fit"),
             "partial_all": ({}, "partial"), "partial_k8": ({"nn_feature_selection": 8}, "partial"),
             "partial_k100": ({"nn_feature_selection": 100}, "partial"),   # K > I
             "parallel_k8": ({"nn_feature_selection": 8}, "parallel"),
             "nonpos_k8": ({"nn_feature_selection": 8, "positive_only": False}, "partial")}
    for name, (cfg, how) in cases.items():
        m = RefSLIMElastic(cfg)
        if how == "fit":
            m.fit(X.copy())
        elif how == "parallel":
            m.fit_in_parallel(X.copy())
        else:
            m.partial_fit_items(X.copy(), list(range(X.shape[1])))
        out.update(csc_parts(m.item_similarity, f"W_{name}"))
    # a mid-size model with K=50 (the headline setting)
    X2 = interaction_matrix(1200, 400, 30000, seed=203).tocsc()
    X2.sort_indices()
    out.update(csc_parts(X2, "X2"))
    m = RefSLIMElastic({"nn_feature_selection": 50}).partial_fit_items(X2.copy(), list(range(400)))
    out.update(csc_parts(m.item_similarity, "W2_k50"))
    np.savez_compressed(os.path.join(OUT, "models.npz"), **out)
    return X, X2, m


# ------------------------------------------------------------------ G4: incremental fits through SLIM (facts 6, 7)
def gen_partial():
    rng = np.random.default_rng(5)
    X = interaction_matrix(150, 50, 1500, seed=303).tocoo()
    order = rng.permutation(X.nnz)
    u, i, v = X.row[order], X.col[order], X.data[order]
    ts = T0 + np.arange(X.nnz, dtype=np.float64) * 37.0
    A = slice(0, 900)
    B = slice(900, 1300)
    C = slice(700, 1000)       # re-sends part of A: additive updates / upserts
    out = {"u": u.astype(np.int64), "i": i.astype(np.int64), "v": v.astype(np.float64), "ts": ts,
           "A": np.array([0, 900]), "B": np.array([900, 1300]), "C": np.array([700, 1000])}
    for name, kw in {"k5": {"nn_feature_selection": 5}, "all": {}, "k5_decay": {"nn_feature_selection": 5, "decay_in_days": 30}}.items():
        model = RefSLIM(min_value=0, max_value=15, **kw)
        steps = [("A", A, False), ("B", B, False), ("C_add", C, False), ("C_upsert", C, True)]
        for label, sl, upsert in steps:
            batch = [(int(a), int(b), float(t), float(r)) for a, b, t, r in zip(u[sl], i[sl], ts[sl], v[sl])]
            model.fit(batch, update_interaction=upsert, progress_bar=False)
            out.update(csc_parts(model.model.item_similarity, f"W_{name}_{label}"))
        users = list(range(0, 150, 7))
        recs = model.recommend_batch(users, top_k=5)
        out[f"rec_{name}"] = pad_lists(recs, 5, -1)
        out[f"rec_users_{name}"] = np.array(users)
    np.savez_compressed(os.path.join(OUT, "partial_fit.npz"), **out)


# ------------------------------------------------------------------ G5 / G6: scoring, top-k, similar items
def gen_scoring(X2, model):
    Xr = X2.tocsr()
    Xr.sort_indices()
    out = {}
    users = list(range(0, 1200, 5))
    out["users"] = np.array(users)
    W32 = model.item_similarity
    W64 = sp.csc_matrix(W32, dtype=np.float64)
    for wname, W in (("f32", W32), ("f64", W64)):
        model.item_similarity = W
        for dense in (False, True):
            for filt in (True, False):
                res = model.recommend_batch(users, Xr, top_k=10, filter_interacted=filt, dense_output=dense, ret_scores=True)
                key = f"{wname}_{'dense' if dense else 'sparse'}_{'filter' if filt else 'nofilter'}"
                out[f"ids_{key}"] = pad_lists([r[0] for r in res], 10, -1)
                out[f"scores_{key}"] = pad_lists([np.asarray(r[1], dtype=np.float64).tolist() for r in res], 10, float("-inf"))
    model.item_similarity = W32
    cands = [3, 399, 17, 250, 251, 252, 8, 120, 77, 301, 5, 64]
    res = model.recommend_batch(users, Xr, candidate_item_ids=cands, top_k=5, ret_scores=True)
    out["cands"] = np.array(cands)
    out["ids_cands"] = pad_lists([r[0] for r in res], 5, -1)
    out["scores_cands"] = pad_lists([np.asarray(r[1], dtype=np.float64).tolist() for r in res], 5, float("-inf"))
    sim_i, sim_s = [], []
    for j in range(400):
        a, b = model.similar_items(j, top_k=6, ret_ndarrays=True)
        sim_i.append(a.tolist())
        sim_s.append(b.astype(np.float64).tolist())
    out["similar_ids"] = pad_lists(sim_i, 6, -1)
    out["similar_scores"] = pad_lists(sim_s, 6, float("-inf"))
    # score-vector export: predict / predict_selected / predict_all (slim_elastic.py:566-626)
    pu = [0, 5, 333, 1199]
    out["predict_users"] = np.array(pu)
    out["predict_dense"] = np.stack([np.asarray(model.predict(u, Xr, dense_output=True)).ravel() for u in pu])
    sp_rows = [model.predict(u, Xr, dense_output=False) for u in pu]
    out["predict_sparse_as_dense"] = np.stack([np.asarray(m.toarray()).ravel() for m in sp_rows])
    out["predict_selected"] = np.stack([np.asarray(model.predict_selected(u, cands, Xr)).ravel() for u in pu])
    out["predict_all_head"] = np.asarray(model.predict_all(Xr[:40], dense_output=True))
    model.item_similarity = W64
    out["predict_dense_f64"] = np.stack([np.asarray(model.predict(u, Xr, dense_output=True)).ravel() for u in pu])
    model.item_similarity = W32
    np.savez_compressed(os.path.join(OUT, "scoring.npz"), **out)


# ------------------------------------------------------------------ G7: interaction store / decay
def gen_store():
    d = {"decay_rate": {}}
    for days in (7, 180, 365):
        d["decay_rate"][str(days)] = RefStore(decay_in_days=days).decay_rate
    rng = np.random.default_rng(11)
    n = 400
    u = rng.integers(0, 30, n)
    i = rng.integers(0, 25, n)
    ts = T0 + np.sort(rng.random(n)) * 40 * 86400
    r = rng.integers(-3, 6, n).astype(float)
    d["events"] = {"u": u.tolist(), "i": i.tolist(), "ts": ts.tolist(), "r": r.tolist()}
    for name, kw, upsert in (("plain", {}, False), ("decay7", {"decay_in_days": 7}, False),
                             ("decay7_upsert", {"decay_in_days": 7}, True), ("clip", {"min_value": -1, "max_value": 4}, False)):
        s = RefStore(**kw)
        for a, b, t, x in zip(u, i, ts, r):
            s.add_interaction(int(a), int(b), float(t), float(x), upsert=upsert)
        d[name] = {"csr": s.to_csr().toarray().astype(np.float64).tolist(),
                   "csc_sel": s.to_csc([1, 3, 5, 24]).toarray().astype(np.float64).tolist(),
                   "csr_sel": s.to_csr([0, 2, 29]).toarray().astype(np.float64).tolist(),
                   "max_timestamp": s.max_timestamp, "shape": list(s.shape),
                   "hot": s.get_hot_items(10, filter_interacted=False),
                   "rating_3_4": s.get_user_item_rating(3, 4), "user_items_5": sorted(s.get_user_items(5))}
    json.dump(d, open(os.path.join(OUT, "store.json"), "w"))


# ------------------------------------------------------------------ G1: the reference's own API scenarios
def gen_api():
    t = T0
    scenarios = {
        "similar_items": [('user_1', 'item_1', t, 5.0), ('user_1', 'item_3', t, 4.0), ('user_1', 'item_4', t, 3.0),
                          ('user_2', 'item_1', t, 3.0), ('user_2', 'item_2', t, -2.0), ('user_2', 'item_4', t, 3.0),
                          ('user_3', 'item_1', t, 4.0), ('user_3', 'item_3', t, 2.0), ('user_3', 'item_4', t, 4.0)],
        "fit_and_recommend": [('user_1', 'item_1', t, 5.0), ('user_2', 'item_2', t, -2.0), ('user_2', 'item_1', t, 3.0),
                              ('user_2', 'item_4', t, 3.0), ('user_1', 'item_3', t, 4.0)],
        "recommend_batch": [('user_1', 'item_1', t, 5.0), ('user_1', 'item_3', t, 4.0), ('user_2', 'item_2', t, 3.0),
                            ('user_2', 'item_4', t, 4.0), ('user_3', 'item_1', t, 2.0), ('user_3', 'item_2', t, 3.0)],
        "int_ids": [(1, 10, t, 5.0), (1, 30, t, 4.0), (2, 20, t, 3.0), (2, 40, t, 4.0), (3, 10, t, 2.0), (3, 20, t, 3.0),
                    (4, 10, t, 1.5), (4, 40, t, 2.5), (4, 30, t, 0.5)],
    }
    out = {}
    m = RefSLIM(); m.fit(scenarios["similar_items"], progress_bar=False)
    out["similar_items"] = {"interactions": scenarios["similar_items"],
                            "similar_item_1": m.similar_items('item_1', top_k=5),
                            "similar_item_1_scores": [[a, float(b)] for a, b in m.similar_items('item_1', top_k=5, ret_scores=True)],
                            "recommend_user_2": m.recommend('user_2', top_k=5)}
    m = RefSLIM(); m.fit(scenarios["fit_and_recommend"], progress_bar=False); m.fit(iter(scenarios["fit_and_recommend"]), progress_bar=False)
    out["fit_and_recommend"] = {"interactions": scenarios["fit_and_recommend"], "recommend_user_1": m.recommend('user_1', top_k=5),
                                "rating_u1_i1": m.interactions.get_user_item_rating(0, 0)}
    m = RefSLIM(); m.fit(scenarios["recommend_batch"], progress_bar=False)
    users = ['user_1', 'user_2', 'user_3']
    out["recommend_batch"] = {"interactions": scenarios["recommend_batch"],
                              "top2": m.recommend_batch(users, top_k=2),
                              "cands": m.recommend_batch(users, candidate_items=['item_1', 'item_2', 'item_3'], top_k=2),
                              "nofilter_u1": m.recommend_batch(['user_1'], top_k=3, filter_interacted=False),
                              "cold": m.recommend_batch(['user_1', 'nobody'], top_k=2)}
    m = RefSLIM(); m.fit(scenarios["int_ids"], progress_bar=False)
    out["int_ids"] = {"interactions": scenarios["int_ids"],
                      "top3": m.recommend_batch([1, 2, 3, 4], top_k=3),
                      "nofilter": m.recommend_batch([1, 2, 3, 4], top_k=3, filter_interacted=False),
                      "cold_99": m.recommend(99, top_k=3),
                      "similar_10": m.similar_items(10, top_k=3)}
    json.dump(out, open(os.path.join(OUT, "api.json"), "w"), indent=1)


# ------------------------------------------------------------------ N3: pickles written by the reference
def gen_reference_pickles():
    """Model files saved by the REFERENCE (data fixtures): rtrec_amd must load them and serve the
    same recommendations."""
    import io
    z = np.load(os.path.join(OUT, "partial_fit.npz"))
    a, b = z["A"]
    batch = [(int(x), int(y), float(t), float(r)) for x, y, t, r in zip(z["u"][a:b], z["i"][a:b], z["ts"][a:b], z["v"][a:b])]
    m = RefSLIM(min_value=0, max_value=15, nn_feature_selection=5, decay_in_days=None)
    m.fit(batch, progress_bar=False)
    buf = io.BytesIO(); m.save(buf)
    open(os.path.join(OUT, "ref_slim_int.pkl"), "wb").write(buf.getvalue())
    users = list(range(0, 150, 5)) + [100000]
    exp = {"int": {"users": users, "recs": m.recommend_batch(users, top_k=5), "similar_3": m.similar_items(3, top_k=4),
                   "csr": m.interactions.to_csr().toarray().tolist(), "hot": m.interactions.get_hot_items(5, filter_interacted=False)}}
    s = RefSLIM()
    t = T0
    inter = [('user_1', 'item_1', t, 5.0), ('user_1', 'item_3', t, 4.0), ('user_1', 'item_4', t, 3.0),
             ('user_2', 'item_1', t, 3.0), ('user_2', 'item_2', t, -2.0), ('user_2', 'item_4', t, 3.0),
             ('user_3', 'item_1', t, 4.0), ('user_3', 'item_3', t, 2.0), ('user_3', 'item_4', t, 4.0)]
    s.fit(inter, progress_bar=False)
    s.register_item_feature('item_1', ['tagA', 'tagB'])
    buf = io.BytesIO(); s.save(buf)
    open(os.path.join(OUT, "ref_slim_str.pkl"), "wb").write(buf.getvalue())
    exp["str"] = {"similar_item_1": s.similar_items('item_1', top_k=5), "rec_user_2": s.recommend('user_2', top_k=5),
                  "item_feature_nnz": int(s.feature_store.build_item_features_matrix(item_ids=[0]).nnz)}
    json.dump(exp, open(os.path.join(OUT, "ref_pickles.json"), "w"))


# ------------------------------------------------------------------ N2: the HTTP shell
def gen_serving():
    """Request/response transcript of the REFERENCE FastAPI app (rtrec/serving/app.py) for a fixed
    request sequence; tests replay it against rtrec_amd.serving.app."""
    from fastapi.testclient import TestClient
    from rtrec.serving.app import create_app
    tok = {"X-Token": "fake_secret_token"}
    str_inter = [{"user": "user1", "item": "item1", "timestamp": 1672531200.0, "rating": 5.0},
                 {"user": "user1", "item": "item2", "timestamp": 1672617600.0, "rating": 3.0},
                 {"user": "user2", "item": "item1", "timestamp": 1672704000.0, "rating": 4.0},
                 {"user": "user2", "item": "item3", "timestamp": 1672704000.0, "rating": 4.0},
                 {"user": "user2", "item": "item4", "timestamp": 1672704000.0, "rating": 3.0},
                 {"user": "user3", "item": "item2", "timestamp": 1672790400.0, "rating": 2.0},
                 {"user": "user3", "item": "item4", "timestamp": 1672790400.0, "rating": 5.0}]
    int_inter = [{"user": 1, "item": 1, "timestamp": 1672531200.0, "rating": 5.0},
                 {"user": 1, "item": 2, "timestamp": 1672617600.0, "rating": 3.0},
                 {"user": 2, "item": 1, "timestamp": 1672704000.0, "rating": 4.0},
                 {"user": 2, "item": 3, "timestamp": 1672704000.0, "rating": 4.0},
                 {"user": 2, "item": 4, "timestamp": 1672704000.0, "rating": 3.0},
                 {"user": 3, "item": 2, "timestamp": 1672790400.0, "rating": 2.0},
                 {"user": 3, "item": 4, "timestamp": 1672790400.0, "rating": 5.0}]
    sessions = {}
    for name, inter, users in (("str", str_inter, ["user1", "user2", "user3", "nobody"]), ("int", int_inter, [1, 2, 3, 99])):
        c = TestClient(create_app())
        steps = [("GET", "/", None, {}), ("POST", "/fit", inter[:5], {"X-Token": "wrong"}), ("POST", "/fit", inter[:5], tok)]
        for u in users:
            steps.append(("POST", "/recommend", {"user": u, "top_k": 5, "filter_interacted": True}, tok))
        steps.append(("POST", "/fit", inter[5:], tok))
        for u in users:
            steps.append(("POST", "/recommend", {"user": u, "top_k": 3, "filter_interacted": False}, tok))
        steps.append(("POST", "/recommend", {"user": users[0]}, {"X-Token": "wrong"}))
        out = []
        for method, path, payload, headers in steps:
            r = c.get(path) if method == "GET" else c.post(path, json=payload, headers=headers)
            out.append({"method": method, "path": path, "json": payload, "headers": headers, "status": r.status_code,
                        "response": r.json()})
        sessions[name] = out
    json.dump(sessions, open(os.path.join(OUT, "serving.json"), "w"), indent=1)


# ------------------------------------------------------------------ N3: ranking metrics + Recommender.evaluate
def gen_evaluate():
    """rtrec/utils/metrics.py:5-313 on seeded (ranked, truth) pairs incl. the empty / short-list corner cases,
    and rtrec/recommender.py:39-82,163-200 end to end: Recommender(SLIM).fit(train) -> evaluate(test)."""
    import pandas as pd
    from rtrec.recommender import Recommender as RefRecommender
    from rtrec.utils import metrics as M
    rng = np.random.default_rng(77)
    pairs = [([], []), ([], [1, 2]), ([3, 4], []), ([1], [1]), ([5, 6, 7], [7]), ([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12], [11, 12, 1])]
    for _ in range(150):
        n_r, n_t = int(rng.integers(0, 14)), int(rng.integers(0, 9))
        pairs.append((rng.permutation(30)[:n_r].tolist(), rng.permutation(30)[:n_t].tolist()))
    names = ["precision", "recall", "f1_score", "ndcg", "hit", "reciprocal_rank", "auc", "true_positives", "average_precision"]
    per_pair = {}
    for size in (1, 5, 10):
        per_pair[str(size)] = [[float(getattr(M, nm)(r, g, size)) for nm in names] for r, g in pairs]
    agg = {str(size): M.compute_scores(iter(pairs), size) for size in (1, 5, 10)}
    agg["mrr_5"] = M.mrr([r for r, _ in pairs], [g for _, g in pairs], 5)
    agg["map_5"] = M.map_score([r for r, _ in pairs], [g for _, g in pairs], 5)

    U, I, n = 300, 120, 7000
    u = (rng.zipf(1.3, n) - 1) % U
    i = (rng.zipf(1.2, n) - 1) % I
    r = np.round(rng.integers(1, 6, n) + rng.random(n), 6)
    ts = T0 + np.sort(rng.random(n)) * 30 * 86400.0
    df = pd.DataFrame({"user": u.astype(int), "item": i.astype(int), "tstamp": ts, "rating": r})
    train, test = df.iloc[:6000], df.iloc[6000:]
    rec = RefRecommender(RefSLIM(nn_feature_selection=8, min_value=0, max_value=15))
    rec.fit(train, batch_size=1000, parallel=False)
    ev = {f"{size}_{int(fi)}": rec.evaluate(test, recommend_size=size, filter_interacted=fi)
          for size in (5, 10) for fi in (True, False)}
    users = sorted(set(test["user"].tolist()))
    recs = rec.recommend_batch(users, top_k=10)
    out = {"metric_names": names, "pairs": pairs, "per_pair": per_pair, "aggregate": agg,
           "e2e": {"train": [train[c].tolist() for c in ("user", "item", "tstamp", "rating")],
                   "test": [test[c].tolist() for c in ("user", "item", "tstamp", "rating")],
                   "model_kwargs": {"nn_feature_selection": 8, "min_value": 0, "max_value": 15},
                   "evaluate": ev, "users": users, "recommend_top10": [[int(x) for x in row] for row in recs]}}
    json.dump(out, open(os.path.join(OUT, "evaluate.json"), "w"))


# ------------------------------------------------------------------ G3 (mid-size): whole models as checksums + long-column known answers
def _crc(a):
    import zlib
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def _column_sweeps(ref_model, X, cols):
    """n_iter_ per target column, through the reference's own per-column objects (FeatureSelectionWrapper + ElasticNet,
    slim_elastic.py:131-154,195-227) and its own column masking (CSCMatrixWrapper, :82-129)."""
    from rtrec.models.internal.slim_elastic import CSCMatrixWrapper
    Xw = CSCMatrixWrapper(X)
    mdl = ref_model.get_model()
    out = []
    for j in cols:
        y = Xw.get_col(j).toarray().ravel()
        keep = X.data[X.indptr[j]:X.indptr[j + 1]].copy()
        Xw.set_col(j, np.zeros_like(keep))
        mdl.fit(X, y)
        out.append(int(mdl.model.n_iter_))
        Xw.set_col(j, keep)
    return np.array(out, dtype=np.int32)


def gen_midsize():
    """VERDICT round 2 item 4 / SURVEY 8c G3: (i) whole-model W at the ML-1M shape and at a 3000 x 1500 structured matrix, K = 50,
    from the real SLIMElastic.partial_fit_items -- stored as CRC32 of the CSC arrays (value BITS) plus every column's n_iter_;
    (ii) ElasticNet known answers for 80 LONG target columns (12k .. 128k entries) of the ML-20M-shape matrix with their 50
    selected features: coefficient bits and n_iter_, where the order of the BLAS reductions in the duality gap (DESIGN D2)
    would bite if it ever did.  The inputs are regenerated from their seeds by the tests (rtrec_amd.synth)."""
    from rtrec_amd.synth import structured_matrix
    out = {}
    for name, X in (("ml1m", interaction_matrix(6040, 3706, 1_000_000, seed=20251003)),
                    ("s3000", structured_matrix(3000, 1500, 200_000, seed=77, n_clusters=12))):
        X = X.tocsc()
        X.sort_indices()
        I = X.shape[1]
        m = RefSLIMElastic({"nn_feature_selection": 50}).partial_fit_items(X.copy(), list(range(I)))
        W = m.item_similarity.tocsc()
        W.sort_indices()
        out[name] = {"shape": list(X.shape), "nnz_X": int(X.nnz), "crc_X": [_crc(X.indptr.astype(np.int32)), _crc(X.indices.astype(np.int32)),
                                                                           _crc(X.data.astype(np.float32))],
                     "W_nnz": int(W.nnz), "W_dtype": str(W.dtype), "W_rows_nonempty": int(np.count_nonzero(np.diff(W.tocsr().indptr))),
                     "crc_W_indptr": _crc(W.indptr.astype(np.int32)), "crc_W_indices": _crc(W.indices.astype(np.int32)),
                     "crc_W_bits": _crc(W.data.astype(np.float32).view(np.uint32)),
                     "n_iter": _column_sweeps(RefSLIMElastic({"nn_feature_selection": 50}), X.copy(), range(I)).tolist()}
        print(f"[golden] {name}: W nnz {W.nnz}, rows {out[name]['W_rows_nonempty']}")
    # (ii) long columns of the ML-20M shape
    X = interaction_matrix(138_493, 26_744, 26_000_000, seed=20251003).tocsc()
    X.sort_indices()
    nnz = np.diff(X.indptr)
    by_len = np.argsort(-nnz, kind="stable")
    cand = by_len[nnz[by_len] >= 12_000]
    pick = np.unique(np.concatenate([cand[:24], cand[np.linspace(0, len(cand) - 1, 56).astype(int)]]))[:80]
    from rtrec.models.internal.slim_elastic import CSCMatrixWrapper
    Xw = CSCMatrixWrapper(X)
    mdl = RefSLIMElastic({"nn_feature_selection": 50}).get_model()
    feats, coefs, iters = [], [], []
    for j in pick:
        y = Xw.get_col(int(j)).toarray().ravel()
        keep = X.data[X.indptr[j]:X.indptr[j + 1]].copy()
        Xw.set_col(int(j), np.zeros_like(keep))
        mdl.fit(X, y)
        c = mdl.sparse_coef_.tocsr()
        o = np.argsort(c.indices, kind="stable")
        feats.append(c.indices[o].astype(np.int32))
        coefs.append(c.data[o].astype(np.float32))
        iters.append(int(mdl.model.n_iter_))
        Xw.set_col(int(j), keep)
    out["long_columns"] = {"shape": [138_493, 26_744], "draws": 26_000_000, "seed": 20251003, "nnz_X": int(X.nnz),
                           "targets": pick.astype(int).tolist(), "target_nnz": nnz[pick].astype(int).tolist(),
                           "features": np.stack(feats).tolist(), "coef_bits": np.stack(coefs).view(np.uint32).tolist(),
                           "n_iter": iters}
    print(f"[golden] long columns: {len(pick)} targets, {nnz[pick].min()} .. {nnz[pick].max()} entries, sweeps {min(iters)} .. {max(iters)}")
    json.dump(out, open(os.path.join(OUT, "midsize.json"), "w"))


# ------------------------------------------------------------------ N4: the call sequence HybridSlimFM makes on its SLIM half
def gen_hybrid_calls():
    """HybridSlimFM (rtrec/models/hybrid.py) drives a SLIMElastic through a fixed set of calls: :122 SLIMElastic(kwargs), :151 /
    :196 partial_fit_items(coo.tocsc(), item_ids[, parallel]), :217 fit(csc, parallel), :225 recommend(...), :267 recommend(...,
    ret_scores=True), :381 recommend_batch(..., ret_scores=False), :409 recommend_batch(..., ret_scores=True), :477
    similar_items(q, top_k, ret_ndarrays=True).  hybrid.py itself needs lightfm / implicit (not installed); its SLIM half is
    the real SLIMElastic, so the sequence is replayed on it and every answer recorded (float ratings: no score ties)."""
    rng = np.random.default_rng(17)
    X = interaction_matrix(260, 90, 5200, seed=404).tocoo()
    order = rng.permutation(X.nnz)
    u, i, v = X.row[order], X.col[order], X.data[order]
    cut = int(X.nnz * 0.7)
    out = {"u": u.tolist(), "i": i.tolist(), "v": v.astype(float).tolist(), "cut": cut, "kwargs": {"nn_feature_selection": 12, "alpha": 0.05}}
    m = RefSLIMElastic(dict(out["kwargs"]))
    A = sp.coo_matrix((v[:cut], (u[:cut], i[:cut])), shape=(260, 90), dtype=np.float32)
    items_a = sorted(set(i[:cut].tolist()))
    m.partial_fit_items(A.tocsc(copy=False), items_a, progress_bar=False)                       # hybrid.py:151
    B = sp.coo_matrix((v, (u, i)), shape=(260, 90), dtype=np.float32)
    items_b = sorted(set(i[cut:].tolist()))
    m.partial_fit_items(B.tocsc(copy=False), items_b, parallel=True, progress_bar=False)        # hybrid.py:196
    out["items_a"], out["items_b"] = items_a, items_b
    out.update({k: (a.tolist() if hasattr(a, "tolist") else a) for k, a in csc_parts(m.item_similarity, "W_after_b").items()})
    Br = B.tocsr()
    users = [0, 7, 33, 120, 259]
    cands = [3, 88, 17, 41, 5, 64, 70]
    rec = {}
    for uu in users:
        row = Br          # (the reference indexes the matrix it is given with the user id: hybrid passes to_csr(select_users=[u]))
        rec[str(uu)] = {
            "plain": m.recommend(uu, row, candidate_item_ids=None, top_k=6, filter_interacted=True, dense_output=False),       # :225
            "scores": [a.tolist() if hasattr(a, "tolist") else a for a in
                       m.recommend(uu, row, candidate_item_ids=None, top_k=6, filter_interacted=True, dense_output=False, ret_scores=True)],  # :267
            "cands": [a.tolist() if hasattr(a, "tolist") else a for a in
                      m.recommend(uu, row, candidate_item_ids=cands, top_k=4, filter_interacted=False, dense_output=False, ret_scores=True)],
            "dense": m.recommend(uu, row, candidate_item_ids=None, top_k=6, filter_interacted=True, dense_output=True)}
    out["users"], out["cands"], out["recommend"] = users, cands, rec
    sub = Br
    out["batch_plain"] = m.recommend_batch(users, sub, candidate_item_ids=None, top_k=6, filter_interacted=True, dense_output=False,
                                           ret_scores=False)                                                                      # :381
    out["batch_scores"] = [[a.tolist() if hasattr(a, "tolist") else a for a in r] for r in
                           m.recommend_batch(users, sub, candidate_item_ids=None, top_k=6, filter_interacted=False, dense_output=False,
                                             ret_scores=True)]                                                                    # :409
    sim = {}
    for q in (0, 5, 41, 89):
        a, b = m.similar_items(q, top_k=5, ret_ndarrays=True)                                                                      # :477
        sim[str(q)] = [a.tolist(), b.astype(float).tolist()]
    out["similar"] = sim
    m.fit(B.tocsc(), parallel=False, progress_bar=False)                                                                            # :217
    out.update({k: (a.tolist() if hasattr(a, "tolist") else a) for k, a in csc_parts(m.item_similarity, "W_after_fit").items()})
    json.dump(out, open(os.path.join(OUT, "hybrid_calls.json"), "w"))
    print("[golden] hybrid call sequence recorded")


# ------------------------------------------------------------------ N4: optim="sgd" (slim_elastic.py:209-222)
def gen_sgd():
    """SLIMElastic({"optim": "sgd", "nn_feature_selection": K, ...}) of the real reference (scikit-learn SGDRegressor behind
    FeatureSelectionWrapper): whole models (CSC arrays, float32 values of the float64 matrix) plus SGDRegressor.n_iter_ of every
    column (the reference does not keep it: the column loop of slim_elastic.py:261-277 is replayed with its own wrapper), a
    partial_fit_items on top (stale / replaced entries), and the failure without feature selection (:273).  Inputs are the
    seeded synthetic matrices of rtrec_amd.synth (regenerated by the tests), float ratings."""
    from sklearn.linear_model import SGDRegressor  # noqa: F401
    out = {"cases": []}
    cases = [dict(U=300, I=40, draws=3000, seed=3, cfg={"nn_feature_selection": 8}),
             dict(U=1200, I=200, draws=30000, seed=3, cfg={"nn_feature_selection": 20}),
             dict(U=2500, I=300, draws=90000, seed=5, cfg={"nn_feature_selection": 50, "alpha": 0.01, "eta0": 0.01}),
             dict(U=800, I=120, draws=20000, seed=7, cfg={"nn_feature_selection": 10, "max_iter": 7}),
             dict(U=900, I=150, draws=24000, seed=11, cfg={"nn_feature_selection": 64, "l1_ratio": 0.5, "tol": 1e-3, "random_state": 7}),
             # more than one feature per wave lane on the device (two up to K = 128, four up to 256)
             dict(U=700, I=130, draws=16000, seed=13, cfg={"nn_feature_selection": 100}),
             dict(U=500, I=140, draws=40000, seed=17, cfg={"nn_feature_selection": 130, "max_iter": 12, "eta0": 0.01})]      # (no X^T y tie at the K-th place: D1)
    for c in cases:
        X = interaction_matrix(c["U"], c["I"], c["draws"], seed=c["seed"]).tocsc()
        X.sort_indices()
        cfg = dict(c["cfg"], optim="sgd")
        m = RefSLIMElastic(dict(cfg))
        m.fit(X.copy())
        W = m.item_similarity.tocsc()
        W.sort_indices()
        n_iter = []
        for j in range(c["I"]):
            model = m.get_model()
            Xm = X.copy()
            y = Xm.getcol(j).toarray().ravel()
            Xm.data[Xm.indptr[j]:Xm.indptr[j + 1]] = 0
            model.fit(Xm, y)
            n_iter.append(int(model.model.n_iter_))
        rec = dict(c, W_dtype=str(W.dtype), W_indptr=W.indptr.tolist(), W_indices=W.indices.tolist(),
                   W_bits=np.ascontiguousarray(W.data, dtype=np.float32).view(np.uint32).tolist(),
                   W_is_float32_valued=bool(np.array_equal(W.data, W.data.astype(np.float32).astype(W.dtype))), n_iter=n_iter)
        if c["I"] == 120:        # an incremental refit on top (float32 branch of partial_fit_items, stale entries survive)
            items = [3, 4, 50, 119]
            m.partial_fit_items(X.copy(), items)
            W2 = m.item_similarity.tocsc()
            W2.sort_indices()
            rec.update(partial_items=items, W2_dtype=str(W2.dtype), W2_indptr=W2.indptr.tolist(), W2_indices=W2.indices.tolist(),
                       W2_bits=np.ascontiguousarray(W2.data, dtype=np.float32).view(np.uint32).tolist())
        out["cases"].append(rec)
        print(f"[golden] sgd {c['U']}x{c['I']} K={cfg['nn_feature_selection']}: W nnz {W.nnz} dtype {W.dtype}, epochs {min(n_iter)}..{max(n_iter)}")
    try:
        RefSLIMElastic({"optim": "sgd"}).fit(interaction_matrix(60, 12, 300, seed=1).tocsc())
        out["no_feature_selection"] = None
    except Exception as e:          # the reference itself fails here: SGDRegressor has no sparse_coef_
        out["no_feature_selection"] = {"type": type(e).__name__, "message": str(e)}
    json.dump(out, open(os.path.join(OUT, "sgd.json"), "w"))
    print("[golden] sgd.json", os.path.getsize(os.path.join(OUT, "sgd.json")), out["no_feature_selection"])


if __name__ == "__main__":
    if "--sgd" in sys.argv:
        gen_sgd()
        sys.exit(0)
    if "--hybrid" in sys.argv:
        gen_hybrid_calls()
        sys.exit(0)
    if "--midsize" in sys.argv:
        gen_midsize()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "evaluate":     # adds evaluate.json without touching the other fixtures
        gen_evaluate()
        sys.exit(0)
    gen_rng()
    gen_cd_columns()
    X, X2, model = gen_models()
    gen_partial()
    gen_scoring(X2, model)
    gen_store()
    gen_api()
    gen_reference_pickles()
    gen_serving()
    gen_evaluate()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
