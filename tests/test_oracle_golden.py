"""The CPU oracle pinned against golden vectors produced by the REAL reference
(tools/gen_golden.py: rtrec + scikit-learn 1.7.2 + scipy 1.15.3, run in the build container).

Bar: bit-exact -- coefficient bits, sweep counts, W structure, score bits, top-k ids.
"""
import json
import os

import numpy as np
import pytest
import scipy.sparse as sp

from rtrec_amd.engine import merge_coefficients

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


def test_rng_matches_sklearn(oracle):
    d = json.load(open(os.path.join(G, "rng.json")))
    assert d["43"]["seed"] == 494155588
    for rs, ref in d.items():
        seed = oracle.sklearn_seed(int(rs))
        assert seed == ref["seed"]
        assert oracle.rand_sequence(seed, 12, 50).tolist() == ref["mod50"]
        assert oracle.rand_sequence(seed, 12, 3707).tolist() == ref["mod3707"]


@pytest.mark.parametrize("name,kw", [("default", {}), ("nonpositive", {"positive": False}),
                                     ("loose", {"tol": 1e-2, "max_iter": 5}), ("strong", {"alpha": 0.5, "l1_ratio": 0.5})])
def test_cd_matches_sklearn_elasticnet(oracle, name, kw):
    z = np.load(os.path.join(G, "cd_columns.npz"))
    X = load_csc(z, "X")
    for j in range(X.shape[1]):
        Xj = X.copy()
        y = Xj[:, j].toarray().ravel()
        Xj.data[Xj.indptr[j]:Xj.indptr[j + 1]] = 0
        w, _, n_iter = oracle.cd(Xj, y, **kw)
        assert n_iter == z[f"{name}_n_iter"][j], f"column {j}"
        assert np.array_equal(bits(w), bits(z[f"{name}_coef"][j])), f"column {j}"


@pytest.mark.parametrize("name,kw", [("serial_all", {}), ("serial_k8", {"nn_feature_selection": 8}),
                                     ("partial_all", {}), ("partial_k8", {"nn_feature_selection": 8}),
                                     ("partial_k100", {"nn_feature_selection": 100}),
                                     ("parallel_k8", {"nn_feature_selection": 8}),
                                     ("nonpos_k8", {"nn_feature_selection": 8, "positive": False})])
def test_fit_columns_matches_slimelastic(oracle, name, kw):
    z = np.load(os.path.join(G, "models.npz"))
    X = load_csc(z, "X")
    W_ref = load_csc(z, f"W_{name}")
    I = X.shape[1]
    ptr, idx, val, _ = oracle.fit_columns(X, np.arange(I), **kw)
    W = merge_coefficients(None, I, idx.astype(np.int64), np.repeat(np.arange(I), np.diff(ptr)), val)
    assert same_matrix(W, W_ref)
    assert str(z[f"W_{name}_dtype"]) == ("float64" if name.startswith("serial") else "float32")


def test_fit_k50_midsize(oracle):
    z = np.load(os.path.join(G, "models.npz"))
    X = load_csc(z, "X2")
    ptr, idx, val, _ = oracle.fit_columns(X, np.arange(400), nn_feature_selection=50)
    W = merge_coefficients(None, 400, idx.astype(np.int64), np.repeat(np.arange(400), np.diff(ptr)), val)
    assert same_matrix(W, load_csc(z, "W2_k50"))


def untied_prefix(scores_row):
    """Length of the leading part of a descending score list that is free of exact ties."""
    n = int(np.sum(np.isfinite(scores_row)))
    for k in range(n - 1):
        if scores_row[k] == scores_row[k + 1]:
            return k
    return n


@pytest.mark.parametrize("w", ["f32", "f64"])
@pytest.mark.parametrize("dense", [False, True])
@pytest.mark.parametrize("filt", [True, False])
def test_recommend_matches_reference(oracle, w, dense, filt):
    zm = np.load(os.path.join(G, "models.npz"))
    zs = np.load(os.path.join(G, "scoring.npz"))
    X = load_csc(zm, "X2").tocsr()
    W = load_csc(zm, "W2_k50").tocsr()
    users = zs["users"]
    key = f"{w}_{'dense' if dense else 'sparse'}_{'filter' if filt else 'nofilter'}"
    ids, sc, cnt = oracle.recommend_batch(X[users], W, top_k=10, filter_interacted=filt, dense=dense, use_f64=(w == "f64"))
    g_ids, g_sc = zs[f"ids_{key}"], zs[f"scores_{key}"]
    for r in range(len(users)):
        n_ref = int(np.sum(g_ids[r] >= 0))
        assert cnt[r] == n_ref
        # sparse path: stable sort -> fully determined.  dense path: numpy's unstable argsort
        # leaves tied (zero) scores in unspecified order -> compare the untied prefix (D1)
        n = n_ref if not dense else untied_prefix(g_sc[r])
        assert ids[r, :n].tolist() == g_ids[r, :n].tolist(), f"user {users[r]}"
        assert np.array_equal(bits(sc[r, :n]), bits(g_sc[r, :n]))


def test_similar_items_matches_reference(oracle):
    zm = np.load(os.path.join(G, "models.npz"))
    zs = np.load(os.path.join(G, "scoring.npz"))
    W = load_csc(zm, "W2_k50")
    for j in range(400):
        oi, ov = oracle.similar_items(W, j, top_k=6)
        n = untied_prefix(zs["similar_scores"][j])
        assert len(oi) == int(np.sum(zs["similar_ids"][j] >= 0))
        assert oi[:n].tolist() == zs["similar_ids"][j, :n].tolist()
        assert np.array_equal(bits(ov[:n]), bits(zs["similar_scores"][j, :n]))


# ------------------------------------------------------------------ mid-size fixtures (tools/gen_golden.py --midsize)
# Whole models at the ML-1M shape and at a structured 3000 x 1500 matrix (real SLIMElastic.partial_fit_items, K = 50) as CRC32 of
# the CSC arrays + every column's n_iter_, and ElasticNet known answers for 80 long columns (12k .. 128k entries) of the ML-20M
# shape: SURVEY 8c G3, where the BLAS reduction order of the duality gap (DESIGN D2) would show if it ever mattered.
def _crc(a):
    import zlib
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def midsize():
    return json.load(open(os.path.join(G, "midsize.json")))


def midsize_matrix(name):
    from rtrec_amd.synth import interaction_matrix, structured_matrix
    if name == "ml1m":
        X = interaction_matrix(6040, 3706, 1_000_000, seed=20251003)
    elif name == "s3000":
        X = structured_matrix(3000, 1500, 200_000, seed=77, n_clusters=12)
    else:
        X = interaction_matrix(138_493, 26_744, 26_000_000, seed=20251003)
    X = X.tocsc()
    X.sort_indices()
    return X


def check_model_crc(ref, W, n_iter):
    W = W.tocsc()
    W.sort_indices()
    assert W.nnz == ref["W_nnz"]
    assert _crc(W.indptr.astype(np.int32)) == ref["crc_W_indptr"]
    assert _crc(W.indices.astype(np.int32)) == ref["crc_W_indices"]
    assert _crc(W.data.astype(np.float32).view(np.uint32)) == ref["crc_W_bits"], "coefficient bits differ from the reference's W"
    assert np.array_equal(np.asarray(n_iter), np.asarray(ref["n_iter"])), "sweep counts differ from scikit-learn's n_iter_"


@pytest.mark.parametrize("name", ["ml1m", "s3000"])
def test_midsize_model_checksums(oracle, name):
    ref = midsize()[name]
    X = midsize_matrix(name)
    assert [_crc(X.indptr.astype(np.int32)), _crc(X.indices.astype(np.int32)), _crc(X.data.astype(np.float32))] == ref["crc_X"], \
        "the synthetic generator drifted: regenerate tests/golden/midsize.json"
    I = X.shape[1]
    ptr, idx, val, nit = oracle.fit_columns(X, np.arange(I), nn_feature_selection=50, n_threads=8)
    W = merge_coefficients(None, I, idx.astype(np.int64), np.repeat(np.arange(I, dtype=np.int64), np.diff(ptr)), val)
    check_model_crc(ref, W, nit)


def test_midsize_long_columns(oracle):
    ref = midsize()["long_columns"]
    X = midsize_matrix("c3")
    assert X.nnz == ref["nnz_X"]
    tg = np.asarray(ref["targets"], dtype=np.int64)
    assert np.array_equal(np.diff(X.indptr)[tg], ref["target_nnz"])
    ptr, idx, val, nit = oracle.fit_columns(X, tg, nn_feature_selection=50, n_threads=8)
    assert np.array_equal(nit, ref["n_iter"])
    assert np.array_equal(np.diff(ptr), np.full(len(tg), 50))
    assert np.array_equal(idx.reshape(len(tg), 50), np.asarray(ref["features"]))
    assert np.array_equal(bits(val).reshape(len(tg), 50), np.asarray(ref["coef_bits"], dtype=np.uint32))


def _sgd_cases():
    return json.load(open(os.path.join(G, "sgd.json")))


def _golden_w(c, prefix="W"):
    bits_ = np.asarray(c[f"{prefix}_bits"], dtype=np.uint32)
    return sp.csc_matrix((bits_.view(np.float32), np.asarray(c[f"{prefix}_indices"]), np.asarray(c[f"{prefix}_indptr"])),
                         shape=(c["I"], c["I"]))


@pytest.mark.parametrize("case", range(7))
def test_sgd_oracle_matches_the_reference(oracle, case):
    """optim="sgd" (slim_elastic.py:209-222; scikit-learn SGDRegressor behind FeatureSelectionWrapper): the oracle's restatement
    of _plain_sgd32 reproduces the real reference's W bit for bit and SGDRegressor.n_iter_ of every column
    (tests/golden/sgd.json, tools/gen_golden.py --sgd)."""
    from rtrec_amd.synth import interaction_matrix
    c = _sgd_cases()["cases"][case]
    X = interaction_matrix(c["U"], c["I"], c["draws"], seed=c["seed"]).tocsc()
    X.sort_indices()
    cfg = dict(c["cfg"])
    K = cfg.pop("nn_feature_selection")
    assert c["W_dtype"] == "float64" and c["W_is_float32_valued"]
    ptr, idx, val, nit = oracle.fit_columns_sgd(X, np.arange(c["I"]), nn_feature_selection=K, **cfg)
    W = merge_coefficients(None, c["I"], idx.astype(np.int64), np.repeat(np.arange(c["I"], dtype=np.int64), np.diff(ptr)), val)
    Wg = _golden_w(c)
    W.sort_indices()
    assert np.array_equal(W.indptr, Wg.indptr) and np.array_equal(W.indices, Wg.indices)
    assert np.array_equal(W.data.view(np.uint32), Wg.data.view(np.uint32))
    assert nit.tolist() == c["n_iter"]


def test_sgd_without_feature_selection_fails_like_the_reference(oracle):
    g = _sgd_cases()["no_feature_selection"]
    assert g == {"type": "AttributeError", "message": "'SGDRegressor' object has no attribute 'sparse_coef_'"}
    with pytest.raises(AttributeError, match="sparse_coef_"):
        oracle.fit_columns_sgd(sp.identity(4, format="csc", dtype=np.float32), [0])



# ---- the CPU model of the device's binade-speculative fold (oracle/fold_model.c) ----
@pytest.mark.parametrize("kind", [-1, 0, 1, 2, 3, 4, 5, 6, 7])
def test_fold_model_equals_the_sequential_sum_on_generated_streams(oracle, kind):
    for G in (1, 2, 4):                                                  # 1.08e6 sums over the nine generators, per window size
        bad, st = oracle.fold_fuzz(4242 + kind + 100 * G, 40_000, 3000, kind, groups_per_window=G)
        assert bad == 0, f"G={G}"
        assert st.entries > 3e7 and st.spec_entries + st.serial_entries == st.entries


def test_fold_model_inside_the_oracle_cd_keeps_every_golden_bit(oracle):
    """The oracle's coordinate descent with its dot products routed through the fold model: coefficients and sweep counts
    of the reference-generated goldens are unchanged (and most entries take the integer path)."""
    z = np.load(os.path.join(G, "cd_columns.npz"))
    X = load_csc(z, "X")
    oracle.set_fold_model(True)
    try:
        for name, kw in [("default", {}), ("nonpositive", {"positive": False})]:
            for j in range(X.shape[1]):
                Xj = X.copy()
                y = Xj[:, j].toarray().ravel()
                Xj.data[Xj.indptr[j]:Xj.indptr[j + 1]] = 0
                w, _, n_iter = oracle.cd(Xj, y, **kw)
                assert n_iter == z[f"{name}_n_iter"][j], f"column {j}"
                assert np.array_equal(bits(w), bits(z[f"{name}_coef"][j])), f"column {j}"
        st = oracle.fold_model_stats()
        assert st.entries > 0 and st.spec_entries + st.serial_entries == st.entries
    finally:
        oracle.set_fold_model(False)
