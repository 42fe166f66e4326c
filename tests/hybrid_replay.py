"""Replay of the call sequence HybridSlimFM makes on its SLIM half (rtrec/models/hybrid.py:122,151,196,217,225,267,381,409,477)
against answers recorded from the real SLIMElastic (tests/golden/hybrid_calls.json, tools/gen_golden.py --hybrid).  Used by the CPU
suite (oracle backend: the host logic of the facade) and the GPU suite (the product path)."""
import json
import os

import numpy as np
import scipy.sparse as sp

G = os.path.join(os.path.dirname(__file__), "golden")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def same(W, ref, prefix):
    R = sp.csc_matrix((np.asarray(ref[f"{prefix}_data"]), np.asarray(ref[f"{prefix}_indices"]), np.asarray(ref[f"{prefix}_indptr"])),
                      shape=tuple(ref[f"{prefix}_shape"]))
    W = W.tocsc()
    W.sort_indices()
    R.sort_indices()
    return (W.shape == R.shape and str(W.dtype) == ref[f"{prefix}_dtype"] and np.array_equal(W.indptr, R.indptr)
            and np.array_equal(W.indices, R.indices) and np.array_equal(bits(W.data), bits(R.data)))


def replay(make_model):
    d = json.load(open(os.path.join(G, "hybrid_calls.json")))
    u, i, v, cut = np.asarray(d["u"]), np.asarray(d["i"]), np.asarray(d["v"], dtype=np.float32), d["cut"]
    m = make_model(dict(d["kwargs"]))                                                                   # hybrid.py:122
    A = sp.coo_matrix((v[:cut], (u[:cut], i[:cut])), shape=(260, 90), dtype=np.float32)
    m.partial_fit_items(A.tocsc(copy=False), d["items_a"], progress_bar=False)                          # :151
    B = sp.coo_matrix((v, (u, i)), shape=(260, 90), dtype=np.float32)
    m.partial_fit_items(B.tocsc(copy=False), d["items_b"], parallel=True, progress_bar=False)           # :196
    assert same(m.item_similarity, d, "W_after_b")
    Br = B.tocsr()
    users, cands = d["users"], d["cands"]
    for uu in users:
        ref = d["recommend"][str(uu)]
        assert m.recommend(uu, Br, candidate_item_ids=None, top_k=6, filter_interacted=True, dense_output=False) == ref["plain"]   # :225
        ids, sc = m.recommend(uu, Br, candidate_item_ids=None, top_k=6, filter_interacted=True, dense_output=False, ret_scores=True)  # :267
        assert list(ids) == ref["scores"][0] and np.array_equal(bits(sc), bits(ref["scores"][1]))
        ids, sc = m.recommend(uu, Br, candidate_item_ids=cands, top_k=4, filter_interacted=False, dense_output=False, ret_scores=True)
        assert list(ids) == ref["cands"][0] and np.array_equal(bits(sc), bits(ref["cands"][1]))
        assert m.recommend(uu, Br, candidate_item_ids=None, top_k=6, filter_interacted=True, dense_output=True) == ref["dense"]
    assert m.recommend_batch(users, Br, candidate_item_ids=None, top_k=6, filter_interacted=True, dense_output=False,
                             ret_scores=False) == d["batch_plain"]                                       # :381
    out = m.recommend_batch(users, Br, candidate_item_ids=None, top_k=6, filter_interacted=False, dense_output=False, ret_scores=True)  # :409
    for (ids, sc), ref in zip(out, d["batch_scores"]):
        assert list(ids) == ref[0] and np.array_equal(bits(sc), bits(ref[1]))
    for q, ref in d["similar"].items():
        ids, sc = m.similar_items(int(q), top_k=5, ret_ndarrays=True)                                    # :477
        assert isinstance(ids, np.ndarray) and isinstance(sc, np.ndarray)
        assert ids.tolist() == ref[0] and np.array_equal(bits(sc), bits(ref[1]))
    m.fit(B.tocsc(), parallel=False, progress_bar=False)                                                 # :217
    assert same(m.item_similarity, d, "W_after_fit")
