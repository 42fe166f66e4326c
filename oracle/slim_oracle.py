"""ctypes front-end of the CPU oracle (oracle/slim_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/slim_oracle.h.  Imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg; never by rtrec_amd.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libslim_oracle.so")
_lib: Optional[C.CDLL] = None

_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class OracleCfg(C.Structure):
    _fields_ = [("l1_reg", C.c_float), ("l2_reg", C.c_float), ("tol", C.c_float),
                ("max_iter", C.c_int32), ("seed", C.c_uint32), ("positive", C.c_int32),
                ("top_features", C.c_int32)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the .so is missing or stale."""
    src = os.path.join(_HERE, "slim_oracle.c")
    hdr = os.path.join(_HERE, "slim_oracle.h")
    fold = os.path.join(_HERE, "fold_model.c")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr), os.path.getmtime(fold)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.slim_oracle_rand_r.restype = C.c_uint32
        L.slim_oracle_rand_r.argtypes = [C.POINTER(C.c_uint32)]
        L.slim_oracle_cd.restype = C.c_int32
        L.slim_oracle_cd.argtypes = [C.c_int32, C.c_int32, _f32p, _i32p, _i32p, _f32p,
                                     C.POINTER(OracleCfg), _f32p, _f32p, _f32p, C.POINTER(C.c_float)]
        L.slim_oracle_feature_scores.restype = None
        L.slim_oracle_feature_scores.argtypes = [C.c_int32, _f32p, _i32p, _i32p, _f32p, C.c_int32, _f32p]
        L.slim_oracle_select_topk.restype = C.c_int32
        L.slim_oracle_select_topk.argtypes = [C.c_int32, _f32p, C.c_int32, _i32p]
        L.slim_oracle_fit_column.restype = C.c_int32
        L.slim_oracle_fit_column.argtypes = [C.c_int32, C.c_int32, _f32p, _i32p, _i32p, C.c_int32,
                                             C.c_double, C.c_double, C.c_double, C.c_int32, C.c_uint32,
                                             C.c_int32, C.c_int32, _i32p, _f32p,
                                             C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.slim_oracle_fit_columns.restype = C.c_int64
        L.slim_oracle_fit_columns.argtypes = [C.c_int32, C.c_int32, _f32p, _i32p, _i32p, C.c_int32, _i32p,
                                              C.c_double, C.c_double, C.c_double, C.c_int32, C.c_uint32,
                                              C.c_int32, C.c_int32, _i64p, _i32p, _f32p, _i32p]
        L.slim_oracle_similar_items.restype = C.c_int32
        L.slim_oracle_similar_items.argtypes = [_i32p, _i32p, _f32p, C.c_int32, C.c_int32, _i32p, _f32p]
        L.slim_oracle_recommend_batch.restype = None
        L.slim_oracle_recommend_batch.argtypes = [C.c_int32, _i32p, _i32p, _f32p, _i32p, _i32p, _f32p,
                                                  C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                  _i32p, _f32p, _i32p]
        L.slim_oracle_fit_column_sgd.restype = C.c_int32
        L.slim_oracle_fit_column_sgd.argtypes = [C.c_int32, C.c_int32, _f32p, _i32p, _i32p, C.c_int32, C.c_double, C.c_double,
                                                 C.c_double, C.c_double, C.c_int32, C.c_uint32, C.c_int32, _i32p, _f32p,
                                                 C.POINTER(C.c_int32)]
        L.slim_oracle_fit_columns_mt.restype = C.c_int32
        L.slim_oracle_fit_columns_mt.argtypes = [C.c_int32, C.c_int32, _f32p, _i32p, _i32p, C.c_int32, _i32p,
                                                 C.c_double, C.c_double, C.c_double, C.c_int32, C.c_uint32,
                                                 C.c_int32, C.c_int32, C.c_int32, _i32p, _i32p, _f32p, _i32p, C.c_int32]
        L.slim_oracle_recommend_batch_mt.restype = C.c_int32
        L.slim_oracle_recommend_batch_mt.argtypes = [C.c_int32, _i32p, _i32p, _f32p, _i32p, _i32p, _f32p,
                                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                     _i32p, _f32p, _i32p, C.c_int32]
        _lib = L
    return _lib


def sklearn_seed(random_state: int = 43) -> int:
    """check_random_state(rs).randint(0, 2**31-1): the xorshift seed of _cd_fast.pyx:367."""
    return int(np.random.RandomState(random_state).randint(0, 2147483647))


def rand_sequence(seed: int, n: int, modulo: int) -> np.ndarray:
    st = C.c_uint32(seed)
    L = lib()
    return np.array([L.slim_oracle_rand_r(C.byref(st)) % modulo for _ in range(n)], dtype=np.int64)


def _csc(X):
    return (np.ascontiguousarray(X.data, dtype=np.float32),
            np.ascontiguousarray(X.indices, dtype=np.int32),
            np.ascontiguousarray(X.indptr, dtype=np.int32))


def cd(X_csc, y: np.ndarray, alpha=0.1, l1_ratio=0.1, tol=1e-4, max_iter=100,
       random_state=43, positive=True) -> Tuple[np.ndarray, float, int]:
    """ElasticNet(...).fit(X_csc, y) restated: returns (coef_, raw gap, n_iter_)."""
    n_samples, n_features = X_csc.shape
    d, i, p = _csc(X_csc)
    cfg = OracleCfg(np.float32(alpha * l1_ratio * n_samples), np.float32(alpha * (1.0 - l1_ratio) * n_samples),
                    np.float32(tol), max_iter, sklearn_seed(random_state), int(bool(positive)), 0)
    w = np.zeros(max(n_features, 1), dtype=np.float32)
    R = np.empty(max(n_samples, 1), dtype=np.float32)
    XtA = np.empty(max(n_features, 1), dtype=np.float32)
    gap = C.c_float(0)
    n_iter = lib().slim_oracle_cd(n_samples, n_features, d, i, p, np.ascontiguousarray(y, dtype=np.float32),
                                  C.byref(cfg), w, R, XtA, C.byref(gap))
    return w[:n_features], float(gap.value), int(n_iter)


def fit_columns_sgd(X_csc, cols, alpha=0.1, l1_ratio=0.1, eta0=0.001, tol=1e-4, max_iter=100, random_state=43,
                    nn_feature_selection=None):
    """optim="sgd" behind FeatureSelectionWrapper (slim_elastic.py:139-154, 209-222): per target column the K selected items
    (ascending) with SGDRegressor's coef_, and its n_iter_ (epochs): (ptr, idx, val, n_iter)."""
    U, I = X_csc.shape
    d, i, p = _csc(X_csc)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    if nn_feature_selection is None:
        raise AttributeError("'SGDRegressor' object has no attribute 'sparse_coef_'")     # what the reference does (:273)
    K = min(int(nn_feature_selection), I)
    ptr = np.zeros(len(cols) + 1, dtype=np.int64)
    idx = np.empty((len(cols), K), dtype=np.int32)
    val = np.empty((len(cols), K), dtype=np.float32)
    nit = np.zeros(len(cols), dtype=np.int32)
    for t, j in enumerate(cols.tolist()):
        it = C.c_int32(0)
        n = lib().slim_oracle_fit_column_sgd(U, I, d, i, p, int(j), alpha, l1_ratio, eta0, tol, max_iter,
                                             sklearn_seed(random_state), K, idx[t], val[t], C.byref(it))
        assert n == K
        nit[t] = it.value
        ptr[t + 1] = ptr[t] + K
    return ptr, idx.reshape(-1), val.reshape(-1), nit


def fit_columns(X_csc, cols, alpha=0.1, l1_ratio=0.1, tol=1e-4, max_iter=100, random_state=43,
                positive=True, nn_feature_selection=None, n_threads=1):
    """Per-column model.sparse_coef_ for each target column: (ptr, idx, val, n_iter).
    n_threads > 1: POSIX threads over the columns (identical results)."""
    U, I = X_csc.shape
    d, i, p = _csc(X_csc)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    K = int(nn_feature_selection) if nn_feature_selection is not None else 0
    cap = min(K, I) if K > 0 else I
    if n_threads > 1 and len(cols) > 0:
        cnt = np.zeros(len(cols), dtype=np.int32)
        idx2 = np.empty((len(cols), cap), dtype=np.int32)
        val2 = np.empty((len(cols), cap), dtype=np.float32)
        nit = np.zeros(len(cols), dtype=np.int32)
        lib().slim_oracle_fit_columns_mt(U, I, d, i, p, len(cols), cols, alpha, l1_ratio, tol, max_iter,
                                         sklearn_seed(random_state), int(bool(positive)), K, cap,
                                         cnt, idx2.reshape(-1), val2.reshape(-1), nit, int(n_threads))
        mask = np.arange(cap)[None, :] < cnt[:, None]
        ptr = np.zeros(len(cols) + 1, dtype=np.int64)
        np.cumsum(cnt, out=ptr[1:])
        return ptr, idx2[mask], val2[mask], nit
    ptr = np.zeros(len(cols) + 1, dtype=np.int64)
    idx = np.empty(max(len(cols) * cap, 1), dtype=np.int32)
    val = np.empty(max(len(cols) * cap, 1), dtype=np.float32)
    nit = np.zeros(max(len(cols), 1), dtype=np.int32)
    total = lib().slim_oracle_fit_columns(U, I, d, i, p, len(cols), cols, alpha, l1_ratio, tol, max_iter,
                                          sklearn_seed(random_state), int(bool(positive)), K,
                                          ptr, idx, val, nit)
    return ptr, idx[:total], val[:total], nit[:len(cols)]


def recommend_batch(Xb_csr, W_csr, top_k=10, filter_interacted=True, dense=False, use_f64=False, n_threads=1):
    """(ids[B,k], scores[B,k], counts[B]) following slim_elastic.py:674-741 + top-k helpers.
    n_threads > 1: POSIX threads over blocks of rows (identical results)."""
    B = Xb_csr.shape[0]
    n_cols = W_csr.shape[1]
    ids = np.empty((B, top_k), dtype=np.int32)
    sc = np.empty((B, top_k), dtype=np.float32)
    cnt = np.empty(max(B, 1), dtype=np.int32)
    fn = lib().slim_oracle_recommend_batch_mt if n_threads > 1 else lib().slim_oracle_recommend_batch
    extra = (int(n_threads),) if n_threads > 1 else ()
    fn(
        B, np.ascontiguousarray(Xb_csr.indptr, dtype=np.int32),
        np.ascontiguousarray(Xb_csr.indices, dtype=np.int32),
        np.ascontiguousarray(Xb_csr.data, dtype=np.float32),
        np.ascontiguousarray(W_csr.indptr, dtype=np.int32),
        np.ascontiguousarray(W_csr.indices, dtype=np.int32),
        np.ascontiguousarray(W_csr.data, dtype=np.float32),
        n_cols, top_k, int(bool(filter_interacted)), int(bool(dense)), int(bool(use_f64)),
        ids.reshape(-1) if B else np.empty(0, np.int32), sc.reshape(-1) if B else np.empty(0, np.float32), cnt, *extra)
    return ids, sc, cnt[:B]


def similar_items(W_csc, item: int, top_k=10):
    cap = max(int(W_csc.indptr[item + 1] - W_csc.indptr[item]), 1)
    k = min(top_k, cap)
    oi = np.empty(max(k, 1), dtype=np.int32)
    ov = np.empty(max(k, 1), dtype=np.float32)
    n = lib().slim_oracle_similar_items(np.ascontiguousarray(W_csc.indptr, dtype=np.int32),
                                        np.ascontiguousarray(W_csc.indices, dtype=np.int32),
                                        np.ascontiguousarray(W_csc.data, dtype=np.float32),
                                        int(item), int(k), oi, ov)
    return oi[:n], ov[:n]


# ---- CPU model of the device's binade-speculative ordered fold (oracle/fold_model.c) ---------------------------
class FoldStats(C.Structure):
    _fields_ = [("entries", C.c_int64), ("spec_entries", C.c_int64), ("serial_entries", C.c_int64), ("passes", C.c_int64)]


def _fold_lib():
    L = lib()
    if not getattr(L, "_fold_ready", False):
        L.fold_model_fold.restype = C.c_float
        L.fold_model_fold.argtypes = [C.c_float, _f32p, C.c_int64, C.POINTER(FoldStats)]
        L.fold_model_sequential.restype = C.c_float
        L.fold_model_sequential.argtypes = [C.c_float, _f32p, C.c_int64]
        L.fold_model_fuzz.restype = C.c_int64
        L.fold_model_fuzz.argtypes = [C.c_uint32, C.c_int64, C.c_int32, C.c_int32, C.POINTER(FoldStats)]
        L._fold_ready = True
    return L


def fold_speculative(products: np.ndarray, acc0: float = 0.0):
    """(sum, FoldStats) of the speculative fold model over float32 `products`, starting from acc0."""
    p = np.ascontiguousarray(products, dtype=np.float32)
    st = FoldStats()
    v = _fold_lib().fold_model_fold(np.float32(acc0), p, p.size, C.byref(st))
    return np.float32(v), st


def fold_sequential(products: np.ndarray, acc0: float = 0.0) -> np.float32:
    """The reference's left-to-right float32 accumulation (_cd_fast.pyx:464-466)."""
    p = np.ascontiguousarray(products, dtype=np.float32)
    return np.float32(_fold_lib().fold_model_sequential(np.float32(acc0), p, p.size))


def fold_fuzz(seed: int, n_folds: int, max_len: int, kind: int = -1, groups_per_window: int = 4):
    """(mismatches, FoldStats) over n_folds generated sums; kind -1 mixes all generators.  groups_per_window: the G of the
    device's fold_groups_spec<G> being modelled (1: single-wave kernel, 4: the multi-wave kernel's consumer)."""
    C.c_int.in_dll(_fold_lib(), "fold_model_groups_per_window").value = int(groups_per_window)
    st = FoldStats()
    bad = _fold_lib().fold_model_fuzz(seed, n_folds, max_len, kind, C.byref(st))
    return int(bad), st


def set_fold_model(on: bool) -> None:
    """Route the coordinate-descent dot products of THIS thread's oracle calls through the fold model (tests only)."""
    L = _fold_lib()
    L.slim_oracle_set_fold_model.restype = None
    L.slim_oracle_set_fold_model.argtypes = [C.c_int]
    L.slim_oracle_set_fold_model(1 if on else 0)


def fold_model_stats() -> FoldStats:
    L = _fold_lib()
    out = (C.c_int64 * 4)()
    L.slim_oracle_fold_model_stats.restype = None
    L.slim_oracle_fold_model_stats(out)
    return FoldStats(out[0], out[1], out[2], out[3])
