"""SLIMElastic: the operator boundary of the SLIM path, served by the MI355X engine.

Mirrors the method set of rtrec.models.internal.slim_elastic.SLIMElastic
(/root/reference/rtrec/models/internal/slim_elastic.py:156-857) -- same constructor config, same
arguments, same return types, same error messages -- so callers written against the reference
(rtrec.models.SLIM, HybridSlimFM) switch by import.  Where the reference loops over item columns
calling scikit-learn's ElasticNet and over users calling scipy/numpy/sorted(), this class uploads
the matrices once and calls the HIP kernels (rtrec_amd/csrc) through rtrec_amd.engine.

W lives on the device between fit and score (engine.DeviceWeights: the fit kernels' output is merged into it and
the score layouts are built from it there).  `item_similarity` is still the reference's public attribute -- a host
scipy.sparse.csc_matrix, pickled, read by HybridSlimFM -- but it is materialised from the device copy only when
somebody reads it; assigning to it replaces the device copy.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp
from numpy import ndarray

from ... import _native
from ...engine import DeviceWeights, SlimEngine, coefficients_to_updates, merge_coefficients


def _default_engine() -> SlimEngine:
    """One engine per process: cuda:LOCAL_RANK, sharded over the default process group if any."""
    import os
    import torch
    rank, world, group = 0, 1, None
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
    except Exception:   # pragma: no cover
        pass
    dev = None
    if torch.cuda.is_available():
        dev = f"cuda:{int(os.environ.get('LOCAL_RANK', torch.cuda.current_device()))}"
    return SlimEngine(device=dev, rank=rank, world_size=world, process_group=group)


class SLIMElastic:
    """Sparse linear method: one elastic-net regression per item column, W = item_similarity."""

    def __init__(self, config: dict = {}, engine: Optional[SlimEngine] = None):
        self.optim_name = config.get("optim", "cd")
        self.eta0 = config.get("eta0", 0.001)
        self.alpha = config.get("alpha", 0.1)
        self.l1_ratio = config.get("l1_ratio", 0.1)
        self.positive_only = config.get("positive_only", True)
        self.max_iter = config.get("max_iter", 100)
        self.tol = config.get("tol", 1e-4)
        self.random_state = config.get("random_state", 43)
        self.nn_feature_selection = config.get("nn_feature_selection", None)
        # extension (not a reference kwarg): fit_mode "exact" (default: coefficients bit-identical to scikit-learn),
        # "gram" (Gram-form coordinate descent: a few 1e-5 relative, several times faster on dense catalogues; also
        # selected by exact=False) or "shuffle" (tree-reduced dot products on the float32 residual) -- DESIGN.md 3.4
        self.fit_mode = config.get("fit_mode", "exact" if config.get("exact", True) else "gram")
        self._item_similarity: Optional[sp.csc_matrix] = None    # host copy of W (None while only the device holds it)
        self._w_dev: Optional[DeviceWeights] = None              # device copy of W (None until first needed)
        self._engine = engine
        self.n_iter_: Optional[np.ndarray] = None    # sweeps per fitted column of the last fit

    # ---------------------------------------------------------------- state
    @property
    def item_similarity(self) -> Optional[sp.csc_matrix]:
        if self._item_similarity is None and self._w_dev is not None:
            # one download, on demand (a column-sharded W -- SlimEngine.shard_w -- is gathered first: a collective call)
            self._item_similarity = self.engine.gather_weights(self._w_dev).to_csc(self.engine.be.torch)
        return self._item_similarity

    def gather_item_similarity(self) -> Optional[sp.csc_matrix]:
        """The host copy of W, explicitly.  With a column-sharded W (SlimEngine.shard_w, several ranks) this is a COLLECTIVE:
        every rank must call it at the same point -- reading `item_similarity`, `save()` and pickling go through it, so a
        model whose W is sharded is saved by calling this on ALL ranks first and pickling on one of them afterwards (the
        host copy is then cached and no further exchange happens).  A call on one rank only blocks in all_gather."""
        return self.item_similarity

    @item_similarity.setter
    def item_similarity(self, W: Optional[sp.csc_matrix]) -> None:
        self._item_similarity = W
        self._w_dev = None

    @property
    def is_fitted(self) -> bool:
        return self._item_similarity is not None or self._w_dev is not None

    @property
    def n_items_fitted(self) -> int:
        """Number of item columns of W (item_similarity.shape[1]) without materialising the host matrix."""
        return self._w_dev.n_items if self._w_dev is not None else self._item_similarity.shape[1]

    def _w_is_f64(self) -> bool:
        return self._w_dev.f64 if self._w_dev is not None else self._item_similarity.dtype == np.float64

    @property
    def engine(self) -> SlimEngine:
        if self._engine is None:
            self._engine = _default_engine()
        return self._engine

    def __getstate__(self) -> Dict[str, Any]:
        state = dict(self.__dict__)
        state["_item_similarity"] = self.item_similarity      # the host matrix is what is pickled
        state["_engine"] = None
        state["_w_dev"] = None
        return state

    def __setstate__(self, state: Dict[str, Any]) -> None:
        # the reference keeps `item_similarity` as a plain attribute (slim_elastic.py:193)
        if "item_similarity" in state:
            state = dict(state)
            state["_item_similarity"] = state.pop("item_similarity")
        self.__dict__.update(state)
        self.__dict__.setdefault("_item_similarity", None)
        self._engine = None
        self._w_dev = None
        self.__dict__.pop("_w_on_device", None)
        self.__dict__.setdefault("n_iter_", None)

    def _check_optim(self) -> None:
        """slim_elastic.py:195-227: "cd" -> ElasticNet, "sgd" -> SGDRegressor, anything else raises."""
        if self.optim_name not in ("cd", "sgd"):
            raise ValueError(f"Invalid Optimizer name: {self.optim_name}")

    # ---------------------------------------------------------------- fit
    def _fit_targets(self, X_csc: sp.csc_matrix, targets: np.ndarray, keep_old: bool, f64: bool) -> None:
        """Fit `targets` on the GPU(s) and write the coefficients back like the reference's LIL loop: into the existing
        W (`keep_old`) or into an empty one.  With feature selection the whole write-back happens on the device
        (SlimEngine.merge_fit) and the host matrix is only built when `item_similarity` is read."""
        self._check_optim()
        eng = self.engine
        if isinstance(X_csc, dict):       # already resident on the device (utils/device_store.py)
            n_items = int(X_csc["n_items"])
            eng.set_interactions_device(X_csc, X_csc["n_users"], n_items)
        else:
            n_items = X_csc.shape[1]
            eng.set_interactions(X_csc)
        targets = np.unique(np.asarray(targets, dtype=np.int64))     # a column listed twice is fitted once
        mine = eng.owned_columns(targets)
        kw = dict(alpha=self.alpha, l1_ratio=self.l1_ratio, positive=self.positive_only, max_iter=self.max_iter,
                  tol=self.tol, random_state=self.random_state, nn_feature_selection=self.nn_feature_selection,
                  **({} if getattr(self, "fit_mode", "exact") == "exact" else {"mode": self.fit_mode}))
        old_dev: Optional[DeviceWeights] = None
        if keep_old and self.is_fitted:
            if self._w_dev is None:
                W = self._item_similarity
                self._w_dev = eng.upload_weights(W if isinstance(W, sp.csc_matrix) else sp.csc_matrix(W))
            old_dev = self._w_dev
        if self.optim_name == "sgd":
            # scikit-learn's SGDRegressor behind FeatureSelectionWrapper (slim_elastic.py:139-154, 209-222).  Without
            # nn_feature_selection the reference fails on the first column it fits (SGDRegressor has no sparse_coef_, :273):
            # so does this -- unless there is nothing to fit.
            if self.nn_feature_selection is None and len(targets) == 0:
                kw.pop("mode", None)
            else:
                if old_dev is not None and old_dev.lossy:
                    raise NotImplementedError("optim='sgd' merges into a W whose values are float32 numbers")
                d_t, d_items, d_coef, d_count, n_iter = eng.fit_columns_sgd(
                    mine, alpha=self.alpha, l1_ratio=self.l1_ratio, eta0=self.eta0, max_iter=self.max_iter, tol=self.tol,
                    random_state=self.random_state, nn_feature_selection=self.nn_feature_selection)
                self.n_iter_ = n_iter
                self._w_dev = eng.merge_fit(old_dev, n_items, f64, d_t, d_items, d_coef, d_count)
                self._item_similarity = None
                eng.set_weights(self._w_dev)
                return
        if self.nn_feature_selection is not None and (old_dev is None or not old_dev.lossy):
            d_t, d_items, d_coef, d_count, n_iter = eng.fit_columns(mine, device_out=True, **kw)
            self.n_iter_ = n_iter
            self._w_dev = eng.merge_fit(old_dev, n_items, f64, d_t, d_items, d_coef, d_count)
            self._item_similarity = None
            eng.set_weights(self._w_dev)
            return
        # all-features fits (output block sized for the host) and float64 matrices that float32 cannot hold
        tg, items, coef, count, n_iter = eng.fit_columns(mine, **kw)
        rows, cols, vals = coefficients_to_updates(tg, items, coef, count)
        self.n_iter_ = n_iter
        if eng.world_size > 1:
            import torch.distributed as dist
            parts: List[Any] = [None] * eng.world_size
            dist.all_gather_object(parts, (rows, cols, vals), group=eng.group)
            rows = np.concatenate([p[0] for p in parts])
            cols = np.concatenate([p[1] for p in parts])
            vals = np.concatenate([p[2] for p in parts])
        W_old = self.item_similarity if keep_old else None
        if W_old is not None and W_old.shape[0] != n_items:
            W_old = W_old.copy()
            W_old.resize((n_items, n_items))
        self.item_similarity = merge_coefficients(W_old, n_items, rows, cols, vals, dtype=np.float64 if f64 else np.float32)

    def _merge_dtype_f64(self) -> bool:
        """dtype of a merge into the existing matrix: float32 when there is none yet (slim_elastic.py:322-327)."""
        return self._w_is_f64() if self.is_fitted else False

    @staticmethod
    def _as_csc(interaction_matrix: Any, err: str) -> sp.csc_matrix:
        if isinstance(interaction_matrix, sp.csc_matrix):
            return interaction_matrix
        if isinstance(interaction_matrix, sp.csr_matrix):
            return interaction_matrix.tocsc()
        raise ValueError(err)

    def fit(self, interaction_matrix: sp.csc_matrix | sp.csr_matrix, parallel: bool = False,
            progress_bar: bool = False) -> "SLIMElastic":
        """Fit every item column.  Serial mode starts from an empty float64 matrix, `parallel`
        merges into the existing float32 one (slim_elastic.py:252 vs :322-327)."""
        if isinstance(interaction_matrix, sp.csc_matrix) and parallel:
            return self.fit_in_parallel(interaction_matrix, progress_bar=progress_bar)
        if isinstance(interaction_matrix, sp.csr_matrix) and parallel:
            logging.warning("Multiprocessing is only supported for CSC format. Fitting in single process.")
        X = self._as_csc(interaction_matrix,
                         "Interaction matrix must be a scipy.sparse.csr_matrix or scipy.sparse.csc_matrix.")
        self._fit_targets(X, np.arange(X.shape[1]), False, True)
        return self

    def fit_in_parallel(self, interaction_matrix: sp.csc_matrix, item_ids: Optional[ndarray] = None,
                        progress_bar: bool = False, chunk_size: int = 100, num_workers: Optional[int] = None
                        ) -> "SLIMElastic":
        """Reference: process pool over item chunks (slim_elastic.py:283-386).  Here the item
        columns are the GPU work queue; chunk_size / num_workers are accepted and ignored."""
        if not isinstance(interaction_matrix, sp.csc_matrix):
            raise ValueError("Interaction matrix must be in CSC format for parallel processing.")
        n_items = interaction_matrix.shape[1]
        targets = np.arange(n_items) if item_ids is None else np.asarray(item_ids, dtype=np.int64)
        self._fit_targets(interaction_matrix, targets, True, self._merge_dtype_f64())
        return self

    def partial_fit(self, interaction_matrix: sp.csr_matrix, user_ids: List[int], parallel: bool = False,
                    progress_bar: bool = False) -> "SLIMElastic":
        """Refit the items the given users interacted with (slim_elastic.py:495-508)."""
        X = interaction_matrix.tocsr()
        items = np.unique(np.concatenate([X.indices[X.indptr[u]:X.indptr[u + 1]] for u in user_ids])
                          if len(user_ids) else np.empty(0, np.int64))
        return self.partial_fit_items(interaction_matrix, items.tolist(), progress_bar=progress_bar)

    def partial_fit_items(self, interaction_matrix: sp.csc_matrix | sp.csr_matrix, updated_items: List[int],
                          parallel: bool = False, progress_bar: bool = False) -> "SLIMElastic":
        """Refit only `updated_items`, keeping every other column of W (slim_elastic.py:510-564)."""
        if isinstance(interaction_matrix, sp.csc_matrix) and parallel:
            return self.fit_in_parallel(interaction_matrix, item_ids=np.array(updated_items), progress_bar=progress_bar)
        X = self._as_csc(interaction_matrix,
                         "Interaction matrix must be a scipy.sparse.csr_matrix or scipy.sparse.csc_matrix.")
        self._fit_targets(X, np.asarray(list(updated_items), dtype=np.int64), True, self._merge_dtype_f64())
        return self

    def fit_device(self, X: Dict[str, Any], parallel: bool = False) -> "SLIMElastic":
        """fit() for a matrix that is already resident on the device (DeviceInteractions.full() plus
        n_users / n_items): serial mode starts from an empty float64 W, `parallel` merges into the
        existing float32 one, exactly like fit / fit_in_parallel."""
        targets = np.arange(int(X["n_items"]))
        if parallel:
            self._fit_targets(X, targets, True, self._merge_dtype_f64())
        else:
            self._fit_targets(X, targets, False, True)
        return self

    def partial_fit_items_device(self, X: Dict[str, Any], updated_items: List[int]) -> "SLIMElastic":
        """partial_fit_items for a matrix that is already resident on the device: `X` is the array set of
        DeviceInteractions.partial() plus n_users / n_items."""
        self._fit_targets(X, np.asarray(list(updated_items), dtype=np.int64), True, self._merge_dtype_f64())
        return self

    # ---------------------------------------------------------------- score
    def _sync_weights(self) -> None:
        """Make the engine score with this model's W (a no-op while it already does)."""
        if self._w_dev is None:
            W = self._item_similarity
            Wc = W if isinstance(W, sp.csc_matrix) else sp.csc_matrix(W)
            self._w_dev = self.engine.upload_weights(Wc, acc_f64=(Wc.dtype == np.float64))
        if self.engine.weights is not self._w_dev:
            self.engine.set_weights(self._w_dev)

    def _topk(self, Xb: sp.csr_matrix, candidate_item_ids: Optional[List[int]], top_k: int, filter_interacted: bool,
              dense_output: bool, row_ids: Optional[Sequence[int]] = None):
        """(ids[B,k], scores[B,k], counts[B]) for the rows of Xb (or rows `row_ids` of the resident X)."""
        self._sync_weights()
        n_items = self.n_items_fitted
        col_rank = cands = None
        if candidate_item_ids is not None:
            mode = _native.TOPK_CANDIDATES
            cands = np.asarray(candidate_item_ids, dtype=np.int64)        # (the engine builds the rank array when it needs one)
            top_k = min(top_k, len(candidate_item_ids))
        else:
            mode = _native.TOPK_DENSE if dense_output else _native.TOPK_SPARSE
            top_k = min(top_k, n_items)
        if top_k <= 0:
            B = Xb.shape[0] if row_ids is None else len(row_ids)
            return np.empty((B, 0), np.int32), np.empty((B, 0), np.float32), np.zeros(B, np.int32)
        if not self.engine.topk_supported(top_k, mode):
            # beyond what the fused kernel selects on the device (top_k > 1023, or a catalogue whose per-tile
            # lists would not fit the merge): scores from the device (score_rows_kernel), selection on the host
            if row_ids is not None:
                Xb = self.engine.rows_csr(row_ids)
            return self._topk_host(Xb, candidate_item_ids, top_k, filter_interacted, mode)
        if row_ids is not None:
            return self.engine.recommend_rows(row_ids, top_k, filter_interacted, mode, col_rank, candidates=cands)
        if Xb.shape[1] != n_items:   # the reference would fail inside scipy on a shape mismatch
            Xb = Xb.copy()
            Xb.resize((Xb.shape[0], n_items))
        return self.engine.recommend_csr(Xb, top_k, filter_interacted, mode, col_rank, candidates=cands)

    def _topk_host(self, Xb: sp.csr_matrix, candidate_item_ids: Optional[List[int]], top_k: int,
                   filter_interacted: bool, mode: int, chunk_rows: int = 256):
        """Large-k path: the score rows come from the device (SlimEngine.predict_csr), the selection follows
        slim_elastic.py:661-672 (candidates), :744-779 (dense) and :782-818 (sparse: stored non-zero products
        only, stable sort over scipy's reverse-first-touch product order) in numpy.  Unspecified tie orders
        (numpy's unstable argsort) follow the same canonical rule as the kernels."""
        W = self.item_similarity
        n_items = W.shape[1]
        Xb = Xb.tocsr()
        if Xb.shape[1] != n_items:
            Xb = Xb.copy()
            Xb.resize((Xb.shape[0], n_items))
        if not Xb.has_sorted_indices:
            Xb = Xb.sorted_indices()
        B = Xb.shape[0]
        dt = np.float64 if W.dtype == np.float64 else np.float32
        ids = np.full((B, top_k), -1, dtype=np.int32)
        scores = np.full((B, top_k), -np.inf, dtype=np.float32)
        counts = np.zeros(B, dtype=np.int32)
        Wr = W.tocsr() if mode == _native.TOPK_SPARSE else None
        cand = np.asarray(candidate_item_ids, dtype=np.int64) if candidate_item_ids is not None else None
        for r0 in range(0, B, chunk_rows):
            S = np.asarray(self.engine.predict_csr(Xb[r0:r0 + chunk_rows]), dtype=dt)
            for q in range(S.shape[0]):
                r = r0 + q
                own = Xb.indices[Xb.indptr[r]:Xb.indptr[r + 1]]
                if cand is not None:
                    sc = S[q, cand]
                    top = np.argsort(sc, kind="stable")[-top_k:][::-1]
                    sel, val = cand[top], sc[top]
                elif mode == _native.TOPK_DENSE:
                    sc = S[q].copy()
                    if filter_interacted:
                        sc[own] = -np.inf
                    top = np.argsort(sc, kind="stable")[-top_k:][::-1]
                    top = top[sc[top] != -np.inf]
                    sel, val = top, sc[top]
                else:
                    sc = S[q]
                    nz = sc != 0
                    if filter_interacted:
                        nz[own] = False
                    cols = np.flatnonzero(nz)
                    # first-touch position of every product column: W rows of the user's items in ascending item order
                    seq = np.concatenate([Wr.indices[Wr.indptr[i]:Wr.indptr[i + 1]] for i in own.tolist()]
                                         or [np.empty(0, np.int32)])
                    ft = np.full(n_items, -1, dtype=np.int64)
                    u, first = np.unique(seq, return_index=True)
                    ft[u] = first
                    order = np.lexsort((-ft[cols], -sc[cols]))[:top_k]
                    sel, val = cols[order], sc[cols][order]
                c = len(sel)
                ids[r, :c], scores[r, :c], counts[r] = sel, val, c
        return ids, scores, counts

    def recommend(self, user_id: int, interaction_matrix: sp.csr_matrix,
                  candidate_item_ids: Optional[List[int]] = None, top_k: int = 10, filter_interacted: bool = True,
                  dense_output: bool = True, ret_scores: bool = False) -> List[int] | Tuple[List[int], ndarray]:
        if not self.is_fitted:
            raise RuntimeError("Model must be fitted before calling predict.")
        out = self.recommend_batch([user_id], interaction_matrix, candidate_item_ids, top_k, filter_interacted,
                                   dense_output, ret_scores)
        return out[0]

    def recommend_batch(self, user_ids: List[int], interaction_matrix: sp.csr_matrix,
                        candidate_item_ids: Optional[List[int]] = None, top_k: int = 10,
                        filter_interacted: bool = True, dense_output: bool = True, ret_scores: bool = False
                        ) -> List[List[int]] | List[Tuple[List[int], ndarray]]:
        if not self.is_fitted:
            raise RuntimeError("Model must be fitted before calling batch_recommend.")
        if len(user_ids) == 0:
            return []
        Xb = interaction_matrix[user_ids, :]
        ids, scores, counts = self._topk(Xb, candidate_item_ids, top_k, filter_interacted, dense_output)
        return self._format(ids, scores, counts, ret_scores)

    @staticmethod
    def _format(ids: ndarray, scores: ndarray, counts: ndarray, ret_scores: bool):
        rows = ids.tolist()                                # one conversion for the whole batch
        if not ret_scores and (len(rows) == 0 or int(counts.min()) == ids.shape[1]):
            return rows                                    # every list is full (the usual case): nothing to cut
        cnt = counts.tolist()
        if not ret_scores:
            for p in np.flatnonzero(counts < ids.shape[1]).tolist():      # cut only the short rows (no B new lists)
                rows[p] = rows[p][:cnt[p]]
            return rows
        return [(row[:c], scores[r, :c].copy()) for r, (row, c) in enumerate(zip(rows, cnt))]

    # ---------------------------------------------------------------- predict (dense / sparse score rows)
    def _not_fitted(self, what: str) -> None:
        if not self.is_fitted:
            raise RuntimeError(f"Model must be fitted before calling {what}.")

    def _predict_rows(self, Xb: sp.csr_matrix, dense_output: bool):
        """Xb . W as the reference's safe_sparse_dot returns it: ndarray if dense_output, else a CSR
        (canonical: sorted indices, exact zeros dropped -- scipy's unsorted product order is not kept)."""
        self._sync_weights()
        n_items = self.n_items_fitted
        Xb = Xb.tocsr()
        if Xb.shape[1] != n_items:
            Xb = Xb.copy()
            Xb.resize((Xb.shape[0], n_items))
        S = self.engine.predict_csr(Xb)
        return S if dense_output else sp.csr_matrix(S)

    def predict(self, user_id: int, interaction_matrix: sp.csr_matrix, dense_output: bool = True):
        """Scores of one user for all items: shape (1, n_items) (slim_elastic.py:566-585)."""
        self._not_fitted("predict")
        return self._predict_rows(interaction_matrix[user_id, :], dense_output)

    def predict_selected(self, user_id: int, item_ids: List[int], interaction_matrix: sp.csr_matrix,
                         dense_output: bool = True):
        """Scores of one user for the given items: shape (1, len(item_ids)) (slim_elastic.py:587-608)."""
        self._not_fitted("predict_selected")
        S = self._predict_rows(interaction_matrix[user_id, :], True)[:, list(item_ids)]
        return S if dense_output else sp.csr_matrix(S)

    def predict_all(self, interaction_matrix: sp.csr_matrix, dense_output: bool = True):
        """Scores of every row of the matrix: shape (n_users, n_items) (slim_elastic.py:610-626)."""
        self._not_fitted("predict_all")
        return self._predict_rows(interaction_matrix, dense_output)

    # ---------------------------------------------------------------- item-to-item
    def similar_items(self, item_id: int, top_k: int = 10, ret_ndarrays: bool = False
                      ) -> List[Tuple[int, float]] | Tuple[ndarray, ndarray]:
        if not self.is_fitted:
            raise RuntimeError("Model must be fitted before calling similar_items.")
        n_items = self.n_items_fitted
        if not 0 <= item_id < n_items or top_k <= 0:
            ids, sc = np.empty(0, np.int32), np.empty(0, np.float32)
        else:
            self._sync_weights()
            i, s, c = self.engine.similar_items([item_id], top_k)
            ids, sc = i[0, :int(c[0])], s[0, :int(c[0])]
        if ret_ndarrays:
            return ids, sc
        return list(zip(ids.tolist(), sc.tolist()))

    def similar_items_batch(self, item_ids: List[int], top_k: int = 10) -> List[List[Tuple[int, float]]]:
        """[similar_items(i, top_k) for i in item_ids] with ONE launch of similar_topk_kernel and one
        download (the reference answers item-to-item queries one column at a time, slim_elastic.py:820-857)."""
        if not self.is_fitted:
            raise RuntimeError("Model must be fitted before calling similar_items.")
        n_items = self.n_items_fitted
        out: List[List[Tuple[int, float]]] = [[] for _ in item_ids]
        valid = [p for p, i in enumerate(item_ids) if 0 <= i < n_items]
        if not valid or top_k <= 0:
            return out
        self._sync_weights()
        ids, sc, cnt = self.engine.similar_items([item_ids[p] for p in valid], top_k)
        id_rows, sc_rows, cnts = ids.tolist(), sc.tolist(), cnt.tolist()
        for r, p in enumerate(valid):
            c = cnts[r]
            out[p] = list(zip(id_rows[r][:c], sc_rows[r][:c]))
        return out
