"""SLIM model: the drop-in for rtrec.models.SLIM (/root/reference/rtrec/models/slim.py:21-149).

Same constructor kwargs (fan out to the interaction store, both Identifiers and SLIMElastic;
unknown keys ignored), same fit / bulk_fit / recommend / recommend_batch / similar_items
behaviour, same pickle payload keys.  The interaction matrix is kept resident in HBM between
calls and scored in place by row id instead of being rebuilt and sliced per request.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Iterable, List, Optional, Tuple

import numpy as np

from ..utils.device_store import DeviceInteractions
from .base import BaseModel
from .internal.slim_elastic import SLIMElastic
from .. import settings

class SLIM(BaseModel):
    def __init__(self, **kwargs: Any):
        super().__init__(**kwargs)
        self.model = SLIMElastic(kwargs)
        self.recorded_item_ids: set = set()
        self._x_on_device: Optional[Tuple[int, float]] = None   # (store version, max_timestamp) of the GPU copy
        self._dev_x: Optional[DeviceInteractions] = None        # X resident in HBM (utils/device_store.py)

    # ------------------------------------------------------------ device-resident X
    def _store_tag(self) -> Any:
        """What a device copy of X must match to be current: the store version, and with time decay also
        max_timestamp (every value is a function of it)."""
        st = self.interactions
        return st.version if st.decay_rate is None else (st.version, st.max_timestamp)

    def _mirror(self, full: bool = False) -> Optional[DeviceInteractions]:
        """The device-resident copy of X, or None where it does not apply (a backend without device arrays).
        Stores with time decay keep their raw values and timestamps resident too and are re-valued on the device
        at every new max_timestamp (utils/device_store.py).  RTREC_AMD_DEVICE_STORE=0 forces the host-export path."""
        if settings.raw("RTREC_AMD_DEVICE_STORE", "1") == "0":
            return None
        be = self.model.engine.be
        if not getattr(be, "supports_device_store", False):
            return None
        if self._dev_x is None:
            self._dev_x = DeviceInteractions(be.torch, be.device)
            self._dev_x.decay_fn = getattr(be, "decay_f32", None)
        return self._dev_x

    def _mirror_synced(self, full: bool = False) -> Optional[DeviceInteractions]:
        """The mirror, brought up to the host store's state (one upload of the store's compacted block -- keys, raw
        values, timestamps -- if it has fallen behind; values are cast / decayed on the device)."""
        mir = self._mirror(full)
        if mir is not None and mir.version != self._store_tag():
            st = self.interactions
            blk = st._compact()
            mir.load_store(blk.key, blk.val, blk.ts if st.decay_rate is not None else None, st.shape[0], st.shape[1],
                           self._store_tag(), rate=st.decay_rate, now=st.max_timestamp)
        return mir

    _MIRROR_APPLY_MAX = 1 << 18     # larger writes (bulk chunks) leave the mirror stale: it is rebuilt on demand

    @property
    def bulk_chunk_rows(self) -> Optional[int]:
        """How many DataFrame rows Recommender hands over per add_interactions_columns call: with the device ingest one
        chunk is one upload + one sort, so the whole frame goes in one piece (64 M rows = 2 GiB of columns)."""
        return (1 << 26) if self._bulk_folder(1 << 26) is not None else None

    def _bulk_folder(self, n: int) -> Any:
        """Bulk batches are sorted, deduplicated and folded on the device (DeviceInteractions.ingest with the backend's
        rtrec_store_fold_device kernel); a batch into an EMPTY store also leaves the mirror in step -- no upload later."""
        from ..utils.interactions import _DEVICE_FOLD_MIN
        st = self.interactions
        if (n < _DEVICE_FOLD_MIN or st.decay_rate is not None or settings.raw("RTREC_AMD_DEVICE_INGEST", "1") == "0"):
            return None
        if self.model._engine is None:
            # ingest alone does not need the GPU (the host store is complete by itself): do not construct the engine --
            # which fails loudly without one -- just to look for a folder
            import torch
            if not torch.cuda.is_available():
                return None
        mir = self._mirror(True)
        fold_fn = None if mir is None else getattr(self.model.engine.be, "fold_pairs", None)
        if fold_fn is None:
            return None
        was_empty = st.is_empty

        def fold(users, items, ts, dl, upsert, lo, hi, lookup):
            res = mir.ingest(users, items, ts, dl, upsert, lo, hi, lookup, fold_fn)
            self._ingested_into_empty = was_empty
            return res
        return fold

    def _ingest(self, interactions: Iterable[Tuple[Any, Any, float, float]], update_interaction: bool
                ) -> Tuple[np.ndarray, np.ndarray]:
        tag0 = self._store_tag()
        uid, iid = super()._ingest(interactions, update_interaction)
        self._stored(tag0, uid, iid)
        return uid, iid

    def _stored(self, tag_before: Any, user_ids: np.ndarray, item_ids: np.ndarray) -> None:
        """A mirror that was in step with the store is advanced by the batch's distinct (user, item)
        pairs -- their new (raw) values and timestamps come from the host store, which owns the semantics."""
        st, mir = self.interactions, self._dev_x
        if mir is not None and mir.ingested is not None:       # the batch was folded on the device
            into_empty, self._ingested_into_empty = getattr(self, "_ingested_into_empty", False), False
            if self._store_tag() != tag_before and (into_empty or mir.version == tag_before):
                mir.adopt_ingested(st.shape[0], st.shape[1], self._store_tag(), merge=not into_empty)
                return
            mir.ingested = None
        if (mir is None or mir.version != tag_before or self._store_tag() == tag_before
                or len(user_ids) > self._MIRROR_APPLY_MAX or settings.raw("RTREC_AMD_DEVICE_STORE", "1") == "0"):
            return
        keys = np.unique(st._keys(user_ids, item_ids))
        _, val, ts = st._lookup(keys)
        if st.decay_rate is None:
            mir.apply(keys >> 32, keys & 0xFFFFFFFF, val.astype(np.float32), st.shape[0], st.shape[1], self._store_tag())
        else:
            mir.apply(keys >> 32, keys & 0xFFFFFFFF, val, st.shape[0], st.shape[1], self._store_tag(), tstamps=ts,
                      now=st.max_timestamp)

    def _device_matrix(self, item_ids: Optional[List[int]]) -> Optional[Dict[str, Any]]:
        """X (or X with only `item_ids`' columns populated) as device arrays, or None -> host export."""
        mir = self._mirror_synced(full=item_ids is None)
        if mir is None:
            return None
        X = dict(mir.full() if item_ids is None else mir.partial(np.asarray(item_ids, dtype=np.int64)))
        X["n_users"], X["n_items"] = mir.n_users, mir.n_items
        return X

    # ------------------------------------------------------------ fit
    def fit(self, interactions: Iterable[Tuple[Any, Any, float, float]], update_interaction: bool = False,
            progress_bar: bool = True) -> "SLIM":
        """Ingest a mini-batch and refit exactly the items it touched (slim.py:29-43): the matrix
        handed to the solver has ONLY those items' columns populated."""
        _, iid = self._ingest(interactions, update_interaction)
        item_ids = np.unique(iid).tolist()
        self._fit_items(item_ids, False, progress_bar)
        return self

    def _fit_items(self, item_ids: List[int], parallel: bool, progress_bar: bool) -> None:
        X = self._device_matrix(item_ids) if item_ids else None
        if X is not None:
            self.model.partial_fit_items_device(X, item_ids)
        else:
            interaction_matrix = self.interactions.to_csc(item_ids)
            self.model.partial_fit_items(interaction_matrix, item_ids, parallel=parallel, progress_bar=progress_bar)
        self._x_on_device = None

    def _record_interactions(self, user_id: int, item_id: int, tstamp: float, rating: float) -> None:
        self.recorded_item_ids.add(item_id)

    def _record_batch(self, user_ids: np.ndarray, item_ids: np.ndarray) -> None:
        self.recorded_item_ids.update(np.unique(item_ids).tolist())

    def _fit_recorded(self, parallel: bool = False, progress_bar: bool = True) -> "SLIM":
        item_ids = sorted(self.recorded_item_ids)
        self._fit_items(item_ids, parallel, progress_bar)
        self.recorded_item_ids.clear()
        return self

    def bulk_fit(self, parallel: bool = False, progress_bar: bool = True) -> "SLIM":
        X = self._device_matrix(None)
        if X is not None:        # resident X: CSR order from the host keys, CSC order by a sort on the device
            self.model.fit_device(X, parallel=parallel)
        else:
            self.model.fit(self.interactions.to_csc(), parallel=parallel, progress_bar=progress_bar)
        self._x_on_device = None
        return self

    # ------------------------------------------------------------ recommend
    def _sync_interactions(self) -> None:
        """Make the GPU copy of X (CSR, decayed to the current max_timestamp) current."""
        stamp = (self.interactions.version, self.interactions.max_timestamp)
        if self._x_on_device != stamp:
            mir = self._mirror(full=True)
            if mir is not None and mir.version == self._store_tag():
                self.model.engine.set_interactions_device(mir.full(), mir.n_users, mir.n_items)
            else:
                self.model.engine.set_interactions(None, self.interactions.to_csr(), need_csc=False)
            self._x_on_device = stamp

    def _recommend(self, user_id: int, candidate_item_ids: Optional[List[int]] = None,
                   user_tags: Optional[List[str]] = None, top_k: int = 10, filter_interacted: bool = True) -> List[int]:
        return self._recommend_hot_batch([user_id], candidate_item_ids=candidate_item_ids, top_k=top_k,
                                         filter_interacted=filter_interacted)[0]

    def _recommend_hot_batch(self, user_ids: List[int], candidate_item_ids: Optional[List[int]] = None,
                             users_tags: Optional[List[List[str]]] = None, top_k: int = 10,
                             filter_interacted: bool = True) -> List[List[int]]:
        if not self.model.is_fitted:
            raise RuntimeError("Model must be fitted before calling batch_recommend.")
        if len(user_ids) == 0:
            return []
        n_users = self.interactions.shape[0]
        ids_arr = np.asarray(user_ids, dtype=np.int64)
        if int(ids_arr.min()) < 0 or int(ids_arr.max()) >= n_users:
            return self._recommend_odd_ids(ids_arr, n_users, candidate_item_ids, top_k, filter_interacted)
        ids, scores, counts = self._hot_topk(ids_arr, candidate_item_ids, top_k, filter_interacted)
        return self.model._format(ids, scores, counts, ret_scores=False)

    def _recommend_hot_arrays(self, user_ids: np.ndarray, candidate_item_ids: Optional[List[int]], top_k: int,
                              filter_interacted: bool) -> Tuple[np.ndarray, np.ndarray]:
        """The kernels' own output -- ids[B, k] and counts[B] as they come off the device in one copy -- with no Python
        object per user or per item (BaseModel.recommend_batch's vectorised route)."""
        if not self.model.is_fitted:
            raise RuntimeError("Model must be fitted before calling batch_recommend.")
        if len(user_ids) == 0:
            return np.empty((0, top_k), np.int32), np.zeros(0, np.int32)
        n_users = self.interactions.shape[0]
        if int(user_ids.min()) < 0 or int(user_ids.max()) >= n_users:
            return super()._recommend_hot_arrays(user_ids, candidate_item_ids, top_k, filter_interacted)
        ids, _, counts = self._hot_topk(user_ids, candidate_item_ids, top_k, filter_interacted)
        return ids, counts

    def _hot_topk(self, ids_arr: np.ndarray, candidate_item_ids: Optional[List[int]], top_k: int, filter_interacted: bool):
        """(ids, scores, counts) arrays for internal user ids inside [0, n_users)."""
        dense_output = not self.item_ids.pass_through
        stamp = (self.interactions.version, self.interactions.max_timestamp)
        n_users = self.interactions.shape[0]
        resident = self._dev_x is not None and self._dev_x.version == self._store_tag()
        if self._x_on_device == stamp or resident or len(ids_arr) * 16 >= n_users:
            # bulk scoring: (re)upload all of X once and score it in place by row id
            self._sync_interactions()
            return self.model._topk(None, candidate_item_ids, top_k, filter_interacted, dense_output, row_ids=ids_arr)
        # online serving after an update: ship only the requested users' rows (like the
        # reference's to_csr(select_users=...), but without the empty rows)
        uniq, inverse = np.unique(ids_arr, return_inverse=True)
        rows, cols, data = self.interactions._triples(select_users=uniq)
        indptr = np.zeros(len(uniq) + 1, dtype=np.int64)
        indptr[1:] = np.bincount(np.searchsorted(uniq, rows), minlength=len(uniq))
        np.cumsum(indptr, out=indptr)
        from scipy.sparse import csr_matrix
        Xb = csr_matrix((data.astype(np.float32), cols.astype(np.int32), indptr.astype(np.int32)),
                        shape=(len(uniq), self.interactions.shape[1]))
        ids, scores, counts = self.model._topk(Xb, candidate_item_ids, top_k, filter_interacted, dense_output)
        return ids[inverse], scores[inverse], counts[inverse]

    def _recommend_odd_ids(self, ids: np.ndarray, n_users: int, candidate_item_ids: Optional[List[int]], top_k: int,
                           filter_interacted: bool) -> List[List[int]]:
        """Internal user ids outside [0, n_users) -- a negative integer user, or a user known only through
        register_user_feature -- never reach the device.  They get what the reference's scipy row indexing
        of to_csr(select_users=...) gives them (slim.py:93, slim_elastic.py:707): IndexError beyond the
        matrix, and for a negative id the row it wraps around to, which is populated only if that user is
        in the same batch."""
        bad = ids[(ids >= n_users) | (ids < -n_users)]
        if len(bad):
            raise IndexError(f"index ({int(bad[0])}) out of range")
        regular = set(ids[ids >= 0].tolist())
        rows = np.where(ids < 0, ids + n_users, ids)
        live = np.array([(i >= 0) or (int(w) in regular) for i, w in zip(ids.tolist(), rows.tolist())], dtype=bool)
        out: List[List[int]] = [[] for _ in ids]
        if live.any():
            got = self._recommend_hot_batch(rows[live].tolist(), candidate_item_ids=candidate_item_ids, top_k=top_k,
                                            filter_interacted=filter_interacted)
            for p, row in zip(np.flatnonzero(live).tolist(), got):
                out[p] = row
        if candidate_item_ids is not None and not live.all():
            # an all-zero row still ranks the candidates (scores 0): argsort(...)[-k:][::-1], slim_elastic.py:730-733
            k = min(top_k, len(candidate_item_ids))
            zero_row = [candidate_item_ids[j] for j in range(len(candidate_item_ids) - 1, len(candidate_item_ids) - 1 - k, -1)]
            for p in np.flatnonzero(~live).tolist():
                out[p] = list(zero_row)
        elif not self.item_ids.pass_through and not live.all():
            # dense mode ranks every item of an all-zero row too: highest ids first (stable-argsort rule, D1)
            n_items = self.model.n_items_fitted
            k = min(top_k, n_items)
            for p in np.flatnonzero(~live).tolist():
                out[p] = list(range(n_items - 1, n_items - 1 - k, -1))
        return out

    def _similar_items(self, query_item_id: int, query_item_tags: Optional[List[str]] = None, top_k: int = 10
                       ) -> List[Tuple[int, float]]:
        return self.model.similar_items(query_item_id, top_k=top_k, ret_ndarrays=False)  # type: ignore

    def _similar_items_batch(self, query_item_ids: List[int], query_item_tags: Optional[List[str]] = None,
                             top_k: int = 10) -> List[List[Tuple[int, float]]]:
        return self.model.similar_items_batch(query_item_ids, top_k=top_k)

    # ------------------------------------------------------------ persistence (slim.py:117-149)
    def _serialize(self) -> dict:
        return {"model": self.model, "interactions": self.interactions, "user_ids": self.user_ids,
                "item_ids": self.item_ids, "feature_store": self.feature_store}

    @classmethod
    def _deserialize(cls, data: dict) -> "SLIM":
        instance = cls()
        instance.model = data["model"]
        instance.interactions = data["interactions"]
        instance.user_ids = data["user_ids"]
        instance.item_ids = data["item_ids"]
        instance.feature_store = data["feature_store"]
        return instance
