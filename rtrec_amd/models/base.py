"""Model boundary: raw ids in, raw ids out.

Public surface of rtrec.models.base.BaseModel (/root/reference/rtrec/models/base.py:21-423):
feature registration (:36-70), add_interactions (:72-94), fit (:107-115), recommend (:135-173),
recommend_batch (:188-269), similar_items (:320-340), get_users_by_items (:342-362),
save/load/loads (:376-405) and the abstract hooks a concrete model fills in.  Batches are
ingested through the columnar store in one vectorised call when the batch is well formed; a
malformed interaction falls back to the reference's per-tuple "log a warning and skip it".
"""
from __future__ import annotations

import logging
import pickle
from abc import ABC, abstractmethod
from io import BytesIO
from typing import Any, Iterable, List, Optional, Tuple, Union

import numpy as np

from ..utils.features import FeatureStore
from ..utils.identifiers import Identifier
from ..utils.interactions import UserItemInteractions

FileLike = Union[BytesIO, Any]


class BaseModel(ABC):
    def __init__(self, **kwargs: Any):
        self.interactions = UserItemInteractions(**kwargs)
        self.user_ids = Identifier(**kwargs)
        self.item_ids = Identifier(**kwargs)
        self.feature_store = FeatureStore()

    # ------------------------------------------------------------ features
    def register_user_feature(self, user: Any, user_tags: List[str]) -> int:
        user_id = self.user_ids.identify(user)
        self.feature_store.put_user_features(user_id, user_tags)
        return user_id

    def clear_user_features(self, user_ids: Optional[List[int]] = None) -> None:
        self.feature_store.clear_user_features(user_ids)

    def register_item_feature(self, item: Any, item_tags: List[str]) -> int:
        item_id = self.item_ids.identify(item)
        self.feature_store.put_item_features(item_id, item_tags)
        return item_id

    def clear_item_features(self, item_ids: Optional[List[int]] = None) -> None:
        self.feature_store.clear_item_features(item_ids)

    # ------------------------------------------------------------ ingest
    def _ingest(self, interactions: Iterable[Tuple[Any, Any, float, float]], update_interaction: bool
                ) -> Tuple[np.ndarray, np.ndarray]:
        """Identify and store a batch.  Returns the internal (user_ids, item_ids) that were stored."""
        rows = interactions if isinstance(interactions, list) else list(interactions)
        if not rows:
            return np.empty(0, np.int64), np.empty(0, np.int64)
        try:
            users, items, tstamps, ratings = zip(*rows)
            ts, dl = np.asarray(tstamps), np.asarray(ratings)
            # numeric columns only: np.asarray(..., dtype=float64) would silently PARSE "3.5", where the
            # reference's `current + delta` raises and the row is skipped (interactions.py:103-105)
            if ts.dtype.kind not in "fiub" or dl.dtype.kind not in "fiub":
                raise ValueError("non-numeric timestamp or rating")
            ts, dl = ts.astype(np.float64), dl.astype(np.float64)
            if ts.shape != (len(rows),) or dl.shape != (len(rows),):
                raise ValueError("malformed interaction batch")
            clean = self._batch_is_homogeneous(users, self.user_ids) and self._batch_is_homogeneous(items, self.item_ids)
        except Exception:
            clean = False
        if clean:
            try:
                uid = self.user_ids.identify_many(users)
                iid = self.item_ids.identify_many(items)
                self.interactions.add_interactions_batch(uid, iid, ts, dl, upsert=update_interaction)
                return uid, iid
            except Exception:
                # nothing has been stored yet (the store validates before it writes) and identify() is
                # idempotent, so the per-interaction path below redoes the batch row by row and skips the
                # offending rows like the reference does
                pass
        # per-interaction path with the reference's swallow-and-warn convention (base.py:86-94)
        u_out: List[int] = []
        i_out: List[int] = []
        for row in rows:
            try:
                user, item, tstamp, rating = row
                user_id = self.user_ids.identify(user)
                item_id = self.item_ids.identify(item)
                self.interactions.add_interaction(user_id, item_id, tstamp, rating, upsert=update_interaction)
                u_out.append(user_id)
                i_out.append(item_id)
            except Exception as e:
                logging.warning(f"Error processing interaction: {e}")
                continue
        return np.asarray(u_out, dtype=np.int64), np.asarray(i_out, dtype=np.int64)

    def add_interactions_columns(self, users: np.ndarray, items: np.ndarray, tstamps: np.ndarray, ratings: np.ndarray,
                                 update_interaction: bool = False, record_interactions: bool = False) -> None:
        """Columnar add_interactions: the same state change as add_interactions(zip(users, items,
        tstamps, ratings)) without materialising a Python tuple per interaction.  Integer id columns
        pass through in one vectorised call; anything else (string ids, negative ids, mixed kinds)
        goes through the tuple path and its per-interaction warn-and-skip handling."""
        users, items = np.asarray(users), np.asarray(items)
        fast = (users.dtype.kind in "iu" and items.dtype.kind in "iu" and len(users) > 0
                and not self.user_ids.force_identify and not self.item_ids.force_identify
                and self.user_ids.pass_through is not False and self.item_ids.pass_through is not False
                and int(users.min()) >= 0 and int(items.min()) >= 0)
        if not fast:
            self.add_interactions(list(zip(users.tolist(), items.tolist(), np.asarray(tstamps).tolist(),
                                           np.asarray(ratings).tolist())),
                                  update_interaction=update_interaction, record_interactions=record_interactions)
            return
        uid = self.user_ids.identify_many(users)
        iid = self.item_ids.identify_many(items)
        tag0 = self._store_tag()
        self.interactions.add_interactions_batch(uid, iid, np.asarray(tstamps, dtype=np.float64),
                                                 np.asarray(ratings, dtype=np.float64), upsert=update_interaction,
                                                 device_fold=self._bulk_folder(len(uid)))
        self._stored(tag0, uid, iid)
        if record_interactions:
            self._record_batch(uid, iid)

    def _bulk_folder(self, n: int) -> Any:
        """A device routine that reduces a bulk batch of n interactions to its distinct pairs (see
        UserItemInteractions.add_interactions_batch), or None: models with a GPU backend override this."""
        return None

    def _store_tag(self) -> Any:
        """What a copy of the store must match to be current (models with time decay add max_timestamp)."""
        return self.interactions.version

    def _stored(self, tag_before: Any, user_ids: np.ndarray, item_ids: np.ndarray) -> None:
        """Hook: the interactions (user_ids, item_ids) have just been written to the store, whose _store_tag() was
        `tag_before`.  Models that mirror the store elsewhere (device-resident X) advance the mirror."""

    @staticmethod
    def _batch_is_homogeneous(objs: Tuple[Any, ...], ident: Identifier) -> bool:
        """True when identify() cannot raise for any element of the batch."""
        all_int = all(isinstance(o, (int, np.integer)) for o in objs)
        if ident.force_identify:
            return True
        if all_int:
            return ident.pass_through is not False and all(o >= 0 for o in objs)
        none_int = not any(isinstance(o, (int, np.integer)) for o in objs)
        if not (none_int and ident.pass_through is not True):
            return False
        try:                       # an unhashable id (a JSON list, ...) raises in identify(): per-row path
            for o in objs:
                hash(o)
        except TypeError:
            return False
        return True

    def add_interactions(self, interactions: Iterable[Tuple[Any, Any, float, float]],
                         update_interaction: bool = False, record_interactions: bool = False) -> None:
        uid, iid = self._ingest(interactions, update_interaction)
        if record_interactions:
            self._record_batch(uid, iid)

    def _record_batch(self, user_ids: np.ndarray, item_ids: np.ndarray) -> None:
        for u, i in zip(user_ids.tolist(), item_ids.tolist()):
            self._record_interactions(u, i, 0.0, 0.0)

    @abstractmethod
    def _record_interactions(self, user_id: int, item_id: int, tstamp: float, rating: float) -> None:
        raise NotImplementedError("_record_interactions method must be implemented in the derived class")

    def fit(self, interactions: Iterable[Tuple[Any, Any, float, float]], update_interaction: bool = False,
            progress_bar: bool = True) -> "BaseModel":
        self.add_interactions(interactions, update_interaction=update_interaction, record_interactions=True)
        return self._fit_recorded(progress_bar=progress_bar)

    @abstractmethod
    def _fit_recorded(self, parallel: bool = False, progress_bar: bool = True) -> "BaseModel":
        raise NotImplementedError("_fit_recorded method must be implemented in the derived class")

    @abstractmethod
    def bulk_fit(self, parallel: bool = True, progress_bar: bool = True) -> "BaseModel":
        raise NotImplementedError("bulk_fit method must be implemented in the derived class")

    # ------------------------------------------------------------ recommend
    def _candidate_ids(self, candidate_items: Optional[List[Any]]) -> Optional[List[int]]:
        if candidate_items is None:
            return None
        out = []
        for item in candidate_items:
            item_id = self.item_ids.get_id(item)
            if item_id is None:
                continue
            if self.item_ids.pass_through and item_id > self.interactions.max_item_id:
                continue
            out.append(item_id)
        return out or None

    def _known_user_id(self, user: Any) -> Optional[int]:
        user_id = self.user_ids.get_id(user)
        if user_id is not None and self.user_ids.pass_through and user_id > self.interactions.max_user_id:
            return None   # an integer id the store has never seen is a cold-start user
        return user_id

    def recommend(self, user: Any, candidate_items: Optional[List[Any]] = None, user_tags: Optional[List[str]] = None,
                  top_k: int = 10, filter_interacted: bool = True) -> List[Any]:
        candidate_item_ids = self._candidate_ids(candidate_items)
        user_id = self._known_user_id(user)
        if user_id is None:
            hot = self.interactions.get_hot_items(top_k, filter_interacted=False)
            if candidate_item_ids is not None:
                hot = [i for i in hot if i in candidate_item_ids]
            return hot
        rec = self._recommend(user_id, candidate_item_ids=candidate_item_ids, user_tags=user_tags, top_k=top_k,
                              filter_interacted=filter_interacted)
        return [self.item_ids.get(i) for i in rec]

    @abstractmethod
    def _recommend(self, user_id: int, candidate_item_ids: Optional[List[int]] = None,
                   user_tags: Optional[List[str]] = None, top_k: int = 10, filter_interacted: bool = True) -> List[int]:
        raise NotImplementedError("_recommend method must be implemented in the derived class")

    def recommend_batch(self, users: List[Any], candidate_items: Optional[List[Any]] = None,
                        users_tags: Optional[List[List[str]]] = None, top_k: int = 10,
                        filter_interacted: bool = True, as_arrays: bool = False) -> Any:
        """rtrec/models/base.py:188-269: hot / cold split, candidates mapped and bounded, one list of raw item ids per user.

        Integer (pass-through) user ids take a vectorised route: ONE compare of the id array against `max_user_id` instead of
        a `get_id` per user, and the array goes to the model's array hook (`_recommend_hot_arrays`) as it is -- the
        per-user Python of the reference's loop is what bounded `B / wall(recommend_batch)` (SURVEY 8d's score metric).

        `as_arrays=True` (an extension; the reference returns lists only) returns `(ids[B, top_k], counts[B])` numpy arrays
        instead of B Python lists: row b's answer is `ids[b, :counts[b]]`, the rest of the row is -1 (None for mapped ids)."""
        arr = self._int_user_array(users) if not users_tags else None
        if arr is not None:
            return self._recommend_batch_int_users(arr, candidate_items, top_k, filter_interacted, as_arrays)
        hot_pos: List[int] = []
        hot_ids: List[int] = []
        cold_pos: List[int] = []
        cold_ids: List[Optional[int]] = []
        for pos, user in enumerate(users):
            uid = self._known_user_id(user)
            if uid is None:
                cold_pos.append(pos)
                cold_ids.append(self.handle_unknown_user(user))
            else:
                hot_pos.append(pos)
                hot_ids.append(uid)
        candidate_item_ids = self._candidate_ids(candidate_items)

        def to_items(rows: List[List[int]]) -> List[List[Any]]:
            if self.item_ids.pass_through:          # integer ids are their own internal ids
                return rows
            return [[self.item_ids.get(i) for i in row] for row in rows]

        if not cold_ids:
            results = to_items(self._recommend_hot_batch(hot_ids, candidate_item_ids=candidate_item_ids,
                                                         users_tags=users_tags, top_k=top_k,
                                                         filter_interacted=filter_interacted))
            return self._lists_as_arrays(results, top_k) if as_arrays else results
        results: List[List[Any]] = [[] for _ in users]
        cold_tags = [users_tags[p] for p in cold_pos] if users_tags else None
        cold = to_items(self._recommend_cold_batch(cold_ids, candidate_item_ids=candidate_item_ids,
                                                   users_tags=cold_tags, top_k=top_k))
        for p, row in zip(cold_pos, cold):
            results[p] = row
        if hot_ids:
            hot_tags = [users_tags[p] for p in hot_pos] if users_tags else None
            hot = to_items(self._recommend_hot_batch(hot_ids, candidate_item_ids=candidate_item_ids,
                                                     users_tags=hot_tags, top_k=top_k,
                                                     filter_interacted=filter_interacted))
            for p, row in zip(hot_pos, hot):
                results[p] = row
        return self._lists_as_arrays(results, top_k) if as_arrays else results

    def _int_user_array(self, users: Any) -> Optional[np.ndarray]:
        """`users` as an int64 array when every element is an integer that passes through unmapped (then `get_id` is the
        identity and the hot / cold split is one compare), else None -> the per-user loop with the reference's checks."""
        ident = self.user_ids
        if ident.force_identify or ident.pass_through is not True:
            return None
        if (type(self).handle_unknown_user is not BaseModel.handle_unknown_user
                or type(self)._recommend_cold_batch is not BaseModel._recommend_cold_batch):
            return None                  # a subclass maps / serves unknown users itself: keep its per-user hooks
        if isinstance(users, np.ndarray):
            return users.astype(np.int64, copy=False) if (users.ndim == 1 and users.dtype.kind in "iu") else None
        if isinstance(users, range):
            return np.arange(users.start, users.stop, users.step, dtype=np.int64)
        if not isinstance(users, (list, tuple)) or len(users) == 0 or not isinstance(users[0], (int, np.integer)):
            return None
        try:
            arr = np.asarray(users)
        except Exception:
            return None
        # (a str / float / None among the ints makes numpy choose another dtype: those batches keep the loop)
        return arr.astype(np.int64, copy=False) if (arr.ndim == 1 and arr.dtype.kind in "iu") else None

    def _recommend_batch_int_users(self, uid: np.ndarray, candidate_items: Optional[List[Any]], top_k: int,
                                   filter_interacted: bool, as_arrays: bool) -> Any:
        candidate_item_ids = self._candidate_ids(candidate_items)
        B = int(uid.shape[0])
        if B == 0:                                       # an empty batch: nothing to score
            return (np.empty((0, top_k), dtype=np.int64), np.zeros(0, dtype=np.int32)) if as_arrays else []
        cold = uid > self.interactions.max_user_id       # an integer id the store has never seen is a cold-start user
        n_cold = int(np.count_nonzero(cold))
        mapped = not self.item_ids.pass_through          # string item ids: internal -> raw through id_to_obj
        hot_rows_list = None
        if n_cold:
            hot_rows_list = self._recommend_cold_batch([None], candidate_item_ids=candidate_item_ids, top_k=top_k)[0]
        h_ids = h_counts = None
        if n_cold < B:
            h_ids, h_counts = self._recommend_hot_arrays(uid if n_cold == 0 else uid[~cold], candidate_item_ids, top_k,
                                                         filter_interacted)
        if not as_arrays:
            # lists: the hot users' rows in one tolist() (int32 as it comes off the device), cut only where a list is short;
            # cold users all get the hot-items list
            hot_lists: List[List[Any]] = []
            if h_ids is not None:
                hot_lists = self._rows_as_lists(h_ids)
                if len(hot_lists) and int(h_counts.min()) < h_ids.shape[1]:
                    # only the SHORT rows are cut (a comprehension over all rows allocates B new lists: with the collector
                    # running, 117 ms for 100,000 users of which 140 were short)
                    for p_ in np.flatnonzero(h_counts < h_ids.shape[1]).tolist():
                        hot_lists[p_] = hot_lists[p_][:int(h_counts[p_])]
                if mapped:
                    get = self.item_ids.get
                    hot_lists = [[get(i) for i in row] for row in hot_lists]
            if n_cold == 0:
                return hot_lists
            cold_row = [self.item_ids.get(i) for i in hot_rows_list] if mapped else hot_rows_list
            rows: List[List[Any]] = [list(cold_row) for _ in range(B)]      # a list of its own per user, like the reference's
            for p_, r_ in zip(np.flatnonzero(~cold).tolist(), hot_lists):
                rows[p_] = r_
            return rows
        if n_cold == 0:
            ids, counts = h_ids, h_counts
        else:
            width = max(top_k, len(hot_rows_list), h_ids.shape[1] if h_ids is not None else 0)
            ids = np.full((B, width), -1, dtype=np.int64)
            counts = np.zeros(B, dtype=np.int32)
            if h_ids is not None:
                hot_pos = np.flatnonzero(~cold)
                ids[hot_pos, :h_ids.shape[1]] = h_ids
                counts[hot_pos] = h_counts
            if hot_rows_list:
                ids[cold, :len(hot_rows_list)] = np.asarray(hot_rows_list, dtype=np.int64)[None, :]
            counts[cold] = len(hot_rows_list)
        if mapped:
            lut = np.empty(len(self.item_ids.id_to_obj) + 1, dtype=object)
            lut[:-1] = self.item_ids.id_to_obj
            lut[-1] = None
            live = np.arange(ids.shape[1])[None, :] < counts[:, None]
            bad = live & ((ids < 0) | (ids >= len(lut) - 1))
            if bad.any():
                from ..utils.identifiers import IdentifierError
                raise IdentifierError(self.item_ids.name, int(ids[bad][0]))
            ids = lut[np.where(live, ids, -1)]
        return ids, counts

    def _recommend_hot_arrays(self, user_ids: np.ndarray, candidate_item_ids: Optional[List[int]], top_k: int,
                              filter_interacted: bool) -> Tuple[np.ndarray, np.ndarray]:
        """Array form of _recommend_hot_batch: (ids[B, k'], counts[B]), internal item ids, row b valid up to counts[b].
        The default goes through the list hook; models with a batched scorer override it."""
        rows = self._recommend_hot_batch(user_ids.tolist(), candidate_item_ids=candidate_item_ids, top_k=top_k,
                                         filter_interacted=filter_interacted)
        return self._lists_as_arrays(rows, top_k)

    @staticmethod
    def _rows_as_lists(ids: np.ndarray) -> List[List[int]]:
        """ids.tolist() with the cyclic collector paused: B new lists of ints hold no cycles, but every 700 allocations
        would start a young-generation scan over them (138k users x 10 items: 81 -> 62 ms; what remains is CPython creating
        1.5 M objects -- the floor of any list-of-lists answer, which is why as_arrays exists)."""
        import gc
        if ids.shape[0] < 4096 or not gc.isenabled():
            return ids.tolist()
        gc.disable()
        try:
            return ids.tolist()
        finally:
            gc.enable()

    @staticmethod
    def _lists_as_arrays(rows: List[List[Any]], top_k: int) -> Tuple[np.ndarray, np.ndarray]:
        counts = np.fromiter((len(r) for r in rows), dtype=np.int32, count=len(rows))
        width = max(int(top_k), int(counts.max()) if len(rows) else 0)
        ints = all(isinstance(x, (int, np.integer)) for r in rows for x in r)
        ids = np.full((len(rows), width), -1, dtype=np.int64) if ints else np.full((len(rows), width), None, dtype=object)
        for b, r in enumerate(rows):
            if r:
                ids[b, :len(r)] = r
        return ids, counts

    def handle_unknown_user(self, user: Any) -> Optional[int]:
        return None

    def _recommend_cold_batch(self, user_ids: List[Optional[int]], candidate_item_ids: Optional[List[int]] = None,
                              users_tags: Optional[List[List[str]]] = None, top_k: int = 10) -> List[List[int]]:
        hot = self.interactions.get_hot_items(top_k, filter_interacted=False)
        if candidate_item_ids is not None:
            hot = [i for i in hot if i in candidate_item_ids]
        return [hot for _ in user_ids]

    def _recommend_hot_batch(self, user_ids: List[int], candidate_item_ids: Optional[List[int]] = None,
                             users_tags: Optional[List[List[str]]] = None, top_k: int = 10,
                             filter_interacted: bool = True) -> List[List[int]]:
        if users_tags:
            assert len(user_ids) == len(users_tags), (
                f"Number of user tags must match the number of users. Got {len(user_ids)} users and "
                f"{len(users_tags)} user tags.")
            return [self._recommend(u, candidate_item_ids=candidate_item_ids, user_tags=t, top_k=top_k,
                                    filter_interacted=filter_interacted) for u, t in zip(user_ids, users_tags)]
        return [self._recommend(u, candidate_item_ids=candidate_item_ids, top_k=top_k,
                                filter_interacted=filter_interacted) for u in user_ids]

    # ------------------------------------------------------------ item-to-item, lookups
    def similar_items(self, query_item: Any, query_item_tags: Optional[List[str]] = None, top_k: int = 10,
                      ret_scores: bool = False) -> List[Tuple[Any, float]] | List[Any]:
        query_item_id = self.item_ids.identify(query_item)   # like the reference, registers unseen items
        if query_item_id is None:
            return []
        pairs = self._similar_items(query_item_id, query_item_tags=query_item_tags, top_k=top_k)
        if ret_scores:
            return [(self.item_ids.get(i), s) for i, s in pairs]
        return [self.item_ids.get(i) for i, _ in pairs]

    def similar_items_batch(self, query_items: List[Any], query_item_tags: Optional[List[str]] = None, top_k: int = 10,
                            ret_scores: bool = False) -> List[Any]:
        """[similar_items(q, ...) for q in query_items] in one pass.  Models that can answer many queries at
        once override _similar_items_batch; the default loops."""
        ids = [self.item_ids.identify(q) for q in query_items]       # same order and side effects as one call per query
        known = [i for i in ids if i is not None]
        answers = iter(self._similar_items_batch(known, query_item_tags=query_item_tags, top_k=top_k))
        out: List[Any] = []
        for i in ids:
            if i is None:
                out.append([])
                continue
            pairs = next(answers)
            out.append([(self.item_ids.get(j), s) for j, s in pairs] if ret_scores else [self.item_ids.get(j) for j, _ in pairs])
        return out

    def _similar_items_batch(self, query_item_ids: List[int], query_item_tags: Optional[List[str]] = None,
                             top_k: int = 10) -> List[List[Tuple[int, float]]]:
        return [self._similar_items(i, query_item_tags=query_item_tags, top_k=top_k) for i in query_item_ids]

    def get_users_by_items(self, items: List[Any]) -> List[Any]:
        item_ids = [i for i in (self.item_ids.get_id(item) for item in items) if i is not None]
        if not item_ids:
            return []
        return [self.user_ids.get(u) for u in self.interactions.get_users_by_items(item_ids)]

    @abstractmethod
    def _similar_items(self, query_item_id: int, query_item_tags: Optional[List[str]] = None, top_k: int = 10
                       ) -> List[Tuple[int, float]]:
        raise NotImplementedError("_similar_items method must be implemented in the derived class")

    # ------------------------------------------------------------ persistence
    def save(self, f: FileLike) -> int:
        return f.write(pickle.dumps(self._serialize(), protocol=pickle.HIGHEST_PROTOCOL))

    @classmethod
    def load(cls, f: FileLike) -> "BaseModel":
        """Load a model saved by rtrec_amd OR by the reference (rtrec.models.SLIM.save)."""
        from ..compat import loads as compat_loads
        return cls._deserialize(compat_loads(f.read()))

    @classmethod
    def loads(cls, data: bytes) -> "BaseModel":
        return cls.load(BytesIO(data))

    @abstractmethod
    def _serialize(self) -> dict:
        raise NotImplementedError("_serialize method must be implemented in the derived class")

    @classmethod
    @abstractmethod
    def _deserialize(cls, data: dict) -> "BaseModel":
        raise NotImplementedError("_deserialize method must be implemented in the derived class")
