"""DataFrame facade: the drop-in for rtrec.recommender.Recommender
(/root/reference/rtrec/recommender.py:20-223).

Same methods, arguments, mini-batching and the same printed wall-clock / "Throughput: N
samples/sec" figure (interactions divided by ingest + fit time, recommender.py:81,126) that
BASELINE.md quotes.  Mini-batches are cut from the DataFrame columns as arrays rather than
through itertuples().
"""
from __future__ import annotations

import math
import time
from typing import Any, Dict, Iterable, Iterator, List, Optional, Tuple

import pandas as pd

from .models.base import BaseModel
from .utils.metrics import compute_scores

_COLUMNS = ["user", "item", "tstamp", "rating"]
_COLUMNAR_CHUNK = 1 << 22


class Recommender:
    def __init__(self, model: BaseModel, use_generator: bool = True):
        self.model = model
        self.use_generator = use_generator

    def get_model(self) -> BaseModel:
        return self.model

    def partial_fit(self, user_interactions: Iterable[Tuple[int, int, int, float]],
                    update_interaction: bool = False) -> "Recommender":
        t0 = time.time()
        self.model.fit(user_interactions, update_interaction=update_interaction, progress_bar=False)
        print(f"Fit completed in {time.time() - t0:.2f} seconds")
        return self

    def _register_tags(self, user_tags, item_tags) -> None:
        for user, tags in (user_tags or {}).items():
            self.model.register_user_feature(user, tags)
        for item, tags in (item_tags or {}).items():
            self.model.register_item_feature(item, tags)

    def _ingest_frame(self, train_data: pd.DataFrame, batch_size: int, update_interaction: bool, record: bool,
                      assume_sorted: bool) -> None:
        frame = train_data[_COLUMNS]
        if not assume_sorted:
            frame = frame.sort_values("tstamp", ascending=True)
        columnar = getattr(self.model, "add_interactions_columns", None)
        if columnar is not None and all(frame[c].dtype.kind in "iuf" for c in _COLUMNS):
            # numeric frame: hand the columns over as arrays.  The store applies a chunk with the same
            # sequential semantics as one add_interaction per row, so mini-batch boundaries (batch_size)
            # do not change the result; only the per-row Python objects of the reference loop go away.
            u, i, t, r = (frame[c].to_numpy() for c in _COLUMNS)
            chunk = getattr(self.model, "bulk_chunk_rows", None) or _COLUMNAR_CHUNK
            for s in range(0, len(frame), chunk):
                e = s + chunk
                columnar(u[s:e], i[s:e], t[s:e], r[s:e], update_interaction=update_interaction,
                         record_interactions=record)
            return
        for batch in Recommender.generate_batches(frame, batch_size, as_generator=self.use_generator):
            self.model.add_interactions(batch, update_interaction=update_interaction, record_interactions=record)

    def fit(self, train_data: pd.DataFrame, user_tags: Optional[Dict[Any, List[str]]] = None,
            item_tags: Optional[Dict[Any, List[str]]] = None, batch_size: int = 1_000,
            update_interaction: bool = False, parallel: bool = False, assume_sorted: bool = True) -> "Recommender":
        """Incremental fit: ingest in mini-batches, then refit the items that were touched."""
        t0 = time.time()
        self._register_tags(user_tags, item_tags)
        self._ingest_frame(train_data, batch_size, update_interaction, True, assume_sorted)
        self.model._fit_recorded(parallel=parallel, progress_bar=True)
        dt = time.time() - t0
        print(f"Fit completed in {dt:.2f} seconds")
        print(f"Throughput: {len(train_data) / dt:.2f} samples/sec")
        return self

    def bulk_fit(self, train_data: pd.DataFrame, user_tags: Optional[Dict[Any, List[str]]] = None,
                 item_tags: Optional[Dict[Any, List[str]]] = None, batch_size: int = 1_000,
                 update_interaction: bool = False, parallel: bool = True, assume_sorted: bool = True) -> "Recommender":
        """Ingest everything, then fit every item column."""
        t0 = time.time()
        self._register_tags(user_tags, item_tags)
        self._ingest_frame(train_data, batch_size, update_interaction, False, assume_sorted)
        self.model.bulk_fit(parallel=parallel, progress_bar=True)
        dt = time.time() - t0
        print(f"Fit completed in {dt:.2f} seconds")
        print(f"Throughput: {len(train_data) / dt:.2f} samples/sec")
        return self

    def recommend(self, user: Any, candidate_items: Optional[List[Any]] = None, user_tags: Optional[List[str]] = None,
                  top_k: int = 10, filter_interacted: bool = True) -> List[Any]:
        return self.model.recommend(user, candidate_items, user_tags, top_k, filter_interacted)

    def recommend_batch(self, users: List[Any], candidate_items: Optional[List[Any]] = None,
                        users_tags: Optional[List[List[str]]] = None, top_k: int = 10,
                        filter_interacted: bool = True, as_arrays: bool = False) -> Any:
        """rtrec/recommender.py:141-151.  `as_arrays=True` (extension): (ids[B, top_k], counts[B]) numpy arrays instead of B
        Python lists -- see BaseModel.recommend_batch."""
        if as_arrays:
            return self.model.recommend_batch(users, candidate_items, users_tags, top_k, filter_interacted, as_arrays=True)
        return self.model.recommend_batch(users, candidate_items, users_tags, top_k, filter_interacted)

    def similar_items(self, query_items: List[Any], query_item_tags: Optional[List[str]] = None, top_k: int = 10,
                      ret_scores: bool = False):
        batch = getattr(self.model, "similar_items_batch", None)
        if batch is not None:        # one kernel launch for all queries instead of one per query
            return batch(query_items, query_item_tags, top_k, ret_scores)
        return [self.model.similar_items(item, query_item_tags, top_k, ret_scores) for item in query_items]

    def evaluate(self, test_data: pd.DataFrame, user_tags: Optional[Dict[Any, List[str]]] = None,
                 recommend_size: int = 10, batch_size=100, filter_interacted: bool = True) -> Dict[str, float]:
        """Average ranking metrics over the users of test_data (columns user, item)."""
        truth = test_data.groupby("user")["item"].apply(list).to_dict()
        users = list(truth.keys())

        def pairs() -> Iterator[Tuple[List[Any], List[Any]]]:
            for s in range(0, len(users), batch_size):
                chunk = users[s:s + batch_size]
                tags = [user_tags.get(u, []) for u in chunk] if user_tags else None
                recs = self.recommend_batch(chunk, users_tags=tags, top_k=recommend_size,
                                            filter_interacted=filter_interacted)
                for u, rec in zip(chunk, recs):
                    yield rec, truth[u]
        return compute_scores(pairs(), recommend_size)

    @staticmethod
    def generate_batches(df: pd.DataFrame, batch_size: int = 1_000, as_generator: bool = False
                         ) -> Iterator[Iterable[Tuple[int, int, int, float]]]:
        """Mini-batches of (user, item, tstamp, rating) tuples in row order."""
        cols = [df[c].tolist() for c in df.columns[:4]]
        for s in range(0, len(df), batch_size):
            batch = list(zip(*(c[s:s + batch_size] for c in cols)))
            yield iter(batch) if as_generator else batch
