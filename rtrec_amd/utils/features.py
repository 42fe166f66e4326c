"""Tag -> feature-id registry for users and items.

SLIM never reads features; the reference keeps the registry on every model so that
register_user_feature / register_item_feature work uniformly
(/root/reference/rtrec/models/base.py:36-70, /root/reference/rtrec/utils/features.py:9-224).
This is the thin host-side equivalent: same method names and return shapes.
"""
from __future__ import annotations

from typing import Dict, Generic, Iterable, List, Optional, TypeVar

import numpy as np
from scipy.sparse import csr_matrix

T = TypeVar("T")


class IndexedSet(Generic[T]):
    """Insertion-ordered set with O(1) key -> position lookup."""

    def __init__(self, iterable: Optional[Iterable[T]] = None) -> None:
        self._pos: Dict[T, int] = {}
        self._keys: List[T] = []
        for item in iterable or ():
            self.add(item)

    def __setstate__(self, state):
        # reference layout: _key_to_index / _index_to_key (rtrec/utils/collections.py:11-12)
        if "_key_to_index" in state:
            state = {"_pos": state["_key_to_index"], "_keys": state["_index_to_key"]}
        self.__dict__.update(state)

    def add(self, key: T) -> int:
        pos = self._pos.get(key)
        if pos is None:
            pos = self._pos[key] = len(self._keys)
            self._keys.append(key)
        return pos

    def index(self, key: T, default: int = -1) -> int:
        return self._pos.get(key, default)

    def __len__(self) -> int:
        return len(self._keys)

    def __contains__(self, key) -> bool:
        return key in self._pos

    def __iter__(self):
        return iter(self._keys)

    def __getitem__(self, index):
        return self._keys[index]


class _Side:
    def __init__(self) -> None:
        self.vocab: IndexedSet = IndexedSet()
        self.by_entity: Dict[int, List[int]] = {}

    def put(self, entity: int, tags: List[str], append: bool) -> None:
        ids = self.by_entity.get(entity, []) if append else []
        for tag in tags:
            fid = self.vocab.add(tag)
            if fid not in ids:
                ids.append(fid)
        self.by_entity[entity] = ids

    def clear(self, entities: Optional[List[int]]) -> None:
        if entities is None:
            self.by_entity.clear()
        else:
            for e in entities:
                self.by_entity.pop(e, None)

    def repr(self, tags: List[str]) -> csr_matrix:
        cols = [f for f in (self.vocab.index(t) for t in tags) if f >= 0]
        return csr_matrix((np.ones(len(cols)), (np.zeros(len(cols)), np.array(cols))),
                          shape=(1, len(self.vocab)), dtype=np.float32)

    def matrix(self, entities: Optional[List[int]], tags: Optional[List[List[str]]], n: Optional[int]):
        if len(self.vocab) == 0:
            return None
        rows: List[int] = []
        cols: List[int] = []
        if entities is None:
            for e, fids in self.by_entity.items():
                rows += [e] * len(fids)
                cols += fids
        elif tags:
            assert len(entities) == len(tags), (
                f"Number of IDs and tags should be equal. Got {len(entities)} IDs and {len(tags)} tags.")
            for e, ts in zip(entities, tags):
                fids = [f for f in (self.vocab.index(t) for t in ts) if f >= 0]
                rows += [e] * len(fids)
                cols += fids
        else:
            for e in entities:
                fids = self.by_entity.get(e, [])
                rows += [e] * len(fids)
                cols += fids
        if n is None:
            n = (max(rows) if rows else 0) + 1
        return csr_matrix((np.ones(len(rows)), (rows, cols)), shape=(n, len(self.vocab)), dtype=np.float32)


class FeatureStore:
    def __init__(self) -> None:
        self._users = _Side()
        self._items = _Side()

    def __setstate__(self, state):
        # reference layout: user_features / item_features (IndexedSet) + *_feature_map dicts
        if "user_features" in state:
            self._users, self._items = _Side(), _Side()
            self._users.vocab, self._users.by_entity = state["user_features"], state["user_feature_map"]
            self._items.vocab, self._items.by_entity = state["item_features"], state["item_feature_map"]
        else:
            self.__dict__.update(state)

    @property
    def user_features(self) -> IndexedSet:
        return self._users.vocab

    @property
    def item_features(self) -> IndexedSet:
        return self._items.vocab

    @property
    def user_feature_map(self) -> Dict[int, List[int]]:
        return self._users.by_entity

    @property
    def item_feature_map(self) -> Dict[int, List[int]]:
        return self._items.by_entity

    def num_user_features(self) -> int:
        return len(self._users.vocab)

    def num_item_features(self) -> int:
        return len(self._items.vocab)

    def clear_user_features(self, user_ids: Optional[List[int]] = None) -> None:
        self._users.clear(user_ids)

    def clear_item_features(self, item_ids: Optional[List[int]] = None) -> None:
        self._items.clear(item_ids)

    def put_user_features(self, user_id: int, user_tags: List[str], append: bool = False) -> None:
        self._users.put(user_id, user_tags, append)

    def put_item_features(self, item_id: int, item_tags: List[str], append: bool = False) -> None:
        self._items.put(item_id, item_tags, append)

    def get_user_feature_repr(self, user_tags: List[str]) -> csr_matrix:
        return self._users.repr(user_tags)

    def get_item_feature_repr(self, item_tags: List[str]) -> csr_matrix:
        return self._items.repr(item_tags)

    def build_user_features_matrix(self, user_ids: Optional[List[int]] = None,
                                   users_tags: Optional[List[List[str]]] = None, num_users: Optional[int] = None):
        return self._users.matrix(user_ids, users_tags, num_users)

    def build_item_features_matrix(self, item_ids: Optional[List[int]] = None,
                                   items_tags: Optional[List[List[str]]] = None, num_items: Optional[int] = None):
        return self._items.matrix(item_ids, items_tags, num_items)
