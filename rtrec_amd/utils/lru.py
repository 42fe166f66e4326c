"""Capacity-bounded recency set with hit counts (cold-start "hot items").

Behaviour follows rtrec.utils.lru.LRUFreqSet (/root/reference/rtrec/utils/lru.py:11-123):
adding an existing key bumps its count and makes it most recent; adding a new key at capacity
first evicts the least recently added/bumped key; get_freq_items() lists keys by count,
descending, ties in recency order (Python's stable sort over the recency-ordered dict).
"""
from __future__ import annotations

from collections import OrderedDict
from collections.abc import MutableSet
from typing import Any, Iterable, Iterator, List, Optional

import numpy as np


class LRUFreqSet(MutableSet):
    def __init__(self, capacity: int):
        if capacity <= 0:
            raise ValueError("Capacity must be greater than 0.")
        self.capacity = capacity
        self.data: "OrderedDict[Any, int]" = OrderedDict()   # key -> hit count, oldest first

    def add(self, value: Any) -> None:
        hits = self.data.pop(value, None)
        if hits is None:
            if len(self.data) >= self.capacity:
                self.data.popitem(last=False)
            hits = 0
        self.data[value] = hits + 1

    def add_many(self, values: Iterable[Any]) -> None:
        """Same end state as calling add() for each value in order."""
        if isinstance(values, np.ndarray):
            if values.dtype.kind in "iu" and values.ndim == 1 and len(values) > 64 and (
                    self._add_many_ints(values) or self._replay_native(values)):
                return
            values = values.tolist()       # python scalars: the keys are handed back to callers
        values = list(values)
        fresh = {v for v in values if v not in self.data}
        if len(self.data) + len(fresh) > self.capacity:
            for v in values:        # evictions depend on the exact interleaving
                self.add(v)
            return
        hits: dict = {}
        for v in values:
            hits[v] = hits.get(v, 0) + 1
        # final recency order = order of LAST occurrence
        seen = set()
        last_order = []
        for v in reversed(values):
            if v not in seen:
                seen.add(v)
                last_order.append(v)
        for v in reversed(last_order):
            self.data[v] = self.data.pop(v, 0) + hits[v]

    def _add_many_ints(self, values: "np.ndarray") -> bool:
        """Vectorised add_many for an integer array: hit counts and the order of LAST occurrence come
        from numpy (bincount + a scatter of positions: with repeated indices the last write wins), the
        dict is touched once per distinct key.  Returns False (nothing done) when the batch would
        overflow the capacity -- evictions depend on the exact interleaving."""
        n = len(values)
        lo, hi = int(values.min()), int(values.max())
        if lo < 0 or hi - lo > max(4 * n, 1 << 22):       # sparse id range: sort-based grouping instead
            uniq, rev_first, counts = np.unique(values[::-1], return_index=True, return_counts=True)
            last_pos = n - 1 - rev_first
        else:
            shifted = values - lo
            counts_all = np.bincount(shifted, minlength=hi - lo + 1)
            last_all = np.zeros(hi - lo + 1, dtype=np.int64)
            last_all[shifted] = np.arange(n, dtype=np.int64)
            present = np.flatnonzero(counts_all)
            uniq, counts, last_pos = present + lo, counts_all[present], last_all[present]
        return self.add_counted(uniq, counts, last_pos)

    def add_counted(self, uniq: "np.ndarray", counts: "np.ndarray", last_pos: "np.ndarray") -> bool:
        """The end state of add() for every value of a batch, given the batch's statistics: its distinct keys, how often each
        occurs and where each occurs LAST (any increasing position).  False (nothing done) when the batch would overflow
        the capacity."""
        keys = uniq.tolist()
        data = self.data
        fresh = sum(1 for k in keys if k not in data)
        if len(data) + fresh > self.capacity:
            return False
        order = np.argsort(last_pos, kind="stable")
        cnt = counts.tolist()
        for j in order.tolist():
            k = keys[j]
            data[k] = data.pop(k, 0) + cnt[j]
        return True

    def _replay_native(self, values: "np.ndarray") -> bool:
        """add() for every value in order -- evictions included -- by librtrec_amd.so's host routine
        (rtrec_lru_replay: an intrusive linked list over dense key indices).  For batches that overflow the
        capacity, where the end state depends on the exact interleaving.  The keys of the state and of the
        batch are first mapped to dense indices 0..D-1, so the routine's list nodes are sized by the number of
        DISTINCT keys, not by the largest id (a sparse or large item-id space used to cost a 16 B x max-id
        allocation per overflowing batch).  Returns False (nothing done) when the library is missing or the
        keys are not integers."""
        try:
            from .. import _native
            lib = _native.load()
        except Exception:
            return False
        data = self.data
        vals = np.ascontiguousarray(values, dtype=np.int64)
        try:
            sk = np.fromiter(data.keys(), dtype=np.int64, count=len(data))
        except (TypeError, ValueError, OverflowError):
            return False
        if len(sk) and not all(isinstance(k, int) for k in data):       # e.g. 1.0 would have been cast silently
            return False
        sc = np.fromiter(data.values(), dtype=np.int64, count=len(data))
        uniq = np.unique(np.concatenate([sk, vals]))
        sk_d, vals_d = np.searchsorted(uniq, sk), np.searchsorted(uniq, vals)
        ok, oc = np.empty(self.capacity, np.int64), np.empty(self.capacity, np.int64)
        n_out = int(lib.rtrec_lru_replay(sk_d.ctypes.data, sc.ctypes.data, len(sk_d), vals_d.ctypes.data, len(vals_d),
                                         self.capacity, max(len(uniq), 1), ok.ctypes.data, oc.ctypes.data))
        if n_out < 0:
            return False
        self.data = OrderedDict(zip(uniq[ok[:n_out]].tolist(), oc[:n_out].tolist()))
        return True

    def discard(self, value: Any) -> None:
        if value not in self.data:
            raise KeyError(value)
        del self.data[value]

    def __contains__(self, key: Any) -> bool:
        return key in self.data

    def __iter__(self) -> Iterator[Any]:
        return iter(self.data)

    def __len__(self) -> int:
        return len(self.data)

    def __repr__(self) -> str:
        return f"LRUFreqSet(capacity={self.capacity}, size={len(self.data)})"

    def get_freq_items(self, n: Optional[int] = None, exclude_items: List[Any] = []) -> Iterator[Any]:
        ranked = sorted(self.data.items(), key=lambda kv: kv[1], reverse=True)
        if len(exclude_items) == 0:
            for key, _ in (ranked if n is None else ranked[:n]):
                yield key
            return
        emitted = 0
        for key, _ in ranked:
            if key in exclude_items:
                continue
            if n is not None and emitted >= n:
                return
            emitted += 1
            yield key
