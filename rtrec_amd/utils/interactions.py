"""Columnar user-item interaction store.

Same public behaviour as rtrec.utils.interactions.UserItemInteractions
(/root/reference/rtrec/utils/interactions.py:14-353: additive update with clipping, upsert,
half-life time decay evaluated against max_timestamp, hot-item tracking, CSR/CSC/COO export
of shape (max_user_id+1, max_item_id+1) in float32), but stored as sorted numpy columns
instead of a dict of dicts so that mini-batches are ingested and exported with array
operations: the 100 M-interaction configurations cannot go through ~7 us/interaction of
Python.

Layout: one int64 key per (user, item) pair, key = user << 32 | item, kept sorted, with two
float64 columns (value, timestamp).  Sorted-by-key IS CSR order, so to_csr() is a cumulative
count and to_csc() one stable argsort.  Recent writes sit in a small sorted delta block that
is merged into the base block geometrically (log-structured), so a 1k-interaction mini-batch
never rewrites the whole store.
"""
from __future__ import annotations

import logging
import math
import time
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
from scipy.sparse import coo_matrix, csc_matrix, csr_matrix

from .lru import LRUFreqSet

_SHIFT = 32
_MASK = (1 << _SHIFT) - 1
_DELTA_MERGE_MIN = 1 << 16
_L0_MAX = 1 << 15              # recent writes: a mini-batch merges into at most this many entries


class _Block:
    """Sorted (key, value, timestamp) columns."""
    __slots__ = ("key", "val", "ts", "_perm", "_iptr", "_rows_im", "_val_im", "_ts_im", "_d32_im")

    def __init__(self, key=None, val=None, ts=None):
        self.key = np.empty(0, np.int64) if key is None else key
        self.val = np.empty(0, np.float64) if val is None else val
        self.ts = np.empty(0, np.float64) if ts is None else ts
        self._reset_index()

    def _reset_index(self) -> None:
        self._perm = None
        self._iptr = None
        self._rows_im = None
        self.values_changed()

    def values_changed(self) -> None:
        """val / ts were overwritten in place (same keys): the item-major value mirrors are stale."""
        self._val_im = None
        self._ts_im = None
        self._d32_im = None

    def __len__(self) -> int:
        return int(self.key.shape[0])

    def __getstate__(self):
        return (self.key, self.val, self.ts)

    def __setstate__(self, st):
        self.key, self.val, self.ts = st
        self._reset_index()

    def item_index(self, n_items: int) -> Tuple[np.ndarray, np.ndarray]:
        """(perm, iptr): positions of the entries in item-major order and the CSC-style pointer over
        items.  Built lazily, valid until the block's key set changes; a catalogue that has grown since
        (items this block has never seen) only pads the pointer."""
        if self._iptr is None:
            items = self.key & _MASK
            own = int(items.max()) + 1 if len(self) else 0
            self._perm = np.argsort(items, kind="stable")        # users stay ascending inside an item
            self._iptr = np.zeros(own + 1, dtype=np.int64)
            np.cumsum(np.bincount(items, minlength=own), out=self._iptr[1:])
        if self._iptr.shape[0] < n_items + 1:
            self._iptr = np.concatenate([self._iptr, np.full(n_items + 1 - self._iptr.shape[0], self._iptr[-1])])
        return self._perm, self._iptr

    def item_major(self, n_items: int, static_values: bool):
        """(iptr, rows, val, ts, data32): the block's columns mirrored in item-major (CSC) order, so the
        export of selected item columns is a copy of contiguous slices instead of three random gathers
        and two sorts.  `data32` (float32 values) is kept only when values do not decay with time."""
        perm, iptr = self.item_index(n_items)
        if self._rows_im is None:
            self._rows_im = (self.key[perm] >> _SHIFT).astype(np.int32)
        if self._val_im is None:
            self._val_im = self.val[perm]
            self._ts_im = self.ts[perm]
            self._d32_im = None
        if static_values and self._d32_im is None:
            self._d32_im = self._val_im.astype(np.float32)
        return iptr, self._rows_im, self._val_im, self._ts_im, self._d32_im

    def find(self, keys: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """(found mask, position) of each key."""
        if len(self) == 0:
            return np.zeros(keys.shape, bool), np.zeros(keys.shape, np.int64)
        if keys.shape[0] >= _NATIVE_FIND_MIN and len(self) >= _NATIVE_FIND_MIN and bool(np.all(keys[1:] >= keys[:-1])):
            lib = _native_lib()                 # ascending needles: galloping search at memory speed
            if lib is not None:
                k = np.ascontiguousarray(keys, dtype=np.int64)
                hay = np.ascontiguousarray(self.key)
                pos = np.empty(k.shape[0], np.int64)
                found = np.empty(k.shape[0], np.uint8)
                if lib.rtrec_store_find_sorted(hay.ctypes.data, len(self), k.ctypes.data, k.shape[0],
                                               pos.ctypes.data, found.ctypes.data, 0) != 0:
                    raise RuntimeError("rtrec_store_find_sorted failed")
                return found.view(bool), pos
        pos = np.searchsorted(self.key, keys)
        pos_c = np.minimum(pos, len(self) - 1)
        return self.key[pos_c] == keys, pos_c


_NATIVE_MERGE_MIN = 1 << 15
_NATIVE_FIND_MIN = 1 << 14
_NATIVE_APPLY_MIN = 1 << 16
_DEVICE_FOLD_MIN = 1 << 18         # batches at least this long are reduced on the device when the caller offers one
_MAX_ROUNDS = 32                   # occurrences of one (user, item) pair per batch that are applied as vectorised rounds
_NATIVE_DECAY_MIN = 1
_native_merge: Any = None          # False: the native library is not available (numpy merge instead)
_native_handle: Any = None


def _native_lib() -> Any:
    """librtrec_amd.so (host routines of csrc/store_host.hip), or None when it has not been built."""
    global _native_handle
    if _native_handle is None:
        try:
            from .. import _native
            _native_handle = _native.load()
        except Exception:
            _native_handle = False
    return _native_handle or None


def _merge_native(old: _Block, new: _Block) -> Optional[_Block]:
    """Two-pointer merge by librtrec_amd.so's host routine (rtrec_store_merge_sorted: memory speed,
    threaded), or None when the library has not been built."""
    global _native_merge
    if _native_merge is None:
        try:
            from .. import _native
            _native_merge = _native.load().rtrec_store_merge_sorted
        except Exception:
            _native_merge = False
    if _native_merge is False:
        return None
    n, m = len(old), len(new)
    cols = [np.ascontiguousarray(a) for a in (old.key, old.val, old.ts, new.key, new.val, new.ts)]
    ko, vo, to = np.empty(n + m, np.int64), np.empty(n + m, np.float64), np.empty(n + m, np.float64)
    p = [a.ctypes.data for a in cols]
    cnt = int(_native_merge(p[0], p[1], p[2], n, p[3], p[4], p[5], m, ko.ctypes.data, vo.ctypes.data, to.ctypes.data, 0))
    if cnt < 0:
        raise RuntimeError("rtrec_store_merge_sorted failed")
    return _Block(ko[:cnt], vo[:cnt], to[:cnt]) if cnt == n + m else _Block(ko[:cnt].copy(), vo[:cnt].copy(), to[:cnt].copy())


def _merge_blocks(old: _Block, new: _Block) -> _Block:
    """Union of two sorted blocks; on equal keys the entry of `new` wins."""
    if len(old) == 0:
        return new
    if len(new) == 0:
        return old
    if len(old) + len(new) >= _NATIVE_MERGE_MIN:
        merged = _merge_native(old, new)
        if merged is not None:
            return merged
    found, pos = old.find(new.key)
    if found.any():   # overwrite in place, append the rest
        old.val[pos[found]] = new.val[found]
        old.ts[pos[found]] = new.ts[found]
        old.values_changed()
        rest = ~found
        if not rest.any():
            return old
        nk, nv, nt = new.key[rest], new.val[rest], new.ts[rest]
    else:
        nk, nv, nt = new.key, new.val, new.ts
    # two sorted runs with distinct keys: the destination of every new entry is its insertion point plus
    # its rank, the old entries fill the remaining slots in order -- one binary search of the smaller run
    # and three streaming scatters instead of a sort of n + m keys and three random gathers
    n, m = len(old), nk.shape[0]
    dst_new = np.searchsorted(old.key, nk) + np.arange(m, dtype=np.int64)
    is_new = np.zeros(n + m, dtype=bool)
    is_new[dst_new] = True
    is_old = ~is_new
    out = []
    for a_old, a_new in ((old.key, nk), (old.val, nv), (old.ts, nt)):
        o = np.empty(n + m, dtype=a_old.dtype)
        o[is_old] = a_old
        o[dst_new] = a_new
        out.append(o)
    return _Block(*out)


def _stable_order(users: np.ndarray, items: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(argsort(keys, kind="stable"), keys in that order).  When (user, item, arrival index) fit one int64
    the composite is VALUE-sorted instead (numpy's vectorised quicksort; the arrival index in the low bits
    both makes the keys distinct -- so an unstable sort yields the stable order -- and is the permutation),
    and the sorted keys are unpacked from the composite instead of gathered through the permutation."""
    n = users.shape[0]
    ib, xb = int(items.max()).bit_length(), max(1, (n - 1).bit_length())
    if n >= (1 << 12) and int(users.max()).bit_length() + ib + xb <= 63:
        comp = (users << (ib + xb)) | (items << xb) | np.arange(n, dtype=np.int64)
        comp.sort()
        order = comp & ((1 << xb) - 1)
        comp >>= xb
        sk = ((comp >> ib) << _SHIFT) | (comp & ((1 << ib) - 1))
        return order, sk
    keys = (users << _SHIFT) | items
    order = np.argsort(keys, kind="stable")
    return order, keys[order]


class UserItemInteractions:
    def __init__(self, min_value: int = -5, max_value: int = 10, decay_in_days: Optional[int] = None,
                 **kwargs: Any) -> None:
        assert max_value > min_value, f"max_value should be greater than min_value {max_value} > {min_value}"
        self.min_value = min_value
        self.max_value = max_value
        # half-life decay, "Time Weight Collaborative Filtering": rate = 1 - ln(2) / days
        self.decay_rate: Optional[float] = None if decay_in_days is None else 1.0 - (math.log(2) / decay_in_days)
        self.hot_items = LRUFreqSet(capacity=kwargs.get("n_recent_hot", 100_000))
        self.all_item_ids: set = set()
        self.max_user_id = 0
        self.max_item_id = 0
        self.max_timestamp = 0.0
        self._base = _Block()
        self._delta = _Block()
        self._l0 = _Block()
        self.version = 0   # bumped on every mutation; device mirrors key their caches on it

    def __setstate__(self, state: Dict[str, Any]) -> None:
        """Own pickles restore as they are; a REFERENCE pickle carries `interactions` as
        {user: {item: (value, tstamp)}} (interactions.py:27) and is converted to sorted columns."""
        nested = state.get("interactions") if isinstance(state.get("interactions"), dict) else None
        if nested is None:
            self.__dict__.update(state)
            self.__dict__.setdefault("_l0", _Block())
            return
        state = {k: v for k, v in state.items() if k != "interactions"}
        self.__dict__.update(state)
        users, items, vals, tss = [], [], [], []
        for u, inner in nested.items():
            for i, (v, t) in inner.items():
                users.append(u); items.append(i); vals.append(v); tss.append(t)
        key = self._keys(np.asarray(users, dtype=np.int64), np.asarray(items, dtype=np.int64))
        order = np.argsort(key, kind="stable")
        self._base = _Block(key[order], np.asarray(vals, dtype=np.float64)[order], np.asarray(tss, dtype=np.float64)[order])
        self._delta = _Block()
        self._l0 = _Block()
        self.version = 0

    # ------------------------------------------------------------------ decay
    def get_decay_rate(self) -> Optional[float]:
        return self.decay_rate

    def set_decay_rate(self, decay_rate: Optional[float]) -> None:
        self.decay_rate = decay_rate
        self.version += 1

    def _apply_decay(self, value: float, last_timestamp: float) -> float:
        if self.decay_rate is None:
            return value
        elapsed_days = (self.max_timestamp - last_timestamp) / 86400.0
        return value * self.decay_rate ** elapsed_days

    def _decay_array(self, val: np.ndarray, ts: np.ndarray, now: Any) -> np.ndarray:
        if self.decay_rate is None:
            return val
        lib = _native_lib() if val.shape[0] >= _NATIVE_DECAY_MIN else None
        if lib is not None:       # libm pow, threaded: the reference's own arithmetic (float ** float), bit for bit
            v, t = np.ascontiguousarray(val, dtype=np.float64), np.ascontiguousarray(ts, dtype=np.float64)
            now_arr = None if np.ndim(now) == 0 else np.ascontiguousarray(now, dtype=np.float64)
            out = np.empty(v.shape[0], np.float64)
            if lib.rtrec_store_decay(v.ctypes.data, t.ctypes.data, v.shape[0], float(self.decay_rate),
                                     now_arr.ctypes.data if now_arr is not None else None,
                                     0.0 if now_arr is not None else float(now), out.ctypes.data, None, 0) != 0:
                raise RuntimeError("rtrec_store_decay failed")
            return out
        return val * np.power(self.decay_rate, (now - ts) / 86400.0)

    # ------------------------------------------------------------------ lookup
    @staticmethod
    def _keys(users: np.ndarray, items: np.ndarray) -> np.ndarray:
        return (users.astype(np.int64) << _SHIFT) | items.astype(np.int64)

    def _lookup(self, keys: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(found, value, timestamp) per key; newer levels shadow older ones (l0 > delta > base)."""
        val = np.zeros(keys.shape, np.float64)
        ts = np.zeros(keys.shape, np.float64)
        fb, pb = self._base.find(keys)
        if fb.any():
            val[fb] = self._base.val[pb[fb]]
            ts[fb] = self._base.ts[pb[fb]]
        fd, pd_ = self._delta.find(keys)
        if fd.any():
            val[fd] = self._delta.val[pd_[fd]]
            ts[fd] = self._delta.ts[pd_[fd]]
        fl, pl = self._l0.find(keys)
        if fl.any():
            val[fl] = self._l0.val[pl[fl]]
            ts[fl] = self._l0.ts[pl[fl]]
        return fb | fd | fl, val, ts

    def _write(self, keys: np.ndarray, val: np.ndarray, ts: np.ndarray, presorted: bool = False) -> None:
        """Store unique keys (any order).  Three sorted levels, newest first: a small l0 that a
        mini-batch merges into (so a 1k-interaction write never rewrites a multi-million-entry block),
        the delta block l0 spills into, and the base block the delta is compacted into once it has
        grown to an eighth of it."""
        if presorted:
            blk = _Block(keys, np.asarray(val, dtype=np.float64), np.asarray(ts, dtype=np.float64))
        else:
            order = np.argsort(keys, kind="stable")
            blk = _Block(keys[order], val[order].astype(np.float64), ts[order].astype(np.float64))
        if len(blk) >= _L0_MAX // 2:          # a bulk chunk goes straight to the delta level (after what is newer than it)
            self._flush_l0(check=False)
            self._delta = _merge_blocks(self._delta, blk)
            self._check_delta()
        else:
            self._l0 = _merge_blocks(self._l0, blk)
            if len(self._l0) >= _L0_MAX:
                self._flush_l0()
        self.version += 1

    def _flush_l0(self, check: bool = True) -> None:
        """Fold the recent-writes level into the delta block (every reader that walks the blocks does
        this first; lookups by key do not need to)."""
        if len(self._l0):
            self._delta = _merge_blocks(self._delta, self._l0)
            self._l0 = _Block()
            if check:
                self._check_delta()

    def _check_delta(self) -> None:
        if len(self._delta) >= max(_DELTA_MERGE_MIN, len(self._base) // 8):
            self._compact()

    def _compact(self) -> _Block:
        self._flush_l0(check=False)
        if len(self._delta):
            self._base = _merge_blocks(self._base, self._delta)
            self._delta = _Block()
        return self._base

    # ------------------------------------------------------------------ ingest
    def add_interaction(self, user_id: int, item_id: int, tstamp: float, delta: float = 1.0,
                        upsert: bool = False) -> None:
        """One interaction: new = clip(decayed(old) + delta) or, with upsert, (delta, tstamp)."""
        # the reference computes `tstamp > now + 180.0` and `current + delta` with the caller's objects: a
        # numeric string raises TypeError there (numpy would parse it), and the caller skips the row
        tstamp, delta = 0.0 + tstamp, (delta if upsert else 0.0 + delta)
        self.add_interactions_batch(np.array([user_id], np.int64), np.array([item_id], np.int64),
                                    np.array([tstamp], np.float64), np.array([delta], np.float64), upsert=upsert)

    def add_interactions_batch(self, users: Sequence[int], items: Sequence[int], tstamps: Sequence[float],
                               deltas: Sequence[float], upsert: bool = False, device_fold: Any = None) -> None:
        """Apply interactions in order with exactly the sequential semantics of add_interaction
        (interactions.py:81-119), vectorised over the interactions that touch distinct pairs.

        device_fold (DeviceInteractions.ingest bound to a GPU backend) reduces a bulk batch to its distinct pairs on the
        device instead -- sort, run detection, per-pair fold, hot-item counts -- and this store writes the result; it is
        not used for stores with time decay (every step's current value depends on the running max_timestamp)."""
        users = np.asarray(users, dtype=np.int64)
        items = np.asarray(items, dtype=np.int64)
        ts = np.ascontiguousarray(tstamps, dtype=np.float64)
        dl = np.ascontiguousarray(deltas, dtype=np.float64)
        n = users.shape[0]
        if n == 0:
            return
        if users.min() < 0 or items.min() < 0 or users.max() > _MASK or items.max() > _MASK:
            raise ValueError("user and item ids must be in [0, 2**32)")
        now = time.time()
        late = ts > now + 180.0
        if late.any():
            logging.warning(f"{int(late.sum())} interaction timestamp(s) are in the future "
                            f"(max {float(ts.max())}, current time {now})")
        # max_timestamp as each interaction sees it (running max of tstamp + 1): only decay looks at it
        decaying = self.decay_rate is not None and not upsert
        seen = np.maximum.accumulate(np.concatenate(([self.max_timestamp], ts + 1.0)))[1:] if decaying else None
        max_ts_after = float(seen[-1]) if decaying else max(self.max_timestamp, float(ts.max()) + 1.0)

        if device_fold is not None and not decaying and n >= _DEVICE_FOLD_MIN:
            empty = self.is_empty
            res = device_fold(users, items, ts, dl, upsert, float(self.min_value), float(self.max_value),
                              None if empty else (lambda keys: self._lookup(keys)[1]))
            self._write(res["keys"], res["val"], res["ts"], presorted=True)
            self.max_timestamp = max_ts_after
            if res["items_present"] is not None:
                self.all_item_ids.update(res["items_present"].tolist())
            else:
                self.all_item_ids.update(np.unique(items).tolist())
            if res["hot"] is None or not self.hot_items.add_counted(*res["hot"]):
                pos = dl > 0
                if pos.any():
                    self.hot_items.add_many(items[pos])
            self.max_user_id = max(self.max_user_id, int(res["user_max"]))
            self.max_item_id = max(self.max_item_id, int(res["item_max"]))
            return

        order, sk = _stable_order(users, items)
        tail_idx = tail_key = None
        first = np.ones(n, bool)
        first[1:] = sk[1:] != sk[:-1]
        if first.all():
            rounds = [(order, sk)]    # distinct pairs: one round, visited in key order (already sorted for the store)
        else:   # occurrence rank of every interaction within its (user, item) group
            start = np.flatnonzero(first)
            grp = np.cumsum(first) - 1
            rank = np.arange(n) - start[grp]
            # Round r = the r-th occurrence of every pair, vectorised over the pairs.  The rounds are cut out of
            # ONE stable sort by rank (O(n log n) for the batch, not O(n) per round), and only the first
            # _MAX_ROUNDS of them are run this way: a pair repeated more often than that (a bot, a replayed
            # stream) has the rest of its sequence folded in one sequential pass below, so the cost of a
            # batch stays O(n) however often one pair repeats (the reference's loop is O(n) too).
            by_rank = np.argsort(rank, kind="stable")
            bounds = np.concatenate(([0], np.cumsum(np.bincount(rank))))
            n_rounds = min(len(bounds) - 1, _MAX_ROUNDS)
            rounds = [(order[by_rank[bounds[r]:bounds[r + 1]]], sk[by_rank[bounds[r]:bounds[r + 1]]])
                      for r in range(n_rounds)]
            if len(bounds) - 1 > n_rounds:
                tail = by_rank[bounds[n_rounds]:]
                tail.sort()                               # back to (key, arrival) order
                tail_idx, tail_key = order[tail], sk[tail]
        lib = _native_lib() if (not decaying and n >= _NATIVE_APPLY_MIN) else None
        for idx, k in rounds:
            if lib is not None:       # gather + add + clip in one threaded pass (rtrec_store_apply_round)
                old = None if upsert else np.ascontiguousarray(self._lookup(k)[1])
                idx_c = np.ascontiguousarray(idx, dtype=np.int64)
                new, ts_k = np.empty(len(idx_c), np.float64), np.empty(len(idx_c), np.float64)
                if lib.rtrec_store_apply_round(idx_c.ctypes.data, len(idx_c), dl.ctypes.data, ts.ctypes.data,
                                               old.ctypes.data if old is not None else None, float(self.min_value),
                                               float(self.max_value), new.ctypes.data, ts_k.ctypes.data, 0) != 0:
                    raise RuntimeError("rtrec_store_apply_round failed")
                self._write(k, new, ts_k, presorted=True)
                continue
            if upsert:
                new = dl[idx]
            else:
                found, old, old_ts = self._lookup(k)
                if decaying:
                    cur = np.where(found & (old != 0.0), self._decay_array(old, old_ts, seen[idx]), 0.0)
                else:
                    cur = old                 # 0.0 where the pair is new
                new = cur + dl[idx]       # max(lo, min(new, hi)) with Python's min / max (interactions.py:106): a NaN ends as lo
                new = np.where(self.max_value < new, float(self.max_value), new)
                new = np.where(new > self.min_value, new, float(self.min_value))
            self._write(k, new, ts[idx], presorted=True)     # every round is a subsequence of the key-sorted order
        if tail_idx is not None:
            self._fold_tail(tail_idx, tail_key, ts, dl, seen, upsert)

        self.max_timestamp = max_ts_after
        i_hi = int(items.max())
        if i_hi <= max(4 * n, 1 << 22):
            self.all_item_ids.update(np.flatnonzero(np.bincount(items, minlength=i_hi + 1)).tolist())
        else:
            self.all_item_ids.update(np.unique(items).tolist())
        pos = dl > 0
        if pos.any():
            self.hot_items.add_many(items[pos])
        self.max_user_id = max(self.max_user_id, int(users.max()))
        self.max_item_id = max(self.max_item_id, int(items.max()))

    def _fold_tail(self, idx: np.ndarray, keys: np.ndarray, ts: np.ndarray, dl: np.ndarray,
                   seen: Optional[np.ndarray], upsert: bool) -> None:
        """Occurrences beyond the vectorised rounds, in (key, arrival) order: every pair's remaining sequence is
        folded one interaction after the other with the reference's own scalar arithmetic (Python floats,
        interactions.py:62-79,103-112), then the pairs are written once."""
        cut = np.flatnonzero(np.concatenate(([True], keys[1:] != keys[:-1], [True])))
        ukeys = keys[cut[:-1]]
        if upsert:                       # the last occurrence wins
            last = idx[cut[1:] - 1]
            self._write(ukeys, dl[last], ts[last], presorted=True)
            return
        _, old, old_ts = self._lookup(ukeys)     # every such pair exists: its earlier occurrences were just stored
        lo, hi, rate = self.min_value, self.max_value, self.decay_rate
        dl_l, ts_l = dl[idx].tolist(), ts[idx].tolist()
        seen_l = seen[idx].tolist() if seen is not None else None
        out_v, out_t = old.tolist(), old_ts.tolist()
        cuts = cut.tolist()
        for g in range(len(ukeys)):
            v, t = out_v[g], out_t[g]
            for q in range(cuts[g], cuts[g + 1]):
                if seen_l is not None and v != 0.0:
                    v = v * rate ** ((seen_l[q] - t) / 86400.0)
                v = max(lo, min(v + dl_l[q], hi))
                t = ts_l[q]
            out_v[g], out_t[g] = v, t
        self._write(ukeys, np.asarray(out_v, np.float64), np.asarray(out_t, np.float64), presorted=True)

    # ------------------------------------------------------------------ queries
    def _user_entries(self, user_id: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        self._flush_l0()
        sb = self._select(self._base, np.array([user_id], np.int64), None)
        sd = self._select(self._delta, np.array([user_id], np.int64), None)
        if len(sd) == 0:
            return self._base.key[sb] & _MASK, self._base.val[sb], self._base.ts[sb]
        key = np.concatenate([self._delta.key[sd], self._base.key[sb]])
        val = np.concatenate([self._delta.val[sd], self._base.val[sb]])
        ts = np.concatenate([self._delta.ts[sd], self._base.ts[sb]])
        order = np.argsort(key, kind="stable")
        key, val, ts = key[order], val[order], ts[order]
        keep = np.ones(len(key), bool)
        keep[1:] = key[1:] != key[:-1]
        return key[keep] & _MASK, val[keep], ts[keep]

    def has_interaction(self, user_id: int, item_id: int) -> bool:
        found, _, _ = self._lookup(self._keys(np.array([user_id]), np.array([item_id])))
        return bool(found[0])

    def get_user_item_rating(self, user_id: int, item_id: int, default_rating: float = 0.0) -> float:
        found, val, ts = self._lookup(self._keys(np.array([user_id]), np.array([item_id])))
        if not found[0] or val[0] == default_rating:
            return default_rating
        return float(self._apply_decay(float(val[0]), float(ts[0])))

    def get_user_items(self, user_id: int, n_recent: Optional[int] = None) -> List[int]:
        items, _, ts = self._user_entries(user_id)
        if len(items) == 0:
            return []
        if n_recent is not None and self.n_users_seen > n_recent:
            order = np.argsort(-ts, kind="stable")
            return items[order][:n_recent].tolist()
        return items.tolist()

    @property
    def n_users_seen(self) -> int:
        blk = self._compact()
        if len(blk) == 0:
            return 0
        u = blk.key >> _SHIFT
        return int(1 + np.count_nonzero(u[1:] != u[:-1]))

    @property
    def nnz(self) -> int:
        return len(self._compact())

    @property
    def is_empty(self) -> bool:
        return len(self._base) + len(self._delta) + len(self._l0) == 0

    def get_all_item_ids(self) -> List[int]:
        return list(self.all_item_ids)

    def get_all_users(self) -> List[int]:
        blk = self._compact()
        return np.unique(blk.key >> _SHIFT).tolist()

    def get_all_non_interacted_items(self, user_id: int) -> List[int]:
        interacted = self.get_user_items(user_id)
        if len(interacted) == 0:
            return list(self.all_item_ids)
        return list(self.all_item_ids.difference(interacted))

    def get_all_non_negative_items(self, user_id: int) -> List[int]:
        items, val, ts = self._user_entries(user_id)
        rated = dict(zip(items.tolist(), np.where(val != 0.0, self._decay_array(val, ts, self.max_timestamp), 0.0).tolist()))
        return [i for i in self.all_item_ids if rated.get(i, 0.0) >= 0.0]

    def get_hot_items(self, n: Optional[int] = None, user_id: Optional[int] = None,
                      filter_interacted: bool = True) -> List[int]:
        interacted: List[int] = []
        if filter_interacted:
            assert user_id is not None, "User ID must be provided to filter interacted items."
            interacted = self.get_user_items(user_id)
        return list(self.hot_items.get_freq_items(n, exclude_items=interacted))

    def get_users_by_items(self, item_ids: List[int]) -> List[int]:
        rows, _, _ = self._triples(select_items=list(item_ids), weights=False)
        return np.unique(rows).tolist()

    # ------------------------------------------------------------------ export
    @staticmethod
    def _ranges(lo: np.ndarray, hi: np.ndarray) -> np.ndarray:
        """Concatenation of the index ranges [lo_k, hi_k)."""
        cnt = hi - lo
        total = int(cnt.sum())
        if total == 0:
            return np.empty(0, np.int64)
        return np.repeat(lo - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) + np.arange(total)

    def _select(self, blk: _Block, users: Optional[np.ndarray], items: Optional[np.ndarray]) -> np.ndarray:
        """Positions of a block's entries restricted to the given users and/or items, without
        scanning the block: user ranges come from the sorted keys, item ranges from the item index."""
        if len(blk) == 0:
            return np.empty(0, np.int64)
        if users is not None:
            sel = self._ranges(np.searchsorted(blk.key, users << _SHIFT), np.searchsorted(blk.key, (users + 1) << _SHIFT))
            if items is not None:
                sel = sel[np.isin(blk.key[sel] & _MASK, items)]
            return sel
        if items is not None:
            if len(blk) < (1 << 15):
                return np.flatnonzero(np.isin(blk.key & _MASK, items))
            perm, iptr = blk.item_index(self.max_item_id + 1)
            it = items[items <= self.max_item_id]
            return np.sort(perm[self._ranges(iptr[it], iptr[it + 1])])
        return np.arange(len(blk))

    def _triples(self, select_users: Optional[Sequence[int]] = None, select_items: Optional[Sequence[int]] = None,
                 weights: bool = True) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(rows, cols, float64 decayed values) in key order (user-major, item ascending).
        Selections read only the selected rows / columns of the base and delta blocks, so a
        mini-batch export costs O(selected entries), not O(all interactions)."""
        users = None if select_users is None else np.unique(np.asarray(list(select_users), dtype=np.int64))
        items = None if select_items is None else np.unique(np.asarray(list(select_items), dtype=np.int64))
        if users is None and items is None:
            blk = self._compact()
            key, val, ts = blk.key, blk.val, blk.ts
        else:
            self._flush_l0()
            sb = self._select(self._base, users, items)
            sd = self._select(self._delta, users, items)
            key = np.concatenate([self._delta.key[sd], self._base.key[sb]])     # delta first: it wins
            val = np.concatenate([self._delta.val[sd], self._base.val[sb]])
            ts = np.concatenate([self._delta.ts[sd], self._base.ts[sb]])
            if len(sd):
                order = np.argsort(key, kind="stable")
                key, val, ts = key[order], val[order], ts[order]
                keep = np.ones(len(key), bool)
                keep[1:] = key[1:] != key[:-1]
                key, val, ts = key[keep], val[keep], ts[keep]
        data = self._decay_array(val, ts, self.max_timestamp) if weights else val
        return key >> _SHIFT, key & _MASK, data

    def to_csr(self, select_users: Optional[List[int]] = None, include_weights: bool = True) -> csr_matrix:
        """U x I CSR; with select_users only those rows are populated (interactions.py:259-289).
        NOTE the reference treats an EMPTY select_users list like None (`if select_users:`)."""
        sel = select_users if select_users else None
        rows, cols, data = self._triples(select_users=sel, weights=include_weights)
        n_u, n_i = self.shape
        indptr = np.zeros(n_u + 1, dtype=np.int64)
        indptr[1:] = np.bincount(rows, minlength=n_u)
        np.cumsum(indptr, out=indptr)
        if include_weights:
            m = csr_matrix((data.astype(np.float32), cols.astype(np.int32), indptr.astype(np.int32)),
                           shape=(n_u, n_i), dtype=np.float32)
        else:
            m = csr_matrix((np.ones(len(rows), dtype=np.int32), cols.astype(np.int32), indptr.astype(np.int32)),
                           shape=(n_u, n_i), dtype=np.int32)
        m.has_sorted_indices = True
        return m

    _ITEM_MAJOR_MIN = 1 << 15      # smaller base blocks are exported through the generic path

    def _csc_selected(self, items: np.ndarray) -> csc_matrix:
        """CSC with only the columns `items` (sorted, unique) populated -- the matrix a mini-batch refit
        works on (slim.py:33-36).  The base block's item-major mirror yields the selected columns as
        contiguous slices; the (small) delta block's entries for those items then overwrite their base
        entry or are inserted at their row position.  Same result as the generic path, ~6x less work."""
        n_u, n_i = self.shape
        static = self.decay_rate is None
        self._flush_l0()
        iptr, rows_im, val_im, ts_im, d32_im = self._base.item_major(n_i, static)
        it = items[items < n_i]
        lo, hi = iptr[it], iptr[it + 1]
        cnt = hi - lo
        sel = self._ranges(lo, hi)                       # contiguous runs: a streaming copy
        rows = rows_im[sel]
        if static:
            data = d32_im[sel]
        else:
            data = self._decay_array(val_im[sel], ts_im[sel], self.max_timestamp).astype(np.float32)
        counts = np.zeros(n_i, dtype=np.int64)
        counts[it] = cnt
        sd = self._select(self._delta, None, it)
        if len(sd):
            kd = self._delta.key[sd]
            it_d, us_d = kd & _MASK, kd >> _SHIFT
            od = np.argsort((it_d << _SHIFT) | us_d, kind="stable")      # item-major, users ascending
            it_d, us_d = it_d[od], us_d[od]
            dat_d = self._decay_array(self._delta.val[sd][od], self._delta.ts[sd][od], self.max_timestamp).astype(np.float32)
            start = np.concatenate(([0], np.cumsum(cnt)))               # slice of each selected item in `rows`
            slot = np.searchsorted(it, it_d)                             # which selected item each delta entry belongs to
            grp = np.flatnonzero(np.concatenate(([True], it_d[1:] != it_d[:-1], [True])))
            pos = np.empty(len(it_d), dtype=np.int64)
            us32 = us_d.astype(np.int32)
            for g in range(len(grp) - 1):                                # one search per item that has delta entries
                a, b = grp[g], grp[g + 1]
                s0, s1 = start[slot[a]], start[slot[a] + 1]
                pos[a:b] = s0 + np.searchsorted(rows[s0:s1], us32[a:b])
            end = start[slot + 1]
            found = pos < end
            found[found] = rows[pos[found]] == us32[found]
            data[pos[found]] = dat_d[found]                              # the delta entry shadows the base entry
            new = ~found
            if new.any():
                rows = np.insert(rows, pos[new], us32[new])
                data = np.insert(data, pos[new], dat_d[new])
                counts += np.bincount(it_d[new], minlength=n_i)
        indptr = np.zeros(n_i + 1, dtype=np.int64)
        np.cumsum(counts, out=indptr[1:])
        m = csc_matrix((data, rows, indptr.astype(np.int32)), shape=(n_u, n_i), dtype=np.float32)
        m.has_sorted_indices = True
        return m

    def to_csc(self, select_items: Optional[List[int]] = None) -> csc_matrix:
        """U x I CSC; with select_items only those columns are populated (interactions.py:291-303)."""
        if select_items is not None and len(self._base) >= self._ITEM_MAJOR_MIN:
            return self._csc_selected(np.unique(np.asarray(list(select_items), dtype=np.int64)))
        rows, cols, data = self._triples(select_items=select_items)
        n_u, n_i = self.shape
        # rows stay ascending inside each column; 16-bit keys take numpy's radix sort (several times
        # faster than the merge sort used for wider integers)
        n_e = cols.shape[0]
        xb = max(1, (n_e - 1).bit_length())
        if n_i <= 65536:
            order = np.argsort(cols.astype(np.uint16), kind="stable")
        elif n_e >= (1 << 12) and int(n_i).bit_length() + xb <= 63:     # value sort of (item, position)
            comp = (cols.astype(np.int64) << xb) | np.arange(n_e, dtype=np.int64)
            comp.sort()
            order = comp & ((1 << xb) - 1)
        else:
            order = np.argsort(cols, kind="stable")
        indptr = np.zeros(n_i + 1, dtype=np.int64)
        indptr[1:] = np.bincount(cols, minlength=n_i)
        np.cumsum(indptr, out=indptr)
        m = csc_matrix((data[order].astype(np.float32), rows[order].astype(np.int32), indptr.astype(np.int32)),
                       shape=(n_u, n_i), dtype=np.float32)
        m.has_sorted_indices = True
        return m

    def to_coo(self, select_users: Optional[List[int]] = None, select_items: Optional[List[int]] = None) -> coo_matrix:
        rows, cols, data = self._triples(select_users=select_users, select_items=select_items)
        return coo_matrix((data, (rows, cols)), shape=self.shape, dtype="float32")

    @property
    def shape(self) -> Tuple[int, int]:
        return self.max_user_id + 1, self.max_item_id + 1

    # the reference exposes the nested dict; materialise it on demand for introspection only
    @property
    def interactions(self) -> Dict[int, Dict[int, Tuple[float, float]]]:
        blk = self._compact()
        out: Dict[int, Dict[int, Tuple[float, float]]] = {}
        for k, v, t in zip(blk.key.tolist(), blk.val.tolist(), blk.ts.tolist()):
            out.setdefault(k >> _SHIFT, {})[k & _MASK] = (v, t)
        return out
