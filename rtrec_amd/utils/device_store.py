"""Device-resident mirror of the interaction matrix X.

The host store (rtrec_amd/utils/interactions.py) owns the reference's semantics -- additive update
with clipping, upsert, decay, hot items (/root/reference/rtrec/utils/interactions.py:81-119).  What
the GPU path needs from it per call is X as CSR + CSC arrays, and for a streaming mini-batch
(SLIM.fit, /root/reference/rtrec/models/slim.py:29-43) the matrix with ONLY the touched items'
columns populated.  Exporting those from the host costs O(selected entries) of numpy work plus a
PCIe upload per mini-batch (C3: ~200 ms + 25 ms for 1000 interactions).

This mirror keeps X in HBM as two sorted COO copies -- row-major keys (user << 32 | item) and
column-major keys (item << 32 | user), each with its float32 values -- and applies a mini-batch as a
merge of its <= batch-size changed pairs (values computed by the host store, so clip / upsert /
sequential-duplicate semantics stay in one place).  CSR / CSC pointers are binary searches over the
keys, and the touched-columns matrix is a gather of contiguous column slices plus a mask-compaction
of the row-major copy: a few passes over HBM, no host export, no upload.

Stores WITH time decay (every value is `raw * rate ** ((max_timestamp - tstamp) / 86400)` in float64,
rtrec/utils/interactions.py:62-79) keep the raw float64 values and timestamps resident as well, and the
float32 matrix is re-evaluated on the device whenever max_timestamp has moved -- one pass of the hand-written
decay kernel (csrc/store_device.hip) whose float32 results are provably the reference's except for the few
entries it flags as too close to a rounding boundary, which the host re-evaluates with libm and patches.

PyTorch tensor ops are the plumbing here (sort / searchsorted / scatter on the resident arrays); the
arrays feed the hand-written fit and score kernels through the same C-ABI as host-built matrices.
The class is device-agnostic (tests run it on CPU tensors against the host exports).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np

_SHIFT = 32
_MASK = (1 << _SHIFT) - 1


class DeviceInteractions:
    def __init__(self, torch: Any, device: Any):
        self.torch = torch
        self.device = device
        self.version: Optional[int] = None      # host store version the mirror equals; None = never built
        self.n_users = 0
        self.n_items = 0
        self._rk = self._rv = self._ck = self._cv = None      # sorted keys + values, both orientations
        self._full: Optional[Dict[str, Any]] = None
        # time decay: raw float64 values and timestamps in both orientations, the rate, the max_timestamp the float32
        # values were last evaluated at (None = stale) and the max_timestamp they are wanted at
        self.rate: Optional[float] = None
        self._rraw = self._rts = self._craw = self._cts = None
        self._now: Optional[float] = None
        self._valued_at: Optional[float] = None
        self.decay_fn: Any = None           # (raw, ts, rate, now) -> float32 tensor; the GPU backend installs the HIP kernel
        self.ingested: Optional[Dict[str, Any]] = None      # the block the last ingest() folded, still on the device

    # ------------------------------------------------------------------ build
    def _dev(self, a: Any, dtype: Any = None):
        """A host array uploaded (cast to `dtype` first) -- or a tensor that already lives on the device, as it is."""
        if self.torch.is_tensor(a):
            want = None if dtype is None else getattr(self.torch, np.dtype(dtype).name)
            return a.to(self.device) if want is None or a.dtype == want else a.to(self.device, want)
        return self.torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(self.device)

    def load_csr(self, indptr: np.ndarray, indices: np.ndarray, data: np.ndarray, n_users: int, n_items: int,
                 version: int) -> None:
        """(Re)build from a host CSR export (sorted indices); the column-major copy is sorted on the device."""
        torch = self.torch
        rptr = self._dev(np.asarray(indptr, dtype=np.int64))
        rcol = self._dev(np.asarray(indices, dtype=np.int64))
        self._rv = self._dev(np.asarray(data, dtype=np.float32))
        rows = torch.repeat_interleave(torch.arange(n_users, device=self.device, dtype=torch.int64), rptr[1:] - rptr[:-1])
        self._rk = (rows << _SHIFT) | rcol
        ck = (rcol << _SHIFT) | rows
        self._ck, order = torch.sort(ck)            # keys are distinct: any sort yields the CSC order
        self._cv = self._rv[order]
        self.rate = self._rraw = self._rts = self._craw = self._cts = self._now = self._valued_at = None
        self.n_users, self.n_items, self.version, self._full = int(n_users), int(n_items), version, None

    def load_csc(self, indptr: np.ndarray, indices: np.ndarray, data: np.ndarray, n_users: int, n_items: int,
                 version: int) -> None:
        """(Re)build from a host CSC export (sorted indices); the row-major copy is sorted on the device --
        instead of a scipy tocsr() on the host and a second upload."""
        torch = self.torch
        cptr = self._dev(np.asarray(indptr, dtype=np.int64))
        crow = self._dev(np.asarray(indices, dtype=np.int64))
        self._cv = self._dev(np.asarray(data, dtype=np.float32))
        cols = torch.repeat_interleave(torch.arange(n_items, device=self.device, dtype=torch.int64), cptr[1:] - cptr[:-1])
        self._ck = (cols << _SHIFT) | crow
        rk = (crow << _SHIFT) | cols
        self._rk, order = torch.sort(rk)
        self._rv = self._cv[order]
        self.rate = self._rraw = self._rts = self._craw = self._cts = self._now = self._valued_at = None
        self.n_users, self.n_items, self.version, self._full = int(n_users), int(n_items), version, None

    def load_store(self, keys: np.ndarray, raw: np.ndarray, ts: Optional[np.ndarray], n_users: int, n_items: int,
                   version: Any, rate: Optional[float] = None, now: Optional[float] = None) -> None:
        """(Re)build from the host store's compacted block: row-major keys (user << 32 | item, ascending), the RAW
        stored values and -- for a store with time decay (rate, now = max_timestamp) -- the timestamps.  No host
        export: the values are cast (or decayed) on the device, the column-major copy is a device sort."""
        torch = self.torch
        self._rk = self._dev(keys, np.int64)
        ck = ((self._rk & _MASK) << _SHIFT) | (self._rk >> _SHIFT)
        self._ck, order = torch.sort(ck)
        self.rate = None if rate is None else float(rate)
        if self.rate is None:
            self._rv = self._dev(raw, np.float32)
            self._cv = self._rv[order]
            self._rraw = self._rts = self._craw = self._cts = None
            self._now = self._valued_at = None
        else:
            self._rraw, self._rts = self._dev(raw, np.float64), self._dev(ts, np.float64)
            self._craw, self._cts = self._rraw[order], self._rts[order]
            self._rv = self._cv = None
            self._now, self._valued_at = float(now), None
        self.n_users, self.n_items, self.version, self._full = int(n_users), int(n_items), version, None

    def _host_decay(self, raw, ts, rate: float, now: float):
        """decay_fn fallback (CPU tensors, tests): libm pow through the host routine, the reference's own arithmetic."""
        from .interactions import _native_lib
        lib = _native_lib()
        v, t = raw.cpu().numpy(), ts.cpu().numpy()
        out = np.empty(v.shape[0], np.float32)
        if lib is not None and v.shape[0]:
            if lib.rtrec_store_decay(v.ctypes.data, t.ctypes.data, v.shape[0], float(rate), None, float(now), None,
                                     out.ctypes.data, 0) != 0:
                raise RuntimeError("rtrec_store_decay failed")
        elif v.shape[0]:
            out = (v * rate ** ((now - t) / 86400.0)).astype(np.float32)
        return self.torch.from_numpy(out).to(self.device)

    def _revalue(self) -> None:
        """Time decay: bring the float32 values of both orientations to the current max_timestamp."""
        if self.rate is None or self._valued_at == self._now:
            return
        fn = self.decay_fn or self._host_decay
        self._rv = fn(self._rraw, self._rts, self.rate, self._now)
        self._cv = fn(self._craw, self._cts, self.rate, self._now)
        self._valued_at = self._now
        self._full = None

    def adopt(self, X: Dict[str, Any], n_users: int, n_items: int, version: int) -> None:
        """Take over a full X the engine has already uploaded (bulk_fit): no host work at all."""
        torch = self.torch
        i64 = torch.int64
        rptr, cptr = X["rptr"].to(i64), X["cptr"].to(i64)
        rows = torch.repeat_interleave(torch.arange(n_users, device=self.device, dtype=i64), rptr[1:] - rptr[:-1])
        cols = torch.repeat_interleave(torch.arange(n_items, device=self.device, dtype=i64), cptr[1:] - cptr[:-1])
        self._rk, self._rv = (rows << _SHIFT) | X["rcol"].to(i64), X["rval"].clone()
        self._ck, self._cv = (cols << _SHIFT) | X["crow"].to(i64), X["cval"].clone()
        self.rate = self._rraw = self._rts = self._craw = self._cts = self._now = self._valued_at = None
        self.n_users, self.n_items, self.version, self._full = int(n_users), int(n_items), version, None

    def warm_up(self, fold_fn: Any = None) -> None:
        """Run every tensor op of this class once on a four-entry matrix: the first use of a sort /
        searchsorted / scatter kernel loads its code object (~0.1-0.3 s per process), which belongs to
        backend construction, not to the first bulk_fit or mini-batch."""
        self.load_csr(np.array([0, 2, 3]), np.array([0, 1, 1]), np.ones(3, np.float32), 2, 2, 0)
        self.apply(np.array([0, 1]), np.array([1, 0]), np.ones(2, np.float32), 2, 2, 1)
        self.adopt(self.full(), 2, 2, 2)
        self.partial(np.array([1]))
        self.load_csc(np.array([0, 1, 3]), np.array([0, 0, 1]), np.ones(3, np.float32), 2, 2, 3)
        self.load_store(np.array([0, 1, (1 << 32) | 1]), np.ones(3), np.zeros(3), 2, 2, 4, rate=0.99, now=86400.0)
        self.apply(np.array([0]), np.array([1]), np.ones(1), 2, 2, 5, tstamps=np.zeros(1), now=2 * 86400.0)
        self.full()
        if fold_fn is not None:
            for wide in (1 << 31, 0):       # both sort keys of ingest()
                self.ingest(np.array([0, 1, 0]) + wide, np.array([1, 0, 1]) + wide, np.zeros(3), np.ones(3), False, 0.0, 5.0,
                            lambda keys: np.zeros(len(keys)), fold_fn)
            self.adopt_ingested(2, 2, 6)

    # ------------------------------------------------------------------ update
    def _merge(self, keys, vals, new_k, new_v):
        """Write (new_k, new_v) -- distinct keys, any order -- into the sorted (keys, vals): existing keys
        are overwritten in place, new ones are inserted at their sorted position.  `vals` / `new_v` are lists of
        value arrays that travel with the keys (one float32 array, or raw values + timestamps)."""
        torch = self.torch
        new_k, order = torch.sort(new_k)
        new_v = [v[order] for v in new_v]
        n = keys.shape[0]
        pos = torch.searchsorted(keys, new_k)
        if n:
            hit = keys[pos.clamp(max=n - 1)] == new_k
            for v, nv in zip(vals, new_v):
                v[pos[hit]] = nv[hit]
        else:
            hit = torch.zeros_like(new_k, dtype=torch.bool)
        miss = ~hit
        m = int(miss.sum())
        if m == 0:
            return keys, vals
        ins_k, ins_pos = new_k[miss], pos[miss]
        out_k = torch.empty(n + m, dtype=keys.dtype, device=keys.device)
        dst_new = ins_pos + torch.arange(m, device=keys.device, dtype=ins_pos.dtype)
        dst_old = torch.arange(n, device=keys.device, dtype=torch.int64) + torch.searchsorted(ins_k, keys)
        out_k[dst_old] = keys
        out_k[dst_new] = ins_k
        out_vals = []
        for v, nv in zip(vals, new_v):
            out_v = torch.empty(n + m, dtype=v.dtype, device=v.device)
            out_v[dst_old] = v
            out_v[dst_new] = nv[miss]
            out_vals.append(out_v)
        return out_k, out_vals

    def apply(self, users: np.ndarray, items: np.ndarray, values: np.ndarray, n_users: int, n_items: int,
              version: Any, tstamps: Optional[np.ndarray] = None, now: Optional[float] = None) -> None:
        """Set X[users[k], items[k]] = values[k] for distinct pairs (the state the host store holds after a
        mini-batch) and grow the shape to (n_users, n_items).  With time decay `values` are the RAW stored values,
        `tstamps` their timestamps and `now` the store's max_timestamp after the batch."""
        u = self._dev(users, np.int64)
        i = self._dev(items, np.int64)
        if self.rate is None:
            v = self._dev(values, np.float32)
            if u.shape[0]:
                self._rk, (self._rv,) = self._merge(self._rk, [self._rv], (u << _SHIFT) | i, [v])
                self._ck, (self._cv,) = self._merge(self._ck, [self._cv], (i << _SHIFT) | u, [v])
        else:
            v = self._dev(values, np.float64)
            t = self._dev(tstamps, np.float64)
            if u.shape[0]:
                self._rk, (self._rraw, self._rts) = self._merge(self._rk, [self._rraw, self._rts], (u << _SHIFT) | i, [v, t])
                self._ck, (self._craw, self._cts) = self._merge(self._ck, [self._craw, self._cts], (i << _SHIFT) | u, [v, t])
            self._now, self._valued_at = float(now), None      # every value is a function of max_timestamp
        self.n_users, self.n_items, self.version, self._full = int(n_users), int(n_items), version, None

    # ------------------------------------------------------------------ bulk ingest
    def ingest(self, users: np.ndarray, items: np.ndarray, tstamps: np.ndarray, deltas: np.ndarray, upsert: bool,
               lo: float, hi: float, lookup: Any, fold_fn: Any) -> Dict[str, Any]:
        """A DataFrame-sized batch of interactions reduced to its distinct (user, item) pairs ON THE DEVICE, with the
        sequential semantics of one add_interaction per row (rtrec/utils/interactions.py:81-119, no time decay):
        the batch is uploaded once, sorted by (user, item, arrival) -- a value sort of the composite when it fits 63 bits,
        like the host store's _stable_order -- cut into runs of equal pairs, and every run is folded in arrival order by
        fold_fn (HipBackend.fold_pairs -> rtrec_store_fold_device) starting from the stored value `lookup(keys)` returns
        (None: the store is empty).  Also counted on the device: the hot-item statistics (LRUFreqSet.add per positive
        delta, interactions.py:115-116) as (item, hits, last arrival) triples and the set of items present.

        Returns host arrays for the store (which stays the owner of the state) -- keys (ascending), val, ts, items_present,
        hot = (items, hits, last_pos) or None -- and keeps the folded block on the device (self.ingested) so that a mirror of
        a previously empty store can adopt it without an upload."""
        torch = self.torch
        n = int(len(users))
        u, i = self._dev(users, np.int64), self._dev(items, np.int64)
        t, d = self._dev(tstamps, np.float64), self._dev(deltas, np.float64)
        u_hi, i_hi = int(u.max()), int(i.max())
        ib, xb = i_hi.bit_length(), max(1, (n - 1).bit_length())
        if u_hi.bit_length() + ib + xb <= 63:
            comp = (u << (ib + xb)) | (i << xb) | torch.arange(n, dtype=torch.int64, device=self.device)
            comp = torch.sort(comp)[0]          # distinct composites: the value sort IS the stable order
            order = comp & ((1 << xb) - 1)
            comp >>= xb
            sk = ((comp >> ib) << _SHIFT) | (comp & ((1 << ib) - 1))
        else:
            sk, order = torch.sort((u << _SHIFT) | i, stable=True)
        del u
        first = torch.ones(n, dtype=torch.bool, device=self.device)
        first[1:] = sk[1:] != sk[:-1]
        begin = torch.nonzero(first).reshape(-1)
        start = torch.cat([begin, torch.tensor([n], dtype=torch.int64, device=self.device)])
        uk = sk[begin]
        keys_h = uk.cpu().numpy()
        old = None
        if not upsert and lookup is not None:
            old = self._dev(lookup(keys_h), np.float64)
        val, ts_k, v32 = fold_fn(order, start, d, t, old, float(lo), float(hi), bool(upsert))
        out: Dict[str, Any] = {"keys": keys_h, "user_max": u_hi, "item_max": i_hi, "hot": None, "items_present": None}
        if i_hi <= max(4 * n, 1 << 22):         # dense id range: per-item counts by scatter (the host store's own criterion)
            pos = d > 0
            arrival = torch.arange(n, dtype=torch.int64, device=self.device)
            hits = torch.bincount(i[pos], minlength=i_hi + 1)
            last = torch.full((i_hi + 1,), -1, dtype=torch.int64, device=self.device)
            last.scatter_reduce_(0, i[pos], arrival[pos], "amax")
            seen = torch.bincount(i, minlength=i_hi + 1) > 0
            hot_items = torch.nonzero(hits).reshape(-1)
            out["hot"] = (hot_items.cpu().numpy(), hits[hot_items].cpu().numpy(), last[hot_items].cpu().numpy())
            out["items_present"] = torch.nonzero(seen).reshape(-1).cpu().numpy()
        out["val"], out["ts"] = val.cpu().numpy(), ts_k.cpu().numpy()
        self.ingested = {"keys": uk, "val32": v32}
        return out

    def adopt_ingested(self, n_users: int, n_items: int, version: Any, merge: bool = False) -> bool:
        """Take over the block the last ingest() folded (stores without time decay): as the whole matrix when the store
        was empty before it, or (merge) as changed pairs of a mirror that was in step with the store."""
        blk, self.ingested = self.ingested, None
        if blk is None:
            return False
        if merge:
            self.apply(blk["keys"] >> _SHIFT, blk["keys"] & _MASK, blk["val32"], n_users, n_items, version)
        else:
            self.load_store(blk["keys"], blk["val32"], None, n_users, n_items, version)
        return True

    # ------------------------------------------------------------------ views
    @property
    def nnz(self) -> int:
        return 0 if self._rk is None else int(self._rk.shape[0])

    def _pointers(self, keys, n: int):
        torch = self.torch
        bounds = torch.arange(n + 1, device=self.device, dtype=torch.int64) << _SHIFT
        return torch.searchsorted(keys, bounds).to(torch.int32)

    def full(self) -> Dict[str, Any]:
        """X as the engine's array set (rptr/rcol/rval + cptr/crow/cval [+ col_nnz, nonneg])."""
        self._revalue()
        if self._full is None:
            torch = self.torch
            cptr = self._pointers(self._ck, self.n_items)
            self._full = {
                "rptr": self._pointers(self._rk, self.n_users), "rcol": (self._rk & _MASK).to(torch.int32), "rval": self._rv,
                "cptr": cptr, "crow": (self._ck & _MASK).to(torch.int32), "cval": self._cv,
                "col_nnz": np.diff(cptr.cpu().numpy().astype(np.int64)),
                "nonneg": bool(self.nnz == 0 or float(self._cv.min()) >= 0.0),
            }
        return self._full

    def partial(self, items: np.ndarray) -> Dict[str, Any]:
        """X with only the columns `items` populated (same shape): what a mini-batch refit works on."""
        torch = self.torch
        F = self.full()
        i64 = torch.int64
        it = self._dev(np.unique(np.asarray(items, dtype=np.int64)))
        it = it[it < self.n_items]
        cptr = F["cptr"].to(i64)
        lo, cnt = cptr[it], cptr[it + 1] - cptr[it]
        total = int(cnt.sum())
        out_start = torch.cumsum(cnt, 0) - cnt
        idx = torch.arange(total, device=self.device, dtype=i64) + torch.repeat_interleave(lo - out_start, cnt)
        counts = torch.zeros(self.n_items, dtype=i64, device=self.device)
        counts[it] = cnt
        cptr_p = torch.zeros(self.n_items + 1, dtype=i64, device=self.device)
        cptr_p[1:] = torch.cumsum(counts, 0)
        cval_p = F["cval"][idx]
        # row-major copy: keep the entries whose item is selected
        mask = torch.zeros(self.n_items, dtype=torch.bool, device=self.device)
        mask[it] = True
        keep = mask[F["rcol"].to(i64)]
        csum = torch.zeros(self.nnz + 1, dtype=i64, device=self.device)
        csum[1:] = torch.cumsum(keep, 0)
        return {
            "rptr": csum[F["rptr"].to(i64)].to(torch.int32), "rcol": F["rcol"][keep], "rval": F["rval"][keep],
            "cptr": cptr_p.to(torch.int32), "crow": F["crow"][idx], "cval": cval_p,
            "col_nnz": counts.cpu().numpy(),
            "nonneg": bool(total == 0 or float(cval_p.min()) >= 0.0),
        }
