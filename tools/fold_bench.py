"""Ordered float32 sums on the device: the binade-speculative fold (csrc/fold_spec.hip.h) against the literal chain.

  python tools/fold_bench.py [--len 55000] [--sums 4096] [--kind drift|zero|int|tie]

Prints ns per entry of one wave (a long single sum) and the chip-wide rate (many sums), and checks every result of
mode 0 against mode 1 bit for bit.  The CPU model of the same control flow is oracle/fold_model.c."""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def stream(kind: str, n: int, rng) -> np.ndarray:
    if kind == "drift":
        return (rng.random(n, dtype=np.float32) * rng.integers(1, 6, n).astype(np.float32) * np.float32(0.37)).astype(np.float32)
    if kind == "zero":
        return ((rng.random(n, dtype=np.float32) - np.float32(0.5)) * rng.integers(1, 6, n).astype(np.float32)).astype(np.float32)
    if kind == "int":
        return rng.integers(0, 26, n).astype(np.float32)
    if kind == "mixed":     # residual-like: mostly positive products, a third negative and smaller
        s = np.where(rng.random(n) < 0.33, -0.3, 1.0).astype(np.float32)
        return (rng.random(n, dtype=np.float32) * s * rng.integers(1, 6, n).astype(np.float32)).astype(np.float32)
    if kind == "tie":
        p = (np.float32(0.5) * rng.integers(-2, 7, n).astype(np.float32)).astype(np.float32)
        p[0] = np.float32(2.0 ** 24 * 1.37)
        return p
    raise ValueError(kind)


def ordered_sums(values, offsets, mode, torch, lib, reps=1):
    out = torch.empty(len(offsets) - 1, dtype=torch.float32, device=values.device)
    st = torch.cuda.current_stream().cuda_stream
    from rtrec_amd import _native
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _native.check(lib.rtrec_slim_ordered_sums(values.data_ptr(), offsets.data_ptr(), len(offsets) - 1, mode, out.data_ptr(), st), "ordered_sums")
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        _native.check(lib.rtrec_slim_ordered_sums(values.data_ptr(), offsets.data_ptr(), len(offsets) - 1, mode, out.data_ptr(), st), "ordered_sums")
    ev1.record()
    torch.cuda.synchronize()
    return out.cpu().numpy(), ev0.elapsed_time(ev1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--len", type=int, default=55000)
    ap.add_argument("--sums", type=int, default=4096)
    ap.add_argument("--kinds", default="drift,mixed,zero,int,tie")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import torch
    from rtrec_amd import _native
    lib = _native.load()
    rng = np.random.default_rng(5)
    res = []
    for kind in a.kinds.split(","):
        for n_sums in (1, a.sums):
            vals = np.concatenate([stream(kind, a.len, rng) for _ in range(min(n_sums, 64))])
            if n_sums > 64:
                vals = np.tile(vals, (n_sums + 63) // 64)[: n_sums * a.len]
            off = (np.arange(n_sums + 1, dtype=np.int64) * a.len)
            dv = torch.from_numpy(vals).cuda()
            do = torch.from_numpy(off).cuda()
            o0, t0 = ordered_sums(dv, do, 0, torch, lib, reps=3)
            o1, t1 = ordered_sums(dv, do, 1, torch, lib, reps=3)
            same = bool(np.array_equal(o0.view(np.uint32), o1.view(np.uint32)))
            r = {"kind": kind, "len": a.len, "sums": n_sums, "bit_equal": same,
                 "spec_ns_per_entry": t0 / (n_sums * a.len) * 1e9 * (1 if n_sums > 1 else 1),
                 "chain_ns_per_entry": t1 / (n_sums * a.len) * 1e9, "spec_s": t0, "chain_s": t1}
            print(json.dumps(r), flush=True)
            res.append(r)
    if a.out:
        with open(a.out, "w") as f:
            for r in res:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
