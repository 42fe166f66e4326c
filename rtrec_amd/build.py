"""Build the gfx950 shared library (librtrec_amd.so) in-tree with hipcc.

`python -m rtrec_amd.build` or `rtrec_amd.build.build_native()`.  hipcc cross-compiles
without a GPU; the resulting .so travels to the GPU box with the source tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "librtrec_amd.so")
SOURCES = ["score.hip", "fit.hip", "store_host.hip", "store_device.hip"]
HEADERS = ["common.hip.h", "score_seg.hip.h", os.path.join("..", "..", "include", "rtrec_amd.h")]
# -ffp-contract=off: the kernels reproduce the reference's float32 rounding sequence, so a
# multiply must never be fused into the following add.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the ROCm toolchain is required to build rtrec_amd")
    return exe


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
