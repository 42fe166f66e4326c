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
SOURCES = ["score.hip", "fit.hip", "store_host.hip", "store_device.hip", "seg_build.hip", "score_refine.hip", "score_cands.hip", "fit_sgd.hip", "score_dense_fill.hip", "score_first_touch.hip", "ordered_fold.hip"]
HEADERS = ["common.hip.h", "score_seg.hip.h", "fold_spec.hip.h", os.path.join("..", "..", "include", "rtrec_amd.h")]
# -ffp-contract=off: the kernels reproduce the reference's float32 rounding sequence, so a
# multiply must never be fused into the following add.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the ROCm toolchain is required to build rtrec_amd")
    return exe


BUILD_INFO = os.path.join(LIB_DIR, "build_info.json")


def _sha256(path: str) -> str:
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def source_sha256() -> str:
    """sha256 over the kernel sources the library is compiled from (csrc/*.hip, their headers, include/rtrec_amd.h) and the
    compiler flags: the same for every rebuild of the same tree, whatever the binary's bytes."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for name in sorted(SOURCES) + sorted(HEADERS):
        path = os.path.normpath(os.path.join(CSRC, name))
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def same_build(a: dict, b: dict) -> bool:
    """Two fingerprints name the same kernels: equal library bytes, or equal sources + flags."""
    a, b = a or {}, b or {}
    return bool((a.get("lib_sha256") and a.get("lib_sha256") == b.get("lib_sha256"))
                or (a.get("src_sha256") and a.get("src_sha256") == b.get("src_sha256")))


def fingerprint() -> dict:
    """Identity of the kernel library a measurement was taken with: sha256 of librtrec_amd.so (what actually ran) plus the
    git commit it was built at (recorded at build time -- the GPU box has no .git).  Profile summaries carry it
    (tools/pmc_round_summary.py) and bench.py attaches a summary to its line only when it matches the running build."""
    import json
    info = {}
    try:
        info = json.load(open(BUILD_INFO))
    except Exception:
        pass
    from . import settings
    ab_lib = settings.raw("RTREC_AMD_LIB")
    lib = ab_lib or LIB_PATH
    sha = _sha256(lib) if os.path.exists(lib) else None
    if info.get("lib_sha256") != sha:           # an A/B library or a rebuild without the record: only the hash is known
        info = {"git_head": None, "git_dirty": None}
    src = None
    if not ab_lib:                                   # (an A/B library was built from some other tree)
        try:
            src = source_sha256()
        except OSError:
            src = None
    return {"lib_sha256": sha, "src_sha256": src, "git_head": info.get("git_head"), "git_dirty": info.get("git_dirty")}


def _write_build_info() -> None:
    import json
    head, dirty = None, None
    try:
        root = os.path.dirname(_HERE)
        head = subprocess.check_output(["git", "rev-parse", "HEAD"], cwd=root, stderr=subprocess.DEVNULL).decode().strip()
        dirty = bool(subprocess.check_output(["git", "status", "--porcelain", "--", "rtrec_amd/csrc", "include"], cwd=root,
                                             stderr=subprocess.DEVNULL).decode().strip())
    except Exception:
        pass
    try:
        with open(BUILD_INFO, "w") as f:
            json.dump({"lib_sha256": _sha256(LIB_PATH), "git_head": head, "git_dirty": dirty}, f)
    except OSError:
        pass


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        os.makedirs(LIB_DIR, exist_ok=True)
        cmd = [_hipcc()] + HIPCC_FLAGS + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd, cwd=CSRC)
        _write_build_info()
    elif not os.path.exists(BUILD_INFO):
        _write_build_info()
    build_ops(force=force, verbose=verbose)
    return LIB_PATH


# The PyTorch-ROCm custom ops (torch.ops.rtrec_amd.*) are registered from C++ (csrc/torch_ops.cpp, TORCH_LIBRARY): host
# code only, built with the host compiler against the torch headers; it binds librtrec_amd.so at run time (dlopen).
OPS_PATH = os.path.join(LIB_DIR, "librtrec_amd_ops.so")
OPS_SOURCE = os.path.join(CSRC, "torch_ops.cpp")


OPS_STAMP = OPS_PATH + ".stamp"      # torch version + C++ ABI flag the ops library was compiled against


def _torch_stamp() -> str:
    import torch
    return f"{torch.__version__} abi={int(getattr(torch._C, '_GLIBCXX_USE_CXX11_ABI', True))}"


def ops_stale() -> bool:
    """The ops library is compiled against torch's headers and C++ ABI: a torch upgrade makes it stale even though no source
    changed -- the stamp file beside it records what it was built for."""
    if not os.path.exists(OPS_PATH):
        return True
    t = os.path.getmtime(OPS_PATH)
    if any(os.path.getmtime(d) > t for d in (OPS_SOURCE, os.path.normpath(os.path.join(CSRC, HEADERS[-1])))):
        return True
    try:
        return open(OPS_STAMP).read().strip() != _torch_stamp()
    except OSError:
        return True


def build_ops(force: bool = False, verbose: bool = False) -> str:
    if not force and not ops_stale():
        return OPS_PATH
    import torch
    from torch.utils import cpp_extension
    os.makedirs(LIB_DIR, exist_ok=True)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cxx = shutil.which("g++") or "g++"
    abi = int(getattr(torch._C, "_GLIBCXX_USE_CXX11_ABI", True))
    cmd = ([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-I/opt/rocm/include"]
           + [f"-I{p}" for p in cpp_extension.include_paths()]
           + [OPS_SOURCE, "-o", OPS_PATH, f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", "-ldl",
              f"-Wl,-rpath,{tlib}"])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=CSRC)
    with open(OPS_STAMP, "w") as f:
        f.write(_torch_stamp() + "\n")
    return OPS_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
