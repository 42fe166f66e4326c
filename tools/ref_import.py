"""Import the upstream reference (read-only, /root/reference) in THIS container only.

Used solely by tools/gen_golden.py to produce golden vectors. Never imported by the
product, the tests, smoke() or bench.py: /root/reference does not exist on the GPU box.

rtrec/models/__init__.py eagerly imports the hybrid / LightFM models, which need the
`implicit` and `lightfm` packages (not installed, no network).  Neither is on the SLIM
path, so empty stand-in *modules* are registered before the import.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def import_reference():
    if "rtrec" in sys.modules:
        return sys.modules["rtrec"]
    for name in ("implicit", "implicit.cpu", "implicit.cpu.topk", "lightfm"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["implicit"].cpu = sys.modules["implicit.cpu"]
    sys.modules["implicit.cpu"].topk = sys.modules["implicit.cpu.topk"]
    sys.modules["implicit.cpu.topk"].topk = lambda *a, **k: (_ for _ in ()).throw(
        RuntimeError("implicit is not installed"))
    sys.modules["lightfm"].LightFM = object
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import rtrec  # noqa: F401
    import rtrec.models  # noqa: F401
    return sys.modules["rtrec"]
