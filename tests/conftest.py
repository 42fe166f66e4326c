import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no librtrec_amd.so (built artefacts stay out of the history): build it before
    the first test needs it -- hipcc cross-compiles gfx950 without a GPU; the host routines of the interaction
    store (merge / lookup / decay / LRU replay) live in the same library."""
    try:
        from rtrec_amd import build
        if build.is_stale() or build.ops_stale():
            build.build_native()
    except Exception as e:      # the tests that need the library then say so themselves
        print(f"[conftest] could not build librtrec_amd.so: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def oracle():
    from oracle import slim_oracle
    slim_oracle.lib()
    return slim_oracle


@pytest.fixture(scope="session")
def engine():
    """One SlimEngine on cuda:0 for the whole GPU session."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from rtrec_amd.engine import SlimEngine
    return SlimEngine(device="cuda:0")
