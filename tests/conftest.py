import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_cpu():
    """The CPU oracle (test infrastructure): builds oracle/libeffort_oracle.so with gcc if needed."""
    from oracle import cpu
    cpu.lib()
    return cpu


@pytest.fixture(scope="session")
def hip_lib_built():
    """Make sure effort_amd/libeffort_hip.so exists (hipcc cross-compiles gfx950 without a GPU)."""
    import effort_amd
    if not os.path.exists(effort_amd._lib.LIB_PATH):
        effort_amd.build()
    return effort_amd._lib.LIB_PATH


# ---- hipGraphs are never destroyed in a GPU test session ---------------------------------------------------------------------
# Destroying a hipGraph that was captured across several streams (fork / join edges: every capture of a context with lanes, every
# multi-stream test) is a use-after-free inside the HIP runtime bundled with this torch wheel (HIP 7.0.51831): a capture -> replay ->
# destroy loop of plain torch ops dies in 3 runs of 4, the same loop that keeps its graphs never does (tools/lab/graph_event_repro.py,
# profiles/r06_heap_hunt.txt).  A test session destroys a few hundred such graphs; so every torch.cuda.CUDAGraph made while
# the suite runs on a GPU takes one reference that is never given back: its destructor never runs, not during the session and not during
# interpreter shutdown -- the process ends like any program that still holds its graphs when it exits, through the ordinary exit path.
_GRAPHS_FOR_LIFE = []


def pytest_sessionstart(session):
    try:
        import torch
    except Exception:                                            # noqa: BLE001
        return
    if not torch.cuda.is_available():
        return
    import ctypes
    orig_new = torch.cuda.CUDAGraph.__new__

    def keeping_new(cls, *a, **k):
        g = orig_new(cls, *a, **k)
        ctypes.pythonapi.Py_IncRef(ctypes.py_object(g))          # (leaked on purpose: see above)
        _GRAPHS_FOR_LIFE.append(g)
        return g
    torch.cuda.CUDAGraph.__new__ = staticmethod(keeping_new)
