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
