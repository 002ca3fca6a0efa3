"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly what
include/effort_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions(names=("effort_hip.h", "effort_hip_debug.h")):
    fns = set()
    for n in names:
        src = open(os.path.join(ROOT, "include", n)).read()
        fns |= set(re.findall(r"EFFORT_API\s+[\w\s\*]+?\b(effort_\w+)\s*\(", src))
    return sorted(fns)


def test_header_declares_the_expected_surface():
    fns = _header_functions(("effort_hip.h",))
    # the lab bench (profiling / tracing / ablation hooks) is not part of the drop-in boundary
    assert not [f for f in fns if f.startswith(("effort_debug_", "effort_kernel_", "effort_enable_kernel_timing", "effort_set_persistent"))]
    for must in ("effort_create", "effort_destroy", "effort_sync", "effort_weights_fp16", "effort_weights_q4",
                 "effort_bucketmul", "effort_bucketmul_q4", "effort_dense_gemv", "effort_convert_fp16",
                 "effort_last_dispatch_count", "effort_calc_dispatch"):
        assert must in fns


def test_library_loads_and_exports_every_declared_symbol(hip_lib_built):
    lib = ctypes.CDLL(hip_lib_built)
    for name in _header_functions():
        assert hasattr(lib, name), f"{name} declared in effort_hip.h but not exported"
    lib.effort_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.effort_version()


def test_python_binding_covers_the_header(hip_lib_built):
    import effort_amd
    assert effort_amd._lib.exported_symbols() == _header_functions()
    effort_amd.lib()        # binds every prototype; raises if one is missing


def test_null_arguments_are_reported_not_crashed(hip_lib_built):
    import effort_amd
    lib = effort_amd.lib()
    assert lib.effort_sync(None) == -1
    assert lib.effort_bucketmul(None, None, None, None, None, 0.25) == -1
    assert lib.effort_set_tuning(None, 0, 0, 0) == -1
    assert lib.effort_last_error(None) == b"null context"


def test_product_has_no_oracle_dependency():
    """The product package must never import or link the oracle (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "effort_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), (dirpath, f)


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import effort_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        effort_amd.gpu()
