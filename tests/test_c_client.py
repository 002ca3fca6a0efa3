"""The drop-in boundary called from C: tests/c_client.c uses nothing but include/effort_hip.h and the HIP runtime API (no
Python, no ctypes, no torch in its process).  It is the stand-in for the Swift shim of INTEGRATION.md -- bucketMul.swift:11,
helpers/gpu.swift:146-196 -- which cannot be built in this image.  Its dump is checked against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from tests.util import make_v, make_w

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENT = os.path.join(ROOT, "tests", "c_client")


def _build():
    """tests/c_client is built by `make -C effort_amd/csrc c_client` (part of __graft_entry__.build()).  Here it is only
    (re)compiled when missing or older than its source -- with gcc directly, NOT through make: a test must never find the
    library "out of date" by some copied timestamp and rebuild the .so the test process has mapped."""
    src = os.path.join(ROOT, "tests", "c_client.c")
    if os.path.exists(CLIENT) and os.path.getmtime(CLIENT) >= os.path.getmtime(src):
        return
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), src, "-o", CLIENT,
                           "-L" + os.path.join(ROOT, "effort_amd"), "-leffort_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                           "-Wl,-rpath,$ORIGIN/../effort_amd", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    assert os.path.exists(CLIENT)


def test_c_client_builds_from_the_header_alone(hip_lib_built):
    """CPU: the client compiles as C11 against effort_hip.h and links the library (no GPU needed to build)."""
    _build()
    src = open(os.path.join(ROOT, "tests", "c_client.c")).read()
    import re
    assert sorted(re.findall(r"#include\s+[<\"]([^>\"]+)[>\"]", src)) == ["effort_hip.h", "hip/hip_runtime_api.h", "stdio.h", "stdlib.h", "string.h"]


@pytest.mark.gpu
@pytest.mark.parametrize("outDim,effort", [(4096, 0.25), (11008, 0.25), (1024, 1.0)])
def test_c_client_matches_oracle(hip_lib_built, oracle_cpu, tmp_path, outDim, effort):
    _build()
    inDim = 4096
    W, v = make_w(outDim, inDim, seed=21), make_v(inDim, seed=22, heavy=True)
    W.tofile(tmp_path / "W.f16")
    v.tofile(tmp_path / "v.f32")
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    res = subprocess.run([CLIENT, str(tmp_path / "W.f16"), str(tmp_path / "v.f32"), str(inDim), str(outDim), repr(effort), str(tmp_path / "out.f32")],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr
    words = res.stdout.split()
    count, cutoff_bits, dropped = int(words[1]), int(words[3], 16), int(words[5])
    b, s, p, oob = oracle_cpu.convert_fp16(W)
    want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, effort)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    assert dropped == oob == 0
    assert count == n                                                             # dispatch.size: exact
    assert cutoff_bits == int(np.float32(cutoff).view(np.uint32))                # BucketMul.cutoff: the float's bits
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
