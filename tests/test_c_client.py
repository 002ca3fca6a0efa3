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


WIDE = os.path.join(ROOT, "tests", "c_client_wide")


def _build(client=CLIENT):
    """The clients can also be built by `make -C effort_amd/csrc check`.  Here one is only (re)compiled when missing or older than
    its source -- with gcc directly, NOT through make: a test must never find the library "out of date" by some copied timestamp
    and rebuild the .so the test process has mapped."""
    src = client + ".c"
    if os.path.exists(client) and os.path.getmtime(client) >= os.path.getmtime(src):
        return
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(rocm, "include"), src, "-o", client,
                           "-L" + os.path.join(ROOT, "effort_amd"), "-leffort_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
                           "-Wl,-rpath,$ORIGIN/../effort_amd", "-Wl,-rpath," + os.path.join(rocm, "lib")])
    assert os.path.exists(client)


def _run(client, args, tmp_path):
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    res = subprocess.run([client] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr + res.stdout
    calls = []
    for line in res.stdout.splitlines():
        w = line.split()
        if w[:1] == ["dispatch"]:
            calls.append((int(w[1]), int(w[3], 16)))
    return calls, res.stdout


def _bits(x):
    return int(np.float32(x).view(np.uint32))


def test_c_client_builds_from_the_header_alone(hip_lib_built):
    """CPU: the client compiles as C11 against effort_hip.h and links the library (no GPU needed to build)."""
    import re
    for client in (CLIENT, WIDE):
        _build(client)
        src = open(client + ".c").read()
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


@pytest.mark.gpu
def test_c_client_q4(hip_lib_built, oracle_cpu, tmp_path):
    """bucketMulQ4 from C (bucketMulQ4.swift:11-17): effort_convert_q4 -> effort_weights_q4 -> effort_bucketmul_q4.  The converted
    bundle the client dumps equals the oracle's restatement of q4_draft.convert byte for byte (outlier TABLE as a set of rows:
    ties in |w| may be ordered differently), the product the oracle's within the bar, dispatch size and cutoff bits exact."""
    from oracle import q4_layout
    inDim = outDim = 4096
    W = make_w(outDim, inDim, seed=31)
    core2 = np.ascontiguousarray(W.T)
    v = make_v(inDim, seed=6, heavy=True)
    core2.tofile(tmp_path / "core2.f16")
    v.tofile(tmp_path / "v.f32")
    _build(WIDE)
    calls, _ = _run(WIDE, ["q4", tmp_path / "core2.f16", tmp_path / "v.f32", inDim, outDim, repr(0.25), tmp_path / "out.f32"], tmp_path)
    L = q4_layout.convert(core2)
    assert np.fromfile(str(tmp_path / "out.f32") + ".buckets", dtype=np.uint16).tobytes() == np.ascontiguousarray(L["buckets"]).view(np.uint16).tobytes()
    assert np.fromfile(str(tmp_path / "out.f32") + ".stats", dtype=np.float32).tobytes() == np.ascontiguousarray(L["bucket.stats"], dtype=np.float32).tobytes()
    assert np.fromfile(str(tmp_path / "out.f32") + ".probes", dtype=np.uint16).tobytes() == np.ascontiguousarray(L["probes"]).view(np.uint16).tobytes()
    ol = np.fromfile(str(tmp_path / "out.f32") + ".outliers", dtype=np.float32).reshape(-1, 4)
    assert ol.shape == L["outliers"].shape
    key = lambda t: t[np.lexsort((t[:, 2], t[:, 1]))]                            # noqa: E731  (rows ordered by (in, out))
    assert np.array_equal(key(ol), key(np.asarray(L["outliers"], dtype=np.float32)))
    want, n, cutoff = oracle_cpu.bucket_mul_q4(v, L["buckets"], L["bucket.stats"], L["probes"], L["outliers"], inDim, outDim, 0.25)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    assert calls == [(n, _bits(cutoff))]
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 2e-5 * np.abs(want).max()


@pytest.mark.gpu
def test_c_client_group_of_three_on_one_input(hip_lib_built, oracle_cpu, tmp_path):
    """The decode loop's Wq|Wk|Wv (runNetwork.swift:132-134) from C: three handles sharing v, ONE effort_bucketmul_group launch; every
    call's output, dispatch size and cutoff bits against the oracle."""
    inDim, outDims = 4096, (4096, 1024, 1024)
    Ws = [make_w(o, inDim, seed=40 + i) for i, o in enumerate(outDims)]
    v = make_v(inDim, seed=41)
    for i, W in enumerate(Ws):
        W.tofile(tmp_path / f"W{i}.f16")
    v.tofile(tmp_path / "v.f32")
    _build(WIDE)
    calls, _ = _run(WIDE, ["group"] + [tmp_path / f"W{i}.f16" for i in range(3)] + [tmp_path / "v.f32", inDim, *outDims, repr(0.25), tmp_path / "out.f32"], tmp_path)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    assert got.size == sum(outDims) and len(calls) == 3
    off = 0
    for i, (W, o) in enumerate(zip(Ws, outDims)):
        b, s, p, oob = oracle_cpu.convert_fp16(W)
        want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, o, 0.25)
        assert calls[i] == (n, _bits(cutoff)), i
        assert np.abs(got[off:off + o] - want).max() <= 2e-5 * np.abs(want).max(), i
        off += o


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_c_client_column_shards_and_all_gather(hip_lib_built, oracle_cpu, tmp_path, world):
    """north_star's multi-GPU shim from C, as far as one GPU reaches: effort_comm_unique_id -> effort_comm_create (a world of one rank)
    -> effort_weights_column_shard(r, world) for every r -> one multiply per shard -> effort_allgather_outputs.  No PyTorch in the
    client's process: the RCCL the library opens is the system ROCm's.  Every shard selects the full matrix's rows (dispatch
    size, cutoff bits), the gathered vector is the oracle's full product."""
    inDim, outDim = 4096, 11008
    W, v = make_w(outDim, inDim, seed=21), make_v(inDim, seed=23, heavy=True)
    W.tofile(tmp_path / "W.f16")
    v.tofile(tmp_path / "v.f32")
    _build(WIDE)
    calls, out = _run(WIDE, ["shard", tmp_path / "W.f16", tmp_path / "v.f32", inDim, outDim, repr(0.25), world, tmp_path / "out.f32"], tmp_path)
    b, s, p, oob = oracle_cpu.convert_fp16(W)
    want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, 0.25)
    assert calls == [(n, _bits(cutoff))] * world
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
