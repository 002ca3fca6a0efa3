"""GPU parity tests: the HIP path (through the C ABI, via the effort_amd host mirror) against the CPU oracle
on the same seeded inputs.

Bars (stated here, used below):
  * integer / selection work is BIT-EXACT: cutoff value, dispatch count, dispatch list entries, the whole
    converted layout (buckets, stats, probes);
  * f32 outputs: the reference itself has no defined summation order (atomic dispatch append, simd_sum,
    atomic float adds), so outputs are compared with  |hip - oracle| <= 2e-5 * max|oracle|  and
    cos-sim >= 0.999999 (north_star asks >= 0.999); two runs of the HIP path must be bit-identical.
"""
import numpy as np
import pytest
import torch

from tests.util import cos, make_v, make_w

_KEEP_ALIVE = []       # events / streams that took part in a hipGraph capture are never destroyed (see test_launches_in_flight_on_several_streams)

pytestmark = pytest.mark.gpu

ATOL_REL = 2e-5
DEV = "cuda:0"


def dev16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(DEV)


def devf(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(DEV)


_CACHE = {}


def converted(oracle_cpu, outDim, inDim, seed=1234, zeros=0, scale=0.02):
    key = (outDim, inDim, seed, zeros, scale)
    if key not in _CACHE:
        W = make_w(outDim, inDim, seed=seed, zeros=zeros, scale=scale)
        b, s, p, oob = oracle_cpu.convert_fp16(W)
        assert oob == 0
        _CACHE[key] = (W, b, s, p)
    return _CACHE[key]


def gpu_weights(ea, W, b, s, p, **kw):
    outDim, inDim = W.shape
    return ea.ExpertWeights(dev16(b), dev16(s), dev16(p), inSize=inDim, outSize=outDim, core=dev16(W).view(torch.float16), **kw)


@pytest.fixture(scope="module")
def ea(hip_lib_built):
    import effort_amd
    assert torch.cuda.is_available()
    effort_amd.gpu(0)
    return effort_amd


def close(out, want):
    if not want.any():                       # nothing dispatched (effort 0 can select no row at all)
        return not out.any()
    tol = ATOL_REL * float(np.abs(want).max() + 1e-30)
    return float(np.abs(out - want).max()) <= tol and cos(out, want) >= 0.999999


# ---------------------------------------------------------------- cutoff + dispatch: exact
@pytest.mark.parametrize("outDim,inDim", [(256, 4096), (1024, 4096), (4096, 4096)])
@pytest.mark.parametrize("heavy", [False, True])
def test_cutoff_and_dispatch_bit_exact(ea, oracle_cpu, outDim, inDim, heavy):
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    v = make_v(inDim, seed=5, heavy=heavy)
    vd = devf(v)
    bm = ea.BucketMul.shared()
    for effort in (0.0, 0.08, 0.25, 0.5, 0.9, 1.0):
        cutoff, _ = oracle_cpu.find_cutoff(v, p, 0, effort)
        disp, n = oracle_cpu.prepare_dispatch(v, s, 0, cutoff, inDim, outDim // 16)
        bm.calcDispatch(vd, ew, None, effort)
        ea.gpu().eval()
        assert np.float32(bm.cutoff).tobytes() == np.float32(cutoff).tobytes(), (effort, bm.cutoff, cutoff)
        n_hip = int(bm.dispatch_size.item())
        assert n_hip == n == ea.gpu().last_dispatch_count()
        got = bm.dispatch[:n].cpu().numpy()
        assert got.tobytes() == disp[:n].tobytes()


def test_cutoff_across_input_scales(ea, oracle_cpu):
    """The same input scaled over twelve decades: the cutoff crosses 128, below which the kernel runs the reference's last
    bisection rounds as a loop and from which it takes their fixed point directly (cutoff_device.h, bisect_to_cell_edge) --
    bit-exact with the oracle either way, through the fused multiply as well as the standalone kernel."""
    outDim, inDim = 1024, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    bm = ea.BucketMul.shared()
    base = make_v(inDim, seed=17, heavy=True)
    seen_small = seen_big = 0
    for k in range(-9, 4):
        for mult in (1.0, 1.7, 3.1):
            v = (base * np.float32(mult * 10.0 ** k)).astype(np.float32)
            vd = devf(v)
            for effort in (0.1, 0.25, 0.6):
                cutoff, _ = oracle_cpu.find_cutoff(v, p, 0, effort)
                bm.calcDispatch(vd, ew, None, effort)
                ea.gpu().eval()
                assert np.float32(bm.cutoff).tobytes() == np.float32(cutoff).tobytes(), (k, mult, effort, bm.cutoff, cutoff)
                out = torch.zeros(outDim, device=DEV)
                ea.bucketMul(vd, ew, None, out, effort)
                ea.gpu().eval()
                assert np.float32(ea.gpu().last_cutoff()).tobytes() == np.float32(cutoff).tobytes(), (k, mult, effort)
                seen_small += cutoff < 128.0
                seen_big += cutoff >= 128.0
    assert seen_small > 10 and seen_big > 10, (seen_small, seen_big)


def test_cutoff_with_exact_zero_inputs(ea, oracle_cpu):
    """Exact zeros among the probe products (zero inputs; Q4 probes zeroed as outliers) stretch the value range down to
    0: the cutoff stays bit-exact at every effort, including effort 1 where the reference bisects towards 0 for its
    full 100 rounds."""
    outDim, inDim = 1024, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    bm = ea.BucketMul.shared()
    for heavy in (False, True):
        v = make_v(inDim, seed=9, heavy=heavy)
        v[::7] = 0.0
        v[5] = -0.0
        vd = devf(v)
        for effort in (0.0, 0.05, 0.5, 0.97, 1.0):
            cutoff, _ = oracle_cpu.find_cutoff(v, p, 0, effort)
            disp, n = oracle_cpu.prepare_dispatch(v, s, 0, cutoff, inDim, outDim // 16)
            bm.calcDispatch(vd, ew, None, effort)
            ea.gpu().eval()
            assert np.float32(bm.cutoff).tobytes() == np.float32(cutoff).tobytes(), (heavy, effort, bm.cutoff, cutoff)
            assert int(bm.dispatch_size.item()) == n
            out = torch.zeros(outDim, device=DEV)
            ea.bucketMul(vd, ew, None, out, effort)
            ea.gpu().eval()
            assert ea.gpu().last_cutoff() == cutoff and ea.gpu().last_dispatch_count() == n


# ---------------------------------------------------------------- FP16 multiply
SHAPES = [(256, 4096), (64, 4096), (1024, 4096), (4096, 4096), (11008, 4096), (4096, 14336)]


@pytest.mark.parametrize("outDim,inDim", SHAPES)
def test_bucketmul_matches_oracle(ea, oracle_cpu, outDim, inDim):
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    out = torch.full((outDim,), float("nan"), device=DEV)
    for heavy in (False, True):
        v = make_v(inDim, seed=8, heavy=heavy)
        vd = devf(v)
        for effort in (0.0, 0.1, 0.25, 0.5, 1.0):
            want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, effort)
            ea.bucketMul(vd, ew, None, out, effort)
            ea.gpu().eval()
            assert ea.gpu().last_dispatch_count() == n, (effort, n)
            assert ea.gpu().last_cutoff() == cutoff
            assert close(out.cpu().numpy(), want), (outDim, inDim, heavy, effort)


def test_bucketmul_default_effort_and_expertmul(ea, oracle_cpu):
    W, b, s, p = converted(oracle_cpu, 256, 4096)
    ew = gpu_weights(ea, W, b, s, p)
    v = make_v(4096, seed=1)
    out = torch.zeros(256, device=DEV)
    ea.expertMul(devf(v), ew, out)                                    # effort defaults to 0.25 (bucketMul.swift:11)
    ea.gpu().eval()
    want, _, _ = oracle_cpu.bucket_mul(v, b, s, p, 4096, 256, 0.25)
    assert close(out.cpu().numpy(), want)


@pytest.mark.parametrize("waves,elems", [(16, 1), (16, 2), (8, 1), (8, 2), (8, 4), (4, 1), (4, 2), (4, 4)])
def test_launch_geometries_agree_and_are_deterministic(ea, oracle_cpu, waves, elems):
    outDim, inDim = 1024, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    v = make_v(inDim, seed=2, heavy=True)
    vd = devf(v)
    want, n, _ = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, 0.3)
    g = ea.gpu()
    try:
        for slices in (0, 8, 12, 20, 64, 512):                        # (12, 20: not multiples of 8 -- the item grid is padded, the call's first block may be padding)
            g.set_tuning(waves, elems, slices)
            o1 = torch.zeros(outDim, device=DEV)
            o2 = torch.zeros(outDim, device=DEV)
            ea.bucketMul(vd, ew, None, o1, 0.3)
            ea.bucketMul(vd, ew, None, o2, 0.3)
            g.eval()
            assert g.last_dispatch_count() == n
            assert torch.equal(o1, o2)                                # fixed summation order
            assert close(o1.cpu().numpy(), want), (waves, elems, slices)
    finally:
        g.set_tuning(0, 0, 0)


def test_experts_and_percentload(ea, oracle_cpu):
    """expNo offsets into a stacked buffer (bucketMul.metal:58) and percentLoad truncates the rank planes
    (loader.swift:157-159)."""
    outDim, inDim = 256, 4096
    mats = [converted(oracle_cpu, outDim, inDim, seed=100 + e) for e in range(3)]
    ews = [gpu_weights(ea, *m) for m in mats]
    stacked = ea.ExpertWeights.stack(ews)
    B = np.concatenate([m[1] for m in mats])
    S = np.concatenate([m[2] for m in mats])
    P = np.concatenate([m[3] for m in mats])
    v = make_v(inDim, seed=4)
    vd = devf(v)
    out = torch.zeros(outDim, device=DEV)
    for e in range(3):
        expNo = torch.tensor([e], dtype=torch.int32, device=DEV)
        want, n, _ = oracle_cpu.bucket_mul(v, B, S, P, inDim, outDim, 0.4, expNo=e)
        ea.bucketMul(vd, stacked, expNo, out, 0.4)
        ea.gpu().eval()
        assert ea.gpu().last_dispatch_count() == n
        assert close(out.cpu().numpy(), want)
        single, _, _ = oracle_cpu.bucket_mul(v, mats[e][1], mats[e][2], mats[e][3], inDim, outDim, 0.4)
        assert np.array_equal(want, single)
    for pl in (8, 4, 1):
        W, b, s, p = mats[0]
        rows = inDim * pl
        ewt = ews[0].truncated(pl)
        want, n, _ = oracle_cpu.bucket_mul(v, b[:rows], s[:rows], p, inDim, outDim, 0.6, percentLoad=pl)
        ea.bucketMul(vd, ewt, None, out, 0.6)
        ea.gpu().eval()
        assert ea.gpu().last_dispatch_count() == n and close(out.cpu().numpy(), want)


# ---------------------------------------------------------------- converter: bit-exact layout
@pytest.mark.parametrize("outDim,inDim,zeros", [(256, 4096, 0), (64, 4096, 0), (4096, 4096, 50), (4160, 4096, 40), (11008, 4096, 0)])
def test_gpu_converter_bit_exact(ea, oracle_cpu, outDim, inDim, zeros):
    W, b, s, p = converted(oracle_cpu, outDim, inDim, seed=77, zeros=zeros)
    t = {}
    ea.bucketize(dev16(W).view(torch.float16), "x.", t)
    ea.gpu().eval()
    assert t["x.buckets"].cpu().numpy().view(np.uint16).tobytes() == b.view(np.uint16).tobytes()
    assert t["x.bucket.stats"].cpu().numpy().view(np.uint16).tobytes() == s.view(np.uint16).tobytes()
    assert t["x.probes"].cpu().numpy().view(np.uint16).tobytes() == p.view(np.uint16).tobytes()


def test_converter_preconditions(ea):
    import effort_amd
    with pytest.raises(effort_amd.EffortError):
        ea.bucketize(torch.zeros((96, 4096), dtype=torch.float16, device=DEV), "", {})
    with pytest.raises(effort_amd.EffortError):
        ea.bucketize(torch.zeros((256, 2048), dtype=torch.float16, device=DEV), "", {})
    with pytest.raises(NotImplementedError):
        ea.bucketize(torch.zeros((256, 4096), dtype=torch.float16, device=DEV), "", {}, goQ8=True)


# ---------------------------------------------------------------- Q4
@pytest.fixture(scope="module")
def q4_case():
    from oracle import q4_layout
    inDim, outDim = 4096, 4096            # probes = diag(core) must have 4096 entries (q4_draft.py:240)
    W = make_w(outDim, inDim, seed=31)
    L = q4_layout.convert(np.ascontiguousarray(W.T))
    return W, L, inDim, outDim


@pytest.mark.parametrize("with_outliers", [True, False])
def test_bucketmul_q4_matches_oracle(ea, oracle_cpu, q4_case, with_outliers):
    W, L, inDim, outDim = q4_case
    ol = L["outliers"] if with_outliers else None
    ew = ea.ExpertWeights(dev16(L["buckets"]), devf(L["bucket.stats"]), dev16(L["probes"]), inSize=inDim, outSize=outDim,
                          outliers=None if ol is None else devf(ol), q4=True)
    out = torch.full((outDim,), float("nan"), device=DEV)                # the call must zero it (expertMul.swift:27)
    for heavy in (False, True):
        v = make_v(inDim, seed=6, heavy=heavy)
        vd = devf(v)
        for effort in (0.0, 0.15, 0.25, 0.5, 1.0):
            want, n, cutoff = oracle_cpu.bucket_mul_q4(v, L["buckets"], L["bucket.stats"], L["probes"], ol, inDim, outDim, effort)
            ea.expertMul(vd, ew, out, effort)
            ea.gpu().eval()
            assert ea.gpu().last_dispatch_count() == n and ea.gpu().last_cutoff() == cutoff
            assert close(out.cpu().numpy(), want), (heavy, effort)
    # Q4 dispatch list in the reference's format (value pre-multiplied by the row mean)
    bm = ea.BucketMulQ4.shared()
    cutoff, _ = oracle_cpu.find_cutoff(v, L["probes"], 0, 0.25)
    disp, n = oracle_cpu.prepare_dispatch_q4(v, L["bucket.stats"], 0, cutoff, inDim, outDim // 32)
    bm.calcDispatch(vd, ew, None, 0.25)
    ea.gpu().eval()
    assert int(bm.dispatch_size.item()) == n and bm.dispatch[:n].cpu().numpy().tobytes() == disp[:n].tobytes()


def test_gpu_q4_converter_matches_oracle(ea, q4_case):
    """effort_amd.q4.convert on the GPU vs the numpy restatement of q4_draft.convert (itself pinned by the
    reference's fixtures): identical buckets, stats, probes and outlier table."""
    W, L, inDim, outDim = q4_case
    out = ea.q4_convert(dev16(np.ascontiguousarray(W.T)).view(torch.float16))
    assert out["buckets"].cpu().numpy().view(np.uint16).tobytes() == L["buckets"].view(np.uint16).tobytes()
    assert out["bucket.stats"].cpu().numpy().tobytes() == L["bucket.stats"].tobytes()
    assert out["probes"].cpu().numpy().view(np.uint16).tobytes() == L["probes"].view(np.uint16).tobytes()
    assert out["outliers"].cpu().numpy().tobytes() == L["outliers"].tobytes()       # same stable order


def test_q4_quality_vs_dense(ea, oracle_cpu, q4_case):
    """playground.swift:16-41: bucketMulQ4 vs basicMul cos-sim (the reference prints a tick above 0.99 on real
    weights at its default effort; on i.i.d. Gaussian weights sign+mean quantisation is coarser)."""
    W, L, inDim, outDim = q4_case
    ew = ea.ExpertWeights(dev16(L["buckets"]), devf(L["bucket.stats"]), dev16(L["probes"]), inSize=inDim, outSize=outDim,
                          outliers=devf(L["outliers"]), core=dev16(W).view(torch.float16), q4=True)
    v = devf(make_v(inDim, seed=12, heavy=True))
    test = torch.zeros(outDim, device=DEV)
    control = torch.zeros(outDim, device=DEV)
    ea.basicMul(v, ew.core, control)
    ea.expertMul(v, ew, test, 1.0)
    assert ea.cosineSimilarityTo(test, control) > 0.93


# ---------------------------------------------------------------- dense baseline, errors
@pytest.mark.parametrize("outDim,inDim", [(1024, 4096), (1027, 4112), (8, 16), (300, 14336)])
def test_dense_gemv_matches_oracle(ea, oracle_cpu, outDim, inDim):
    """basicMul (helpers/mps.swift:14-47): the streaming kernel (csrc/gemv.hip) and the rocBLAS backend against the f32
    restatement; ragged sizes: rows not a multiple of the 8 a workgroup takes, inDim not a multiple of the 512-element chunk."""
    W = make_w(outDim, inDim, seed=3)
    v = make_v(inDim, seed=9, heavy=True)
    want = oracle_cpu.dense_gemv(W, v, round_v_to_f16=True)             # v.asFloat16(), mps.swift:19
    g = ea.gpu()
    for rocblas in (False, True):
        out = torch.full((outDim,), float("nan"), device=DEV)
        try:
            g.set_dense_backend(rocblas)
            ea.basicMul(devf(v), dev16(W).view(torch.float16), out)
            g.eval()
        finally:
            g.set_dense_backend(False)
        # own kernel: f32 sums of exact f16 products in another order than the oracle's (1e-5 band); rocBLAS's internal
        # accumulation is a third-party detail, as MPS's is for the reference (1e-3 band)
        tol = 1e-3 if rocblas else 1e-5
        assert np.allclose(out.cpu().numpy(), want, rtol=tol, atol=tol * np.abs(want).max()), rocblas


def test_error_reporting(ea, oracle_cpu):
    import effort_amd
    W, b, s, p = converted(oracle_cpu, 256, 4096)
    ew = gpu_weights(ea, W, b, s, p)
    v = devf(make_v(4096))
    out = torch.zeros(256, device=DEV)
    with pytest.raises(effort_amd.EffortError, match="EFFORT_ERR_EFFORT"):
        ea.bucketMul(v, ew, None, out, 1.5)
    with pytest.raises(ValueError):
        ea.bucketMulQ4(v, ew, None, out, 0.25)                         # FP16 bundle into the Q4 call
    with pytest.raises(ValueError):
        ea.bucketMul(v[:100].contiguous(), ew, None, out, 0.25)        # v too short
    with pytest.raises(effort_amd.EffortError):                        # outDim % 16 != 0, bucketMul.swift:73
        ea.ExpertWeights(torch.zeros((4096 * 16, 15), dtype=torch.int16, device=DEV), dev16(s), dev16(p), inSize=4096, outSize=250).handle
    # a Q4 outlier table naming elements outside the matrix (e.g. a full-matrix table handed to a column shard) is refused
    qb = torch.zeros((4096 * 8, 8), dtype=torch.int16, device=DEV)
    qs = torch.zeros((4096 * 8, 2), dtype=torch.float32, device=DEV)
    for bad in ([[0.5, 4096.0, 3.0, 0.0]], [[0.5, 7.0, 256.0, 0.0]], [[0.5, float("nan"), 3.0, 0.0]], [[0.5, -1.0, 3.0, 0.0]]):
        with pytest.raises(effort_amd.EffortError, match="outlier"):
            ea.ExpertWeights(qb, qs, dev16(p), inSize=4096, outSize=256, outliers=torch.tensor(bad, device=DEV), q4=True).handle
    ea.ExpertWeights(qb, qs, dev16(p), inSize=4096, outSize=256, outliers=torch.tensor([[0.5, 4095.0, 255.0, 0.0]], device=DEV), q4=True).handle
    with pytest.raises(effort_amd.EffortError, match="f16"):           # the 4-byte index entry keeps the value as f16 (the table comes from an f16 matrix)
        ea.ExpertWeights(qb, qs, dev16(p), inSize=4096, outSize=256, outliers=torch.tensor([[0.1, 5.0, 5.0, 0.0]], device=DEV), q4=True).handle


def test_weights_rewritten_in_place_need_a_refresh(ea, oracle_cpu):
    """The fixed-point scale comes from a snapshot of the weights (rankBound).  Buffers rewritten in place with LARGER
    weights (loader.swift's buffers are mutable) are picked up by effort_weights_refresh: the product matches the oracle
    again; the bound itself is readable and scales with the weights."""
    outDim, inDim = 1024, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    v = make_v(inDim, seed=4)
    vd = devf(v)
    out = torch.zeros(outDim, device=DEV)
    ea.bucketMul(vd, ew, None, out, 0.5)
    ea.gpu().eval()
    bound0 = ew.rank_bound()[0]
    W2 = (W.astype(np.float32) * 16).astype(np.float16)               # exact scaling: same order, same positions
    b2, s2, p2, _ = oracle_cpu.convert_fp16(W2)
    ew.buckets.copy_(dev16(b2).reshape(ew.buckets.shape))
    ew.stats.copy_(dev16(s2).reshape(ew.stats.shape))
    ew.probes.copy_(dev16(p2).reshape(ew.probes.shape))
    ew.refresh()
    assert abs(ew.rank_bound()[0] / bound0 - 16.0) < 0.05
    want, n, cutoff = oracle_cpu.bucket_mul(v, b2, s2, p2, inDim, outDim, 0.5)
    ea.bucketMul(vd, ew, None, out, 0.5)
    ea.gpu().eval()
    assert ea.gpu().last_dispatch_count() == n and ea.gpu().last_cutoff() == cutoff and close(out.cpu().numpy(), want)
    # ... and by a PERSISTENT launch, which stages the row means from the compact copy made at registration (refreshed too)
    g = ea.gpu()
    outs = [torch.full((outDim,), float("nan"), device=DEV) for _ in range(32)]
    try:
        g.set_tuning(8, 1, 64)                           # 64 slices per call: thousands of items
        ea.bucketMulGroup([(vd, ew, None, o, 0.5) for o in outs])
        g.eval()
    finally:
        g.set_tuning(0, 0, 0)
    for i in (0, 13, 31):
        assert g.last_dispatch_count(i) == n and g.last_cutoff(i) == cutoff and close(outs[i].cpu().numpy(), want), i


def test_converter_reports_dropped_elements(ea, oracle_cpu):
    """effort_convert_status: rows where zero padding ties with real zeros overfill bucket 0 exactly as the reference's
    preBucketize does (convert.metal:40-61); the drops are counted instead of passing silently.  The oracle counts the same."""
    outDim, inDim = 4160, 4096
    W = make_w(outDim, inDim, seed=3, zeros=3000)
    W[:, 7] = 0                                                       # a column of the HF matrix = one whole input row of zeros
    b, s, p, oob = oracle_cpu.convert_fp16(W)
    ea.gpu().convert_status()
    t = {}
    ea.bucketize(dev16(W).view(torch.float16), "", t)
    ea.gpu().eval()
    assert t["buckets"].cpu().numpy().view(np.uint16).tobytes() == np.ascontiguousarray(b).view(np.uint16).tobytes()
    assert ea.gpu().convert_status() == oob and ea.gpu().convert_status() == 0


# ---------------------------------------------------------------- full-size properties (BASELINE config B)
def test_full_size_properties(ea, oracle_cpu):
    outDim, inDim = 11008, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    v = make_v(inDim, seed=21, heavy=True)
    vd, v2 = devf(v), devf(v * 2)
    a, a2, c = (torch.zeros(outDim, device=DEV) for _ in range(3))
    counts = []
    for effort in (0.1, 0.25, 0.5, 0.75, 1.0):
        ea.bucketMul(vd, ew, None, a, effort)
        n1 = ea.gpu().last_dispatch_count()
        ea.bucketMul(v2, ew, None, a2, effort)
        n2 = ea.gpu().last_dispatch_count()
        ea.bucketMul(vd, ew, None, c, effort)
        ea.gpu().eval()
        assert n1 == n2 and torch.equal(a * 2, a2)                      # exact power-of-two scale invariance
        assert torch.equal(a, c)                                        # run-to-run determinism
        counts.append(n1)
    assert counts == sorted(counts)
    assert abs(counts[1] / (16 * inDim) - 0.25) < 0.05
    dense = torch.zeros(outDim, device=DEV)
    ea.basicMul(vd, ew.core, dense)
    assert ea.cosineSimilarityTo(a, dense) > 0.9999                     # effort 1.0 vs dense (benchmark.swift:166-177)


def test_column_shards_reproduce_the_full_product(ea, oracle_cpu):
    """Multi-GPU partition exercised on one device: the two column shards select the same rows and
    their concatenated outputs equal the unsharded product."""
    from effort_amd.sharded import ShardedExpertWeights, shardedExpertMulGroup
    outDim, inDim = 1024, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    v = make_v(inDim, seed=33)
    vd = devf(v)
    want, n, _ = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, 0.5)
    halves = []
    for r in range(2):
        sh = ew.column_shard(r, 2)
        o = torch.zeros(outDim // 2, device=DEV)
        ea.bucketMul(vd, sh, None, o, 0.5)
        ea.gpu().eval()
        assert ea.gpu().last_dispatch_count() == n
        halves.append(o.cpu().numpy())
    assert close(np.concatenate(halves), want)
    one = ShardedExpertWeights(ew, outDim, 0, 1)
    o = torch.zeros(outDim, device=DEV)
    shardedExpertMulGroup(vd, [one], [o], 0.5)
    ea.gpu().eval()
    assert close(o.cpu().numpy(), want)


# ---------------------------------------------------------------- grouped launches
@pytest.mark.parametrize("split", [False, True])
def test_group_launch_equals_single_calls(ea, oracle_cpu, split):
    """effort_bucketmul_group: n independent calls (different shapes, inputs, efforts) in one launch.  Selection is
    exact whatever the launch geometry: every call's cutoff / dispatch.size equal the single launch's and the
    oracle's.  With the geometry pinned the outputs are the single launches' bit for bit; with the default geometry
    (fatter row slices the more calls share the launch) they agree to the f32 rounding of the slice sums."""
    shapes = [(4096, 4096), (1024, 4096), (256, 4096), (1024, 4096), (4096, 4096)]
    efforts = [0.25, 0.5, 1.0, 0.08, 0.0]
    g = ea.gpu()
    for pinned in (True, False):
        if pinned:
            g.set_tuning(8, 2, 32)
        calls, singles, wants = [], [], []
        for i, ((outDim, inDim), effort) in enumerate(zip(shapes, efforts)):
            W, b, s, p = converted(oracle_cpu, outDim, inDim)
            ew = gpu_weights(ea, W, b, s, p)
            v = make_v(inDim, seed=40 + i, heavy=bool(i & 1))
            vd = devf(v)
            single = torch.zeros(outDim, device=DEV)
            ea.bucketMul(vd, ew, None, single, effort)
            g.eval()
            singles.append((single.cpu().numpy(), g.last_dispatch_count(), g.last_cutoff()))
            wants.append(oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, effort))
            calls.append((vd, ew, None, torch.full((outDim,), float("nan"), device=DEV), effort))
        g.set_split_cutoff(split)
        try:
            for _ in range(2):
                ea.bucketMulGroup(calls)
                g.eval()
                for i, (call, (single, n, cutoff), (want, n_or, cutoff_or)) in enumerate(zip(calls, singles, wants)):
                    got = call[3].cpu().numpy()
                    if pinned:
                        assert got.tobytes() == single.tobytes(), i
                    assert g.last_dispatch_count(i) == n == n_or and g.last_cutoff(i) == cutoff == cutoff_or
                    assert close(got, want) and close(got, single), i
        finally:
            g.set_split_cutoff(False)
            g.set_tuning(0, 0, 0)
    with pytest.raises(ValueError):
        ea.bucketMulGroup(calls * 7)                          # more than 32


@pytest.mark.parametrize("per_cu,n_calls", [(1, 16), (2, 16), (2, 27)])
def test_group_launch_persistent_workgroups(ea, oracle_cpu, per_cu, n_calls):
    """A group with more work items than the chip holds runs as persistent workgroups pulling items from per-XCD queues
    (one workgroup evaluates several items, possibly of different calls): same bits as single launches, launch after
    launch (the queues rewind)."""
    outDim, inDim = 4096, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    g = ea.gpu()
    calls, singles = [], []
    for i in range(n_calls):                         # 16 = what fits the kernel arguments; more go through the device table
        vd = devf(make_v(inDim, seed=60 + i, heavy=bool(i & 1)))
        effort = (0.1, 0.25, 0.5, 1.0)[i % 4]
        single = torch.zeros(outDim, device=DEV)
        ea.bucketMul(vd, ew, None, single, effort)
        g.eval()
        singles.append((single.cpu().numpy(), g.last_dispatch_count(), g.last_cutoff()))
        calls.append((vd, ew, None, torch.full((outDim,), float("nan"), device=DEV), effort))
    try:
        g.set_persistent(per_cu)
        g.set_tuning(8, 1, 64)                      # 4 tiles x 64 slices per call: thousands of items
        for _ in range(3):
            for c in calls:
                c[3].fill_(float("nan"))
            ea.bucketMulGroup(calls)
            g.eval()
            for i, (call, (single, n, cutoff)) in enumerate(zip(calls, singles)):
                assert g.last_dispatch_count(i) == n and g.last_cutoff(i) == cutoff
                assert close(call[3].cpu().numpy(), single), i
    finally:
        g.set_persistent(-1)
        g.set_tuning(0, 0, 0)


def test_plain_and_persistent_forms_of_one_launch_agree(ea, oracle_cpu):
    """Round 6: on a context without lanes an FP16 launch stays a PLAIN grid (lean kernel, every workgroup its own cutoff) up to six items
    per CU -- 20 calls on a 4096 x 11008 matrix: 960 items, the third and fourth round of workgroups placed by the dispatcher -- where it
    used to go persistent from two per CU on.  Same geometry, integer accumulation: the two forms give the SAME BITS, launch after launch,
    and both match the oracle."""
    outDim, inDim, n_calls = 11008, 4096, 20
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    g = ea.gpu()
    hv = [make_v(inDim, seed=300 + i, heavy=bool(i & 1)) for i in range(n_calls)]
    efforts = [(0.1, 0.25, 0.5, 1.0)[i % 4] for i in range(n_calls)]
    calls = [(devf(hv[i]), ew, None, torch.full((outDim,), float("nan"), device=DEV), efforts[i]) for i in range(n_calls)]
    got = {}
    try:
        for form, per_cu in (("heuristic", -1), ("persistent", 2), ("plain", 0)):
            g.set_persistent(per_cu)
            for rep in range(2):
                for c in calls:
                    c[3].fill_(float("nan"))
                ea.bucketMulGroup(calls)
                g.eval()
                outs = [c[3].cpu().numpy().copy() for c in calls]
                meta = [(g.last_dispatch_count(i), g.last_cutoff(i)) for i in range(n_calls)]
                if form in got:
                    assert all(np.array_equal(a, b_) for a, b_ in zip(outs, got[form][0])), form       # launch after launch
                got[form] = (outs, meta)
    finally:
        g.set_persistent(-1)
    for form in ("persistent", "plain"):
        assert got[form][1] == got["heuristic"][1], form
        assert all(np.array_equal(a, b_) for a, b_ in zip(got[form][0], got["heuristic"][0])), form
    for i in (0, 1, 6, 19):
        want, n, cutoff = oracle_cpu.bucket_mul(hv[i], b, s, p, inDim, outDim, efforts[i])
        assert got["heuristic"][1][i] == (n, cutoff), i
        assert close(got["heuristic"][0][i], want), i


def test_row_reuse_policy_changes_no_bit(ea, oracle_cpu):
    """effort_set_row_reuse: the bucket-row stream with the ordinary cache policy instead of nt -- a second copy of the streaming loop,
    chosen per item.  Speed only: a lone call, a plain group and a persistent group give the same bits under either policy (and the
    oracle's dispatch count and cutoff)."""
    outDim, inDim, n_calls = 11008, 4096, 12
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    g = ea.gpu()
    hv = [make_v(inDim, seed=700 + i, heavy=bool(i & 1)) for i in range(n_calls)]
    efforts = [(0.1, 0.25, 0.5, 1.0)[i % 4] for i in range(n_calls)]
    calls = [(devf(hv[i]), ew, None, torch.full((outDim,), float("nan"), device=DEV), efforts[i]) for i in range(n_calls)]
    got = {}
    try:
        for reuse in (False, True, False):
            g.set_row_reuse(reuse)
            for form, per_cu, k in (("lone", -1, 1), ("plain", 0, n_calls), ("persistent", 2, n_calls)):
                g.set_persistent(per_cu)
                for c in calls:
                    c[3].fill_(float("nan"))
                ea.bucketMulGroup(calls[:k])
                g.eval()
                outs = [c[3].cpu().numpy().copy() for c in calls[:k]]
                meta = [(g.last_dispatch_count(i), g.last_cutoff(i)) for i in range(k)]
                if form in got:
                    assert meta == got[form][1], (form, reuse)
                    assert all(np.array_equal(a, b_) for a, b_ in zip(outs, got[form][0])), (form, reuse)
                got[form] = (outs, meta)
    finally:
        g.set_persistent(-1)
        g.set_row_reuse(False)
    for i in (0, 1, 6):
        want, n, cutoff = oracle_cpu.bucket_mul(hv[i], b, s, p, inDim, outDim, efforts[i])
        assert got["plain"][1][i] == (n, cutoff), i
        assert close(got["plain"][0][i], want), i


def test_row_reuse_policy_changes_no_bit_q4(ea, oracle_cpu, q4_case):
    """The Q4 kernels hold the two copies of the streaming loop too (same launch geometry -> same bits, outliers included)."""
    W, L, inDim, outDim = q4_case
    ew = ea.ExpertWeights(dev16(L["buckets"]), devf(L["bucket.stats"]), dev16(L["probes"]), inSize=inDim, outSize=outDim,
                          outliers=devf(L["outliers"]), q4=True)
    g = ea.gpu()
    calls = [(devf(make_v(inDim, seed=720 + k, heavy=k == 2)), ew, None, torch.full((outDim,), float("nan"), device=DEV), effort)
             for k, effort in enumerate((0.25, 0.5, 1.0, 0.1))]
    got = None
    try:
        for reuse in (False, True, False):
            g.set_row_reuse(reuse)
            for c in calls:
                c[3].fill_(float("nan"))
            ea.bucketMulGroup(calls)
            g.eval()
            cur = ([c[3].cpu().numpy().copy() for c in calls], [(g.last_dispatch_count(i), g.last_cutoff(i)) for i in range(len(calls))])
            if got is not None:
                assert cur[1] == got[1], reuse
                assert all(np.array_equal(a, b_) for a, b_ in zip(cur[0], got[0])), reuse
            got = cur
    finally:
        g.set_row_reuse(False)
    assert all(np.isfinite(o).all() for o in got[0])


def test_cutoff_jobs_under_graph_replay(ea, oracle_cpu):
    """A persistent 32-call launch evaluates each call's cutoff once, in a job its items wait for (flag in device memory,
    lowered by the last workgroup out).  Replayed from ONE hipGraph with the inputs changed in place between replays, every
    replay must see the cutoffs of ITS inputs -- a flag or value left over from the previous replay would show here."""
    outDim, inDim, n_calls = 4096, 4096, 32
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    g = ea.gpu()
    vs = [torch.zeros(inDim, device=DEV) for _ in range(n_calls)]
    outs = [torch.zeros(outDim, device=DEV) for _ in range(n_calls)]
    calls = [(vs[i], ew, None, outs[i], (0.1, 0.25, 0.5, 0.8)[i % 4]) for i in range(n_calls)]
    try:
        g.set_tuning(8, 1, 32)                     # 4 tiles x 32 slices per call = 4096 items: a persistent launch
        ea.bucketMulGroup(calls)                   # warm
        g.eval()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            ea.bucketMulGroup(calls)
        g._bind_stream()
    finally:
        g.set_tuning(0, 0, 0)
    for rep in range(6):
        hv = [make_v(inDim, seed=1000 * rep + i, heavy=bool((i + rep) & 1)) * (1.0 + rep) for i in range(n_calls)]
        for i in range(n_calls):
            vs[i].copy_(devf(hv[i]))
            outs[i].fill_(float("nan"))
        graph.replay()
        g.eval()
        for i in (0, 5, 17, 31, rep):
            want, n, cutoff = oracle_cpu.bucket_mul(hv[i], b, s, p, inDim, outDim, calls[i][4])
            assert g.last_cutoff(i) == cutoff and g.last_dispatch_count(i) == n, (rep, i)
            assert close(outs[i].cpu().numpy(), want), (rep, i)


def test_group_launch_q4(ea, oracle_cpu, q4_case):
    W, L, inDim, outDim = q4_case
    ews = [ea.ExpertWeights(dev16(L["buckets"]), devf(L["bucket.stats"]), dev16(L["probes"]), inSize=inDim, outSize=outDim,
                            outliers=devf(L["outliers"]) if k != 1 else None, q4=True) for k in range(3)]
    calls, wants = [], []
    for k, effort in enumerate((0.25, 0.5, 1.0)):
        v = make_v(inDim, seed=50 + k, heavy=k == 2)
        wants.append(oracle_cpu.bucket_mul_q4(v, L["buckets"], L["bucket.stats"], L["probes"], L["outliers"] if k != 1 else None,
                                              inDim, outDim, effort))
        calls.append((devf(v), ews[k], None, torch.full((outDim,), float("nan"), device=DEV), effort))
    ea.bucketMulGroup(calls)
    ea.gpu().eval()
    for k, (call, (want, n, cutoff)) in enumerate(zip(calls, wants)):
        assert ea.gpu().last_dispatch_count(k) == n and ea.gpu().last_cutoff(k) == cutoff
        assert close(call[3].cpu().numpy(), want), k
    with pytest.raises(ValueError):
        ea.bucketMulGroup([calls[0], (calls[0][0],) + (gpu_weights(ea, *converted(oracle_cpu, 256, 4096)),) + calls[0][2:]])   # mixed kinds


@pytest.mark.parametrize("inDim,outDim", [(4096, 4160), (16448, 4096)])
def test_q4_outlier_tables(ea, oracle_cpu, inDim, outDim):
    """calcOutliers (bucketMulQ4.metal:13-21) on tables the converter never writes but the format allows: every entry
    on one output, duplicates of one (input, output) pair, outputs without entries, entries on the first and the last
    output, shuffled table order -- over a synthetic Q4 bundle (any nibble pattern is a valid bucket word).  4160
    outputs end in a ragged tile ((outDim/16) % 4 == 0 is the reference's own precondition, bucketMul.swift:76); 16448
    inputs exceed the LDS copy of v (gathered from memory)."""
    rng = np.random.default_rng(inDim + outDim)
    rows, cols = inDim * 8, outDim // 32
    buckets = rng.integers(0, 65536, size=(rows, cols), dtype=np.uint16)
    mean = np.abs(rng.normal(0, 0.02, size=rows)).astype(np.float32)
    stats = np.stack([mean, mean], axis=1)
    probes = rng.normal(0, 0.02, size=4096).astype(np.float16)

    def table(n, outs):
        t = np.zeros((n, 4), np.float32)
        t[:, 0] = rng.normal(0, 0.3, size=n).astype(np.float16)       # values of an f16 matrix (q4_draft.py:58-67)
        t[:, 1] = rng.integers(0, inDim, size=n)
        t[:, 2] = outs if not np.isscalar(outs) else np.full(n, outs)
        return t
    tables = {
        "one output": table(5000, 77),
        "first and last": np.concatenate([table(300, 0), table(300, outDim - 1)]),
        "duplicates": np.repeat(table(64, rng.integers(0, outDim, size=64)), 7, axis=0),
        "sparse, shuffled": table(20000, rng.choice(np.arange(0, outDim, 3), size=20000)),
        "dense": table(200000, rng.integers(0, outDim, size=200000)),
    }
    v = make_v(inDim, seed=8, heavy=True)
    vd = devf(v)
    g = ea.gpu()
    for name, ol in tables.items():
        ew = ea.ExpertWeights(dev16(buckets), devf(stats), dev16(probes), inSize=inDim, outSize=outDim, outliers=devf(ol), q4=True)
        for tune, effort in (((0, 0, 0), 0.25), ((8, 1, 16), 0.1), ((8, 2, 24), 0.5)):
            want, n, cutoff = oracle_cpu.bucket_mul_q4(v, buckets, stats, probes, ol, inDim, outDim, effort)
            out = torch.full((outDim,), float("nan"), device=DEV)
            try:
                g.set_tuning(*tune)
                ea.bucketMulQ4(vd, ew, None, out, effort)
                g.eval()
            finally:
                g.set_tuning(0, 0, 0)
            assert g.last_dispatch_count() == n and g.last_cutoff() == cutoff
            assert close(out.cpu().numpy(), want), (name, tune, effort)


# ---------------------------------------------------------------- on-disk bucket format + model converter driver
def test_model_file_roundtrip(ea, oracle_cpu, tmp_path):
    """convertMistral on the GPU (one synthetic layer: hidden 4096, kv 256, ffn 1024) -> safetensors shards + index ->
    ExpertWeights loaded back by name (loader.swift:60-167) -> bucketMul equals the oracle on the same matrix; loading
    with percentLoad < 16 reads only the first rank planes; the FFN name pattern stacks experts."""
    from effort_amd import bucketfile as bf
    hidden, kv, ffn = 4096, 256, 1024
    shapes = {"self_attn.q_proj": (hidden, hidden), "self_attn.k_proj": (kv, hidden), "self_attn.v_proj": (kv, hidden),
              "self_attn.o_proj": (hidden, hidden), "mlp.gate_proj": (ffn, hidden), "mlp.up_proj": (ffn, hidden),
              "mlp.down_proj": (hidden, 4096)}          # (down_proj kept square: inDim >= 4096 is a bucketize precondition)
    src = {"model.norm.weight": torch.ones(hidden).half(), "lm_head.weight": torch.zeros(8, hidden).half(),
           "model.embed_tokens.weight": torch.zeros(8, hidden).half(),
           "model.layers.0.input_layernorm.weight": torch.ones(hidden).half(),
           "model.layers.0.post_attention_layernorm.weight": torch.ones(hidden).half()}
    mats = {}
    for k, (o, i) in shapes.items():
        mats[k] = make_w(o, i, seed=70 + len(mats))
        src[f"model.layers.0.{k}.weight"] = torch.from_numpy(mats[k])
    saver = bf.convertMistral(src, bf.TensorSaver(str(tmp_path), "buckets-FP16"), numLayers=1)
    saver.save()
    L = bf.TensorLoader(str(tmp_path), "buckets-FP16")
    v = make_v(hidden, seed=3)
    vd = devf(v)
    # attention matrix by element name; layout identical to the oracle's conversion
    ew = bf.loadExpertWeights(L, "layers.0.attention.wk")
    b, s, p, _ = oracle_cpu.convert_fp16(mats["self_attn.k_proj"])
    assert ew.buckets.cpu().numpy().view(np.uint16).tobytes() == b.view(np.uint16).tobytes()
    assert (ew.inSize, ew.outSize, ew.percentLoad) == (hidden, kv, 16) and ew.core is not None
    out = torch.zeros(kv, device=DEV)
    ea.expertMul(vd, ew, out, 0.3)
    ea.gpu().eval()
    want, n, _ = oracle_cpu.bucket_mul(v, b, s, p, hidden, kv, 0.3)
    assert ea.gpu().last_dispatch_count() == n and close(out.cpu().numpy(), want)
    # FFN matrix by (prefix, wId), percentLoad 16 and 4
    b, s, p, _ = oracle_cpu.convert_fp16(mats["mlp.gate_proj"])
    for pl in (16, 4):
        ew = bf.loadExpertWeights(L, "layers.0.feed_forward.experts.", "w1", inDim=hidden, outDim=ffn, numExperts=1, percentLoad=pl)
        assert ew.buckets.shape == (1, hidden * pl, ffn // 16)
        out = torch.zeros(ffn, device=DEV)
        ea.bucketMul(vd, ew, None, out, 0.5)
        ea.gpu().eval()
        want, n, _ = oracle_cpu.bucket_mul(v, b[:hidden * pl], s[:hidden * pl], p, hidden, ffn, 0.5, percentLoad=pl)
        assert ea.gpu().last_dispatch_count() == n and close(out.cpu().numpy(), want), pl
    with pytest.raises(KeyError):
        bf.loadExpertWeights(L, "layers.0.attention.nope", inDim=hidden, outDim=kv)


def test_ragged_input_dimension(ea, oracle_cpu):
    """inDim that no slice size divides (4500 = 35 slices of 128 + 20 rows; 8 of 512 + 404): the last slice is partial, in
    lone launches and in groups (fatter slices), FP16 with and without stacked experts."""
    outDim, inDim = 256, 4500
    mats = [converted(oracle_cpu, outDim, inDim, seed=200 + e) for e in range(2)]
    ews = [gpu_weights(ea, *m) for m in mats]
    stacked = ea.ExpertWeights.stack(ews)
    B = np.concatenate([m[1] for m in mats]); S = np.concatenate([m[2] for m in mats]); P = np.concatenate([m[3] for m in mats])
    g = ea.gpu()
    calls, wants = [], []
    for i in range(9):
        v = make_v(inDim, seed=300 + i, heavy=bool(i & 1))
        e, effort = i & 1, (0.1, 0.3, 0.7, 1.0)[i % 4]
        want, n, cutoff = oracle_cpu.bucket_mul(v, B, S, P, inDim, outDim, effort, expNo=e)
        expNo = torch.tensor([e], dtype=torch.int32, device=DEV)
        out = torch.full((outDim,), float("nan"), device=DEV)
        ea.bucketMul(devf(v), stacked, expNo, out, effort)
        g.eval()
        assert g.last_dispatch_count() == n and g.last_cutoff() == cutoff and close(out.cpu().numpy(), want), i
        calls.append((devf(v), stacked, expNo, torch.full((outDim,), float("nan"), device=DEV), effort))
        wants.append((want, n, cutoff))
    ea.bucketMulGroup(calls)
    g.eval()
    for i, (call, (want, n, cutoff)) in enumerate(zip(calls, wants)):
        assert g.last_dispatch_count(i) == n and g.last_cutoff(i) == cutoff and close(call[3].cpu().numpy(), want), i


def test_model_file_roundtrip_q4(ea, oracle_cpu, tmp_path):
    """q4_convert.py's flow on the GPU: convertMistral(q4=True) -> shards named like the reference's Q4 model -> bundles
    loaded back (stats f32x2, outliers) -> bucketMulQ4 equals the oracle run on the numpy restatement of q4_draft.convert."""
    from effort_amd import bucketfile as bf
    from oracle import q4_layout
    hidden = 4096
    src = {"model.norm.weight": torch.ones(hidden).half(), "lm_head.weight": torch.zeros(8, hidden).half(),
           "model.embed_tokens.weight": torch.zeros(8, hidden).half(),
           "model.layers.0.input_layernorm.weight": torch.ones(hidden).half(),
           "model.layers.0.post_attention_layernorm.weight": torch.ones(hidden).half()}
    mats = {}
    for k in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
        mats[k] = make_w(hidden, hidden, seed=90 + len(mats))
        src[f"model.layers.0.{k}.weight"] = torch.from_numpy(mats[k])
    saver = bf.convertMistral(src, bf.TensorSaver(str(tmp_path), "model", pad_total=False), numLayers=1, q4=True)
    saver.save()
    L = bf.TensorLoader(str(tmp_path), "model")
    assert not L.hasTensor("layers.0.attention.wk.buckets") and L.hasTensor("layers.0.attention.wk.core")     # q4_convert.py:57
    v = make_v(hidden, seed=8)
    vd = devf(v)
    for name, key in (("layers.0.attention.wq", "self_attn.q_proj"), ("layers.0.feed_forward.experts.0.w3", "mlp.up_proj")):
        ew = bf.loadExpertWeights(L, name, q4=True)
        want_layout = q4_layout.convert(np.ascontiguousarray(mats[key].T))
        assert ew.q4 and ew.percentLoad == 8 and ew.outliers.shape == want_layout["outliers"].shape
        assert ew.buckets.cpu().numpy().view(np.uint16).tobytes() == want_layout["buckets"].view(np.uint16).tobytes()
        out = torch.full((hidden,), float("nan"), device=DEV)
        ea.expertMul(vd, ew, out, 0.3)
        ea.gpu().eval()
        want, n, cutoff = oracle_cpu.bucket_mul_q4(v, want_layout["buckets"], want_layout["bucket.stats"], want_layout["probes"],
                                                   want_layout["outliers"], hidden, hidden, 0.3)
        assert ea.gpu().last_dispatch_count() == n and ea.gpu().last_cutoff() == cutoff and close(out.cpu().numpy(), want), name


def test_fused_prologues_and_residual(ea, oracle_cpu):
    """effort_bucketmul_group_fused: the input derived in the launch (silu gate / rmsNorm) selects exactly the rows, and
    gives the output (+ residual), that materialising it first with the glue kernels gives."""
    import ctypes as C
    from effort_amd import _lib
    outDim, inDim = 1024, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    g = ea.gpu()
    lib = _lib.lib()
    P = lambda t: C.c_void_p(t.data_ptr())                                               # noqa: E731
    x1, x3 = devf(make_v(inDim, seed=71)), devf(make_v(inDim, seed=72, heavy=True))
    hvec = devf(make_v(inDim, seed=73, heavy=True))
    wn = torch.from_numpy((1 + 0.1 * np.random.default_rng(5).standard_normal(inDim)).astype(np.float16)).to(DEV)
    resid = devf(make_v(outDim, seed=74))
    # --- silu gate: materialised by effort_silu_mul, then a plain multiply
    x2 = torch.zeros(inDim, device=DEV)
    g._bind_stream()
    g.check(lib.effort_silu_mul(g.ctx, P(x1), P(x3), P(x2), inDim), "silu")
    plain = torch.zeros(outDim, device=DEV)
    ea.bucketMul(x2, ew, None, plain, 0.3)
    g.eval()
    n0, c0 = g.last_dispatch_count(), g.last_cutoff()
    want, n_or, c_or = oracle_cpu.bucket_mul(x2.cpu().numpy(), b, s, p, inDim, outDim, 0.3)
    assert n0 == n_or and c0 == c_or
    fused = resid.clone()
    ea.bucketMulGroup([(x1, ew, None, fused, 0.3, {"gate": x3, "resid": fused})])          # out = resid + product, in place
    g.eval()
    assert g.last_dispatch_count() == n0 and g.last_cutoff() == c0
    assert torch.equal(fused, resid + plain)
    # --- rmsNorm: materialised by effort_add_rmsnorm_mul; the fused prologue sums the squares in that kernel's order, so the
    #     input -- hence cutoff, selection and product -- is the same to the bit
    hn = torch.zeros(inDim, device=DEV)
    hc = hvec.clone()
    g.check(lib.effort_add_rmsnorm_mul(g.ctx, P(hc), None, P(wn), P(hn), inDim), "rmsnorm")
    plain2 = torch.zeros(outDim, device=DEV)
    ea.bucketMul(hn, ew, None, plain2, 0.5)
    g.eval()
    n1, c1 = g.last_dispatch_count(), g.last_cutoff()
    want2, n2_or, c2_or = oracle_cpu.bucket_mul(hn.cpu().numpy(), b, s, p, inDim, outDim, 0.5)
    assert n1 == n2_or and c1 == c2_or
    fused2 = torch.zeros(outDim, device=DEV)
    ea.bucketMulGroup([(hvec, ew, None, fused2, 0.5, {"norm": wn}), (x1, ew, None, plain, 0.3)])   # mixed with a plain call
    g.eval()
    assert g.last_dispatch_count(0) == n1 and g.last_cutoff(0) == c1     # exact: same input bits
    assert torch.equal(fused2, plain2)
    want1, n1_or, c1_or = oracle_cpu.bucket_mul(x1.cpu().numpy(), b, s, p, inDim, outDim, 0.3)     # the plain call sharing the launch
    assert g.last_dispatch_count(1) == n1_or and g.last_cutoff(1) == c1_or and close(plain.cpu().numpy(), want1)
    for tune in ((4, 2, 0), (16, 1, 0), (2, 4, 0)):                       # 256 / 1024 / 128 threads standing for that kernel's 1024
        g.set_tuning(*tune)
        try:
            f3 = torch.zeros(outDim, device=DEV)
            ea.bucketMulGroup([(hvec, ew, None, f3, 0.5, {"norm": wn})])
            g.eval()
            assert g.last_dispatch_count(0) == n1 and g.last_cutoff(0) == c1 and close(f3.cpu().numpy(), want2), tune
        finally:
            g.set_tuning(0, 0, 0)
    with pytest.raises(ValueError):
        ea.bucketMulGroup([(x1, ew, None, fused, 0.3, {"gate": x3, "norm": wn})])


def test_randomized_groups_against_oracle(ea, oracle_cpu):
    """Seeded sweep: random shapes (ragged inDim, outDim from half a tile to several), efforts, percentLoad, group sizes and
    tunings -- selection exact, outputs within the bar, for every call of every launch."""
    rng = np.random.default_rng(20240807)
    shapes = [(64, 4096), (512, 4096), (1024, 4500), (2048, 4096), (4160, 4608)]    # outDim divides 4096 or exceeds it (convert.swift:210-215)
    bank = {sh: converted(oracle_cpu, sh[0], sh[1], seed=500 + i) for i, sh in enumerate(shapes)}
    g = ea.gpu()
    try:
        for trial in range(6):
            n = int(rng.integers(1, 12))
            tune = [(0, 0, 0), (8, 1, 0), (16, 2, 0), (4, 4, 0), (8, 2, 24), (8, 4, 8)][trial]
            g.set_tuning(*tune)
            calls, wants = [], []
            for i in range(n):
                outDim, inDim = shapes[int(rng.integers(len(shapes)))]
                W, b, s, p = bank[(outDim, inDim)]
                pl = int(rng.choice([16, 16, 8, 3]))
                rows = inDim * pl
                ew = gpu_weights(ea, W, b[:rows], s[:rows], p, percentLoad=pl)
                effort = float(rng.choice([0.0, 0.03, 0.25, 0.6, 1.0]))
                v = make_v(inDim, seed=int(rng.integers(1 << 30)), heavy=bool(rng.integers(2)))
                if rng.integers(4) == 0:
                    v[rng.integers(inDim, size=40)] = 0.0
                wants.append(oracle_cpu.bucket_mul(v, b[:rows], s[:rows], p, inDim, outDim, effort, percentLoad=pl))
                calls.append((devf(v), ew, None, torch.full((outDim,), float("nan"), device=DEV), effort))
            ea.bucketMulGroup(calls)
            g.eval()
            for i, (call, (want, cnt, cutoff)) in enumerate(zip(calls, wants)):
                assert g.last_dispatch_count(i) == cnt and g.last_cutoff(i) == cutoff, (trial, i)
                assert close(call[3].cpu().numpy(), want), (trial, i, tune)
    finally:
        g.set_tuning(0, 0, 0)


# ---------------------------------------------------------------- BASELINE.json configs at full size
@pytest.fixture(scope="module")
def q4_11008():
    from oracle import q4_layout
    inDim, outDim = 4096, 11008
    W = make_w(outDim, inDim, seed=77)
    return W, q4_layout.convert(np.ascontiguousarray(W.T)), inDim, outDim


def test_q4_config3_full_size(ea, oracle_cpu, q4_11008):
    """BASELINE.json configs[2]: bucketMulQ4 at 4096 x 11008 with the converter's own 2 % outlier table
    (oracle/q4_layout.convert = q4_draft.convert), efforts 0.1 .. 1.0, as a lone call and as 16 calls of one launch (the
    bench geometry of the Q4 line), against eo_bucketmul_q4 (bucketMulQ4.metal:13-92)."""
    W, L, inDim, outDim = q4_11008
    assert abs(L["outliers"].shape[0] / (inDim * outDim) - 0.02) < 0.001
    ew = ea.ExpertWeights(dev16(L["buckets"]), devf(L["bucket.stats"]), dev16(L["probes"]), inSize=inDim, outSize=outDim,
                          outliers=devf(L["outliers"]), q4=True)
    g = ea.gpu()
    efforts = (0.1, 0.25, 0.5, 1.0)
    v = make_v(inDim, seed=11)
    vd = devf(v)
    out = torch.full((outDim,), float("nan"), device=DEV)
    for effort in efforts:                                                   # lone calls
        want, n, cutoff = oracle_cpu.bucket_mul_q4(v, L["buckets"], L["bucket.stats"], L["probes"], L["outliers"], inDim, outDim, effort)
        ea.bucketMulQ4(vd, ew, None, out, effort)
        g.eval()
        assert g.last_dispatch_count() == n and g.last_cutoff() == cutoff, effort
        assert close(out.cpu().numpy(), want), effort
    hv = [make_v(inDim, seed=300 + i, heavy=bool(i & 1)) for i in range(16)]       # 16 calls per launch
    outs = [torch.full((outDim,), float("nan"), device=DEV) for _ in range(16)]
    ea.bucketMulGroup([(devf(hv[i]), ew, None, outs[i], efforts[i % 4]) for i in range(16)])
    g.eval()
    for i in range(16):
        want, n, cutoff = oracle_cpu.bucket_mul_q4(hv[i], L["buckets"], L["bucket.stats"], L["probes"], L["outliers"], inDim, outDim, efforts[i % 4])
        assert g.last_dispatch_count(i) == n and g.last_cutoff(i) == cutoff, i
        assert close(outs[i].cpu().numpy(), want), i


def test_bench_geometry_32_matrices_from_graph(ea, oracle_cpu):
    """The bench's 32-call launch: 32 DISTINCT converted 4096 x 11008 matrices, one call each at 25 % effort, one heuristic
    group (persistent workgroups, cutoff jobs), replayed from a hipGraph -- every output, dispatch count and cutoff against
    the oracle, on a second replay with the input changed in place."""
    _bench_geometry(ea, oracle_cpu, ea.gpu(), 4096, 11008, 32)


def _bench_geometry(ea, oracle_cpu, g, inDim, outDim, n_calls):
    gen = torch.Generator(device=DEV)
    ews, host = [], []
    for k in range(n_calls):
        gen.manual_seed(9000 + k)
        W = (torch.randn((outDim, inDim), generator=gen, device=DEV, dtype=torch.float32) * 0.02).to(torch.float16)
        ew = ea.ExpertWeights.from_core(W)                                    # product converter (bit-exact vs the oracle: test_gpu_converter_bit_exact)
        ew.core = None
        ews.append(ew)
        host.append((ew.buckets[0].cpu().numpy().view(np.uint16), ew.stats[0].cpu().numpy().view(np.uint16), ew.probes[0].cpu().numpy().view(np.uint16)))
    vdev = torch.zeros(inDim, device=DEV)
    outs = [torch.zeros(outDim, device=DEV) for _ in range(n_calls)]
    calls = [(vdev, ews[i], None, outs[i], 0.25) for i in range(n_calls)]
    ea.bucketMulGroup(calls)                                                  # warm
    g.eval()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ea.bucketMulGroup(calls)
    g._bind_stream()
    for rep, heavy in enumerate((False, True)):
        v = make_v(inDim, seed=42 + rep, heavy=heavy)
        vdev.copy_(devf(v))
        for o in outs:
            o.fill_(float("nan"))
        graph.replay()
        g.eval()
        for i in range(n_calls):
            want, n, cutoff = oracle_cpu.bucket_mul(v, *host[i], inDim, outDim, 0.25)
            assert g.last_cutoff(i) == cutoff and g.last_dispatch_count(i) == n, (rep, i)
            assert close(outs[i].cpu().numpy(), want), (rep, i)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_column_shards_of_the_baseline_shapes(ea, oracle_cpu, q4_case, q4_11008, world):
    """BASELINE.json configs[3] on one device: ShardedExpertWeights.from_full for every rank of a world of 2 / 4 / 8, for
    4096 x 11008 (86 bucket columns per rank at world 8: a ragged tile), 4096 x 4096, 4096 x 1024 (Wk / Wv: 8 columns per rank
    at world 8) and 14336 x 4096 (W2), FP16, and Q4 with outliers --
    every rank's dispatch count and cutoff are the full call's, and the concatenated outputs are the full product."""
    from effort_amd.sharded import ShardedExpertWeights
    g = ea.gpu()
    cases = []
    for outDim, inD in ((11008, 4096), (4096, 4096), (1024, 4096), (4096, 14336)):          # W1/W3-like, Wq/Wo, Wk/Wv (8 columns per rank at world 8), W2
        W, b, s, p = converted(oracle_cpu, outDim, inD)
        cases.append(("fp16", outDim, gpu_weights(ea, W, b, s, p), lambda v, e, b=b, s=s, p=p, outDim=outDim, inD=inD: oracle_cpu.bucket_mul(v, b, s, p, inD, outDim, e)))
    for (W, L, inDim, outDim) in (q4_11008, q4_case):
        ew = ea.ExpertWeights(dev16(L["buckets"]), devf(L["bucket.stats"]), dev16(L["probes"]), inSize=inDim, outSize=outDim,
                              outliers=devf(L["outliers"]), q4=True)
        cases.append(("q4", outDim, ew, lambda v, e, L=L, inDim=inDim, outDim=outDim: oracle_cpu.bucket_mul_q4(
            v, L["buckets"], L["bucket.stats"], L["probes"], L["outliers"], inDim, outDim, e)))
    vin = {4096: make_v(4096, seed=91, heavy=True), 14336: make_v(14336, seed=92, heavy=True)}
    vdev = {k: devf(x) for k, x in vin.items()}
    for kind, outDim, full, oracle in cases:
        v, vd = vin[full.inSize], vdev[full.inSize]
        want, n, cutoff = oracle(v, 0.5)
        parts = []
        for r in range(world):
            sh = ShardedExpertWeights.from_full(full, r, world)
            assert sh.localOut * world == outDim and sh.local.outSize == sh.localOut
            o = torch.full((sh.localOut,), float("nan"), device=DEV)
            ea.expertMul(vd, sh.local, o, 0.5)
            g.eval()
            assert g.last_dispatch_count() == n and g.last_cutoff() == cutoff, (kind, outDim, world, r)
            parts.append(o.cpu().numpy())
        assert close(np.concatenate(parts), want), (kind, outDim, world)
    # the shards of several matrices sharing v go out as ONE grouped launch per rank (shardedExpertMulGroup's default multiply)
    from effort_amd.sharded import _default_mul_group
    fp = [c for c in cases if c[0] == "fp16" and c[2].inSize == 4096]
    v, vd = vin[4096], vdev[4096]
    for r in (0, world - 1):
        shs = [ShardedExpertWeights.from_full(c[2], r, world) for c in fp]
        outs = [torch.full((sh.localOut,), float("nan"), device=DEV) for sh in shs]
        _default_mul_group(vd, [sh.local for sh in shs], outs, 0.5, None)
        g.eval()
        for (kind, outDim, full, oracle), sh, o in zip(fp, shs, outs):
            want, n, cutoff = oracle(v, 0.5)
            assert close(o.cpu().numpy(), want[r * sh.localOut:(r + 1) * sh.localOut]), (outDim, world, r)


def test_aligned_row_pitch_is_bit_identical(ea, oracle_cpu, q4_11008):
    """effort_weights_align_rows: the handle's own copy of the buckets with rows on 128-byte lines (1376 -> 1408 bytes for
    11008 outputs; Q4: 688 -> 768) gives the same bits as the converter's pitch -- lone calls, a grouped launch, and after
    a refresh of rewritten weights."""
    outDim, inDim = 11008, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    plain, padded = gpu_weights(ea, W, b, s, p), gpu_weights(ea, W, b, s, p)
    assert padded.align_rows() == 1408 and padded.align_rows() == 1408          # idempotent
    g = ea.gpu()
    v = make_v(inDim, seed=17, heavy=True)
    vd = devf(v)
    for effort in (0.1, 0.25, 1.0):
        o1, o2 = torch.zeros(outDim, device=DEV), torch.zeros(outDim, device=DEV)
        ea.bucketMul(vd, plain, None, o1, effort)
        g.eval()
        n1, c1 = g.last_dispatch_count(), g.last_cutoff()
        ea.bucketMul(vd, padded, None, o2, effort)
        g.eval()
        assert (g.last_dispatch_count(), g.last_cutoff()) == (n1, c1) and torch.equal(o1, o2), effort
    outs = [torch.zeros(outDim, device=DEV) for _ in range(12)]
    ea.bucketMulGroup([(vd, padded if i & 1 else plain, None, outs[i], 0.25) for i in range(12)])
    g.eval()
    want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, 0.25)
    for i in range(12):                              # (the last two calls of this launch are cut into thinner slices, api.hip thinFrom: their own rounding grid)
        assert g.last_dispatch_count(i) == n and torch.equal(outs[i], outs[0 if i < 10 else 10]) and close(outs[i].cpu().numpy(), want), i
    o2 = torch.zeros(outDim, device=DEV)
    ea.bucketMul(vd, padded, None, o2, 0.25)
    g.eval()
    # rewritten weights reach the copy through refresh
    padded.buckets.copy_(dev16(np.ascontiguousarray(b).view(np.uint16) ^ np.uint16(0x8000)).reshape(padded.buckets.shape))   # every weight negated
    padded.refresh()
    o3 = torch.zeros(outDim, device=DEV)
    ea.bucketMul(vd, padded, None, o3, 0.25)
    g.eval()
    assert close(o3.cpu().numpy(), -want) and not torch.equal(o3, o2)          # (exactly -o2 up to the rounding mode's ties: floor(x + 1/2))
    # Q4
    Wq, L, inDim, outDim = q4_11008
    mk = lambda: ea.ExpertWeights(dev16(L["buckets"]), devf(L["bucket.stats"]), dev16(L["probes"]), inSize=inDim, outSize=outDim,   # noqa: E731
                                  outliers=devf(L["outliers"]), q4=True)
    q1, q2 = mk(), mk()
    assert q2.align_rows() == 768
    o1, o2 = torch.zeros(outDim, device=DEV), torch.zeros(outDim, device=DEV)
    ea.bucketMulQ4(vd, q1, None, o1, 0.25)
    ea.bucketMulQ4(vd, q2, None, o2, 0.25)
    g.eval()
    assert torch.equal(o1, o2)


def test_launches_in_flight_on_several_streams(ea, oracle_cpu):
    """The bench's job shape: ONE hipGraph whose steps run on four HIP streams, one effort_ctx (own scratch: slabs, tickets,
    queues, cutoffs) per stream, up to four grouped launches in flight.  Every step has its own input vector; every output
    of every step, its dispatch counts and cutoffs are checked against the oracle -- shared scratch between contexts, or a
    queue / flag left over by a launch that overlapped another, would show here."""
    outDim, inDim, n_mats, S, steps = 4096, 4096, 12, 4, 8
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    plain, padded = gpu_weights(ea, W, b, s, p), gpu_weights(ea, W, b, s, p)
    padded.align_rows()
    ews = [padded if k & 1 else plain for k in range(n_mats)]
    ctxs = [ea.gpu(0)] + [ea.Gpu(0) for _ in range(S - 1)]
    streams = [None] + [torch.cuda.Stream() for _ in range(S - 1)]
    vs = [torch.zeros(inDim, device=DEV) for _ in range(steps)]
    outs = [[torch.zeros(outDim, device=DEV) for _ in range(n_mats)] for _ in range(steps)]
    efforts = (0.1, 0.25, 0.5)
    for c in ctxs:
        c.set_tuning(8, 1, 32)                         # 4 tiles x 32 slices per call: persistent launches with cutoff jobs

    def step(i):
        ea.bucketMulGroup([(vs[i], ews[k], None, outs[i][k], efforts[(i + k) % 3]) for k in range(n_mats)], gpu=ctxs[i % S])

    # (fork / join through events that outlive the graph, not Stream.wait_stream: its temporary events are destroyed inside the
    #  capture they took part in, which this runtime does not survive reliably -- tools/lab/graph_event_repro.py)
    fork_ev, join_ev = torch.cuda.Event(), [torch.cuda.Event() for _ in range(S)]
    _KEEP_ALIVE.extend([fork_ev] + join_ev + streams[1:])

    def enqueue():
        s0 = torch.cuda.current_stream()
        fork_ev.record(s0)
        for k in range(1, S):
            streams[k].wait_event(fork_ev)
        for i in range(steps):
            if i % S == 0:
                step(i)
            else:
                with torch.cuda.stream(streams[i % S]):
                    step(i)
        for k in range(1, S):
            join_ev[k].record(streams[k])
            s0.wait_event(join_ev[k])
    try:
        enqueue()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            enqueue()
        for c in ctxs:
            c._bind_stream()
    finally:
        for c in ctxs:
            c.set_tuning(0, 0, 0)
    for rep in range(2):
        hv = [make_v(inDim, seed=7000 + 10 * rep + i, heavy=bool(i & 1)) for i in range(steps)]
        for i in range(steps):
            vs[i].copy_(devf(hv[i]))
            for o in outs[i]:
                o.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        for i in range(steps):
            for k in range(n_mats):
                want, n, cutoff = oracle_cpu.bucket_mul(hv[i], b, s, p, inDim, outDim, efforts[(i + k) % 3])
                assert close(outs[i][k].cpu().numpy(), want), (rep, i, k)
                if i >= steps - S:                 # the contexts remember their LAST launch
                    assert ctxs[i % S].last_dispatch_count(k) == n and ctxs[i % S].last_cutoff(k) == cutoff, (rep, i, k)
    for c in ctxs[1:]:
        c.close()


def test_hip_q4_converter(ea, q4_case, q4_11008):
    """effort_convert_q4 (HIP kernels behind the C ABI: csrc/convert_q4.hip): the reference-generated fixtures
    (tests/golden/q4_*.npz: what q4_draft.convert returned), the full-size layouts of the oracle's restatement (4096x4096,
    4096x11008: bit-identical buckets, stats, probes, outlier table in the same order), numpy's pairwise summation at the
    row lengths that matter, ties at the 2 % boundary and signed zeros; and the tensor-op host mirror agrees."""
    import glob
    import os
    from effort_amd import q4
    for path in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "q4_*.npz"))):
        gd = np.load(path)
        out = ea.q4_convert(torch.from_numpy(gd["core2"]).to(DEV))
        assert np.array_equal(out["buckets"].cpu().numpy().view(np.uint16), gd["buckets_u16"]), path
        assert np.array_equal(out["bucket.stats"].cpu().numpy(), gd["bucket_stats"]), path
        assert np.array_equal(out["probes"].cpu().numpy().view(np.uint16), gd["probes"].view(np.uint16)), path
        a, b = out["outliers"].cpu().numpy(), gd["outliers"]                 # same set (the reference's order is its unstable sort's)
        assert np.array_equal(a[np.lexsort((a[:, 2], a[:, 1]))], b[np.lexsort((b[:, 2], b[:, 1]))]), path
    for W, L, inDim, outDim in (q4_case, q4_11008):
        out = ea.q4_convert(dev16(np.ascontiguousarray(W.T)).view(torch.float16))
        assert out["buckets"].cpu().numpy().view(np.uint16).tobytes() == L["buckets"].view(np.uint16).tobytes()
        assert out["bucket.stats"].cpu().numpy().tobytes() == L["bucket.stats"].tobytes()
        assert out["probes"].cpu().numpy().view(np.uint16).tobytes() == L["probes"].view(np.uint16)[:min(inDim, outDim)].tobytes()
        assert out["outliers"].cpu().numpy().tobytes() == L["outliers"].tobytes()
    # many ties at the boundary (values from a 16-level grid), -0.0 among them, ragged pairwise lengths (nb = 36 .. 1500)
    rng = np.random.default_rng(12)
    from oracle import q4_layout
    for inDim, outDim in ((96, 288), (64, 1056), (40, 12000)):
        core2 = (rng.integers(-8, 9, size=(inDim, outDim)) / 16).astype(np.float16)
        core2[rng.integers(0, inDim, 50), rng.integers(0, outDim, 50)] = np.float16(-0.0)
        L = q4_layout.convert(core2)
        for out in (ea.q4_convert(torch.from_numpy(core2).to(DEV)), q4.convert_tensor_ops(torch.from_numpy(core2).to(DEV))):
            assert out["buckets"].cpu().numpy().view(np.uint16).tobytes() == L["buckets"].view(np.uint16).tobytes(), (inDim, outDim)
            assert out["bucket.stats"].cpu().numpy().tobytes() == L["bucket.stats"].tobytes(), (inDim, outDim)
            assert out["outliers"].cpu().numpy().tobytes() == L["outliers"].tobytes(), (inDim, outDim)
            assert out["probes"].cpu().numpy().view(np.uint16).tobytes() == L["probes"].view(np.uint16).tobytes()


@pytest.mark.parametrize("kind", ["gauss", "heavy", "zeros", "few", "tiny", "ties"])
def test_row_selection_across_exits_and_ties(ea, oracle_cpu, kind):
    """Row selection through the fused multiply for inputs that leave the reference's bisection through every exit -- the count
    exits (gauss, heavy), bounds closer than 1e-5 / the fixed point (tiny values), zeros (a count table that starts at the smallest
    nonzero value), few distinct values and exact ties at the threshold (whole rank planes share the cutoff's own score) -- must be
    EXACT (dispatch.size, cutoff bits) and the product within the bar, for a lone call and as a group of three on one input.
    (Written for round 5's speculative selection -- rows selected with a bracket of the cutoff before it is known, the bisection
    run under the stream, open rows decided afterwards: correct on all of these and 5 us SLOWER per lone call, branch
    `speculative-selection`, DESIGN.md 8 -- and kept: it is the one test that drives ties and degenerate exits through the
    multiply itself.)"""
    inDim, outDim = 4096, 4096
    rng = np.random.default_rng({"gauss": 1, "heavy": 2, "zeros": 3, "few": 4, "tiny": 5, "ties": 6}[kind])
    W = make_w(outDim, inDim, seed=55)
    v = rng.standard_normal(inDim).astype(np.float32)
    if kind == "heavy":
        v = (v * np.exp(2.0 * rng.standard_normal(inDim))).astype(np.float32)
    elif kind == "zeros":
        v[rng.integers(0, inDim, 1500)] = 0
    elif kind == "few":
        v = rng.choice(np.array([0.5, 1.0, 2.0, 3.0], np.float32), inDim)
        W[np.arange(inDim), np.arange(inDim)] = rng.choice(np.array([0.01, 0.02], np.float16), inDim)       # the probes: the diagonal (convert.metal:14-31)
    elif kind == "tiny":
        v = (v * 1e-9).astype(np.float32)
    elif kind == "ties":
        v = np.sign(v).astype(np.float32)                       # |v| = 1 everywhere: a row's score is its mean alone -- whole rank planes tie
        W = (np.sign(W.astype(np.float32)) * np.float32(0.015625)).astype(np.float16)
    b, s, p, oob = oracle_cpu.convert_fp16(W)
    assert oob == 0
    ew = gpu_weights(ea, W, b, s, p)
    vd = devf(v)
    g = ea.gpu()
    outs = [torch.full((outDim,), float("nan"), device=DEV) for _ in range(3)]
    for effort in (0.02, 0.1, 0.25, 0.5, 0.9, 1.0):
        want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, effort)
        ea.bucketMul(vd, ew, None, outs[0], effort)                             # a lone call: the lean kernel
        g.eval()
        assert g.last_dispatch_count() == n and np.float32(g.last_cutoff()).tobytes() == np.float32(cutoff).tobytes(), (kind, effort, g.last_dispatch_count(), n)
        assert close(outs[0].cpu().numpy(), want), (kind, effort)
        ea.bucketMulGroup([(vd, ew, None, o, effort) for o in outs])           # three calls of one launch (another slicing)
        g.eval()
        for i, o in enumerate(outs):
            assert g.last_dispatch_count(i) == n and g.last_cutoff(i) == cutoff, (kind, effort, i)
            assert close(o.cpu().numpy(), want), (kind, effort, i)


def test_timing_hooks_of_the_shipped_library(ea, oracle_cpu):
    """The shipped kernels carry no stamp code (tests/test_abi.py::test_shipped_kernels_carry_no_lab_code): the device-clock modes are
    refused with a message naming the lab library, the HIP-event mode still times launches, and results are what they are without it."""
    if ea.lib().effort_is_lab_build():
        pytest.skip("EFFORT_HIP_LIB points at a lab build")
    outDim, inDim = 1024, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    g = ea.Gpu(0)
    v = make_v(inDim, seed=21)
    want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, 0.25)
    out = torch.zeros(outDim, device=DEV)
    try:
        for mode in (2, 3):
            with pytest.raises(ea.EffortError, match="libeffort_hip_lab"):
                g.enable_kernel_timing(mode)
        with pytest.raises(ea.EffortError, match="libeffort_hip_lab"):
            g.kernel_clock()
        with pytest.raises(ea.EffortError, match="libeffort_hip_lab"):
            g.debug_stamps()
        g.enable_kernel_timing(1)
        for _ in range(3):
            ea.bucketMul(devf(v), ew, None, out, 0.25, gpu=g)
        t = g.kernel_timing()
        assert t["samples"] == 3 and 1.0 < t["mul_us"] < 1000.0
        g.enable_kernel_timing(0)
        g.eval()
        assert close(out.cpu().numpy(), want) and g.last_dispatch_count() == n and g.last_cutoff() == cutoff
    finally:
        g.close()


def test_slice_counts_that_are_not_multiples_of_eight(ea, oracle_cpu, q4_11008):
    """Slices are dealt to the XCDs in rounds of 8, so a count like 5 or 12 pads the item grid, and the FIRST block of a call can be padding
    (it is for a call whose rotation puts slices >= the count there).  Round 6's Q4 rule launches 10..16 calls as one round of 5-8 tall slices
    (api.hip q4_one_round), so such counts are the default path now: every call's dispatch.size, BucketMul.cutoff (the hook is written by the
    item of tile 0 / slice 0, which used to be taken for the first block) and product against the oracle -- Q4 groups of 10, 13 and 16 on the
    default geometry, then forced slice counts, lone and grouped, FP16 too."""
    W, L, inDim, outDim = q4_11008
    ew = ea.ExpertWeights(dev16(L["buckets"]), devf(L["bucket.stats"]), dev16(L["probes"]), inSize=inDim, outSize=outDim,
                          outliers=devf(L["outliers"]), q4=True)
    g = ea.Gpu(0)
    efforts = (0.1, 0.25, 0.5)
    hv = [make_v(inDim, seed=700 + i, heavy=bool(i & 1)) for i in range(16)]
    wants = [oracle_cpu.bucket_mul_q4(hv[i], L["buckets"], L["bucket.stats"], L["probes"], L["outliers"], inDim, outDim, efforts[i % 3]) for i in range(16)]
    try:
        for tune, n in (((0, 0, 0), 16), ((0, 0, 0), 13), ((0, 0, 0), 10), ((8, 1, 5), 16), ((8, 2, 10), 7), ((8, 1, 21), 3), ((8, 1, 5), 1), ((8, 2, 12), 1)):
            g.set_tuning(*tune)
            outs = [torch.full((outDim,), float("nan"), device=DEV) for _ in range(n)]
            ea.bucketMulGroup([(devf(hv[i]), ew, None, outs[i], efforts[i % 3]) for i in range(n)], gpu=g)
            g.eval()
            for i in range(n):
                want, cnt, cutoff = wants[i]
                assert g.last_dispatch_count(i) == cnt and g.last_cutoff(i) == cutoff, (tune, n, i)
                assert close(outs[i].cpu().numpy(), want), (tune, n, i)
        # FP16: 12 and 20 slices (a 4096-row input cannot go below 8 with 8-wave workgroups), a group whose calls rotate through every XCD offset
        oD, iD = 1024, 4096
        Wf, b, s, p = converted(oracle_cpu, oD, iD)
        ewf = gpu_weights(ea, Wf, b, s, p)
        for tune in ((8, 1, 12), (8, 2, 20), (8, 4, 12)):
            g.set_tuning(*tune)
            outs = [torch.full((oD,), float("nan"), device=DEV) for _ in range(9)]
            ea.bucketMulGroup([(devf(hv[i]), ewf, None, outs[i], efforts[i % 3]) for i in range(9)], gpu=g)
            g.eval()
            for i in range(9):
                want, cnt, cutoff = oracle_cpu.bucket_mul(hv[i], b, s, p, iD, oD, efforts[i % 3])
                assert g.last_dispatch_count(i) == cnt and g.last_cutoff(i) == cutoff and close(outs[i].cpu().numpy(), want), (tune, i)
    finally:
        g.set_tuning(0, 0, 0)
        g.close()


def test_geometry_rules_of_round_six(ea, oracle_cpu, q4_11008):
    """The launch-geometry rules round 6 added, pinned through the slice-count hook (effort_debug_slice_counts): a lone 4096 -> 14336 call --
    the reference's timed shape -- takes 32 slices (24 = 171 rows in blocks of 256 slots before: pick_slices' power-of-two rule); a 16-call
    Q4 group on a context WITHOUT lanes launches as one round of 5 tall slices per call, on a context WITH lanes as 8 (api.hip q4_one_round);
    a pair of Q4 calls 16 slices; an FP16 launch without lanes whose last round of workgroups would be nearly empty cuts its last two calls into twice
    the slices.  Every product against the oracle."""
    W, L, inDim, outDim = q4_11008
    ew = ea.ExpertWeights(dev16(L["buckets"]), devf(L["bucket.stats"]), dev16(L["probes"]), inSize=inDim, outSize=outDim,
                          outliers=devf(L["outliers"]), q4=True)
    g = ea.Gpu(0)
    v = make_v(inDim, seed=41)
    want, cnt, cutoff = oracle_cpu.bucket_mul_q4(v, L["buckets"], L["bucket.stats"], L["probes"], L["outliers"], inDim, outDim, 0.25)
    try:
        # (third session: Q4 groups of 3 .. 9 calls without lanes take one E = 1 item per CU, counted on the padded item ranges: 3 calls 13 slices, 6 calls 6, 8 calls 5;
        #  9 calls do not fit such a round and stay at E = 2 x 8; with lanes 3 calls keep 8)
        for lanes, n, slices in ((1, 16, 5), (4, 16, 8), (1, 2, 16), (1, 12, 6), (1, 32, 8), (1, 3, 13), (1, 6, 6), (1, 8, 5), (1, 9, 8), (4, 3, 8),
                                  (1, 22, 16), (1, 24, 16), (1, 20, 8), (1, 26, 8), (4, 22, 8)):     # (Q4 thin last calls: 22 / 24 calls are 16 / 64 items over a round of two per CU)
            g.set_overlap(lanes)
            outs = [torch.full((outDim,), float("nan"), device=DEV) for _ in range(n)]
            ea.bucketMulGroup([(devf(v), ew, None, o, 0.25) for o in outs], gpu=g)
            g.eval()
            assert len(g.slice_counts(n - 1)) == slices, (lanes, n, len(g.slice_counts(n - 1)))
            for i in (0, n - 1):
                assert g.last_dispatch_count(i) == cnt and g.last_cutoff(i) == cutoff and close(outs[i].cpu().numpy(), want), (lanes, n, i)
        g.set_overlap(1)
        oD, iD = 14336, 4096
        Wf, b, s, p = converted(oracle_cpu, oD, iD, seed=77)
        ewf = gpu_weights(ea, Wf, b, s, p)
        out = torch.zeros(oD, device=DEV)
        ea.bucketMul(devf(v), ewf, None, out, 0.25, gpu=g)
        g.eval()
        assert len(g.slice_counts(0)) == 32
        wantf, cntf, cutf = oracle_cpu.bucket_mul(v, b, s, p, iD, oD, 0.25)
        assert g.last_dispatch_count() == cntf and g.last_cutoff() == cutf and close(out.cpu().numpy(), wantf)
        # FP16, third session: a launch that has the chip to itself and whose last round of workgroups would be nearly empty (at most 7/32 full: 11 / 12 / 13 / 23 calls of
        # 48 items are 16 / 64 / 112 / 80 items over a multiple of 512; 14 and 24 calls, 160 and 128, are not) cuts its LAST TWO calls into twice the slices (api.hip do_group: thinFrom); not with lanes
        oD, iD = 11008, 4096
        Wf, b, s, p = converted(oracle_cpu, oD, iD)
        ewf = gpu_weights(ea, Wf, b, s, p)
        hv = [make_v(iD, seed=900 + i, heavy=bool(i & 1)) for i in range(3)]
        wants = [oracle_cpu.bucket_mul(hv[i], b, s, p, iD, oD, e) for i, e in enumerate((0.25, 0.5, 0.1))]
        for lanes, n, first, last in ((1, 11, 8, 16), (1, 12, 8, 16), (1, 13, 8, 16), (1, 23, 8, 16), (1, 14, 8, 8), (1, 16, 8, 8), (1, 24, 8, 8), (4, 12, 8, 8), (1, 10, 8, 8)):
            g.set_overlap(lanes)
            outs = [torch.full((oD,), float("nan"), device=DEV) for _ in range(n)]
            ea.bucketMulGroup([(devf(hv[i % 3]), ewf, None, outs[i], (0.25, 0.5, 0.1)[i % 3]) for i in range(n)], gpu=g)
            g.eval()
            got = [len(g.slice_counts(i)) for i in range(n)]
            thin = 3 if n == 13 else 2                    # (13 calls are 112 items over a round: three thin calls -- 144 items -- cover them, two would not)
            assert got == [first] * (n - thin) + [last] * thin, (lanes, n, got)
            for i in (0, 1, n - 3, n - 2, n - 1):
                wantf, cntf, cutf = wants[i % 3]
                assert g.last_dispatch_count(i) == cntf and g.last_cutoff(i) == cutf and close(outs[i].cpu().numpy(), wantf), (lanes, n, i)
        # FP16, third session: a group of >= 8 calls on a SMALL matrix takes enough slices to give every CU an item (api.hip pick_slices: fill) --
        # 8 calls on 4096 x 4096: 16 slices (8 x 2 tiles x 16 = 256 items; 8 slices left half the CUs idle); and a group of >= 3 calls that fits one round of CUs
        # takes as many slices as keep it at one item per CU, counted as the items are dealt to the XCDs (a call's range padded to a multiple of 8): 9 calls 12
        # slices (9 x 24 = 216), 11 calls 8 (12 slices would be 264)
        oD, iD = 4096, 4096
        Ws, bs, ss, ps = converted(oracle_cpu, oD, iD)
        ews = gpu_weights(ea, Ws, bs, ss, ps)
        wants = [oracle_cpu.bucket_mul(hv[i], bs, ss, ps, iD, oD, e) for i, e in enumerate((0.25, 0.5, 0.1))]
        for n, slices in ((8, 16), (9, 12), (11, 8), (16, 8)):
            outs = [torch.full((oD,), float("nan"), device=DEV) for _ in range(n)]
            ea.bucketMulGroup([(devf(hv[i % 3]), ews, None, outs[i], (0.25, 0.5, 0.1)[i % 3]) for i in range(n)], gpu=g)
            g.eval()
            assert [len(g.slice_counts(i)) for i in range(n)] == [slices] * n, (n, len(g.slice_counts(0)))
            for i in (0, 1, 2, n - 1):
                wantf, cntf, cutf = wants[i % 3]
                assert g.last_dispatch_count(i) == cntf and g.last_cutoff(i) == cutf and close(outs[i].cpu().numpy(), wantf), (n, i)
        # ... 3 / 4 calls of 4096 x 11008: 13 / 10 slices (E = 2: 3 x 80 = 240, 4 x 64 = 256 padded items); 6 / 7 calls: E = 4 (their E = 2 items overflow one round
        # even at 8 slices) at 13 / 10 slices; 5 calls stay at E = 2 x 8 (240 items)
        oD, iD = 11008, 4096
        wants = [oracle_cpu.bucket_mul(hv[i], b, s, p, iD, oD, e) for i, e in enumerate((0.25, 0.5, 0.1))]
        for n, slices in ((3, 13), (4, 10), (5, 8), (6, 13), (7, 10), (-3, 8)):          # (-3: three calls on a context WITH lanes keep the 8 fat slices)
            g.set_overlap(4 if n < 0 else 1)
            n = abs(n)
            outs = [torch.full((oD,), float("nan"), device=DEV) for _ in range(n)]
            ea.bucketMulGroup([(devf(hv[i % 3]), ewf, None, outs[i], (0.25, 0.5, 0.1)[i % 3]) for i in range(n)], gpu=g)
            g.eval()
            assert [len(g.slice_counts(i)) for i in range(n)] == [slices] * n, (n, len(g.slice_counts(0)))
            for i in range(n):
                wantf, cntf, cutf = wants[i % 3]
                assert g.last_dispatch_count(i) == cntf and g.last_cutoff(i) == cutf and close(outs[i].cpu().numpy(), wantf), (n, i)
    finally:
        g.set_overlap(1)
        g.close()
