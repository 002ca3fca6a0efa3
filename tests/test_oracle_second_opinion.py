"""The C oracle (oracle/effort_oracle.c) against an independent numpy restatement of the same Metal / Swift text
(oracle/metal_numpy.py): cutoff bits, loop counts, dispatch lists and products over seeded inputs, including inputs built
to leave findCutoff32 through each of its exits.  Not a pin of the reference -- a guard against transcription errors."""
import numpy as np
import pytest

from oracle import metal_numpy as mn
from tests.util import make_v, make_w

IN, OUT = 4096, 256


@pytest.fixture(scope="module")
def layout(oracle_cpu):
    W = make_w(OUT, IN, seed=4242)
    return oracle_cpu.convert_fp16(W)[:3]


def _inputs(rng, kind):
    v = rng.standard_normal(IN).astype(np.float32)
    if kind == 1:
        v *= np.exp(rng.standard_normal(IN)).astype(np.float32)              # heavy-tailed
    elif kind == 2:
        v = np.round(v * 2).astype(np.float32) / 2                           # few distinct magnitudes: counts jump over the target
    elif kind == 3:
        v[rng.random(IN) < 0.6] = 0.0                                        # many exact zeros
    elif kind == 4:
        v = (v * np.float32(1e-4)).astype(np.float32)                        # tiny products: the bounds meet (< 1e-5) before the counts do
    return v


def test_cutoff_and_loops_match_on_200_inputs(oracle_cpu, layout):
    b, s, p = layout
    rng = np.random.default_rng(20260927)
    exits = {}
    for k in range(200):
        v = _inputs(rng, k % 5)
        probes = p if k % 7 else np.full(4096, np.float16(0.0123))           # constant probes: products proportional to |v|
        effort = float(rng.choice([0.0, 0.02, 0.1, 0.25, 0.3, 0.5, 0.75, 0.97, 1.0]))
        q = oracle_cpu.effort_to_q(effort)
        assert q == int(float(4095) * (1 - effort))
        want, loops, why = mn.find_cutoff32(v, probes, 0, q)
        got, got_loops = oracle_cpu.find_cutoff(v, probes, 0, effort)
        assert np.float32(got).tobytes() == np.float32(want).tobytes() and got_loops == loops, (k, effort, why)
        exits[why] = exits.get(why, 0) + 1
    assert exits.get("count", 0) and exits.get("counts", 0) and exits.get("bounds", 0), exits     # every ordinary exit was taken


def test_hundred_loop_exit(oracle_cpu):
    """`if (loops>100)`: two clusters of equal values around the target rank keep |maxCount - minCount| >= 3 and the count off
    the target while the bracket shrinks float by float; large magnitudes keep maxBound - minBound above 1e-5."""
    v = np.ones(IN, np.float32)
    v[: IN // 2] = np.float32(3.0)
    probes = np.full(4096, np.float16(1.0))
    effort = 0.25                                                             # target rank 1025: between the clusters' counts 0 / 2048 / 4096
    want, loops, why = mn.find_cutoff32(v, probes, 0, int(4095 * (1 - effort)))
    got, got_loops = oracle_cpu.find_cutoff(v, probes, 0, effort)
    assert np.float32(got).tobytes() == np.float32(want).tobytes() and got_loops == loops
    assert why == "loops" and loops == 101


def test_dispatch_and_product_match(oracle_cpu, layout):
    b, s, p = layout
    rng = np.random.default_rng(7)
    for k in range(12):
        v = _inputs(rng, k % 4)
        effort = [0.1, 0.25, 0.5, 1.0][k % 4]
        out_np, n_np, cut_np, loops, why, disp_np = mn.full_mul(v, b, s, p, IN, OUT, effort)
        out_c, n_c, cut_c = oracle_cpu.bucket_mul(v, b, s, p, IN, OUT, effort)
        assert n_c == n_np and np.float32(cut_c).tobytes() == np.float32(cut_np).tobytes(), (k, why)
        disp_c, n2 = oracle_cpu.prepare_dispatch(v, s, 0, cut_c, IN, OUT // 16)
        assert n2 == n_np and np.array_equal(disp_c[:n2].view(np.uint32), disp_np.view(np.uint32)), k      # the list, bit for bit (ascending rows)
        # products: same per-thread sums; only bucketIntegrate's simd_sum order differs between the two restatements
        scale = float(np.abs(out_c).max()) + 1e-30
        assert float(np.abs(out_np - out_c).max()) <= 2e-6 * scale, k
        tmp = mn.bucket_mul(b, mn.round_up_and_zero(disp_np), OUT // 16, 32)
        # ... and with the C oracle's butterfly order the integrate is bit-identical too
        sref = tmp[:, :OUT].copy()
        d = 16
        while d >= 1:
            sref[:d] = sref[:d] + sref[d:2 * d]
            d //= 2
        assert np.array_equal(sref[0], out_c), k


def test_bisection_tail_reaches_the_cell_edge():
    """The claim the HIP kernel's closed-form tail rests on (cutoff_device.h, bisect_to_cell_edge): once findCutoff32's bounds sit in
    adjacent bfloat cells, its remaining rounds -- the count test decided by `newBound >= X`, X the first float of the upper cell --
    end on X itself whenever X >= 128 and fewer than 60 rounds are spent, in at most 19 more rounds.  The loop below is the
    reference's (bucketMul.metal:199-246 with the count replaced by its known outcome), in float32."""
    f = np.float32
    rng = np.random.default_rng(7)

    def tail(nb, mn, mx, X, loops):
        while True:
            loops += 1
            if nb >= X:
                mx = nb
            else:
                mn = nb
            prev = nb
            nb = f(f(mx + mn) / f(2))
            if f(mx - mn) < f(0.00001) or loops > 100 or nb == prev:
                return nb, loops

    checked = longest = 0
    for _ in range(20000):
        X = f(int(rng.integers(128, 256)) * 2.0 ** (int(rng.integers(0, 40)) - 7))          # a bfloat pattern >= 1
        pat = int(X.view(np.uint32)) >> 16
        lo_cell = np.uint32((pat - 1) << 16).view(np.float32)
        hi_end = np.uint32((pat + 1) << 16).view(np.float32)
        mn = f(lo_cell + (X - lo_cell) * f(rng.random()))
        mn = mn if mn < X else lo_cell
        mx = f(X + (hi_end - X) * f(rng.random()))
        mx = mx if mx < hi_end else X
        l0 = int(rng.integers(1, 60))
        got, l1 = tail(f(f(mx + mn) / f(2)), mn, mx, X, l0)
        if X >= 128:
            checked += 1
            longest = max(longest, l1 - l0)
            assert got == X, (X, mn, mx, got)
    assert checked > 10000 and longest <= 19, (checked, longest)
