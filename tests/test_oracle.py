"""CPU tests: pin the oracle (known-answer material of the reference + golden fixtures + invariants).

FP16 parity vs Swift/Metal is UNPINNED (nothing of the reference's FP16 path runs here and its tests hold
no vectors for it); what is checked: the docs' worked layout example, the Q4 fixtures produced by the
reference's own q4_draft.py, the reference's own acceptance threshold (cos-sim >= 0.99 vs dense,
playground.swift:37-41) and structural invariants of the layout.
"""
import glob
import os

import numpy as np
import pytest

from tests.util import cos, make_v, make_w

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


# ---------------------------------------------------------------- number formats
def test_half_conversions_exhaustive(oracle_cpu):
    lib = oracle_cpu.lib()
    allh = np.arange(65536, dtype=np.uint32).astype(np.uint16)
    ref = allh.view(np.float16).astype(np.float32)
    got = np.array([lib.eo_h2f(int(h)) for h in allh], np.float32)
    ok = (got == ref) | (np.isnan(got) & np.isnan(ref))
    assert ok.all()
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-8, 1e-4, 1.0, 100.0, 7e4)])
    xs = np.concatenate([xs, ref[~np.isnan(ref)], np.float32([0.0, -0.0, 65504, 65520, 65519.99, 2.98e-8, 2.9802322e-8, 8.9e-8])])
    want = xs.astype(np.float16).view(np.uint16)
    got = np.array([lib.eo_f2h(float(x)) for x in xs], np.uint16)
    assert (got == want).all()


def test_bf16_round(oracle_cpu):
    import torch
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.standard_normal(5000).astype(np.float32) * s for s in (1e-3, 1.0, 2000.0)] + [np.float32([999.0, 1000.0, 0.0])])
    want = torch.from_numpy(xs).to(torch.bfloat16).to(torch.float32).numpy()
    got = np.array([oracle_cpu.bf16r(float(x)) for x in xs], np.float32)
    assert (got == want).all()
    assert oracle_cpu.bf16r(999.0) == 1000.0      # the clamp value of findCutoff32's min reduction


# ---------------------------------------------------------------- docs known-answer tests
def test_docs_layout_example(oracle_cpu):
    """docs/bucketmul.html:42-231: first row of the 12x12 example, buckets of 4."""
    row = np.float16([.46, .87, -.19, .27, .18, -.39, -.29, -.62, -.81, -.34, -.84, .33])
    ranked, oob = oracle_cpu.bucketize_row(row, 4)
    assert oob == 0
    want_vals = [[.87, -.62, -.84], [.46, -.39, -.81], [.27, -.29, -.34], [-.19, .18, .33]]
    want_pos = [[1, 3, 2], [0, 1, 0], [3, 2, 1], [2, 0, 3]]
    bits = ranked.view(np.uint16)
    assert (bits & 3).tolist() == want_pos                     # position in the low bits (bucket size 4)
    vals = (bits & 0xFFFC).view(np.float16).astype(np.float32)
    assert np.allclose(vals, want_vals, rtol=4e-3)             # low 2 mantissa bits were overwritten
    means = np.abs(np.float32(want_vals)).mean(axis=1)
    assert np.allclose(means, [0.777, 0.553, 0.300, 0.233], atol=1e-3)
    assert np.allclose(np.abs(ranked.astype(np.float32)).mean(axis=1), means, rtol=4e-3)


def test_docs_cutoff_example_cosine(oracle_cpu):
    """docs/equations.html:312-358: keeping only products >= 100 gives cos-sim 0.999."""
    W = np.float32([[1, 13, 2], [0.1, 1, 8], [1, 3, 256]])
    v = np.float32([1000, 10, 1])
    full = W @ v
    prods = W * v
    approx = np.where(prods >= 100, prods, 0).sum(axis=1).astype(np.float32)
    assert full.tolist() == [1132, 118, 1286] and approx.tolist() == [1130, 100, 1256]
    assert abs(oracle_cpu.cosine(approx, full) - 0.99989) < 2e-5


def test_docs_layout_example_second_row(oracle_cpu):
    """docs/bucketmul.html: the second input row (v_1) of the 12x12 example -- its four rank rows and their means as printed
    (0.747 / 0.503 / 0.28 / 0.16).  The page prints -0.42 with position 2; in its own matrix -0.42 is element 1 of the
    third bucket [.5, -.42, -.23, .02], which is what the layout (and this test) holds."""
    row = np.float16([-.87, .11, .03, .5, .43, .87, -.49, .59, .5, -.42, -.23, .02])
    ranked, oob = oracle_cpu.bucketize_row(row, 4)
    assert oob == 0
    want_vals = [[-.87, .87, .5], [.5, .59, -.42], [.11, -.49, -.23], [.03, .43, .02]]
    want_pos = [[0, 1, 0], [3, 3, 1], [1, 2, 2], [2, 0, 3]]
    bits = ranked.view(np.uint16)
    assert (bits & 3).tolist() == want_pos
    vals = (bits & 0xFFFC).view(np.float16).astype(np.float32)
    assert np.allclose(vals, want_vals, rtol=4e-3, atol=2e-3)    # the low 2 mantissa bits carry the position
    means = np.abs(np.float32(want_vals)).mean(axis=1)
    assert np.allclose(means, [0.747, 0.503, 0.28, 0.16], atol=4e-3)
    assert np.allclose(np.abs(ranked.astype(np.float32)).mean(axis=1), means, rtol=2e-2, atol=2e-3)
    # the first row's last rank row: the page prints 0.233 once and 0.223 once; (.19 + .18 + .33) / 3 = 0.2333
    assert abs(float(np.abs(np.float32([-.19, .18, .33])).mean()) - 0.2333) < 1e-4


def test_docs_cutoff_example_through_prepare_dispatch(oracle_cpu):
    """docs/equations.html:312-358 driven through the oracle's prepareDispatch (bucketMul.metal:47-79): inputs (1, 10, 1000),
    each input row's weights sorted descending -- 256 8 2 / 13 3 1 / 1 1 0.1 with output positions 3 2 1 / 1 3 2 / 1 3 2 --
    and the page's cutoff of 100.  A rank row here is one weight, so its mean IS the weight and the keep test
    cutoff < (1e5 * mean) * |v| is the page's  el_v * el_w  against the cutoff.  The page keeps the product that equals
    100 exactly ("the 0.1 weight made the cut") while its pseudocode says ">": any cutoff in (30, 100) * 1e5 gives the
    page's kept set; 99e5 is used.  Result (1130, 100, 1256), cos-sim 0.99989 against (1132, 118, 1286)."""
    inDim, ranks = 3, 3
    v = np.float32([1, 10, 1000])
    w = np.float32([[256, 8, 2], [13, 3, 1], [1, 1, 0.1]])                 # [input j][rank]
    pos = np.array([[3, 2, 1], [1, 3, 2], [1, 3, 2]]) - 1                   # output index of that weight
    stats = np.zeros((ranks * inDim, 4), np.float16)
    for r in range(ranks):
        for j in range(inDim):
            stats[r * inDim + j, :] = w[j, r]                               # rank-major rows: row = rank * inDim + j (convert.metal:83-100)
    disp, n = oracle_cpu.prepare_dispatch(v, stats, 0, 99e5, inDim, 1, percentLoad=ranks)
    assert n == 5
    rows = disp[:n, 1].astype(int).tolist()
    assert rows == [0, 1, 2, 5, 8]                                          # 256*1, 13*10, 1*1000 | 1*1000 | 0.1*1000
    assert disp[:n, 0].tolist() == [1, 10, 1000, 1000, 1000]
    out = np.zeros(3, np.float32)
    for val, row in disp[:n]:
        r, j = divmod(int(row), inDim)
        out[pos[j, r]] += np.float32(val) * np.float32(np.float16(w[j, r]))
    assert np.allclose(out, [1130, 100, 1256], rtol=3e-4)                   # f16(0.1) * 1000 = 99.98
    assert abs(oracle_cpu.cosine(out, np.float32([1132, 118, 1286])) - 0.99989) < 2e-5
    # nine products in all: five kept = "55 % of the calculations"
    assert abs(n / 9 - 0.55) < 0.01


# ---------------------------------------------------------------- Q4 layout / multiply pinned by the reference's q4_draft.py
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "q4_*.npz"))), ids=os.path.basename)
def test_q4_layout_matches_reference_fixture(path):
    from oracle import q4_layout
    g = np.load(path)
    L = q4_layout.convert(g["core2"])
    assert np.array_equal(L["buckets"].view(np.uint16), g["buckets_u16"])
    assert np.array_equal(L["bucket.stats"], g["bucket_stats"])
    assert np.array_equal(L["probes"].view(np.uint16), g["probes"].view(np.uint16))
    a, b = L["outliers"], g["outliers"]                       # same set; the reference's order is unstable-sort order
    assert np.array_equal(a[np.lexsort((a[:, 2], a[:, 1]))], b[np.lexsort((b[:, 2], b[:, 1]))])
    assert np.array_equal(np.abs(a[:, 0]), np.abs(b[:, 0]))   # and the same |value| sequence
    ov2 = q4_layout.draft_mul_no_effort(L, g["v"], g["core2"].shape[1])
    assert np.array_equal(ov2, g["output_vector2"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "q4_*.npz"))), ids=os.path.basename)
def test_q4_oracle_multiply_matches_reference_draft(path, oracle_cpu):
    """The C restatement of prepareDispatchQ4 + bucketMulQ4 with every row dispatched must equal the draft's
    output_vector2 (captured from the reference) plus the one documented difference: the Metal kernel treats
    a zero nibble as positive (bucketMulQ4.metal:77) whereas the draft uses sign(0) == 0 (q4_draft.py:224)."""
    from oracle import q4_layout
    g = np.load(path)
    core2, v = g["core2"], g["v"]
    inDim, outDim = core2.shape
    disp, n = oracle_cpu.prepare_dispatch_q4(v, g["bucket_stats"], 0, -1.0, inDim, outDim // 32)
    assert n == inDim * 8                                       # cutoff -1: every row passes
    D = oracle_cpu.round_up_pad(disp, n)
    out = oracle_cpu.bucket_mul_q4_dispatch(g["buckets_u16"], disp, D, outDim // 32, outDim)
    L = q4_layout.convert(core2)
    corr = np.zeros(outDim)
    base = np.arange(outDim // 8) * 8
    for i in range(inDim * 8):
        z = L["_vals_rows"][i] == 0
        corr[(base + L["_pos_rows"][i])[z]] += np.float64(np.float32(v[i // 8]) * np.float32(L["_avg"][i]))
    want = g["output_vector2"] + corr
    assert np.allclose(out, want, rtol=2e-5, atol=2e-5 * np.abs(want).max())


def test_q4_outliers_oracle(oracle_cpu):
    g = np.load(os.path.join(GOLDEN, "q4_64x256.npz"))
    v = g["v"]
    out = np.zeros(256, np.float32)
    oracle_cpu.lib().eo_calc_outliers(oracle_cpu._p(np.ascontiguousarray(v)), oracle_cpu._p(np.ascontiguousarray(g["outliers"])),
                                      __import__("ctypes").c_uint64(len(g["outliers"])), oracle_cpu._p(out))
    want = np.zeros(256)
    for val, i, o, _ in g["outliers"]:
        want[int(o)] += float(v[int(i)]) * float(val)
    assert np.allclose(out, want, rtol=1e-5, atol=1e-6)


# ---------------------------------------------------------------- FP16 converter invariants
@pytest.fixture(scope="module")
def conv_small(oracle_cpu):
    W = make_w(256, 4096, seed=5)
    buckets, stats, probes, oob = oracle_cpu.convert_fp16(W)
    return W, buckets, stats, probes, oob


def _check_layout(W, buckets, stats, probes, ignore_zero=False):
    outDim, inDim = W.shape
    C = outDim // 16
    b = buckets.view(np.uint16).reshape(16, inDim, C)            # [rank][inRow][bucket]
    pos = (b & 15).astype(np.int64)
    col = np.arange(C)[None, None, :] * 16 + pos
    orig = W.T.view(np.uint16)[np.arange(inDim)[None, :, None], col]       # W[col][inRow]
    same = ((orig & 0xFFF0) | pos.astype(np.uint16)) == b
    if ignore_zero:
        # a slot holding (+-)0 is either a real zero weight or the slot a displaced zero never reached
        # (sortAbs's zero padding ties with real zeros, model.swift:664-676): numerically identical
        same |= (b & 0x7FF0) == 0
        assert ((b & 0x7FF0) == 0).sum() <= ((W.view(np.uint16) & 0x7FF0) == 0).sum()
    assert same.all()
    if not ignore_zero:
        assert (np.sort(pos, axis=0) == np.arange(16)[:, None, None]).all()   # 16 ranks = a permutation of positions
    mag = (b & 0x7FF0).astype(np.int64)
    assert (mag[:-1] >= mag[1:]).all()                            # sorted by |w| descending along rank
    mean = np.abs(buckets.astype(np.float64)).mean(axis=1)
    st = stats.astype(np.float64)
    assert (st[:, 0:1] == st).all()                               # all four lanes equal (convert.metal:115-118)
    assert np.allclose(st[:, 3], mean, rtol=1.5e-3)               # f16 rounding of the f32 mean
    rep = 1 if outDim >= 4096 else 4096 // outDim
    want = np.array([W[i // rep, i // rep + i % rep] for i in range(4096)], np.float16)
    assert np.array_equal(probes.view(np.uint16), want.view(np.uint16))


def test_convert_fp16_invariants(conv_small):
    W, buckets, stats, probes, oob = conv_small
    assert oob == 0 and buckets.shape == (4096 * 16, 16) and stats.shape == (4096 * 16, 4)
    _check_layout(W, buckets, stats, probes)


def test_convert_fp16_is_sorted_like_a_stable_sort_without_ties(oracle_cpu):
    """Where |w| has no ties inside a row the bitonic network must agree with any correct sort."""
    rng = np.random.default_rng(3)
    row = rng.permutation(np.arange(1, 65, dtype=np.float32) / 64).astype(np.float16) * rng.choice([-1, 1], 64).astype(np.float16)
    ranked, oob = oracle_cpu.bucketize_row(row, 16)
    bits = ranked.view(np.uint16)
    for bkt in range(4):
        members = row[bkt * 16:(bkt + 1) * 16]
        order = np.argsort(-np.abs(members.astype(np.float32)), kind="stable")
        assert ((bits[:, bkt] & 15) == order).all()


def test_convert_fp16_preconditions(oracle_cpu):
    with pytest.raises(ValueError):
        oracle_cpu.convert_fp16(make_w(96, 4096))        # 4096 % 96 != 0 (convert.swift:210)
    with pytest.raises(ValueError):
        oracle_cpu.convert_fp16(make_w(256, 2048))       # inDim < 4096 (convert.swift:212)


def test_convert_fp16_nonpow2_with_exact_zeros(oracle_cpu):
    """outDim = 4160 is padded to 8192 by sortAbs; planted +-0 weights tie with the padding.  The restatement
    follows the reference literally (bucket 0 may receive padding entries); real non-zero weights are intact."""
    W = make_w(4160, 4096, seed=9, zeros=40)
    buckets, stats, probes, oob = oracle_cpu.convert_fp16(W)
    assert oob == 0
    _check_layout(W, buckets, stats, probes, ignore_zero=True)


# ---------------------------------------------------------------- cutoff / dispatch / multiply
def test_effort_to_q(oracle_cpu):
    assert [oracle_cpu.effort_to_q(e) for e in (0.25, 0.5, 1.0, 0.0)] == [3071, 2047, 0, 4095]


def test_find_cutoff_selects_requested_probe_rank(conv_small, oracle_cpu):
    W, buckets, stats, probes, _ = conv_small
    v = make_v(4096)
    vals = np.array([oracle_cpu.bf16r(abs(np.float32(np.float32(100000.0) * v[j]) * np.float32(oracle_cpu.bf16r(float(probes[j])))))
                     for j in range(4096)], np.float32)
    prev = np.inf
    for effort in (0.1, 0.25, 0.5, 0.75, 1.0):
        cutoff, loops = oracle_cpu.find_cutoff(v, probes, 0, effort)
        above = int((vals > cutoff).sum())
        want = 4096 - oracle_cpu.effort_to_q(effort)
        assert loops <= 101 and abs(above - want) <= 6, (effort, above, want)
        assert cutoff <= prev
        prev = cutoff


def test_bucketmul_oracle_against_dense(conv_small, oracle_cpu):
    """The reference's own acceptance test: cos-sim vs the dense product (playground.swift:37-41 uses > 0.99)."""
    W, buckets, stats, probes, _ = conv_small
    for heavy in (False, True):
        v = make_v(4096, heavy=heavy)
        dense = oracle_cpu.dense_gemv(W, v)
        sims, counts = [], []
        for effort in (1.0, 0.5, 0.25, 0.1):
            out, n, cutoff = oracle_cpu.bucket_mul(v, buckets, stats, probes, 4096, 256, effort)
            sims.append(cos(out, dense))
            counts.append(n)
        # i.i.d. Gaussian weights are the worst case for the method (real LLM weights/activations are
        # heavier-tailed: docs/ryc/ryc0.3.png shows 0.99 at 22 %); these bands are for this synthetic input
        assert sims[0] > 0.9999 and sims[1] > 0.98 and sims[2] > 0.9 and sims[3] > 0.75, sims
        assert sims == sorted(sims, reverse=True)
        assert counts == sorted(counts, reverse=True) and counts[0] <= 4096 * 16
        assert abs(counts[2] / (4096 * 16) - 0.25) < 0.1


def test_bucketmul_oracle_dispatch_consistency(conv_small, oracle_cpu):
    W, buckets, stats, probes, _ = conv_small
    v = make_v(4096, seed=7)
    cutoff, _ = oracle_cpu.find_cutoff(v, probes, 0, 0.3)
    disp, n = oracle_cpu.prepare_dispatch(v, stats, 0, cutoff, 4096, 16)
    rows = (disp[:n, 1] / 16).astype(np.int64)
    assert (np.diff(rows) > 0).all()                             # ascending bucket rows
    mean = stats.astype(np.float32)[:, 3]
    keep = cutoff < (np.float32(100000.0) * mean) * np.abs(v[np.arange(4096 * 16) % 4096])
    assert np.array_equal(np.nonzero(keep)[0], rows)
    assert np.array_equal(disp[:n, 0], v[rows % 4096])
    D = oracle_cpu.round_up_pad(disp, n)
    assert D % 2048 == 0 and D > n and not disp[n:D].any()
    out = oracle_cpu.bucket_mul_dispatch(buckets, disp, D, 16, 256)
    full, n2, c2 = oracle_cpu.bucket_mul(v, buckets, stats, probes, 4096, 256, 0.3)
    assert n2 == n and c2 == cutoff and np.array_equal(out, full)
    # independent f64 evaluation of the same selection
    want = np.zeros(256)
    bu = buckets.view(np.uint16)
    for r in rows:
        np.add.at(want, np.arange(16) * 16 + (bu[r] & 15), float(v[r % 4096]) * buckets[r].astype(np.float64))
    assert np.allclose(out, want, rtol=1e-5, atol=1e-5 * np.abs(want).max())


def test_bucketmul_oracle_scale_invariance(conv_small, oracle_cpu):
    """Scaling v by a power of two scales every product and the cutoff exactly: same rows, out scales exactly."""
    W, buckets, stats, probes, _ = conv_small
    v = make_v(4096, seed=11)
    a, na, _ = oracle_cpu.bucket_mul(v, buckets, stats, probes, 4096, 256, 0.25)
    b, nb, _ = oracle_cpu.bucket_mul(v * 4, buckets, stats, probes, 4096, 256, 0.25)
    assert na == nb and np.array_equal(a * 4, b)
