"""CPU model of the HIP cutoff's lookup-free, block-parallel bisection (effort_amd/csrc/cutoff_device.h, "THE BISECTION WITHOUT ITS
LOOKUPS") against the oracle's findCutoff32.  The device code replaces every use of a count inside the reference's loop --
the steering comparison and the two count-driven exit tests -- with comparisons of the threshold's bf16 CELL against five
order statistics of the values; this restates that control flow in numpy f32 arithmetic -- including the hand-over to the
closed-form tail and the ballot rounds for value ranges wider than the table -- asserts every shortcut against the count
it stands for, and compares the float BITS with the C oracle over seeded inputs that leave the loop through every exit.  The five
order statistics are also found the way the device finds them since round 5 -- from the HISTOGRAM of the cells, one wave, two
levels (`_device_order_statistic`), over the cells between the bounds and over the fixed window plain grids count into before the
value range is known -- and must be the same cells.  (The device code itself is compared with the oracle by the -m gpu tests: test_cutoff_*.)"""
import numpy as np
import pytest

F = np.float32
CAP = 4096                      # cells the device table holds (kCutoffBinsPerThread * 512 threads)
NO_LO, NO_HI = 0xFFFF0000, 0xFFF00000


def _bf16(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def _pat(f):
    return int(np.asarray(F(f)).view(np.uint32)) >> 16


def _cell_edge_tail(nb, lo, hi, X, loops):
    """bisect_to_cell_edge: the loop form (the closed form is checked against it on the GPU)."""
    while True:
        loops += 1
        if nb >= X:
            hi = nb
        else:
            lo = nb
        prev = nb
        nb = F((hi + lo) / F(2))
        if F(hi - lo) < F(0.00001) or loops > 100 or nb == prev:
            return nb


K_BLK = 16
WINDOW_BASE = (127 - 10) << 7          # kCutoffWindowBase: the cell of 2^-10


def _device_order_statistic(hist, base, above, k):
    """T(k) as wave 0 computes it from the histogram (cutoff_device.h, "one wave turns the histogram into the five order
    statistics"): lane L sums its CAP / 64 consecutive cells; a suffix scan over the lanes gives the count above each lane's last
    cell; whole lanes whose last cell still reaches k count in full, the boundary lane's cells are scanned the same way."""
    seg = CAP // 64
    lanes = hist.reshape(64, seg).sum(axis=1)
    total = int(lanes.sum())
    c1 = above + total - np.cumsum(lanes)                          # count(last cell of lane L) = everything in later lanes, + above
    allGE = above + total
    sgAll = int((c1 >= k).sum())                                   # (the counts fall with the lane: a prefix)
    assert all(c1[i] >= c1[i + 1] for i in range(63))
    sg = min(sgAll, 63)
    cells = hist[sg * seg:(sg + 1) * seg]
    cnt = int(c1[sg]) + int(cells.sum()) - np.cumsum(cells)        # count(cell) = later cells of the lane + what lies behind the lane
    n = sg * seg + int((cnt >= k).sum())
    if k <= 0 or sgAll >= 64:
        return 0xFFFFFFFF
    if k > 4096 or allGE < k:
        return 0
    return base + n


def _bits(f):
    return int(np.asarray(F(f)).view(np.uint32))


def _block_rounds(nb, lo, hi, pLo, pHi, catLo, catHi, loops, T):
    """cutoff_device.h, "THE ROUNDS, kBlk AT A TIME": a block runs the bare recurrence of the bounds for K_BLK rounds (every lane
    the same), round r leaving its midpoint in lane r; lane r then reconstructs round r -- the bounds after it are the latest
    midpoints at or before r that went each way -- and evaluates the reference's exits; the first lane that stops hands over.
    Returns (fin, nb, lo, hi, pLo, pHi, loops) where the loop stops (fin: an exit of the reference; else adjacent cells)."""
    tM2, tM1, tM, tP1, tP2 = T
    tMs = 0xFFFFFFFF if tM >= 0x10000 else tM << 16
    if pHi == pLo + 1:
        return (False, nb, lo, hi, pLo, pHi, loops)
    while True:
        a, l, h, rec = nb, lo, hi, []
        for r in range(K_BLK):                                   # the recurrence: runs on past the stop, harmlessly
            rec.append(a)
            below = _bits(a) >= tMs
            h, l = (a, l) if below else (h, a)
            a = F((h + l) / F(2))
        went = [_bits(x) >= tMs for x in rec]
        for r in range(K_BLK):                                   # "lane r"
            hs = [j for j in range(r + 1) if went[j]]
            ls = [j for j in range(r + 1) if not went[j]]
            hr, lr = (rec[hs[-1]] if hs else hi), (rec[ls[-1]] if ls else lo)
            pHr, pLr = (_bits(hr) >> 16 if hs else pHi), (_bits(lr) >> 16 if ls else pLo)
            cHr = int(pHr < tM1) + int(pHr < tM2) if hs else catHi
            cLr = int(pLr < tP1) + int(pLr < tP2) if ls else catLo
            p = _bits(rec[r]) >> 16
            nbr = F((hr + lr) / F(2))
            finr = (tP1 <= p < tM) or bool(F(hr - lr) < F(0.00001)) or cHr > cLr or loops + r + 1 > 100 or bool(nbr == rec[r])
            if finr or pHr == pLr + 1 or r == K_BLK - 1:
                state = (nbr, lr, hr, pLr, pHr, cLr, cHr)
                stopped = finr or pHr == pLr + 1
                break
        nb, lo, hi, pLo, pHi, catLo, catHi = state
        loops += r + 1
        if stopped:
            return (bool(finr), nb, lo, hi, pLo, pHi, loops)


def model_cutoff(v, probes_u16, q):
    pr = _bf16(probes_u16.view(np.float16).astype(np.float32))
    t = (F(100000.0) * v.astype(np.float32)).astype(np.float32)
    vals = _bf16(np.abs((t * pr).astype(np.float32)))
    vp = (vals.view(np.uint32) >> 16).astype(np.int64)
    effort = 4096 - q
    pmin, pmax = int(vp.min()), int(vp.max())
    nz = vp[vp > 0]
    pminNZ = min(int(nz.min()) if nz.size else 0xFFFF, pmax)
    frompat = lambda p: np.asarray(np.uint32(p << 16)).view(np.float32)[()]          # noqa: E731
    lo, hi = F(min(frompat(pmin), F(1000.0))), F(frompat(pmax))
    nb = F((lo + hi) / F(2))
    loops, minC, maxC, pLo, pHi, done = 0, 4096, 0, NO_LO, NO_HI, False
    CA = lambda p: int((vp > p).sum())                                               # noqa: E731

    def rnd(cnt):                                                                    # `round` of the device code
        nonlocal loops, lo, hi, nb, minC, maxC, pLo, pHi
        loops += 1
        p = _pat(nb)
        if cnt < effort:
            hi, maxC, pHi = nb, cnt, p
        else:
            lo, minC, pLo = nb, cnt, p
        prev = nb
        nb = F((hi + lo) / F(2))
        if cnt == effort or F(hi - lo) < F(0.00001) or abs(maxC - minC) < 3 or loops > 100:
            return True
        return nb == prev
    low = lambda: max(pLo, pminNZ) if pLo != NO_LO else pminNZ                       # noqa: E731
    top = lambda: pHi if pHi != NO_HI else pmax                                      # noqa: E731
    while not done and pHi != pLo + 1 and top() - low() + 1 > CAP:
        done = rnd(CA(_pat(nb)))
    if done:
        return nb
    if pHi == pLo + 1:
        return _cell_edge_tail(nb, lo, hi, frompat(pHi), loops)
    base, tp = low(), top()
    above = maxC if pHi != NO_HI else 0
    allGE = int((vp >= base).sum())

    def count_above(p):                                                              # the table lookup
        if p < base:
            return allGE
        if p > tp:
            return above
        return int((vp > p).sum())
    tbl = [count_above(base + c) if base + c <= tp else above for c in range(CAP)]
    assert all(tbl[c] >= tbl[c + 1] for c in range(CAP - 1))

    def first_cell_below(k):                                                         # T(k): count(p) >= k  <=>  p < T(k)
        if k <= 0:
            return 0xFFFFFFFF
        if k > 4096 or allGE < k:
            return 0
        n = sum(1 for c in tbl if c >= k)
        return 0xFFFFFFFF if n == CAP else base + n
    m = effort
    tM2, tM1, tM, tP1, tP2 = (first_cell_below(m + d) for d in (-2, -1, 0, 1, 2))
    # the device finds the same five cells from the HISTOGRAM, one wave, two levels (cutoff_device.h, round 5) -- from the cells
    # [base, top], or, on plain grids, from a fixed window counted before the range was known
    hist = np.bincount((vp[(vp >= base) & (vp <= tp)] - base).astype(np.int64), minlength=CAP)[:CAP]
    assert [_device_order_statistic(hist, base, above, m + d) for d in (-2, -1, 0, 1, 2)] == [tM2, tM1, tM, tP1, tP2]
    if pLo == NO_LO and pHi == NO_HI and pminNZ >= WINDOW_BASE and pmax - WINDOW_BASE < CAP:
        win = np.bincount((vp[vp >= WINDOW_BASE] - WINDOW_BASE).astype(np.int64), minlength=CAP)[:CAP]
        assert [_device_order_statistic(win, WINDOW_BASE, 0, m + d) for d in (-2, -1, 0, 1, 2)] == [tM2, tM1, tM, tP1, tP2]
    # count(hi) as "how many of m-1, m-2 it reaches", count(lo) as "how many of m+1, m+2": |maxCount - minCount| < 3 <=> catHi > catLo
    catHi = int(pHi < tM1) + int(pHi < tM2) if pHi != NO_HI else int(maxC >= m - 1) + int(maxC >= m - 2)
    catLo = int(pLo < tP1) + int(pLo < tP2) if pLo != NO_LO else int(minC >= m + 1) + int(minC >= m + 2)
    blk = _block_rounds(nb, lo, hi, pLo, pHi, catLo, catHi, loops, (tM2, tM1, tM, tP1, tP2))       # the device's block form, from the same state
    fin = False
    while not fin and pHi != pLo + 1:
        p = _pat(nb)
        cnt = count_above(p)                                                         # (the model checks every shortcut against the count)
        below = p >= tM
        assert below == (cnt < m)
        loops += 1
        if below:
            hi, pHi, catHi = nb, p, int(p < tM1) + int(p < tM2)
        else:
            lo, pLo, catLo = nb, p, int(p < tP1) + int(p < tP2)
        prev = nb
        nb = F((hi + lo) / F(2))
        cntEq = tP1 <= p < tM
        dLt3 = catHi > catLo
        mx = count_above(pHi) if pHi != NO_HI else maxC
        mn = count_above(pLo) if pLo != NO_LO else minC
        assert cntEq == (cnt == m) and dLt3 == (abs(mx - mn) < 3), (cntEq, cnt, m, dLt3, mx, mn)
        fin = cntEq or bool(F(hi - lo) < F(0.00001)) or dLt3 or loops > 100 or bool(nb == prev)
    # the block form stops in the same round with the same state, bit for bit
    same = lambda x, y: np.asarray(F(x)).view(np.uint32) == np.asarray(F(y)).view(np.uint32)        # noqa: E731
    assert blk[0] == fin and same(blk[1], nb) and same(blk[2], lo) and same(blk[3], hi) and blk[4:] == (pLo, pHi, loops), (blk, fin, nb, lo, hi, pLo, pHi, loops)
    if fin:
        return nb
    return _cell_edge_tail(nb, lo, hi, frompat(pHi), loops)


def _inputs(seed, kind):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(4096).astype(np.float32)
    pr = (rng.standard_normal(4096) * 0.02).astype(np.float16)
    if kind == "heavy":
        v = (v * np.exp(2.0 * rng.standard_normal(4096))).astype(np.float32)
    elif kind == "zeros":
        v[rng.integers(0, 4096, 1500)] = 0
        pr[rng.integers(0, 4096, 300)] = 0
    elif kind == "wide":                                   # > 64 octaves of range: ballot rounds before the table
        v = (v * np.exp2(rng.integers(-60, 40, 4096).astype(np.float32))).astype(np.float32)
    elif kind == "few":                                    # few distinct values: the count == effort / counts exits
        v = rng.choice(np.array([0.5, 1.0, 2.0, 3.0], np.float32), 4096)
        pr = rng.choice(np.array([0.01, 0.02], np.float16), 4096)
    elif kind == "tiny":
        v = (v * 1e-9).astype(np.float32)
    return v, pr.view(np.uint16)


@pytest.mark.parametrize("kind", ["gauss", "heavy", "zeros", "wide", "few", "tiny"])
def test_lookup_free_bisection_model_matches_oracle(oracle_cpu, kind):
    mism = 0
    for seed in range(40):
        v, pr = _inputs(seed, kind)
        for effort in (0.0, 0.02, 0.1, 0.25, 0.5, 0.9, 1.0):
            want, _ = oracle_cpu.find_cutoff(v, pr, 0, effort)
            got = model_cutoff(v, pr, oracle_cpu.effort_to_q(effort))
            if np.float32(want).view(np.uint32) != np.float32(got).view(np.uint32):
                mism += 1
    assert mism == 0
