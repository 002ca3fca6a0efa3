"""A SECOND, independent restatement of the FP16 hot path -- findCutoff32 -> prepareDispatch -> roundUp/zeroRange32 ->
bucketMul -> bucketIntegrate -- written in numpy from the Metal / Swift text (bucketMul.metal:11-247,
bucketMul.swift:34-88), not from oracle/effort_oracle.c.  TEST INFRASTRUCTURE: tests/test_oracle_second_opinion.py
cross-checks the C oracle against it.  It is not a pin (nothing here ran on the reference's hardware); it catches
transcription errors, since the two restatements share no code and are organised differently: this one keeps the
kernels' thread structure (1024 threads x 4 values, simdgroups of 32, threadgroup variables), the C file is scalar.

Everything is float32 arithmetic in numpy scalars / arrays; bfloat = float32 rounded to nearest even on 16 bits.
"""
from __future__ import annotations

import numpy as np

F = np.float32
CUTOFF_SCALE = 100000          # `#define CUTOFF_SCALE 100000` (an int: int * float -> float), bucketMul.metal:33


def bfloat(x):
    """float -> bfloat -> float (round to nearest, ties to even), elementwise."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    r = np.where(nan, u | 0x00400000, r) & 0xFFFF0000
    return r.astype(np.uint32).view(np.float32)


def find_cutoff32(v: np.ndarray, probes_half: np.ndarray, expNo: int, _effort: int):
    """kernel findCutoff32 (bucketMul.metal:141-247), 1024 threads in one threadgroup (bucketMul.swift:41).  Returns
    (out[0], loops, which exit fired)."""
    v = np.asarray(v, dtype=np.float32)
    pr = np.asarray(probes_half).view(np.float16).reshape(-1)
    effort = 4096 - int(_effort)                                             # uint effort = 4096-_effort
    ids = np.arange(1024)
    # per thread: bfloat4 myVal; myVal[i] = bfloat(abs(CUTOFF_SCALE * v[4*id+i] * bfloat(probes[4*id+i+expNo*4096])))
    myVal = np.empty((1024, 4), np.float32)
    for i in range(4):
        pv = bfloat(pr[4 * ids + i + expNo * 4096].astype(np.float32))        # bfloat(half)
        t = (F(CUTOFF_SCALE) * v[4 * ids + i]).astype(np.float32)             # left to right: (100000 * v) ...
        myVal[:, i] = bfloat(np.abs((t * pv).astype(np.float32)))              # ... * bfloat(probe); abs; bfloat
    myMax = np.maximum(F(-999), myVal.max(axis=1))                            # float myMax = -999, max over the four
    myMin = np.minimum(F(999), myVal.min(axis=1))
    # simd_min / simd_max over simdgroups of 32 threads; lane 0 stores them as bfloat in tgMin / tgMax [32]
    tgMin = bfloat(myMin.reshape(32, 32).min(axis=1))
    tgMax = bfloat(myMax.reshape(32, 32).max(axis=1))
    # simdgroup 0: lane l reads tgMin[l] / tgMax[l]; minBound = simd_min, maxBound = simd_max (threadgroup floats)
    minBound, maxBound = F(tgMin.min()), F(tgMax.max())
    newBound = F((minBound + maxBound) / F(2))
    loops = 0
    minCount, maxCount = 4096, 0                                              # `minCount = 4096;` and the threadgroup initialiser `maxCount = 0`
    while True:
        loops += 1
        myAbove = (myVal > newBound).sum(axis=1)                              # per thread, its four values
        tgAbove = myAbove.reshape(32, 32).sum(axis=1)                         # simd_sum per simdgroup
        countAbove = int(tgAbove.sum())                                       # simdgroup 0: simd_sum over tgAbove
        if countAbove < effort:
            maxBound, maxCount = newBound, countAbove
        else:
            minBound, minCount = newBound, countAbove
        newBound = F((maxBound + minBound) / F(2))
        globalCount = countAbove
        if globalCount == effort:
            return newBound, loops, "count"
        if F(maxBound - minBound) < F(0.00001):
            return newBound, loops, "bounds"
        if abs(maxCount - minCount) < 3:
            return newBound, loops, "counts"
        if loops > 100:
            return newBound, loops, "loops"


def prepare_dispatch(v, stats_half4, expNo, cutoff, chunkSize, rowsCount, colsCount, expertSize, nThreads):
    """kernel prepareDispatch (bucketMul.metal:47-79); thread id walks rows [chunkSize*id + off, +chunkSize).  The reference
    appends with an atomic counter (any order); threads are taken in id order here.  Returns float2 entries [n, 2]."""
    v = np.asarray(v, dtype=np.float32)
    st = np.asarray(stats_half4).view(np.float16).reshape(-1, 4)
    off = expertSize * expNo
    i = np.arange(off, off + chunkSize * nThreads)                            # all threads' rows, ascending = thread order
    s3 = st[i, 3].astype(np.float32)                                          # float(s[3])
    val = v[i % rowsCount]
    lhs = ((F(CUTOFF_SCALE) * s3).astype(np.float32) * np.abs(val)).astype(np.float32)     # CUTOFF_SCALE * float(s[3]) * abs(val), left to right
    keep = F(cutoff) < lhs
    rows = i[keep]
    return np.stack([val[keep], (rows.astype(np.uint32) * np.uint32(colsCount)).astype(np.float32)], axis=1)   # {val, float(i*colsCount)}


def round_up_and_zero(dispatch: np.ndarray, number: int = 2048, threads: int = 2048) -> np.ndarray:
    """roundUp: size = (1 + size/number) * number; zeroRange32 with `threads` threads zeroes [prev, new) (bucketMul.swift:57-58)."""
    prev = dispatch.shape[0]
    new = (1 + prev // number) * number
    assert new - prev <= threads
    return np.concatenate([dispatch, np.zeros((new - prev, 2), np.float32)])


def bucket_mul(weights_half: np.ndarray, dispatch: np.ndarray, cols: int, groups: int = 32) -> np.ndarray:
    """kernel bucketMul (bucketMul.metal:83-117), grid [cols, groups]: result[y*16384 + x*16 + i].  All threads of a group
    step through its dispatch slice together (vectorised over x), so every thread's sums see the rows in slice order."""
    w16 = np.asarray(weights_half).view(np.uint16).reshape(-1)
    D = dispatch.shape[0]
    per = D // groups
    result = np.zeros((groups, 16384), np.float32)
    xs = np.arange(cols)
    for y in range(groups):
        myVal = np.zeros((cols, 16), np.float32)
        rowOffset = y * D // groups
        for r in range(per):                                                  # (STEP only unrolls the loop)
            d0, d1 = dispatch[rowOffset + r]
            w = w16[int(d1) + xs]                                             # half w = weights[int(d[1]) + id.x]
            vv = (F(d0) * w.view(np.float16).astype(np.float32)).astype(np.float32)       # float v = d[0]*float(w)
            myVal[xs, w & 15] += vv                                           # myVal[pos] += v; the other fifteen += 0
        result[y, : cols * 16] = myVal.reshape(-1)
    return result


def bucket_integrate(tmpMulVec: np.ndarray, outDim: int) -> np.ndarray:
    """kernel bucketIntegrate (bucketMul.metal:122-137): out[i] = simd_sum over lanes tiisg of tmpMulVec[i + tiisg*16384].
    simd_sum's order is not specified; lanes are added in order 0..31 here (the C oracle uses a butterfly): outputs agree
    to rounding, not to the bit."""
    out = np.zeros(outDim, np.float32)
    for l in range(32):
        out = (out + tmpMulVec[l, :outDim]).astype(np.float32)
    return out


def full_mul(v, buckets, stats, probes, inDim: int, outDim: int, effort: float, expNo: int = 0, percentLoad: int = 16):
    """BucketMul.fullMul (bucketMul.swift:34-88).  Returns (out, dispatch count before padding, cutoff, loops, exit)."""
    q = int(float(4096 - 1) * (1 - effort))                                   # Int(Double(probesCount-1)*(1-effort))
    cutoff, loops, why = find_cutoff32(v, probes, expNo, q)
    statsRows = inDim * percentLoad                                           # ew.stats.rows per expert
    cols = outDim // 16
    disp = prepare_dispatch(v, stats, expNo, cutoff, 4, inDim, cols, percentLoad * inDim, statsRows // 4)
    n = disp.shape[0]
    padded = round_up_and_zero(disp)
    tmp = bucket_mul(buckets, padded, cols, 32)
    return bucket_integrate(tmp, outDim), n, float(cutoff), loops, why, disp
