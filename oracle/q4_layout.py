"""numpy restatement of the reference's Q4 weight layout (test infrastructure only).

Follows ``q4_draft.py:70-322`` (``convert(core2)``), vectorised.  Pinned by tests/golden/q4_*.npz,
which were produced by importing the reference's q4_draft.py in the authoring container
(tests/golden/make_q4_golden.py).

Input ``core2`` is ``W.T``: f16 ``[inDim, outDim]`` (q4_convert.py:54,63).
"""
from __future__ import annotations

import numpy as np


def extract_outliers(core: np.ndarray, perc: float = 0.02):
    """q4_draft.py:71-102.  The reference uses ``np.argsort(-abs)`` with numpy's default (unstable)
    kind, so the order of equal-|w| entries -- and which of several equal entries straddling the
    2 % boundary are taken -- is unspecified there; this restatement uses a stable sort (|w| desc,
    flat index asc)."""
    flat = core.flatten()
    absv = np.abs(flat)
    cnt = int(len(flat) * perc)
    order = np.argsort(-absv, kind="stable")[:cnt]
    rows, cols = np.unravel_index(order, core.shape)
    table = np.zeros((cnt, 4), np.float32)                      # rearrangeOutliers, :58-67
    table[:, 0] = flat[order]
    table[:, 1] = rows
    table[:, 2] = cols
    core = core.copy()
    core[rows, cols] = 0
    return table, core


def convert(core2: np.ndarray, perc: float = 0.02) -> dict:
    assert core2.dtype == np.float16 and core2.ndim == 2
    inDim, outDim = core2.shape
    assert outDim % 32 == 0                                      # :299 needs whole 16-bit words
    outliers, core = extract_outliers(np.ascontiguousarray(core2), perc)

    nb = outDim // 8
    r = core.reshape(inDim, nb, 8)
    # :117-134 -- per bucket of 8 outputs, argsort(-|w|); numpy sorts 8 elements by insertion -> stable
    idx = np.argsort(-np.abs(r), axis=-1, kind="stable")
    svals = np.take_along_axis(r, idx, axis=-1)                  # [inDim, nb, rank]
    # :147-168 -- output_rows[inRow*8 + rank] = [(value, bucket*8 + pos) for each bucket]
    vals_rows = np.ascontiguousarray(svals.transpose(0, 2, 1).reshape(inDim * 8, nb))
    pos_rows = np.ascontiguousarray(idx.transpose(0, 2, 1).reshape(inDim * 8, nb)).astype(np.uint16)

    # :179-194,244-245 -- avg |value| per row: np.mean over a float16 array (float32 pairwise
    # accumulation, float16 result), then stored as float32 in both lanes
    avg = np.empty(inDim * 8, np.float16)
    for i in range(inDim * 8):
        avg[i] = np.mean(np.abs(vals_rows[i]))
    bucket_stats = np.empty((inDim * 8, 2), np.float32)
    bucket_stats[:, 0] = avg
    bucket_stats[:, 1] = avg

    # :264-318 -- nibble = (8 if value < 0 else 0) + idx % 8 ; 4 nibbles per 16-bit word, first item highest
    nib = (vals_rows < 0).astype(np.uint16) * 8 + pos_rows
    words = (nib[:, 0::4] << 12) | (nib[:, 1::4] << 8) | (nib[:, 2::4] << 4) | nib[:, 3::4]
    buckets = np.ascontiguousarray(words.astype(np.uint16)).view(np.float16)

    return {
        "probes": np.diag(core).copy(),                          # :240,248 (after outlier removal)
        "bucket.stats": bucket_stats,
        "buckets": buckets,                                      # f16 view [inDim*8, outDim/32]
        "outliers": outliers,
        # extras (not reference outputs) used by the tests
        "_vals_rows": vals_rows,
        "_pos_rows": pos_rows,
        "_avg": avg,
    }


def draft_mul_no_effort(layout: dict, v: np.ndarray, outDim: int) -> np.ndarray:
    """The draft's effort-free multiply ``output_vector2`` (q4_draft.py:208-228): float64 accumulation of
    ``scalar * sign(value) * avg`` (sign(0) == 0 there, unlike the Metal kernel which treats a zero
    nibble as positive), outliers excluded."""
    vals_rows, pos_rows, avg = layout["_vals_rows"], layout["_pos_rows"], layout["_avg"]
    nb = vals_rows.shape[1]
    out = np.zeros(outDim)
    base = np.arange(nb) * 8
    for i in range(vals_rows.shape[0]):
        scalar = v[i // 8]
        contribution = scalar * np.sign(vals_rows[i]) * avg[i]
        out[base + pos_rows[i]] += contribution
    return out
