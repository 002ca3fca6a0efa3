"""Q4 weight layout converter -- mirror of the reference's ``q4_draft.convert(core2)`` (q4_draft.py:70-322).

The reference converts one matrix with pure-Python loops over every weight (minutes per matrix); this is the same
transformation as batched tensor operations (PyTorch on the GPU, or on the CPU for small inputs), bit-identical
in buckets, stats and probes and identical in the outlier set:

* the top 2 % weights by |w| become ``outliers`` f32 ``[n, 4]`` = (value, inIdx, outIdx, 0) and are zeroed in place
  (:71-102).  Ties in |w| are ordered by flat index (the reference's unstable argsort leaves them unspecified);
* per input row, every 8 consecutive outputs are sorted by |w| descending (stable, :117-134); rank row
  ``inRow*8 + rank`` collects one element per bucket (:147-168);
* ``bucket.stats`` = mean|rank row| as numpy computes ``np.mean`` of a float16 array -- float32 PAIRWISE
  accumulation, float32 division, float16 result -- stored as f32 in both lanes (:179-194,244-245).  numpy's
  pairwise order (blocks of 8 partial sums up to 128 elements, recursive halving above) is reproduced exactly;
* 4-bit codes ``(w < 0 ? 8 : 0) + idx % 8``, four per 16-bit word, first bucket in the top nibble (:264-318);
* ``probes`` = diagonal after outlier removal (:240).

``core2`` is ``W.T``: f16 ``[inDim, outDim]`` (q4_convert.py:54,63), ``outDim % 32 == 0``.
"""
from __future__ import annotations

import torch


def _np_pairwise_sum_f32(a: torch.Tensor) -> torch.Tensor:
    """numpy's pairwise summation (loops_utils.h.src, PW_BLOCKSIZE 128) over the last axis of a float32 [R, n]
    tensor, every operation in float32 in numpy's order."""
    n = a.shape[1]
    if n < 8:
        res = torch.zeros(a.shape[0], dtype=torch.float32, device=a.device)
        for i in range(n):
            res = res + a[:, i]
        return res
    if n <= 128:
        r = [a[:, k].clone() for k in range(8)]
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = r[k] + a[:, i + k]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[:, i]
            i += 1
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _np_pairwise_sum_f32(a[:, :n2]) + _np_pairwise_sum_f32(a[:, n2:])


def convert(core2: torch.Tensor, perc: float = 0.02) -> dict:
    """GPU tensors go through the C ABI (effort_convert_q4: HIP kernels, csrc/convert_q4.hip); CPU tensors through the
    tensor-op restatement below (the host mirror: small inputs, no GPU needed) -- both bit-identical to the reference."""
    if core2.dtype != torch.float16 or core2.dim() != 2:
        raise ValueError("core2 must be a float16 matrix [inDim, outDim] (= W.T)")
    inDim, outDim = core2.shape
    if outDim % 32:
        raise ValueError("outDim must be a multiple of 32 (q4_draft.py:299)")
    if core2.is_cuda:
        return _convert_hip(core2.contiguous(), perc)
    return convert_tensor_ops(core2, perc)


def _convert_hip(core2: torch.Tensor, perc: float) -> dict:
    import ctypes as C

    from . import _lib
    from .runtime import gpu as _gpu
    inDim, outDim = core2.shape
    dev = core2.device
    g, lib = _gpu(dev.index), _lib.lib()
    n = int(lib.effort_q4_outlier_count(inDim, outDim, float(perc)))
    out = {"buckets": torch.empty((inDim * 8, outDim // 32), dtype=torch.int16, device=dev),
           "bucket.stats": torch.empty((inDim * 8, 2), dtype=torch.float32, device=dev),
           "probes": torch.empty(min(inDim, outDim), dtype=torch.float16, device=dev),
           "outliers": torch.empty((n, 4), dtype=torch.float32, device=dev)}
    p = lambda t: C.c_void_p(t.data_ptr())                                       # noqa: E731
    g._bind_stream()
    g.check(lib.effort_convert_q4(g.ctx, p(core2), inDim, outDim, C.c_double(float(perc)), p(out["buckets"]), p(out["bucket.stats"]),
                                  p(out["probes"]), p(out["outliers"]) if n else None), "q4_convert")
    return out


def convert_tensor_ops(core2: torch.Tensor, perc: float = 0.02) -> dict:
    inDim, outDim = core2.shape
    core = core2.contiguous().clone()
    dev = core.device

    # ---- outliers (:71-102) ---------------------------------------------------------------------------
    flat = core.view(-1)
    cnt = int(flat.numel() * perc)
    # |w| descending, flat index ascending on ties: sort the magnitude bit patterns (monotone for halfs)
    mag = (flat.view(torch.int16).to(torch.int32) & 0x7FFF)
    order = torch.sort(-mag, stable=True).indices[:cnt]
    outliers = torch.zeros((cnt, 4), dtype=torch.float32, device=dev)
    outliers[:, 0] = flat[order].to(torch.float32)
    outliers[:, 1] = (order // outDim).to(torch.float32)
    outliers[:, 2] = (order % outDim).to(torch.float32)
    flat[order] = 0

    # ---- buckets of 8, sorted by |w| descending (stable) (:117-168) ---------------------------------------
    nb = outDim // 8
    r = core.view(inDim, nb, 8)
    rmag = r.view(torch.int16).to(torch.int32) & 0x7FFF
    idx = torch.sort(-rmag, dim=-1, stable=True).indices                       # [in, nb, rank] -> position
    svals = torch.gather(r, -1, idx)
    vals_rows = svals.transpose(1, 2).reshape(inDim * 8, nb).contiguous()     # rank row = inRow*8 + rank
    pos_rows = idx.transpose(1, 2).reshape(inDim * 8, nb).contiguous()

    # ---- stats: numpy's float16 mean (:179-194) --------------------------------------------------------
    s = _np_pairwise_sum_f32(vals_rows.abs().to(torch.float32))
    avg = (s / torch.tensor(float(nb), dtype=torch.float32, device=dev)).to(torch.float16)
    stats = torch.empty((inDim * 8, 2), dtype=torch.float32, device=dev)
    stats[:, 0] = avg.to(torch.float32)
    stats[:, 1] = stats[:, 0]

    # ---- 4-bit codes, 4 per word, first bucket in the top nibble (:264-318) ------------------------------
    nib = (vals_rows < 0).to(torch.int32) * 8 + pos_rows.to(torch.int32)
    words = (nib[:, 0::4] << 12) | (nib[:, 1::4] << 8) | (nib[:, 2::4] << 4) | nib[:, 3::4]
    words = torch.where(words >= 32768, words - 65536, words).to(torch.int16).contiguous()   # u16 bit pattern

    return {
        "probes": torch.diagonal(core).contiguous(),                             # :240 (after outlier removal)
        "bucket.stats": stats,
        "buckets": words,                                                        # int16 view of the u16 words
        "outliers": outliers,
    }
