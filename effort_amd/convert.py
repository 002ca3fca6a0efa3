"""``bucketize`` -- mirror of convert.swift:209-260 (FP16 part), running on the GPU through the C ABI.

    bucketize(w, outTensorsPref, tensors, goQ8=False)

fills ``tensors[pref+"bucket.stats"]`` (f16 [inDim*16, 4]), ``tensors[pref+"probes"]`` (f16 [4096]) and
``tensors[pref+"buckets"]`` (f16 [inDim*16, outDim/16]) exactly like the reference (convert.swift:253-259).
The Q8 tail of the reference (convert.swift:262-331) was abandoned there and is not provided.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .runtime import gpu as _gpu


def aligned_row_pitch(outDim: int) -> int:
    """Bytes between bucket rows that puts every row on a 128-byte line (effort_aligned_row_pitch)."""
    return (outDim // 16 * 2 + 127) // 128 * 128


def bucketize(w: torch.Tensor, outTensorsPref: str, tensors: dict, goQ8: bool = False, rowPitch: int = 0):
    """``rowPitch`` (bytes, 0 = dense): the buckets tensor is then a [inDim*16, outDim/16] VIEW of rows ``rowPitch`` bytes
    apart (``.contiguous()`` gives the reference's dense layout back); see effort_weights_fp16_pitched."""
    if goQ8:
        raise NotImplementedError("Q8 was abandoned in the reference (expertMul.swift:35); only the FP16 layout exists")
    if not (w.is_cuda and w.dtype == torch.float16 and w.dim() == 2 and w.is_contiguous()):
        raise ValueError("w must be a contiguous f16 CUDA matrix [outDim, inDim]")
    outDim, inDim = w.shape
    # convert.swift:210-215 preconditions are re-checked (and reported) by the C call
    cols = outDim // 16
    pitchCols = rowPitch // 2 if rowPitch else cols
    if pitchCols < cols or rowPitch % 8:
        raise ValueError("rowPitch must be a multiple of 8 bytes >= 2 * outDim/16")
    storage = torch.zeros((inDim * 16, pitchCols), dtype=torch.float16, device=w.device) if pitchCols != cols else \
        torch.empty((inDim * 16, cols), dtype=torch.float16, device=w.device)
    buckets = storage[:, :cols]
    stats = torch.empty((inDim * 16, 4), dtype=torch.float16, device=w.device)
    probes = torch.empty(4096, dtype=torch.float16, device=w.device)
    g = _gpu(w.device.index)
    g._bind_stream()
    p = lambda t: C.c_void_p(t.data_ptr())
    g.check(_lib.lib().effort_convert_fp16_pitched(g.ctx, p(w), outDim, inDim, p(storage), pitchCols * 2, p(stats), p(probes)), "bucketize")
    tensors[outTensorsPref + "bucket.stats"] = stats
    tensors[outTensorsPref + "probes"] = probes
    tensors[outTensorsPref + "buckets"] = buckets
