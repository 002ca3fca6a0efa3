"""ctypes binding of oracle/effort_oracle.c (test infrastructure only; see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libeffort_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "effort_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libeffort_oracle.so"])
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.eo_h2f.restype = C.c_float
        _lib.eo_h2f.argtypes = [C.c_uint16]
        _lib.eo_f2h.restype = C.c_uint16
        _lib.eo_f2h.argtypes = [C.c_float]
        _lib.eo_bf16r.restype = C.c_float
        _lib.eo_bf16r.argtypes = [C.c_float]
        _lib.eo_effort_to_q.restype = C.c_uint32
        _lib.eo_effort_to_q.argtypes = [C.c_double]
        _lib.eo_cosine.restype = C.c_float
        _lib.eo_bucketmul_full.restype = C.c_int64
        _lib.eo_bucketmul_q4_full.restype = C.c_int64
        _lib.eo_prepare_dispatch.restype = C.c_uint32
        _lib.eo_prepare_dispatch_q4.restype = C.c_uint32
        _lib.eo_round_up_pad.restype = C.c_uint32
        _lib.eo_convert_fp16.restype = C.c_int
        _lib.eo_bucketize_row.restype = C.c_int
    return _lib


def _p(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _u16(a) -> np.ndarray:
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16:
        a = a.view(np.uint16)
    assert a.dtype == np.uint16
    return a


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- scalar helpers
def effort_to_q(effort: float) -> int:
    return int(lib().eo_effort_to_q(float(effort)))


def bf16r(x: float) -> float:
    return float(lib().eo_bf16r(float(x)))


# ---------------------------------------------------------------- converter
def convert_fp16(W: np.ndarray, bSize: int = 16):
    """bucketize() FP16 (convert.swift:209-260).  W: f16 [outDim, inDim] -> (buckets, stats, probes, oob)."""
    Wu = _u16(W)
    outDim, inDim = Wu.shape
    buckets = np.zeros((inDim * bSize, outDim // bSize), np.uint16)
    stats = np.zeros((inDim * bSize, 4), np.uint16)
    probes = np.zeros(4096, np.uint16)
    rc = lib().eo_convert_fp16(_p(Wu), C.c_uint32(outDim), C.c_uint32(inDim), C.c_uint32(bSize),
                               _p(buckets), _p(stats), _p(probes))
    if rc < 0:
        raise ValueError(f"bucketize precondition violated (code {rc})")
    return buckets.view(np.float16), stats.view(np.float16), probes.view(np.float16), rc


def bucketize_row(row_vals: np.ndarray, bSize: int):
    v = _u16(np.asarray(row_vals, dtype=np.float16))
    n = v.shape[0]
    ranked = np.zeros((bSize, n // bSize), np.uint16)
    rc = lib().eo_bucketize_row(_p(v), C.c_uint32(n), C.c_uint32(bSize), _p(ranked))
    return ranked.view(np.float16), rc


# ---------------------------------------------------------------- cutoff / dispatch
def find_cutoff(v, probes, expNo: int, effort: float):
    v = _f32(v)
    pr = _u16(probes).reshape(-1)
    cutoff = C.c_float(0)
    loops = C.c_int(0)
    lib().eo_find_cutoff(_p(v), _p(pr), C.c_uint32(expNo), C.c_uint32(effort_to_q(effort)),
                         C.byref(cutoff), C.byref(loops))
    return float(cutoff.value), int(loops.value)


def prepare_dispatch(v, stats, expNo: int, cutoff: float, inDim: int, cols: int, percentLoad: int = 16):
    """prepareDispatch in ascending bucket-row order.  stats: f16 [E*inDim*percentLoad, 4]."""
    v = _f32(v)
    st = _u16(stats).reshape(-1, 4)
    statsRows = inDim * percentLoad
    disp = np.zeros((statsRows + 2048, 2), np.float32)
    n = lib().eo_prepare_dispatch(_p(v), _p(st), C.c_uint32(expNo), C.c_float(cutoff), C.c_uint32(statsRows),
                                  C.c_uint32(inDim), C.c_uint32(cols), C.c_uint32(percentLoad * inDim), _p(disp))
    return disp, int(n)


def prepare_dispatch_q4(v, stats_f2, expNo: int, cutoff: float, inDim: int, cols: int):
    v = _f32(v)
    st = _f32(stats_f2).reshape(-1, 2)
    statsRows = inDim * 8
    disp = np.zeros((statsRows + 2048, 2), np.float32)
    n = lib().eo_prepare_dispatch_q4(_p(v), _p(st), C.c_uint32(expNo), C.c_float(cutoff), C.c_uint32(statsRows),
                                     C.c_uint32(cols), C.c_uint32(8 * inDim), _p(disp))
    return disp, int(n)


def round_up_pad(disp: np.ndarray, n: int) -> int:
    return int(lib().eo_round_up_pad(_p(disp), C.c_uint32(n)))


# ---------------------------------------------------------------- multiplies
class Scratch:
    """BucketMul singleton scratch (bucketMul.swift:19-32,52): dispatch + tmpMulVec."""

    def __init__(self, max_rows: int = 229376 * 2):
        self.dispatch = np.zeros((max_rows + 2048, 2), np.float32)
        self.tmp = np.zeros((32, 16384), np.float32)


def bucket_mul(v, buckets, stats, probes, inDim: int, outDim: int, effort: float, expNo: int = 0,
               percentLoad: int = 16, scratch: Scratch | None = None):
    """Full FP16 bucketMul (bucketMul.swift:11-90).  Returns (out f32[outDim], dispatchCount, cutoff)."""
    v = _f32(v)
    b = _u16(buckets)
    st = _u16(stats)
    pr = _u16(probes)
    sc = scratch or Scratch(inDim * percentLoad)
    out = np.zeros(outDim, np.float32)
    cutoff = C.c_float(0)
    n = lib().eo_bucketmul_full(_p(v), _p(b), _p(st), _p(pr), C.c_uint32(expNo), C.c_double(effort),
                                C.c_uint32(inDim), C.c_uint32(outDim), C.c_uint32(percentLoad),
                                _p(out), _p(sc.dispatch), _p(sc.tmp), C.byref(cutoff))
    if n < 0:
        raise ValueError("bucketMul precondition violated")
    return out, int(n), float(cutoff.value)


def bucket_mul_q4(v, buckets, stats_f2, probes, outliers, inDim: int, outDim: int, effort: float,
                  expNo: int = 0, scratch: Scratch | None = None):
    """Full Q4 call: out.zero() + bucketMulQ4 + calcOutliers (expertMul.swift:25-28, bucketMulQ4.swift:54-63)."""
    v = _f32(v)
    b = _u16(buckets)
    st = _f32(stats_f2)
    pr = _u16(probes)
    sc = scratch or Scratch(inDim * 8)
    out = np.zeros(outDim, np.float32)
    cutoff = C.c_float(0)
    if outliers is not None and len(outliers):
        ol = _f32(outliers)
        olp, nol = _p(ol), ol.shape[0]
    else:
        olp, nol = None, 0
    n = lib().eo_bucketmul_q4_full(_p(v), _p(b), _p(st), _p(pr), olp, C.c_uint64(nol), C.c_uint32(expNo),
                                   C.c_double(effort), C.c_uint32(inDim), C.c_uint32(outDim), _p(out),
                                   _p(sc.dispatch), C.byref(cutoff))
    if n < 0:
        raise ValueError("bucketMulQ4 precondition violated")
    return out, int(n), float(cutoff.value)


def bucket_mul_dispatch(buckets, disp: np.ndarray, D: int, cols: int, outDim: int) -> np.ndarray:
    """bucketMul + bucketIntegrate kernels on an explicit (already padded) dispatch list."""
    b = _u16(buckets)
    tmp = np.zeros((32, 16384), np.float32)
    out = np.zeros(outDim, np.float32)
    lib().eo_bucket_mul(_p(b), _p(disp), C.c_uint32(D), C.c_uint32(cols), C.c_uint32(32), _p(tmp))
    lib().eo_bucket_integrate(_p(tmp), _p(out), C.c_uint32(outDim))
    return out


def bucket_mul_q4_dispatch(buckets, disp: np.ndarray, D: int, cols: int, outDim: int) -> np.ndarray:
    """bucketMulQ4 kernel on an explicit (already padded) dispatch list, out pre-zeroed, no outliers."""
    b = _u16(buckets)
    out = np.zeros(outDim, np.float32)
    lib().eo_bucket_mul_q4(_p(b), _p(disp), C.c_uint32(D), C.c_uint32(cols), C.c_uint32(32), _p(out))
    return out


def dense_gemv(W, v, round_v_to_f16: bool = False):
    Wu = _u16(W)
    outDim, inDim = Wu.shape
    v = _f32(v)
    out = np.zeros(outDim, np.float32)
    lib().eo_dense_gemv(_p(Wu), _p(v), _p(out), C.c_uint32(outDim), C.c_uint32(inDim), C.c_int(int(round_v_to_f16)))
    return out


def cosine(a, b) -> float:
    a = _f32(a)
    b = _f32(b)
    return float(lib().eo_cosine(_p(a), _p(b), C.c_uint32(a.shape[0])))
