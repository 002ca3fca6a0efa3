"""Host-side mirror of the reference's multiply surface, forwarding to the C ABI.

    bucketMul(v:by:expNo:out:effort:)     bucketMul.swift:11-15
    bucketMulQ4(v:by:expNo:out:effort:)   bucketMulQ4.swift:11-17
    expertMul(v:by:[expNo:]out:effort:)   expertMul.swift:20-38
    basicMul(v:by:out:)                   helpers/mps.swift:14-20
    class BucketMul / BucketMulQ4         bucketMul.swift:18-90 / bucketMulQ4.swift:18-87

Argument meaning follows the reference: ``v`` / ``out`` are f32 device vectors (VectorFloat), ``by`` an
ExpertWeights, ``expNo`` a DEVICE scalar holding the expert index (ScalarFloat read as uint; ``None`` =
the reference's ``tmpExpZero``), ``effort`` a host float.  Calls enqueue on the current stream and
return; read results after ``gpu().eval()`` (or any stream-ordered torch op).  Violated preconditions
raise (the reference asserts).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .runtime import gpu as _gpu
from .weights import ExpertWeights


def _p(t: torch.Tensor | None):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _check_vec(name: str, t: torch.Tensor, n: int):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= n):
        raise ValueError(f"{name} must be a contiguous f32 CUDA vector with at least {n} elements")


def _check_expno(expNo: torch.Tensor | None):
    if expNo is not None and not (expNo.is_cuda and expNo.element_size() == 4 and expNo.numel() >= 1):
        raise ValueError("expNo must be a 4-byte device scalar (ScalarFloat reinterpreted as uint)")


def bucketMul(v: torch.Tensor, by: ExpertWeights, expNo: torch.Tensor | None, out: torch.Tensor, effort: float = 0.25,
              gpu=None):
    """``gpu``: an extra ``effort_amd.Gpu`` context (own scratch) to run independent multiplies concurrently on
    another stream; default = the device's shared context, i.e. the reference's BucketMul.shared."""
    (BucketMul.shared() if gpu is None else BucketMul(gpu.device, gpu)).fullMul(v, by, expNo, out, effort)


def bucketMulQ4(v: torch.Tensor, by: ExpertWeights, expNo: torch.Tensor | None, out: torch.Tensor, effort: float = 0.25,
                gpu=None):
    (BucketMulQ4.shared() if gpu is None else BucketMulQ4(gpu.device, gpu)).fullMul(v, by, expNo, out, effort)


def _marshal(calls, q4: bool):
    """ctypes arrays of a list of calls [(v, by, expNo, out, effort[, extras])]: (n, ws, vs, es, outs, eff, pre, aux, res);
    the last three are None when no call folds glue into the launch."""
    n = len(calls)
    P = C.c_void_p * n
    addr = lambda x: None if x is None else (x.data_ptr() if isinstance(x, torch.Tensor) else (x.value if hasattr(x, "value") else int(x)))   # noqa: E731
    ws = P(*[addr(c[1].handle) for c in calls])
    vs = P(*[addr(c[0]) for c in calls])
    es = P(*[addr(c[2]) for c in calls])
    outs = P(*[addr(c[3]) for c in calls])
    eff = (C.c_double * n)(*[float(c[4]) for c in calls])
    extras = [c[5] if len(c) > 5 and c[5] else {} for c in calls]
    if not any(extras):
        return n, ws, vs, es, outs, eff, None, None, None
    if q4:
        raise ValueError("fused prologues / epilogues are implemented for FP16 bundles")
    pre, aux, res = [], [], []
    for c, x in zip(calls, extras):
        if set(x) - {"gate", "norm", "resid"} or ("gate" in x and "norm" in x):
            raise ValueError("a call takes one of gate= / norm=, and optionally resid=")
        if "gate" in x:
            _check_vec("gate", x["gate"], c[1].inSize)
        if "norm" in x:
            w = x["norm"]
            if not (w.is_cuda and w.element_size() == 2 and w.is_contiguous() and w.numel() >= c[1].inSize):
                raise ValueError("norm weights must be a contiguous f16 CUDA vector of inSize elements")
        if "resid" in x:
            _check_vec("resid", x["resid"], c[1].outSize)
        pre.append(1 if "gate" in x else 2 if "norm" in x else 0)
        aux.append(addr(x.get("gate", x.get("norm"))))
        res.append(addr(x.get("resid")))
    return n, ws, vs, es, outs, eff, (C.c_int * n)(*pre), P(*aux), P(*res)


def bucketMulGroup(calls, gpu=None):
    """One launch for up to 32 independent multiplies: ``calls`` = [(v, by, expNo, out, effort), ...], all FP16 or all
    Q4 bundles.  Same results as calling bucketMul / bucketMulQ4 on each; the group is how independent projections of
    the decode loop (Wq|Wk|Wv, W1|W3 -- runNetwork.swift:132-134,178-182) keep the whole chip busy.

    A call may carry a 6th element, a dict folding the loop's neighbouring element-wise steps into the launch (FP16):
    ``{"gate": x3}`` -- input = silu(v) * x3 (runNetwork.swift:181); ``{"norm": w}`` -- input = rmsNorm(v) * w (:121-122);
    ``{"resid": h}`` -- out = h + product (:172,183; ``h`` may be ``out`` itself)."""
    calls = [tuple(c) for c in calls]
    if not 1 <= len(calls) <= 32:
        raise ValueError("a group holds 1..32 calls")
    q4 = calls[0][1].q4
    cls = BucketMulQ4 if q4 else BucketMul
    bm = cls.shared() if gpu is None else cls(gpu.device, gpu)
    for c in calls:
        bm._validate(c[0], c[1], c[2], c[3])
    n, ws, vs, es, outs, eff, pre, aux, res = _marshal(calls, q4)
    g = bm.gpu
    g._bind_stream()
    if pre is not None:
        g.check(_lib.lib().effort_bucketmul_group_fused(g.ctx, n, ws, vs, es, outs, eff, pre, aux, res), "bucketMulGroup")
        return
    fn = _lib.lib().effort_bucketmul_q4_group if q4 else _lib.lib().effort_bucketmul_group
    g.check(fn(g.ctx, n, ws, vs, es, outs, eff), "bucketMulGroup")


def basicMul(v: torch.Tensor, by: torch.Tensor, out: torch.Tensor, gpu=None):
    """Dense f16 GEMV, ``by`` = core matrix [outDim, inDim]; asserts of helpers/mps.swift:15-18."""
    assert by.shape[0] == out.numel() and by.shape[1] == v.numel() and by.shape[1] % 16 == 0
    _check_vec("v", v, by.shape[1])
    _check_vec("out", out, by.shape[0])
    assert by.is_cuda and by.element_size() == 2 and by.is_contiguous()
    g = gpu if gpu is not None else _gpu(v.device.index)
    g._bind_stream()
    g.check(_lib.lib().effort_dense_gemv(g.ctx, _p(by), _p(v), _p(out), by.shape[1], by.shape[0]), "basicMul")


def expertMul(v: torch.Tensor, by: ExpertWeights, out: torch.Tensor, effort: float = 0.25, expNo: torch.Tensor | None = None):
    """expertMul.swift:24-38: Q4 bundle -> (out.zero() +) bucketMulQ4, or dense basicMul when the buckets are
    missing; FP16 bundle -> bucketMul.  (Q8 was abandoned in the reference: it asserts false.)"""
    if by.q4:
        if by.bucketsLoaded:
            bucketMulQ4(v, by, expNo, out, effort)     # the out.zero() of :27 happens inside the C call
        else:
            basicMul(v, by.core, out)
    else:
        bucketMul(v, by, expNo, out, effort)


class BucketMul:
    """bucketMul.swift:18-90.  The reference's singleton owns dispatch / cutoff scratch; here that scratch
    lives in the per-device context, and ``dispatch`` is only materialised by ``calcDispatch``."""
    probesCount = 4096
    maxDispatchSize = 229376 * 2
    _q4 = False
    _shared: dict = {}

    @classmethod
    def shared(cls):
        d = torch.cuda.current_device()
        key = (cls, d)
        if key not in cls._shared:
            cls._shared[key] = cls(d)
        return cls._shared[key]

    def __init__(self, device: int, gpu=None):
        self.gpu = gpu if gpu is not None else _gpu(device)
        self.dispatch: torch.Tensor | None = None          # float2[] as [n, 2]
        self.dispatch_size: torch.Tensor | None = None      # dispatch.size (device u32)

    def _validate(self, v, ew, expNo, out):
        if ew.q4 != self._q4:
            raise ValueError("wrong weight kind for this multiply (FP16 vs Q4)")
        _check_vec("v", v, ew.inSize)
        if out is not None:
            _check_vec("out", out, ew.outSize)
        _check_expno(expNo)
        assert ew.probes.shape[-1] == 4096, "probes implemented for 4096 only"     # bucketMul.swift:36

    def calcDispatch(self, v, eWeights: ExpertWeights, expNo, effort: float):
        """bucketMul.swift:34-47: cutoff + dispatch list, materialised in the reference's float2 format
        (ascending bucket-row order).  The fused multiply does not need this; tests and tools do."""
        ew = eWeights
        self._validate(v, ew, expNo, None)
        rows = ew.inSize * ew.percentLoad
        if self.dispatch is None or self.dispatch.shape[0] < rows:
            self.dispatch = torch.empty((rows, 2), dtype=torch.float32, device=v.device)
            self.dispatch_size = torch.zeros(1, dtype=torch.int32, device=v.device)
        g = self.gpu
        g._bind_stream()
        g.check(_lib.lib().effort_calc_dispatch(g.ctx, ew.handle, _p(v), _p(expNo), float(effort), _p(self.dispatch),
                                                _p(self.dispatch_size)), "calcDispatch")

    def fullMul(self, v, ew: ExpertWeights, expNo, out, effort: float):
        self._validate(v, ew, expNo, out)
        g = self.gpu
        g._bind_stream()
        fn = _lib.lib().effort_bucketmul_q4 if self._q4 else _lib.lib().effort_bucketmul
        g.check(fn(g.ctx, ew.handle, _p(v), _p(expNo), _p(out), float(effort)), "bucketMulQ4" if self._q4 else "bucketMul")

    @property
    def cutoff(self) -> float:
        return self.gpu.last_cutoff()


class BucketMulQ4(BucketMul):
    _q4 = True


def cosineSimilarityTo(a: torch.Tensor, b: torch.Tensor) -> float:
    """VectorFloat.cosineSimilarityTo (model.swift:511-519)."""
    _check_vec("a", a, a.numel())
    _check_vec("b", b, a.numel())
    g = _gpu(a.device.index)
    g._bind_stream()
    x = C.c_float()
    g.check(_lib.lib().effort_cosine(g.ctx, _p(a), _p(b), a.numel(), C.byref(x)), "cosineSimilarityTo")
    return float(x.value)
