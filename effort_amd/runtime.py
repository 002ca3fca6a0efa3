"""Host-side mirror of ``class Gpu`` (helpers/gpu.swift:19-222), reduced to what the bucketMul path uses.

The reference enqueues kernels into one serial compute encoder and runs them at ``gpu.eval()``
(helpers/gpu.swift:109-119).  Here work is enqueued on a HIP stream (PyTorch's current stream for the
device, so torch ops and bucketMul calls interleave in order) and ``eval()`` waits for it.
PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class Gpu:
    def __init__(self, device: int | None = None):
        if not torch.cuda.is_available():
            raise RuntimeError("effort_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self._lib = _lib.lib()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self.ctx = self._lib.effort_create(self.device, C.c_void_p(stream))
        self.has_comm = False
        if not self.ctx:
            raise RuntimeError("effort_create failed: " + (self._lib.effort_last_error(None) or b"").decode())
        self._stream = stream

    # -- helpers ---------------------------------------------------------------------------------
    def _bind_stream(self):
        """Follow torch's current stream.  ALL per-call scratch (partial tiles, arrival tickets, item queues, cutoffs) lives in
        this one context, so two streams must never run its launches concurrently: when the stream changes, the new stream
        first waits for the work already enqueued on the old one.  (Inside a graph capture the capture's own stream order
        applies, and no cross-stream wait may be recorded: independent multiplies that should overlap take one Gpu context
        each -- ``effort_amd.Gpu(device)`` -- as bench.py's Step does.)"""
        cur = torch.cuda.current_stream(self.device)
        stream = cur.cuda_stream
        if stream != self._stream:
            old = getattr(self, "_stream_obj", None)
            if old is not None and not torch.cuda.is_current_stream_capturing():
                try:
                    cur.wait_stream(old)
                except RuntimeError:
                    pass                                   # the old stream belonged to a finished capture
            self._lib.effort_set_stream(self.ctx, C.c_void_p(stream))
            self._stream = stream
        self._stream_obj = cur

    def check(self, rc: int, where: str):
        if rc != 0:
            detail = self._lib.effort_last_error(self.ctx)
            raise _lib.EffortError(rc, where, detail.decode() if detail else "")

    # -- reference surface -------------------------------------------------------------------------
    def eval(self):
        """gpu.eval(): commit + waitUntilCompleted."""
        self.check(self._lib.effort_sync(self.ctx), "gpu.eval")

    def set_overlap(self, lanes: int = 4):
        """Up to ``lanes`` independent multiply launches of this context in flight at once (effort_set_overlap): each goes to
        an internal stream with its own scratch, ordered after the context's stream and after earlier multiplies it depends
        on.  ``join()`` (or ``eval()``, or any other call on the context) makes the stream wait for them -- call it before
        consuming an output with your own stream work, and before ending a graph capture."""
        self._bind_stream()
        self.check(self._lib.effort_set_overlap(self.ctx, int(lanes)), "set_overlap")

    def join(self):
        self.check(self._lib.effort_join(self.ctx), "join")

    def set_row_reuse(self, reuse: bool = True):
        """Cache policy of the bucket-row stream (effort_set_row_reuse): non-temporal by default (a row is read once per call);
        ``reuse=True`` keeps the ordinary policy for launches in flight that read the SAME matrices (a batch on one set of
        weights).  Speed only."""
        self.check(self._lib.effort_set_row_reuse(self.ctx, int(bool(reuse))), "set_row_reuse")

    # -- multi-GPU: one process per GPU, RCCL over xGMI (effort_comm_*) -----------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0 makes the 128-byte id; the host program ships it to the other ranks (effort_comm_unique_id)."""
        buf = (C.c_char * 128)()
        rc = _lib.lib().effort_comm_unique_id(buf)
        if rc != 0:
            raise _lib.EffortError(rc, "comm_unique_id")
        return bytes(buf)

    def comm_create(self, rank: int, world: int, uid: bytes):
        """This context's RCCL communicator (collective over the world: every rank calls it with the same id)."""
        if len(uid) != 128:
            raise ValueError("the communicator id is 128 bytes")
        self._bind_stream()
        self.check(self._lib.effort_comm_create(self.ctx, int(rank), int(world), C.c_char_p(uid)), "comm_create")
        self.has_comm = True

    def comm_destroy(self):
        self.check(self._lib.effort_comm_destroy(self.ctx), "comm_destroy")
        self.has_comm = False

    @property
    def comm_world(self) -> int:
        return int(self._lib.effort_comm_world(self.ctx))

    @property
    def comm_rank(self) -> int:
        return int(self._lib.effort_comm_rank(self.ctx))

    def allgather_outputs(self, send: torch.Tensor, recv: torch.Tensor, count: int | None = None):
        """recv f32 [world][count] = every rank's ``count`` outputs, enqueued on the context's stream after the multiplies
        that wrote them (effort_allgather_outputs: ncclAllGather over xGMI)."""
        n = int(send.numel() if count is None else count)
        if not (send.is_cuda and recv.is_cuda and send.dtype == torch.float32 and recv.dtype == torch.float32 and send.is_contiguous()
                and recv.is_contiguous() and send.numel() >= n and recv.numel() >= n * self.comm_world):
            raise ValueError("allgather_outputs: contiguous f32 CUDA tensors, recv holding world * count elements")
        self._bind_stream()
        self.check(self._lib.effort_allgather_outputs(self.ctx, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), n), "allgather_outputs")

    def hook_lane(self, lane: int):
        """Point last_dispatch_count / last_cutoff / slice_counts at lane ``lane``'s last launch (overlap mode; test hook)."""
        self.check(self._lib.effort_debug_hook_lane(self.ctx, int(lane)), "hook_lane")

    # -- extras ------------------------------------------------------------------------------------
    def set_tuning(self, waves: int = 0, elems: int = 0, slices: int = 0):
        self.check(self._lib.effort_set_tuning(self.ctx, waves, elems, slices), "set_tuning")

    def set_split_cutoff(self, split: bool = True):
        self.check(self._lib.effort_set_split_cutoff(self.ctx, int(split)), "set_split_cutoff")

    def enable_kernel_timing(self, mode: int = 1):
        """1: HIP events + device clock, 2: device clock only (graph safe), 0: off."""
        self._bind_stream()
        self.check(self._lib.effort_enable_kernel_timing(self.ctx, int(mode)), "enable_kernel_timing")

    def kernel_clock(self):
        us, n = C.c_double(), C.c_int()
        self.check(self._lib.effort_kernel_clock(self.ctx, C.byref(us), C.byref(n)), "kernel_clock")
        return {"mul_us": us.value, "launches": n.value}

    def set_persistent(self, wg_per_cu: int = -1):
        self.check(self._lib.effort_set_persistent(self.ctx, int(wg_per_cu)), "set_persistent")

    def set_dense_backend(self, rocblas: bool = False):
        """basicMul through the library's hssgemv (True) or the streaming HIP kernel (False, default)."""
        self.check(self._lib.effort_set_dense_backend(self.ctx, int(bool(rocblas))), "set_dense_backend")

    def convert_status(self) -> int:
        """Elements the bucketize() calls since the last query could not place (effort_convert_status); reads and clears."""
        n = C.c_int(0)
        self.check(self._lib.effort_convert_status(self.ctx, C.byref(n)), "convert_status")
        return int(n.value)

    def debug_stamps(self):
        buf = (C.c_ulonglong * 32)()
        self.check(self._lib.effort_debug_stamps(self.ctx, buf), "debug_stamps")
        return list(buf)

    def slice_counts(self, idx: int = 0, n: int = 4096):
        """Kept rows per row slice of call ``idx`` of the most recent launch (test hook)."""
        buf = (C.c_uint32 * n)()
        k = self._lib.effort_debug_slice_counts(self.ctx, int(idx), buf, n)
        if k < 0:
            self.check(k, "slice_counts")
        return list(buf[:k])

    def debug_trace(self, n: int = 4096):
        """Per-item records of the most recent multiply launch in timing mode 3: list of 8-tuples of ints."""
        buf = (C.c_ulonglong * (12 * n))()
        self.check(self._lib.effort_debug_trace(self.ctx, buf, n), "debug_trace")
        return [tuple(buf[8 * i:8 * i + 8]) + tuple(buf[8 * n + 4 * i:8 * n + 4 * i + 4]) for i in range(n)]

    def kernel_timing(self):
        mul, cut, integ, n = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        self.check(self._lib.effort_kernel_timing(self.ctx, C.byref(mul), C.byref(cut), C.byref(integ), C.byref(n)), "kernel_timing")
        return {"mul_us": mul.value, "cutoff_us": cut.value, "integrate_us": integ.value, "samples": n.value}

    def last_dispatch_count(self, idx: int = 0) -> int:
        """dispatch.size of call ``idx`` of the most recent (group) launch."""
        n = C.c_uint32()
        self.check(self._lib.effort_group_dispatch_count(self.ctx, int(idx), C.byref(n)), "last_dispatch_count")
        return int(n.value)

    def last_cutoff(self, idx: int = 0) -> float:
        x = C.c_float()
        self.check(self._lib.effort_group_cutoff(self.ctx, int(idx), C.byref(x)), "last_cutoff")
        return float(x.value)

    def close(self):
        if getattr(self, "ctx", None):
            self._lib.effort_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_gpus: dict[int, Gpu] = {}


def gpu(device: int | None = None) -> Gpu:
    """The process-wide ``gpu`` object of the reference (helpers/gpu.swift:17), one per device."""
    if not torch.cuda.is_available():
        raise RuntimeError("effort_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")
    d = torch.cuda.current_device() if device is None else int(device)
    g = _gpus.get(d)
    if g is None:
        g = _gpus[d] = Gpu(d)
    return g
