"""``ExpertWeights`` -- the per-matrix weight bundle of the reference (loader.swift:46-167).

Same fields and meaning: ``buckets`` [E, inSize*percentLoad, cols], ``stats`` [E, inSize*percentLoad, 4]
(Q4: f32 [E, inSize*8, 2]), ``probes`` [E, 4096], optional ``outliers`` f32 [n, 4] and dense ``core``
f16 [outSize, inSize]; ``expertSize = percentLoad*inSize`` (loader.swift:50).  Tensors live in HBM as
torch CUDA tensors (device memory plumbing only); the C-ABI handle borrows their pointers.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .runtime import gpu as _gpu


def _ptr(t: torch.Tensor | None):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class ExpertWeights:
    def __init__(self, buckets: torch.Tensor, stats: torch.Tensor, probes: torch.Tensor, inSize: int, outSize: int,
                 percentLoad: int | None = None, numExperts: int = 1, outliers: torch.Tensor | None = None,
                 core: torch.Tensor | None = None, q4: bool = False):
        self.q4 = bool(q4)                                    # goQ4 (main.swift:47) per bundle
        self.inSize = int(inSize)
        self.outSize = int(outSize)
        self.percentLoad = int(percentLoad) if percentLoad is not None else (8 if q4 else 16)   # loader.swift:64
        self.numExperts = int(numExperts)
        self.core = core
        self.outliers = outliers
        self.bucketsLoaded = True
        dev = buckets.device
        if dev.type != "cuda":
            raise ValueError("ExpertWeights tensors must be on the GPU")
        cols = self.outSize // 32 if q4 else self.outSize // 16
        rows = self.inSize * self.percentLoad
        # Matrix3D shapes of loader.swift:70,76 (bit-level dtype: 16-bit words / f16x4 or f32x2 stats)
        # ``buckets`` may be a view of PITCHED storage -- rows more than ``cols`` words apart, e.g. on whole 128-byte lines as
        # ``from_core`` / ``bucketize(rowPitch=...)`` write them (effort_weights_fp16_pitched): kept as is, [E, rows, cols]
        # strided; anything else is made dense like the reference's Matrix3D.
        b = buckets.view(torch.int16) if buckets.element_size() == 2 else buckets
        if b.dim() == 2:
            b = b.unsqueeze(0)
        if (not q4 and tuple(b.shape) == (self.numExperts, rows, cols) and b.stride(2) == 1 and b.stride(1) > cols
                and b.stride(1) % 4 == 0 and (self.numExperts == 1 or b.stride(0) == rows * b.stride(1))):
            self.buckets = b
            self.rowPitch = b.stride(1) * 2
        else:
            self.buckets = buckets.contiguous().view(torch.int16).reshape(self.numExperts, rows, cols)
            self.rowPitch = cols * 2
        if q4:
            self.stats = stats.contiguous().to(torch.float32).reshape(self.numExperts, rows, 2)
        else:
            self.stats = stats.contiguous().view(torch.int16).reshape(self.numExperts, rows, 4)
        self.probes = probes.contiguous().view(torch.int16).reshape(self.numExperts, 4096)
        if outliers is not None:
            self.outliers = outliers.contiguous().to(torch.float32).reshape(-1, 4)
        self._gpu = _gpu(dev.index)
        self._handle = None

    @property
    def expertSize(self) -> int:
        return self.percentLoad * self.inSize

    @property
    def handle(self):
        """effort_w* registered with the context (lazily, once)."""
        if self._handle is None:
            g, lib = self._gpu, _lib.lib()
            g._bind_stream()
            if self.q4:
                n = 0 if self.outliers is None else self.outliers.shape[0]
                h = lib.effort_weights_q4(g.ctx, _ptr(self.buckets), _ptr(self.stats), _ptr(self.probes), _ptr(self.outliers),
                                          n, self.inSize, self.outSize, self.numExperts)
            else:
                h = lib.effort_weights_fp16_pitched(g.ctx, _ptr(self.buckets), self.rowPitch, _ptr(self.stats), _ptr(self.probes),
                                                    self.inSize, self.outSize, self.percentLoad, self.numExperts)
            if not h:
                detail = lib.effort_last_error(g.ctx)
                raise _lib.EffortError(-2, "ExpertWeights", detail.decode() if detail else "")
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.lib().effort_weights_free(self._handle)
                self._handle = None
        except Exception:
            pass

    # -- constructors ------------------------------------------------------------------------------
    @classmethod
    def from_core(cls, core: torch.Tensor, aligned: bool = True) -> "ExpertWeights":
        """Dense f16 matrix [outSize, inSize] -> FP16 bundle via the GPU converter (= bucketize()).  ``aligned`` (default):
        the converter writes the bucket rows on whole 128-byte lines (same values, rows ``aligned_row_pitch`` bytes apart; the
        multiply streams them ~10 % faster than the reference's dense rows when HBM is saturated) -- ``buckets`` is then a
        strided view; ``aligned=False`` gives the reference's dense tensor."""
        from .convert import aligned_row_pitch, bucketize
        tensors: dict[str, torch.Tensor] = {}
        bucketize(core, "", tensors, goQ8=False, rowPitch=aligned_row_pitch(core.shape[0]) if aligned else 0)
        return cls(tensors["buckets"], tensors["bucket.stats"], tensors["probes"], inSize=core.shape[1],
                   outSize=core.shape[0], core=core)

    @classmethod
    def stack(cls, experts: list["ExpertWeights"]) -> "ExpertWeights":
        """Mixtral-style expert stacking: all experts in one buffer, selected by expNo (loader.swift:113-166)."""
        e0 = experts[0]
        return cls(torch.cat([e.buckets for e in experts]), torch.cat([e.stats for e in experts]),
                   torch.cat([e.probes for e in experts]), e0.inSize, e0.outSize, e0.percentLoad,
                   numExperts=sum(e.numExperts for e in experts), outliers=e0.outliers, q4=e0.q4)

    def truncated(self, percentLoad: int) -> "ExpertWeights":
        """Load only the first ``percentLoad`` rank slices (loader.swift:157-159, copyFrom(mySize: true))."""
        assert not self.q4 and 1 <= percentLoad <= self.percentLoad
        rows = self.inSize * percentLoad
        return ExpertWeights(self.buckets[:, :rows].contiguous(), self.stats[:, :rows].contiguous(), self.probes,
                             self.inSize, self.outSize, percentLoad, self.numExperts, core=self.core)

    def align_rows(self) -> int:
        """For bundles held in the reference's dense layout (loaded from a file, ``from_core(aligned=False)``): give the handle
        its own copy of the buckets with every row on a 128-byte line (effort_weights_align_rows).  No-op for bundles that are
        aligned already (``from_core`` writes them so).  Call before capturing launches into a graph.  Returns the row pitch."""
        self._gpu.check(_lib.lib().effort_weights_align_rows(self.handle), "ExpertWeights.align_rows")
        return int(_lib.lib().effort_weights_row_pitch(self.handle))

    def refresh(self):
        """The buffers were rewritten in place: re-read the bound the multiply's fixed-point scale comes from."""
        if self._handle is not None:
            self._gpu.check(_lib.lib().effort_weights_refresh(self._handle), "ExpertWeights.refresh")

    def rank_bound(self) -> list[float]:
        buf = (C.c_float * self.numExperts)()
        self._gpu.check(_lib.lib().effort_weights_get_bound(self.handle, buf), "ExpertWeights.rank_bound")
        return list(buf)

    def set_rank_bound(self, bound):
        buf = (C.c_float * self.numExperts)(*[float(x) for x in bound])
        self._gpu.check(_lib.lib().effort_weights_set_bound(self.handle, buf), "ExpertWeights.set_rank_bound")

    def column_shard(self, rank: int, world: int) -> "ExpertWeights":
        """Bucket-column (output) shard for multi-GPU (effort_weights_column_shard): columns [rank*C/G, (rank+1)*C/G) of every
        bucket row as a VIEW of this bundle's buffers -- no copy --, stats and probes shared (they are row-global, so every rank
        selects the same rows), the full matrix's fixed-point bound (every rank rounds on the same grid), and for Q4 the slice of
        the outlier index on those outputs.  Any split into an even number of columns per rank is valid (the multiply masks a
        ragged last tile: 11008 outputs over 8 ranks = 86 columns each).  Keep the full bundle alive while the shard is used."""
        from .sharded import shard_columns, shard_outliers
        lib = _lib.lib()
        unit = 32 if self.q4 else 16
        if (self.buckets.shape[2] // world) % 2:
            # an ODD number of columns per rank (Q4 only: 11008 outputs over 8 ranks = 43 words): a view would start rows on odd
            # 16-bit words, which the dword row loads cannot take -- this rank gets its own dense copy of its columns instead
            b = shard_columns(self.buckets, rank, world)
            per = b.shape[2]
            sh = ExpertWeights(b, self.stats, self.probes, self.inSize, per * unit, self.percentLoad, self.numExperts,
                               outliers=shard_outliers(self.outliers, rank, world, self.outSize),
                               core=None if self.core is None else self.core[rank * per * unit:(rank + 1) * per * unit], q4=self.q4)
            sh.set_rank_bound(self.rank_bound())
            return sh
        h = lib.effort_weights_column_shard(self.handle, int(rank), int(world))
        if not h:
            detail = lib.effort_last_error(self._gpu.ctx)
            raise _lib.EffortError(-2, "ExpertWeights.column_shard", detail.decode() if detail else "")
        per = self.buckets.shape[2] // world
        sh = object.__new__(ExpertWeights)
        sh.__dict__.update(self.__dict__)
        sh.buckets = self.buckets[:, :, rank * per:(rank + 1) * per]              # (a strided view, like the handle's)
        sh.outSize = per * unit
        sh.outliers = shard_outliers(self.outliers, rank, world, self.outSize)
        sh.core = None if self.core is None else self.core[rank * per * unit:(rank + 1) * per * unit]
        sh._handle = h
        sh._full = self                                                            # the view borrows the full bundle's buffers and index
        return sh
