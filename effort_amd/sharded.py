"""Multi-GPU bucketMul: bucket-column (output) sharding + one RCCL all-gather.

The reference is single-device (helpers/gpu.swift:36-38); sharding is new.  One process per GPU
(torch.distributed, backend "nccl" == RCCL over xGMI).  Rank g holds bucket columns
[g*C/G, (g+1)*C/G) of every bucket row of a matrix, plus the FULL stats and probes -- those are
row-global (convert.metal:105-119), so every rank computes the identical cutoff and selects the identical
rows (dispatch count and cutoff are bit-identical), and the concatenation of the per-rank outputs is the single-GPU
result to the multiply's rounding (a shard takes the full matrix's fixed-point bound; its launch geometry -- hence the
grid its partial sums are rounded on -- may differ from the unsharded call's: |delta| <= 2e-5 max|out|).  The only exchange
is an all-gather of outDim/G f32 per rank (KB-scale: latency-bound over xGMI, never bandwidth-bound), so
matrices that share an input vector (Wq|Wk|Wv, W1|W3) are gathered together in ONE collective.

The shards and the collective are the C ABI's (include/effort_hip.h: effort_weights_column_shard -- a view, no copy --,
effort_comm_create, effort_allgather_outputs); this module is their host mirror.  ``init_comm`` uses torch.distributed only
to ship the communicator id between the processes the launcher started; the CPU tests inject the multiply and gather
through gloo.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch
import torch.distributed as dist


def shard_columns(buckets: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Columns [rank*C/world, (rank+1)*C/world) of a [..., rows, C] bucket tensor (any device)."""
    C = buckets.shape[-1]
    if C % world:
        raise ValueError(f"{C} bucket columns do not divide across {world} ranks")
    per = C // world                                 # any even split: 4096x11008 over 8 ranks = 86 columns (1376 outputs) each
    return buckets[..., rank * per:(rank + 1) * per].contiguous()


def shard_outliers(outliers: torch.Tensor | None, rank: int, world: int, outDim: int):
    """Q4 outliers (value, inIdx, outIdx, 0) whose output falls in this rank's slice, re-based to it."""
    if outliers is None:
        return None
    if outDim % world:
        raise ValueError(f"{outDim} outputs do not divide across {world} ranks")
    per = outDim // world
    lo, hi = rank * per, (rank + 1) * per
    m = (outliers[:, 2] >= lo) & (outliers[:, 2] < hi)
    ol = outliers[m].clone()
    ol[:, 2] -= lo
    return ol


class ShardedExpertWeights:
    """This rank's column shard of one matrix.  ``local`` is whatever the multiply backend consumes: an
    ``effort_amd.ExpertWeights`` on the GPU path."""

    def __init__(self, local, outSize: int, rank: int, world: int):
        self.local = local
        self.outSize = int(outSize)                  # full (gathered) output size
        self.rank, self.world = int(rank), int(world)
        if self.outSize % self.world:
            raise ValueError("outSize must divide across ranks")
        self.localOut = self.outSize // self.world

    @classmethod
    def from_full(cls, full, rank: int | None = None, world: int | None = None):
        """Shard a full GPU ExpertWeights (every rank holds or builds the same full bundle, keeps its slice)."""
        rank = dist.get_rank() if rank is None else rank
        world = dist.get_world_size() if world is None else world
        return cls(full.column_shard(rank, world), full.outSize, rank, world)


def _default_mul(v, by, out, effort, expNo):
    from .bucket_mul import expertMul
    expertMul(v, by, out, effort, expNo)


def _default_mul_group(v, bys, outs, effort, expNo):
    """The shards of matrices sharing ``v`` are independent calls: one grouped kernel launch (effort_bucketmul_group)
    when they are all bucketed bundles of one kind, expertMul one by one otherwise (dense fallback of expertMul.swift:29)."""
    from .bucket_mul import bucketMulGroup, expertMul
    same = all(b.q4 == bys[0].q4 for b in bys) and all((not b.q4) or b.bucketsLoaded for b in bys)
    if same and 1 < len(bys) <= 32:
        bucketMulGroup([(v, b, expNo, o, effort) for b, o in zip(bys, outs)])
    else:
        for b, o in zip(bys, outs):
            expertMul(v, b, o, effort, expNo)


def shardedExpertMulGroup(v: torch.Tensor, bys: Sequence[ShardedExpertWeights], outs: Sequence[torch.Tensor], effort: float = 0.25,
                          expNo: torch.Tensor | None = None, group=None,
                          mul: Callable = _default_mul, scratch: dict | None = None):
    """expertMul for several matrices sharing the input ``v`` with a single all-gather.

    Each rank multiplies its column shards into one contiguous send buffer [sum(localOut)], the collective
    gathers [world, sum(localOut)], and the slices are scattered to ``outs`` (each f32 [outSize])."""
    world = bys[0].world
    total = sum(b.localOut for b in bys)
    key = (v.device, total, world)
    sc = scratch if scratch is not None else _SCRATCH
    if key not in sc:
        sc[key] = (torch.empty(total, dtype=torch.float32, device=v.device),
                   torch.empty((world, total), dtype=torch.float32, device=v.device))
    send, recv = sc[key]
    off, pieces = 0, []
    for b in bys:
        pieces.append(send[off:off + b.localOut])
        off += b.localOut
    if mul is _default_mul:
        _default_mul_group(v, [b.local for b in bys], pieces, effort, expNo)
    else:
        for b, piece in zip(bys, pieces):
            mul(v, b.local, piece, effort, expNo)
    g = None
    if v.is_cuda and mul is _default_mul:
        from .runtime import gpu as _gpu
        g = _gpu(v.device.index)
    if g is not None and g.has_comm and g.comm_world == world:
        g.allgather_outputs(send, recv.view(-1), total)          # the C ABI's collective (effort_allgather_outputs: RCCL over xGMI)
    elif world == 1:
        recv[0].copy_(send)
    else:
        dist.all_gather_into_tensor(recv.view(-1), send, group=group)     # (CPU tests: gloo, with the multiply injected)
    off = 0
    for b, out in zip(bys, outs):
        out.view(world, b.localOut).copy_(recv[:, off:off + b.localOut])
        off += b.localOut


def shardedExpertMul(v: torch.Tensor, by: ShardedExpertWeights, out: torch.Tensor, effort: float = 0.25,
                     expNo: torch.Tensor | None = None, group=None, mul: Callable = _default_mul):
    """Sharded drop-in for expertMul(v:by:out:effort:) (expertMul.swift:20-38)."""
    shardedExpertMulGroup(v, [by], [out], effort, expNo, group, mul)


def init_comm(g=None, group=None):
    """Give the device's context its RCCL communicator (effort_comm_create), with torch.distributed as the host program that ships
    rank 0's 128-byte id to the other ranks.  One process per GPU; call once after dist.init_process_group."""
    from .runtime import gpu as _gpu
    g = g if g is not None else _gpu(torch.cuda.current_device())
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [g.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    g.comm_create(rank, world, box[0])
    return g


class ColumnShardedGroups:
    """The decode loop's launch groups with every matrix column-sharded over ``world`` ranks (BASELINE config 4): a group
    = the matrices that share an input vector (Wq|Wk|Wv, W1|W3; Wo and W2 alone -- runNetwork.swift:121-183), ONE grouped launch of
    this rank's shards and ONE all-gather (effort_allgather_outputs) per group.  The glue folded into the launches stays folded:
    the input prologues read full, replicated vectors; a residual epilogue adds this rank's SLICE of the residual, and when the
    output IS the residual vector (h += wo(attn), h += w2(...)) the gather runs in place on it -- no staging, no copies.

    ``ranks``: the ranks this process computes -- ``[rank]`` in a real run; ALL of them to EMULATE a world on one GPU (every rank's
    launch one after the other, their slices written where the gather would put them, no collective): identical results, and the
    way a column split's decode is validated and its per-rank launches timed where there is one GPU (``gather=False`` with one
    rank: that rank's kernels alone)."""

    def __init__(self, world: int, rank: int = 0, emulate: bool = False, gpu=None, gather: bool = True, mul_group=None, allgather=None):
        """``mul_group(calls)`` / ``allgather(send, recv_flat, count)``: the launch and the collective (default: bucketMulGroup on the
        context and its effort_allgather_outputs); the CPU tests inject a CPU multiply and gloo here."""
        self.world, self.rank, self.emulate, self.gather = int(world), int(rank), bool(emulate), bool(gather)
        self.ranks = list(range(self.world)) if emulate else [self.rank]
        self.g = gpu
        self._mul_group, self._allgather = mul_group, allgather
        self._shards: dict = {}
        self._bufs: dict = {}

    def _ctx(self, v):
        if self.g is None:
            from .runtime import gpu as _gpu
            self.g = _gpu(v.device.index)
        return self.g

    def shards(self, ew):
        """{rank: column shard} of a full bundle, made once (views: effort_weights_column_shard)."""
        k = id(ew)
        if k not in self._shards:
            self._shards[k] = (ew, {r: ew.column_shard(r, self.world) for r in self.ranks})
            for sh in self._shards[k][1].values():
                getattr(sh, "handle", None)                       # (registered now, not inside a graph capture)
        return self._shards[k][1]

    def _launch(self, v, calls):
        if self._mul_group is not None:
            return self._mul_group(calls)
        from .bucket_mul import bucketMulGroup
        bucketMulGroup(calls, gpu=self._ctx(v))

    def _collect(self, v, send, recv_flat, count):
        if self._allgather is not None:
            return self._allgather(send, recv_flat, count)
        self._ctx(v).allgather_outputs(send, recv_flat, count)

    def mul(self, v, items, effort):
        """items = [(full bundle, out (f32 [outSize], full), extras dict or None), ...] sharing the input ``v``."""
        W = self.world
        los = [ew.outSize // W for ew, _, _ in items]
        inplace = len(items) == 1 and bool(items[0][2]) and items[0][2].get("resid") is items[0][1]
        if inplace:                                               # h += product: every rank its slice of h, the gather in place on h
            ew, out, kw = items[0]
            lo = los[0]
            for r in self.ranks:
                sl = out[r * lo:(r + 1) * lo]
                self._launch(v, [(v, self.shards(ew)[r], None, sl, effort, dict(kw, resid=sl))])
            if not self.emulate and self.gather:
                self._collect(v, out[self.rank * lo:(self.rank + 1) * lo], out, lo)
            return
        total = sum(los)
        key = (v.device, tuple(los))
        if key not in self._bufs:
            self._bufs[key] = (torch.zeros(total, dtype=torch.float32, device=v.device), torch.zeros((W, total), dtype=torch.float32, device=v.device))
        send, recv = self._bufs[key]
        for r in self.ranks:
            calls, off = [], 0
            for (ew, out, kw), lo in zip(items, los):
                piece = recv[r, off:off + lo] if self.emulate else send[off:off + lo]
                kw2 = dict(kw) if kw else None
                if kw2 and "resid" in kw2:
                    kw2["resid"] = kw2["resid"][r * lo:(r + 1) * lo]
                calls.append((v, self.shards(ew)[r], None, piece, effort, kw2) if kw2 else (v, self.shards(ew)[r], None, piece, effort))
                off += lo
            self._launch(v, calls)
        if not self.emulate:
            if not self.gather:
                return
            self._collect(v, send, recv.view(-1), total)
        off = 0
        for (ew, out, kw), lo in zip(items, los):
            out.view(W, lo).copy_(recv[:, off:off + lo])
            off += lo


_SCRATCH: dict = {}
