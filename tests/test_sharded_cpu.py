"""world_size-2 gloo test of the multi-GPU path (column sharding + one all-gather) on CPU.

The sharding / gather logic of effort_amd.sharded is exercised with the CPU oracle injected as the multiply
backend (tests may use the oracle; the product default is the HIP path).  Property: every rank selects the
same rows (stats and probes are replicated), so the gathered output equals the single-device result
bit for bit.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.util import make_v, make_w

IN, OUT = 4096, 256


class _Local:
    def __init__(self, buckets, stats, probes, outDim):
        self.buckets, self.stats, self.probes, self.outDim = buckets, stats, probes, outDim


def _oracle_mul(v, by, out, effort, expNo):
    from oracle import cpu
    res, n, cutoff = cpu.bucket_mul(v.numpy(), by.buckets, by.stats, by.probes, IN, by.outDim, effort)
    out.copy_(torch.from_numpy(res))
    by.last = (n, cutoff)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from effort_amd.sharded import ShardedExpertWeights, shard_columns, shardedExpertMul, shardedExpertMulGroup
        from oracle import cpu
        v = torch.from_numpy(make_v(IN, seed=3))
        mats = []
        for seed in (21, 22):
            W = make_w(OUT, IN, seed=seed)
            buckets, stats, probes, _ = cpu.convert_fp16(W)
            full, n_full, cut_full = cpu.bucket_mul(v.numpy(), buckets, stats, probes, IN, OUT, 0.5)
            lb = shard_columns(torch.from_numpy(buckets.view(np.int16)), rank, world).numpy().view(np.float16)
            by = ShardedExpertWeights(_Local(lb, stats, probes, OUT // world), OUT, rank, world)
            mats.append((by, full, n_full, cut_full))
        # single matrix
        out = torch.zeros(OUT)
        shardedExpertMul(v, mats[0][0], out, 0.5, mul=_oracle_mul)
        ok1 = np.array_equal(out.numpy(), mats[0][1]) and mats[0][0].local.last == (mats[0][2], mats[0][3])
        # two matrices sharing v, ONE collective
        outs = [torch.zeros(OUT), torch.zeros(OUT)]
        shardedExpertMulGroup(v, [m[0] for m in mats], outs, 0.5, mul=_oracle_mul)
        ok2 = all(np.array_equal(o.numpy(), m[1]) for o, m in zip(outs, mats))
        q.put((rank, bool(ok1), bool(ok2)))
    finally:
        dist.destroy_process_group()


def test_sharded_bucketmul_two_ranks_gloo(oracle_cpu):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True, True), (1, True, True)]


def test_shard_helpers():
    from effort_amd.sharded import shard_columns, shard_outliers
    b = torch.arange(2 * 6 * 8).reshape(2, 6, 8)
    parts = [shard_columns(b, r, 4) for r in range(4)]
    assert torch.equal(torch.cat(parts, dim=-1), b)
    ol = torch.tensor([[0.5, 3, 0, 0], [1.5, 2, 65, 0], [-2.0, 1, 127, 0], [3.0, 0, 64, 0]])
    got = [shard_outliers(ol, r, 2, 128) for r in range(2)]
    assert got[0][:, 2].tolist() == [0] and got[1][:, 2].tolist() == [1, 63, 0]
