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


def test_column_sharding_of_11008_over_8_ranks(oracle_cpu):
    """The BASELINE shape through shard_columns / shard_outliers for a world of 8: 688 bucket columns -> 86 per rank (1376
    outputs: (outDim/16) % 4 != 0, which only the reference's own kernel minds).  Every rank computes the full call's
    cutoff and dispatch list from the replicated stats / probes; its columns of the product are bit-equal to the full
    product's (findCutoff32 / prepareDispatch / bucketMul / bucketIntegrate restated by the oracle, per column)."""
    from effort_amd.sharded import shard_columns, shard_outliers
    from oracle import cpu
    inDim, outDim, world = 4096, 11008, 8
    W = make_w(outDim, inDim, seed=5)
    buckets, stats, probes, _ = cpu.convert_fp16(W)
    v = make_v(inDim, seed=8, heavy=True)
    full, n_full, cutoff = cpu.bucket_mul(v, buckets, stats, probes, inDim, outDim, 0.25)
    cols = outDim // 16
    bt = torch.from_numpy(np.ascontiguousarray(buckets).view(np.int16).reshape(1, 16 * inDim, cols))
    got = []
    for r in range(world):
        lb = shard_columns(bt, r, world)
        assert lb.shape[-1] == 86
        lbn = lb.numpy().view(np.uint16).reshape(16 * inDim, 86)
        cut_r, _ = cpu.find_cutoff(v, probes, 0, 0.25)
        disp, n = cpu.prepare_dispatch(v, stats, 0, cut_r, inDim, 86)
        assert (n, cut_r) == (n_full, cutoff)
        D = cpu.round_up_pad(disp, n)
        got.append(cpu.bucket_mul_dispatch(lbn, disp, D, 86, 86 * 16))
    assert np.array_equal(np.concatenate(got), full)
    # outliers follow their output column, re-based to the shard
    rng = np.random.default_rng(1)
    ol = np.zeros((1000, 4), np.float32)
    ol[:, 0] = rng.normal(size=1000)
    ol[:, 1] = rng.integers(0, inDim, 1000)
    ol[:, 2] = rng.integers(0, outDim, 1000)
    parts = [shard_outliers(torch.from_numpy(ol), r, world, outDim) for r in range(world)]
    assert sum(p.shape[0] for p in parts) == 1000
    for r, p in enumerate(parts):
        assert ((p[:, 2] >= 0) & (p[:, 2] < outDim // world)).all()
        back = p.clone()
        back[:, 2] += r * (outDim // world)
        keep = (ol[:, 2] >= r * 1376) & (ol[:, 2] < (r + 1) * 1376)
        assert np.array_equal(back.numpy(), ol[keep])
