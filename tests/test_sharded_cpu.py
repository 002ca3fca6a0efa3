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


# ---- the decode loop's launch groups, column-sharded (ColumnShardedGroups): a layer's dependent chain over two gloo ranks --------
class _Bundle:
    """A CPU stand-in for ExpertWeights: the oracle's layout, column_shard() as shard_columns() makes it."""

    def __init__(self, buckets, stats, probes, inDim, outDim):
        self.buckets, self.stats, self.probes, self.inSize, self.outSize = buckets, stats, probes, inDim, outDim

    def column_shard(self, rank, world):
        from effort_amd.sharded import shard_columns
        cols = self.outSize // 16
        bt = torch.from_numpy(np.ascontiguousarray(self.buckets).view(np.int16).reshape(1, 16 * self.inSize, cols))
        lb = shard_columns(bt, rank, world).numpy().view(np.float16).reshape(16 * self.inSize, cols // world)
        return _Bundle(lb, self.stats, self.probes, self.inSize, self.outSize // world)


def _oracle_group(calls):
    """bucketMulGroup with the glue folded in, restated with the oracle: input prologues (matrix.metal:25-35, aux.metal:113-152),
    residual epilogue (runNetwork.swift:172,183)."""
    from oracle import cpu
    for c in calls:
        v, by, _, out, effort = c[:5]
        kw = c[5] if len(c) > 5 and c[5] else {}
        x = v.numpy().astype(np.float32)
        if "gate" in kw:
            x = (kw["gate"].numpy() * x / (np.float32(1.0) + np.exp(-x))).astype(np.float32)
        if "norm" in kw:
            x = ((x / np.sqrt(np.float32((x.astype(np.float64) ** 2).mean()) + np.float32(1e-5))) * kw["norm"].numpy().astype(np.float32)).astype(np.float32)
        res, n, cutoff = cpu.bucket_mul(x, by.buckets, by.stats, by.probes, by.inSize, by.outSize, effort)
        if "resid" in kw:
            res = res + kw["resid"].numpy()
        out.copy_(torch.from_numpy(res.astype(np.float32)))


def _layer(G, B, mats, effort):
    """wo -> w1|w3 -> w2 -> wq|wk|wv (runNetwork.swift:121-183), the decoder's sharded token step without the attention."""
    G.mul(B["attn"], [(mats["wo"], B["h"], {"resid": B["h"]})], effort)
    G.mul(B["h"], [(mats["w1"], B["x1"], {"norm": B["n1"]}), (mats["w3"], B["x3"], {"norm": B["n1"]})], effort)
    G.mul(B["x1"], [(mats["w2"], B["h"], {"gate": B["x3"], "resid": B["h"]})], effort)
    G.mul(B["h"], [(mats["wq"], B["xq"], {"norm": B["n2"]}), (mats["wk"], B["xk"], {"norm": B["n2"]})], effort)


def _layer_setup():
    from oracle import cpu
    shapes = {"wo": (IN, IN), "w1": (IN, IN), "w3": (IN, IN), "w2": (IN, IN), "wq": (IN, IN), "wk": (256, IN)}      # (out, in)
    mats = {}
    for i, (name, (o, ii)) in enumerate(shapes.items()):
        b, s, p, _ = cpu.convert_fp16(make_w(o, ii, seed=60 + i))
        mats[name] = _Bundle(b, s, p, ii, o)
    B = {"attn": torch.from_numpy(make_v(IN, seed=70)), "h": torch.from_numpy(make_v(IN, seed=71)), "x1": torch.zeros(IN), "x3": torch.zeros(IN),
         "xq": torch.zeros(IN), "xk": torch.zeros(256),
         "n1": torch.from_numpy((1.0 + 0.1 * make_v(IN, seed=72)).astype(np.float16)), "n2": torch.from_numpy((1.0 + 0.1 * make_v(IN, seed=73)).astype(np.float16))}
    return mats, B


def _layer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from effort_amd.sharded import ColumnShardedGroups
        mats, B = _layer_setup()
        ref = {k: t.clone() for k, t in B.items()}
        _layer(ColumnShardedGroups(1, 0, emulate=True, mul_group=_oracle_group), ref, mats, 0.5)           # one device: the full matrices
        gathers = []

        def allgather(send, recv_flat, count):
            gathers.append(count)
            dist.all_gather_into_tensor(recv_flat[:world * count], send[:count].clone())                      # (gloo; the product's is effort_allgather_outputs)
        _layer(ColumnShardedGroups(world, rank, mul_group=_oracle_group, allgather=allgather), B, mats, 0.5)
        ok = all(torch.equal(B[k], ref[k]) for k in ("h", "x1", "x3", "xq", "xk"))                            # same rows on every rank => the same bits
        emu = {k: t.clone() for k, t in _layer_setup()[1].items()}
        _layer(ColumnShardedGroups(world, emulate=True, mul_group=_oracle_group), emu, mats, 0.5)             # every rank in one process: no collective
        ok_emu = all(torch.equal(emu[k], ref[k]) for k in ("h", "x1", "x3", "xq", "xk"))
        q.put((rank, bool(ok), bool(ok_emu), gathers))
    finally:
        dist.destroy_process_group()


def test_column_sharded_layer_chain_two_ranks_gloo(oracle_cpu):
    """ColumnShardedGroups -- what the sharded Decoder and bench.py's layer_latency leg run -- over a world of two gloo ranks with the
    oracle as the multiply: a layer's dependent chain (residuals gathered IN PLACE on h, w1|w3 and wq|wk in ONE collective each)
    gives every rank the single-device vectors bit for bit; four collectives per layer; the one-process emulation of the world too."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_layer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True, True, [IN // 2, IN, IN // 2, IN // 2 + 128]), (1, True, True, [IN // 2, IN, IN // 2, IN // 2 + 128])]
