"""GPU tests at the decode loop's full layer shapes (Mistral-7B: 4096 / 14336 / 1024): a decoder layer's dependent multiplies -- wo ->
w1|w3 -> w2 -> wq|wk|wv of the next layer, rmsNorm / silu / residual folded in (runNetwork.swift:121-183) -- as the four launches
the decode loop issues, every call against the CPU oracle fed the GPU's own input vectors; and the multi-GPU entry points of the C
ABI as far as one GPU reaches.  (Round 4's one-chain-launch variant of the same seven calls lives on branch `chain-launch`.)"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.test_gpu_parity import DEV, close, ea  # noqa: F401  (ea: module fixture)

pytestmark = pytest.mark.gpu


def _host(ew):
    return (ew.buckets[0].contiguous().cpu().numpy().view(np.uint16), ew.stats[0].cpu().numpy().view(np.uint16),
            ew.probes[0].cpu().numpy().view(np.uint16))


@pytest.fixture(scope="module")
def layers(ea):
    from effort_amd.decode import MistralConfig, Model
    cfg = MistralConfig(numLayers=2)                                          # Mistral-7B shapes: 4096 / 14336 / 1024
    return Model.random(cfg, seed=11, keep_cores=False).layers


def _buffers(seed):
    gen = torch.Generator(device=DEV)
    gen.manual_seed(seed)
    f = lambda n, s=1.0: torch.randn(n, generator=gen, device=DEV) * s         # noqa: E731
    return {"attn": f(4096), "h": f(4096), "x1": f(14336), "x3": f(14336), "xq": f(4096), "xk": f(1024), "xv": f(1024)}


def _stages(L, Ln, B, e):
    return [[(B["attn"], L.wo, None, B["h"], e, {"resid": B["h"]})],
            [(B["h"], L.w1, None, B["x1"], e, {"norm": L.ffnNorm}), (B["h"], L.w3, None, B["x3"], e, {"norm": L.ffnNorm})],
            [(B["x1"], L.w2, None, B["h"], e, {"gate": B["x3"], "resid": B["h"]})],
            [(B["h"], Ln.wq, None, B["xq"], e, {"norm": Ln.attnNorm}), (B["h"], Ln.wk, None, B["xk"], e, {"norm": Ln.attnNorm}),
             (B["h"], Ln.wv, None, B["xv"], e, {"norm": Ln.attnNorm})]]


@pytest.mark.parametrize("effort", [0.0, 0.25, 0.6, 1.0])
def test_layer_launches_match_the_oracle(ea, oracle_cpu, layers, effort):
    L, Ln = layers
    g = ea.gpu()
    lib = ea.lib()
    P = lambda t: C.c_void_p(t.data_ptr())                                      # noqa: E731
    # --- the four launches, keeping every intermediate
    S = _buffers(3)
    h0 = S["h"].clone()
    snap, counts = {}, []
    for k, st in enumerate(_stages(L, Ln, S, effort)):
        ea.bucketMulGroup(st)
        g.eval()
        counts += [(g.last_dispatch_count(i), g.last_cutoff(i)) for i in range(len(st))]
        snap[k] = {n: t.clone() for n, t in S.items()}
    # --- every call against the oracle on the GPU's own input
    def oracle(ew, vin, want_out, resid=None):
        want, n, cutoff = oracle_cpu.bucket_mul(vin.cpu().numpy(), *_host(ew), ew.inSize, ew.outSize, effort)
        if resid is not None:
            want = want + resid.cpu().numpy()
        assert close(want_out.cpu().numpy(), want)
        return n, cutoff
    g._bind_stream()
    got = [oracle(L.wo, S["attn"], snap[0]["h"], h0)]
    hn = torch.zeros(4096, device=DEV)
    g.check(lib.effort_add_rmsnorm_mul(g.ctx, P(snap[0]["h"].clone()), None, P(L.ffnNorm), P(hn), 4096), "rmsnorm")
    got += [oracle(L.w1, hn, snap[1]["x1"]), oracle(L.w3, hn, snap[1]["x3"])]
    x2 = torch.zeros(14336, device=DEV)
    g.check(lib.effort_silu_mul(g.ctx, P(snap[1]["x1"]), P(snap[1]["x3"]), P(x2), 14336), "silu")
    got += [oracle(L.w2, x2, snap[2]["h"], snap[1]["h"])]
    g.check(lib.effort_add_rmsnorm_mul(g.ctx, P(snap[2]["h"].clone()), None, P(Ln.attnNorm), P(hn), 4096), "rmsnorm")
    got += [oracle(Ln.wq, hn, snap[3]["xq"]), oracle(Ln.wk, hn, snap[3]["xk"]), oracle(Ln.wv, hn, snap[3]["xv"])]
    assert got == counts                                                                          # dispatch.size and cutoff bits


def test_layer_replays_are_race_free(ea, layers):
    """The layer's four launches from a hipGraph, many replays back to back with the layers swapping roles (the buffers are
    rewritten every replay, so a stale line in an L1 or in another XCD's L2 shows up as a wrong bit): every replay's outputs
    equal the first's -- also with launches in flight on other lanes beside them."""
    L, Ln = layers
    g = ea.gpu()
    B = _buffers(9)
    init = {n: t.clone() for n, t in B.items()}

    def step():
        for n in B:
            B[n].copy_(init[n])
        for st in _stages(L, Ln, B, 0.25) + _stages(Ln, L, B, 0.25):       # a second layer consuming the first one's h
            ea.bucketMulGroup(st)
    step()
    g.eval()
    want = {n: t.clone() for n, t in B.items()}
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, capture_error_mode="thread_local"):
        step()
    g._bind_stream()
    for rep in range(30):
        gr.replay()
        if rep % 6 == 5:
            g.eval()
            for n in B:
                assert torch.equal(B[n], want[n]), (rep, n)
    g.set_overlap(4)
    try:
        junk = [torch.zeros(14336, device=DEV) for _ in range(8)]
        for rep in range(6):
            for j in junk:
                ea.bucketMul(init["attn"], L.w1, None, j, 0.5)
            step()
            g.eval()
            for n in B:
                assert torch.equal(B[n], want[n]), (rep, n)
    finally:
        g.set_overlap(1)


def test_c_abi_communicator_world_of_one(ea, oracle_cpu):
    """The multi-GPU entry points of the C ABI as far as one GPU reaches: an RCCL communicator of one rank
    (effort_comm_create), column shards as views of a registered bundle (effort_weights_column_shard: FP16, worlds 2 and 8),
    their products gathered by effort_allgather_outputs -- ncclAllGather on the context's stream -- and the gathered vector
    against the oracle's full product; selection bit-exact on every shard."""
    from effort_amd.sharded import ShardedExpertWeights, shardedExpertMulGroup
    from tests.test_gpu_parity import converted, devf, gpu_weights
    from tests.util import make_v
    inDim, outDim = 4096, 11008
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    g = ea.Gpu(0)
    g.comm_create(0, 1, ea.Gpu.comm_unique_id())
    assert g.has_comm and g.comm_world == 1 and g.comm_rank == 0
    v = make_v(inDim, seed=31, heavy=True)
    vd = devf(v)
    want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, 0.25)
    for world in (2, 8):
        got = torch.zeros(outDim, device=DEV)
        send = torch.zeros(outDim, device=DEV)
        per = outDim // world
        shards = [ew.column_shard(r, world) for r in range(world)]
        assert all(sh.outSize == per and sh.buckets.data_ptr() == ew.buckets.data_ptr() + r * per // 16 * 2 for r, sh in enumerate(shards))   # views
        for r, sh in enumerate(shards):                                    # what rank r would run; here one after the other
            ea.bucketMul(vd, sh, None, send[r * per:(r + 1) * per], 0.25, gpu=g)
            g.eval()
            assert g.last_dispatch_count() == n and g.last_cutoff() == cutoff
        g.allgather_outputs(send, got)                                     # world of one: the whole vector through RCCL
        g.eval()
        assert close(got.cpu().numpy(), want)
    with pytest.raises(ea.EffortError):                                       # a shard's bound is the full handle's: refresh that one
        shards[0].refresh()
    with pytest.raises((ValueError, ea.EffortError)):                         # 688 columns do not split over 3 ranks
        ew.column_shard(0, 3)
    g.comm_destroy()
    assert not g.has_comm
    g.close()


def test_column_shard_views_outlive_a_freed_parent(ea, oracle_cpu):
    """effort_weights_free on a full handle with live column-shard views must not leave them dangling (the header asks for the
    shards to be freed first; the library keeps the parent's registration data until its last view is gone), and a view's
    buffer descriptor ends at the END of the full allocation: the last rank's ragged last tile of the LAST row reads nothing
    past it (a 4096 x 11008 shard of a world of 8 has 86 columns: a 128-column tile's lanes 43..63 lie beyond the row)."""
    import ctypes as C
    from tests.test_gpu_parity import converted, dev16, devf
    from tests.util import make_v
    lib = ea.lib()
    g = ea.Gpu(0)
    inDim, outDim, world = 4096, 11008, 8
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    bd, sd, pd = dev16(b), dev16(s), dev16(p)
    full = lib.effort_weights_fp16(g.ctx, C.c_void_p(bd.data_ptr()), C.c_void_p(sd.data_ptr()), C.c_void_p(pd.data_ptr()), inDim, outDim, 16, 1)
    assert full
    views = [lib.effort_weights_column_shard(full, r, world) for r in range(world)]
    assert all(views)
    assert not lib.effort_weights_column_shard(views[0], 0, 2)                 # a view of a view is refused
    lib.effort_weights_free(full)                                                # deferred: the views still read its row means
    v = make_v(inDim, seed=5)
    vd = devf(v)
    want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, 1.0)      # effort 1.0: the last row of the buffer is streamed
    per = outDim // world
    got = torch.zeros(outDim, device=DEV)
    for r in range(world):
        g.check(lib.effort_bucketmul(g.ctx, views[r], C.c_void_p(vd.data_ptr()), None, C.c_void_p(got[r * per:].data_ptr()), 1.0), "bucketmul")
        g.eval()
        assert g.last_dispatch_count() == n
    assert close(got.cpu().numpy(), want)
    for vw in views:
        lib.effort_weights_free(vw)                                              # the last one frees the parent too
    g.close()
