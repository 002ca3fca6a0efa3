"""GPU tests of the features round 3 added around the multiply: launches of ONE context in flight together
(effort_set_overlap), the pitched bucket layout, weights refreshed under a captured graph, and the bench's timed
configuration as a whole.  Through the C ABI, against the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.test_gpu_parity import DEV, close, converted, dev16, devf, ea, gpu_weights  # noqa: F401  (ea: module fixture)
from tests.util import make_v

pytestmark = pytest.mark.gpu


def _host(ew):
    """Oracle inputs of a bundle: dense uint16 arrays (the buckets may be a pitched view)."""
    return (ew.buckets[0].contiguous().cpu().numpy().view(np.uint16), ew.stats[0].cpu().numpy().view(np.uint16),
            ew.probes[0].cpu().numpy().view(np.uint16))


def test_pitched_layout_is_bit_identical(ea, oracle_cpu):
    """from_core(aligned=True) has the converter write the bucket rows on whole 128-byte lines (effort_convert_fp16_pitched +
    effort_weights_fp16_pitched: 1376 -> 1408 bytes for 11008 outputs, no second copy): same layout values, same stats /
    probes, and bit-identical products, lone and in a 32-call persistent launch (compact means, E = 4)."""
    outDim, inDim = 11008, 4096
    gen = torch.Generator(device=DEV)
    gen.manual_seed(77)
    W = (torch.randn((outDim, inDim), generator=gen, device=DEV, dtype=torch.float32) * 0.02).to(torch.float16)
    dense, pitched = ea.ExpertWeights.from_core(W, aligned=False), ea.ExpertWeights.from_core(W, aligned=True)
    lib = ea.lib()
    assert pitched.rowPitch == 1408 and dense.rowPitch == 1376
    assert lib.effort_weights_row_pitch(pitched.handle) == 1408 and lib.effort_aligned_row_pitch(outDim) == 1408
    assert pitched.align_rows() == 1408                                       # nothing to do: no second copy
    assert torch.equal(dense.buckets, pitched.buckets) and torch.equal(dense.stats, pitched.stats) and torch.equal(dense.probes, pitched.probes)
    b, s, p = _host(pitched)
    g = ea.gpu()
    v = make_v(inDim, seed=5, heavy=True)
    vd = devf(v)
    for effort in (0.1, 0.25, 1.0):
        o1, o2 = torch.zeros(outDim, device=DEV), torch.zeros(outDim, device=DEV)
        ea.bucketMul(vd, dense, None, o1, effort)
        g.eval()
        n1, c1 = g.last_dispatch_count(), g.last_cutoff()
        ea.bucketMul(vd, pitched, None, o2, effort)
        g.eval()
        assert (g.last_dispatch_count(), g.last_cutoff()) == (n1, c1) and torch.equal(o1, o2), effort
    want, n, cutoff = oracle_cpu.bucket_mul(v, b, s, p, inDim, outDim, 0.25)
    outs = [torch.full((outDim,), float("nan"), device=DEV) for _ in range(32)]
    ea.bucketMulGroup([(vd, pitched if i & 1 else dense, None, outs[i], 0.25) for i in range(32)])
    g.eval()
    for i in range(32):
        assert g.last_dispatch_count(i) == n and g.last_cutoff(i) == cutoff and torch.equal(outs[i], outs[0]) and close(outs[i].cpu().numpy(), want), i
    # a pitch the library did not choose (a multiple of 8 bytes >= 2 * cols), and a bad one
    from effort_amd.convert import bucketize
    t = {}
    bucketize(W, "", t, rowPitch=1376 + 40)
    odd = ea.ExpertWeights(t["buckets"], t["bucket.stats"], t["probes"], inSize=inDim, outSize=outDim)
    assert odd.rowPitch == 1416
    o3, o4 = torch.zeros(outDim, device=DEV), torch.zeros(outDim, device=DEV)
    ea.bucketMul(vd, odd, None, o3, 0.25)
    ea.bucketMul(vd, dense, None, o4, 0.25)
    g.eval()
    assert torch.equal(o3, o4) and close(o3.cpu().numpy(), want) and g.last_dispatch_count() == n
    with pytest.raises(ValueError):
        bucketize(W, "", {}, rowPitch=1380)


def test_refresh_keeps_captured_graphs_valid(ea, oracle_cpu):
    """A launch copies the handle's pointers by value and a captured hipGraph bakes them in: effort_weights_refresh recomputes
    the bound and the compact means IN PLACE, so a graph captured before the refresh replays on the rewritten weights with
    the new bound (lone call: 8-byte stats path; 32-call persistent launch: compact means)."""
    outDim, inDim = 1024, 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ew = gpu_weights(ea, W, b, s, p)
    g = ea.gpu()
    v = make_v(inDim, seed=8)
    vd = devf(v)
    lone = torch.zeros(outDim, device=DEV)
    many = [torch.zeros(outDim, device=DEV) for _ in range(32)]

    def enqueue():
        ea.bucketMul(vd, ew, None, lone, 0.5)
        g.set_tuning(8, 1, 64)                           # 64 slices per call: a persistent launch with cutoff jobs and compact means
        ea.bucketMulGroup([(vd, ew, None, o, 0.5) for o in many])
        g.set_tuning(0, 0, 0)
    try:
        enqueue()
        g.eval()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            enqueue()
        g._bind_stream()
        W2 = (W.astype(np.float32) * 16).astype(np.float16)           # exact scaling: same order, same positions, 16x the bound
        b2, s2, p2, _ = oracle_cpu.convert_fp16(W2)
        ew.buckets.copy_(dev16(b2).reshape(ew.buckets.shape))
        ew.stats.copy_(dev16(s2).reshape(ew.stats.shape))
        ew.probes.copy_(dev16(p2).reshape(ew.probes.shape))
        ew.refresh()
        lone.fill_(float("nan"))
        for o in many:
            o.fill_(float("nan"))
        graph.replay()
        g.eval()
        want, n, cutoff = oracle_cpu.bucket_mul(v, b2, s2, p2, inDim, outDim, 0.5)
        assert close(lone.cpu().numpy(), want)
        for i in (0, 7, 31):
            assert g.last_dispatch_count(i) == n and g.last_cutoff(i) == cutoff and close(many[i].cpu().numpy(), want), i
    finally:
        g.set_tuning(0, 0, 0)


def test_overlap_orders_dependent_launches(ea, oracle_cpu):
    """effort_set_overlap: launches of ONE context go to internal lanes.  Independent ones may overlap; one that reads an
    earlier launch's output (RAW), overwrites it (WAW) or overwrites its input (WAR) is ordered after it.  A chain mixing all
    three, eager and replayed from a hipGraph, gives the bits of the same chain on a single lane."""
    inDim = outDim = 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    W2, b2, s2, p2 = converted(oracle_cpu, outDim, inDim, seed=99)
    ewA, ewB = gpu_weights(ea, W, b, s, p), gpu_weights(ea, W2, b2, s2, p2)
    ctx = ea.Gpu(0)
    v0 = devf(make_v(inDim, seed=3, heavy=True))
    bufs = {k: torch.zeros(inDim, device=DEV) for k in "vxyzw"}

    def chain():
        bufs["v"].copy_(v0)
        ea.bucketMul(bufs["v"], ewA, None, bufs["x"], 0.25, gpu=ctx)        # x = A v
        ea.bucketMul(bufs["v"], ewB, None, bufs["y"], 0.5, gpu=ctx)         # y = B v          (independent of the first: may overlap)
        ea.bucketMul(bufs["x"], ewB, None, bufs["z"], 0.25, gpu=ctx)        # z = B x          (RAW on x)
        ea.bucketMul(bufs["y"], ewA, None, bufs["v"], 0.25, gpu=ctx)        # v = A y          (WAR on v: launches 1, 2 read it; RAW on y)
        ea.bucketMul(bufs["z"], ewA, None, bufs["x"], 0.5, gpu=ctx)         # x = A z          (WAW on x, WAR vs launch 3; RAW on z)
        ea.bucketMulGroup([(bufs["v"], ewA, None, bufs["w"], 0.25), (bufs["x"], ewB, None, bufs["y"], 0.25)], gpu=ctx)   # RAW on v and x, WAW on y
        ctx.join()
    res = {}
    for lanes in (1, 4):
        ctx.set_overlap(lanes)
        for k in bufs:
            bufs[k].zero_()
        chain()
        ctx.eval()
        res[lanes] = {k: t.clone() for k, t in bufs.items()}
    for k in bufs:
        assert torch.equal(res[1][k], res[4][k]), k
    # the same from a graph (the lanes fork from and rejoin the capturing stream), twice
    ctx.set_overlap(4)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        chain()
    ctx._bind_stream()
    for _ in range(2):
        for k in "xyzw":
            bufs[k].fill_(float("nan"))
        graph.replay()
        ctx.eval()
        for k in bufs:
            assert torch.equal(res[1][k], bufs[k]), k
    # and against the oracle: the first link
    want, n, cutoff = oracle_cpu.bucket_mul(v0.cpu().numpy(), b2, s2, p2, inDim, outDim, 0.5)
    ctx.set_overlap(1)
    y = torch.zeros(outDim, device=DEV)
    ea.bucketMul(v0, ewB, None, y, 0.5, gpu=ctx)
    ctx.eval()
    assert close(y.cpu().numpy(), want) and ctx.last_dispatch_count() == n
    ctx.close()


def test_a_capture_that_begins_with_launches_pending_on_the_lanes(ea, oracle_cpu):
    """Lanes and hipGraph captures (effort_hip.h, effort_set_overlap).  Launches enqueued on the lanes BEFORE a capture of the
    context's own stream begins are real work on the lanes' queues; a launch inside the capture that depends on them would have to
    wait, from a capturing stream, on an event recorded outside the capture -- which stream capture forbids.  The library refuses it
    with a message that names the remedy (join before beginning the capture) instead of failing somewhere inside HIP, touches no stream
    doing so (the capture stays valid), and after the join the same capture goes through, bit-identical with one lane.  Lanes left
    pending by a capture that ENDED are dropped from the bookkeeping (the graph's own edges order them)."""
    inDim = outDim = 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim)
    ewA = gpu_weights(ea, W, b, s, p)
    st = torch.cuda.Stream(device=DEV)
    v0 = devf(make_v(inDim, seed=13, heavy=True))
    x, y, z = (torch.zeros(inDim, device=DEV) for _ in range(3))
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        ctx = ea.Gpu(0)                                                          # bound to `st`: the stream that is going to capture
        ctx.set_overlap(1)
        ea.bucketMul(v0, ewA, None, x, 0.25, gpu=ctx)
        ea.bucketMul(x, ewA, None, y, 0.5, gpu=ctx)
        ctx.eval()
        want_x, want_y = x.clone(), y.clone()
        ctx.set_overlap(4)
        x.zero_(); y.zero_()
        ea.bucketMul(v0, ewA, None, x, 0.25, gpu=ctx)                            # lane 0 now holds real work, not joined
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
            with pytest.raises(ea.EffortError, match="before beginning the capture"):
                ea.bucketMul(x, ewA, None, y, 0.5, gpu=ctx)                      # RAW on x: would need lane 0's pre-capture work
            with pytest.raises(ea.EffortError, match="before beginning the capture"):
                ctx.join()                                                       # ... and so would a join inside the capture
            z.zero_()                                                            # (something to capture: the refused calls left the capture untouched and valid)
        del graph
        ctx.join()                                                               # the remedy, outside the capture
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st, capture_error_mode="thread_local"):
            ea.bucketMul(x, ewA, None, y, 0.5, gpu=ctx)
            ctx.join()
        y.fill_(float("nan"))
        graph.replay()
        ctx.eval()
        assert torch.equal(x, want_x) and torch.equal(y, want_y)
        # a capture that ended with its lanes joined leaves nothing behind; one more eager launch and a join are ordinary
        ea.bucketMul(v0, ewA, None, z, 0.25, gpu=ctx)
        ctx.eval()
        assert torch.equal(z, want_x)
        ctx.close()


def test_long_dependent_chains_without_a_join(ea, oracle_cpu):
    """The reference's style is one gpu.eval() after arbitrarily many enqueues (helpers/gpu.swift:109-119).  Two interleaved
    chains x <- A x of 800 dependent lone calls each (1 600 launches, no effort_join in between) under effort_set_overlap(4):
    the lanes' hazard bookkeeping is bounded (1 024 recorded ranges per lane), and when it fills up the lanes are JOINED --
    the dependencies are kept, never dropped.  Bit-identical with the same chains on one lane, eager and with the whole
    chain captured into one hipGraph."""
    inDim = outDim = 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim, scale=1.0 / 64.0)          # gain ~ 1 per step: the values stay alive
    W2, b2, s2, p2 = converted(oracle_cpu, outDim, inDim, seed=99, scale=1.0 / 64.0)
    ewA, ewB = gpu_weights(ea, W, b, s, p), gpu_weights(ea, W2, b2, s2, p2)
    ctx = ea.Gpu(0)
    x0, u0 = devf(make_v(inDim, seed=3)), devf(make_v(inDim, seed=4, heavy=True))
    x, y, u, w = (torch.zeros(inDim, device=DEV) for _ in range(4))
    steps = 400                                                                 # two calls per chain and step

    def chains():
        x.copy_(x0); u.copy_(u0)
        for _ in range(steps):
            ea.bucketMul(x, ewA, None, y, 0.5, gpu=ctx)                         # y = A x   (RAW on x, WAR on y)
            ea.bucketMul(u, ewB, None, w, 0.5, gpu=ctx)                         # w = B u   (the other chain: another lane)
            ea.bucketMul(y, ewA, None, x, 0.5, gpu=ctx)                         # x = A y
            ea.bucketMul(w, ewB, None, u, 0.5, gpu=ctx)                         # u = B w
    res = {}
    for lanes in (1, 4):
        ctx.set_overlap(lanes)
        chains()
        ctx.eval()
        res[lanes] = (x.clone(), u.clone())
        assert torch.isfinite(res[lanes][0]).all() and float(res[lanes][0].abs().max()) > 0 and float(res[lanes][1].abs().max()) > 0
    assert torch.equal(res[1][0], res[4][0]) and torch.equal(res[1][1], res[4][1])
    ctx.set_overlap(4)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        chains()
        ctx.join()
    ctx._bind_stream()
    x.zero_(); u.zero_()
    graph.replay()
    ctx.eval()
    assert torch.equal(res[1][0], x) and torch.equal(res[1][1], u)
    # the first link against the oracle
    want, n, cutoff = oracle_cpu.bucket_mul(x0.cpu().numpy(), b, s, p, inDim, outDim, 0.5)
    ctx.set_overlap(1)
    ea.bucketMul(x0, ewA, None, y, 0.5, gpu=ctx)
    ctx.eval()
    assert close(y.cpu().numpy(), want) and ctx.last_dispatch_count() == n
    ctx.close()


def test_a_chain_moves_lanes_when_the_stream_keeps_answering_busy(ea, oracle_cpu):
    """A dependent chain lives on one lane and forks from the context's stream only when that stream is busy; four busy answers in
    a row (a lane that shares its hardware queue with the caller's stream makes the idle test lie: DESIGN 4.2 iv) and the chain
    MOVES to an idle lane, which takes the old lane's address ranges over.  Here the stream IS busy -- the caller parks a sleep
    kernel on it before every call -- so every link forks and the move happens; a second chain and a launch that depends on both run
    beside it.  Bit-identical with one lane; the order after the caller's own work is kept (the input is written by a copy the caller
    enqueues on the stream between the calls)."""
    inDim = outDim = 4096
    W, b, s, p = converted(oracle_cpu, outDim, inDim, scale=1.0 / 64.0)
    W2, b2, s2, p2 = converted(oracle_cpu, outDim, inDim, seed=77, scale=1.0 / 64.0)
    ewA, ewB = gpu_weights(ea, W, b, s, p), gpu_weights(ea, W2, b2, s2, p2)
    ctx = ea.Gpu(0)
    x0, u0 = devf(make_v(inDim, seed=5)), devf(make_v(inDim, seed=6))
    x, y, u, w, z, k = (torch.zeros(inDim, device=DEV) for _ in range(6))

    def run():
        x.copy_(x0); u.copy_(u0)
        for i in range(12):
            torch.cuda._sleep(200_000)                                          # the caller's stream is busy when the call arrives
            k.copy_(x if i % 2 == 0 else y)                                     # ... with work the launch must follow: its input
            ea.bucketMul(k, ewA, None, y if i % 2 == 0 else x, 0.5, gpu=ctx)    # one chain, ping-pong through x / y
            if i % 3 == 0:
                ea.bucketMul(u, ewB, None, w, 0.5, gpu=ctx)                     # another chain, elsewhere
                ea.bucketMul(w, ewB, None, u, 0.5, gpu=ctx)
        ea.bucketMul(x, ewB, None, z, 0.5, gpu=ctx)                             # depends on the moved chain
        ctx.eval()
        return x.clone(), y.clone(), u.clone(), z.clone()
    ctx._bind_stream()
    ctx.set_overlap(1)
    one = run()
    ctx.set_overlap(4)
    four = run()
    again = run()                                                               # (the move is allowed once between joins: a second batch moves again)
    for a_, b_, c_ in zip(one, four, again):
        assert torch.isfinite(a_).all() and float(a_.abs().max()) > 0
        assert torch.equal(a_, b_) and torch.equal(a_, c_)
    ctx.set_overlap(1)
    ctx.close()


def test_timed_configuration_as_a_whole(ea, oracle_cpu):
    """bench.py's timed job, as a whole: 4096 x 11008 matrices converted on line-aligned rows (pitch 1408), 32 calls per
    launch at 25 % effort on the heuristic geometry (persistent workgroups, cutoff jobs, COMPACT means, E = 4), ONE context
    with four launches in flight (effort_set_overlap), eight steps in one hipGraph, every step its own input vector and
    output set, the steps in flight on DISJOINT matrix sets -- every output of every step against the oracle, and the
    dispatch counts and cutoffs of each lane's last launch."""
    inDim, outDim, n_calls, lanes, steps, n_sets = 4096, 11008, 32, 4, 8, 2
    gen = torch.Generator(device=DEV)
    sets, host = [], []
    for k in range(n_sets * n_calls):
        gen.manual_seed(7000 + k)
        W = (torch.randn((outDim, inDim), generator=gen, device=DEV, dtype=torch.float32) * 0.02).to(torch.float16)
        ew = ea.ExpertWeights.from_core(W)                                    # aligned rows: what bench.make_weights does
        ew.core = None
        assert ew.rowPitch == 1408
        host.append(_host(ew))
        if k % n_calls == 0:
            sets.append([])
        sets[-1].append(ew)
    ctx = ea.Gpu(0)
    ctx.set_overlap(lanes)
    vs = [torch.zeros(inDim, device=DEV) for _ in range(steps)]
    outs = [[torch.zeros(outDim, device=DEV) for _ in range(n_calls)] for _ in range(steps)]

    def enqueue():
        for i in range(steps):
            ea.bucketMulGroup([(vs[i], sets[i % n_sets][k], None, outs[i][k], 0.25) for k in range(n_calls)], gpu=ctx)
        ctx.join()
    enqueue()
    ctx.eval()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        enqueue()
    ctx._bind_stream()
    hv = [make_v(inDim, seed=500 + i, heavy=bool(i & 1)) for i in range(steps)]
    for i in range(steps):
        vs[i].copy_(devf(hv[i]))
        for o in outs[i]:
            o.fill_(float("nan"))
    graph.replay()
    graph.replay()
    ctx.eval()
    for i in range(steps):
        for k in range(n_calls):
            want, n, cutoff = oracle_cpu.bucket_mul(hv[i], *host[(i % n_sets) * n_calls + k], inDim, outDim, 0.25)
            assert close(outs[i][k].cpu().numpy(), want), (i, k)
            if i >= steps - lanes:                                            # lane i % lanes ran step i last: its hooks are readable
                ctx.hook_lane(i % lanes)
                assert ctx.last_dispatch_count(k) == n and ctx.last_cutoff(k) == cutoff, (i, k)
    ctx.close()
