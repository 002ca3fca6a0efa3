#!/usr/bin/env python
"""bench.py -- bucketMul throughput on MI355X (driver contract: DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--group 32] [--effort 0.25]

Workload (BASELINE.json configs[1]): Mistral-7B-FFN-shaped matrix 4096 x 11008, fp16 buckets, bucketMul at
25 % effort (the north-star operating point), plus an effort sweep 10..100 % in the same JSON line.
One STEP = one pass of the hot path over one batch of synthetic input = 32 bucketMul calls, one per DISTINCT
converted matrix (rotation i % 32 exactly like benchmarks/benchmark.swift:206,255 -- 2.9 GB of buckets, so
reads come from HBM, not the 256 MB Infinity Cache), all on the same input vector, each writing its own output
vector.  Inputs are resident in HBM before the timed region.  The 32 calls of a step are replayed from ONE
hipGraph on ONE stream, so the host is not in the timed path (the reference's timeIt, helpers/timeit.swift:10-34,
likewise enqueues everything and waits once).  The calls of a step are independent (as Wq|Wk|Wv or W1|W3 are in the
decode loop), so they are issued `--group` at a time through effort_bucketmul_group: ONE kernel launch per group.
`by_group_size` in the output gives the same step at 1, 2, 3, 4, 8, 16 and 32 calls per launch; group size 1 is the
dependent-chain latency (every call waits for the previous one).  `two_streams` spreads the launches over two HIP
streams (the head of one launch then overlaps the tail of another).

value            = effective (dense-equivalent) GB/s = 2*inDim*outDim bytes per call / time per call, whole job
                   over all ranks.
tokens_per_s     = the reference's projection 1/(t_call * 4 * 32) (helpers/timeit.swift:26,33-34).
roofline         = dominant kernel (bucket_mul_kernel: a whole group of calls in one launch): algorithmic bytes per
                   launch / its average launch duration = the timed region / its launches.  Kernels of one stream do
                   not overlap, so this is also what `rocprofv3 --kernel-trace --stats` reports for the same command.
cpu_baseline     = the CPU oracle (a port: the reference ships no CPU path) on the host cores, bounded sample.
decode           = BASELINE.json configs[4]: end-to-end greedy decode of a random-init Mistral-7B-shaped model through
                   effort_amd/decode.py (one hipGraph per token): tokens/s dense vs effort 100 % / 25 %, KL vs dense.

N > 1 (one process per GPU, RCCL): independent matrices are partitioned across the ranks (every rank owns 32
distinct matrices; weak scaling) and the output vectors of a step are exchanged with ONE all-gather (north_star:
"partition independent weight matrices ... RCCL all-gather of the output vectors").  `--partition columns` runs
the bucket-column sharding of SURVEY 8e instead (strong scaling).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IN_DIM, OUT_DIM = 4096, 11008
N_MATS = 32
SWEEP = [0.10, 0.15, 0.20, 0.25, 0.30, 0.40, 0.50, 0.60, 0.70, 0.80, 0.90, 1.00]
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_FILE = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")   # written from a rocprofv3 --pmc pass (tools/pmc_traffic.py)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(D: int, inDim: int, outDim: int) -> int:
    """SURVEY 8d / BASELINE.md per-call bytes: kept bucket rows + stats + probes + v + out.  (The fused kernel
    writes no global dispatch list, so the reference formula's 8*D term is dropped.)"""
    return D * (outDim // 16) * 2 + 16 * inDim * 8 + 4096 * 2 + 4 * inDim + 4 * outDim


def mul_kernel_bytes(D: int, inDim: int, outDim: int) -> int:
    """What bucket_mul_kernel must move per CALL it serves (a launch serves `group` calls)."""
    return algorithmic_bytes(D, inDim, outDim)


def make_weights(ea, n, inDim, outDim, seed0, dev, keep_core=True, q4=False):
    ews = []
    gen = torch.Generator(device=dev)
    for k in range(n):
        gen.manual_seed(seed0 + k)
        W = (torch.randn((outDim, inDim), generator=gen, device=dev, dtype=torch.float32) * 0.02).to(torch.float16)
        if q4:                                       # product converter (effort_amd/q4.py, = q4_draft.convert)
            t = ea.q4_convert(W.t().contiguous())
            ew = ea.ExpertWeights(t["buckets"], t["bucket.stats"], t["probes"], inSize=inDim, outSize=outDim,
                                  outliers=t["outliers"], core=W, q4=True)
        else:
            ew = ea.ExpertWeights.from_core(W)      # product converter (GPU bucketize)
        if not keep_core:
            ew.core = None
        ew.handle
        ews.append(ew)
    ea.gpu().eval()
    return ews


class Step:
    """One step = one call per (matrix, output) item.  `group` calls per launch on the context's stream; with K > 1
    the launches are additionally spread round-robin over K HIP streams / contexts (used for the dense baseline,
    which has no grouped form).  capture() returns the step as one hipGraph."""

    def __init__(self, ea, device, K=1):
        self.ea, self.K = ea, K
        self.ctxs = [ea.gpu(device)] if K == 1 else [ea.Gpu(device) for _ in range(K)]
        self.streams = [None] if K == 1 else [torch.cuda.Stream(device=device) for _ in range(K)]

    def _enqueue(self, fn, chunks):
        if self.K == 1:
            for ch in chunks:
                fn(self.ctxs[0], ch)
            return
        s0 = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(s0)
        for i, ch in enumerate(chunks):
            with torch.cuda.stream(self.streams[i % self.K]):
                fn(self.ctxs[i % self.K], ch)
        for st in self.streams:
            s0.wait_stream(st)

    def capture(self, fn, chunks):
        self._enqueue(fn, chunks)                    # warm: handles, kernel attributes, rocBLAS workspaces
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enqueue(fn, chunks)
        for c in self.ctxs:
            c._bind_stream()
        return g


def chunked(items, n):
    return [items[i:i + n] for i in range(0, len(items), n)]


def time_replays(g, steps, warmup, barrier=None, after=None):
    for _ in range(warmup):
        g.replay()
        if after:
            after()
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
        if after:
            after()
    torch.cuda.synchronize()
    if barrier:
        barrier()
    return (time.perf_counter() - t0) / steps


def cpu_baseline(ews, v, effort, inDim, outDim, budget_s=12.0):
    """CPU oracle ("port") on the host cores: same converted weights (4 of the matrices), same v, same effort."""
    import numpy as np

    from oracle import cpu
    mats = []
    for ew in ews[:4]:
        mats.append((ew.buckets[0].cpu().numpy().view(np.float16), ew.stats[0].cpu().numpy().view(np.float16),
                     ew.probes[0].cpu().numpy().view(np.float16)))
    vh = v.cpu().numpy()
    sc = cpu.Scratch(inDim * 16)
    cpu.bucket_mul(vh, *mats[0], inDim, outDim, effort, scratch=sc)                # warm
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        out, D, _ = cpu.bucket_mul(vh, *mats[n % 4], inDim, outDim, effort, scratch=sc)
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(2 * inDim * outDim / dt / 1e9, 3), "unit": "GB/s", "cores": os.cpu_count(),
            "kind": "port", "us_per_call": round(dt * 1e6, 1),
            "sample": f"{n} bucketMul calls at effort {effort} over 4 of the {N_MATS} converted {inDim}x{outDim} matrices "
                      f"(OpenMP over bucket columns, {os.cpu_count()} threads), {budget_s:.0f} s budget"}, out, (n - 1) % 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--effort", type=float, default=0.25)
    ap.add_argument("--group", type=int, default=32, help="independent calls per kernel launch (1..32)")
    ap.add_argument("--partition", choices=["matrices", "columns"], default="matrices")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the end-to-end decode section (BASELINE.json configs[4])")
    ap.add_argument("--headline-only", action="store_true", help="only the timed job (for rocprofv3 passes: every bucket_mul_kernel dispatch is then the timed configuration)")
    ap.add_argument("--tune", default="0,0,0", help="waves,elems,slices of the multiply kernel (0,0,0 = heuristic)")
    args = ap.parse_args()
    G = max(1, min(32, args.group))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):      # (the env knob exercises the collective path on a 1-GPU box)
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rank != 0:
            os.dup2(2, 1)        # only rank 0 owns stdout (RCCL prints banners there); the others' goes to stderr
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import effort_amd as ea
    g = ea.gpu(local)
    g.set_tuning(*(int(x) for x in args.tune.split(",")))

    inDim, outDim = IN_DIM, OUT_DIM
    t_setup = time.perf_counter()
    columns = world > 1 and args.partition == "columns"
    seed0 = 1234 if (columns or world == 1) else 1234 + rank * N_MATS
    ews_full = make_weights(ea, N_MATS, inDim, outDim, seed0, dev, keep_core=(rank == 0))
    if columns:
        from effort_amd.sharded import ShardedExpertWeights
        ews = [ShardedExpertWeights.from_full(e, rank, world).local for e in ews_full]
        localOut = outDim // world
    else:
        ews, localOut = ews_full, outDim
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    v = torch.randn(inDim, generator=gen, device=dev, dtype=torch.float32)
    outs_all = torch.zeros((N_MATS, localOut), device=dev)
    outs = [outs_all[k] for k in range(N_MATS)]
    gathered = torch.zeros((world, N_MATS * localOut), device=dev) if dist else None
    torch.cuda.synchronize()
    log(f"[rank {rank}] setup {time.perf_counter() - t_setup:.1f} s: {N_MATS} matrices {inDim}x{outDim} converted on the GPU")

    def barrier():
        if dist:
            dist.barrier()

    def mul(effort):
        return lambda ctx, chunk: ea.bucketMulGroup([(v, ew, None, o, effort) for ew, o in chunk], gpu=ctx)

    items = list(zip(ews, outs))
    one = Step(ea, local)

    # ---------------- the timed job: K steps at the headline effort --------------------------------
    graph = one.capture(mul(args.effort), chunked(items, G))
    D = g.last_dispatch_count((N_MATS - 1) % G)
    if dist:
        # Pipelined exchange: step i's all-gather runs on a communication stream while step i+1 computes into the other
        # of two output buffers (the steps are independent); a buffer is recomputed only after its gather has finished.
        outs_b = torch.zeros((N_MATS, localOut), device=dev)
        graph_b = one.capture(mul(args.effort), chunked(list(zip(ews, [outs_b[k] for k in range(N_MATS)])), G))
        gathered_b = torch.zeros_like(gathered)
        comm = torch.cuda.Stream(device=dev)
        bufs = [(graph, outs_all, gathered, torch.cuda.Event(), torch.cuda.Event()),
                (graph_b, outs_b, gathered_b, torch.cuda.Event(), torch.cuda.Event())]

        def run(n):
            main = torch.cuda.current_stream()
            for i in range(n):
                gr, src, dst, computed, gathered_ev = bufs[i & 1]
                main.wait_event(gathered_ev)             # the previous gather out of this buffer is done
                gr.replay()
                computed.record(main)
                with torch.cuda.stream(comm):
                    comm.wait_event(computed)
                    dist.all_gather_into_tensor(dst.view(-1), src.view(-1))
                    gathered_ev.record(comm)
            main.wait_stream(comm)

        run(args.warmup)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        barrier()
        dt = (time.perf_counter() - t0) / args.steps
    else:
        dt = time_replays(graph, args.steps, args.warmup, barrier)
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    calls_per_step = N_MATS * (world if not columns else 1)          # whole job
    t_call = dt / N_MATS                                            # per-rank time per bucketMul call
    eff_bytes = 2 * inDim * outDim
    value = calls_per_step * eff_bytes / dt / 1e9

    result = {
        "metric": "effective GB/s + tokens/s vs effort %, Mistral-7B FFN 4096x11008 fp16",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 5), "higher_is_better": True, "scaling": "strong" if columns else "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"bucketMul {inDim}x{outDim} fp16 buckets, effort {args.effort}, {N_MATS} distinct matrices rotated "
                               f"(one call each per step), fixed-point accumulate (f32 out); {G} independent calls per fused "
                               f"kernel launch, one stream, one hipGraph", "effort": args.effort, "matrices_per_step": N_MATS,
                   "inDim": inDim, "outDim": outDim, "calls_per_launch": G, "kernel_geometry(waves,elems,slices)": args.tune if args.tune != "0,0,0" else "heuristic",
                   "partition": ("columns" if columns else "matrices") if world > 1 else "none", "dispatch_rows": D},
        "us_per_call": round(t_call * 1e6, 3),
        "tokens_per_s": round(1.0 / (t_call * 4 * 32), 2),
    }

    if rank == 0 and world == 1 and not args.headline_only:
        kb = mul_kernel_bytes(D, inDim, outDim)
        launches = (N_MATS + G - 1) // G
        # ---------------- roofline of the dominant kernel, in the timed configuration -----------------
        g.enable_kernel_timing(2)                        # device wall clock inside the kernel (graph safe)
        gt = one.capture(mul(args.effort), chunked(items, G))
        for _ in range(5):
            gt.replay()
        g.kernel_clock()
        for _ in range(20):
            gt.replay()
        kc = g.kernel_clock()
        kus, nl = kc["mul_us"], kc["launches"]
        del gt
        g.enable_kernel_timing(1)                        # HIP events on the launch stream, queue pre-filled
        torch.cuda._sleep(20_000_000)                    # keep the GPU busy while the host enqueues
        for r in range(4):
            for ch in chunked(items, G):
                ea.bucketMulGroup([(v, ew, None, o, args.effort) for ew, o in ch])
        evt = g.kernel_timing()
        g.enable_kernel_timing(0)
        traffic = None
        try:
            with open(PMC_FILE) as f:
                pmc = json.load(f)
            if pmc.get("calls_per_launch") == G and abs(pmc.get("effort", -1) - args.effort) < 1e-9:
                traffic = pmc["hbm_bytes_per_launch"]
        except Exception:
            pmc = None
        t_launch = dt / launches                         # launch-to-launch in the timed graph (includes the gap between kernels)
        # The kernel's average launch duration = the timed region / the launches it holds: one stream, back-to-back launches
        # of one hipGraph, so this is what HIP events around the region give and what `rocprofv3 --kernel-trace --stats`
        # reports per launch (profiles/).  The in-kernel device clock (first workgroup start -> last workgroup end) and
        # per-launch HIP events outside a graph are given beside it.
        t_kernel = t_launch
        result["roofline"] = {
            "bound": "hbm", "kernel": "bucket_mul_kernel", "achieved": round(G * kb / t_kernel / 1e9, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(G * kb / t_kernel / 1e9 / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "traffic_source": (f"rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE pass of this command ({os.path.relpath(PMC_FILE, ROOT)}), "
                               "corrected as MI355X_MICROARCH.md prescribes") if traffic else None,
            "calls_per_launch": G, "bytes_per_launch": G * kb, "bytes_per_call": kb, "kernel_us": round(t_kernel * 1e6, 3),
            "kernel_us_source": "timed region / launches (one stream, back-to-back launches of one hipGraph)",
            "kernel_us_device_clock": round(kus, 3), "launches_sampled_device_clock": nl,
            "frac_device_clock": round(G * kb / kus / 1e3 / HBM_PEAK_GBPS, 4),
            "kernel_us_hip_events_outside_graph": round(evt["mul_us"], 3),
        }
        # ---------------- the same step at other group sizes (1 = dependent-chain latency) ------------
        by = {}
        for n in (1, 2, 3, 4, 8, 16, 32):
            gn = one.capture(mul(args.effort), chunked(items, n))
            tn = time_replays(gn, 40, 10) / N_MATS
            by[str(n)] = {"us_per_call": round(tn * 1e6, 3), "effective_GBps": round(eff_bytes / tn / 1e9, 1),
                          "achieved_GBps": round(kb / tn / 1e9, 1), "tokens_per_s": round(1.0 / (tn * 4 * 32), 1)}
            del gn
        result["by_group_size"] = by
        two = Step(ea, local, 2)
        for c in two.ctxs:
            c.set_tuning(*(int(x) for x in args.tune.split(",")))
        tw = {}
        for n in (8, 16, 32):
            gn = two.capture(mul(args.effort), chunked(items, n))
            tn = time_replays(gn, 40, 10) / N_MATS
            tw[str(n)] = {"us_per_call": round(tn * 1e6, 3), "effective_GBps": round(eff_bytes / tn / 1e9, 1),
                          "achieved_GBps": round(kb / tn / 1e9, 1), "frac_of_hbm_peak": round(kb / tn / 1e9 / HBM_PEAK_GBPS, 4)}
            del gn
        result["two_streams"] = tw
        ts = by["1"]["us_per_call"] * 1e-6
        # ---------------- dense baseline (basicMul over the rotating cores) ---------------------------
        four = Step(ea, local, 4)
        dense_out = [torch.zeros(outDim, device=dev) for _ in range(4)]
        ditems = [[(ew, dense_out[i % 4])] for i, ew in enumerate(ews)]

        def dense(ctx, chunk):
            for ew, o in chunk:
                ea.basicMul(v, ew.core, o, gpu=ctx)
        # basicMul (helpers/mps.swift:14-47) twice: through rocBLAS' hssgemv -- the library the north star names -- and
        # through the package's own streaming kernel (csrc/gemv.hip), the default backend of effort_dense_gemv
        for name, rocblas in (("dense_rocblas", True), ("dense_hip_kernel", False)):
            for ctx in one.ctxs + four.ctxs:
                ctx.set_dense_backend(rocblas)
            td1 = time_replays(one.capture(dense, ditems), 30, 5) / N_MATS
            tdk = time_replays(four.capture(dense, ditems), 30, 5) / N_MATS
            result[name] = {"us_per_call_serial": round(td1 * 1e6, 3), "us_per_call_4_streams": round(tdk * 1e6, 3),
                            "GBps": round(eff_bytes / min(td1, tdk) / 1e9, 1), "frac_of_hbm_peak": round(eff_bytes / min(td1, tdk) / 1e9 / HBM_PEAK_GBPS, 4),
                            "speedup_at_effort": round(min(td1, tdk) / t_call, 3), "speedup_serial_vs_serial": round(td1 / ts, 3)}
        # ---------------- effort sweep ----------------------------------------------------------------
        if not args.no_sweep:
            sweep = []
            for e in SWEEP:
                ge = one.capture(mul(e), chunked(items, G))
                De = g.last_dispatch_count((N_MATS - 1) % G)
                te = time_replays(ge, 40, 10) / N_MATS
                ea.basicMul(v, ews[N_MATS - 1].core, dense_out[0])
                cs = ea.cosineSimilarityTo(outs[N_MATS - 1], dense_out[0])
                sweep.append({"effort": e, "dispatch_rows": De, "us_per_call": round(te * 1e6, 3),
                              "effective_GBps": round(eff_bytes / te / 1e9, 1),
                              "achieved_GBps": round(algorithmic_bytes(De, inDim, outDim) / te / 1e9, 1),
                              "frac_of_hbm_peak": round(algorithmic_bytes(De, inDim, outDim) / te / 1e9 / HBM_PEAK_GBPS, 4),
                              "tokens_per_s": round(1.0 / (te * 4 * 32), 1), "cos_vs_dense": round(cs, 5)})
                del ge
            result["sweep"] = sweep
            # heavy-tailed input (a real rms-normed state has outlier channels): v * exp(N(0,1)), same seeds
            vh = v * torch.exp(torch.randn(inDim, generator=gen, device=dev, dtype=torch.float32))
            heavy = {}
            for e in (0.25, 0.5):
                fh = lambda ctx, chunk: ea.bucketMulGroup([(vh, ew, None, o, e) for ew, o in chunk], gpu=ctx)    # noqa: E731
                gh = one.capture(fh, chunked(items, G))
                Dh = g.last_dispatch_count((N_MATS - 1) % G)
                th = time_replays(gh, 40, 10) / N_MATS
                ea.basicMul(vh, ews[N_MATS - 1].core, dense_out[0])
                heavy[str(e)] = {"dispatch_rows": Dh, "us_per_call": round(th * 1e6, 3),
                                 "achieved_GBps": round(algorithmic_bytes(Dh, inDim, outDim) / th / 1e9, 1),
                                 "frac_of_hbm_peak": round(algorithmic_bytes(Dh, inDim, outDim) / th / 1e9 / HBM_PEAK_GBPS, 4),
                                 "cos_vs_dense": round(ea.cosineSimilarityTo(outs[N_MATS - 1], dense_out[0]), 5)}
                del gh
            result["heavy_tailed_input"] = heavy
        # ---------------- the other shapes / formats BASELINE.json names ------------------------------------
        if not args.no_sweep:
            def quick(ews_x, outDim_x, inDim_x, effort, n, q4=False):
                vx = v if inDim_x == inDim else torch.randn(inDim_x, generator=gen, device=dev, dtype=torch.float32)
                ox = [torch.zeros(outDim_x, device=dev) for _ in ews_x]
                fnx = lambda ctx, ch: ea.bucketMulGroup([(vx, ew, None, o, effort) for ew, o in ch], gpu=ctx)    # noqa: E731
                gx = one.capture(fnx, chunked(list(zip(ews_x, ox)), n))
                Dx = g.last_dispatch_count((len(ews_x) - 1) % n)
                tx = time_replays(gx, 40, 10) / len(ews_x)
                del gx
                if q4:
                    nol = ews_x[0].outliers.shape[0]
                    ab = Dx * (outDim_x // 32) * 2 + 8 * inDim_x * 8 + 4096 * 2 + 4 * inDim_x + 4 * outDim_x + 16 * nol
                else:
                    ab = algorithmic_bytes(Dx, inDim_x, outDim_x)
                r = {"us_per_call": round(tx * 1e6, 3), "dispatch_rows": Dx, "effective_GBps": round(2 * inDim_x * outDim_x / tx / 1e9, 1),
                     "achieved_GBps": round(ab / tx / 1e9, 1), "frac_of_hbm_peak": round(ab / tx / 1e9 / HBM_PEAK_GBPS, 4)}
                if q4:       # SURVEY 8d prices an outlier at the reference's 16 bytes; the registered index holds 8
                    r["achieved_GBps_8B_outliers"] = round((ab - 8 * nol) / tx / 1e9, 1)
                return r
            other = {}
            sq = make_weights(ea, 16, 4096, 4096, 4321, dev, keep_core=False)
            other["4096x4096 fp16"] = {f"effort {e}, {n} per launch": quick(sq, 4096, 4096, e, n) for e in (0.5, 0.25) for n in (1, 16)}
            del sq
            dn = make_weights(ea, 16, 14336, 4096, 5321, dev, keep_core=False)       # W2 of the FFN: 14336 -> 4096
            other["14336x4096 fp16"] = {f"effort {e}, {n} per launch": quick(dn, 4096, 14336, e, n) for e in (0.25,) for n in (1, 16)}
            del dn
            q4w = make_weights(ea, 16, inDim, outDim, 6321, dev, keep_core=False, q4=True)
            other["4096x11008 q4 (bucketMulQ4, 2 % outliers)"] = {f"effort {e}, {n} per launch": quick(q4w, outDim, inDim, e, n, q4=True)
                                                                  for e in (0.25,) for n in (1, 16)}
            del q4w
            result["other_configs"] = other
        # ---------------- end-to-end greedy decode (BASELINE.json configs[4]; random-init Mistral-7B shapes) ----------
        if not args.no_decode and not args.no_sweep:
            try:
                from effort_amd.decode import Decoder, MistralConfig, Model, kl_divergence
                g.set_tuning(0, 0, 0)
                model = Model.random(MistralConfig(), seed=1)
                dec = Decoder(model, maxTokens=64)
                prompt, ntok = [1, 733, 16289, 28793, 22557], 48
                g.set_dense_backend(True)                    # the dense path through rocBLAS (the north star's baseline) ...
                _, dt_r, _ = dec.run(prompt, ntok, dense=True)
                g.set_dense_backend(False)                   # ... and through the package's own GEMV (also the LM head of the effort runs)
                dec._graphs.clear()
                ids_d, dt_d, lg_d = dec.run(prompt, ntok, dense=True, collect_logits=True)
                forced = prompt + ids_d[len(prompt) - 1:-1]
                dsec = {"model": "Mistral-7B shapes, 32 layers, random-init weights (no checkpoints offline)", "tokens": ntok,
                        "dense_rocblas_tokens_per_s": round(1 / dt_r, 1), "dense_hip_kernel_tokens_per_s": round(1 / dt_d, 1), "effort": {}}
                for e in (1.0, 0.25):
                    _, dt_e, _ = dec.run(prompt, ntok, effort=e)
                    _, _, lg_e = dec.run(forced, ntok, effort=e, forced=True, collect_logits=True)
                    dsec["effort"][str(e)] = {"tokens_per_s": round(1 / dt_e, 1), "ms_per_token": round(dt_e * 1e3, 3),
                                              "speedup_vs_dense_rocblas": round(dt_r / dt_e, 3), "speedup_vs_dense_hip_kernel": round(dt_d / dt_e, 3),
                                              "kl_vs_dense": round(kl_divergence(lg_d, lg_e), 5)}
                result["decode"] = dsec
                del dec, model
            except Exception as ex:
                result["decode"] = {"error": repr(ex)}
        # ---------------- CPU baseline -----------------------------------------------------------------
        if not args.no_cpu:
            try:
                import numpy as np
                cb, cpu_out, k_last = cpu_baseline(ews, v, args.effort, inDim, outDim)
                graph.replay()
                torch.cuda.synchronize()
                hip = outs[k_last].cpu().numpy()
                cb["gpu_vs_cpu_max_rel_err"] = float(np.abs(hip - cpu_out).max() / (np.abs(cpu_out).max() + 1e-30))
                result["cpu_baseline"] = cb
            except Exception as ex:  # the oracle is optional infrastructure for the bench
                result["cpu_baseline"] = {"error": repr(ex)}

    if rank == 0:
        # RCCL writes a version banner to C stdout, which (a pipe) only drains at exit: push it out first so that the JSON
        # line is the last thing this process prints
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(result), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
