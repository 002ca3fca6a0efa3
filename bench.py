#!/usr/bin/env python
"""bench.py -- bucketMul throughput on MI355X (driver contract: DESIGN.md "Measurement").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--group 32] [--streams 4] [--effort 0.25]

Workload (BASELINE.json configs[1]): Mistral-7B-FFN-shaped matrix 4096 x 11008, fp16 buckets, bucketMul at 25 % effort (the
north-star operating point), plus an effort sweep 10..100 % in the full record.  One STEP = one pass of the hot path over one
batch of synthetic input = 32 bucketMul calls, one per DISTINCT converted matrix (rotation i % 32 exactly like
benchmarks/benchmark.swift:206,255), all on the same input vector, each writing its own output vector; inputs are resident in
HBM before the timed region.  The calls of a step are independent (as Wq|Wk|Wv or W1|W3 are in the decode loop): they are
issued `--group` at a time through effort_bucketmul_group, ONE kernel launch per group.  The K steps of the job are independent
too: they go through ONE context with effort_set_overlap(S) -- the library keeps up to S = `--streams` launches in flight on
its own lanes -- and every step in flight multiplies its OWN 32 matrices (S x 32 distinct matrices, 11.8 GB: nothing a
concurrent launch reads can come out of the 256 MB Infinity Cache on another launch's behalf).  The job is one hipGraph,
repeated until the timed region is >= 50 ms; the host is not in the timed path (the reference's timeIt,
helpers/timeit.swift:10-34, likewise enqueues everything and waits once).  After the timed replays EVERY output set the timed
graph wrote is checked against the CPU oracle (S x 32 outputs).

What this file measures (the driver's one command): the headline, `roofline` (+ the in-run PMC traffic passes, `single_stream`),
`cpu_baseline`, `by_group_size`, the effort sweep, the dense baselines, config 0/1's shape and config 3 (Q4), the decode
scalars, a slim `timeit_protocol`.  Everything else round 5's bench carried -- by_streams, four_contexts, shared_matrices,
shard_projection, the timeit variants, quality sweeps, the one-GPU layer_latency -- lives in tools/bench_extra.py (writes
gpurun_out/bench_extra.json).

value            = effective (dense-equivalent) GB/s = 2*inDim*outDim bytes per call / time per call, whole job over all ranks.
tokens_per_s     = the reference's projection 1/(t_call * 4 * 32) (helpers/timeit.swift:26,33-34).
roofline         = dominant kernel (bucket_mul_kernel: a whole group of calls in one launch).  achieved = algorithmic bytes the
                   timed region moved / the timed region; `single_stream` gives the same job with ONE launch in flight (launch
                   duration = timed region / launches, which is what `rocprofv3 --kernel-trace --stats` reports: profiles/).
steps            = K as asked for (the driver's flag); `timed_steps` = K x repetitions is what the timed region really holds
                   (`timed_region_ms`), `steps_requested` repeats K.

N > 1: `python bench.py --gpus N` STARTS ITS N RANKS ITSELF when no launcher did (WORLD_SIZE unset): one process per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set, rank 0's line relayed; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` the launcher's environment is honoured as before.
Partition (`config.partition`): "matrices" -- independent matrices on different ranks (every rank its own S x 32: weak scaling;
north_star: "partition independent weight matrices ... RCCL all-gather of the output vectors") is `value`; the bucket-column
split of SURVEY 8e ("columns", strong scaling) and config 4's per-layer latency case are measured beside it (`multi_gpu`).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # the CPU oracle's OpenMP workers must sleep, not spin, between its calls: timed GPU sections follow
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "effective GB/s + tokens/s vs effort %, Mistral-7B FFN 4096x11008 fp16"
IN_DIM, OUT_DIM = 4096, 11008
N_MATS = 32
SWEEP = [0.10, 0.15, 0.20, 0.25, 0.30, 0.40, 0.50, 0.60, 0.70, 0.80, 0.90, 1.00]
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PMC_FILES = [os.path.join(ROOT, "profiles", f"r0{r}_pmc_traffic.json") for r in (6, 5, 4, 3, 2)]   # rocprofv3 --pmc passes folded by tools/pmc_traffic.py
COMPACT_LIMIT = 4096        # bytes: the driver keeps ~8 KB of stdout tail; the last line must fit with room to spare
AUX_DEADLINE_S = int(os.environ.get("BENCH_AUX_DEADLINE_S", "420"))      # N > 1: the auxiliary legs' budget after the headline (see aux_overdue)
ALIGN_ROWS = True    # (--no-align) the converter writes the bucket rows on whole 128-byte lines (effort_convert_fp16_pitched): no second copy


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(D: int, inDim: int, outDim: int) -> int:
    """SURVEY 8d / BASELINE.md per-call bytes: kept bucket rows + stats + probes + v + out.  (The fused kernel
    writes no global dispatch list, so the reference formula's 8*D term is dropped.)"""
    return D * (outDim // 16) * 2 + 16 * inDim * 8 + 4096 * 2 + 4 * inDim + 4 * outDim


def mul_kernel_bytes(D: int, inDim: int, outDim: int) -> int:
    """What bucket_mul_kernel must move per CALL it serves (a launch serves `group` calls): the SURVEY formula."""
    return algorithmic_bytes(D, inDim, outDim)


def moved_bytes(D: int, inDim: int, outDim: int) -> int:
    """The same with what a persistent FP16 launch actually stages for the keep test: the 2-byte compact row means
    instead of the 8-byte stats entries."""
    return D * (outDim // 16) * 2 + 16 * inDim * 2 + 4096 * 2 + 4 * inDim + 4 * outDim


def q4_bytes(D: int, inDim: int, outDim: int, n_outliers: int, outlier_bytes: int = 16) -> int:
    """bucketMulQ4 per call: kept rows of outDim/32 words + 8-byte stats entries of the 8*inDim rows + probes + v + out
    + the outlier table (16 bytes an entry in the reference; the registered index holds 4)."""
    return D * (outDim // 32) * 2 + 8 * inDim * 8 + 4096 * 2 + 4 * inDim + 4 * outDim + outlier_bytes * n_outliers


def _lds_atomic_peak():
    """Chip-wide ds_add_u32 rate in G atomics/s: the microbenchmark's elements per ns and CU x 256 CUs."""
    for name in ("r05_q4_microbench_scatter.txt", "r04_q4_microbench_scatter.txt"):
        try:
            for line in open(os.path.join(ROOT, "profiles", name)):
                if "ds_add_u32" in line:
                    return float(line.split("elem/ns/CU")[0].split()[-1]) * 256, f"profiles/{name} (tools/lab/microbench.hip on an MI355X)"
        except Exception:                                        # noqa: BLE001
            pass
    return 13 * 2.4 * 256, "13 ds_add_u32 per clock and CU (DESIGN 4.1) x 2.4 GHz x 256 CUs"


LDS_ATOMIC_PEAK, LDS_ATOMIC_SRC = _lds_atomic_peak()


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(result: dict) -> str:
    """The LAST stdout line: the driver's contract keys plus `roofline` and `cpu_baseline`, numbers and short strings only
    (the reference's timeIt prints two numbers, helpers/timeit.swift:33-34).  Everything else goes to gpurun_out/bench_full.json
    and stderr.  Always < COMPACT_LIMIT bytes (tests/test_abi.py::test_bench_compact_line_fits_the_driver_tail)."""
    out = _pick(result, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                         "dtype", "data"))
    cfg = dict(result.get("config", {}))
    if len(str(cfg.get("workload", ""))) > 200:
        cfg["workload"] = str(cfg["workload"])[:197] + "..."
    cfg.pop("kernel_geometry(waves,elems,slices)", None)
    out["config"] = cfg
    out.update(_pick(result, ("us_per_call", "tokens_per_s", "timed_region_ms", "timed_steps", "steps_requested", "bytes_per_launch", "error")))
    rf = result.get("roofline")
    if isinstance(rf, dict):
        r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "traffic_measured_in_run",
                       "bytes_per_launch", "calls_per_launch", "launches_in_flight", "kernel_us", "frac_moved_bytes", "scope"))
        if isinstance(rf.get("single_stream"), dict):
            r["single_stream"] = _pick(rf["single_stream"], ("frac", "kernel_us", "kernel_us_hip_events_outside_graph"))
        out["roofline"] = r
    cb = result.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "us_per_call", "error", "gpu_vs_cpu_max_rel_err", "gpu_vs_cpu_outputs_checked",
                       "gpu_vs_cpu_dispatch_counts_checked", "gpu_vs_cpu_dispatch_count_or_nan_mismatches"))
        if "sample" in cb:
            c["sample"] = str(cb["sample"])[:160]
        c0 = cb.get("config0_4096x4096_effort_0.5")
        if isinstance(c0, dict):
            c["config0_4096x4096_effort_0.5"] = _pick(c0, ("value", "unit", "cores", "us_per_call", "error"))
        out["cpu_baseline"] = c
    # a few scalars of the sections that went to the full record
    extra = {}
    try:
        by = result["by_group_size"]
        extra["lone_call_us"] = by["1"]["us_per_call"]
        extra["three_per_launch_us_per_call"] = by["3"]["us_per_call"]
        extra["one_group_of_32_us_per_call"] = by["32"]["us_per_call"]
    except Exception:                                        # noqa: BLE001
        pass
    try:
        tp = result["timeit_protocol"]
        tps = {e: x["spd_tps"] for e, x in tp["as_written"].items()}
        for k, name in (("as_written_overlap4", "0.25_overlap4"),):
            if k in tp:
                tps[name] = tp[k]["0.25"]["spd_tps"]
        tps["dense"] = tp["dense_hip_kernel_3x_wq"]["spd_tps"]
        extra["timeit_tps"] = tps
    except Exception:                                        # noqa: BLE001
        pass
    try:
        extra["dense_rocblas_speedup"] = result["dense_rocblas"]["speedup_at_effort"]
        extra["dense_hip_kernel_speedup"] = result["dense_hip_kernel"]["speedup_at_effort"]
    except Exception:                                        # noqa: BLE001
        pass
    try:
        d = result["decode"]
        extra["decode_tps"] = {"dense_rocblas": d["dense_rocblas_tokens_per_s"], "dense_hip_kernel": d["dense_hip_kernel_tokens_per_s"],
                               **{e: x["tokens_per_s"] for e, x in d["effort"].items()}}
    except Exception:                                        # noqa: BLE001
        pass
    try:
        ll = result["layer_latency"]
        extra["layer_latency_us_effort_0.5"] = _pick(ll, ("us_per_layer_unsharded", "us_per_layer_with_gathers", "projected_kernel_only", "error"))
    except Exception:                                        # noqa: BLE001
        pass
    try:
        q = result["other_configs"]["4096x11008 q4 (bucketMulQ4, 2 % outliers)"]
        extra["q4_us_per_call"] = {k.replace("effort 0.25, ", ""): x["us_per_call"] for k, x in q.items()}
        extra["q4_frac_of_hbm_peak_moved_bytes"] = {k.replace("effort 0.25, ", ""): x["frac_of_hbm_peak_4B_outliers"] for k, x in q.items()}
    except Exception:                                        # noqa: BLE001
        pass
    mg = result.get("multi_gpu")
    if isinstance(mg, dict):
        m = _pick(mg, ("partition", "ms_per_step_kernel_only", "ms_per_step_with_all_gather", "steps_per_round", "aborted"))
        for part in ("columns", "matrices_4096x4096", "columns_4096x4096"):
            if isinstance(mg.get(part), dict):
                m[part] = _pick(mg[part], ("ms_per_step_kernel_only", "ms_per_step_with_all_gather", "whole_job_effective_GBps", "per_rank_frac_of_hbm_peak", "error"))
        if isinstance(mg.get("layer_latency"), dict):
            m["layer_latency"] = _pick(mg["layer_latency"], ("effort", "us_per_layer_kernel_only", "us_per_layer_with_gathers", "us_per_layer_unsharded", "error"))
        out["multi_gpu"] = m
    out.update(_pick(result, ("rccl_ranks", "launched_by")))
    if extra:
        out["extra"] = extra
    out["full_record"] = result.get("full_record", "gpurun_out/bench_full.json")
    line = json.dumps(out, separators=(",", ":"))
    for victim in ("extra", "multi_gpu"):                    # never exceed the limit: shed the optional sections first
        if len(line) >= COMPACT_LIMIT and victim in out:
            del out[victim]
            line = json.dumps(out, separators=(",", ":"))
    assert len(line) < COMPACT_LIMIT, len(line)
    return line


def make_weights(ea, n, inDim, outDim, seed0, dev, keep_core=True, q4=False):
    import torch
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
            ew = ea.ExpertWeights.from_core(W, aligned=ALIGN_ROWS)      # product converter (GPU bucketize)
        if not keep_core:
            ew.core = None
        ew.handle
        ews.append(ew)
    ea.gpu().eval()
    return ews


def chunked(items, n):
    return [items[i:i + n] for i in range(0, len(items), n)]


_EVENTS_FOR_LIFE = []       # see Job: events that took part in a capture are never destroyed
_GRAPHS_FOR_LIFE = []       # ... and neither is a captured graph: keep() below


def keep(graph):
    """Every hipGraph this process captures lives until the process exits (and the process leaves through os._exit: main()).  Destroying
    a graph exec shortly after its last launch is what took round 5's bench down once in ~40 runs (`free(): invalid pointer`): a
    use-after-free inside the HIP runtime's graph code that plain torch ops reproduce without this library in the process
    (tools/lab/graph_event_repro.py: capture -> replay -> destroy in a loop dies under the checking allocator, the same loop that KEEPS
    its graphs does not).  A bench's few hundred graphs cost a few MB."""
    _GRAPHS_FOR_LIFE.append(graph)
    return graph


class Job:
    """K independent steps as ONE hipGraph: step i is enqueued on stream i % S through context i % S (a context owns the
    scratch of its launches) into output set i % S; the streams fork from / join the capturing stream inside the graph.
    `step(ctx, slot)` enqueues one step's launches on the current stream.  S = 1: everything on the capturing stream.
    (The N > 1 legs, the dense baseline and tools/bench_extra.py's `four_contexts`; the one-GPU headline goes through LaneJob.)

    The fork / join edges go through events that live AS LONG AS THE PROCESS, not through Stream.wait_stream: wait_stream creates a
    temporary event per call and destroys it at once -- inside a capture, while the graph it became part of is still being built --
    and on this runtime (ROCm 7.x) an event destroyed after it took part in a capture leaves a dangling reference that a LATER
    hipGraphLaunch trips over, or that surfaces as glibc's `free(): invalid pointer` a few frees later: round 5's one-in-forty bench
    abort (round 6's hunts: tools/lab/graph_event_repro.py; the library's own lanes have pooled their events for the same
    reason since round 3, api.hip)."""

    def __init__(self, ea, device, streams=1, tune=(0, 0, 0)):
        import torch
        self.ea, self.S = ea, max(1, streams)
        self.ctxs = [ea.gpu(device)] + [ea.Gpu(device) for _ in range(self.S - 1)]
        self.streams = [None] + [torch.cuda.Stream(device=device) for _ in range(self.S - 1)]
        self.fork_ev = torch.cuda.Event()
        self.join_ev = [torch.cuda.Event() for _ in range(self.S)]
        _EVENTS_FOR_LIFE.extend([self.fork_ev] + self.join_ev + self.streams[1:])
        for c in self.ctxs:
            c.set_tuning(*tune)

    def _enqueue(self, step, nsteps):
        import torch
        s0 = torch.cuda.current_stream()
        used = min(self.S, nsteps)
        if used > 1:
            self.fork_ev.record(s0)
        for k in range(1, used):
            self.streams[k].wait_event(self.fork_ev)
        for i in range(nsteps):
            k = i % self.S
            if k == 0:
                step(self.ctxs[0], 0)
            else:
                with torch.cuda.stream(self.streams[k]):
                    step(self.ctxs[k], k)
        for k in range(1, used):
            self.join_ev[k].record(self.streams[k])
            s0.wait_event(self.join_ev[k])

    def capture(self, step, nsteps):
        import torch
        self._enqueue(step, min(nsteps, self.S))     # warm: handles, kernel attributes, rocBLAS workspaces
        torch.cuda.synchronize()
        if hasattr(step, "reset"):
            step.reset()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):   # (another thread -- RCCL's watchdog -- may query events meanwhile)
            self._enqueue(step, nsteps)
        for c in self.ctxs:
            c._bind_stream()
        return keep(g)

    def last_dispatch_count(self, nsteps, idx):
        return self.ctxs[(nsteps - 1) % self.S].last_dispatch_count(idx)


class LaneJob:
    """K independent steps as ONE hipGraph through ONE context: the library keeps up to `lanes` launches in flight
    (effort_set_overlap) and orders each launch after the earlier ones it depends on; step i writes output set i % lanes."""

    def __init__(self, ea, device, lanes=1, tune=(0, 0, 0), ctx=None):
        self.S = max(1, lanes)
        self.ctx = ctx if ctx is not None else ea.Gpu(device)
        self.ctx.set_tuning(*tune)
        self.ctx.set_overlap(self.S)
        self.ctxs = [self.ctx]

    def _enqueue(self, step, nsteps):
        for i in range(nsteps):
            step(self.ctx, i % self.S)
        self.ctx.join()                              # the lanes rejoin the (capturing) stream

    def capture(self, step, nsteps):
        import torch
        self._enqueue(step, min(nsteps, self.S))     # warm: handles, kernel attributes
        torch.cuda.synchronize()
        if hasattr(step, "reset"):
            step.reset()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._enqueue(step, nsteps)
        self.ctx._bind_stream()
        return keep(g)

    def last_dispatch_count(self, nsteps, idx):
        return self.ctx.last_dispatch_count(idx)


def time_graph(g_timed, g_warm, barrier=None, reps=1):
    """W warm-up steps (one replay of the warm-up graph), then the timed graph `reps` times: seconds per replay."""
    import torch
    (g_warm if g_warm is not None else g_timed).replay()      # (no warm-up graph: one untimed replay of the timed one -- the first replay of a graph uploads it)
    if barrier:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g_timed.replay()
    torch.cuda.synchronize()
    if barrier:
        barrier()
    return (time.perf_counter() - t0) / reps


def cpu_baseline(ews, v, effort, inDim, outDim, budget_s=10.0, nmat=4):
    """CPU oracle ("port") on the host cores: same converted weights (a few of the matrices), same v, same effort.  Timed in a
    process of its own (oracle/cpu_bench.py): one thread per physical core, bound, spinning between the port's parallel
    regions -- this process keeps its own OpenMP workers asleep for the sake of the GPU timings."""
    import shutil
    import subprocess
    import tempfile

    import numpy as np
    d = tempfile.mkdtemp(prefix="effort_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for k, ew in enumerate(ews[:nmat]):
            np.save(os.path.join(d, f"b{k}.npy"), ew.buckets[0].contiguous().cpu().numpy().view(np.float16))
            np.save(os.path.join(d, f"s{k}.npy"), ew.stats[0].cpu().numpy().view(np.float16))
            np.save(os.path.join(d, f"p{k}.npy"), ew.probes[0].cpu().numpy().view(np.float16))
        np.save(os.path.join(d, "v.npy"), v.cpu().numpy())
        best = None
        logical = os.cpu_count() or 1
        quota = logical                                   # the container's CPU quota (cgroup v2 cpu.max = "<quota> <period>"): threads beyond it are throttled
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                quota = max(1, min(logical, int(q) // int(per)))
        except Exception:
            pass
        for threads in sorted({quota, max(1, quota // 2), min(logical, 2 * quota)}, reverse=True):
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_WAIT_POLICY="active", OMP_PROC_BIND="close", OMP_PLACES="cores", GOMP_SPINCOUNT="100000000")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), d, str(inDim), str(outDim), str(effort),
                                  str(budget_s / 3), str(nmat)], env=env, capture_output=True, text=True, timeout=120 + 4 * budget_s)
            r = json.loads(out.stdout.strip().split("\n")[-1])
            r["threads"] = threads
            if best is None or r["seconds_per_call"] < best["seconds_per_call"]:
                best = r
        dt, n = best["seconds_per_call"], best["calls"]
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return {"value": round(2 * inDim * outDim / dt / 1e9, 3), "unit": "GB/s", "cores": best["threads"], "host_logical_cpus": logical, "container_cpu_quota": quota,
            "kind": "port", "us_per_call": round(dt * 1e6, 1),
            "sample": f"{n} bucketMul calls, effort {effort}, {nmat} converted {inDim}x{outDim} matrices rotated, {budget_s / 3:.0f} s per thread count",
            "sample_note": f"{n} bucketMul calls at effort {effort} over {nmat} of the converted {inDim}x{outDim} matrices (the CPU port: bucket rows "
                      f"streamed by (group, column block) tasks, OpenMP, {best['threads']} bound spinning threads: the fastest of "
                      f"{quota} / {max(1, quota // 2)} / {min(logical, 2 * quota)}; the container may use {quota} of the host's {logical} logical CPUs), "
                      f"{budget_s / 3:.0f} s each"}


def measured_traffic(effort, group, budget_s=150):
    """HBM-side bytes per launch of the timed configuration, MEASURED in this run: two rocprofv3 --pmc passes (FETCH_SIZE, then
    WRITE_SIZE; counters in their own runs with --kernel-trace only, as MI355X_MICROARCH.md prescribes) of this script's
    --headline-only job in a child process, folded by tools/pmc_traffic.py.  None if rocprofv3 is missing or a pass fails."""
    import shutil
    import signal
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 not on PATH"
    d = tempfile.mkdtemp(prefix="effort_pmc_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = ["rocprofv3", "--pmc", name, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(d, name), "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "24", "--warmup", "4", "--headline-only", "--effort", str(effort), "--group", str(group)]
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, start_new_session=True)
            try:
                _, err = p.communicate(timeout=budget_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                return None, f"rocprofv3 --pmc {name}: no result within {budget_s} s"
            if p.returncode != 0:
                return None, f"rocprofv3 --pmc {name} exited {p.returncode}: {(err or '')[-200:]}"
        out = os.path.join(d, "traffic.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), "--fetch", os.path.join(d, "FETCH_SIZE"), "--write", os.path.join(d, "WRITE_SIZE"),
                            "--group", str(group), "--effort", str(effort), "--out", out], capture_output=True, text=True, timeout=60)
        if r.returncode != 0:
            return None, "tools/pmc_traffic.py: " + (r.stderr or r.stdout)[-200:]
        with open(out) as f:
            return json.load(f), None
    except Exception as ex:                                  # noqa: BLE001
        return None, repr(ex)
    finally:
        shutil.rmtree(d, ignore_errors=True)


def timeit_protocol(ea, g, dev, efforts=(1.0, 0.5, 0.25), repeats=3000, variants=(("as_written", 1, 1),), from_graph=False,
                    dense=(("dense_hip_kernel", False),)):
    """The reference's own timing loop (helpers/timeit.swift:10-34 driven by benchmarks/benchmark.swift:245-257,
    goQuickBucketPerformance): 1000 warm-up calls, eval, then `repeats` SINGLE bucketMul calls -- expertMul(v, layers[i % 32].w1,
    out: test, effort) on 32 rotating 4096 -> 14336 matrices, one enqueue per call, one eval at the end; it prints
    tpt = ms_per_call * 4 * 32 and spd = 1000 / tpt (projected tokens/s).  Here every call is one effort_bucketmul through the C
    ABI (ctypes, arguments prebuilt: the host must not be what is timed; its enqueue time is reported beside the total).
    `variants` = (name, lanes, output vectors): the loop as written (every call writes the SAME output vector, so the calls are
    ordered whatever the lanes); tools/bench_extra.py adds the same with effort_set_overlap(4), four rotating output vectors under
    overlap, the loop from a hipGraph, and the dense line through rocBLAS."""
    import ctypes as C

    import torch
    lib = ea.lib()
    inDim, outDim = 4096, 14336
    ws = make_weights(ea, 32, inDim, outDim, 777, dev, keep_core=False)
    hs = [C.c_void_p(ew.handle) if not isinstance(ew.handle, C.c_void_p) else ew.handle for ew in ws]
    gen = torch.Generator(device=dev)
    gen.manual_seed(43)
    v = torch.randn(inDim, generator=gen, device=dev)
    tests = [torch.zeros(outDim, device=dev) for _ in range(4)]
    vp, tp = C.c_void_p(v.data_ptr()), [C.c_void_p(t.data_ptr()) for t in tests]
    fn, ctx = lib.effort_bucketmul, g.ctx
    g._bind_stream()
    res = {"shape": f"{inDim}x{outDim}", "matrices": 32, "warmup_calls": 1000, "timed_calls": repeats,
           "formula": "tpt_ms = ms_per_call * 4 * 32; spd_tps = 1000 / tpt_ms (helpers/timeit.swift:33-34)",
           "reference_readoffs_tps": {"source": "docs/ryc/ryc0.1.png via BASELINE.md section 1 (Apple Silicon, hardware not stated)",
                                      "mps_dense": 26, "1.0": 15, "0.5": 26, "0.25": 47, "0.2": 55, "0.1": 87}}

    def loop(n, s, nout):
        rc = 0
        for i in range(n):
            rc |= fn(ctx, hs[i & 31], vp, None, tp[i % nout], s)
        return rc
    for name, lanes, nout in variants:
        g.set_overlap(lanes)
        sec = {}
        for s in efforts:
            s = float(s)
            assert loop(1000, s, nout) == 0
            g.eval()
            t0 = time.perf_counter()
            assert loop(repeats, s, nout) == 0
            t1 = time.perf_counter()
            g.eval()
            t2 = time.perf_counter()
            epl = (t2 - t0) / repeats * 1e3                                   # ms per call
            sec[str(s)] = {"us_per_call": round(epl * 1e3, 3), "tpt_ms": round(epl * 4 * 32, 3), "spd_tps": round(1000.0 / (epl * 4 * 32), 1),
                           "host_enqueue_us_per_call": round((t1 - t0) / repeats * 1e6, 3)}
        res[name] = sec
    g.set_overlap(1)
    if from_graph:      # the same loop with the host taken out: 320 calls (ten rotations) captured into ONE hipGraph, replayed ten times
        sec = {}
        for s in efforts:
            s = float(s)
            assert loop(64, s, 1) == 0
            g.eval()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                g._bind_stream()                                                  # (the raw ABI calls go to the context's stream: the capturing one)
                assert loop(320, s, 1) == 0
            g._bind_stream()
            keep(gr).replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                gr.replay()
            torch.cuda.synchronize()
            epl = (time.perf_counter() - t0) / 3200 * 1e3
            sec[str(s)] = {"us_per_call": round(epl * 1e3, 3), "tpt_ms": round(epl * 4 * 32, 3), "spd_tps": round(1000.0 / (epl * 4 * 32), 1)}
            del gr
        res["as_written_from_a_graph"] = sec
    # the MPS line of the same benchmark: three dense 4096 x 4096 multiplies per iteration, multiplier 4/3 (benchmark.swift:237-241)
    cores = [(torch.randn((4096, 4096), generator=gen, device=dev) * 0.02).to(torch.float16) for _ in range(32)]
    dfn = lib.effort_dense_gemv
    cps = [C.c_void_p(c.data_ptr()) for c in cores]
    ctl = torch.zeros(4096, device=dev)
    cp = C.c_void_p(ctl.data_ptr())
    for backend, rocblas in dense:
        g.set_dense_backend(rocblas)

        def dloop(n):
            rc = 0
            for i in range(n):
                for _ in range(3):
                    rc |= dfn(ctx, cps[i & 31], vp, cp, 4096, 4096)
            return rc
        assert dloop(1000) == 0
        g.eval()
        t0 = time.perf_counter()
        assert dloop(repeats) == 0
        g.eval()
        epl = (time.perf_counter() - t0) / repeats * 1e3
        res[backend + "_3x_wq"] = {"us_per_iteration": round(epl * 1e3, 3), "tpt_ms": round(epl * 4 * 32 * 4 / 3, 3), "spd_tps": round(1000.0 / (epl * 4 * 32 * 4 / 3), 1)}
    g.set_dense_backend(False)
    return res


def layer_latency(ea, g, dev, rank=0, world=1, effort=0.5, n_layers=8, reps=60, project=(2, 4, 8)):
    """BASELINE config 4's LATENCY case ("all 32 layers' Wq/Wk/Wv/Wo + FFN matrices sharded across 8 x MI355X, RCCL all-gather over
    xGMI, effort 50 %"): one Mistral-7B layer's seven matrices column-sharded, in the decode loop's dependent order with the glue
    folded in -- wo -> w1|w3 -> w2 -> wq|wk|wv of the next layer (runNetwork.swift:121-183) -- ONE grouped launch of this rank's
    shards and ONE effort_allgather_outputs per group (in place on h for wo / w2): effort_amd.sharded.ColumnShardedGroups, what
    Decoder(world=G) runs.  `n_layers` weight sets are rotated (3.4 GB: no set finds its rows in the Infinity Cache); the chain
    is one hipGraph, us per LAYER = replay time / n_layers.  Reported per rank: kernel-only (no collectives) and with the
    gathers; plus, on this GPU alone, rank 0's kernel-only chain of a world of 2 / 4 / 8 (`projected_kernel_only`: what a rank of
    such a world launches, without its collectives)."""
    import torch

    from effort_amd.decode import MistralConfig, Model
    from effort_amd.sharded import ColumnShardedGroups
    model = Model.random(MistralConfig(numLayers=n_layers), seed=3, keep_cores=False)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    f = lambda n: torch.randn(n, generator=gen, device=dev)                      # noqa: E731
    B = {"h": f(4096), "attn": f(4096), "x1": f(14336), "x3": f(14336), "xq": f(4096), "xk": f(1024), "xv": f(1024)}
    h0 = B["h"].clone()

    def chain(G):
        for n in range(n_layers):
            L, Ln = model.layers[n], model.layers[(n + 1) % n_layers]
            an, fnw = {"norm": Ln.attnNorm}, {"norm": L.ffnNorm}
            G.mul(B["attn"], [(L.wo, B["h"], {"resid": B["h"]})], effort)
            G.mul(B["h"], [(L.w1, B["x1"], fnw), (L.w3, B["x3"], fnw)], effort)
            G.mul(B["x1"], [(L.w2, B["h"], {"gate": B["x3"], "resid": B["h"]})], effort)
            G.mul(B["h"], [(Ln.wq, B["xq"], an), (Ln.wk, B["xk"], an), (Ln.wv, B["xv"], an)], effort)
        B["h"].copy_(h0)                                                          # (the state stays bounded from replay to replay)

    def timed(G, graph=True):
        chain(G)                                                                  # warm: shards registered, kernel attributes
        torch.cuda.synchronize()
        if graph:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                chain(G)
            g._bind_stream()
            run = keep(gr).replay
        else:
            run = lambda: chain(G)                                                # noqa: E731
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / reps / n_layers * 1e6, 2)

    res = {"effort": effort, "layers_rotated": n_layers, "world": world, "rank": rank,
           "order": "wo -> w1|w3 -> w2 -> wq|wk|wv (next layer), glue folded in, one gather per group (in place on h for wo / w2)",
           "us_per_layer_unsharded": timed(ColumnShardedGroups(1, 0, emulate=True, gpu=g))}
    res["us_per_layer_kernel_only"] = timed(ColumnShardedGroups(world, rank, gpu=g, gather=False)) if world > 1 else res["us_per_layer_unsharded"]
    if g.has_comm and g.comm_world == world:
        try:
            if world > 1:     # collectives of a REAL world are enqueued eagerly: a capture that goes wrong on one rank would hang the others, and no bench line at all is worse than a host-inclusive figure
                res["us_per_layer_with_gathers"] = timed(ColumnShardedGroups(world, rank, gpu=g), graph=False)
                res["with_gathers_from"] = "eager enqueues (host in the loop: 8 launches + 4 collectives + 4 scatter copies per layer)"
            else:
                res["us_per_layer_with_gathers"] = timed(ColumnShardedGroups(world, rank, gpu=g))
                res["with_gathers_from"] = "one hipGraph (collectives captured)"
        except Exception as ex:                                                   # noqa: BLE001  (a runtime that cannot capture the collective)
            res["with_gathers_graph_error"] = repr(ex)[:200]
            try:
                torch.cuda.synchronize()
                res["us_per_layer_with_gathers"] = timed(ColumnShardedGroups(world, rank, gpu=g), graph=False)
                res["with_gathers_from"] = "eager enqueues (host in the loop)"
            except Exception as ex2:                                              # noqa: BLE001
                res["error"] = repr(ex2)[:200]
    if world == 1:
        proj = {}
        for Gw in project:
            try:
                proj[str(Gw)] = timed(ColumnShardedGroups(Gw, 0, gpu=g, gather=False))
            except Exception as ex:                                               # noqa: BLE001
                proj[str(Gw)] = repr(ex)[:120]
        res["projected_kernel_only"] = proj
        res["projected_kernel_only_note"] = "rank 0's launches of a world of G on THIS GPU, no collectives: us per layer"
    return res


def oracle_outputs(ews, v, effort, inDim, outDim, idxs):
    """The CPU oracle's product for the matrices `idxs` (test infrastructure used as the CHECKER of the bench's outputs)."""
    import numpy as np

    from oracle import cpu
    vh = v.cpu().numpy()
    sc = cpu.Scratch(inDim * 16)
    res = {}
    for k in idxs:
        ew = ews[k]
        out, D, cutoff = cpu.bucket_mul(vh, ew.buckets[0].cpu().numpy().view(np.float16), ew.stats[0].cpu().numpy().view(np.float16),
                                        ew.probes[0].cpu().numpy().view(np.float16), inDim, outDim, effort, scratch=sc)
        res[k] = (out, D, cutoff)
    return res


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--effort", type=float, default=0.25)
    ap.add_argument("--group", type=int, default=32, help="independent calls per kernel launch (1..32)")
    ap.add_argument("--streams", type=int, default=4, help="steps (launches) in flight: the context's lanes (effort_set_overlap)")
    ap.add_argument("--partition", choices=["matrices", "columns"], default="matrices",
                    help="accepted for compatibility and ignored: N > 1 times both partitions (matrices = `value`, columns beside it in multi_gpu)")
    ap.add_argument("--no-sweep", action="store_true", help="headline + roofline + cpu_baseline only")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes (roofline.traffic then comes from the committed profile)")
    ap.add_argument("--no-decode", action="store_true", help="skip the end-to-end decode section (BASELINE.json configs[4])")
    ap.add_argument("--headline-only", action="store_true", help="only the timed job (for rocprofv3 passes: every bucket_mul_kernel dispatch is then the timed configuration)")
    ap.add_argument("--headline-shared", action="store_true", help="round 2's job: every step in flight on the SAME 32 matrices")
    ap.add_argument("--row-reuse", action="store_true", help="effort_set_row_reuse(1) on the timed job's context: the ordinary cache policy on the row stream (what a --headline-shared caller asks for; default nt)")
    ap.add_argument("--no-align", action="store_true", help="stream the converter's rows as they are (2*cols bytes apart) instead of on whole 128-byte lines")
    ap.add_argument("--tune", default="0,0,0", help="waves,elems,slices of the multiply kernel (0,0,0 = heuristic)")
    return ap.parse_args(argv)


class Bench:
    """The state every section shares: the device, the contexts, S disjoint sets of 32 converted matrices, the input vector, one
    output set per step in flight.  tools/bench_extra.py builds the same object for its sections."""

    def __init__(self, args, rank=0, world=1, local=0, dist=None):
        import torch

        import effort_amd as ea
        global ALIGN_ROWS
        ALIGN_ROWS = not args.no_align
        self.args, self.rank, self.world, self.local, self.dist = args, rank, world, local, dist
        self.torch, self.ea = torch, ea
        self.G = max(1, min(32, args.group))
        self.S = max(1, min(8, args.streams))
        self.tune = tuple(int(x) for x in args.tune.split(","))
        self.dev = torch.device("cuda", local)
        self.g = ea.gpu(local)
        self.g.set_tuning(*self.tune)
        self.inDim, self.outDim = IN_DIM, OUT_DIM
        self.eff_bytes = 2 * IN_DIM * OUT_DIM
        t_setup = time.perf_counter()
        # S disjoint sets of 32 matrices: every step in flight multiplies its own (set 0 keeps the dense cores: dense baseline, cos-sim)
        self.seed0 = 1234 if world == 1 else 1234 + rank * N_MATS * self.S
        self.n_sets = 1 if args.headline_shared else self.S
        lean = args.headline_only or args.no_sweep
        self.ew_sets = [make_weights(ea, N_MATS, IN_DIM, OUT_DIM, self.seed0 + k * N_MATS, self.dev, keep_core=(rank == 0 and (k == 0 or not lean)))
                        for k in range(self.n_sets)]
        self.ews = self.ew_sets[0]
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(42)
        self.v = torch.randn(IN_DIM, generator=self.gen, device=self.dev, dtype=torch.float32)
        self.out_sets = [torch.zeros((N_MATS, OUT_DIM), device=self.dev) for _ in range(self.S)]          # one output set per step in flight
        torch.cuda.synchronize()
        log(f"[rank {rank}] setup {time.perf_counter() - t_setup:.1f} s: {self.n_sets} x {N_MATS} matrices {IN_DIM}x{OUT_DIM} converted on the GPU")
        # the headline job: ONE context, the library overlaps its launches (effort_set_overlap).  N > 1 keeps round 2's job (four
        # contexts on four streams) under the all-gather pipeline.
        self.job = Job(ea, local, self.S, self.tune) if dist else LaneJob(ea, local, self.S, self.tune)
        self.one = LaneJob(ea, local, 1, self.tune, ctx=self.g)
        if getattr(args, "row_reuse", False):
            for cx in self.job.ctxs:
                cx.set_row_reuse(True)
        self.launches_per_step = (N_MATS + self.G - 1) // self.G

    def barrier(self):
        if self.dist:
            self.dist.barrier()

    def mul_step(self, effort, weights=None, vec=None, sets=None, group=None, wsets=None):
        """One step: the group launches of the matrices in `weights` (or, `wsets` given, of the slot's own set: step in flight
        k multiplies wsets[k % len(wsets)]), outputs into set `slot`."""
        ea = self.ea
        weights = self.ews if weights is None else weights
        vec = self.v if vec is None else vec
        sets = self.out_sets if sets is None else sets
        group = self.G if group is None else group

        def step(ctx, slot):
            ws = wsets[slot % len(wsets)] if wsets else weights
            items = [(vec, ew, None, sets[slot][k], effort) for k, ew in enumerate(ws)]
            for ch in chunked(items, group):
                ea.bucketMulGroup(ch, gpu=ctx)
        return step

    def rate(self, tn, kb):
        return {"us_per_call": round(tn * 1e6, 3), "us_per_step": round(tn * 1e6 * N_MATS, 2), "effective_GBps": round(self.eff_bytes / tn / 1e9, 1),
                "achieved_GBps": round(kb / tn / 1e9, 1), "frac_of_hbm_peak": round(kb / tn / 1e9 / HBM_PEAK_GBPS, 4)}

    # ---------------- the timed job on ONE GPU: K steps at the headline effort ------------------------------------------------
    def headline(self):
        args, job, S = self.args, self.job, self.S
        self.head_step = self.mul_step(args.effort, wsets=self.ew_sets)
        g_warm = job.capture(self.head_step, args.warmup) if args.warmup > 0 else None
        self.g_timed = job.capture(self.head_step, args.steps)
        self.D = job.last_dispatch_count(args.steps, (N_MATS - 1) % self.G)
        # the K-step job is repeated until the timed region is >= 50 ms (a 2.5 ms region reads 5 % slow: clocks, caches).  The
        # repetitions are captured into ONE graph, timed with one launch: a graph's lanes drain at its end, so replaying a 20-step
        # graph spends ~7 % of its time filling and draining the four launches in flight -- an artefact of chopping the job into
        # replays, not of the job (the K-step graph replayed `reps` times is reported beside it)
        est = time_graph(self.g_timed, g_warm, self.barrier)
        self.reps = reps = max(1, int(0.06 / max(est, 1e-6)) + 1)
        self.dt_replayed = time_graph(self.g_timed, g_warm, self.barrier, reps=reps) / args.steps
        if reps > 1:
            g_long = job.capture(self.head_step, args.steps * reps)
            time_graph(g_long, None)                                  # (a graph's first replay uploads it)
            self.dt = time_graph(g_long, g_warm, self.barrier) / (args.steps * reps)
            del g_long
        else:
            self.dt = self.dt_replayed
        self.in_flight = min(S, args.steps)
        self.dt_kernel = self.dt

    # ---------------- N > 1: rounds of steps, one all-gather per round under the next round's compute --------------------------
    def make_exchange(self):
        """Every rank: its 32 matrices per step; the steps run a ROUND (16 S steps: one hipGraph with S steps in flight, as on one
        GPU) and ONE all-gather per round exchanges the round's output vectors.  Pipelined: round r's all-gather runs on a
        communication stream while round r+1 computes into the other of two buffers.  The collective is the C ABI's
        (effort_comm_create / effort_allgather_outputs: ncclAllGather on a context's stream); torch.distributed only ships the
        communicator id between the ranks."""
        torch, ea, args, S, G, dev, world, dist, job = self.torch, self.ea, self.args, self.S, self.G, self.dev, self.world, self.dist, self.job
        from effort_amd.sharded import init_comm
        comm = torch.cuda.Stream(device=dev)
        self.cg = cg = init_comm(ea.Gpu(self.local))
        bench = self

        def round_step(weights, send, vec):
            """Step i of a round writes output set i of `send` ([R, 32, localOut]); Job hands out (ctx, slot = i % S)."""
            count = [0]

            def step(ctx, slot):
                i = count[0]
                count[0] += 1
                ws = weights[slot % len(weights)] if isinstance(weights[0], list) else weights
                items = [(vec, ew, None, send[i % send.shape[0]][k], args.effort) for k, ew in enumerate(ws)]
                for ch in chunked(items, G):
                    ea.bucketMulGroup(ch, gpu=ctx)
            step.reset = lambda: count.__setitem__(0, 0)
            return step

        class Exchange:
            def __init__(self, weights, localOut, vec=None):
                self.lo = localOut
                self.vec = bench.v if vec is None else vec
                self.R = 16 * S                          # steps per round: sixteen per stream -- a round is one hipGraph whose launches drain at its end, so long rounds
                self.send = [torch.zeros((self.R, N_MATS, localOut), device=dev) for _ in range(2)]
                self.recv = [torch.zeros(world * self.R * N_MATS * localOut, device=dev) for _ in range(2)]
                self.ev = [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(2)]
                self.weights, self.graphs = weights, {}

            def graph(self, b, n):                       # n steps (<= R) into buffer b: step i -> stream i % S, output set i
                if (b, n) not in self.graphs:
                    self.graphs[(b, n)] = job.capture(round_step(self.weights, self.send[b], self.vec), n)
                return self.graphs[(b, n)]

            def run(self, nsteps, exchange=True):
                main = torch.cuda.current_stream()
                rounds = [self.R] * (nsteps // self.R) + ([nsteps % self.R] if nsteps % self.R else [])
                for r, n in enumerate(rounds):
                    b = r & 1
                    if exchange:
                        main.wait_event(self.ev[b][1])    # the previous gather out of this buffer is done
                    self.graph(b, n).replay()
                    if exchange:
                        self.ev[b][0].record(main)
                        with torch.cuda.stream(comm):
                            comm.wait_event(self.ev[b][0])
                            cnt = n * N_MATS * self.lo
                            cg.allgather_outputs(self.send[b].view(-1)[:cnt], self.recv[b][:world * cnt], cnt)
                            self.ev[b][1].record(comm)
                main.wait_stream(comm)

            def timed(self, exchange, nsteps=None):
                nsteps = nsteps or args.steps
                for b in (0, 1):                         # captures happen outside the timed region
                    for n in {self.R, nsteps % self.R, args.warmup % self.R} - {0}:
                        self.graph(b, n)
                self.run(args.warmup, exchange)
                bench.barrier()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                self.run(nsteps, exchange)
                torch.cuda.synchronize()
                bench.barrier()
                x = torch.tensor([(time.perf_counter() - t0) / nsteps], device=dev, dtype=torch.float64)
                dist.all_reduce(x, op=dist.ReduceOp.MAX)
                return float(x.item())
        self.Exchange = Exchange

    def headline_dist(self):
        args = self.args
        self.make_exchange()
        ex = self.Exchange(self.ew_sets, OUT_DIM)
        ex.timed(False)                                      # (first pass: every graph's first replay uploads it)
        # as on one GPU the K-step job is repeated back to back until the timed region is >= 50 ms (the same count on every rank:
        # from the max-over-ranks estimate); the rounds and their all-gathers simply continue across the repetitions
        self.reps = reps = max(1, int(0.06 / max(ex.timed(True) * args.steps, 1e-6)) + 1)
        ex.timed(False, args.steps * reps)                   # (uploads the remainder round's graph)
        self.dt_kernel = ex.timed(False, args.steps * reps)  # the steps without the exchange (kernel only)
        self.dt = ex.timed(True, args.steps * reps)
        self.D = self.job.ctxs[0].last_dispatch_count((N_MATS - 1) % self.G)
        self.in_flight = min(self.S, args.steps)
        self.dt_replayed = self.dt
        self.g_timed = None
        self.head_step = self.mul_step(args.effort, wsets=self.ew_sets)
        del ex

    def partitions(self, iD, oD, sets, full):
        """Both partitions of one shape, per rank: `sets` = this rank's own S x 32 matrices (matrix partition), `full` = the S x 32
        matrices every rank shards by columns.  Kernel-only and with the round's all-gather; fraction of the HBM roofline per rank."""
        args, world, rank, G = self.args, self.world, self.rank, self.G
        out = {}
        n = args.steps * self.reps
        exm = self.Exchange(sets, oD)
        exm.timed(False)
        exm.timed(False, n)
        km, am = exm.timed(False, n), exm.timed(True, n)
        Dm = self.job.ctxs[0].last_dispatch_count((N_MATS - 1) % G)
        bm = mul_kernel_bytes(Dm, iD, oD)
        out["matrices"] = {"ms_per_step_kernel_only": round(km * 1e3, 5), "ms_per_step_with_all_gather": round(am * 1e3, 5),
                           "per_rank_achieved_GBps": round(N_MATS * bm / km / 1e9, 1), "per_rank_frac_of_hbm_peak": round(N_MATS * bm / km / 1e9 / HBM_PEAK_GBPS, 4),
                           "whole_job_effective_GBps": round(world * N_MATS * 2 * iD * oD / am / 1e9, 1), "scaling": "weak"}
        del exm
        out["columns"] = self.columns(iD, oD, full)
        return out

    def columns(self, iD, oD, full):
        """The bucket-column split: every rank multiplies ITS columns of the same matrices; every step in flight shards
        its own set (with one set, four launches in flight streamed the SAME matrices and a rank's 25 % working set -- 90 MB at
        G = 8 -- sat in the 256 MB Infinity Cache: round 4's `columns` read 0.83-0.87 of "HBM" for that reason)."""
        args, world, rank, G = self.args, self.world, self.rank, self.G
        n = args.steps * self.reps
        shs = []
        for fs in full:
            row = []
            for e in fs:
                sh = e.column_shard(rank, world)
                sh.handle
                row.append(sh)
            shs.append(row)
        exs = self.Exchange(shs, oD // world)
        exs.timed(False)
        exs.timed(False, n)
        kc, ac = exs.timed(False, n), exs.timed(True, n)
        Dc = self.job.ctxs[0].last_dispatch_count((N_MATS - 1) % G)
        bc = Dc * (oD // 16 // world) * 2 + 16 * iD * 8 + 4096 * 2 + 4 * iD + 4 * (oD // world)      # a rank's algorithmic bytes per call: its columns of the kept rows, all the stats
        r = {"partition": f"bucket columns: {oD // 16 // world} of {oD // 16} columns per rank, stats / probes replicated (strong scaling: the same {len(full)} x 32 matrices on every N, every step in flight on its own set)",
             "ms_per_step_kernel_only": round(kc * 1e3, 5), "ms_per_step_with_all_gather": round(ac * 1e3, 5),
             "per_rank_achieved_GBps": round(N_MATS * bc / kc / 1e9, 1), "per_rank_frac_of_hbm_peak": round(N_MATS * bc / kc / 1e9 / HBM_PEAK_GBPS, 4),
             "whole_job_effective_GBps": round(N_MATS * 2 * iD * oD / ac / 1e9, 1), "scaling": "strong", "columns_per_rank": oD // 16 // world,
             "all_gather_bytes_per_rank_per_round": 16 * self.S * N_MATS * (oD // world) * 4}
        del exs, shs
        return r

    def multi_gpu_legs(self, result):
        """N > 1, behind the headline: the column split at the headline shape, both partitions at 4096 x 4096 (north_star's other
        shape), config 4's per-layer latency case.  Each leg is a collective job: a failure is recorded, never fatal."""
        torch, ea, args, S, dev, rank, world = self.torch, self.ea, self.args, self.S, self.dev, self.rank, self.world
        mg = result["multi_gpu"]
        try:
            full = self.ew_sets if self.seed0 == 1234 else [make_weights(ea, N_MATS, IN_DIM, OUT_DIM, 1234 + k * N_MATS, dev, keep_core=False) for k in range(self.n_sets)]
            mg["columns"] = self.columns(IN_DIM, OUT_DIM, full)
            del full
        except Exception as ex:                                  # noqa: BLE001
            mg["columns"] = {"error": repr(ex)[:300]}
        try:
            sq_sets = [make_weights(ea, N_MATS, 4096, 4096, 5000 + rank * N_MATS * S + k * N_MATS, dev, keep_core=False) for k in range(S)]
            sq_full = [make_weights(ea, N_MATS, 4096, 4096, 5000 + k * N_MATS, dev, keep_core=False) for k in range(S)] if rank else sq_sets
            p = self.partitions(4096, 4096, sq_sets, sq_full)
            mg["matrices_4096x4096"], mg["columns_4096x4096"] = p["matrices"], p["columns"]
            del sq_sets, sq_full
        except Exception as ex:                                  # noqa: BLE001
            mg["matrices_4096x4096"] = {"error": repr(ex)[:300]}
        mg["note"] = ("per rank and N: `matrices` = every rank its own 32 matrices per step (weak scaling, the headline), `columns` = every rank its bucket columns "
                      "of the SAME 32 matrices (strong scaling); kernel-only and with the round's RCCL all-gather (effort_allgather_outputs) under the next round's "
                      "compute; fractions against 8 TB/s per GPU")
        try:     # config 4's latency case: a layer's dependent chain, column-sharded, one gather per group, effort 0.5
            torch.cuda.empty_cache()
            mg["layer_latency"] = layer_latency(ea, self.cg, dev, rank, world)
        except Exception as ex:                                  # noqa: BLE001
            mg["layer_latency"] = {"error": repr(ex)[:300]}

    # ---------------- roofline of the dominant kernel, in the timed configuration ----------------------------------------------
    def roofline(self):
        torch, args, G, S, g, one = self.torch, self.args, self.G, self.S, self.g, self.one
        kb = mul_kernel_bytes(self.D, IN_DIM, OUT_DIM)
        traffic, pmc_src = None, None
        for pf in PMC_FILES:
            try:
                with open(pf) as f:
                    pmc = json.load(f)
                if pmc.get("calls_per_launch") == G and abs(pmc.get("effort", -1) - args.effort) < 1e-9:
                    traffic, pmc_src = pmc["hbm_bytes_per_launch"], os.path.relpath(pf, ROOT)
                    break
            except Exception:
                pass
        traffic_src = (f"static: a committed rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE pass of this command on another run ({pmc_src}), "
                       "corrected as MI355X_MICROARCH.md prescribes; NOT measured in this run") if traffic else None
        in_run = False
        if not args.no_pmc:
            log("roofline.traffic: two rocprofv3 --pmc passes of the headline job in a child process ...")
            mt, why = measured_traffic(args.effort, G)
            if mt and mt.get("hbm_bytes_per_launch"):
                static = traffic
                traffic, in_run = mt["hbm_bytes_per_launch"], True
                traffic_src = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (own child processes, --kernel-trace only) of "
                               f"`bench.py --headline-only` on this box, {mt['dispatches_averaged']} dispatches averaged; FETCH_SIZE x2 as MI355X_MICROARCH.md "
                               f"prescribes for gfx950, WRITE_SIZE as reported" + (f"; the committed pass of an earlier run reads {static}" if static else ""))
            else:
                log(f"roofline.traffic: not measured ({why}); falling back to the committed pass")
                if traffic_src:
                    traffic_src += f" (the in-run pass failed: {why})"
        t_launch = self.dt / self.launches_per_step      # the timed region's share per launch
        mb = moved_bytes(self.D, IN_DIM, OUT_DIM)
        rf = {
            "bound": "hbm", "kernel": "bucket_mul_kernel", "achieved": round(G * kb / t_launch / 1e9, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(G * kb / t_launch / 1e9 / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "traffic_source": traffic_src, "traffic_measured_in_run": in_run,
            "traffic_over_algorithmic": round(traffic / (G * kb), 4) if traffic else None,
            "calls_per_launch": G, "bytes_per_launch": G * kb, "bytes_per_call": kb, "launches_in_flight": self.in_flight,
            "bytes_per_call_note": "SURVEY 8d formula (8-byte stats entries); the persistent launch stages 2-byte compact means instead: frac_moved_bytes",
            "frac_moved_bytes": round(G * mb / t_launch / 1e9 / HBM_PEAK_GBPS, 4),
            "kernel_us": round(t_launch * 1e6, 3),
            "kernel_us_source": ("timed region / launches.  With one launch in flight this is the kernel's duration; with "
                                 f"{self.in_flight} in flight it is the chip's time per launch (each launch lasts about {self.in_flight}x as long and "
                                 "rocprofv3 reports that), so `achieved` is the rate of the CHIP over the timed region: profiles/r0N_span.json "
                                 "recomputes it from a kernel trace (span = first start .. last end); see single_stream"),
        }
        # the same job with ONE launch in flight: launch duration = timed region / launches (kernels do not overlap)
        if S > 1:
            one.S = len(self.ew_sets)                        # (slots rotate over the weight / output sets; still one launch in flight)
            g1w = one.capture(self.mul_step(args.effort, wsets=self.ew_sets), min(args.warmup, 10))
            g1 = one.capture(self.mul_step(args.effort, wsets=self.ew_sets), max(args.steps, 2 * len(self.ew_sets)))
            one.S = 1
            dt1 = time_graph(g1, g1w, reps=max(1, self.reps // 2)) / max(args.steps, 2 * len(self.ew_sets))
            del g1w, g1
        else:
            dt1 = self.dt
        g.enable_kernel_timing(1)                        # HIP events on the launch stream, queue pre-filled
        torch.cuda._sleep(20_000_000)                    # keep the GPU busy while the host enqueues
        for r in range(4):
            self.mul_step(args.effort)(g, 0)
        evt = g.kernel_timing()
        g.enable_kernel_timing(0)
        t1 = dt1 / self.launches_per_step
        rf["single_stream"] = {
            "achieved": round(G * kb / t1 / 1e9, 1), "frac": round(G * kb / t1 / 1e9 / HBM_PEAK_GBPS, 4), "kernel_us": round(t1 * 1e6, 3),
            "kernel_us_source": "timed region / launches (one stream, back-to-back kernel nodes of one hipGraph): the launch duration rocprofv3 --kernel-trace --stats reports for this configuration",
            "kernel_us_device_clock_note": "the shipped kernels carry no stamp code since round 6 (libeffort_hip_lab.so does: tools/qbench.py with EFFORT_HIP_LIB=lab)",
            "kernel_us_hip_events_outside_graph": round(evt["mul_us"], 3)}
        return rf

    def by_group_size(self):
        """The step at 1 ... 32 calls per launch, one launch in flight (1 = the dependent-chain latency of a lone call)."""
        kb = mul_kernel_bytes(self.D, IN_DIM, OUT_DIM)
        by = {}
        for n in (1, 2, 3, 4, 8, 16, 32):
            gn = self.one.capture(self.mul_step(self.args.effort, group=n), 8)
            tn = time_graph(gn, None, reps=6) / 8 / N_MATS
            by[str(n)] = {"us_per_call": round(tn * 1e6, 3), "effective_GBps": round(self.eff_bytes / tn / 1e9, 1),
                          "achieved_GBps": round(kb / tn / 1e9, 1), "tokens_per_s": round(1.0 / (tn * 4 * 32), 1)}
            del gn
        return by

    def dense_baseline(self, result, t_call, t_lone):
        """basicMul (helpers/mps.swift:14-47) twice: through rocBLAS' hssgemv -- the library the north star names -- and through the
        package's own streaming kernel (csrc/gemv.hip), the default backend of effort_dense_gemv; every step in flight on its own 32
        cores, like the multiply's job."""
        torch, ea, dev = self.torch, self.ea, self.dev
        self.dense_sets = dense_sets = [torch.zeros((N_MATS, OUT_DIM), device=dev) for _ in range(4)]
        four = Job(ea, self.local, 4, self.tune)
        core_sets = [ws for ws in self.ew_sets if ws[0].core is not None]

        def dense_step(ctx, slot):
            for k, ew in enumerate(core_sets[slot % len(core_sets)]):
                ea.basicMul(self.v, ew.core, dense_sets[slot][k], gpu=ctx)
        for name, rocblas in (("dense_rocblas", True), ("dense_hip_kernel", False)):
            for ctx in self.one.ctxs + four.ctxs:
                ctx.set_dense_backend(rocblas)
            td1 = time_graph(self.one.capture(dense_step, 4), None, reps=3) / 4 / N_MATS
            tdk = time_graph(four.capture(dense_step, 8), None, reps=3) / 8 / N_MATS
            result[name] = {"us_per_call_serial": round(td1 * 1e6, 3), "us_per_call_4_streams": round(tdk * 1e6, 3),
                            "GBps": round(self.eff_bytes / min(td1, tdk) / 1e9, 1), "frac_of_hbm_peak": round(self.eff_bytes / min(td1, tdk) / 1e9 / HBM_PEAK_GBPS, 4),
                            "speedup_at_effort": round(min(td1, tdk) / t_call, 3), "speedup_serial_vs_serial": round(td1 / t_lone, 3),
                            "distinct_cores_in_flight": len(core_sets) * N_MATS}
        del four

    def effort_sweep(self):
        """BASELINE config 2's sweep, the headline job at every effort (24-step graphs, four in flight on disjoint sets); the output
        of each point is kept for the oracle check after the timed sections."""
        import numpy as np
        torch, ea, job, S, G = self.torch, self.ea, self.job, self.S, self.G
        last = N_MATS - 1
        sweep, got_all = [], []
        slotL = (24 - 1) % S                             # the slot (output set, weight set) of a 24-step graph's last step
        self.ewsL = ewsL = self.ew_sets[slotL % len(self.ew_sets)]
        dense_out = self.dense_sets[0]
        for e in SWEEP:
            ge = job.capture(self.mul_step(e, wsets=self.ew_sets), 24)
            De = job.last_dispatch_count(24, (N_MATS - 1) % G)
            te = time_graph(ge, None, reps=2) / 24 / N_MATS
            row = {"effort": e, "dispatch_rows": De, "us_per_call": round(te * 1e6, 3),
                   "effective_GBps": round(self.eff_bytes / te / 1e9, 1),
                   "achieved_GBps": round(algorithmic_bytes(De, IN_DIM, OUT_DIM) / te / 1e9, 1),
                   "frac_of_hbm_peak": round(algorithmic_bytes(De, IN_DIM, OUT_DIM) / te / 1e9 / HBM_PEAK_GBPS, 4),
                   "tokens_per_s": round(1.0 / (te * 4 * 32), 1)}
            if ewsL[last].core is not None:
                ea.basicMul(self.v, ewsL[last].core, dense_out[0])
                row["cos_vs_dense"] = round(ea.cosineSimilarityTo(self.out_sets[slotL][last], dense_out[0]), 5)
            got_all.append(self.out_sets[slotL][last].cpu().numpy().astype(np.float64))
            sweep.append(row)
            del ge
        self.sweep_check = (sweep, got_all, last)        # against the oracle after the timed sections (its threads would disturb them)
        return sweep

    def make_sets(self, n, inDim_x, outDim_x, seed, q4=False):
        """S disjoint sets of n matrices: every step in flight multiplies its own."""
        return [make_weights(self.ea, n, inDim_x, outDim_x, seed + 100 * k, self.dev, keep_core=False, q4=q4) for k in range(self.S)]

    def quick(self, sets_w, outDim_x, inDim_x, effort, n, streams, q4=False):
        """n calls per launch over the matrices of `sets_w`, one launch in flight (streams == 1) or S."""
        torch, one, job, S = self.torch, self.one, self.job, self.S
        vx = self.v if inDim_x == IN_DIM else torch.randn(inDim_x, generator=self.gen, device=self.dev, dtype=torch.float32)
        nm = len(sets_w[0])
        sets_x = [torch.zeros((nm, outDim_x), device=self.dev) for _ in range(max(streams, len(sets_w)))]
        jb = one if streams == 1 else job
        nst = 8 if streams == 1 else 16
        if streams == 1:
            one.S = len(sets_w)                  # (one launch in flight, rotating through the sets)
        try:
            gx = jb.capture(self.mul_step(effort, vec=vx, sets=sets_x, group=n, wsets=sets_w), nst)
        finally:
            one.S = 1
        Dx = jb.last_dispatch_count(nst, (nm - 1) % n)
        tx = time_graph(gx, None, reps=3) / nst / nm
        del gx
        if q4:
            nol = sets_w[0][0].outliers.shape[0]
            ab = q4_bytes(Dx, inDim_x, outDim_x, nol)
        else:
            ab = algorithmic_bytes(Dx, inDim_x, outDim_x)
        r = {"us_per_call": round(tx * 1e6, 3), "dispatch_rows": Dx, "effective_GBps": round(2 * inDim_x * outDim_x / tx / 1e9, 1),
             "achieved_GBps": round(ab / tx / 1e9, 1), "frac_of_hbm_peak": round(ab / tx / 1e9 / HBM_PEAK_GBPS, 4)}
        if q4:       # SURVEY 8d prices an outlier at the reference's 16 bytes; the registered index holds 4: the bytes actually moved
            mb = q4_bytes(Dx, inDim_x, outDim_x, nol, 4)
            r["achieved_GBps_4B_outliers"] = round(mb / tx / 1e9, 1)
            r["frac_of_hbm_peak_4B_outliers"] = round(mb / tx / 1e9 / HBM_PEAK_GBPS, 4)
            # what bounds the Q4 multiply is its LDS scatter, not memory: four integer LDS atomics per 16-bit word of a kept row
            # (one per nibble), against the measured ds_add_u32 rate of the chip (tools/lab/microbench.hip)
            atomics = Dx * (outDim_x // 32) * 4          # (the outlier phase has none since round 5: a lane sums its output in a register)
            r["roofline_lds_atomic"] = {"bound": "lds_atomic", "atomics_per_call": atomics, "achieved_Gatomics_per_s": round(atomics / tx / 1e9, 1),
                                        "peak_Gatomics_per_s": round(LDS_ATOMIC_PEAK, 1), "frac": round(atomics / tx / 1e9 / LDS_ATOMIC_PEAK, 4),
                                        "peak_source": LDS_ATOMIC_SRC}
        return r

    def three(self, sets_w, outDim_x, inDim_x, effort, q4=False):
        S = self.S
        return {f"effort {effort}, 1 per launch": self.quick(sets_w, outDim_x, inDim_x, effort, 1, 1, q4),
                f"effort {effort}, 16 per launch": self.quick(sets_w, outDim_x, inDim_x, effort, 16, 1, q4),
                f"effort {effort}, 16 per launch, {S} in flight": self.quick(sets_w, outDim_x, inDim_x, effort, 16, S, q4)}

    def other_configs(self):
        """The other shapes / formats BASELINE.json names: config 0/1's 4096 x 4096 (FP16, 50 % and 25 %) and config 3 (bucketMulQ4
        4096 x 11008 at 25 % with the converter's 2 % outlier tables)."""
        other = {}
        sets_w = self.make_sets(16, 4096, 4096, 4321)
        other["4096x4096 fp16"] = {}
        for e in (0.5, 0.25):
            other["4096x4096 fp16"].update(self.three(sets_w, 4096, 4096, e))
        del sets_w
        q4w = self.make_sets(16, IN_DIM, OUT_DIM, 6321, q4=True)
        other["4096x11008 q4 (bucketMulQ4, 2 % outliers)"] = self.three(q4w, OUT_DIM, IN_DIM, 0.25, q4=True)
        del q4w
        return other

    def decode(self):
        """BASELINE config 5: end-to-end greedy decode (random-init Mistral-7B shapes; one hipGraph per token): tokens/s dense through
        rocBLAS / through the package's own GEMV / effort 100 % / 25 %, KL vs dense.  (Quality on structured weights:
        tools/bench_extra.py.)"""
        from effort_amd.decode import Decoder, MistralConfig, Model, kl_divergence
        g = self.g
        g.set_tuning(0, 0, 0)
        model = Model.random(MistralConfig(), seed=1)
        dec = Decoder(model, maxTokens=64)
        prompt, ntok = [1, 733, 16289, 28793, 22557], 48
        g.set_dense_backend(True)                    # the dense path through rocBLAS (the north star's baseline) ...
        _, dt_r, _ = dec.run(prompt, ntok, dense=True)
        g.set_dense_backend(False)                   # ... and through the package's own GEMV (also the LM head of the effort runs)
        _GRAPHS_FOR_LIFE.extend(dec._graphs.values())        # (keep(): no graph is destroyed before the process exits)
        dec._graphs.clear()
        ids_d, dt_d, lg_d = dec.run(prompt, ntok, dense=True, collect_logits=True)
        forced = prompt + ids_d[len(prompt) - 1:-1]
        dsec = {"model": "Mistral-7B shapes, 32 layers, random-init weights (no checkpoints offline)", "tokens": ntok,
                "dense_rocblas_tokens_per_s": round(1 / dt_r, 1), "dense_hip_kernel_tokens_per_s": round(1 / dt_d, 1), "effort": {}}
        for e in (1.0, 0.5, 0.25):
            _, dt_e, _ = dec.run(prompt, ntok, effort=e)
            _, _, lg_e = dec.run(forced, ntok, effort=e, forced=True, collect_logits=True)
            dsec["effort"][str(e)] = {"tokens_per_s": round(1 / dt_e, 1), "ms_per_token": round(dt_e * 1e3, 3),
                                      "speedup_vs_dense_rocblas": round(dt_r / dt_e, 3), "speedup_vs_dense_hip_kernel": round(dt_d / dt_e, 3),
                                      "kl_vs_dense": round(kl_divergence(lg_d, lg_e), 5)}
        _GRAPHS_FOR_LIFE.extend(dec._graphs.values())
        del dec, model
        self.g.set_tuning(*self.tune)
        return dsec

    def cpu_check(self):
        """cpu_baseline + EVERY output set the timed graph wrote (its last replay: step k in flight multiplied ew_sets[k] into set k)
        against the oracle, 128 dispatch counts, every sweep point."""
        import numpy as np
        torch, ea, args, S, G, job = self.torch, self.ea, self.args, self.S, self.G, self.job
        cb = cpu_baseline(self.ews, self.v, args.effort, IN_DIM, OUT_DIM)
        for o in self.out_sets:
            o.fill_(float("nan"))
        # the timed graph itself, once more, into cleared output sets (BENCH_FORCE_DIST on one GPU timed rounds: S steps of the same job)
        (self.g_timed or job.capture(self.mul_step(args.effort, wsets=self.ew_sets), min(S, args.steps))).replay()
        torch.cuda.synchronize()
        timed_outputs = [o.clone() for o in self.out_sets[:min(S, args.steps)]]
        worst, bad, checked, counts_checked, refs_by_set = 0.0, 0, 0, 0, {}
        for k, hip_set in enumerate(timed_outputs):
            wk = self.ew_sets[k % len(self.ew_sets)]
            hip = hip_set.cpu().numpy()
            ref = oracle_outputs(wk, self.v, args.effort, IN_DIM, OUT_DIM, range(N_MATS))
            for i in range(N_MATS):
                want, Do, _ = ref[i]
                worst = max(worst, float(np.abs(hip[i] - want).max() / (np.abs(want).max() + 1e-30)))
                bad += int(not np.isfinite(hip[i]).all())
                checked += 1
            refs_by_set[k] = ref
        # dispatch counts (the reference's dispatch.size hook): one more step per lane, eagerly, then every lane's hooks
        if G == N_MATS and isinstance(job, LaneJob):
            job._enqueue(self.head_step, len(timed_outputs))
            torch.cuda.synchronize()
            for k in range(len(timed_outputs)):
                job.ctx.hook_lane(k)
                for i in range(N_MATS):
                    bad += int(job.ctx.last_dispatch_count(i) != refs_by_set[k][i][1])
                    counts_checked += 1
        if getattr(self, "sweep_check", None):           # every sweep point's output against the oracle at that effort
            sw, gots, last = self.sweep_check
            for row, got in zip(sw, gots):
                want, Do, _ = oracle_outputs(self.ewsL, self.v, row["effort"], IN_DIM, OUT_DIM, [last])[last]
                row["cos_vs_oracle"] = round(float(got @ want.astype(np.float64) / (np.linalg.norm(got) * np.linalg.norm(want) + 1e-300)), 9)
                row["dispatch_rows_oracle"] = int(Do)
        try:             # BASELINE.json configs[0]: one 4096x4096 bucketMul at 50 % effort on the CPU path
            sq4 = make_weights(ea, 4, 4096, 4096, 4321, self.dev, keep_core=False)
            cb["config0_4096x4096_effort_0.5"] = cpu_baseline(sq4, self.v, 0.5, 4096, 4096, budget_s=5.0)
            del sq4
        except Exception as ex:                              # noqa: BLE001
            cb["config0_4096x4096_effort_0.5"] = {"error": repr(ex)}
        cb["gpu_vs_cpu_max_rel_err"] = worst
        cb["gpu_vs_cpu_outputs_checked"] = checked
        cb["gpu_vs_cpu_outputs_checked_note"] = f"all {len(timed_outputs)} output sets of the timed graph's last replay, each against the oracle on the matrices its step multiplied"
        cb["gpu_vs_cpu_dispatch_counts_checked"] = counts_checked
        cb["gpu_vs_cpu_dispatch_count_or_nan_mismatches"] = bad
        return cb


def dist_env():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_dist(backend="nccl", device=None):
    """One process per GPU: the process group of the launcher's (or launch_ranks') environment.  127.0.0.1: the container's hostname
    may not resolve."""
    import torch.distributed as dist
    world, rank, _ = dist_env()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")        # (BENCH_FORCE_DIST on a bare box: a launcher sets it otherwise)
    if rank != 0:
        os.dup2(2, 1)        # only rank 0 owns stdout (RCCL prints banners there); the others' goes to stderr
    kw = {"device_id": device} if device is not None else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def emit(result, rank=0, done=None):
    """Rank 0: the FULL record (every section) to gpurun_out/bench_full.json and stderr; the LAST stdout line is the compact one
    (< 4 KB) the driver parses (round 4's 23 KB line overflowed its stdout tail: BENCH_r04.parsed = null)."""
    if rank != 0:
        return
    try:            # RCCL writes a version banner to C stdout, which (a pipe) only drains at exit: push it out first so that the JSON line is the last thing this process prints
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:                                        # noqa: BLE001
        pass
    full = json.dumps(result)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_full.json"), "w") as f:
            f.write(full + "\n")
    except OSError as ex:
        log(f"bench_full.json not written: {ex!r}")
    log("FULL_RECORD " + full)
    sys.stderr.flush()
    if done is None or not done.is_set():
        if done is not None:
            done.set()
        print(compact_line(result), flush=True)


def main(argv=None):
    args = parse_args(argv)
    world, rank, local = dist_env()
    if os.environ.get("BENCH_STUB"):
        return stub_main(args)
    import torch
    if world != args.gpus and (world > 1 or not os.environ.get("BENCH_FORCE_DIST")):
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):      # (the env knob exercises the collective path on a 1-GPU box)
        dist = init_dist("nccl", torch.device("cuda", local))
    b = Bench(args, rank, world, local, dist)
    G, S = b.G, b.S
    if dist:
        b.headline_dist()
    else:
        b.headline()
    dt, D, reps = b.dt, b.D, b.reps
    calls_per_step = N_MATS * world                                  # whole job
    t_call = dt / N_MATS                                             # per-rank time per bucketMul call
    value = calls_per_step * b.eff_bytes / dt / 1e9
    kb = mul_kernel_bytes(D, IN_DIM, OUT_DIM)
    result = {
        "metric": METRIC,
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"bucketMul {IN_DIM}x{OUT_DIM} fp16 buckets, effort {args.effort}, {N_MATS} distinct matrices rotated "
                               f"(one call each per step), fixed-point accumulate (f32 out); {G} independent calls per fused "
                               f"kernel launch; the job's steps are independent: " + (f"rounds of {16 * S} steps (one hipGraph each, four contexts on four streams), "
                               f"one all-gather of the round's outputs per round under the next round's compute, {b.in_flight} step(s) " if dist else
                               f"ONE hipGraph through ONE context, {b.in_flight} step(s) ") +
                               f"in flight{'' if dist else ' (effort_set_overlap)'}, each on its own {N_MATS} matrices", "effort": args.effort, "matrices_per_step": N_MATS,
                   "distinct_matrices": b.n_sets * N_MATS,
                   "inDim": IN_DIM, "outDim": OUT_DIM, "calls_per_launch": G, "steps_in_flight": b.in_flight,
                   "bucket_row_pitch_bytes": (OUT_DIM // 16 * 2 + 127) // 128 * 128 if ALIGN_ROWS else OUT_DIM // 16 * 2,
                   "kernel_geometry(waves,elems,slices)": args.tune if args.tune != "0,0,0" else "heuristic",
                   "partition": "matrices" if world > 1 else "none", "dispatch_rows": D},
        "bytes_per_launch": G * kb, "us_per_call": round(t_call * 1e6, 3),
        "tokens_per_s": round(1.0 / (t_call * 4 * 32), 2),
        "steps_requested": args.steps, "timed_steps": args.steps * reps, "timed_region_ms": round(dt * args.steps * reps * 1e3, 3), "timed_replays": 1,
        "timed_region_note": (f"the {args.steps}-step job {reps} times back to back: rounds of {16 * S} steps (one hipGraph each), a round's all-gather under the next round's compute" if dist else
                              f"the {args.steps}-step job {reps} times back to back in ONE hipGraph, one launch; the {args.steps}-step graph replayed {reps} times instead (its lanes drain at every replay's end): {b.dt_replayed * 1e3:.5f} ms per step"),
    }
    if os.environ.get("BENCH_LAUNCHED"):
        result["launched_by"] = "bench.py --gpus N (its own launcher: launch_ranks)"
    aux_done = threading.Event()
    if dist:
        # the headline above is complete; what follows on N > 1 are auxiliary legs full of collectives.  One rank failing inside
        # one (and skipping its collective) would park the others in RCCL for ever and the driver would get NO line: a timer
        # on every rank prints the headline as measured (rank 0) and ends the process instead
        def aux_overdue():
            if aux_done.is_set():
                return
            aux_done.set()
            if rank == 0:
                result.setdefault("multi_gpu", {})["aborted"] = f"auxiliary legs still running {AUX_DEADLINE_S} s after the headline; line printed by the watchdog"
                sys.stderr.flush()
                print(compact_line(result), flush=True)
            os._exit(0)
        watchdog = threading.Timer(AUX_DEADLINE_S, aux_overdue)
        watchdog.daemon = True
        watchdog.start()
        result["rccl_ranks"] = dist.get_world_size()
        result["multi_gpu"] = {"partition": "matrices (weak scaling: every rank its own 32 matrices per step; north_star's partition: `value`)",
                               "ms_per_step_kernel_only": round(b.dt_kernel * 1e3, 5), "ms_per_step_with_all_gather": round(dt * 1e3, 5),
                               "steps_per_round": 16 * S, "all_gather_bytes_per_rank_per_round": 16 * S * N_MATS * OUT_DIM * 4}
        # the dominant kernel's roofline on N > 1: PER RANK (every rank streams its own matrices from its own HBM), from the kernel-only
        # time of the timed rounds; traffic is measured at N = 1 only (a rocprofv3 child per rank would outlive the driver's patience)
        kbl, t_launch = G * kb, b.dt_kernel / b.launches_per_step
        result["roofline"] = {"bound": "hbm", "kernel": "bucket_mul_kernel", "achieved": round(kbl / t_launch / 1e9, 1), "peak": HBM_PEAK_GBPS,
                              "unit": "GB/s", "frac": round(kbl / t_launch / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                              "bytes_per_launch": kbl, "calls_per_launch": G, "launches_in_flight": b.in_flight, "kernel_us": round(t_launch * 1e6, 3),
                              "scope": "per rank (one GPU's HBM), slowest rank, kernel-only time of the timed rounds; the all-gather of the outputs is in ms_per_step"}
        if not args.headline_only:
            b.multi_gpu_legs(result)

    if rank == 0 and world == 1 and not args.headline_only:
        result["roofline"] = b.roofline()
        if not args.no_sweep:
            result["by_group_size"] = by = b.by_group_size()
            try:
                result["timeit_protocol"] = timeit_protocol(b.ea, b.ea.Gpu(local), b.dev)
            except Exception as ex:                              # noqa: BLE001
                result["timeit_protocol"] = {"error": repr(ex)}
            b.dense_baseline(result, t_call, by["1"]["us_per_call"] * 1e-6)
            result["sweep"] = b.effort_sweep()
            try:
                result["other_configs"] = b.other_configs()
            except Exception as ex:                              # noqa: BLE001
                result["other_configs"] = {"error": repr(ex)}
            if not args.no_decode:
                try:
                    result["decode"] = b.decode()
                except Exception as ex:                          # noqa: BLE001
                    result["decode"] = {"error": repr(ex)}
        if not args.no_cpu:
            try:
                result["cpu_baseline"] = b.cpu_check()
            except Exception as ex:                              # noqa: BLE001  (the oracle is optional infrastructure for the bench)
                result["cpu_baseline"] = {"error": repr(ex)}
    emit(result, rank, aux_done)
    aux_done.set()
    if dist:
        dist.destroy_process_group()
    # leave without tearing anything down: the graphs kept alive (keep()) are not destroyed at interpreter exit either
    sys.stdout.flush()
    sys.stderr.flush()
    # (not under --headline-only: that is the job rocprofv3 wraps, and a profiler flushes its trace in exit handlers)
    if (os.environ.get("BENCH_CHILD") or os.environ.get("BENCH_LAUNCHED") or int(os.environ.get("WORLD_SIZE", "1")) > 1) and not args.headline_only:
        import atexit
        atexit._run_exitfuncs()          # (Python-level exit hooks still run; what is skipped is the teardown of the CUDA objects)
        os._exit(0)
    return 0


def stub_main(args):
    """BENCH_STUB=1: the launch contract without a GPU -- what tests/test_abi.py::test_bench_launches_its_own_ranks runs over gloo.
    Every rank goes through the SAME steps as the real N > 1 job around its timed region (process group from the launcher's
    environment, W untimed warm-up steps, barrier, EXACTLY K steps, barrier, max over ranks, rank 0 prints the one line through
    compact_line); the step itself is a CPU stand-in (a small tensor op + one all-gather through gloo), measured, not claimed as
    bucketMul: the line says `stub`."""
    import torch
    world, rank, _ = dist_env()
    if os.environ.get("BENCH_STUB_FAIL_RANK") == str(rank):      # (the launcher's failure path: a rank that dies before the rendezvous)
        sys.exit(3)
    dist = init_dist(os.environ.get("BENCH_STUB_BACKEND", "gloo")) if world > 1 else None
    x = torch.ones(4096)
    recv = torch.zeros(world * 4096)

    def step():
        y = x * 1.0001
        if dist:
            dist.all_gather_into_tensor(recv, y)
        return y
    for _ in range(args.warmup):
        step()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item()) / max(1, args.steps)
    result = {"metric": METRIC, "value": round(world * N_MATS * 2 * IN_DIM * OUT_DIM / dt / 1e9, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps,
              "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
              "data": "synthetic", "config": {"workload": "STUB: the launch contract over gloo, no GPU work (BENCH_STUB=1)", "partition": "matrices" if world > 1 else "none"},
              "rccl_ranks": world, "stub": True, "full_record": None}
    if os.environ.get("BENCH_LAUNCHED"):
        result["launched_by"] = "bench.py --gpus N (its own launcher: launch_ranks)"
    if rank == 0:
        print(compact_line(result), flush=True)
    if dist:
        dist.destroy_process_group()
    return 0


def _last_record(text):
    for cand in reversed(text.strip().split("\n")):
        try:
            d = json.loads(cand)
            if isinstance(d, dict) and "metric" in d and "value" in d:
                return d
        except ValueError:
            continue
    return None


def error_line(n, why):
    """A contract-shaped line that says why nothing was measured (value null: the driver records the run as unmeasured)."""
    return json.dumps({"metric": METRIC, "value": None, "unit": "GB/s", "n_gpus": n, "error": why, "higher_is_better": True, "scaling": "weak",
                       "vs_baseline": None, "dtype": "f16", "data": "synthetic", "config": {"workload": "not run", "partition": "matrices"}}, separators=(",", ":"))


def launch_ranks(n, argv, timeout_s=None):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks here -- one process per GPU, the
    environment torch.distributed.run would give them (RANK / LOCAL_RANK / WORLD_SIZE, MASTER_ADDR 127.0.0.1, a free MASTER_PORT) --
    relay rank 0's line, and end every rank if one fails (a rank that died would leave the others in a collective for ever).
    Fewer than N visible GPUs: one JSON error line, exit code 2."""
    import signal
    import socket
    import subprocess
    timeout_s = timeout_s or int(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1500"))
    if not os.environ.get("BENCH_STUB"):
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(error_line(n, f"--gpus {n} asked for, {have} GPU(s) visible to this process"), flush=True)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BENCH_LAUNCHED="1",
                HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, start_new_session=True))
    out0, failed, t_end = b"", None, time.time() + timeout_s
    try:
        import selectors
        sel = selectors.DefaultSelector()
        sel.register(procs[0].stdout, selectors.EVENT_READ)
        open0 = True
        while True:
            if open0:
                for key, _ in sel.select(timeout=0.5):
                    chunk = os.read(key.fileobj.fileno(), 65536)
                    if chunk:
                        out0 += chunk
                    else:
                        sel.unregister(key.fileobj)
                        open0 = False
            else:
                time.sleep(0.2)
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
                break
            if all(c == 0 for c in codes) and not open0:
                break
            if time.time() > t_end:
                failed = f"no result within {timeout_s} s"
                break
    finally:
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL if failed else signal.SIGTERM)
                except OSError:
                    pass
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:                                    # noqa: BLE001
                pass
    line = _last_record(out0.decode(errors="replace"))
    if line is not None:                                         # (rank 0 may have printed its line before another rank failed in teardown)
        if failed:
            line["note"] = (line.get("note", "") + "; " if line.get("note") else "") + f"launcher: {failed} after rank 0's line"
        print(json.dumps(line, separators=(",", ":")), flush=True)
        return 0
    print(error_line(n, f"launcher: {failed or 'rank 0 printed no record'}"), flush=True)
    return 1


def guarded():
    """N = 1: the measurement runs in a CHILD process and this one relays its line: should the child ever die without a record, the
    contract-complete subset (headline + roofline + cpu_baseline: --no-sweep --no-pmc) is run instead, then the headline alone;
    what was dropped is named in the line's `note`.  (Round 5 added this after one run in ~40 ended in glibc's `free(): invalid
    pointer`; DESIGN 5 has what round 6's hunt found.)"""
    import subprocess
    env = dict(os.environ, BENCH_CHILD="1")
    note = None
    for extra in ([], ["--no-sweep", "--no-pmc"], ["--headline-only"]):
        p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + extra, env=env, stdout=subprocess.PIPE)
        line = _last_record(p.stdout.decode(errors="replace"))
        if line is not None:
            if note:
                line["note"] = note
            print(json.dumps(line, separators=(",", ":")), flush=True)
            return 0
        note = (note + "; " if note else "") + f"the run with flags {extra or ['(default)']} ended with exit code {p.returncode} and no record"
        log("bench.py: " + note + " -- falling back")
    return 1


def entry(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and not os.environ.get("BENCH_CHILD"):
        n = parse_args(argv).gpus
        if n > 1:                                                # no launcher around us: be the launcher
            return launch_ranks(n, argv)
    if os.environ.get("BENCH_CHILD") or os.environ.get("BENCH_NO_GUARD") or world > 1 or os.environ.get("BENCH_FORCE_DIST") or os.environ.get("BENCH_STUB") \
            or "--headline-only" in argv:
        return main(argv)
    return guarded()


if __name__ == "__main__":
    sys.exit(entry())
