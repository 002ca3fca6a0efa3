#!/usr/bin/env python
"""tools/bench_extra.py -- the sections round 5's bench.py carried behind the headline and the driver's one command no longer runs
(VERDICT r05 item 7): `by_streams`, `shared_matrices`, `four_contexts`, the timeit-protocol variants, the quality sweeps
(heavy-tailed input, a structured matrix, the structured decode model), the other shapes, `shard_projection` (BASELINE config 4's
column split projected on one GPU) and the one-GPU `layer_latency` (a world of one through RCCL + projected ranks).

    python tools/bench_extra.py [--sections a,b,...] [--steps 20 --warmup 5 ...bench.py's flags]

Same state object as bench.py (`bench.Bench`: S disjoint sets of 32 converted 4096 x 11008 matrices, one context with lanes), same
timing helpers.  Writes gpurun_out/bench_extra.json and prints a short summary line; each section is on its own try / except.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402

SECTIONS = ("by_streams", "shared_matrices", "four_contexts", "timeit_protocol", "heavy_tailed_input", "sweep_structured", "other_shapes",
            "shard_projection", "decode_quality", "layer_latency")


def by_streams(b, res):
    """ONE context, n launches in flight (effort_set_overlap), each step on its own matrices (one long graph, like the headline)."""
    kb = B.mul_kernel_bytes(b.D, B.IN_DIM, B.OUT_DIM)
    torch = b.torch
    while len(b.out_sets) < 4:
        b.out_sets.append(torch.zeros((B.N_MATS, B.OUT_DIM), device=b.dev))
    bs = {}
    for ns in (1, 2, 3, 4):
        jb = b.job if ns == b.S else B.LaneJob(b.ea, b.local, ns, b.tune)
        gn = jb.capture(b.mul_step(b.args.effort, wsets=b.ew_sets[:max(1, min(ns, len(b.ew_sets)))]), 192)
        bs[str(ns)] = b.rate(B.time_graph(gn, None) / 192 / B.N_MATS, kb)
        del gn
    res["by_streams"] = bs
    res["by_streams_note"] = "one effort_ctx, effort_set_overlap(n): the library keeps n launches in flight; every step in flight on its own 32 matrices"


def shared_matrices(b, res):
    """Round 2's job: every step in flight on the SAME 32 matrices (what the Infinity Cache can contribute)."""
    kb = B.mul_kernel_bytes(b.D, B.IN_DIM, B.OUT_DIM)
    gn = b.job.capture(b.mul_step(b.args.effort), 192)
    res["shared_matrices"] = b.rate(B.time_graph(gn, None) / 192 / B.N_MATS, kb)
    # ... and with the row stream's ordinary cache policy, what such a caller asks for (effort_set_row_reuse: the policy is captured with the launch)
    b.job.ctx.set_row_reuse(True)
    try:
        gr = b.job.capture(b.mul_step(b.args.effort), 192)
    finally:
        b.job.ctx.set_row_reuse(False)
    res["shared_matrices_row_reuse"] = b.rate(B.time_graph(gr, None) / 192 / B.N_MATS, kb)


def four_contexts(b, res):
    """The overlap built by the CALLER from four contexts on four streams (round 2's way) instead of the library's lanes."""
    kb = B.mul_kernel_bytes(b.D, B.IN_DIM, B.OUT_DIM)
    four = B.Job(b.ea, b.local, b.S, b.tune)
    gn = four.capture(b.mul_step(b.args.effort, wsets=b.ew_sets), 192)
    res["four_contexts"] = b.rate(B.time_graph(gn, None) / 192 / B.N_MATS, kb)
    del gn, four


def timeit_full(b, res):
    res["timeit_protocol"] = B.timeit_protocol(b.ea, b.ea.Gpu(b.local), b.dev, efforts=(1.0, 0.7, 0.5, 0.25, 0.15),
                                               variants=(("as_written", 1, 1), ("as_written_overlap4", 4, 1), ("four_outputs_overlap4", 4, 4)),
                                               from_graph=True, dense=(("dense_hip_kernel", False), ("dense_rocblas", True)))


def heavy_tailed_input(b, res):
    """A real rms-normed state has outlier channels: v * exp(N(0,1)), same seeds."""
    torch, ea, job, S, G = b.torch, b.ea, b.job, b.S, b.G
    vh = b.v * torch.exp(torch.randn(B.IN_DIM, generator=b.gen, device=b.dev, dtype=torch.float32))
    slotL, last = (24 - 1) % S, B.N_MATS - 1
    ewsL = b.ew_sets[slotL % len(b.ew_sets)]
    dense = torch.zeros(B.OUT_DIM, device=b.dev)
    heavy = {}
    for e in (0.25, 0.5):
        gh = job.capture(b.mul_step(e, vec=vh, wsets=b.ew_sets), 24)
        Dh = job.last_dispatch_count(24, (B.N_MATS - 1) % G)
        th = B.time_graph(gh, None, reps=2) / 24 / B.N_MATS
        ea.basicMul(vh, ewsL[last].core, dense)
        heavy[str(e)] = {"dispatch_rows": Dh, "us_per_call": round(th * 1e6, 3),
                         "achieved_GBps": round(B.algorithmic_bytes(Dh, B.IN_DIM, B.OUT_DIM) / th / 1e9, 1),
                         "frac_of_hbm_peak": round(B.algorithmic_bytes(Dh, B.IN_DIM, B.OUT_DIM) / th / 1e9 / B.HBM_PEAK_GBPS, 4),
                         "cos_vs_dense": round(ea.cosineSimilarityTo(b.out_sets[slotL][last], dense), 5)}
        del gh
    res["heavy_tailed_input"] = heavy


def sweep_structured(b, res):
    """One STRUCTURED matrix (effort_amd.decode.structured_matrix) and a state as a norm layer with outlier channels leaves it: the
    reference's own check (benchmarks/benchmark.swift:166-177: cos-sim of expertMul vs basicMul)."""
    from effort_amd.decode import structured_matrix, structured_norm_weights
    torch, ea = b.torch, b.ea
    Ws = structured_matrix(B.OUT_DIM, B.IN_DIM, b.gen, b.dev)
    es = ea.ExpertWeights.from_core(Ws)
    es.handle
    x = torch.randn(B.IN_DIM, generator=b.gen, device=b.dev, dtype=torch.float32)
    vs_ = (x / x.pow(2).mean().sqrt()) * structured_norm_weights(B.IN_DIM, b.gen, b.dev).float()
    os_, od_ = torch.zeros(B.OUT_DIM, device=b.dev), torch.zeros(B.OUT_DIM, device=b.dev)
    ea.basicMul(vs_, Ws, od_)
    ss = []
    for e in B.SWEEP:
        ea.bucketMul(vs_, es, None, os_, e)
        ss.append({"effort": e, "dispatch_rows": b.g.last_dispatch_count(), "cos_vs_dense": round(ea.cosineSimilarityTo(os_, od_), 5)})
    res["sweep_structured"] = ss


def other_shapes(b, res):
    other = {}
    for name, (iD, oD, seed, efforts) in {"4096x14336 fp16 (the reference's timed shape)": (4096, 14336, 7321, (0.25,)),
                                           "14336x4096 fp16": (14336, 4096, 5321, (0.25,))}.items():
        sets_w = b.make_sets(16, iD, oD, seed)
        other[name] = {}
        for e in efforts:
            other[name].update(b.three(sets_w, oD, iD, e))
        del sets_w
    res["other_shapes"] = other


def shard_times(b, full_sets, inD, outD, effort, per_launch):
    """Per-rank kernel-only time of a bucket-column split over G GPUs, on this one: a launch = the rank's column shards of
    `per_launch` matrices (rank i % G of matrix i); the steps in flight work disjoint matrix sets."""
    from effort_amd.sharded import ShardedExpertWeights
    torch, one, job, S = b.torch, b.one, b.job, b.S
    rows = {}
    for Gw in (1, 2, 4, 8):
        if (outD // 16) % Gw or (outD // Gw) % 32:
            continue
        sh_sets = full_sets if Gw == 1 else [[ShardedExpertWeights.from_full(e, i % Gw, Gw).local for i, e in enumerate(fs)] for fs in full_sets]
        for fs in sh_sets:
            for x in fs:
                x.handle
                if B.ALIGN_ROWS:
                    x.align_rows()
        lo = outD // Gw
        vx = b.v if inD == B.IN_DIM else torch.randn(inD, generator=b.gen, device=b.dev, dtype=torch.float32)
        nm = len(sh_sets[0])
        sets_x = [torch.zeros((nm, lo), device=b.dev) for _ in range(max(S, len(sh_sets)))]
        r = {}
        for nm_, jb, nst in (("1 launch in flight", one, 8), (f"{S} in flight", job, 16)):
            if jb is one:
                one.S = len(sh_sets)
            try:
                gx = jb.capture(b.mul_step(effort, vec=vx, sets=sets_x, group=per_launch, wsets=sh_sets), nst)
            finally:
                one.S = 1
            Dx = jb.last_dispatch_count(nst, (nm - 1) % per_launch)
            est = B.time_graph(gx, None, reps=2)
            tx = B.time_graph(gx, None, reps=max(4, int(0.05 / max(est, 1e-6)) + 1)) / nst     # per step = per rank per step; timed over >= 50 ms like the headline
            del gx
            ab = nm * B.algorithmic_bytes(Dx, inD, lo)
            r[nm_] = {"us_per_step_per_rank": round(tx * 1e6, 2), "frac_of_hbm_peak": round(ab / tx / 1e9 / B.HBM_PEAK_GBPS, 4)}
        r["columns_per_rank"] = outD // 16 // Gw
        r["row_pitch_bytes"] = sh_sets[0][0].align_rows() if B.ALIGN_ROWS else outD // 16 // Gw * 2
        rows[str(Gw)] = r
        del sh_sets
    for Gw, r in rows.items():
        for k2 in list(r):
            if isinstance(r[k2], dict):
                r[k2]["kernel_only_scaling_efficiency"] = round(rows["1"][k2]["us_per_step_per_rank"] / (int(Gw) * r[k2]["us_per_step_per_rank"]), 3)
    return rows


def shard_projection(b, res):
    """BASELINE.json configs[3] projected on ONE GPU (rank r holds columns [r*C/G, (r+1)*C/G) of every matrix, stats /
    probes replicated; its kernel-only work is a launch of column shards.  Strong scaling: efficiency = t(G=1) / (G * t(G)))."""
    sp = {"note": "per-rank kernel-only time of a bucket-column split, measured on one GPU: a launch of `calls per launch` column shards "
                  "(rank i % G of matrix i), 25 % effort, the steps in flight on disjoint matrix sets; efficiency = t(G=1) / (G * t(G)); "
                  "the all-gather of the outputs is not in it"}
    try:
        sp["4096x11008, 32 calls per launch"] = shard_times(b, b.ew_sets, B.IN_DIM, B.OUT_DIM, 0.25, 32)
    except Exception as ex:                                      # noqa: BLE001
        sp["4096x11008, 32 calls per launch"] = {"error": repr(ex)}
    for iD, oD, seed in ((4096, 4096, 4321), (4096, 14336, 7321), (14336, 4096, 5321)):
        sets_w = b.make_sets(16, iD, oD, seed)
        try:
            sp[f"{iD}x{oD}, 16 calls per launch"] = shard_times(b, sets_w, iD, oD, 0.25, 16)
        except Exception as ex:                                  # noqa: BLE001
            sp[f"{iD}x{oD}, 16 calls per launch"] = {"error": repr(ex)}
        del sets_w
    res["shard_projection"] = sp


def decode_quality(b, res):
    """Quality on STRUCTURED synthetic weights (heavy tails, channel scales, outlier norm channels: what trained models have and
    i.i.d. Gaussians lack), the reference's protocol (benchmarks/benchmark.swift:128-156): greedy text at effort 1.0, then
    teacher-forced predictions at every effort against the effort-1.0 ones.  Not evidence of quality on trained weights (no
    checkpoints offline): reported, never headlined."""
    from effort_amd.decode import Decoder, MistralConfig, Model, kl_divergence
    torch = b.torch
    b.g.set_tuning(0, 0, 0)
    torch.cuda.empty_cache()
    prompt = [1, 733, 16289, 28793, 22557]
    model = Model.random(MistralConfig(), seed=2, structured=True)
    ntq = 224
    dec = Decoder(model, maxTokens=ntq + 8)
    B._GRAPHS_FOR_LIFE.append(dec._graphs)                        # (the dict itself: whatever it collects stays alive)
    ids_1, _, _ = dec.run(prompt, ntq, effort=1.0)
    forced = prompt + ids_1[len(prompt) - 1:-1]
    _, _, lg_dn = dec.run(forced, ntq, dense=True, forced=True, collect_logits=True)
    _, _, lg_1 = dec.run(forced, ntq, effort=1.0, forced=True, collect_logits=True)
    control = lg_1.argmax(-1)
    q = {"model": "Mistral-7B shapes, 32 layers, STRUCTURED random weights (effort_amd.decode.structured_matrix)", "tokens": ntq,
         "protocol": "benchmarks/benchmark.swift:128-156: teacher-forced on the effort-1.0 greedy text; agreement = predictions equal to the effort-1.0 predictions",
         "effort": {}}
    for e in (1.0, 0.7, 0.5, 0.35, 0.25, 0.15, 0.1):
        _, dt_e, _ = dec.run(prompt, 40, effort=e)
        _, _, lg_e = dec.run(forced, ntq, effort=e, forced=True, collect_logits=True)
        q["effort"][str(e)] = {"agreement_vs_effort_1.0": round(float((lg_e.argmax(-1) == control).float().mean()), 4),
                               "agreement_vs_dense": round(float((lg_e.argmax(-1) == lg_dn.argmax(-1)).float().mean()), 4),
                               "kl_vs_dense": round(kl_divergence(lg_dn, lg_e), 5), "tokens_per_s": round(1 / dt_e, 1)}
    res["decode_quality_structured"] = q
    del dec, model


def layer_latency(b, res):
    """BASELINE config 4's latency case on ONE GPU: a world of one through RCCL + projected ranks."""
    b.torch.cuda.empty_cache()
    lg = b.ea.Gpu(b.local)
    lg.comm_create(0, 1, b.ea.Gpu.comm_unique_id())
    try:
        res["layer_latency"] = B.layer_latency(b.ea, lg, b.dev, 0, 1)
    finally:
        lg.comm_destroy()
        del lg


RUN = {"by_streams": by_streams, "shared_matrices": shared_matrices, "four_contexts": four_contexts, "timeit_protocol": timeit_full,
       "heavy_tailed_input": heavy_tailed_input, "sweep_structured": sweep_structured, "other_shapes": other_shapes,
       "shard_projection": shard_projection, "decode_quality": decode_quality, "layer_latency": layer_latency}


def main():
    argv = sys.argv[1:]
    sections = list(SECTIONS)
    if "--sections" in argv:
        i = argv.index("--sections")
        sections = [s for s in argv[i + 1].split(",") if s]
        del argv[i:i + 2]
    args = B.parse_args(argv)
    import torch
    torch.cuda.set_device(0)
    b = B.Bench(args)
    b.headline()                                                  # (the sections price themselves against the headline's dispatch count)
    res = {"what": "bench.py's auxiliary sections (tools/bench_extra.py)", "headline_us_per_call": round(b.dt / B.N_MATS * 1e6, 3), "dispatch_rows": b.D}
    for name in sections:
        t0 = time.perf_counter()
        try:
            RUN[name](b, res)
        except Exception as ex:                                  # noqa: BLE001
            res[name] = {"error": repr(ex)[:400]}
        B.log(f"bench_extra: {name} {time.perf_counter() - t0:.1f} s")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_extra.json"), "w") as f:
        json.dump(res, f)
        f.write("\n")
    print(json.dumps({k: (v if not isinstance(v, dict) or len(json.dumps(v)) < 300 else "...") for k, v in res.items()}), flush=True)
    sys.stderr.flush()
    os._exit(0)          # (bench.keep(): no captured graph is destroyed, at exit either)


if __name__ == "__main__":
    main()
