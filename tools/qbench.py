#!/usr/bin/env python
"""Quick A/B timing of grouped bucketMul launches (hipGraph replays over 32 rotating matrices).

    python tools/qbench.py --configs "0,0,0:-1;16,4,0:1" [--shape 4096x11008] [--effort 0.25] [--group 32] [--reps 3] [--q4 0]

A config is waves,elems,slices:workgroups-per-CU.  Prints per config and repetition: us per launch (host clock around 200
replays) and the in-kernel device-clock span.  Library under test: $EFFORT_HIP_LIB or the in-tree build.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x11008")
    ap.add_argument("--effort", type=float, default=0.25)
    ap.add_argument("--group", type=int, default=32)
    ap.add_argument("--mats", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--q4", type=int, default=0)
    ap.add_argument("--configs", default="0,0,0:-1")
    ap.add_argument("--tag", default="")
    ap.add_argument("--streams", type=int, default=1, help="spread the launches of a step round-robin over K streams / contexts")
    ap.add_argument("--steps-per-graph", type=int, default=1, help="steps captured into one graph")
    ap.add_argument("--overlap", type=int, default=1, help="K > 1: ONE context with effort_set_overlap(K); the steps of a graph write K rotating output sets")
    ap.add_argument("--no-outliers", type=int, default=0, help="Q4: register the bundles without their outlier tables")
    ap.add_argument("--fused", default="", help="comma list of gate,norm,resid: every call derives its input / adds its residual in the launch (effort_bucketmul_group_fused)")
    ap.add_argument("--no-align", type=int, default=0, help="1: the reference's dense rows (2 * cols bytes apart) instead of rows on whole 128-byte lines")
    ap.add_argument("--row-reuse", type=int, default=0, help="1: effort_set_row_reuse(1), the ordinary cache policy on the row stream (default: nt)")
    ap.add_argument("--tails", default="", help="lab library: comma list of EFFORT_TAIL_CALLS values (the last k calls of a group at EFFORT_TAIL_MULT x the slices), each timed in turn")
    ap.add_argument("--split", type=int, default=0, help="1: the cutoffs in a kernel of their own before the multiply (the device-clock span then covers the multiply alone)")
    args = ap.parse_args()
    inDim, outDim = (int(x) for x in args.shape.split("x"))
    import effort_amd as ea
    import bench
    from bench import make_weights
    if args.no_align:
        bench.ALIGN_ROWS = False
    dev = torch.device("cuda", 0)
    g = ea.gpu(0)
    lab = bool(getattr(ea.lib(), "effort_is_lab_build", lambda: 1)())     # (device-clock stamps exist in the lab library only; an older A/B build has them too)
    ews = make_weights(ea, args.mats, inDim, outDim, 1234, dev, keep_core=False, q4=bool(args.q4))
    if args.q4 and args.no_outliers:
        ews = [ea.ExpertWeights(e.buckets, e.stats, e.probes, inSize=inDim, outSize=outDim, q4=True) for e in ews]
        for e in ews:
            e.handle
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    v = torch.randn(inDim, generator=gen, device=dev)
    outs = [torch.zeros(outDim, device=dev) for _ in ews]
    osets = [[torch.zeros(outDim, device=dev) for _ in ews] for _ in range(max(1, args.overlap))]
    if args.overlap > 1:
        g.set_overlap(args.overlap)
    if args.row_reuse:
        g.set_row_reuse(True)
    fz = {}
    if "gate" in args.fused:
        fz["gate"] = torch.randn(inDim, generator=gen, device=dev)
    if "norm" in args.fused:
        fz["norm"] = (1 + 0.1 * torch.randn(inDim, generator=gen, device=dev)).to(torch.float16)
    resid = torch.randn(outDim, generator=gen, device=dev) if "resid" in args.fused else None
    items = list(zip(ews, outs))
    chunks = [items[i:i + args.group] for i in range(0, len(items), args.group)]
    tails = [t for t in args.tails.split(",") if t] or [None]
    for rep in range(args.reps):
      for tail in tails:
        if tail is not None:
            os.environ["EFFORT_TAIL_CALLS"] = tail           # (read by the lab library at every call)
        for cfg in args.configs.split(";"):
            tune, per = cfg.split(":")
            g.set_tuning(*(int(x) for x in tune.split(",")))
            g.set_persistent(int(per))
            g.set_split_cutoff(bool(args.split))
            K = args.streams
            if K > 1 and not hasattr(main, "ctxs"):
                main.ctxs = [ea.Gpu(0) for _ in range(K)]
                main.sts = [torch.cuda.Stream() for _ in range(K)]
                main.fork = torch.cuda.Event()                     # (events that outlive the graphs: bench.Job says why not wait_stream)
                main.join = [torch.cuda.Event() for _ in range(K)]
            if K > 1:
                for c in main.ctxs:
                    c.set_tuning(*(int(x) for x in tune.split(",")))
                    c.set_persistent(int(per))

            def run():
                for rep_ in range(args.steps_per_graph):
                    if args.overlap > 1:
                        oo = osets[rep_ % args.overlap]
                        for c0 in range(0, len(ews), args.group):
                            ea.bucketMulGroup([(v, ews[k], None, oo[k], args.effort) for k in range(c0, min(len(ews), c0 + args.group))])
                    elif K == 1:
                        for ch in chunks:
                            if fz or resid is not None:
                                ea.bucketMulGroup([(v, ew, None, o, args.effort, dict(fz, **({"resid": resid} if resid is not None else {}))) for ew, o in ch])
                            else:
                                ea.bucketMulGroup([(v, ew, None, o, args.effort) for ew, o in ch])
                    else:                                # step `rep_` of the graph goes to stream rep_ % K (own context: own scratch)
                        s0 = torch.cuda.current_stream()
                        st, cx = main.sts[rep_ % K], main.ctxs[rep_ % K]
                        if rep_ == 0:
                            main.fork.record(s0)
                        if rep_ < K:
                            st.wait_event(main.fork)
                        with torch.cuda.stream(st):
                            for ch in chunks:
                                ea.bucketMulGroup([(v, ew, None, o, args.effort) for ew, o in ch], gpu=cx)
                if K > 1:
                    for st, ev in zip(main.sts, main.join):
                        ev.record(st)
                        torch.cuda.current_stream().wait_event(ev)
                if args.overlap > 1:
                    g.join()

            def timed(n):
                run()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    run()
                g._bind_stream()
                for _ in range(30):
                    gr.replay()
                torch.cuda.synchronize()
                if lab:
                    g.kernel_clock()
                t0 = time.perf_counter()
                for _ in range(n):
                    gr.replay()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) / n / len(chunks) / args.steps_per_graph
            g.enable_kernel_timing(0)
            dt = timed(300)
            span = "n/a (product library: EFFORT_HIP_LIB=lab has the stamps)"
            if lab:
                g.enable_kernel_timing(2)
                timed(50)
                span = f"{g.kernel_clock()['mul_us']:8.2f} us"
                g.enable_kernel_timing(0)
            print(f"{args.tag}{'' if tail is None else '-tc' + tail} rep {rep} cfg {cfg:14s} effort {args.effort} group {args.group}: {dt * 1e6:8.2f} us/launch  device-clock span {span}", flush=True)


if __name__ == "__main__":
    main()
