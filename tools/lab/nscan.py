#!/usr/bin/env python
"""us per launch and per call of n-call group launches, n = 1 .. N, for one shape (hipGraph replays, rotating matrices): a scan for holes in the
launch-geometry heuristics -- the time per launch should grow smoothly with n.

    python tools/lab/nscan.py --shape 4096x4096 [--effort 0.25] [--ns 1,2,3,...] [--q4 0]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x4096")
    ap.add_argument("--effort", type=float, default=0.25)
    ap.add_argument("--ns", default="1,2,3,4,5,6,7,8,9,10,11,12,14,16,20,24,28,32")
    ap.add_argument("--q4", type=int, default=0)
    ap.add_argument("--mats", type=int, default=0)
    ap.add_argument("--mix", default="", help="comma list of shapes sharing one inDim (e.g. 4096x4096,4096x1024,4096x1024 = Wq | Wk | Wv): a launch of n calls cycles through them")
    ap.add_argument("--tune", default="0,0,0", help="waves,elems,slices (effort_set_tuning; 0 = the heuristic)")
    ap.add_argument("--overlap", type=int, default=1, help="K > 1: effort_set_overlap(K) -- the launches of a replay are independent and up to K are in flight")
    args = ap.parse_args()
    inDim, outDim = (int(x) for x in args.shape.split("x"))
    import effort_amd as ea
    from bench import make_weights
    dev = torch.device("cuda", 0)
    g = ea.gpu(0)
    ns = [int(x) for x in args.ns.split(",")]
    nm = args.mats or max(32, min(96, (1 << 31) // (inDim * outDim * 2)))          # enough matrices that a launch does not find its rows in the caches
    if args.mix:
        shapes = [tuple(int(x) for x in sh.split("x")) for sh in args.mix.split(",")]
        assert all(sh[0] == shapes[0][0] for sh in shapes)
        inDim = shapes[0][0]
        nm = args.mats or 48
        per = [make_weights(ea, (nm + len(shapes) - 1) // len(shapes), sh[0], sh[1], 1234 + 100 * k, dev, keep_core=False, q4=bool(args.q4)) for k, sh in enumerate(shapes)]
        ews = [per[i % len(shapes)][i // len(shapes)] for i in range(nm)]
        outs_dim = [shapes[i % len(shapes)][1] for i in range(nm)]
        args.shape = "mix(" + args.mix + ")"
    else:
        ews = make_weights(ea, nm, inDim, outDim, 1234, dev, keep_core=False, q4=bool(args.q4))
        outs_dim = [outDim] * nm
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    v = torch.randn(inDim, generator=gen, device=dev)
    outs = [torch.zeros(outs_dim[i], device=dev) for i in range(len(ews))]
    keep = []
    g.set_tuning(*(int(x) for x in args.tune.split(",")))
    if args.overlap > 1:
        g.set_overlap(args.overlap)
    for n in ns:
        chunks = [list(range(i, i + n)) for i in range(0, nm - n + 1, n)]

        def run():
            for ch in chunks:
                ea.bucketMulGroup([(v, ews[k], None, outs[k], args.effort) for k in ch])
            if args.overlap > 1:
                g.join()
        run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        g._bind_stream()
        keep.append(gr)
        for _ in range(10):
            gr.replay()
        torch.cuda.synchronize()
        reps = max(20, 4000 // len(chunks) // max(1, n // 4))
        t0 = time.perf_counter()
        for _ in range(reps):
            gr.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps / len(chunks)
        sl = len(g.slice_counts(0)), len(g.slice_counts(n - 1))
        print(f"{args.shape} effort {args.effort} q4 {args.q4}{' lanes ' + str(args.overlap) if args.overlap > 1 else ''} n {n:2d}: {dt * 1e6:8.2f} us/launch {dt * 1e6 / n:7.2f} us/call  slices {sl[0]}/{sl[1]}", flush=True)


if __name__ == "__main__":
    main()
