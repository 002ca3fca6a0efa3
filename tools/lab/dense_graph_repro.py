#!/usr/bin/env python
"""Round 6 hunt for the `free(): invalid pointer`: the first hunt (tools/lab/heap_hunt.sh, 26 full bench runs under the
checking allocator) died in bench.py's DENSE BASELINE section -- `time_graph(one.capture(dense_step, 4), ...)`: basicMul (effort_dense_gemv)
captured into a hipGraph, replayed, the graph destroyed.  This loops exactly that, one variant per process so that the culprit is named:

    python tools/lab/dense_graph_repro.py --backend rocblas|hip --job lane|four [--iters 300] [--mats 8]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="rocblas")
    ap.add_argument("--job", default="lane")
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--mats", type=int, default=8)
    ap.add_argument("--keep", type=int, default=0, help="1: keep every graph alive (never destroy one)")
    a = ap.parse_args()
    import effort_amd as ea
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    cores = [(torch.randn((11008, 4096), generator=gen, device=dev) * 0.02).to(torch.float16) for _ in range(a.mats)]
    v = torch.randn(4096, generator=gen, device=dev)
    outs = [torch.zeros((a.mats, 11008), device=dev) for _ in range(4)]
    g = ea.gpu(0)
    job = B.LaneJob(ea, 0, 1, ctx=g) if a.job == "lane" else B.Job(ea, 0, 4)
    for c in job.ctxs:
        c.set_dense_backend(a.backend == "rocblas")

    def step(ctx, slot):
        for k, W in enumerate(cores):
            ea.basicMul(v, W, outs[slot][k], gpu=ctx)
    kept = []
    t0 = time.time()
    for it in range(a.iters):
        gr = job.capture(step, 4 if a.job == "lane" else 8)
        B.time_graph(gr, None, reps=3)
        if a.keep:
            kept.append(gr)
        del gr
        if it % 50 == 49:
            print(f"{a.backend} {a.job}: {it + 1} captures + replays + destroys, {time.time() - t0:.0f} s", flush=True)
    print(f"{a.backend} {a.job}: done, {a.iters} iterations clean", flush=True)


if __name__ == "__main__":
    main()
