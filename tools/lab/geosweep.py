#!/usr/bin/env python
"""Launch-geometry sweep of the decode loop's multiply launches (lone calls and small groups, possibly of mixed shapes).

    python tools/lab/geosweep.py --launch 4096x4096,4096x1024,4096x1024 [--effort 0.25] [--configs "8,2,0;8,2,48;16,1,32"]

A launch = one effort_bucketmul_group of the listed shapes (in x out) on ONE input vector.  Per configuration
(waves,elems,slices; 0 = heuristic): microseconds per launch from a hipGraph of 16 back-to-back launches over rotating
weight sets, best of 3.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launch", required=True)
    ap.add_argument("--effort", type=float, default=0.25)
    ap.add_argument("--configs", default="0,0,0")
    ap.add_argument("--sets", type=int, default=8)
    a = ap.parse_args()
    import effort_amd as ea
    from bench import make_weights
    dev = torch.device("cuda", 0)
    g = ea.gpu(0)
    shapes = [tuple(int(x) for x in s.split("x")) for s in a.launch.split(",")]
    inDim = shapes[0][0]
    assert all(s[0] == inDim for s in shapes)
    wsets = []
    for k in range(a.sets):
        wsets.append([make_weights(ea, 1, i, o, 100 * k + j, dev, keep_core=False)[0] for j, (i, o) in enumerate(shapes)])
    v = torch.randn(inDim, device=dev)
    outs = [[torch.zeros(o, device=dev) for (_, o) in shapes] for _ in range(a.sets)]
    res = []
    for cfg in a.configs.split(";"):
        W, E, S = (int(x) for x in cfg.split(","))
        try:
            g.set_tuning(W, E, S)

            def run():
                for rep in range(16):
                    k = rep % a.sets
                    ea.bucketMulGroup([(v, ew, None, o, a.effort) for ew, o in zip(wsets[k], outs[k])])
            run()
            g.eval()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                run()
            g._bind_stream()
            for _ in range(10):
                gr.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(40):
                    gr.replay()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 40 / 16)
            res.append((best * 1e6, cfg))
            del gr
        except Exception as ex:
            res.append((float("inf"), cfg + " " + repr(ex)[:60]))
        finally:
            g.set_tuning(0, 0, 0)
    base = [r for r in res if r[1] == "0,0,0"]
    res.sort()
    print(f"launch {a.launch} effort {a.effort}: heuristic {base[0][0]:.2f} us" if base else f"launch {a.launch}")
    for t, cfg in res[:8]:
        print(f"    {cfg:12s} {t:8.2f} us")


if __name__ == "__main__":
    main()
