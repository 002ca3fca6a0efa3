#!/usr/bin/env python
"""VERDICT r05 item 4, measured BEFORE building it into the library: one group of n independent calls as ONE launch against the
same n calls as k launches of n/k on k lanes of one context (effort_set_overlap(k)), the lanes JOINED after every group -- what
effort_bucketmul_group would do inside one call if it split a big group across its lanes (second launch's head under the first's
stream, its tail under the first's tail).  Steps rotate over 4 disjoint sets of 32 matrices (nothing out of the Infinity Cache);
8 groups per hipGraph, host clock around 100 replays.

    python tools/lab/split_probe.py [--shape 4096x11008] [--effort 0.25] [--groups 16,32] [--splits 1,2,4]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x11008")
    ap.add_argument("--effort", type=float, default=0.25)
    ap.add_argument("--groups", default="16,32")
    ap.add_argument("--splits", default="1,2,4")
    ap.add_argument("--uneven", type=int, default=0, help="1: the first launch gets 3/4 of the calls when split in two")
    args = ap.parse_args()
    inDim, outDim = (int(x) for x in args.shape.split("x"))
    import effort_amd as ea
    from bench import make_weights
    dev = torch.device("cuda", 0)
    sets = [make_weights(ea, 32, inDim, outDim, 1234 + 32 * k, dev, keep_core=False) for k in range(4)]
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    v = torch.randn(inDim, generator=gen, device=dev)
    outs = [[torch.zeros(outDim, device=dev) for _ in range(32)] for _ in range(4)]
    for rep in range(2):
        for n in (int(x) for x in args.groups.split(",")):
            for k in (int(x) for x in args.splits.split(",")):
                g = ea.Gpu(0)
                g.set_overlap(max(1, k))

                def run():
                    for step in range(8):
                        ws, os_ = sets[step % 4], outs[step % 4]
                        for c0 in range(0, 32, n):                     # the groups of a step, one after the other (each joined)
                            calls = [(v, ws[i], None, os_[i], args.effort) for i in range(c0, c0 + n)]
                            if k == 1:
                                ea.bucketMulGroup(calls, gpu=g)
                            else:
                                if args.uneven and k == 2:
                                    cut = [0, n * 3 // 4, n]
                                else:
                                    cut = [n * j // k for j in range(k + 1)]
                                for j in range(k):
                                    ea.bucketMulGroup(calls[cut[j]:cut[j + 1]], gpu=g)
                                g.join()
                run()
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    run()
                g._bind_stream()
                for _ in range(10):
                    gr.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(100):
                    gr.replay()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 100 / 8 / 32
                print(f"rep {rep} group {n:2d} as {k} launch(es){' (3/4 + 1/4)' if args.uneven and k == 2 else ''}: {dt * 1e6:7.3f} us per call = {dt * 1e6 * n:7.2f} us per group", flush=True)
                del gr
                g.close()


if __name__ == "__main__":
    main()
