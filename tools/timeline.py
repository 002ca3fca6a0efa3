#!/usr/bin/env python
"""Per-item timeline of one multiply launch (timing mode 3: every work item leaves its phase stamps).

    python tools/timeline.py [--shape 4096x11008] [--effort 0.25] [--groups 32,3,1] [--tune 0,0,0] [--q4 0] [--out gpurun_out/timeline.json]

Prints, per group size: the launch span, when the first / median / last workgroup reaches each phase, how many
workgroups are in the streaming phase over time, and the per-item phase durations.  The raw records go to --out.
"""
import argparse
import json
import os
os.environ.setdefault("EFFORT_HIP_LIB", "lab")     # stamps / traces live in libeffort_hip_lab.so (the shipped kernels carry none)
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x11008")
    ap.add_argument("--effort", type=float, default=0.25)
    ap.add_argument("--groups", default="32,3,1")
    ap.add_argument("--tune", default="0,0,0")
    ap.add_argument("--q4", type=int, default=0)
    ap.add_argument("--persistent", type=int, default=-1)
    ap.add_argument("--out", default="gpurun_out/timeline.json")
    ap.add_argument("--replay", type=int, default=0, help="trace the last of N back-to-back graph replays instead of one eager launch")
    args = ap.parse_args()
    inDim, outDim = (int(x) for x in args.shape.split("x"))
    import effort_amd as ea
    from bench import make_weights
    dev = torch.device("cuda", 0)
    g = ea.gpu(0)
    g.set_tuning(*(int(x) for x in args.tune.split(",")))
    g.set_persistent(args.persistent)
    nm = max(int(x) for x in args.groups.split(","))
    ews = make_weights(ea, nm, inDim, outDim, 1234, dev, keep_core=False, q4=bool(args.q4))
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    v = torch.randn(inDim, generator=gen, device=dev)
    outs = [torch.zeros(outDim, device=dev) for _ in ews]
    khz = 100000.0
    dump = {}
    for n in (int(x) for x in args.groups.split(",")):
        calls = [(v, ew, None, o, args.effort) for ew, o in zip(ews[:n], outs[:n])]
        for _ in range(3):
            ea.bucketMulGroup(calls)
        g.eval()
        g.enable_kernel_timing(3)
        if args.replay:                                  # sustained: the launch traced is the last of `replay` back-to-back graph replays
            ea.bucketMulGroup(calls)
            g.eval()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                ea.bucketMulGroup(calls)
            g._bind_stream()
            for _ in range(args.replay):
                gr.replay()
            g.eval()
        else:
            ea.bucketMulGroup(calls)
            g.eval()
        rec = np.array(g.debug_trace(4096), dtype=np.uint64)
        g.enable_kernel_timing(0)
        rec = rec[rec[:, 2] != 0]
        jobs = rec[(rec[:, 0] >> np.uint64(63)) != 0]
        items = rec[(rec[:, 0] >> np.uint64(63)) == 0]
        t0 = float(rec[:, 2].min())
        us = lambda x: (np.asarray(x, dtype=np.float64) - t0) / khz * 1e3
        ph = us(items[:, 2:8])                              # start, staged, cutoff, selected, streamed, handed over
        xcc = (items[:, 1] & np.uint64(0xF)).astype(int)
        nkept = ((items[:, 1] >> np.uint64(8)) & np.uint64(0xFFFF)).astype(int)
        tile = ((items[:, 1] >> np.uint64(24)) & np.uint64(0xFF)).astype(int)
        cols = outDim // (32 if args.q4 else 16)
        ntile = int(tile.max()) + 1
        tw = 64
        while tw * ntile < cols:
            tw *= 2
        nbytes = nkept * np.minimum(tw, cols - tile * tw) * 2.0
        wg = ((items[:, 0] >> np.uint64(32)) & np.uint64(0x7FFFFFFF)).astype(int)
        print(f"== {args.shape} effort {args.effort} q4={args.q4} group {n}: {len(items)} items, {len(jobs)} cutoff jobs, {len(set(wg))} workgroups; span {ph[:, 5].max():.1f} us")
        if len(jobs):
            j = us(jobs[:, 2:4])
            print(f"   cutoff jobs: start {j[:, 0].min():.1f}..{j[:, 0].max():.1f}, end {j[:, 1].min():.1f}..{j[:, 1].max():.1f}")
        names = ["start", "staged", "cutoff", "selected", "streamed", "handed"]
        for i, nme in enumerate(names):
            q = np.percentile(ph[:, i], [0, 10, 50, 90, 100])
            print(f"   {nme:9s} min {q[0]:7.1f}  p10 {q[1]:7.1f}  med {q[2]:7.1f}  p90 {q[3]:7.1f}  max {q[4]:7.1f}")
        d = np.diff(ph, axis=1)
        for i, nme in enumerate(["stage", "cutoff", "select", "stream", "handoff"]):
            q = np.percentile(d[:, i], [0, 50, 100])
            print(f"   d.{nme:8s} min {q[0]:6.1f}  med {q[1]:6.1f}  max {q[2]:6.1f}  sum/items {d[:, i].mean():6.2f}")
        end = ph[:, 5].max()
        grid = np.arange(0, end + 5, 5.0)
        streaming = [(int(((ph[:, 3] <= t) & (ph[:, 4] > t)).sum())) for t in grid]
        alive = [(int(((ph[:, 0] <= t) & (ph[:, 5] > t)).sum())) for t in grid]
        dur = np.maximum(ph[:, 4] - ph[:, 3], 0.01)
        # piecewise-even rate: each quarter of an item's rows between its progress stamps (falls back to the whole phase)
        qs = us(items[:, 8:11])
        knots = np.concatenate([ph[:, 3:4], qs, ph[:, 4:5]], axis=1)
        okq = (items[:, 8:11] != 0).all(axis=1) & (np.diff(knots, axis=1) > 0).all(axis=1)
        rate = []
        for t in grid:
            r = 0.0
            for q in range(4):
                m = okq & (knots[:, q] <= t) & (knots[:, q + 1] > t)
                r += float((nbytes[m] / 4 / (knots[m, q + 1] - knots[m, q])).sum())
            m = (~okq) & (ph[:, 3] <= t) & (ph[:, 4] > t)
            r += float((nbytes[m] / dur[m]).sum())
            rate.append(r / 1e6)
        qd = np.diff(knots[okq], axis=1)
        if len(qd):
            print("   quarter durations of the streaming phase (us), median:", np.median(qd, axis=0).round(1), " items with stamps:", int(okq.sum()))
        print(f"   bytes {nbytes.sum() / 1e6:.1f} MB, per-item stream rate GB/s: med {np.median(nbytes / dur) / 1e3:.1f}; whole-launch {nbytes.sum() / end / 1e6:.2f} TB/s")
        print("   t(us)     " + " ".join(f"{int(t):4d}" for t in grid))
        print("   TB/s(est) " + " ".join(f"{r:4.1f}" for r in rate))
        print("   streaming " + " ".join(f"{s:4d}" for s in streaming))
        print("   in item   " + " ".join(f"{s:4d}" for s in alive))
        # items per workgroup and first / second item timing
        order = np.argsort(ph[:, 0])
        per = {}
        for k in order:
            per.setdefault(wg[k], []).append(k)
        if len(jobs):
            jwg = ((jobs[:, 0] >> np.uint64(32)) & np.uint64(0x7FFFFFFF)).astype(int)
            print("   job workgroups:", sorted(jwg.tolist())[:40])
            print("   next item after job / total:", jobs[:8, 4].tolist(), jobs[:8, 5].tolist())
            print("   items run by job workgroups:", [len(per.get(w, [])) for w in sorted(jwg.tolist())][:40])
            print("   workgroups without items:", sorted(set(range(int(wg.max()) + 1)) - set(wg.tolist()))[:40])
        cnt = np.bincount([len(x) for x in per.values()])
        print("   items per workgroup histogram:", {i: int(c) for i, c in enumerate(cnt) if c})
        for x in range(8):
            m = xcc == x
            if m.any():
                print(f"   xcc {x}: {int(m.sum())} items, last end {ph[m, 5].max():.1f}, stream sum {d[m, 3].sum():.0f}")
        dump[str(n)] = {"knots": knots.tolist(), "nbytes": nbytes.tolist(), "ph_us": ph.tolist(), "xcc": xcc.tolist(), "wg": wg.tolist(), "item": (items[:, 0] & np.uint64(0xFFFFFFFF)).astype(int).tolist(),
                        "hwid": ((items[:, 1] >> np.uint64(32))).astype(int).tolist()}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(dump, f)


if __name__ == "__main__":
    main()
