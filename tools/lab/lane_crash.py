#!/usr/bin/env python
"""Repro hunt: segfault in hipGraph replay of a multi-lane job after another multi-lane context was destroyed."""
import faulthandler
import os
import sys

import torch

if not os.environ.get('LD_PRELOAD'):
    faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import effort_amd as ea  # noqa: E402
from bench import LaneJob, make_weights  # noqa: E402

mode = sys.argv[1]
dev = torch.device("cuda", 0)
inDim, outDim, n = 4096, 4096, 8
ews = make_weights(ea, n, inDim, outDim, 1, dev, keep_core=False)
v = torch.randn(inDim, device=dev)
sets = [torch.zeros((n, outDim), device=dev) for _ in range(4)]


def step(ctx, slot):
    ea.bucketMulGroup([(v, ew, None, sets[slot][k], 0.25) for k, ew in enumerate(ews)], gpu=ctx)


def run(lanes, steps=48):
    jb = LaneJob(ea, 0, lanes)
    g = jb.capture(step, steps)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print(f"mode {mode}: {lanes} lanes, {steps} steps ok", flush=True)
    return jb, g


if mode == "destroy":
    jb, g = run(2)
    del g, jb
    run(3)
elif mode == "keep":
    a = run(2)
    b = run(3)
elif mode == "three":
    run(3)
elif mode == "three20":
    run(3, 20)
elif mode == "four48":
    run(4, 48)
elif mode.startswith("bench"):
    inDim, outDim, n = 4096, 11008, 32
    nsets = 4
    ew_sets = [make_weights(ea, n, inDim, outDim, 1234 + 32 * k, dev, keep_core=False) for k in range(nsets)]
    sets = [torch.zeros((n, outDim), device=dev) for _ in range(4)]

    def mk(ws):
        def st(ctx, slot):
            ea.bucketMulGroup([(v, ew, None, sets[slot][k], 0.25) for k, ew in enumerate(ws[slot % len(ws)])], gpu=ctx)
        return st
    keep = []
    if "job" in mode:
        job = LaneJob(ea, 0, 4)
        gj = job.capture(mk(ew_sets), 20)
        gj.replay()
        if "sync" in mode:
            torch.cuda.synchronize()
        keep.append((job, gj))
    for lanes in (1, 2, 3, 4):
        jb = LaneJob(ea, 0, lanes)
        g = jb.capture(mk(ew_sets[:lanes]), 48)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        print(f"mode {mode}: {lanes} lanes ok", flush=True)
        if "keep" in mode:
            keep.append((jb, g))
        del g
print("done", mode, flush=True)
