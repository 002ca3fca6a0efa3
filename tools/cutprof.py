#!/usr/bin/env python
"""Phase stamps of item 0 of a lone call (device wall clock, 10 ns): where findCutoff32's time goes."""
import os, sys
os.environ.setdefault("EFFORT_HIP_LIB", "lab")     # stamps / traces live in libeffort_hip_lab.so (the shipped kernels carry none)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import effort_amd as ea
from bench import make_weights
dev = torch.device("cuda", 0)
g = ea.gpu(0)
for shape in ((4096, 11008), (4096, 4096)):
    ews = make_weights(ea, 4, shape[0], shape[1], 1234, dev, keep_core=False)
    gen = torch.Generator(device=dev); gen.manual_seed(42)
    v = torch.randn(shape[0], generator=gen, device=dev)
    out = torch.zeros(shape[1], device=dev)
    for effort in (0.25, 0.5, 1.0):
        for rep in range(3):
            g.enable_kernel_timing(2)
            ea.bucketMul(v, ews[rep], None, out, effort)
            g.eval()
            st = g.debug_stamps()
            g.enable_kernel_timing(0)
        c = st[0:8]; it = st[8:16]
        us = lambda a, b: (b - a) / 100.0
        ghz = (c[7] - c[6]) / max(1.0, (c[3] - c[0]) * 10.0)
        print(f"{shape} effort {effort}: shader clock during the cutoff {ghz:.2f} GHz; cutoff total {us(c[0], c[3]):.2f} us = minmax+ballot {us(c[0], c[1]):.2f} + table {us(c[1], c[2]):.2f} + bisection {us(c[2], c[3]):.2f}; loops {c[5] // 1000} passes {c[5] % 1000}; "
              f"item0: stage {us(it[0], it[1]):.2f} cutoff {us(it[1], it[2]):.2f} select {us(it[2], it[3]):.2f} stream {us(it[3], it[4]):.2f} handoff {us(it[4], it[5]):.2f} rows {it[6]}")
