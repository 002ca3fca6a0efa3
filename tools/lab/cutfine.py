#!/usr/bin/env python
"""Shader-clock stamps inside findCutoff32 of item 0 of a lone call (a -DEFFORT_CUT_FINE lab build: tools/build_variant_all.sh fine
"-DEFFORT_CUT_FINE"; EFFORT_HIP_LIB=build/variants/fine.so python tools/lab/cutfine.py)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import effort_amd as ea
from bench import make_weights
dev = torch.device("cuda", 0)
g = ea.gpu(0)
names = ["values+adds", "dpp min/max", "barrier A", "range read", "table read+sum", "T(k)", "rounds", "tail", "exchange"]
for shape in ((4096, 11008),):
    ews = make_weights(ea, 4, shape[0], shape[1], 1234, dev, keep_core=False)
    gen = torch.Generator(device=dev); gen.manual_seed(42)
    v = torch.randn(shape[0], generator=gen, device=dev)
    out = torch.zeros(shape[1], device=dev)
    for effort in (0.25, 0.5, 1.0):
        for rep in range(4):
            g.enable_kernel_timing(2)
            ea.bucketMul(v, ews[rep], None, out, effort)
            g.eval()
            st = g.debug_stamps()
            g.enable_kernel_timing(0)
        c = st[0:8]; f = st[8:18]
        seq = [c[6]] + [x for x in f[:9]]
        d = [seq[i + 1] - seq[i] for i in range(9)]
        print(f"{shape} effort {effort}: total {seq[-1] - seq[0]} cycles; " + "; ".join(f"{n} {x}" for n, x in zip(names, d)) + f"; loops {c[5] // 1000}")
