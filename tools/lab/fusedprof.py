#!/usr/bin/env python
"""Phase stamps of item 0 of a lone call, plain against fused prologues / epilogue (tools only)."""
import os, sys
os.environ.setdefault("EFFORT_HIP_LIB", "lab")     # stamps / traces live in libeffort_hip_lab.so (the shipped kernels carry none)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import effort_amd as ea
from bench import make_weights
dev = torch.device("cuda", 0)
g = ea.gpu(0)
for shape, kinds in (((14336, 4096), ("", "gate", "resid")), ((4096, 4096), ("", "norm", "resid"))):
    ews = make_weights(ea, 4, shape[0], shape[1], 1234, dev, keep_core=False)
    gen = torch.Generator(device=dev); gen.manual_seed(42)
    v = torch.randn(shape[0], generator=gen, device=dev)
    x3 = torch.randn(shape[0], generator=gen, device=dev)
    wn = (1 + 0.1 * torch.randn(shape[0], generator=gen, device=dev)).to(torch.float16)
    res = torch.randn(shape[1], generator=gen, device=dev)
    out = torch.zeros(shape[1], device=dev)
    for kind in kinds:
        extra = {"gate": {"gate": x3}, "norm": {"norm": wn}, "resid": {"resid": res}}.get(kind)
        for rep in range(4):
            g.enable_kernel_timing(2)
            item = (v, ews[rep], None, out, 0.25) + ((extra,) if extra else ())
            ea.bucketMulGroup([item])
            g.eval()
            st = g.debug_stamps()
            kc = g.kernel_clock()
            g.enable_kernel_timing(0)
        c = st[0:8]; it = st[8:16]
        us = lambda a, b: (b - a) / 100.0
        print(f"{shape} {kind or 'plain':6s}: span {kc['mul_us']:.2f}; cutoff total {us(c[0], c[3]):.2f}; item0: stage {us(it[0], it[1]):.2f} cutoff {us(it[1], it[2]):.2f} select {us(it[2], it[3]):.2f} "
              f"stream {us(it[3], it[4]):.2f} handoff {us(it[4], it[5]):.2f} reduce(tile 0) {us(st[15], st[16]):.2f} rows {it[6]}")
