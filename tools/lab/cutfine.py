#!/usr/bin/env python
"""Shader-clock stamps inside findCutoff32 of item 0 of a lone call (a -DEFFORT_CUT_FINE lab build: tools/build_variant_all.sh fine
"-DEFFORT_CUT_FINE"; EFFORT_HIP_LIB=build/variants/fine.so python tools/lab/cutfine.py)."""
import os, sys
os.environ.setdefault("EFFORT_HIP_LIB", "lab")     # stamps / traces live in libeffort_hip_lab.so (the shipped kernels carry none)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import effort_amd as ea
from bench import make_weights
dev = torch.device("cuda", 0)
g = ea.gpu(0)
names = ["entry stamp", "values+adds", "dpp min/max + barrier A", "range read", "table read+sum", "T(k)", "rounds", "tail", "exchange"]
for shape in ((4096, 11008),):
    ews = make_weights(ea, 4, shape[0], shape[1], 1234, dev, keep_core=False)
    gen = torch.Generator(device=dev); gen.manual_seed(42)
    v = torch.randn(shape[0], generator=gen, device=dev)
    out = torch.zeros(shape[1], device=dev)
    for effort in (0.25, 0.5, 1.0):
        # as the decode loop and the bench issue it: lone calls on rotating matrices, captured into a hipGraph, replayed
        g.enable_kernel_timing(2)
        def run():
            for rep in range(4):
                ea.bucketMul(v, ews[rep], None, out, effort)
        run(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        g._bind_stream()
        for _ in range(5):
            gr.replay()
        torch.cuda.synchronize()
        st = g.debug_stamps()
        g.enable_kernel_timing(0)
        # host32[k] = tstamp[8 + k]: [6] clock at the cutoff's entry; [8..10] kernel entry (after the touch loads), before / after locate_item;
        # [11..14] inside the cutoff: range known, histogram read + summed, order statistics found, rounds done; [18..22] kernel entry (before the
        # touch), item start, stage issued, cut inputs asked, all landed
        c, en, f, pz = st[0:8], st[8:11], st[11:15], st[18:23]
        print(f"{shape} effort {effort} (cycles): kernel entry -> touched {en[0] - pz[0]} -> locate_item {en[1] - pz[0]} -> located {en[2] - pz[0]} -> item start {pz[1] - pz[0]} "
              f"-> stage issued {pz[2] - pz[0]} -> cut inputs asked {pz[3] - pz[0]} -> all landed {pz[4] - pz[0]} -> cutoff entry {c[6] - pz[0]} -> range known {f[0] - pz[0]} "
              f"-> histogram read and summed +{f[1] - f[0]} -> order statistics +{f[2] - f[1]} -> rounds +{f[3] - f[2]} ({c[5] // 1000} rounds)")
