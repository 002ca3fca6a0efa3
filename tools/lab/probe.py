#!/usr/bin/env python
"""Ablation probe for the multiply kernel (profiling aid): run with EFFORT_DEBUG=<mask> in the environment.

    EFFORT_DEBUG=0 python tools/lab/probe.py ; EFFORT_DEBUG=1 python tools/lab/probe.py ; ...

Prints the kernel's device-clock duration per launch geometry, plus (mask 0 only) a calibration read of
the same bytes with a plain torch reduction so the box's achievable HBM rate is on the same page.
"""
import json
import os
os.environ.setdefault("EFFORT_HIP_LIB", "lab")     # stamps / traces live in libeffort_hip_lab.so (the shipped kernels carry none)
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import effort_amd as ea
    from bench import make_weights, mul_kernel_bytes
    shape = os.environ.get("PROBE_SHAPE", "4096x11008")
    inDim, outDim = (int(x) for x in shape.split("x"))
    mats = int(os.environ.get("PROBE_MATS", "12"))
    dev = torch.device("cuda", 0)
    g = ea.gpu(0)
    ews = make_weights(ea, mats, inDim, outDim, 1234, dev, keep_core=False)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    v = torch.randn(inDim, generator=gen, device=dev)
    outs = [torch.zeros(outDim, device=dev) for _ in ews]
    dbg = int(os.environ.get("EFFORT_DEBUG", "0"))
    if dbg == 0:
        big = torch.cat([e.buckets.view(-1) for e in ews]).view(torch.int32)
        for _ in range(2):
            big.sum()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            big.sum()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(json.dumps({"calibration": "torch int32 sum", "GB": round(big.numel() * 4 / 1e9, 2), "GBps": round(big.numel() * 4 / dt / 1e9, 0)}), flush=True)
        del big
    cfgs = os.environ.get("PROBE_CFGS", "16,1,0;16,1,64;8,1,96;4,1,192;4,1,384").split(";")
    for effort in (0.25, 1.0):
        for cfg in cfgs:
            W, E, S = (int(x) for x in cfg.split(","))
            g.set_tuning(W, E, S)
            g.enable_kernel_timing(2)
            for r in range(3):
                for ew, o in zip(ews, outs):
                    ea.bucketMul(v, ew, None, o, effort)
                if r == 0:
                    g.kernel_clock()
            g.eval()
            D = g.last_dispatch_count()
            clk = g.kernel_clock()
            kb = mul_kernel_bytes(D, inDim, outDim)
            print(json.dumps({"debug": dbg, "effort": effort, "W": W, "E": E, "S": S, "D": D, "mul_us": round(clk["mul_us"], 2),
                              "GBps": round(kb / clk["mul_us"] / 1e3, 0)}), flush=True)
    g.set_tuning(0, 0, 0)


if __name__ == "__main__":
    main()
