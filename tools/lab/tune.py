#!/usr/bin/env python
"""Sweep the multiply kernel's launch geometry (waves/workgroup, elements/lane, row slices) on one GPU.

    python tools/lab/tune.py [--shape 4096x11008] [--efforts 0.25,1.0] [--mats 16]

Prints one line per configuration: per-call time from hipGraph replays over rotating matrices, and the
multiply kernel's own duration from the device wall clock.  Used to pick the heuristics in api.hip.
"""
import argparse
import json
import os
os.environ.setdefault("EFFORT_HIP_LIB", "lab")     # stamps / traces live in libeffort_hip_lab.so (the shipped kernels carry none)
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="4096x11008")
    ap.add_argument("--efforts", default="0.25,1.0")
    ap.add_argument("--mats", type=int, default=16)
    ap.add_argument("--split", type=int, default=0)
    ap.add_argument("--q4", type=int, default=0)
    ap.add_argument("--persistent", type=int, default=-1, help="workgroups per CU of persistent group launches (0 = plain grid)")
    ap.add_argument("--groups", default="", help="comma list: group sizes (calls per launch, effort_bucketmul_group) to try")
    ap.add_argument("--streams", default="1", help="comma list: numbers of concurrent streams/contexts to try")
    ap.add_argument("--configs", default="16,1,0;16,1,24;16,1,32;16,1,64;16,2,0;16,2,32;8,1,0;8,1,48;8,1,96;8,2,0;8,2,48;8,4,0;8,4,32;4,1,0;4,2,0;4,4,0")
    args = ap.parse_args()
    inDim, outDim = (int(x) for x in args.shape.split("x"))
    import effort_amd as ea
    from bench import make_weights, mul_kernel_bytes
    dev = torch.device("cuda", 0)
    g = ea.gpu(0)
    ews = make_weights(ea, args.mats, inDim, outDim, 1234, dev, keep_core=False, q4=bool(args.q4))
    mulfn = ea.bucketMulQ4 if args.q4 else ea.bucketMul
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    v = torch.randn(inDim, generator=gen, device=dev)
    outs = [torch.zeros(outDim, device=dev) for _ in ews]
    rows = []
    for effort in (float(x) for x in args.efforts.split(",")):
        for cfg in args.configs.split(";"):
            W, E, S = (int(x) for x in cfg.split(","))
            try:
                g.set_tuning(W, E, S)
                g.set_split_cutoff(bool(args.split))
                g.set_persistent(args.persistent)
                g.enable_kernel_timing(2)
                for ew, o in zip(ews, outs):
                    mulfn(v, ew, None, o, effort)
                g.eval()
                D = g.last_dispatch_count()
                over = {}
                for K in [int(x) for x in args.streams.split(",") if int(x) > 1]:
                    ctxs = [ea.Gpu(0) for _ in range(K)]
                    sts = [torch.cuda.Stream() for _ in range(K)]
                    for c in ctxs:
                        c.set_tuning(W, E, S)
                        c.set_split_cutoff(bool(args.split))
                    go = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(go):
                        s0 = torch.cuda.current_stream()
                        for st in sts:
                            st.wait_stream(s0)
                        for i, (ew, o) in enumerate(zip(ews, outs)):
                            with torch.cuda.stream(sts[i % K]):
                                mulfn(v, ew, None, o, effort, gpu=ctxs[i % K])
                        for st in sts:
                            s0.wait_stream(st)
                    for _ in range(5):
                        go.replay()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(30):
                        go.replay()
                    torch.cuda.synchronize()
                    over[K] = round((time.perf_counter() - t0) / 30 / len(ews) * 1e6, 2)
                    if not args.split:
                        ctxs[0].enable_kernel_timing(2)
                        go2 = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(go2):
                            s0 = torch.cuda.current_stream()
                            for st in sts:
                                st.wait_stream(s0)
                            for i, (ew, o) in enumerate(zip(ews, outs)):
                                with torch.cuda.stream(sts[i % K]):
                                    mulfn(v, ew, None, o, effort, gpu=ctxs[i % K])
                            for st in sts:
                                s0.wait_stream(st)
                        for _ in range(5):
                            go2.replay()
                        torch.cuda.synchronize()
                        stx = ctxs[0].debug_stamps()
                        over[f"{K}_wg0_phases"] = [round((stx[9 + i] - stx[8 + i]) / 100.0, 2) for i in range(5)] + [round((stx[i] - stx[8]) / 100.0, 2) for i in (17, 18, 19, 16)]
                        del go2
                    del go
                    for c in ctxs:
                        c.close()
                for G in [int(x) for x in args.groups.split(",") if x]:
                    g.enable_kernel_timing(0)            # the stamps cost contended atomics: throughput is measured without
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg):
                        for i in range(0, len(ews), G):
                            ea.bucketMulGroup([(v, ew, None, o, effort) for ew, o in zip(ews[i:i + G], outs[i:i + G])])
                    for _ in range(5):
                        gg.replay()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(30):
                        gg.replay()
                    torch.cuda.synchronize()
                    over[f"g{G}_nostamp"] = round((time.perf_counter() - t0) / 30 / len(ews) * 1e6, 2)
                    del gg
                    g.enable_kernel_timing(2)
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg):
                        for i in range(0, len(ews), G):
                            ea.bucketMulGroup([(v, ew, None, o, effort) for ew, o in zip(ews[i:i + G], outs[i:i + G])])
                    g._bind_stream()
                    for _ in range(5):
                        gg.replay()
                    g.kernel_clock()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(30):
                        gg.replay()
                    torch.cuda.synchronize()
                    over[f"g{G}"] = round((time.perf_counter() - t0) / 30 / len(ews) * 1e6, 2)
                    over[f"g{G}_kernel_us"] = round(g.kernel_clock()["mul_us"], 2)
                    stg = g.debug_stamps()
                    over[f"g{G}_wg0"] = [round((stg[9 + i] - stg[8 + i]) / 100.0, 2) for i in range(5)]
                    over[f"g{G}_ramp_us"] = round(stg[19] / (35 * ((len(ews) + G - 1) // G)) / 100.0, 2)   # 5 + 30 replays since the stamps were reset
                    over[f"g{G}_wgmax,streammax,reduce_us"] = [round(stg[20] / 100.0, 2), round(stg[21] / 100.0, 2), round((stg[16] - stg[15]) / 100.0, 2)]
                    over[f"g{G}_wgmean"] = [round(stg[24 + i] / max(1, stg[29]) / 100.0, 2) for i in range(5)]
                    del gg
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    for ew, o in zip(ews, outs):
                        mulfn(v, ew, None, o, effort)
                g._bind_stream()
                for _ in range(5):
                    gr.replay()
                g.kernel_clock()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    gr.replay()
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / 30 / len(ews)
                clk = g.kernel_clock()
                st = g.debug_stamps()
                g.enable_kernel_timing(1)            # event-to-event intervals (include the launch gap before each kernel)
                torch.cuda._sleep(10_000_000)
                for _ in range(4):
                    for ew, o in zip(ews, outs):
                        mulfn(v, ew, None, o, effort)
                ev = g.kernel_timing()
                kb = mul_kernel_bytes(D, inDim, outDim)
                row = {"shape": args.shape, "effort": effort, "W": W, "E": E, "S": S, "D": D, "call_us": round(t * 1e6, 2),
                       "mul_us": round(clk["mul_us"], 2), "mul_GBps": round(kb / clk["mul_us"] / 1e3, 0),
                       "eff_GBps": round(2 * inDim * outDim / t / 1e9, 0), "call_us_by_streams": over,
                       "ev_mul_us": round(ev["mul_us"], 2),
                       "item0_phases_us(stage,cutoff,select,stream,handoff)": [round((st[9 + i] - st[8 + i]) / 100.0, 2) for i in range(5)],
                       "item0_rows": st[14], "reduce_us(tile0)": round((st[16] - st[15]) / 100.0, 2),
                       "cutoff(setup,table)us": [round((st[1] - st[0]) / 100.0, 2), round((st[2] - st[1]) / 100.0, 2)],
                       "cutoff_loops,ballot_passes": [st[5] // 1000, st[5] % 1000]}
            except Exception as ex:
                row = {"shape": args.shape, "effort": effort, "W": W, "E": E, "S": S, "error": repr(ex)[:100]}
            rows.append(row)
            print(json.dumps(row), flush=True)
    g.set_tuning(0, 0, 0)
    g.enable_kernel_timing(0)


if __name__ == "__main__":
    main()
