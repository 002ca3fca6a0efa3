#!/usr/bin/env python
"""What a decoder layer's multiplies cost as the decode loop issues them -- four dependent launches (wo -> w1|w3 -> w2 ->
wq|wk|wv of the next layer, glue folded in) -- against each launch alone and against the same seven calls as ONE independent
group launch (the bound: no boundaries, no ramps, no dependency stalls).  (The one-chain-launch variant of round 4: branch chain-launch.)

    python tools/layer_probe.py [--effort 0.25] [--layers 8]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--effort", type=float, default=0.25)
    ap.add_argument("--persistent", type=int, default=-1, help="workgroups per CU of persistent launches (-1: heuristic = 2)")
    ap.add_argument("--layers", type=int, default=8, help="distinct weight sets rotated through (cache honesty)")
    ap.add_argument("--reps", type=int, default=200)
    args = ap.parse_args()
    import effort_amd as ea
    from effort_amd.decode import MistralConfig, Model
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    cfg = MistralConfig(numLayers=args.layers)
    model = Model.random(cfg, seed=3, keep_cores=False)
    g = ea.gpu(0)
    g.set_persistent(args.persistent)
    e = args.effort
    f = lambda n: torch.randn(n, device=dev)                                    # noqa: E731
    h, attn, x1, x3, xq, xk, xv = f(4096), f(4096), f(14336), f(14336), f(4096), f(1024), f(1024)
    h0 = h.clone()

    def launches(L, Lnext, which):
        if "wo" in which:
            ea.bucketMulGroup([(attn, L.wo, None, h, e, {"resid": h})])
        if "w13" in which:
            ea.bucketMulGroup([(h, L.w1, None, x1, e, {"norm": L.ffnNorm}), (h, L.w3, None, x3, e, {"norm": L.ffnNorm})])
        if "w2" in which:
            ea.bucketMulGroup([(x1, L.w2, None, h, e, {"gate": x3, "resid": h})])
        if "qkv" in which:
            ea.bucketMulGroup([(h, Lnext.wq, None, xq, e, {"norm": Lnext.attnNorm}), (h, Lnext.wk, None, xk, e, {"norm": Lnext.attnNorm}),
                               (h, Lnext.wv, None, xv, e, {"norm": Lnext.attnNorm})])

    def seven(L, Lnext):              # the same seven multiplies as one launch of independent calls (plain inputs: no glue)
        ea.bucketMulGroup([(attn, L.wo, None, h, e), (h0, L.w1, None, x1, e), (h0, L.w3, None, x3, e), (x1, L.w2, None, attn, e),
                           (h0, Lnext.wq, None, xq, e), (h0, Lnext.wk, None, xk, e), (h0, Lnext.wv, None, xv, e)])

    def timed(fn):
        def run():
            for n in range(args.layers):
                fn(model.layers[n], model.layers[(n + 1) % args.layers])
            h.copy_(h0)
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
        for _ in range(args.reps):
            gr.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.reps / args.layers * 1e6

    res = {"effort": e}
    for name, which in (("wo", ("wo",)), ("w13", ("w13",)), ("w2", ("w2",)), ("qkv", ("qkv",)), ("four_dependent_launches", ("wo", "w13", "w2", "qkv"))):
        res[name + "_us"] = round(timed(lambda L, Ln, w=which: launches(L, Ln, w)), 2)
    res["seven_independent_calls_one_launch_us"] = round(timed(seven), 2)
    res["sum_of_lone_us"] = round(res["wo_us"] + res["w13_us"] + res["w2_us"] + res["qkv_us"], 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
