#!/usr/bin/env python
"""Per-item timeline of ONE chain launch (timing mode 3): when the items of every call start, get past the stage wait and the
staged loads, have their cutoff, their selection, their rows streamed and their slab handed over.

    python tools/chain_trace.py [--effort 0.25] [--out gpurun_out/chain_trace.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--effort", type=float, default=0.25)
    ap.add_argument("--slice-mult", type=int, default=1, help="chain launches: row slices per call x this")
    ap.add_argument("--persistent", type=int, default=-1, help="workgroups per CU of persistent launches (-1: heuristic = 2)")
    ap.add_argument("--out", default="gpurun_out/chain_trace.json")
    ap.add_argument("--separate", type=int, default=0, help="1: trace the four launches of their own instead (each one's records)")
    args = ap.parse_args()
    import effort_amd as ea
    from effort_amd.decode import MistralConfig, Model
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = Model.random(MistralConfig(numLayers=2), seed=3, keep_cores=False)
    L, Ln = model.layers
    g = ea.gpu(0)
    g.set_chain_tuning(args.slice_mult)
    g.set_persistent(args.persistent)
    e = args.effort
    f = lambda n: torch.randn(n, device=dev)                                    # noqa: E731
    h, attn, x1, x3, xq, xk, xv = f(4096), f(4096), f(14336), f(14336), f(4096), f(1024), f(1024)
    stages = [[(attn, L.wo, None, h, e, {"resid": h})],
              [(h, L.w1, None, x1, e, {"norm": L.ffnNorm}), (h, L.w3, None, x3, e, {"norm": L.ffnNorm})],
              [(x1, L.w2, None, h, e, {"gate": x3, "resid": h})],
              [(h, Ln.wq, None, xq, e, {"norm": Ln.attnNorm}), (h, Ln.wk, None, xk, e, {"norm": Ln.attnNorm}), (h, Ln.wv, None, xv, e, {"norm": Ln.attnNorm})]]
    names = ["wo", "w1", "w3", "w2", "wq", "wk", "wv"]
    for _ in range(3):
        ea.bucketMulChain(stages)
    g.eval()
    g.enable_kernel_timing(3)
    ea.bucketMulChain(stages)
    g.eval()
    rec = np.array(g.debug_trace(4096), dtype=np.uint64)
    g.enable_kernel_timing(0)
    rec = rec[rec[:, 2] != 0]
    khz = 100000.0
    t0 = float(rec[:, 2].min())
    ph = (rec[:, 2:8].astype(np.float64) - t0) / khz * 1e3            # start, staged, cutoff, selected, streamed, handed over
    ci = ((rec[:, 1] >> np.uint64(4)) & np.uint64(0xF)).astype(int)
    wg = ((rec[:, 0] >> np.uint64(32)) & np.uint64(0x7FFFFFFF)).astype(int)
    print(f"chain at effort {e}: {len(rec)} items on {len(set(wg.tolist()))} workgroups, span {ph[:, 5].max():.1f} us")
    out = {}
    for c, nme in enumerate(names):
        m = ci == c
        if not m.any():
            continue
        q = lambda col: np.percentile(ph[m, col], [0, 50, 100]).round(1).tolist()     # noqa: E731
        d = np.diff(ph[m], axis=1)
        print(f"  {nme:3s} {int(m.sum()):4d} items | start {q(0)} staged {q(1)} cutoff {q(2)} selected {q(3)} streamed {q(4)} handed {q(5)}"
              f" | med d: wait+stage {np.median(d[:, 0]):.1f} cutoff {np.median(d[:, 1]):.1f} select {np.median(d[:, 2]):.1f} stream {np.median(d[:, 3]):.1f} handoff {np.median(d[:, 4]):.1f}")
        out[nme] = {"items": int(m.sum()), "ph_us": ph[m].round(2).tolist()}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(out, open(args.out, "w"))


if __name__ == "__main__":
    main()
