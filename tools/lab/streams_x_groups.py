#!/usr/bin/env python
"""Throughput of the bench step for (calls per launch) x (HIP streams the launches are spread over)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import effort_amd as ea  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    effort = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
    ews = bench.make_weights(ea, 32, bench.IN_DIM, bench.OUT_DIM, 1234, dev, keep_core=False)
    gen = torch.Generator(device=dev)
    gen.manual_seed(42)
    v = torch.randn(bench.IN_DIM, generator=gen, device=dev)
    outs = [torch.zeros(bench.OUT_DIM, device=dev) for _ in ews]
    items = list(zip(ews, outs))
    per = int(os.environ.get("PERSISTENT", "-1"))
    for K in (1, 2, 3):
        st = bench.Step(ea, 0, K)
        for c in st.ctxs:
            c.set_persistent(per)
            c.set_tuning(*(int(x) for x in os.environ.get("TUNE", "0,0,0").split(",")))
        row = {"streams": K}
        for G in (1, 2, 4, 8, 16, 32):
            fn = lambda ctx, ch: ea.bucketMulGroup([(v, ew, None, o, effort) for ew, o in ch], gpu=ctx)
            g = st.capture(fn, bench.chunked(items, G))
            row[f"g{G}_us_per_call"] = round(bench.time_replays(g, 40, 10) / 32 * 1e6, 2)
            del g
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
