#!/usr/bin/env python
"""A/B of decode-loop speed under library knobs, one model build: tokens/s dense (own GEMV) and effort runs with the
row prefetch of lone calls on / off.

    python tools/decode_ab.py [--layers 32] [--tokens 48] [--efforts 0.25,1.0]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effort_amd.decode import Decoder, MistralConfig, Model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=48)
    ap.add_argument("--efforts", default="0.25,1.0")
    ap.add_argument("--fused-glue", type=int, default=0)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    model = Model.random(MistralConfig(numLayers=a.layers), seed=1)
    dec = Decoder(model, maxTokens=max(64, a.tokens + 8), fused_glue=bool(a.fused_glue))
    prompt = [1, 733, 16289, 28793, 22557]
    out = {}
    dec.g.set_dense_backend(False)
    _, dt_d, _ = dec.run(prompt, a.tokens, dense=True)
    out["dense_hip_kernel_tokens_per_s"] = round(1 / dt_d, 1)
    for e in (float(x) for x in a.efforts.split(",")):
        for pf in (0, 1, 0, 1):
            if hasattr(dec.g, "set_prefetch"):
                dec.g.set_prefetch(bool(pf))
            dec._graphs.clear()
            _, dt_e, _ = dec.run(prompt, a.tokens, effort=e)
            out.setdefault(f"effort {e}", []).append({"prefetch": pf, "tokens_per_s": round(1 / dt_e, 1), "vs_dense": round(dt_d / dt_e, 3)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
