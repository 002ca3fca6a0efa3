#!/usr/bin/env python
"""A/B of decode-loop speed under library knobs, one model build: tokens/s dense (own GEMV) and effort runs with the
row prefetch of lone calls on / off.

    python tools/lab/decode_ab.py [--layers 32] [--tokens 48] [--efforts 0.25,1.0]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from effort_amd.decode import Decoder, MistralConfig, Model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=48)
    ap.add_argument("--efforts", default="0.25,1.0")
    ap.add_argument("--fused-glue", default="", help="comma list of norm,gate,resid (or 1 = all) folded into the multiplies")
    ap.add_argument("--tunes", default="0,0,0", help="semicolon list of set_tuning triples (waves,elems,slices) to time, e.g. 0,0,0;8,1,0")
    ap.add_argument("--split", type=int, default=0, help="also time every knob set with the cutoffs in a kernel of their own")
    a = ap.parse_args()
    torch.cuda.set_device(0)
    model = Model.random(MistralConfig(numLayers=a.layers), seed=1)
    fg = True if a.fused_glue == "1" else tuple(x for x in a.fused_glue.split(",") if x)
    dec = Decoder(model, maxTokens=max(64, a.tokens + 8), fused_glue=fg)
    prompt = [1, 733, 16289, 28793, 22557]
    out = {}
    dec.g.set_dense_backend(False)
    _, dt_d, _ = dec.run(prompt, a.tokens, dense=True)
    out["dense_hip_kernel_tokens_per_s"] = round(1 / dt_d, 1)
    knobs = [{"tune": t} for t in a.tunes.split(";")]
    for e in (float(x) for x in a.efforts.split(",")):
        for kn in knobs:
            for k in ("EFFORT_X_NARROW", "EFFORT_X_FULL", "EFFORT_X_GT"):
                os.environ.pop(k, None)
            dec.g.set_tuning(*(int(x) for x in kn["tune"].split(",")))
            for split in ((0, 1) if a.split else (0,)):
                dec.g.set_split_cutoff(bool(split))
                dec._graphs.clear()
                _, dt_e, _ = dec.run(prompt, a.tokens, effort=e)
                out.setdefault(f"effort {e}", []).append({"knobs": kn, "split_cutoff": split, "tokens_per_s": round(1 / dt_e, 1), "vs_dense": round(dt_d / dt_e, 3)})
        dec.g.set_split_cutoff(False)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
