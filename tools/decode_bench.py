#!/usr/bin/env python
"""BASELINE.json configs[4]: end-to-end greedy decode of a Mistral-7B-shaped model (random-init weights: no
checkpoints here) through the decode loop of effort_amd/decode.py, effort 25 % vs 100 % vs dense: tokens/s and KL
divergence of the logits against the dense path (teacher-forced on the dense run's tokens).

    python tools/decode_bench.py [--layers 32] [--tokens 64] [--efforts 1.0,0.5,0.25]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from effort_amd.decode import Decoder, MistralConfig, Model, kl_divergence  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--efforts", default="1.0,0.5,0.25")
    ap.add_argument("--model-dir", default=None, help="a bucketed model on disk (e.g. kolinko/mistral-buckets: buckets-FP16.safetensors.index.json "
                                                      "+ shards) instead of random-init weights")
    ap.add_argument("--model-name", default="buckets-FP16")
    ap.add_argument("--percent-load", type=int, default=16)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    cfg = MistralConfig(numLayers=a.layers)
    t0 = time.time()
    if a.model_dir:
        from effort_amd.bucketfile import TensorLoader
        model = Model.load(TensorLoader(a.model_dir, a.model_name), cfg, percentLoad=a.percent_load)
    else:
        model = Model.random(cfg, seed=1)
    torch.cuda.synchronize()
    print(f"model: {a.layers} layers, 7 bucketized matrices each, {'loaded' if a.model_dir else 'built + converted'} in {time.time() - t0:.1f} s", file=sys.stderr)
    dec = Decoder(model, maxTokens=max(64, a.tokens + 8))
    prompt = [1, 733, 16289, 28793, 22557]
    dec.g.set_dense_backend(True)                  # dense through rocBLAS' hssgemv, then through the package's own GEMV (the default)
    _, dt_r, _ = dec.run(prompt, a.tokens, dense=True)
    dec.g.set_dense_backend(False)
    dec._graphs.clear()
    ids_d, dt_d, lg_d = dec.run(prompt, a.tokens, dense=True, collect_logits=True)
    forced = prompt + ids_d[len(prompt) - 1:-1]
    out = {"model": f"Mistral-7B shapes, {a.layers} layers, " + (f"loaded from {a.model_dir}" if a.model_dir else "random init"), "tokens": a.tokens, "prompt_tokens": len(prompt),
           "dense_rocblas": {"ms_per_token": round(dt_r * 1e3, 3), "tokens_per_s": round(1 / dt_r, 1)},
           "dense_hip_kernel": {"ms_per_token": round(dt_d * 1e3, 3), "tokens_per_s": round(1 / dt_d, 1)}, "effort": {}}
    for e in (float(x) for x in a.efforts.split(",")):
        ids_e, dt_e, _ = dec.run(prompt, a.tokens, effort=e)                      # free-running greedy: the speed
        _, _, lg_e = dec.run(forced, a.tokens, effort=e, forced=True, collect_logits=True)
        agree = sum(int(x == y) for x, y in zip(lg_e.argmax(-1).tolist(), lg_d.argmax(-1).tolist())) / a.tokens
        out["effort"][str(e)] = {"ms_per_token": round(dt_e * 1e3, 3), "tokens_per_s": round(1 / dt_e, 1),
                                 "speedup_vs_dense_rocblas": round(dt_r / dt_e, 3), "speedup_vs_dense_hip_kernel": round(dt_d / dt_e, 3),
                                 "kl_vs_dense": round(kl_divergence(lg_d, lg_e), 5),
                                 "top1_agreement_vs_dense": round(agree, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
