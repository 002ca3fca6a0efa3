#!/usr/bin/env python
"""Whole-model converter: HF Mistral-7B safetensors -> bucketed model (convert.swift:59-127 / q4_convert.py:29-81).

    python tools/convert_model.py <hf_dir> <out_dir> [--q4] [--layers 32]

<hf_dir> holds model.safetensors.index.json + shards as published by mistralai/Mistral-7B-Instruct-v0.2.  FP16 writes
buckets-FP16-*.safetensors (+ index), Q4 writes model-*.safetensors (+ index), with the reference's tensor names, so
the result is what `kolinko/mistral-buckets` ships (loader.swift:286).  Needs the GPU (effort_amd converters).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from effort_amd import bucketfile as bf  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--q4", action="store_true")
    ap.add_argument("--layers", type=int, default=32)
    a = ap.parse_args()
    src = bf.TensorLoader(a.src, "model")
    saver = bf.TensorSaver(a.dst, "model" if a.q4 else "buckets-FP16", pad_total=not a.q4)
    t0 = time.time()
    bf.convertMistral(src, saver, numLayers=a.layers, q4=a.q4, log=lambda *m: print(*m, f"[{time.time() - t0:.0f} s]", flush=True))
    print("saved", saver.save())


if __name__ == "__main__":
    main()
