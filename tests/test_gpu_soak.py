"""Randomized soak of the hot path against the oracle: FP16 and Q4 bundles, lone calls and groups of up to 32, launch geometries,
percentLoad, zeros and heavy tails in the input, launches kept in flight on the context's lanes (effort_set_overlap) -- every call
of every launch: dispatch.size and cutoff bits exact, the product within the bar.  A few seconds by default (it rides in the
`-m gpu` suite); EFFORT_SOAK_SECONDS=300 EFFORT_SOAK_SEED=7 python -m pytest tests/test_gpu_soak.py -m gpu  for a long run."""
import os
import time

import numpy as np
import pytest
import torch

from tests.test_gpu_parity import DEV, close, converted, dev16, devf, ea, gpu_weights  # noqa: F401  (ea: module fixture)
from tests.util import make_v

pytestmark = pytest.mark.gpu


def _q4_bundle(rng, inDim, outDim, n_outliers):
    """A synthetic Q4 bundle (any nibble pattern is a valid bucket word) with a random outlier table of f16 values."""
    rows, cols = inDim * 8, outDim // 32
    buckets = rng.integers(0, 65536, size=(rows, cols), dtype=np.uint16)
    mean = np.abs(rng.normal(0, 0.02, size=rows)).astype(np.float32)
    stats = np.stack([mean, mean], axis=1)
    probes = rng.normal(0, 0.02, size=4096).astype(np.float16)
    ol = np.zeros((n_outliers, 4), np.float32)
    ol[:, 0] = rng.normal(0, 0.3, size=n_outliers).astype(np.float16)
    ol[:, 1] = rng.integers(0, inDim, size=n_outliers)
    ol[:, 2] = rng.integers(0, outDim, size=n_outliers)
    return buckets, stats, probes, ol


def test_randomized_soak(ea, oracle_cpu):
    seconds = float(os.environ.get("EFFORT_SOAK_SECONDS", "6"))
    rng = np.random.default_rng(int(os.environ.get("EFFORT_SOAK_SEED", "20250927")))
    shapes = [(64, 4096), (512, 4096), (1024, 4500), (2048, 4096), (4160, 4608), (4096, 4096)]     # outDim divides 4096 or exceeds it (convert.swift:210-215)
    bank = {sh: converted(oracle_cpu, sh[0], sh[1], seed=900 + i) for i, sh in enumerate(shapes)}
    q4bank = [(_q4_bundle(rng, 4096, o, n), 4096, o) for o, n in ((1024, 3000), (4160, 40000), (2048, 0))]
    q4ews = [ea.ExpertWeights(dev16(b), devf(s), dev16(p), inSize=i, outSize=o, outliers=devf(ol) if len(ol) else None, q4=True)
             for (b, s, p, ol), i, o in q4bank]
    g = ea.Gpu(0)
    t_end, trials, calls_checked = time.time() + seconds, 0, 0
    try:
        while time.time() < t_end or trials < 4:
            q4 = bool(rng.integers(3) == 0)
            n = int(rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 16, 17, 22, 24, 32]))       # (12, 16: Q4's one-round rule; 3 .. 9: the one-item-per-CU rounds, any slice count; 8 .. 13, 22, 24: thin last calls)
            tune = [(0, 0, 0), (0, 0, 0), (0, 0, 0), (8, 1, 0), (8, 2, 24), (16, 2, 0)][int(rng.integers(6))]
            lanes = int(rng.choice([1, 1, 4]))
            g.set_overlap(lanes)
            g.set_tuning(*tune)
            launches = []
            for _ in range(lanes):                                     # `lanes` independent launches in flight, each into its own outputs
                calls, wants = [], []
                for _i in range(n if lanes == 1 else min(n, 8)):
                    effort = float(rng.choice([0.0, 0.03, 0.1, 0.25, 0.6, 1.0]))
                    if q4:
                        k = int(rng.integers(len(q4bank)))
                        (b, s, p, ol), inDim, outDim = q4bank[k]
                        ew = q4ews[k]
                        v = make_v(inDim, seed=int(rng.integers(1 << 30)), heavy=bool(rng.integers(2)))
                        wants.append(oracle_cpu.bucket_mul_q4(v, b, s, p, ol if len(ol) else None, inDim, outDim, effort))
                    else:
                        outDim, inDim = shapes[int(rng.integers(len(shapes)))]
                        W, b, s, p = bank[(outDim, inDim)]
                        pl = int(rng.choice([16, 16, 8, 3]))
                        rows = inDim * pl
                        ew = gpu_weights(ea, W, b[:rows], s[:rows], p, percentLoad=pl)
                        v = make_v(inDim, seed=int(rng.integers(1 << 30)), heavy=bool(rng.integers(2)))
                        if rng.integers(4) == 0:
                            v[rng.integers(inDim, size=40)] = 0.0
                        wants.append(oracle_cpu.bucket_mul(v, b[:rows], s[:rows], p, inDim, outDim, effort, percentLoad=pl))
                    calls.append((devf(v), ew, None, torch.full((outDim,), float("nan"), device=DEV), effort))
                launches.append((calls, wants))
            for calls, _ in launches:
                ea.bucketMulGroup(calls, gpu=g)
            g.eval()
            for li, (calls, wants) in enumerate(launches):
                if lanes > 1:
                    g.hook_lane(li)
                for i, (call, (want, cnt, cutoff)) in enumerate(zip(calls, wants)):
                    if lanes == 1 or len(launches) == lanes:
                        assert g.last_dispatch_count(i) == cnt and g.last_cutoff(i) == cutoff, (trials, q4, tune, lanes, li, i)
                    assert close(call[3].cpu().numpy(), want), (trials, q4, tune, lanes, li, i)
                    calls_checked += 1
            trials += 1
    finally:
        g.set_tuning(0, 0, 0)
        g.set_overlap(1)
        g.close()
    assert trials >= 4 and calls_checked > 0
