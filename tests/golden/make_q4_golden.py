"""Generate tests/golden/q4_*.npz by running the REFERENCE's q4_draft.convert.

Run in the authoring container only (needs /root/reference; the GPU box has no reference):

    python tests/golden/make_q4_golden.py

Each fixture holds the seeded input (core2 = W.T as f16, v as f32) and everything the reference
returned / printed for it: probes, bucket.stats, buckets (uint16 view), outliers and the draft's
effort-free multiply ``output_vector2`` (captured from its print call, q4_draft.py:228).
Seeds are advanced until the 2 %-outlier boundary has no |w| tie, because the reference's
unstable argsort leaves that case unspecified.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def run_reference(core2, v):
    sys.path.insert(0, REF)
    import q4_draft  # noqa: E402  (reference module; imported here only)

    captured = []
    q4_draft.print = lambda *a, **k: captured.append(a)     # module-level shadow of builtins.print
    q4_draft.v = v                                            # convert() reads a module global `v` (:209)
    out = q4_draft.convert(core2)
    ov2 = [a[0] for a in captured if len(a) == 1 and isinstance(a[0], np.ndarray)
           and a[0].shape == (core2.shape[1],) and a[0].dtype == np.float64]
    assert len(ov2) == 1
    return out, ov2[0]


def boundary_is_unique(core2, perc=0.02):
    a = np.sort(np.abs(core2.astype(np.float32)).ravel())[::-1]
    cnt = int(a.size * perc)
    return a[cnt - 1] != a[cnt]


def make(name, inDim, outDim, seed, scale=0.02, heavy=False):
    while True:
        rng = np.random.default_rng(seed)
        W = (rng.standard_normal((outDim, inDim)) * scale).astype(np.float16)   # HF layout [out, in]
        core2 = W.T                                                             # q4_convert.py:54,63
        if boundary_is_unique(core2):
            break
        seed += 1000
    v = rng.standard_normal(inDim).astype(np.float32)
    if heavy:
        v = (v * np.exp(rng.standard_normal(inDim))).astype(np.float32)
    out, ov2 = run_reference(core2, v)
    np.savez_compressed(
        os.path.join(HERE, f"q4_{name}.npz"),
        core2=np.ascontiguousarray(core2), v=v, seed=np.int64(seed),
        probes=out["probes"], bucket_stats=out["bucket.stats"],
        buckets_u16=np.ascontiguousarray(out["buckets"]).view(np.uint16),
        outliers=out["outliers"], output_vector2=ov2)
    print(name, "seed", seed, "buckets", out["buckets"].shape, "outliers", out["outliers"].shape)


if __name__ == "__main__":
    make("64x256", 64, 256, 11)
    make("96x160", 96, 160, 12, heavy=True)
    make("32x512", 32, 512, 13, scale=0.5)
    make("16x2048", 16, 2048, 14)            # rank rows of 256 elements: numpy's pairwise sum recurses once
    make("8x11008", 8, 11008, 15)            # the FFN width: rank rows of 1376 elements (uneven recursion)
