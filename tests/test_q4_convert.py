"""The product's Q4 converter (effort_amd/q4.py, batched tensor ops) against the fixtures produced by the
reference's q4_draft.convert (CPU tensors here; the same code runs on the GPU in test_gpu_parity)."""
import glob
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "q4_*.npz"))), ids=os.path.basename)
def test_product_q4_converter_matches_reference_fixture(path):
    from effort_amd.q4 import convert
    g = np.load(path)
    out = convert(torch.from_numpy(g["core2"]))
    assert np.array_equal(out["buckets"].numpy().view(np.uint16), g["buckets_u16"])
    assert np.array_equal(out["bucket.stats"].numpy(), g["bucket_stats"])
    assert np.array_equal(out["probes"].numpy().view(np.uint16), g["probes"].view(np.uint16))
    a, b = out["outliers"].numpy(), g["outliers"]
    assert np.array_equal(a[np.lexsort((a[:, 2], a[:, 1]))], b[np.lexsort((b[:, 2], b[:, 1]))])


def test_pairwise_sum_is_numpys():
    from effort_amd.q4 import _np_pairwise_sum_f32
    rng = np.random.default_rng(0)
    for n in (1, 5, 8, 9, 63, 128, 129, 256, 344, 1376, 2048, 3000):
        a = np.abs(rng.standard_normal((7, n)).astype(np.float16))
        want = np.array([np.add.reduce(row, dtype=np.float32) for row in a], np.float32)
        got = _np_pairwise_sum_f32(torch.from_numpy(a).to(torch.float32)).numpy()
        assert np.array_equal(got, want), n


def test_q4_converter_preconditions():
    from effort_amd.q4 import convert
    with pytest.raises(ValueError):
        convert(torch.zeros((8, 48), dtype=torch.float16))
    with pytest.raises(ValueError):
        convert(torch.zeros((8, 64), dtype=torch.float32))
