"""Shared synthetic-input helpers for the tests (SURVEY.md section 8d inputs)."""
import numpy as np


def make_w(outDim: int, inDim: int, seed: int = 1234, scale: float = 0.02, zeros: int = 0) -> np.ndarray:
    """HF-layout weight matrix f16 [outDim, inDim] ~ N(0, scale^2); optionally plant exact zeros."""
    rng = np.random.default_rng(seed)
    W = (rng.standard_normal((outDim, inDim), dtype=np.float32) * scale).astype(np.float16)
    if zeros:
        r = rng.integers(0, outDim, zeros)
        c = rng.integers(0, inDim, zeros)
        W[r, c] = 0
        W[r[: zeros // 2], c[: zeros // 2]] = np.float16(-0.0)
    return W


def make_v(inDim: int, seed: int = 42, heavy: bool = False) -> np.ndarray:
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(inDim, dtype=np.float32)
    if heavy:
        v = (v * np.exp(rng.standard_normal(inDim, dtype=np.float32))).astype(np.float32)
    return v


def cos(a, b) -> float:
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
