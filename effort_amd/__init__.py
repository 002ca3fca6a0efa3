"""effort_amd -- MI355X (gfx950) implementation of kolinko/effort's bucketMul hot path.

Hand-written HIP kernels behind a C ABI (include/effort_hip.h, built as effort_amd/libeffort_hip.so) and
this thin host mirror of the reference's Swift interface for the path: ``bucketMul``, ``bucketMulQ4``,
``expertMul``, ``basicMul``, ``ExpertWeights``, ``BucketMul``, ``bucketize``, ``gpu().eval()``; plus the on-disk bucket
format and the model converter driver (``TensorSaver``, ``TensorLoader``, ``convertMistral``, ``loadExpertWeights``).
PyTorch is used for device memory, streams and torch.distributed only.  There is no CPU fallback.
"""
from ._lib import EffortError, build, lib  # noqa: F401


def __getattr__(name):
    # torch-dependent parts are imported lazily so that `import effort_amd` (and the ABI checks) work
    # on a machine without a GPU.
    if name in ("gpu", "Gpu"):
        from . import runtime as _g
        return getattr(_g, name)
    if name in ("ExpertWeights",):
        from . import weights as _w
        return getattr(_w, name)
    if name in ("bucketMul", "bucketMulQ4", "bucketMulGroup", "expertMul", "basicMul", "BucketMul", "BucketMulQ4", "cosineSimilarityTo"):
        from . import bucket_mul as _b
        return getattr(_b, name)
    if name == "bucketize":
        from . import convert as _c
        return _c.bucketize
    if name == "q4_convert":
        from . import q4 as _q
        return _q.convert
    if name in ("TensorSaver", "TensorLoader", "convertMistral", "loadExpertWeights"):
        from . import bucketfile as _f
        return getattr(_f, name)
    if name in ("ShardedExpertWeights", "shardedExpertMul"):
        from . import sharded as _s
        return getattr(_s, name)
    raise AttributeError(name)
