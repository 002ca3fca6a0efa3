"""ctypes binding of libeffort_hip.so (include/effort_hip.h; the profiling hooks of include/effort_hip_debug.h).

There is NO fallback: if the HIP library is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LAB_LIB_PATH = os.path.join(_HERE, "libeffort_hip_lab.so")      # the lab bench: same sources built -DEFFORT_LAB (device-clock stamps, traces, ablation switches)
# (the override: A/B builds of the kernels and the lab library, tools/ only; "lab" = the in-tree lab library)
LIB_PATH = (LAB_LIB_PATH if os.environ.get("EFFORT_HIP_LIB") == "lab" else os.environ.get("EFFORT_HIP_LIB")) or os.path.join(_HERE, "libeffort_hip.so")

ERRORS = {
    -1: "EFFORT_ERR_ARG", -2: "EFFORT_ERR_SHAPE", -3: "EFFORT_ERR_EFFORT", -4: "EFFORT_ERR_HIP",
    -5: "EFFORT_ERR_KIND", -6: "EFFORT_ERR_CONVERT", -7: "EFFORT_ERR_BLAS", -8: "EFFORT_ERR_COMM",
}


class EffortError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: {ERRORS.get(code, code)} {detail}".strip())


def build(verbose: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8", "all"]      # (the library alone: hipcc; the C-header check and tests/c_client are built by the test suite)
    res = subprocess.run(cmd, capture_output=not verbose, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libeffort_hip.so failed:\n" + (res.stdout or "") + (res.stderr or ""))
    return LIB_PATH


_P = C.c_void_p
_SIGS = {
    "effort_create": (_P, [C.c_int, _P]),
    "effort_destroy": (None, [_P]),
    "effort_set_stream": (C.c_int, [_P, _P]),
    "effort_sync": (C.c_int, [_P]),
    "effort_set_overlap": (C.c_int, [_P, C.c_int]),
    "effort_set_row_reuse": (C.c_int, [_P, C.c_int]),
    "effort_join": (C.c_int, [_P]),
    "effort_last_error": (C.c_char_p, [_P]),
    "effort_version": (C.c_char_p, []),
    "effort_weights_fp16": (_P, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "effort_weights_fp16_pitched": (_P, [_P, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "effort_aligned_row_pitch": (C.c_int, [C.c_int]),
    "effort_weights_q4": (_P, [_P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "effort_weights_free": (None, [_P]),
    "effort_weights_refresh": (C.c_int, [_P]),
    "effort_weights_align_rows": (C.c_int, [_P]),
    "effort_weights_row_pitch": (C.c_int, [_P]),
    "effort_weights_get_bound": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "effort_weights_set_bound": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "effort_weights_column_shard": (_P, [_P, C.c_int, C.c_int]),
    "effort_comm_unique_id": (C.c_int, [_P]),
    "effort_comm_create": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "effort_comm_destroy": (C.c_int, [_P]),
    "effort_comm_rank": (C.c_int, [_P]),
    "effort_comm_world": (C.c_int, [_P]),
    "effort_allgather_outputs": (C.c_int, [_P, _P, _P, C.c_int]),
    "effort_bucketmul": (C.c_int, [_P, _P, _P, _P, _P, C.c_double]),
    "effort_bucketmul_q4": (C.c_int, [_P, _P, _P, _P, _P, C.c_double]),
    "effort_bucketmul_group": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "effort_bucketmul_q4_group": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "effort_bucketmul_group_fused": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "effort_group_dispatch_count": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32)]),
    "effort_group_cutoff": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float)]),
    "effort_debug_occupancy": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "effort_set_persistent": (C.c_int, [_P, C.c_int]),
    "effort_debug_hook_lane": (C.c_int, [_P, C.c_int]),
    "effort_add_rmsnorm_mul": (C.c_int, [_P, _P, _P, _P, _P, C.c_int]),
    "effort_rope_kv": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "effort_attention": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int]),
    "effort_rope_attention": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "effort_silu_mul": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "effort_fetch_row": (C.c_int, [_P, _P, _P, _P, C.c_int]),
    "effort_argmax": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_int]),
    "effort_decode_status": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "effort_convert_status": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "effort_q4_outlier_count": (C.c_int64, [C.c_int, C.c_int, C.c_double]),
    "effort_convert_q4": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_double, _P, _P, _P, _P]),
    "effort_top2_softmax": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "effort_mix2": (C.c_int, [_P, _P, _P, _P, _P, C.c_int]),
    "effort_dense_gemv": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int]),
    "effort_set_dense_backend": (C.c_int, [_P, C.c_int]),
    "effort_last_dispatch_count": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "effort_last_cutoff": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "effort_calc_dispatch": (C.c_int, [_P, _P, _P, _P, C.c_double, _P, _P]),
    "effort_convert_fp16": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P]),
    "effort_convert_fp16_pitched": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    "effort_cosine": (C.c_int, [_P, _P, _P, C.c_int, C.POINTER(C.c_float)]),
    "effort_set_split_cutoff": (C.c_int, [_P, C.c_int]),
    "effort_set_tuning": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "effort_is_lab_build": (C.c_int, []),
    "effort_enable_kernel_timing": (C.c_int, [_P, C.c_int]),
    "effort_debug_stamps": (C.c_int, [_P, C.POINTER(C.c_ulonglong)]),
    "effort_debug_slice_counts": (C.c_int, [_P, C.c_int, C.POINTER(C.c_uint32), C.c_int]),
    "effort_debug_trace": (C.c_int, [_P, C.POINTER(C.c_ulonglong), C.c_int]),
    "effort_kernel_clock": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "effort_kernel_timing": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C effort_amd/csrc`). "
                "effort_amd has no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own libamdhip64 / librocblas.  Import torch FIRST so that this
        # library's NEEDED entries resolve to the copies torch already mapped (same SONAMEs): one HIP
        # runtime per process, so torch's device pointers and streams are valid here.  (A non-Python host
        # -- the Swift shim of INTEGRATION.md -- simply uses the system ROCm.)
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            if (name.startswith("effort_debug_") or name == "effort_is_lab_build") and os.environ.get("EFFORT_HIP_LIB") and not hasattr(l, name):
                continue                 # (an A/B build of an older tree may lack a profiling hook)
            fn = getattr(l, name)        # AttributeError if the ABI and the header drifted apart
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def exported_symbols() -> list[str]:
    return sorted(_SIGS)
