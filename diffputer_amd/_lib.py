"""ctypes binding of libdiffputer_hip.so (the C ABI declared in include/diffputer.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc, gfx950) and loaded from
`diffputer_amd/csrc/`.  There is no CPU fallback: if the shared object is missing this module
raises, so a GPU run can never silently go through another code path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdiffputer_hip.so")

DP_DTYPE_F32 = 0
DP_DTYPE_BF16 = 1
DP_ACCUMULATE = 1
DP_NO_FINAL_LEVEL = 2
DP_STATIC_SCRATCH_BYTES = 32 << 20

# every symbol include/diffputer.h declares: (restype, argtypes)
_vp, _i, _i64, _u64, _u32, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_uint32, C.c_float, C.c_size_t
SIGNATURES = {
    "dp_version": (_i, []),
    "dp_last_error": (C.c_char_p, []),
    "dp_plan_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "dp_plan_set_weights": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "dp_plan_destroy": (None, [_vp]),
    "dp_workspace_bytes": (_sz, [_vp, _i64]),
    "dp_impute": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _u64, _u64, _i, _vp, _vp, _i, _vp, _sz, _vp]),
    "dp_impute_static": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _u64, _u64, _i, _vp, _vp, _i, _vp]),
    "dp_denoise": (_i, [_vp, _vp, _f, _vp, _vp, _i64, _vp, _sz, _vp]),
    "dp_sample_step": (_i, [_vp, _i, _f, _f, _vp, _vp, _vp, _i64, _u64, _u64, _u32, _vp, _sz, _vp]),
    "dp_plan_schedule": (_i, [_vp, C.POINTER(_f)]),
    "dp_plan_levels": (_i, [_vp, C.POINTER(_f)]),
    "dp_plan_embedding": (_i, [_vp, C.POINTER(_f)]),
    "dp_plan_profile": (_i, [_vp, _i]),
    "dp_plan_profile_read": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(_i)]),
    "dp_plan_profile_clock": (_i, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)]),
    "dp_plan_gemm_rows": (_i64, [_vp, _i64]),
    "dp_plan_gemm_kernel": (C.c_char_p, [_vp, _i]),
    "dp_tune": (_i, [C.c_char_p, _i]),
    "dp_linear": (_i, [_i, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "dp_masked_mean_std": (_i, [_vp, _vp, _i64, _i, _vp, _vp, _vp, _sz, _vp]),
    "dp_masked_errors": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp]),
    "dp_mstep_workspace_bytes": (_sz, [_i, _i, _i64]),
    "dp_mstep_fwd_bwd": (_i, [_i, _i, _i64, _vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _vp, _sz, _vp]),
    "dp_noise_field": (_i, [_vp, _i64, _i, _u64, _u64, _i, _u32, _i, _vp]),
}

_lib = None


class DiffputerError(RuntimeError):
    pass


def load():
    """Load the HIP library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DiffputerError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc); there is no CPU fallback.")
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's): import it first so the one HIP runtime in
    # the process is the runtime that owns torch's device pointers and streams.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().dp_last_error().decode("utf-8", "replace")
        raise DiffputerError(f"{what} failed (code {rc}): {msg}")
