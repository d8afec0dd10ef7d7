"""ctypes binding of the C ABI declared in include/qs_amd.h (libqs_hip.so, built by quadruped-springs_amd/build.py).

There is no CPU path: if the library is missing or no HIP device is usable this module raises."""
import ctypes as C
import os

from .config import QsConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QS_LIB_PATH") or os.path.join(_HERE, "libqs_hip.so")   # QS_LIB_PATH: kernel experiments (another build of the same ABI)

ABI_VERSION = 5   # QS_ABI_VERSION of include/qs_amd.h this file was written against

EXPORTS = (
    "qs_create", "qs_destroy", "qs_set_stream", "qs_reset", "qs_reset_to", "qs_get_obs", "qs_step", "qs_step_fused", "qs_get_state", "qs_set_state",
    "qs_info_dim", "qs_get_info", "qs_set_params", "qs_stats", "qs_enable_timing", "qs_last_step_kernel_ms",
    "qs_settle_lanes", "qs_host_step_begin", "qs_host_step_end", "qs_set_trace", "qs_counter", "qs_counters_async", "qs_set_demo", "qs_set_demo_counter", "qs_last_error", "qs_version", "qs_abi_version",
    "qs_norm_create", "qs_norm_destroy", "qs_norm_dims", "qs_norm_set_stream", "qs_norm_set_stats", "qs_norm_get_stats", "qs_norm_reset", "qs_norm_step",
    "qs_norm_step_io", "qs_host_set_norm",
)



class HostResult(C.Structure):
    """qs_host_result (include/qs_amd.h): where the results of a host-path step lie in the handle's page-locked host memory."""
    _fields_ = [("obs", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p), ("truncated", C.c_void_p), ("terminal_rows", C.c_void_p),
                ("terminal_cap", C.c_int32)]


class NormIO(C.Structure):
    """qs_norm_io (include/qs_amd.h): the arrays of a step for qs_norm_step_io and, optionally, where the normalised ones go."""
    _fields_ = [("obs", C.c_void_p), ("rew", C.c_void_p), ("done", C.c_void_p), ("trunc", C.c_void_p), ("term_obs", C.c_void_p), ("tail_rows", C.c_void_p),
                ("tail_cap", C.c_int32),
                ("out_obs", C.c_void_p), ("out_rew", C.c_void_p), ("out_done", C.c_void_p), ("out_trunc", C.c_void_p), ("out_tail", C.c_void_p),
                ("raw_obs", C.c_void_p), ("raw_rew", C.c_void_p), ("tail_count", C.c_void_p)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python quadruped-springs_amd/build.py` (needs hipcc). "
                           "The simulation step has no CPU fallback.")
    # The library must share ONE HIP runtime with PyTorch (device pointers and streams cross the boundary).  torch
    # wheels bundle their own libamdhip64 / libhsa-runtime64; importing torch first makes the loader satisfy this
    # library's DT_NEEDED entry with the runtime torch already mapped.  Loaded the other way round the process ends up
    # with two HSA runtimes and the second one sees no device (measured on the MI355X box, tools/diag_hip_runtime.py).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    lib.qs_create.argtypes = [C.POINTER(QsConfig), i32, C.POINTER(vp)]
    lib.qs_destroy.argtypes = [vp]
    lib.qs_destroy.restype = None
    lib.qs_set_stream.argtypes = [vp, vp]
    lib.qs_reset.argtypes = [vp, vp]
    lib.qs_reset_to.argtypes = [vp, vp, vp]
    lib.qs_get_obs.argtypes = [vp, vp]
    lib.qs_step.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.qs_step_fused.argtypes = [vp, vp, vp]
    lib.qs_get_state.argtypes = [vp, vp]
    lib.qs_set_state.argtypes = [vp, vp]
    lib.qs_info_dim.argtypes = [vp, i32]
    lib.qs_get_info.argtypes = [vp, i32, vp]
    lib.qs_set_params.argtypes = [vp, i32, vp]
    lib.qs_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.qs_enable_timing.argtypes = [vp, i32]
    lib.qs_last_step_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.qs_settle_lanes.argtypes = [vp, C.c_int]
    lib.qs_host_step_begin.argtypes = [vp, vp]
    lib.qs_host_step_end.argtypes = [vp, C.POINTER(HostResult)]
    lib.qs_set_trace.argtypes = [vp, C.c_int, vp]
    lib.qs_set_demo.argtypes = [vp, vp, C.c_int]
    lib.qs_set_demo_counter.argtypes = [vp, vp, vp]
    lib.qs_counter.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
    lib.qs_counters_async.argtypes = [vp, vp]
    f32, f64, pd = C.c_float, C.c_double, C.POINTER(C.c_double)
    lib.qs_norm_create.argtypes = [i32, i32, f64, f64, f64, f64, i32, C.POINTER(vp)]
    lib.qs_norm_destroy.argtypes = [vp]
    lib.qs_norm_dims.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.qs_norm_destroy.restype = None
    lib.qs_norm_set_stream.argtypes = [vp, vp]
    lib.qs_norm_set_stats.argtypes = [vp, vp, vp, f64, f64, f64, f64]
    lib.qs_norm_get_stats.argtypes = [vp, vp, vp, pd, pd, pd, pd]
    lib.qs_norm_reset.argtypes = [vp, vp, i32, i32]
    lib.qs_norm_step.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.qs_norm_step_io.argtypes = [vp, C.POINTER(NormIO), i32, i32, i32]
    lib.qs_host_set_norm.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.qs_last_error.restype = C.c_char_p
    lib.qs_version.restype = C.c_char_p
    # (QS_ALLOW_ABI_MISMATCH=1: the A/B tools that time an older round's library through QS_LIB_PATH on entry points that did not change)
    if os.environ.get("QS_ALLOW_ABI_MISMATCH") != "1" and (not hasattr(lib, "qs_abi_version") or lib.qs_abi_version() != ABI_VERSION):
        have = lib.qs_abi_version() if hasattr(lib, "qs_abi_version") else "none (a library of round 4 or earlier)"
        raise RuntimeError(f"{LIB_PATH} speaks ABI {have}, this binding was written against ABI {ABI_VERSION} of include/qs_amd.h: rebuild it "
                           "(python quadruped-springs_amd/build.py --force)")
    _lib = lib
    return lib


def source_sha(lib=None):
    """The fingerprint of the source tree the loaded library says it was compiled from (build.py's source_fingerprint(), passed to the
    compiler as QS_SOURCE_SHA and returned inside qs_version()); None for a library that does not say (round 5 or earlier)."""
    import re
    m = re.search(r"source ([0-9a-f]{64})", (lib or load()).qs_version().decode())
    return m.group(1) if m else None


def check(rc):
    if rc != 0:
        raise RuntimeError("qs_amd: " + load().qs_last_error().decode())
