"""ctypes binding of liboa_icp.so (include/oa_icp.h) -- the only way Python reaches the HIP kernels.

There is no fallback: if the shared library is missing or no GPU is usable the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboa_icp.so")
LIB_EXP_PATH = os.path.join(_HERE, "liboa_icp_exp.so")      # the same library + the experiments (csrc/oa_families.hpp: OA_EXPERIMENTS)

OA_OK = 0
OA_E_BAD_ARG = -1
OA_E_HIP = -2
OA_E_TOO_FEW_PAIRS = -3
OA_E_SINGULAR = -4
OA_E_NO_DEVICE = -5
OA_E_STATE = -6
OA_E_BAD_THRESH = -7
OA_E_CAPACITY = -8
OA_E_RCCL = -9
OA_EXCHANGE_AUTO = -1
OA_EXCHANGE_MAILBOX = 0
OA_EXCHANGE_RCCL = 1
OA_NSUMS = 24

# every symbol include/oa_icp.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "oa_device_count", "oa_create", "oa_create_multi", "oa_num_devices", "oa_set_exchange", "oa_release_cached_memory",
    "oa_destroy", "oa_last_error", "oa_version", "oa_set_stream",
    "oa_set_search_mode",
    "oa_set_target", "oa_set_target_mesh", "oa_set_source", "oa_set_normals", "oa_set_matrices", "oa_get_matrix_world", "oa_num_selected",
    "oa_reset_seeds", "oa_get_stat",
    "oa_make_pairs", "oa_nn_search", "oa_kabsch", "oa_affine_from_points", "oa_kabsch_from_sums", "oa_get_pivot",
    "oa_iterate", "oa_run", "oa_get_history", "oa_run_begin", "oa_iter_partial", "oa_iter_finish", "oa_run_end",
    "oa_get_search_ms", "oa_measure_valu_ceiling", "oa_exchange_note",
]


class Settings(C.Structure):
    _fields_ = [("iters", C.c_int32), ("use_target", C.c_int32), ("with_scale", C.c_int32),
                ("early_exit", C.c_int32), ("thresh", C.c_double), ("target_d", C.c_double)]


class Report(C.Structure):
    _fields_ = [("iters_done", C.c_int32), ("converged", C.c_int32), ("status", C.c_int32),
                ("reserved", C.c_int32), ("last_K", C.c_int64), ("last_translation", C.c_double),
                ("mean_dist", C.c_double), ("std_dist", C.c_double), ("mean_rot_angle", C.c_double),
                ("nn_ms_total", C.c_double), ("loop_ms", C.c_double)]


class OaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("oa_icp error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


_libs = {}


def load(experiments: bool = False):
    """Load liboa_icp.so (experiments=True: liboa_icp_exp.so, the flavour the A/B tests use).  Raises (loudly) when the HIP
    extension has not been built."""
    if experiments in _libs:
        return _libs[experiments]
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7.  If liboa_icp.so pulled in the
    # system copy first, torch would later fail with "No HIP GPUs are available"; importing torch first makes both
    # bind to the same runtime.  Without torch installed the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # Another BUILD of the same library (the sanitizer pass, A/B experiments) -- only when OA_ICP_LIB_DEBUG=1 says that the
    # override is meant, and never silently: an environment variable alone does not redirect what a production import loads.
    path = LIB_EXP_PATH if experiments else LIB_PATH
    override = os.environ.get("OA_ICP_LIB")
    if override:
        if os.environ.get("OA_ICP_LIB_DEBUG") == "1":
            import sys
            print("object_alignment_amd: loading %s instead of %s (OA_ICP_LIB, OA_ICP_LIB_DEBUG=1)" % (override, LIB_PATH), file=sys.stderr)
            path = override
        else:
            import warnings
            warnings.warn("OA_ICP_LIB is set but ignored: set OA_ICP_LIB_DEBUG=1 as well to load another build of liboa_icp.so")
    if not os.path.exists(path):
        raise RuntimeError(
            "object_alignment_amd: %s is missing -- build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp, fp, dp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int64)
    L.oa_device_count.restype = C.c_int
    L.oa_create.argtypes = [C.POINTER(vp), C.c_int]
    L.oa_create_multi.argtypes = [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]
    L.oa_num_devices.argtypes = [vp]
    L.oa_set_exchange.argtypes = [vp, C.c_int]
    L.oa_release_cached_memory.restype = None
    L.oa_destroy.argtypes = [vp]
    L.oa_destroy.restype = None
    L.oa_last_error.restype = C.c_char_p
    L.oa_version.restype = C.c_char_p
    L.oa_set_stream.argtypes = [vp, vp]
    L.oa_set_search_mode.argtypes = [vp, C.c_int]
    L.oa_set_target.argtypes = [vp, vp, C.c_int64, C.c_int]
    L.oa_set_target_mesh.argtypes = [vp, vp, C.c_int64, C.c_int, C.POINTER(C.c_int32), C.c_int64]
    L.oa_set_source.argtypes = [vp, vp, C.c_int64, C.c_int, ip, C.c_int64, C.c_int32, C.c_int32, C.c_int32]
    L.oa_set_normals.argtypes = [vp, fp, C.c_int64, fp, C.c_int64, C.c_double]
    L.oa_set_matrices.argtypes = [vp, fp, fp]
    L.oa_get_matrix_world.argtypes = [vp, fp]
    L.oa_num_selected.argtypes = [vp]
    L.oa_reset_seeds.argtypes = [vp]
    L.oa_get_stat.argtypes = [vp, C.c_int, dp]
    L.oa_num_selected.restype = C.c_int64
    L.oa_make_pairs.argtypes = [vp, C.c_double, C.c_int, dp, dp, C.c_int64, ip, dp]
    L.oa_nn_search.argtypes = [vp, ip, fp, dp]
    L.oa_kabsch.argtypes = [vp, dp, dp, C.c_int64, C.c_int64, C.c_int, dp]
    L.oa_affine_from_points.argtypes = [vp, dp, dp, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, dp]
    L.oa_kabsch_from_sums.argtypes = [vp, dp, dp, C.c_int, dp]
    L.oa_get_pivot.argtypes = [vp, dp]
    L.oa_iterate.argtypes = [vp, C.POINTER(Settings), dp, dp]
    L.oa_run.argtypes = [vp, C.POINTER(Settings), C.POINTER(Report)]
    L.oa_get_history.argtypes = [vp, C.c_int32, dp, fp, ip, dp, dp]
    L.oa_exchange_note.argtypes = [vp]
    L.oa_exchange_note.restype = C.c_char_p
    L.oa_get_search_ms.argtypes = [vp, C.c_int32, dp]
    L.oa_measure_valu_ceiling.argtypes = [vp, C.c_double, dp]
    L.oa_run_begin.argtypes = [vp, C.POINTER(Settings)]
    L.oa_iter_partial.argtypes = [vp, vp]
    L.oa_iter_finish.argtypes = [vp, vp]
    L.oa_run_end.argtypes = [vp, C.POINTER(Report)]
    _libs[experiments] = L
    return L


def check(rc, lib=None):
    if rc != OA_OK:
        raise OaError(rc, (lib if lib is not None else load()).oa_last_error().decode("utf-8", "replace"))
    return rc


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def as_f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a
