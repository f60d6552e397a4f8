"""IcpEngine: thin Python object over one `oa_ctx` (one GPU).

Holds no arithmetic of its own -- every number comes back from liboa_icp.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi as capi

REF_VALUEERROR = "input arrays are of wrong shape or type"    # functions/general.py:157


@dataclass
class RunResult:
    iters_done: int
    converged: bool
    matrix_world: np.ndarray          # float32 4x4: align_obj.matrix_world after the loop
    last_K: int
    last_translation: float
    mean_dist: float
    std_dist: float
    mean_rot_angle: float
    nn_ms_total: float
    loop_ms: float
    step_M: np.ndarray                # n x 4 x 4 float64, the per-iteration affine_matrix_from_points result
    step_new: np.ndarray              # n x 4 x 4 float32, new_mat (operators/icp_align.py:116-119)
    step_K: np.ndarray
    step_stats: np.ndarray            # n x 2 [mean, std]
    step_trans: np.ndarray


def _device_ptr(x):
    """(pointer, on_device, keepalive) for a numpy array or a torch tensor."""
    if hasattr(x, "data_ptr") and hasattr(x, "is_cuda"):          # torch tensor, without importing torch
        if x.is_cuda:
            import torch
            t = x.detach().to(dtype=torch.float32).contiguous().reshape(-1, 3)
            # the library reads the tensor on its own stream: whatever produced it on torch's stream (including the
            # conversion above) has to be finished first
            torch.cuda.current_stream(t.device).synchronize()
            return C.c_void_p(t.data_ptr()), 1, t, t.shape[0]
        x = x.detach().cpu().numpy()
    a = capi.as_f32(np.asarray(x)).reshape(-1, 3)
    return C.c_void_p(a.ctypes.data), 0, a, a.shape[0]


def resolve_devices(spec):
    """A device list from an int, a sequence, "all", or a string like "0,1,2,3" (what OA_DEVICES may hold)."""
    if spec is None:
        return None
    if isinstance(spec, str):
        spec = spec.strip()
        if spec.lower() == "all":
            return list(range(device_count()))
        return [int(t) for t in spec.replace(";", ",").split(",") if t.strip() != ""]
    if isinstance(spec, (int, np.integer)):
        return [int(spec)]
    return [int(d) for d in spec]


class IcpEngine:
    """One context: one GPU (`device`, the HIP ordinal -- LOCAL_RANK in one-process-per-GPU runs), or several GPUs of
    this process (`devices=[0, 1, ...]`, oa_create_multi): the source is sharded over them, the target replicated,
    and run() / iterate() join the devices' sums inside the library every iteration.  `exchange`: "auto" (default:
    RCCL's all-reduce over xGMI when the devices are distinct and librccl loads, else the mailbox), "rccl" or
    "mailbox"."""

    def __init__(self, device: int = 0, devices=None, exchange=None, experiments: bool = False):
        # experiments: liboa_icp_exp.so -- the default library + the A/B predecessors and measured-but-not-kept variants
        # (csrc/oa_families.hpp); the OA_NN_SORT / OA_NN_MFMA / OA_TRI_RING / OA_GRID_STATS / OA_TRI_SHARE knobs only act there
        self._L = capi.load(experiments=experiments)
        h = C.c_void_p()
        devs = resolve_devices(devices)
        if devs is None:
            self._chk(self._L.oa_create(C.byref(h), int(device)))
            self.devices = [int(device)]
            self.multi = False
        else:
            arr = (C.c_int * len(devs))(*devs)
            self._chk(self._L.oa_create_multi(C.byref(h), arr, len(devs)))
            self.devices = devs
            self.multi = True
        self._h = h
        self.device = self.devices[0]
        self.n_target = 0
        self.n_selected = 0
        # who uploaded the geometry last (GpuBVH objects sharing one engine check these before they trust its state)
        self.target_owner = None
        self.source_owner = None
        if exchange is not None:
            self.set_exchange(exchange)

    def _chk(self, rc):
        return capi.check(rc, self._L)

    def set_exchange(self, mode):
        """'auto', 'rccl' (ncclAllReduce over xGMI) or 'mailbox' (all-gather through peer-mapped device mailboxes --
        pinned host memory without peer access --, rank-ordered sum)."""
        code = ({"auto": capi.OA_EXCHANGE_AUTO, "mailbox": capi.OA_EXCHANGE_MAILBOX, "rccl": capi.OA_EXCHANGE_RCCL}[mode]
                if isinstance(mode, str) else int(mode))
        self._chk(self._L.oa_set_exchange(self._h, code))

    # ---- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._L.oa_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_stream(self, stream_handle):
        """stream_handle: integer hipStream_t, e.g. torch.cuda.current_stream().cuda_stream (0 = the legacy
        default stream); None = the context's private stream."""
        h = C.c_void_p(-1) if stream_handle is None else C.c_void_p(int(stream_handle))
        self._chk(self._L.oa_set_stream(self._h, h))

    def set_search_mode(self, mode):
        """'auto' (default), 'brute' (north-star LDS-tiled brute force), 'grid' (uniform-grid exact search, far
        queries finished by the tree) or 'bvh' (every query through the bounding-box tree).
        All modes return identical correspondences."""
        code = {"auto": -1, "brute": 0, "grid": 1, "bvh": 2}[mode] if isinstance(mode, str) else int(mode)
        self._chk(self._L.oa_set_search_mode(self._h, code))

    # ---- uploads
    def set_target(self, xyz):
        p, on_dev, keep, n = _device_ptr(xyz)
        self._chk(self._L.oa_set_target(self._h, p, n, on_dev))
        self.n_target = n
        self.target_owner = None
        del keep

    def set_target_mesh(self, xyz, tris):
        """Surface mode: closest point on the base mesh's triangles (BVHTree.find_nearest semantics).
        tris: (n, 3) vertex indices (triangulate quads/ngons first)."""
        p, on_dev, keep, n = _device_ptr(xyz)
        t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        self._chk(self._L.oa_set_target_mesh(self._h, p, n, on_dev, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t)))
        self.n_target = n
        self.target_owner = None
        del keep

    def set_source(self, xyz, vlist=None, stride=0, shard_index=0, shard_count=1):
        p, on_dev, keep, n = _device_ptr(xyz)
        if vlist is not None:
            vl = np.ascontiguousarray(vlist, dtype=np.int64)
            vp, nv = capi.iptr(vl), len(vl)
        else:
            vl, vp, nv = None, None, 0
        self._chk(self._L.oa_set_source(self._h, p, n, on_dev, vp, nv, int(stride), int(shard_index), int(shard_count)))
        self.n_selected = int(self._L.oa_num_selected(self._h))
        self.source_owner = None
        del keep, vl

    def set_normals(self, src_normals, tgt_normals=None, max_angle_deg=45.0):
        """Extension (not in the reference): drop pairs whose world-space normals differ by more than max_angle_deg.
        Call after set_source / set_target*; src_normals=None switches the test off."""
        if src_normals is None:
            self._chk(self._L.oa_set_normals(self._h, None, 0, None, 0, 0.0))
            return
        sn = capi.as_f32(src_normals).reshape(-1, 3)
        tn = capi.as_f32(tgt_normals).reshape(-1, 3) if tgt_normals is not None else None
        self._chk(self._L.oa_set_normals(self._h, capi.fptr(sn), len(sn), capi.fptr(tn) if tn is not None else None,
                                          len(tn) if tn is not None else 0, float(max_angle_deg)))

    def set_matrices(self, mx_align, mx_base):
        a, b = capi.as_f32(mx_align, (4, 4)), capi.as_f32(mx_base, (4, 4))
        self._chk(self._L.oa_set_matrices(self._h, capi.fptr(a), capi.fptr(b)))

    def reset_seeds(self):
        """Forget the previous searches' answers (the next search starts cold; results are unaffected)."""
        self._chk(self._L.oa_reset_seeds(self._h))

    STATS = {"grid_cells": 1, "tri_grid_cells": 2, "tri_grid_entries": 3, "n_tris": 4, "surface": 5, "cache_bytes": 6,
             "brute_kernel": 7, "exchange": 8, "rccl_ranks": 9, "enqueue_us": 10, "host_threads": 11,
             "fast_iterations": 12, "handover_entries": 13, "handover_wave_max": 14, "enqueued_min": 15, "enqueued_max": 16,
             "watchdog_aborts": 17, "nn_ms_min": 18, "nn_ms_max": 19, "safe_radii": 20,
             "tri_ring": 21, "tri_ring_accepts": 22, "exchange_us": 23, "rccl_fallbacks": 24, "rccl_ranks_last": 25, "search_clock_mhz": 26, "brute_queue_wgs": 27}
    EXCHANGE_NAMES = {-1: None, 0: "mailbox (pinned host memory)", 1: "rccl", 2: "mailbox (peer-mapped device memory)"}

    def exchange_info(self):
        """What a multi-device engine's loops exchange their sums through: {"exchange": name, "rccl_ranks": n}."""
        note = self._L.oa_exchange_note(self._h)
        return {"exchange": self.EXCHANGE_NAMES.get(int(self.stat("exchange"))), "rccl_ranks": int(self.stat("rccl_ranks")),
                "host_threads": int(self.stat("host_threads")), "note": note.decode("utf-8", "replace") if note else "",
                "rccl_ranks_last": int(self.stat("rccl_ranks_last")), "rccl_fallbacks": int(self.stat("rccl_fallbacks"))}

    def stat(self, name) -> float:
        v = C.c_double(0.0)
        self._chk(self._L.oa_get_stat(self._h, self.STATS[name] if isinstance(name, str) else int(name), C.byref(v)))
        return float(v.value)

    def search_ms(self, max_n=1 << 16) -> np.ndarray:
        """Search time (ms) of every iteration of the last run (oa_get_search_ms)."""
        out = np.zeros(int(max_n), np.float64)
        n = self._L.oa_get_search_ms(self._h, int(max_n), capi.dptr(out))
        if n < 0:
            self._chk(n)
        return out[:n].copy()

    def valu_ceiling(self, target_ms=5.0) -> dict:
        """What the vector ALUs issue right now (oa_measure_valu_ceiling): v_add_f32 (the issue rate) and v_min3_f32 (the half-rate class) on every SIMD."""
        out = np.zeros(4, np.float64)
        self._chk(self._L.oa_measure_valu_ceiling(self._h, float(target_ms), capi.dptr(out)))
        return {"tlaneops": float(out[0]), "shader_clock_mhz": float(out[1]), "ms": float(out[2]), "tlaneops_min3": float(out[3])}

    def enqueued_iterations(self):
        """Iterations the host enqueued for every child in the last run() (OA_STAT_ENQUEUED_CHILD + i): all equal, whatever
        the host threads saw of their devices while they enqueued (docs/HISTORY.md 4.7)."""
        return [int(self.stat(1000 + i)) for i in range(len(self.devices))] if self.multi else [int(self.stat("enqueued_max"))]

    def matrix_world(self) -> np.ndarray:
        out = np.empty((4, 4), np.float32)
        self._chk(self._L.oa_get_matrix_world(self._h, capi.fptr(out)))
        return out

    def pivot(self) -> np.ndarray:
        out = np.empty(3, np.float64)
        self._chk(self._L.oa_get_pivot(self._h, capi.dptr(out)))
        return out

    # ---- contract 1
    def make_pairs(self, thresh, calc_stats=False):
        cap = max(1, self.n_selected)
        A = np.zeros((3, cap), np.float64)
        B = np.zeros((3, cap), np.float64)
        K = C.c_int64(0)
        ds = np.zeros(2, np.float64)
        self._chk(self._L.oa_make_pairs(self._h, float(thresh), int(bool(calc_stats)), capi.dptr(A), capi.dptr(B),
                                         cap, C.byref(K), capi.dptr(ds)))
        k = int(K.value)
        d_stats = [float(ds[0]), float(ds[1])] if calc_stats else None
        return np.ascontiguousarray(A[:, :k]), np.ascontiguousarray(B[:, :k]), d_stats

    def nn_search(self, want_output=True):
        """Nearest target vertex per selected source point: (idx int64[n], d2 float32[n], kernel_ms)."""
        ms = C.c_double(0.0)
        if want_output:
            idx = np.empty(max(1, self.n_selected), np.int64)
            d2 = np.empty(max(1, self.n_selected), np.float32)
            self._chk(self._L.oa_nn_search(self._h, capi.iptr(idx), capi.fptr(d2), C.byref(ms)))
            return idx[: self.n_selected], d2[: self.n_selected], float(ms.value)
        self._chk(self._L.oa_nn_search(self._h, None, None, C.byref(ms)))
        return None, None, float(ms.value)

    # ---- contract 2
    def kabsch(self, A, B, scale=False, horn=False) -> np.ndarray:
        """horn: the rotation through Horn's quaternion (the reference's usesvd=False branch, functions/general.py:191-206)
        instead of the SVD of the covariance -- the same optimum, another route to it."""
        A = np.ascontiguousarray(A, np.float64)
        B = np.ascontiguousarray(B, np.float64)
        M = np.empty((4, 4), np.float64)
        rc = self._L.oa_kabsch(self._h, capi.dptr(A), capi.dptr(B), A.shape[1], A.shape[1], int(bool(scale)) | (2 if horn else 0), capi.dptr(M))
        if rc == capi.OA_E_TOO_FEW_PAIRS:
            raise ValueError(REF_VALUEERROR)
        self._chk(rc)
        return M

    def affine_from_points(self, v0, v1, shear=True, scale=True) -> np.ndarray:
        """affine_matrix_from_points in full: any ndims in 2..64, shear (affine) or rigid / similarity."""
        v0 = np.ascontiguousarray(v0, np.float64)
        v1 = np.ascontiguousarray(v1, np.float64)
        n, K = v0.shape
        M = np.empty((n + 1, n + 1), np.float64)
        rc = self._L.oa_affine_from_points(self._h, capi.dptr(v0), capi.dptr(v1), n, K, K, int(bool(shear)), int(bool(scale)),
                                           capi.dptr(M))
        if rc == capi.OA_E_TOO_FEW_PAIRS:
            raise ValueError(REF_VALUEERROR)
        self._chk(rc)
        return M

    def kabsch_from_sums(self, sums, pivot=None, scale=False) -> np.ndarray:
        s = np.ascontiguousarray(sums, np.float64).reshape(capi.OA_NSUMS)
        pv = np.ascontiguousarray(pivot, np.float64).reshape(3) if pivot is not None else None
        M = np.empty((4, 4), np.float64)
        rc = self._L.oa_kabsch_from_sums(self._h, capi.dptr(s), capi.dptr(pv) if pv is not None else None,
                                         int(bool(scale)), capi.dptr(M))
        if rc == capi.OA_E_TOO_FEW_PAIRS:
            raise ValueError(REF_VALUEERROR)
        self._chk(rc)
        return M

    # ---- the loop
    @staticmethod
    def _settings(iters, thresh, target_d, use_target, with_scale, early_exit):
        return capi.Settings(int(iters), int(bool(use_target)), int(bool(with_scale)), int(bool(early_exit)),
                             float(thresh), float(target_d))

    def _history(self, n):
        n = max(0, int(n))
        m = max(1, n)
        sM = np.zeros((m, 4, 4), np.float64)
        sN = np.zeros((m, 4, 4), np.float32)
        sK = np.zeros(m, np.int64)
        sS = np.zeros((m, 2), np.float64)
        sT = np.zeros(m, np.float64)
        got = self._L.oa_get_history(self._h, m, capi.dptr(sM), capi.fptr(sN), capi.iptr(sK), capi.dptr(sS), capi.dptr(sT))
        got = max(0, min(int(got), n))
        return sM[:got], sN[:got], sK[:got], sS[:got], sT[:got]

    def _result(self, rep) -> RunResult:
        sM, sN, sK, sS, sT = self._history(rep.iters_done)
        return RunResult(rep.iters_done, bool(rep.converged), self.matrix_world(), int(rep.last_K),
                         rep.last_translation, rep.mean_dist, rep.std_dist, rep.mean_rot_angle, rep.nn_ms_total,
                         rep.loop_ms, sM, sN, sK, sS, sT)

    def run(self, iters=50, thresh=0.5, target_d=0.01, use_target=True, with_scale=False, early_exit=True) -> RunResult:
        st = self._settings(iters, thresh, target_d, use_target, with_scale, early_exit)
        rep = capi.Report()
        rc = self._L.oa_run(self._h, C.byref(st), C.byref(rep))
        if rc == capi.OA_E_TOO_FEW_PAIRS:
            # the reference raises out of affine_matrix_from_points in iteration n, after iterations 0..n-1 have been
            # applied to the objects (operators/icp_align.py:121-127): the partial result travels with the exception
            err = ValueError(REF_VALUEERROR)
            try:
                err.partial = self._result(rep)
            except Exception:
                err.partial = None
            raise err
        self._chk(rc)
        return self._result(rep)

    def iterate(self, thresh=0.5, target_d=0.01, use_target=True, with_scale=False):
        """One iteration (modal operator step).  Returns (M float64 4x4, stats dict)."""
        st = self._settings(1, thresh, target_d, use_target, with_scale, False)
        M = np.empty((4, 4), np.float64)
        s = np.empty(6, np.float64)
        rc = self._L.oa_iterate(self._h, C.byref(st), capi.dptr(M), capi.dptr(s))
        if rc == capi.OA_E_TOO_FEW_PAIRS:
            raise ValueError(REF_VALUEERROR)
        self._chk(rc)
        return M, dict(K=int(s[0]), mean_dist=s[1], std_dist=s[2], translation=s[3], rot_angle=s[4],
                       converged=bool(s[5]))

    # ---- split phase (one process per GPU)
    def run_begin(self, iters=50, thresh=0.5, target_d=0.01, use_target=True, with_scale=False, early_exit=True):
        st = self._settings(iters, thresh, target_d, use_target, with_scale, early_exit)
        self._chk(self._L.oa_run_begin(self._h, C.byref(st)))

    def iter_partial(self, sums_device_ptr: int):
        self._chk(self._L.oa_iter_partial(self._h, C.c_void_p(sums_device_ptr)))

    def iter_finish(self, sums_device_ptr: int):
        self._chk(self._L.oa_iter_finish(self._h, C.c_void_p(sums_device_ptr)))

    def run_end(self) -> RunResult:
        rep = capi.Report()
        rc = self._L.oa_run_end(self._h, C.byref(rep))
        if rc == capi.OA_E_TOO_FEW_PAIRS:
            raise ValueError(REF_VALUEERROR)
        self._chk(rc)
        return self._result(rep)


def device_count() -> int:
    return int(capi.load().oa_device_count())


def shard_bounds(n_selected: int, shard_index: int, shard_count: int):
    """[begin, end) of shard `shard_index` -- the same contiguous split oa_set_source applies."""
    per = -(-n_selected // shard_count) if shard_count > 0 else n_selected
    begin = min(n_selected, per * shard_index)
    return begin, min(n_selected, begin + per)
