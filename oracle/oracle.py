"""CPU oracle for the ICP hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``object_alignment_amd/`` may import this module.  It is used by
``tests/``, ``__graft_entry__.smoke()``, ``tools/gen_golden.py`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / timed CPU baseline, never
as the thing shipped.

Two layers:

* ``liboa_oracle.so`` (``oracle/oa_oracle.c``, built by ``oracle/Makefile``): the C
  restatement -- float32 mathutils arithmetic, brute-force and KD-tree nearest
  vertex, ``make_pairs``, Kabsch, the operator loop.
* numpy restatements in this file of ``affine_matrix_from_points``
  (/root/reference/functions/general.py:146-167,179-190,208-216, using
  ``numpy.linalg.svd`` exactly as the reference does) and float32 ``Vector`` /
  ``Matrix`` stand-ins for Blender's ``mathutils`` (used to drive the *imported*
  reference in ``tools/gen_golden.py``).

Parity status: pinned against reference-derived fixtures in ``tests/golden``
(see ``tests/test_oracle_golden.py``) for everything that lives in
/root/reference; the Blender-side float32 arithmetic and the BVH are restated
from API knowledge and are "parity unpinned" (SURVEY.md section 8c).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboa_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc).  Returns the .so path."""
    src = os.path.join(_HERE, "oa_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboa_oracle.so"])
    return _LIB_PATH


class Settings(C.Structure):
    _fields_ = [("iters", C.c_int32), ("sample", C.c_int32), ("use_target", C.c_int32),
                ("with_scale", C.c_int32), ("thresh", C.c_double), ("target_d", C.c_double)]


class Report(C.Structure):
    _fields_ = [("iters_done", C.c_int32), ("converged", C.c_int32), ("status", C.c_int32),
                ("pad", C.c_int32), ("last_translation", C.c_double), ("mean_dist", C.c_double),
                ("std_dist", C.c_double), ("last_K", C.c_int64)]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp, dp, ip, vp = (C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_void_p)
        L.oo_mat4_mul_vec3.argtypes = [fp, fp, fp]
        L.oo_mat4_mul.argtypes = [fp, fp, fp]
        L.oo_mat4_inverted.argtypes = [fp, fp]
        L.oo_mat4_inverted.restype = C.c_int
        L.oo_vec3_length.argtypes = [fp]
        L.oo_vec3_length.restype = C.c_double
        L.oo_dist2.argtypes = [fp, fp]
        L.oo_dist2.restype = C.c_float
        L.oo_nn_brute.argtypes = [fp, C.c_int64, fp, C.c_int64, ip, fp]
        L.oo_kd_build.argtypes = [fp, C.c_int64]
        L.oo_kd_build.restype = vp
        L.oo_kd_free.argtypes = [vp]
        L.oo_kd_query.argtypes = [vp, fp, C.c_int64, ip, fp, C.c_int]
        L.oo_max_threads.restype = C.c_int
        i32p = C.POINTER(C.c_int32)
        L.oo_closest_on_tri.argtypes = [fp, fp, fp, fp, fp]
        L.oo_nn_tri_brute.argtypes = [fp, C.c_int64, fp, i32p, C.c_int64, ip, fp, fp]
        L.oo_make_pairs.argtypes = [fp, C.c_int64, ip, C.c_int64, C.c_int, fp, C.c_int64, vp, i32p, C.c_int64, fp, fp, C.c_double, fp, fp,
                                    C.c_double, C.c_int, C.c_int, dp, dp, C.c_int64, dp, ip]
        L.oo_make_pairs.restype = C.c_int64
        L.oo_kabsch.argtypes = [dp, dp, C.c_int64, C.c_int64, C.c_int, dp]
        L.oo_kabsch.restype = C.c_int
        L.oo_icp_run.argtypes = [fp, C.c_int64, ip, C.c_int64, fp, C.c_int64, vp, i32p, C.c_int64, fp, fp, C.c_double, fp, fp,
                                 C.POINTER(Settings), C.c_int, C.POINTER(Report), dp, fp, ip, dp, dp]
        L.oo_icp_run.restype = C.c_int
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


# ---------------------------------------------------------------- mathutils-like float32 ops

def mat4_mul(a, b):
    a, b = _f32(a, (4, 4)), _f32(b, (4, 4))
    out = np.empty((4, 4), np.float32)
    lib().oo_mat4_mul(_f(a), _f(b), _f(out))
    return out


def set_inverse_rule(rule: int):
    """0 (default): Matrix.inverted() as Blender computes it -- float adjoint / float determinant; 1: the rule of rounds 1-4
    (adjugate in double, rounded once).  Only tests/test_oracle_golden.py switches it, to measure what the choice moves."""
    L = lib()
    L.oo_set_inverse_rule.argtypes = [C.c_int]
    L.oo_set_inverse_rule.restype = None
    L.oo_set_inverse_rule(int(rule))


def mat4_inverted(a):
    a = _f32(a, (4, 4))
    out = np.empty((4, 4), np.float32)
    if not lib().oo_mat4_inverted(_f(a), _f(out)):
        raise ValueError("Matrix.invert(ed): matrix does not have an inverse")
    return out


def mat4_mul_vec3(m, v):
    m, v = _f32(m, (4, 4)), _f32(v, (3,))
    out = np.empty(3, np.float32)
    lib().oo_mat4_mul_vec3(_f(m), _f(v), _f(out))
    return out


def vec3_length(v):
    return float(lib().oo_vec3_length(_f(_f32(v, (3,)))))


# ---------------------------------------------------------------- nearest vertex

def nn_brute(queries, target):
    q, t = _f32(queries).reshape(-1, 3), _f32(target).reshape(-1, 3)
    idx = np.empty(len(q), np.int64)
    d2 = np.empty(len(q), np.float32)
    lib().oo_nn_brute(_f(q), len(q), _f(t), len(t), _i(idx), _f(d2))
    return idx, d2


class KDTree:
    """Exact nearest vertex (same answers as :func:`nn_brute`, ties included)."""

    def __init__(self, target):
        self.target = _f32(target).reshape(-1, 3)
        self._h = lib().oo_kd_build(_f(self.target), len(self.target))

    def query(self, queries, nthreads=0):
        q = _f32(queries).reshape(-1, 3)
        idx = np.empty(len(q), np.int64)
        d2 = np.empty(len(q), np.float32)
        lib().oo_kd_query(self._h, _f(q), len(q), _i(idx), _f(d2), int(nthreads))
        return idx, d2

    def __del__(self):
        try:
            if self._h:
                lib().oo_kd_free(self._h)
                self._h = None
        except Exception:
            pass


def max_threads():
    return int(lib().oo_max_threads())


# ---------------------------------------------------------------- make_pairs / kabsch / loop (C)

def _tris(tris):
    if tris is None:
        return None, None, 0
    t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    return t, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t)


def _normals(normals, max_angle_deg):
    """normals = (source normals [n_verts,3], target vertex normals [nt,3] or None in surface mode) or None."""
    if normals is None:
        return None, None, -2.0
    sn = _f32(normals[0]).reshape(-1, 3)
    tn = _f32(normals[1]).reshape(-1, 3) if normals[1] is not None else None
    return sn, tn, math.cos(max_angle_deg * 3.14159265358979323846 / 180.0)


def nn_tri_brute(queries, verts, tris):
    """Closest point on the triangle surface (BVHTree.find_nearest semantics): (face, co1, d2)."""
    q, v = _f32(queries).reshape(-1, 3), _f32(verts).reshape(-1, 3)
    t, tp, nt = _tris(tris)
    face = np.empty(len(q), np.int64)
    co1 = np.empty((len(q), 3), np.float32)
    d2 = np.empty(len(q), np.float32)
    lib().oo_nn_tri_brute(_f(q), len(q), _f(v), tp, nt, _i(face), _f(co1), _f(d2))
    return face, co1, d2


def make_pairs(src, target, mx_align, mx_base, thresh, vlist=None, sample=0, calc_stats=False,
               kd: KDTree | None = None, nthreads=0, return_nn=False, tris=None, normals=None, max_angle_deg=45.0):
    """Restates /root/reference/functions/general.py:257-329 with a nearest-vertex provider.

    Returns (A, B, d_stats) with A, B float64[3, K]; raises TypeError-equivalent
    (here: ValueError) when thresh <= 0 because the reference returns None there.
    """
    src = _f32(src).reshape(-1, 3)
    tgt = _f32(target).reshape(-1, 3)
    m1, m2 = _f32(mx_align, (4, 4)), _f32(mx_base, (4, 4))
    if vlist is not None:
        vl = np.ascontiguousarray(vlist, dtype=np.int64)
        n_all = len(vl)
    else:
        vl, n_all = None, len(src)
    step = sample if sample > 1 else 1
    cap = max(1, (n_all + step - 1) // step)
    A = np.zeros((3, cap), np.float64)
    B = np.zeros((3, cap), np.float64)
    ds = np.zeros(2, np.float64)
    nn = np.empty(cap, np.int64)
    tk, tp, ntri = _tris(tris)
    sn, tn, cmin = _normals(normals, max_angle_deg)
    K = lib().oo_make_pairs(_f(src), len(src), _i(vl) if vl is not None else None, n_all, int(sample),
                            _f(tgt), len(tgt), kd._h if kd is not None else None, tp, ntri,
                            _f(sn) if sn is not None else None, _f(tn) if tn is not None else None, cmin, _f(m1), _f(m2),
                            float(thresh), int(bool(calc_stats)), int(nthreads), _d(A), _d(B), cap, _d(ds), _i(nn))
    if K == -1:
        raise ValueError("make_pairs: thresh must be > 0 (the reference returns None here)")
    if K == -2:
        raise ValueError("Matrix.invert(ed): matrix does not have an inverse")
    if K < 0:
        raise RuntimeError("oo_make_pairs failed: %d" % K)
    A = np.ascontiguousarray(A[:, :K])
    B = np.ascontiguousarray(B[:, :K])
    d_stats = [float(ds[0]), float(ds[1])] if calc_stats else None
    if return_nn:
        return A, B, d_stats, nn[: (n_all + step - 1) // step]
    return A, B, d_stats


def kabsch_c(A, B, scale=False):
    A = np.ascontiguousarray(A, np.float64)
    B = np.ascontiguousarray(B, np.float64)
    if A.ndim != 2 or A.shape != B.shape or A.shape[0] != 3:
        raise ValueError("input arrays are of wrong shape or type")
    M = np.empty((4, 4), np.float64)
    rc = lib().oo_kabsch(_d(A), _d(B), A.shape[1], A.shape[1], int(bool(scale)), _d(M))
    if rc != 0:
        raise ValueError("input arrays are of wrong shape or type")
    return M


def affine_matrix_from_points(v0, v1, shear=False, scale=False, usesvd=True):
    """numpy restatement of /root/reference/functions/general.py:105-217: the shear branch (:168-178) and the SVD
    branch (:179-190, :208-212), any ndims >= 2.  (Horn's quaternion branch, usesvd=False, is not restated: it
    minimises the same objective; the golden fixtures hold the reference's own Horn results.)"""
    if not usesvd and not shear and np.shape(v0)[0] == 3:
        raise NotImplementedError("oracle does not restate the quaternion branch")
    v0 = np.array(v0, dtype=np.float64, copy=True)
    v1 = np.array(v1, dtype=np.float64, copy=True)
    ndims = v0.shape[0]
    if ndims < 2 or v0.shape[1] < ndims or v0.shape != v1.shape:      # :150
        raise ValueError("input arrays are of wrong shape or type")    # :157
    c0 = np.mean(v0, axis=1)                                           # :160
    c1 = np.mean(v1, axis=1)                                           # :164
    a = v0 - c0.reshape(ndims, 1)
    b = v1 - c1.reshape(ndims, 1)
    M = np.identity(ndims + 1)
    if shear:
        stacked = np.concatenate((a, b), axis=0)                       # :170
        _, _, vh = np.linalg.svd(stacked.T)                            # :171
        vh = vh[:ndims].T                                              # :172
        t = np.dot(vh[ndims:2 * ndims], np.linalg.pinv(vh[:ndims]))    # :173-176
        M[:ndims, :ndims] = t                                          # :177-178
    else:
        u, s, vh = np.linalg.svd(b @ a.T)                              # :181
        R = u @ vh                                                     # :183
        if np.linalg.det(R) < 0.0:                                     # :184
            R = R - np.outer(u[:, ndims - 1], vh[ndims - 1, :] * 2.0)  # :186
        M[:ndims, :ndims] = R                                          # :189-190
        if scale:
            M[:ndims, :ndims] *= math.sqrt(np.sum(b * b) / np.sum(a * a))  # :208-212
    M0 = np.identity(ndims + 1)
    M0[:ndims, ndims] = -c0
    M1inv = np.identity(ndims + 1)
    M1inv[:ndims, ndims] = c1
    M = M1inv @ (M @ M0)                                               # :215
    M /= M[ndims, ndims]                                               # :216
    return M


def icp_run(src, target, mx_align, mx_base, *, iters=50, sample=2, thresh=0.5, target_d=0.01,
            use_target=True, with_scale=False, vlist=None, kd: KDTree | None = None, nthreads=0, tris=None,
            normals=None, max_angle_deg=45.0):
    """Restates /root/reference/operators/icp_align.py:91-151.  Returns a dict."""
    src = _f32(src).reshape(-1, 3)
    tgt = _f32(target).reshape(-1, 3)
    m1 = _f32(mx_align, (4, 4)).copy()
    m2 = _f32(mx_base, (4, 4))
    vl = np.ascontiguousarray(vlist, dtype=np.int64) if vlist is not None else None
    st = Settings(int(iters), int(sample), int(bool(use_target)), int(bool(with_scale)), float(thresh), float(target_d))
    rep = Report()
    n = max(1, int(iters))
    step_M = np.zeros((n, 4, 4), np.float64)
    step_new = np.zeros((n, 4, 4), np.float32)
    step_K = np.zeros(n, np.int64)
    step_stats = np.zeros((n, 2), np.float64)
    step_trans = np.zeros(n, np.float64)
    tk, tp, ntri = _tris(tris)
    sn, tn, cmin = _normals(normals, max_angle_deg)
    rc = lib().oo_icp_run(_f(src), len(src), _i(vl) if vl is not None else None, len(vl) if vl is not None else 0,
                          _f(tgt), len(tgt), kd._h if kd is not None else None, tp, ntri,
                          _f(sn) if sn is not None else None, _f(tn) if tn is not None else None, cmin, _f(m1), _f(m2),
                          C.byref(st), int(nthreads), C.byref(rep), _d(step_M), _f(step_new), _i(step_K),
                          _d(step_stats), _d(step_trans))
    d = rep.iters_done
    return dict(status=rc, iters_done=d, converged=bool(rep.converged), matrix_world=m1,
                step_M=step_M[:d], step_new=step_new[:d], step_K=step_K[:d], step_stats=step_stats[:d],
                step_trans=step_trans[:d], last_translation=rep.last_translation,
                mean_dist=rep.mean_dist, std_dist=rep.std_dist, last_K=int(rep.last_K))


# ---------------------------------------------------------------- vlist mask (restated)

def build_vlist(n_verts, include=None, exclude=None):
    """Restates /root/reference/operators/icp_align.py:56-80.

    include / exclude: None (group absent) or a list of (vertex_index, weight)
    memberships of the 'icp_include' / 'icp_exclude' vertex group.
    """
    if include is not None:
        # a vertex can hold one membership per group; keep weight > 0.9   (:59-65)
        out = [int(v) for v, w in sorted(include, key=lambda t: t[0]) if np.float32(w) > 0.9]
        return out
    if exclude is not None:
        member = {int(v): np.float32(w) for v, w in exclude}
        out = []
        for v in range(n_verts):                                        # :69-76
            if v not in member:
                out.append(v)
            elif member[v] < 0.1:
                out.append(v)
        return out
    return list(range(n_verts))                                         # :80


# ---------------------------------------------------------------- float32 stand-ins for mathutils
# Used by tools/gen_golden.py to drive the IMPORTED reference make_pairs.

class Vector:
    __slots__ = ("v",)

    def __init__(self, seq):
        self.v = np.array(seq, dtype=np.float32).reshape(3)

    def __getitem__(self, i):
        return float(self.v[i])

    def __len__(self):
        return 3

    def __sub__(self, other):
        return Vector(self.v - other.v)          # float32 elementwise

    @property
    def length(self):
        return vec3_length(self.v)

    def __repr__(self):
        return "Vector(%r)" % (self.v.tolist(),)


class _Row:
    def __init__(self, m, r):
        self._m, self._r = m, r

    def __getitem__(self, c):
        return float(self._m.m[self._r, c])

    def __setitem__(self, c, val):
        self._m.m[self._r, c] = np.float32(val)


class Matrix:
    __slots__ = ("m",)

    def __init__(self, rows=None):
        self.m = np.identity(4, dtype=np.float32) if rows is None else np.array(rows, dtype=np.float32).reshape(4, 4)

    @staticmethod
    def Identity(n):
        assert n == 4
        return Matrix()

    def __getitem__(self, r):
        return _Row(self, r)

    def inverted(self):
        return Matrix(mat4_inverted(self.m))

    def __matmul__(self, other):
        if isinstance(other, Matrix):
            return Matrix(mat4_mul(self.m, other.m))
        if isinstance(other, Vector):
            return Vector(mat4_mul_vec3(self.m, other.v))
        return NotImplemented

    def to_translation(self):
        return Vector(self.m[:3, 3])

    def copy(self):
        return Matrix(self.m.copy())


class _Vert:
    __slots__ = ("co", "index", "groups")

    def __init__(self, co, index):
        self.co, self.index = Vector(co), index


class _Mesh:
    def __init__(self, xyz):
        self.vertices = [_Vert(p, i) for i, p in enumerate(np.asarray(xyz, dtype=np.float32).reshape(-1, 3))]


class MeshObject:
    """Duck-typed Blender object: .matrix_world, .data.vertices[i].co"""

    def __init__(self, xyz, matrix_world=None, name="obj"):
        self.data = _Mesh(xyz)
        self.matrix_world = Matrix(matrix_world) if matrix_world is not None else Matrix()
        self.name = name


class NearestVertexBVH:
    """find_nearest provider: closest target *vertex* (SURVEY.md D2), via the oracle's KD-tree."""

    def __init__(self, target_xyz, brute=False):
        self.t = _f32(target_xyz).reshape(-1, 3)
        self.kd = None if brute else KDTree(self.t)

    def find_nearest(self, co):
        q = np.array([co[0], co[1], co[2]], dtype=np.float32).reshape(1, 3)
        idx, d2 = (nn_brute(q, self.t) if self.kd is None else self.kd.query(q, nthreads=1))
        i = int(idx[0])
        return Vector(self.t[i]), None, i, math.sqrt(float(d2[0]))


# ---------------------------------------------------------------- reference-style cost structure (bench tier T1)

def make_pairs_python_loop(src, target, mx_align, mx_base, thresh, kd: KDTree, vlist=None, calc_stats=False, timing=None):
    """The reference's make_pairs COST STRUCTURE (functions/general.py:280-321): an interpreter-level loop with
    float32 4x4 transforms on Python objects and one tree query per vertex (stdout writes omitted).  Same results as
    :func:`make_pairs`; used only to time what a per-vertex Python loop costs (tier T1).
    timing (a dict): receives 'loop_s', the seconds spent in the function body the reference times -- the objects the
    reference is HANDED (the align object's vertices, the matrices) are built before the clock starts, and the base
    object's vertices, which make_pairs never touches, are not built at all."""
    import time as _time
    align = MeshObject(src, mx_align)
    mx1, mx2 = align.matrix_world, Matrix(mx_base)
    tgt = _f32(target).reshape(-1, 3)
    _t0 = _time.perf_counter()
    imx1, imx2 = mx1.inverted(), mx2.inverted()
    verts1, verts2, dists = [], [], []
    for vert_ind in (range(len(align.data.vertices)) if vlist is None else vlist):
        vert = align.data.vertices[vert_ind]
        co_find = imx2 @ (mx1 @ vert.co)
        idx, _ = kd.query(co_find.v.reshape(1, 3), nthreads=1)
        co1 = Vector(tgt[int(idx[0])])
        dist = (mx2 @ co_find - mx2 @ co1).length
        if dist < thresh:
            verts1.append(vert.co)
            verts2.append(imx1 @ (mx2 @ co1))
            if calc_stats:
                dists.append(dist)
    A = np.zeros((3, len(verts1)))
    B = np.zeros((3, len(verts1)))
    for i in range(len(verts1)):
        A[0][i], A[1][i], A[2][i] = verts1[i][0], verts1[i][1], verts1[i][2]
        B[0][i], B[1][i], B[2][i] = verts2[i][0], verts2[i][1], verts2[i][2]
    stats = [float(np.mean(dists)), float(np.std(dists))] if calc_stats else None
    if timing is not None:
        timing["loop_s"] = _time.perf_counter() - _t0
    return A, B, stats
