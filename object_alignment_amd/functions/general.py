"""Drop-in mirror of the reference's functions/general.py for the ICP hot path.

Same names, positional signatures and error behaviour as
  /root/reference/functions/general.py:257  make_pairs
  /root/reference/functions/general.py:105  affine_matrix_from_points
plus the alias `calc_target_matrix` that BASELINE.json's north_star names (the reference has no function of
that name -- SURVEY.md D1).  All arithmetic runs in liboa_icp.so on the GPU; this file only adapts arguments.
"""
from __future__ import annotations

import numpy as np

from ..engine import IcpEngine, REF_VALUEERROR

__all__ = ["make_pairs", "affine_matrix_from_points", "calc_target_matrix", "quaternion_matrix", "vector_norm", "GpuBVH",
           "AlignObject", "default_engine", "invalidate_cached_geometry", "close_default_engines"]

_EPS = np.finfo(float).eps * 4.0                  # /root/reference/functions/general.py:30

_default_engines = {}


def default_engine(device: int = 0, devices=None) -> IcpEngine:
    """Process-wide engine used by the free functions and operators (created on first use).  `devices` (a list,
    "all", "0,1,2,3"; default: the OA_DEVICES environment variable) selects a multi-GPU engine that shards the source
    over those GPUs inside the library (oa_create_multi); without it, one context on `device`."""
    import os
    from ..engine import resolve_devices
    devs = resolve_devices(devices if devices is not None else (os.environ.get("OA_DEVICES") or None))
    key = ("multi",) + tuple(devs) if devs is not None and len(devs) > 1 else ("single", devs[0] if devs else int(device))
    eng = _default_engines.get(key)
    if eng is None or getattr(eng, "_h", None) is None:
        eng = IcpEngine(devices=devs) if key[0] == "multi" else IcpEngine(key[1])
        _default_engines[key] = eng
    return eng


def close_default_engines():
    """Destroy the process-wide engines (their device memory, and -- once no context is left -- the library's
    allocation cache, go back to the driver).  They are created again on the next use."""
    for eng in list(_default_engines.values()):
        try:
            eng.close()
        except Exception:
            pass
    _default_engines.clear()


# ------------------------------------------------------------------ object adapters

def _matrix_to_np(m) -> np.ndarray:
    if isinstance(m, np.ndarray):
        return np.ascontiguousarray(m, dtype=np.float32).reshape(4, 4)
    return np.array([[float(m[r][c]) for c in range(4)] for r in range(4)], dtype=np.float32)


def _coords_of(obj) -> np.ndarray:
    """n x 3 float32 local coordinates of a Blender-like object or an AlignObject."""
    if hasattr(obj, "xyz"):
        return obj.xyz
    verts = obj.data.vertices
    if hasattr(verts, "foreach_get"):                        # real bpy mesh: one C call, never cached
        flat = np.empty(len(verts) * 3, dtype=np.float32)
        verts.foreach_get("co", flat)
        return flat.reshape(-1, 3)
    # duck-typed vertices: converted on EVERY call, as the reference re-reads vertices[i].co on every call
    # (/root/reference/functions/general.py:280-284).  Rounds 2-3 remembered the converted array on the object and trusted it
    # while a probe of ~64 vertices still matched -- an in-place edit of any other vertex went unseen.  One
    # flat generator pass is ~0.4 us a vertex; callers with large static clouds hand over an AlignObject (an array).
    n = len(verts)
    return np.fromiter((c for v in verts for c in (v.co[0], v.co[1], v.co[2])), dtype=np.float32, count=3 * n).reshape(-1, 3)


def invalidate_cached_geometry(obj=None):
    """Forget what the host side remembers about vertex lists (converted int64 arrays).  Kept for callers of rounds 2-3;
    nothing is trusted without a full comparison any more, so this is never needed for correctness."""
    _VLIST_CACHE.clear()


def _tris_of(obj):
    """(n, 3) int32 triangles of a Blender-like object / AlignObject, or None when it has no faces."""
    t = getattr(obj, "tris", None)
    if t is not None:
        return np.ascontiguousarray(t, dtype=np.int32).reshape(-1, 3)
    data = getattr(obj, "data", None)
    if data is None:
        return None
    lt = getattr(data, "loop_triangles", None)
    if lt is not None:                                        # real bpy mesh
        if hasattr(data, "calc_loop_triangles"):
            data.calc_loop_triangles()
        if len(lt) == 0:
            return None
        if hasattr(lt, "foreach_get"):
            flat = np.empty(len(lt) * 3, dtype=np.int32)
            lt.foreach_get("vertices", flat)
            return flat.reshape(-1, 3)
        return np.array([[int(v) for v in t.vertices] for t in lt], dtype=np.int32).reshape(-1, 3)
    polys = getattr(data, "polygons", None)
    if polys is not None and len(polys):                      # duck-typed faces: fan triangulation
        out = []
        for p in polys:
            v = [int(i) for i in p.vertices]
            out += [(v[0], v[k], v[k + 1]) for k in range(1, len(v) - 1)]
        return np.array(out, dtype=np.int32).reshape(-1, 3) if out else None
    return None


def evaluated_base(base_obj, context_or_depsgraph=None):
    """The object whose geometry the base search uses: the EVALUATED base object (modifiers applied) when a depsgraph
    is available, as `BVHTree.FromObject(base_obj, context.evaluated_depsgraph_get())` does (operators/icp_align.py:
    52-53); the object itself otherwise (duck-typed objects, arrays)."""
    dg = context_or_depsgraph
    if dg is not None and hasattr(dg, "evaluated_depsgraph_get"):
        try:
            dg = dg.evaluated_depsgraph_get()
        except Exception:
            dg = None
    if dg is not None and hasattr(base_obj, "evaluated_get"):
        try:
            return base_obj.evaluated_get(dg)
        except Exception:
            pass
    return base_obj


class AlignObject:
    """Blender-free stand-in for an object: local coordinates + matrix_world (float32 4x4) (+ optional triangles)."""

    def __init__(self, xyz, matrix_world=None, name="object", tris=None):
        self.tris = None if tris is None else np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        self.matrix_world = (np.identity(4, dtype=np.float32) if matrix_world is None
                             else np.ascontiguousarray(matrix_world, dtype=np.float32).reshape(4, 4))
        self.name = name
        self.type = "MESH"


class GpuBVH:
    """What `base_bvh` is in this build: the base object's geometry resident on the GPU, with its search structures
    (uniform grid + box tree, csrc/oa_grid.hpp, oa_tri.hpp, oa_bvh.hpp).

    Mirrors `BVHTree.FromObject(base_obj, depsgraph)` (operators/icp_align.py:53): an object with faces is searched
    for the closest point on its triangle SURFACE (what `find_nearest` returns); an object without faces -- a point
    cloud -- for its nearest vertex.
    """

    def __init__(self, target_xyz, engine: IcpEngine | None = None, tris=None):
        # (a multi-device engine works too: every GPU pairs its shard of the source, the library merges the pairs back
        #  into vlist order)
        self.engine = engine if engine is not None else default_engine()
        self.target = np.ascontiguousarray(target_xyz, dtype=np.float32).reshape(-1, 3)
        self.tris = tris
        self._src_key = None
        self._bind_target()

    def _bind_target(self):
        # The engine may be shared (the process-wide default): another GpuBVH, IcpAlign.run or the modal operator may
        # have uploaded their own geometry since.  The engine remembers who uploaded last; when it is not this tree,
        # target AND source go up again -- the reference's base_bvh is self-contained, so must this one be.
        if self.engine.target_owner is self:
            return
        if self.tris is not None:   # surface mode: closest point on the triangles, as BVHTree.find_nearest does
            self.engine.set_target_mesh(self.target, self.tris)
        else:                       # no faces (point cloud): nearest vertex
            self.engine.set_target(self.target)
        self.engine.target_owner = self
        self._src_key = None

    @classmethod
    def FromObject(cls, base_obj, depsgraph=None, engine: IcpEngine | None = None, surface=True):
        """Like BVHTree.FromObject: when the object has faces the tree answers with the closest SURFACE point;
        an object without faces (or surface=False) is searched as a vertex cloud.  With a depsgraph the EVALUATED
        object (modifiers applied) is used, as Blender's BVHTree.FromObject(obj, depsgraph) does."""
        obj = evaluated_base(base_obj, depsgraph)
        return cls(_coords_of(obj), engine, _tris_of(obj) if surface else None)

    def find_nearest(self, co, distance=None):
        """`BVHTree.find_nearest(origin, distance)` for ONE point in the base object's local space, as the reference asks of
        its `base_bvh` (functions/general.py:297): (location, normal, index, distance), or four Nones when nothing lies within
        `distance`.  Surface trees answer with the closest point on the nearest triangle, its geometric normal (Blender's
        normal_tri_v3, normalised) and the triangle's index; vertex trees with the nearest vertex, no normal, its index.
        One point through the device searches: the same kernels, the same (d2, index) rule as every other search.  The
        engine's source is this point afterwards -- the next make_pairs binds its own again."""
        p = np.ascontiguousarray(np.asarray([co[0], co[1], co[2]], dtype=np.float32).reshape(1, 3))
        if not np.all(np.isfinite(p)):
            return (None, None, None, None)
        eng = self.engine
        self._bind_target()
        eng.set_source(p, vlist=None, stride=1)
        eng.source_owner = None
        self._src_key = None
        eye = np.identity(4, dtype=np.float32)
        eng.set_matrices(eye, eye)
        idx, _d2, _ = eng.nn_search()
        A, B, _ = eng.make_pairs(1e300, False)                     # identity matrices: B is the closest point itself
        if B.shape[1] != 1:
            return (None, None, None, None)
        loc = B[:, 0].astype(np.float32)
        diff = (p[0] - loc).astype(np.float32)
        dist = float(np.float32(np.sqrt(np.float32(np.dot(diff.astype(np.float64), diff.astype(np.float64))))))
        if distance is not None and not dist <= float(distance):
            return (None, None, None, None)
        i = int(idx[0])
        normal = None
        if self.tris is not None:
            a, b, c = (self.target[int(v)] for v in np.asarray(self.tris).reshape(-1, 3)[i])
            e1, e2 = (a - b).astype(np.float32), (b - c).astype(np.float32)
            n = np.array([e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]], dtype=np.float32)
            d = np.float32(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
            normal = (n * (np.float32(1.0) / np.float32(np.sqrt(d)))).astype(np.float32) if d > np.float32(1e-35) else np.zeros(3, np.float32)
        return (loc, normal, i, dist)

    def _bind_source(self, xyz, vlist, sample):
        # re-upload only when something changed.  The reference re-reads vertices[i].co on every call
        # (functions/general.py:284), so "changed" has to mean the CONTENT: a hash over the bytes of the coordinates and of
        # the vertex list (an array edited in place keeps its id; sums -- round 2's fingerprint -- survive a permutation).
        self._bind_target()
        key = (xyz.shape, _content_hash(xyz), None if vlist is None else (len(vlist), _content_hash(vlist)), sample)
        if key != self._src_key or self.engine.source_owner is not self:
            self.engine.set_source(xyz, vlist=vlist, stride=sample)
            self.engine.source_owner = self
            self._src_key = key


def _content_hash(a) -> int:
    """64-bit hash of an array's bytes: xxh3 when the module is there (~10 GB/s: 1 ms for a 1M-vertex mesh), else
    blake2b (~1 GB/s).  NaN-safe and order-sensitive by construction (bytes, not values); an empty array hashes its shape
    (a zero-size 2-D view cannot be cast to bytes)."""
    a = np.ascontiguousarray(a)
    if a.size == 0:
        return hash(("empty", a.shape, a.dtype.str)) & 0xFFFFFFFFFFFFFFFF
    buf = a.reshape(-1).view(np.uint8)
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(buf)
    except ImportError:
        import hashlib
        return int.from_bytes(hashlib.blake2b(buf, digest_size=8).digest(), "little")


_VLIST_CACHE = {}


def _vlist_array(vlist):
    """int64 array of a vertex list.  The operators hand over a Python list (as the reference builds it) and reuse it
    every iteration: converting a million-element list costs ~60 ms, so the converted array is remembered per list object
    -- together with a COPY of the list, and it is only reused when the list still equals that copy, element for element
    (`list == list` runs in C with an identity shortcut per element: ~3-5 ms for a million indices; the reference re-reads
    vlist on every call, /root/reference/functions/general.py:274-284, so any in-place edit has to be seen.  Round 3
    compared 32 probed elements: VERDICT r3)."""
    if isinstance(vlist, np.ndarray):
        return np.ascontiguousarray(vlist, dtype=np.int64)
    hit = _VLIST_CACHE.get(id(vlist))
    if hit is not None and hit[2] is vlist and hit[0] == vlist:   # (the list itself is kept alive: its id cannot be recycled)
        return hit[1]
    arr = np.ascontiguousarray(vlist, dtype=np.int64)
    if len(_VLIST_CACHE) > 8:
        _VLIST_CACHE.clear()
    _VLIST_CACHE[id(vlist)] = (list(vlist), arr, vlist)
    return arr


def quaternion_matrix(quaternion):
    """Homogeneous 4x4 rotation matrix of a quaternion (w, x, y, z), any length; the identity for a (near-)zero quaternion.
    Same contract and the same float64 operations as the reference's helper (functions/general.py:38-62, pinned by
    tests/golden/general_helpers.npz): a dozen flops on the host -- the Horn branch of the solve itself runs on the device."""
    q = np.array(quaternion, dtype=np.float64).reshape(-1)[:4]
    nq = float(np.dot(q, q))
    if nq < _EPS:
        return np.identity(4)
    q = q * np.sqrt(2.0 / nq)
    w, x, y, z = q
    ww = np.identity(4)
    ww[0, :3] = (1.0 - y * y - z * z, x * y - z * w, x * z + y * w)
    ww[1, :3] = (x * y + z * w, 1.0 - x * x - z * z, y * z - x * w)
    ww[2, :3] = (x * z - y * w, y * z + x * w, 1.0 - x * x - y * y)
    return ww


def vector_norm(data, axis=None, out=None):
    """Euclidean length of an array along `axis` (functions/general.py:66-102): a float for 1-D input without `out`, else an
    array (at least 1-D); with `out` the result is written there and None returned.  The input is never modified."""
    sq = np.array(data, dtype=np.float64)
    if out is None and sq.ndim == 1:
        return float(np.sqrt(np.dot(sq, sq)))
    sq *= sq
    if out is None:
        res = np.atleast_1d(np.sum(sq, axis=axis))
        np.sqrt(res, res)
        return res
    np.sum(sq, axis=axis, out=out)
    np.sqrt(out, out)
    return None


def make_pairs(align_obj, base_obj, base_bvh, vlist, thresh, sample=0, calc_stats=False):
    """Same contract as the reference's make_pairs (functions/general.py:257-329).

    vlist: vertex indices of align_obj to use; returns (A, B, d_stats) with A, B float64[3, K] in
    align_obj LOCAL space, or None when thresh <= 0 (the reference falls off the end, :277).
    """
    if not thresh > 0:
        return None
    if base_bvh is None:
        base_bvh = GpuBVH.FromObject(base_obj)
    if not isinstance(base_bvh, GpuBVH):
        raise TypeError("base_bvh must be an object_alignment_amd GpuBVH (GpuBVH.FromObject(base_obj))")
    xyz = _coords_of(align_obj)
    vl = _vlist_array(vlist)
    base_bvh._bind_source(xyz, vl, int(sample))
    eng = base_bvh.engine
    eng.set_matrices(_matrix_to_np(align_obj.matrix_world), _matrix_to_np(base_obj.matrix_world))
    return eng.make_pairs(thresh, calc_stats)


def affine_matrix_from_points(v0, v1, shear=True, scale=True, usesvd=True):
    """Same contract as the reference's affine_matrix_from_points (functions/general.py:105-217), every branch:
    shear=True (the signature's default) is the affine estimate of Hartley & Zisserman (:168-178); shear=False the
    rigid (scale=False) or similarity (scale=True) transform through the SVD of the covariance (:179-190, :208-212).
    v0, v1: (ndims, K), ndims >= 2 (the device path is laid out for up to 64 dimensions; the reference takes any, no caller
    passes anything but 3).  The 3-D rigid / similarity case -- the one the ICP operators use -- takes the 24-sum solve of
    the loop (oa_kabsch); everything else the general device path (oa_affine_from_points)."""
    v0 = np.asarray(v0, dtype=np.float64)        # the reference copies (:146-147) because it centres in place; nothing
    v1 = np.asarray(v1, dtype=np.float64)        # is modified here
    if v0.ndim != 2 or v1.ndim != 2:
        raise ValueError(REF_VALUEERROR)
    ndims = v0.shape[0]
    if ndims < 2 or v0.shape[1] < ndims or v0.shape != v1.shape:       # :150
        raise ValueError(REF_VALUEERROR)                                # :157
    if ndims > 64:
        raise ValueError("affine_matrix_from_points: at most 64 dimensions on the device path (got %d)" % ndims)
    # usesvd=False (3-D, no shear): Horn's quaternion branch (:191-206) -- the eigenvector of the 4 x 4 matrix N, on the device
    # (csrc/oa_kernels.hpp: rotation_from_covariance_horn).  As in the reference, usesvd is ignored when shear is set or ndims != 3.
    if ndims == 3 and not shear:
        return default_engine().kabsch(v0, v1, scale=bool(scale), horn=not usesvd)
    return default_engine().affine_from_points(v0, v1, shear=bool(shear), scale=bool(scale))


def calc_target_matrix(A, B, scale=False):
    """north_star's name for the solve: affine_matrix_from_points(A, B, shear=False, scale=scale, usesvd=True)."""
    return affine_matrix_from_points(A, B, shear=False, scale=scale, usesvd=True)
