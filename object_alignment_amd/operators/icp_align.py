"""Drop-in mirror of the reference's operators/icp_align.py (OBJECT_OT_icp_align).

`IcpAlign.run` is the Blender-free form of `execute` (operators/icp_align.py:82-161): the whole
`while n < iters and not converged` loop runs device-resident in liboa_icp.so (oa_run).
`OBJECT_OT_icp_align` keeps the operator surface (bl_idname / bl_label / bl_options / poll / execute) and
adapts duck-typed or real Blender objects to it.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from ..engine import IcpEngine, RunResult
from ..functions.general import _coords_of, _matrix_to_np, _tris_of, default_engine, evaluated_base

try:                                              # inside Blender the operator registers as usual
    import bpy as _bpy                            # noqa: F401
    from bpy.types import Operator as _OperatorBase
except Exception:                                 # outside Blender it is a plain class
    _bpy = None
    _OperatorBase = object


@dataclass
class IcpSettings:
    """Names and defaults of the add-on preferences the loop reads (lib/preferences.py:31-72)."""
    icp_iterations: int = 50
    redraw_frequency: int = 10
    use_sample: bool = False          # never consulted by the reference either
    sample_fraction: float = 0.5
    min_start: float = 0.5
    target_d: float = 0.01
    use_target: bool = True
    take_m_with: bool = False
    align_meth: str = "0"             # '0' RIGID, '1' ROT_LOC_SCALE
    # not a preference of the reference: which GPUs the loop may use.  None = one GPU (or what the OA_DEVICES environment
    # variable lists); "all" / [0, 1, ...] = the source cloud is sharded over those GPUs inside the library
    devices: object = None


_prefs = IcpSettings()


def get_addon_preferences() -> IcpSettings:
    """Stand-in for functions/common/blender.py:48-55; returns the process-wide settings object."""
    return _prefs


def build_vlist(align_obj):
    """Vertex indices the operator aligns with, from the object's `icp_include` / `icp_exclude` vertex groups
    (semantics of operators/icp_align.py:56-80): an include group wins and keeps memberships heavier than 0.9;
    otherwise an exclude group drops every vertex whose membership weighs 0.1 or more; no group = every vertex."""
    mesh = getattr(align_obj, "data", None)
    if mesh is None:
        return list(range(len(_coords_of(align_obj))))
    group_index = {g.name: g.index for g in (getattr(align_obj, "vertex_groups", None) or ())}

    def weights_in(vertex, gi):
        return [m.weight for m in vertex.groups if m.group == gi]

    if "icp_include" in group_index:
        gi = group_index["icp_include"]
        # one entry per qualifying membership, exactly as the reference appends inside its inner loop
        return [v.index for v in mesh.vertices for w in weights_in(v, gi) if w > 0.9]
    if "icp_exclude" in group_index:
        gi = group_index["icp_exclude"]
        keep = []
        for v in mesh.vertices:
            ws = weights_in(v, gi)
            if not ws:
                keep.append(v.index)
            else:
                keep.extend(v.index for w in ws if w < 0.1)
        return keep
    return [v.index for v in mesh.vertices]


def vlist_for_engine(align_obj):
    """build_vlist() for the engine: None when the object has neither an `icp_include` nor an `icp_exclude` group --
    the engine then takes every vertex itself, and a million-element Python list is neither built nor converted."""
    names = {g.name for g in (getattr(align_obj, "vertex_groups", None) or ())}
    if "icp_include" in names or "icp_exclude" in names:
        return build_vlist(align_obj)
    return None


def vlist_from_weights(n_verts, include=None, exclude=None):
    """Array form of the same mask: include / exclude are None or iterables of (vertex_index, weight)."""
    if include is not None:
        return [int(v) for v, w in sorted(include, key=lambda t: t[0]) if np.float32(w) > 0.9]
    if exclude is not None:
        member = {int(v): np.float32(w) for v, w in exclude}
        return [v for v in range(n_verts) if v not in member or member[v] < 0.1]
    return list(range(n_verts))


class IcpAlign:
    """The ICP loop of OBJECT_OT_icp_align.execute without Blender."""

    def __init__(self, settings: IcpSettings | None = None, engine: IcpEngine | None = None):
        self.settings = settings if settings is not None else get_addon_preferences()
        self.engine = engine if engine is not None else default_engine(devices=getattr(self.settings, "devices", None))

    def run(self, source_xyz, target_xyz, mx_align, mx_base, vlist=None, early_exit=True,
            target_tris=None) -> RunResult:
        """target_tris: (n, 3) triangles of the base mesh -> closest point on the surface (the reference's BVH
        semantics); None -> nearest target vertex (point-cloud targets, BASELINE's configurations)."""
        s = self.settings
        thresh = s.min_start                                   # :83
        factor = round(1 / s.sample_fraction)                  # :89  (ZeroDivisionError at 0, as the reference)
        if not thresh > 0:
            # make_pairs returns None and `(A, B, d_stats) = None` raises   (:101, general.py:277)
            raise TypeError("cannot unpack non-iterable NoneType object")
        eng = self.engine
        if target_tris is not None:
            eng.set_target_mesh(target_xyz, target_tris)
        else:
            eng.set_target(target_xyz)
        eng.set_source(source_xyz, vlist=vlist, stride=factor)
        eng.set_matrices(mx_align, mx_base)
        return eng.run(iters=s.icp_iterations, thresh=thresh, target_d=s.target_d, use_target=s.use_target,
                       with_scale=(s.align_meth == "1"), early_exit=early_exit)


def report_lines(res: RunResult, settings, seconds=None):
    """What the reference's execute prints when its loop ends (operators/icp_align.py:145-160), from the device's report: the
    convergence line or 'Maxed out iterations', then the last translation, the last d_stats and the mean of the 5-slot
    rotation ring -- same wording, same %f formats."""
    lines = []
    if settings.use_target and res.iters_done > 0:
        lines.append('Converged in %s iterations' % str(res.iters_done) if res.converged else 'Maxed out iterations')   # :145 / :154
        lines.append('Final Translation: %f ' % res.last_translation)                                                    # :146 / :155
        lines.append('Final Avg Dist: %f' % res.mean_dist)                                                               # :147 / :156
        lines.append('Final St Dev %f' % res.std_dist)                                                                   # :148 / :157
        lines.append('Avg last 5 rotation angle: %f' % res.mean_rot_angle)                                               # :149 / :158
    if seconds is not None:
        lines.append('Aligned obj in %f sec' % seconds)                                                                  # :160
    return lines


def _assign_matrix(obj, new_np):
    old = obj.matrix_world
    if isinstance(old, np.ndarray):
        obj.matrix_world = np.array(new_np, dtype=np.float32)
        return
    try:
        obj.matrix_world = type(old)([[float(x) for x in row] for row in new_np])
    except Exception:
        obj.matrix_world = np.array(new_np, dtype=np.float32)


def execute_alignment(op, context):
    """The body of `execute` of BOTH ICP operators: the reference's OBJECT_OT_icp_align.execute (operators/icp_align.py:47-161)
    and OBJECT_OT_icp_align_feedback.execute (operators/icp_align_feedback.py:130-235) are the same loop -- the second lacks
    the per-iteration timing prints, nothing else -- so both classes run this."""
    settings = get_addon_preferences()
    align_obj = context.object
    base_obj = next(o for o in context.selected_objects if o != align_obj)
    try:
        align_obj.rotation_mode = 'QUATERNION'
    except Exception:
        pass
    import time
    start = time.time()                                     # :51
    vlist = vlist_for_engine(align_obj)
    base_geo = evaluated_base(base_obj, context)            # BVHTree.FromObject(base_obj, depsgraph)  (:52-53)
    failure = None
    try:
        res = IcpAlign(settings).run(_coords_of(align_obj), _coords_of(base_geo),
                                     _matrix_to_np(align_obj.matrix_world), _matrix_to_np(base_obj.matrix_world),
                                     vlist=vlist, target_tris=_tris_of(base_geo))
    except ValueError as exc:
        # fewer than 3 pairs in iteration n: the reference has already applied iterations 0..n-1 to align_obj and
        # the m_* objects when affine_matrix_from_points raises (:109 after :121-127 of the earlier passes)
        res = getattr(exc, "partial", None)
        if res is None:
            raise
        failure = exc
    _assign_matrix(align_obj, res.matrix_world)
    if settings.take_m_with:                                # :123-127, replayed in iteration order
        from .. import _hostmath
        scene = getattr(context, "scene", None) or getattr(getattr(_bpy, "context", None), "scene", None)
        for obj in (scene.objects if scene is not None else []):
            if obj.name[:2] == "m_":
                m = _matrix_to_np(obj.matrix_world)
                for new_mat in res.step_new:
                    m = _hostmath.mat4_mul(m, new_mat)
                _assign_matrix(obj, m)
                if hasattr(obj, "update_tag"):
                    obj.update_tag()
    if hasattr(align_obj, "update_tag"):
        align_obj.update_tag()
    if hasattr(context, "view_layer") and hasattr(context.view_layer, "update"):
        context.view_layer.update()
    op.last_result = res
    # the reference prints its summary (:145-160); here it also goes to Blender's info area
    op.last_report = report_lines(res, settings, time.time() - start)
    for line in op.last_report:
        print(line)
        if hasattr(op, "report"):
            try:
                op.report({'INFO'}, line)
            except Exception:
                pass
    if failure is not None:
        raise failure
    return {'FINISHED'}


class OBJECT_OT_icp_align(_OperatorBase):
    """Iterative-closest-point alignment of the active object onto the other selected object"""
    bl_idname = "object.align_icp"
    bl_label = "ICP Align"
    bl_options = {'REGISTER', 'UNDO'}

    @classmethod
    def poll(cls, context):
        # exactly two selected objects, the active one a mesh (operators/icp_align.py:41-45)
        return len(context.selected_objects) == 2 and context.object.type == 'MESH'

    def execute(self, context):
        return execute_alignment(self, context)
