"""Landmark pre-alignment (`object.align_picked_points`).

The reference's operator (operators/align_pick_points.py) is mostly a two-viewport picking UI; the arithmetic is its
`align_obj` method (:415-465): pair the picked points, solve a rigid or similarity transform with
`affine_matrix_from_points`, right-multiply the align object's float32 `matrix_world`, drag the `m_*` objects along.
That arithmetic is what lives here, on the same engine entry point the ICP loop uses (`oa_kabsch`); the picking UI
(ray casts, split viewports, GPU overlay drawing) is out of scope.

Conventions kept from the reference:
  * picks on the align object are stored in its LOCAL space (:162); picks on the base object are stored in the
    ALIGN object's local space, `align.matrix_world.inverted() @ base.matrix_world @ hit` (:184);
  * unequal pick counts are cut to the shorter list (:417-422);
  * `align_meth` '0' = rigid, '1' = rotation + translation + uniform scale (:442-445);
  * fewer than three pairs: the solver's ValueError("input arrays are of wrong shape or type") propagates.
"""
from __future__ import annotations

import numpy as np

from .. import _hostmath
from ..functions.general import _matrix_to_np, affine_matrix_from_points
from .icp_align import _OperatorBase, _assign_matrix, _bpy, get_addon_preferences


def base_pick_to_align_local(mx_align, mx_base, hit):
    """A pick on the base object (base-local `hit`) expressed in the align object's local space
    (operators/align_pick_points.py:184), with mathutils' float32 rounding: (A^-1 @ B) @ hit."""
    both = _hostmath.mat4_mul(_hostmath.mat4_inverted(_matrix_to_np(mx_align)), _matrix_to_np(mx_base))
    return _hostmath.mat4_mul_vec3(both, np.asarray(hit, dtype=np.float32))


class LandmarkAlign:
    """Blender-free form of `align_obj` (operators/align_pick_points.py:415-465)."""

    def __init__(self, settings=None):
        self.settings = settings if settings is not None else get_addon_preferences()

    def solve(self, align_points, base_points):
        """(M float64 4x4, new_mat float32 4x4) from paired picks, both in the align object's local space."""
        n = min(len(align_points), len(base_points))                      # :417-422
        A = np.zeros((3, n))
        B = np.zeros((3, n))
        for i in range(n):                                                # :427-432 (float32 picks widened to float64)
            A[:, i] = np.asarray(align_points[i], dtype=np.float32)[:3]
            B[:, i] = np.asarray(base_points[i], dtype=np.float32)[:3]
        with_scale = str(self.settings.align_meth) == '1'                 # :442-445
        M = affine_matrix_from_points(A, B, shear=False, scale=with_scale, usesvd=True)
        return M, M.astype(np.float32)                                    # new_mat[n][m] = M[n][m] (:450-453)

    def apply(self, matrix_world, new_mat):
        """matrix_world @ new_mat in float32 (:457)."""
        return _hostmath.mat4_mul(_matrix_to_np(matrix_world), new_mat)


class OBJECT_OT_align_pick_points(_OperatorBase):
    """Align two objects from landmark points picked on each of them"""
    bl_idname = "object.align_picked_points"
    bl_label = "Align: Picked Points"
    bl_options = {'REGISTER', 'UNDO'}

    @classmethod
    def poll(cls, context):
        active = getattr(context, "object", None)
        return len(context.selected_objects) == 2 and bool(active) and active.type == 'MESH'

    def begin(self, context):
        """What `invoke` sets up besides the UI (:262-361): the two objects and the empty pick lists."""
        self.obj_align = context.object
        self.obj_base = next(o for o in context.selected_objects if o != self.obj_align)
        self.align_points = []
        self.base_points = []
        return self

    def pick_align(self, hit_local):
        self.align_points.append(np.asarray(hit_local, dtype=np.float32)[:3])

    def pick_base(self, hit_base_local):
        self.base_points.append(base_pick_to_align_local(self.obj_align.matrix_world, self.obj_base.matrix_world,
                                                         hit_base_local))

    def undo_pick(self, on_align):
        pts = self.align_points if on_align else self.base_points        # :190-195
        if pts:
            pts.pop()

    def align_obj(self, context):
        settings = get_addon_preferences()
        M, new_mat = LandmarkAlign(settings).solve(self.align_points, self.base_points)
        n = min(len(self.align_points), len(self.base_points))
        self.align_points, self.base_points = self.align_points[:n], self.base_points[:n]
        _assign_matrix(self.obj_align, _hostmath.mat4_mul(_matrix_to_np(self.obj_align.matrix_world), new_mat))
        if settings.take_m_with:                                          # :458-462
            scene = getattr(context, "scene", None) or getattr(getattr(_bpy, "context", None), "scene", None)
            for obj in (scene.objects if scene is not None else ()):
                if obj.name[:2] == "m_":
                    _assign_matrix(obj, _hostmath.mat4_mul(_matrix_to_np(obj.matrix_world), new_mat))
                    if hasattr(obj, "update_tag"):
                        obj.update_tag()
        if hasattr(self.obj_align, "update_tag"):
            self.obj_align.update_tag()
        if hasattr(context, "view_layer") and hasattr(context.view_layer, "update"):
            context.view_layer.update()
        self.last_M = M
        return M
