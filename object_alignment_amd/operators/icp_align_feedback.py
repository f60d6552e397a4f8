"""Mirror of the reference's modal operator operators/icp_align_feedback.py (`object.align_icp_redraw`).

Same pacing as the reference: a window timer fires `modal`, every TIMER event runs `redraw_frequency` iterations
(one `oa_iterate` each, i.e. the engine's per-tick step), the viewport redraws in between, and the loop ends on
convergence or after the iteration budget.  Two quirks of the reference are kept on purpose and one is not:

  * kept: `modal` tests `total_iters <= iters` (icp_align_feedback.py:113), so it runs `iters + 1` iterations;
  * kept: the convergence ring, d_stats gating and the `m_*` objects follow the same order as `iterate` (:250-288);
  * not kept: the reference's `iterate` reads an undefined name `take_m_with` (:267) and raises NameError on every
    tick; here `self.take_m_with` (read from the preferences in `invoke`, :101) is used, which is the evident intent.
"""
from __future__ import annotations

import numpy as np

from ..functions.general import _coords_of, _matrix_to_np, _tris_of, default_engine
from .icp_align import _OperatorBase, _assign_matrix, _bpy, build_vlist, get_addon_preferences


class OBJECT_OT_icp_align_feedback(_OperatorBase):
    """Uses ICP alignment to iteratevely aligne two objects and redraws every n iterations.  Slower but better to diagnose errors"""
    bl_idname = "object.align_icp_redraw"
    bl_label = "ICP Align Redraw"
    bl_options = {'REGISTER', 'UNDO'}

    _timer = None

    @classmethod
    def poll(cls, context):
        condition_1 = len(context.selected_objects) == 2
        condition_2 = context.object and context.object.type == 'MESH'
        return condition_1 and condition_2

    # ---- invoke: icp_align_feedback.py:47-102
    def invoke(self, context, event):
        wm = getattr(context, "window_manager", None)
        if wm is not None:
            self._timer = wm.event_timer_add(time_step=0.01, window=getattr(context, "window", None))
            wm.modal_handler_add(self)
        settings = get_addon_preferences()
        self.align_meth = settings.align_meth
        self.align_obj = context.object
        self.base_obj = [obj for obj in context.selected_objects if obj != self.align_obj][0]
        try:
            self.align_obj.rotation_mode = 'QUATERNION'
        except Exception:
            pass
        self.vlist = build_vlist(self.align_obj)
        self.thresh = settings.min_start
        self.sample_fraction = settings.sample_fraction
        self.iters = settings.icp_iterations
        self.target_d = settings.target_d
        self.use_target = settings.use_target
        self.take_m_with = settings.take_m_with
        self.sample_factor = round(1 / self.sample_fraction)
        self.redraw_frequency = settings.redraw_frequency
        self.total_iters = 0
        self.converged = False
        self.conv_t_list = [self.target_d * 2] * 5
        self.conv_r_list = [None] * 5
        self.d_stats = None
        if not self.thresh > 0:
            raise TypeError("cannot unpack non-iterable NoneType object")   # make_pairs would return None (:252)
        self.engine = default_engine()
        tris = _tris_of(self.base_obj)
        if tris is not None:
            self.engine.set_target_mesh(_coords_of(self.base_obj), tris)
        else:
            self.engine.set_target(_coords_of(self.base_obj))
        self.engine.set_source(_coords_of(self.align_obj), vlist=self.vlist, stride=self.sample_factor)
        self.engine.set_matrices(_matrix_to_np(self.align_obj.matrix_world), _matrix_to_np(self.base_obj.matrix_world))
        return {'RUNNING_MODAL'}

    # ---- modal: icp_align_feedback.py:104-122
    def modal(self, context, event):
        if event.type in {'RIGHTMOUSE', 'ESC'}:
            return self.cancel(context)
        if event.type == 'TIMER':
            area = getattr(context, "area", None)
            if area is not None:
                area.tag_redraw()
            for _ in range(0, self.redraw_frequency):
                if self.total_iters <= self.iters and not self.converged:
                    self.iterate(context)
                    self.total_iters += 1
                else:
                    return self.finish(context)
            return {'RUNNING_MODAL'}
        return {'PASS_THROUGH'}

    def execute(self, context):
        return {"CANCELLED"}

    def cancel(self, context):
        wm = getattr(context, "window_manager", None)
        if wm is not None and self._timer is not None:
            wm.event_timer_remove(self._timer)
        return {"CANCELLED"}

    # ---- iterate: icp_align_feedback.py:250-288 (one device iteration)
    def iterate(self, context):
        from .. import _hostmath
        M, st = self.engine.iterate(thresh=self.thresh, target_d=self.target_d, use_target=self.use_target,
                                    with_scale=(self.align_meth == '1'))
        new_mat = M.astype(np.float32)
        _assign_matrix(self.align_obj, self.engine.matrix_world())
        if self.take_m_with:
            scene = getattr(context, "scene", None) or getattr(getattr(_bpy, "context", None), "scene", None)
            for obj in (scene.objects if scene is not None else []):
                if obj.name[:2] == "m_":
                    _assign_matrix(obj, _hostmath.mat4_mul(_matrix_to_np(obj.matrix_world), new_mat))
                    if hasattr(obj, "update_tag"):
                        obj.update_tag()
        if hasattr(self.align_obj, "update_tag"):
            self.align_obj.update_tag()
        self.d_stats = [st["mean_dist"], st["std_dist"]] if self.use_target else None
        if self.d_stats:
            i = self.total_iters % 5
            self.conv_t_list[i] = st["translation"]
            self.conv_r_list[i] = st["rot_angle"]
            if all(d < self.target_d for d in self.conv_t_list):
                self.converged = True

    def finish(self, context):
        wm = getattr(context, "window_manager", None)
        if wm is not None and self._timer is not None:
            wm.event_timer_remove(self._timer)
        return {'FINISHED'}
