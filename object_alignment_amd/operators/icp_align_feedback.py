"""Modal form of the ICP operator (`object.align_icp_redraw`): the reference's operators/icp_align_feedback.py,
rebuilt on the engine's single-iteration entry point `oa_iterate`.

Pacing is the reference's: a window timer drives `modal`; each TIMER event runs a burst of `redraw_frequency`
device iterations and lets the viewport redraw; the run ends when the translation ring reports convergence or the
iteration budget is spent.  Deliberate choices about the reference's quirks:

  * kept -- the budget test is inclusive (icp_align_feedback.py:113 compares with `<=`), so up to
    `icp_iterations + 1` iterations run;
  * kept -- convergence is only evaluated when distance statistics are being computed (`use_target`), using a
    five-slot ring of translation magnitudes initialised to twice the target (:93-94 of icp_align.py, :98-99 here);
  * `execute` (EXEC_DEFAULT, Redo) is the non-modal loop of the reference's :130-235 -- the same code as OBJECT_OT_icp_align's;
  * fixed -- the reference's `iterate` refers to an unbound name (`take_m_with`, :267) and therefore raises
    NameError on its first tick; the value read from the preferences during `invoke` is used instead.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .. import _hostmath
from ..functions.general import _coords_of, _matrix_to_np, _tris_of, default_engine, evaluated_base
from .icp_align import _OperatorBase, _assign_matrix, _bpy, execute_alignment, get_addon_preferences, vlist_for_engine

RING = 5


@dataclass
class _Run:
    """Everything one modal run needs, captured once in invoke()."""
    align_obj: object
    base_obj: object
    thresh: float
    target_d: float
    budget: int
    burst: int
    use_target: bool
    with_scale: bool
    drag_m_objects: bool
    stride: int = 1
    done: int = 0
    converged: bool = False
    last_stats: dict | None = None
    ring_t: list = field(default_factory=list)
    ring_r: list = field(default_factory=list)


class OBJECT_OT_icp_align_feedback(_OperatorBase):
    """ICP alignment of two objects with a viewport redraw every few iterations (slower, easier to diagnose)"""
    bl_idname = "object.align_icp_redraw"
    bl_label = "ICP Align Redraw"
    bl_options = {'REGISTER', 'UNDO'}

    _timer = None
    _run: _Run | None = None

    # ------------------------------------------------------------------ Blender operator protocol
    @classmethod
    def poll(cls, context):
        active = getattr(context, "object", None)
        return len(context.selected_objects) == 2 and bool(active) and active.type == 'MESH'

    def execute(self, context):
        # what bpy.ops.object.align_icp_redraw('EXEC_DEFAULT') and Redo run: the whole loop, no timer, no redraws -- the reference's
        # second copy of OBJECT_OT_icp_align.execute (operators/icp_align_feedback.py:130-235; rounds 1-5 returned CANCELLED here)
        return execute_alignment(self, context)

    def invoke(self, context, event):
        prefs = get_addon_preferences()
        align = context.object
        base = next(o for o in context.selected_objects if o != align)
        if not prefs.min_start > 0:   # make_pairs would hand back None and the unpack in iterate() would raise
            raise TypeError("cannot unpack non-iterable NoneType object")
        run = _Run(align_obj=align, base_obj=base, thresh=prefs.min_start, target_d=prefs.target_d,
                   budget=prefs.icp_iterations, burst=prefs.redraw_frequency, use_target=bool(prefs.use_target),
                   with_scale=(prefs.align_meth == '1'), drag_m_objects=bool(prefs.take_m_with))
        run.ring_t = [run.target_d * 2.0] * RING
        run.ring_r = [None] * RING
        run.stride = round(1 / prefs.sample_fraction)
        self._run = run
        self._upload(run, stride=run.stride, context=context)
        try:
            align.rotation_mode = 'QUATERNION'
        except Exception:
            pass
        wm = getattr(context, "window_manager", None)
        if wm is not None:
            self._timer = wm.event_timer_add(time_step=0.01, window=getattr(context, "window", None))
            wm.modal_handler_add(self)
        return {'RUNNING_MODAL'}

    def modal(self, context, event):
        kind = event.type
        if kind in ('RIGHTMOUSE', 'ESC'):
            return self.cancel(context)
        if kind != 'TIMER':
            return {'PASS_THROUGH'}
        area = getattr(context, "area", None)
        if area is not None:
            area.tag_redraw()
        run = self._run
        for _ in range(run.burst):
            if run.converged or run.done > run.budget:
                return self.finish(context)
            self.iterate(context)
            run.done += 1
        return {'RUNNING_MODAL'}

    def cancel(self, context):
        self._drop_timer(context)
        return {"CANCELLED"}

    def finish(self, context):
        self._drop_timer(context)
        return {'FINISHED'}

    # ------------------------------------------------------------------ one device iteration
    def iterate(self, context):
        run = self._run
        # the engine is the process-wide one: a plain ICP call or a make_pairs between two timer ticks may have replaced
        # its geometry and matrices -- then this run's state goes up again (from the pose reached so far) before it steps
        eng = self.engine
        if eng.target_owner is not self or eng.source_owner is not self:
            self._upload(run, stride=run.stride, context=context)
        M, stats = self.engine.iterate(thresh=run.thresh, target_d=run.target_d, use_target=run.use_target,
                                       with_scale=run.with_scale)
        _assign_matrix(run.align_obj, self.engine.matrix_world())
        if run.drag_m_objects:
            self._apply_to_m_objects(context, M.astype(np.float32))
        if hasattr(run.align_obj, "update_tag"):
            run.align_obj.update_tag()
        run.last_stats = stats if run.use_target else None
        if run.last_stats is not None:
            slot = run.done % RING
            run.ring_t[slot] = stats["translation"]
            run.ring_r[slot] = stats["rot_angle"]
            run.converged = max(run.ring_t) < run.target_d

    # ------------------------------------------------------------------ helpers
    def _upload(self, run, stride, context=None):
        eng = self.engine = default_engine(devices=getattr(get_addon_preferences(), "devices", None))
        base_geo = evaluated_base(run.base_obj, context)         # the evaluated mesh, as the reference's BVH (:57)
        tris = _tris_of(base_geo)
        if tris is None:
            eng.set_target(_coords_of(base_geo))
        else:
            eng.set_target_mesh(_coords_of(base_geo), tris)
        eng.set_source(_coords_of(run.align_obj), vlist=vlist_for_engine(run.align_obj), stride=stride)
        eng.set_matrices(_matrix_to_np(run.align_obj.matrix_world), _matrix_to_np(run.base_obj.matrix_world))
        eng.target_owner = eng.source_owner = self                  # GpuBVH and this operator check these before they trust the engine

    @staticmethod
    def _apply_to_m_objects(context, new_mat):
        scene = getattr(context, "scene", None) or getattr(getattr(_bpy, "context", None), "scene", None)
        for obj in (scene.objects if scene is not None else ()):
            if obj.name.startswith("m_"):
                _assign_matrix(obj, _hostmath.mat4_mul(_matrix_to_np(obj.matrix_world), new_mat))
                if hasattr(obj, "update_tag"):
                    obj.update_tag()

    def _drop_timer(self, context):
        wm = getattr(context, "window_manager", None)
        if wm is not None and self._timer is not None:
            wm.event_timer_remove(self._timer)
            self._timer = None

    # attribute views used by callers/tests
    @property
    def converged(self):
        return bool(self._run and self._run.converged)

    @property
    def total_iters(self):
        return self._run.done if self._run else 0
