"""Optional Blender shell: makes the engine an actual add-on.

Only the ICP part of the reference's UI is mirrored -- the two operators this build implements
(`object.align_icp`, `object.align_icp_redraw`), the preferences the loop reads (same property names, defaults
and ranges as /root/reference/lib/preferences.py:31-72) and a sidebar panel with the two buttons
(/root/reference/ui/__init__.py:29-110 shows more: pick-points, include/exclude painting, updater; those are
out of scope).  Importing this module outside Blender is harmless: `register()` raises.

Install: zip the `object_alignment_amd` package (with liboa_icp.so inside) and enable it as an add-on, or
`import object_alignment_amd.blender_addon as a; a.register()` from Blender's Python console.
"""
from __future__ import annotations

bl_info = {
    "name": "Object Alignment (MI355X ICP engine)",
    "author": "oa-icp",
    "version": (0, 1, 0),
    "blender": (3, 2, 2),
    "location": "View3D > Sidebar > Alignment",
    "description": "ICP alignment of two objects; the iterate loop runs on an AMD MI355X through liboa_icp.so",
    "category": "Object",
}

try:
    import bpy
    from bpy.props import BoolProperty, EnumProperty, FloatProperty, IntProperty
    from bpy.types import AddonPreferences, Panel
except Exception:          # not inside Blender
    bpy = None

_classes = []

if bpy is not None:
    from .operators import icp_align as _icp_align
    from .operators.icp_align import OBJECT_OT_icp_align
    from .operators.icp_align_feedback import OBJECT_OT_icp_align_feedback

    class ObjectAlignmentPreferences(AddonPreferences):
        bl_idname = __package__

        icp_iterations: IntProperty(name="ICP Iterations", default=50)
        redraw_frequency: IntProperty(
            name="Redraw Iterations", description="Iterations per viewport redraw in the modal operator", default=10)
        use_sample: BoolProperty(name="Use Sample", description="Kept for compatibility; not read by the loop", default=False)
        sample_fraction: FloatProperty(
            name="Sample Fraction", description="Stride through the vertex list is round(1 / fraction)",
            default=0.5, min=0, max=1)
        min_start: FloatProperty(
            name="Minimum Starting Dist", description="World-space distance above which a pair is ignored",
            default=0.5, min=0, max=20)
        target_d: FloatProperty(
            name="Target Translation",
            description="Converged once the last five iterations all moved the object by less than this", default=0.01,
            min=0, max=10)
        use_target: BoolProperty(
            name="Use Target", description="Compute pair-distance statistics and test convergence every iteration",
            default=True)
        take_m_with: BoolProperty(
            name="Take m_ Objects with", description="Also move every scene object whose name starts with 'm_'",
            default=False)
        align_meth: EnumProperty(items=[("0", "RIGID", "0"), ("1", "ROT_LOC_SCALE", "1")], name="Alignment Method",
                                 description="Rigid, or rotation + translation + uniform scale", default="0")

        def draw(self, context):
            col = self.layout.column()
            for name in ("icp_iterations", "redraw_frequency", "sample_fraction", "min_start", "target_d", "use_target",
                         "take_m_with", "align_meth"):
                col.prop(self, name)

    class VIEW3D_PT_object_alignment(Panel):
        bl_space_type = "VIEW_3D"
        bl_region_type = "UI"
        bl_label = "Object Alignment"
        bl_idname = "VIEW3D_PT_object_alignment_amd"
        bl_category = "Alignment"

        def draw(self, context):
            col = self.layout.column(align=True)
            col.operator("object.align_icp")
            col.operator("object.align_icp_redraw")
            prefs = _addon_prefs()
            if prefs is not None:
                box = self.layout.box()
                for name in ("icp_iterations", "sample_fraction", "min_start", "target_d", "use_target"):
                    box.prop(prefs, name)

    def _addon_prefs():
        try:
            return bpy.context.preferences.addons[__package__].preferences
        except Exception:
            return None

    def _get_addon_preferences():
        """functions/common/blender.py:48-55: the registered AddonPreferences, else the plain settings object."""
        prefs = _addon_prefs()
        return prefs if prefs is not None else _icp_align._prefs

    _classes = [ObjectAlignmentPreferences, OBJECT_OT_icp_align, OBJECT_OT_icp_align_feedback, VIEW3D_PT_object_alignment]


def register():
    if bpy is None:
        raise RuntimeError("object_alignment_amd.blender_addon.register() needs Blender's bpy")
    for cls in _classes:
        bpy.utils.register_class(cls)
    _icp_align.get_addon_preferences = _get_addon_preferences
    from .operators import icp_align_feedback as _fb
    _fb.get_addon_preferences = _get_addon_preferences


def unregister():
    if bpy is None:
        return
    for cls in reversed(_classes):
        bpy.utils.unregister_class(cls)
