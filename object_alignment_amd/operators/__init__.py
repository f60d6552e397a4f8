from .icp_align import IcpAlign, IcpSettings, OBJECT_OT_icp_align, build_vlist, get_addon_preferences  # noqa: F401
from .icp_align_feedback import OBJECT_OT_icp_align_feedback  # noqa: F401
from .align_pick_points import LandmarkAlign, OBJECT_OT_align_pick_points, base_pick_to_align_local  # noqa: F401
