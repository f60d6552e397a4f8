from .icp_align import IcpAlign, IcpSettings, OBJECT_OT_icp_align, build_vlist, get_addon_preferences  # noqa: F401
