"""MI355X-native drop-in for the per-pixel light transport of MomentsInGraphics/path_tracer.

The product is the HIP library in csrc/ (C ABI in include/pt_hip.h) and the C host in host/;
this package only binds them for Python callers.
"""
from . import hip_api  # noqa: F401
