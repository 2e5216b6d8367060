"""det3d_amd -- MI355X-native hot path of Det3D (voxelize -> VFE -> sparse 3D conv -> RPN
-> head -> rotated NMS) behind the reference's own reader/backbone/neck/head registry and
config API.  HIP kernels + C ABI live in det3d_amd/csrc (include/d3d.h); everything in
this package is the host-side mirror of the reference interfaces for that path.
"""
__version__ = "0.1.0"

from .config import Config, ConfigDict  # noqa: F401
from .builder import build_box_coder, build_detector, get_downsample_factor  # noqa: F401
