"""mm2d3d_amd: MI355X-native hot path of CVLAB-Unibo/MM2D3D (2D RGB-D branch, 3D sparse-voxel branch,
2D<->3D projection, data-parallel training) behind the reference's plugin surface.

Host code is Python on PyTorch-ROCm; all arithmetic of the hot path runs in hand-written gfx950 kernels
reached through the C-ABI library ``libmm2d3d_hip.so`` (see include/mm2d3d.h).
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
