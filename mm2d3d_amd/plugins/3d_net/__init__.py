"""Model plugin ``3d_net`` (reference contract: train.py:522-531)."""
from mm2d3d_amd.net3d import Net3DSeg as Model
from mm2d3d_amd.net3d import dependencies, signature

__all__ = ["Model", "signature", "dependencies"]
