"""Model plugin ``2d_net`` (reference contract: train.py:522-531 imports the module by bare name and reads
``Model``, ``signature``, ``dependencies``).  Put ``mm2d3d_amd/plugins`` on ``sys.path`` (``mm2d3d_amd.plugins.install()``)."""
from mm2d3d_amd.net2d import Net2DSeg as Model
from mm2d3d_amd.net2d import dependencies, signature

__all__ = ["Model", "signature", "dependencies"]
