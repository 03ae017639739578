"""Dense 2D layer modules of the RGB-D branch.

One module class per torch.nn layer the reference's 2D net instantiates (2d_net/backbones.py:13-65,
2d_net/model.py:35-82): same constructor signatures, parameter names, shapes and default init, so
``state_dict``s interchange with the reference (and with torchvision's resnet34 keys).

STATUS (round 1): these classes currently inherit torch.nn's forward (MIOpen / rocBLAS underneath) - an INTERIM
so the full training step can be measured end to end.  The hand-written MFMA implicit-GEMM kernels replace the
forwards class by class (tracked in DESIGN.md, section "2D branch"); nothing else in the package calls torch.nn
convolution directly.
"""
from __future__ import annotations

import torch.nn as nn


class Conv2d(nn.Conv2d):
    pass


class ConvTranspose2d(nn.ConvTranspose2d):
    pass


class BatchNorm2d(nn.BatchNorm2d):
    pass


class ReLU(nn.ReLU):
    pass


class MaxPool2d(nn.MaxPool2d):
    pass


class AvgPool2d(nn.AvgPool2d):
    pass


class Dropout(nn.Dropout):
    pass
