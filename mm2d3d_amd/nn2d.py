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

from . import conv2d as _c2d


class Conv2d(nn.Conv2d):
    """HIP implicit-GEMM (bf16 MFMA) when Cin, Cout are multiples of 64; the 3/1-channel stems and the 6-class 1x1
    heads still take torch's path (interim, 1.3 % of the branch's FLOPs)."""

    def forward(self, x):
        k = self.kernel_size
        if x.is_cuda and _c2d.hip_eligible(self.in_channels, self.out_channels, k[0], k[1], self.stride[0], self.padding[0],
                                          self.dilation[0], self.groups) and self.stride[0] == self.stride[1] \
                and self.padding[0] == self.padding[1] and self.padding_mode == "zeros":
            return _c2d.Conv2dFn.apply(x, self.weight, self.bias, self.stride[0], self.padding[0])
        return super().forward(x)


class ConvTranspose2d(nn.ConvTranspose2d):
    def forward(self, x, output_size=None):
        if x.is_cuda and self.kernel_size == (2, 2) and self.stride == (2, 2) and self.padding == (0, 0) \
                and self.output_padding == (0, 0) and self.groups == 1 and self.in_channels % 64 == 0 \
                and self.out_channels % 64 == 0 and output_size is None:
            return _c2d.ConvTranspose2dFn.apply(x, self.weight, self.bias)
        return super().forward(x, output_size)


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
