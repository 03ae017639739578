"""3D branch: per-point RGB gate -> sparse U-Net -> linear heads, on the HIP operator surface.

Mirrors the plugin the reference loads by name (``3d_net``: /root/reference/.../3d_net/model.py:23-95,
scn_unet.py:8-126): same constructor arguments, same ``state_dict`` keys, same return tuple.
Differences recorded in SURVEY.md section 2.1: ``preds`` always carries ``confidence`` (the vKITTI copy
does, the others crash at train.py:233 without it).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import scn
from .scn import ops

DIMENSION = 3

signature = (
    {"depth": np.zeros([1, 3, 3000], dtype=np.float32)},
    {"seg": np.zeros([1, 1, 3000], dtype=np.float32)},
)
dependencies = [f"numpy>={np.__version__}", f"torch=={torch.__version__}"]


def _add_block(seq, a, b, residual_blocks, leakiness):
    body = scn.Sequential()
    body.add(scn.BatchNormLeakyReLU(a, leakiness=leakiness)).add(scn.SubmanifoldConvolution(DIMENSION, a, b, 3, False))
    if not residual_blocks:  # VGG style (scn_unet.py:48-53)
        seq.add(body)
        return
    body.add(scn.BatchNormLeakyReLU(b, leakiness=leakiness)).add(scn.SubmanifoldConvolution(DIMENSION, b, b, 3, False))
    skip = scn.Identity() if a == b else scn.NetworkInNetwork(a, b, False)
    seq.add(scn.ConcatTable().add(skip).add(body)).add(scn.AddTable())


def UNet(dimension, reps, nPlanes, residual_blocks=False, downsample=(2, 2), leakiness=0, n_input_planes=-1):
    """Recursive sparse U-Net (scn_unet.py:8-87): module indices equal the reference's, so checkpoints interchange."""
    assert dimension == DIMENSION
    p0 = nPlanes[0]
    seq = scn.Sequential()
    for r in range(reps):
        _add_block(seq, n_input_planes if (r == 0 and n_input_planes != -1) else p0, p0, residual_blocks, leakiness)
    if len(nPlanes) > 1:
        down_up = (
            scn.Sequential()
            .add(scn.BatchNormLeakyReLU(p0, leakiness=leakiness))
            .add(scn.Convolution(dimension, p0, nPlanes[1], downsample[0], downsample[1], False))
            .add(UNet(dimension, reps, nPlanes[1:], residual_blocks, downsample, leakiness))
            .add(scn.BatchNormLeakyReLU(nPlanes[1], leakiness=leakiness))
            .add(scn.Deconvolution(dimension, nPlanes[1], p0, downsample[0], downsample[1], False))
        )
        seq.add(scn.ConcatTable().add(scn.Identity()).add(down_up))
        seq.add(scn.JoinTable())
        for r in range(reps):
            _add_block(seq, p0 * (2 if r == 0 else 1), p0, residual_blocks, leakiness)
    return seq


class UNetSCN(nn.Module):
    def __init__(self, in_channels=1, m=16, block_reps=1, residual_blocks=False, full_scale=4096, num_planes=7, bn_momentum=None):
        """``bn_momentum`` (not in the reference): keep-fraction of the batch-norm running statistics for every layer of this
        net, see scn.DEFAULT_BN_MOMENTUM (0.99: the recalled constructor default of the pinned SparseConvNet commit; 0.9 is SURVEY.md A.5's / the docstring's reading)."""
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = m
        n_planes = [(n + 1) * m for n in range(num_planes)]
        prev = scn.DEFAULT_BN_MOMENTUM[0]
        if bn_momentum is not None:
            scn.set_default_bn_momentum(bn_momentum)
        try:
            self.layer1 = scn.InputLayer(DIMENSION, full_scale, mode=4)
            self.layer1.prebuild_levels = num_planes
            self.layer2 = scn.SubmanifoldConvolution(DIMENSION, in_channels, m, 3, False)
            self.layer3 = UNet(DIMENSION, block_reps, n_planes, residual_blocks)
            self.layer4 = scn.BatchNormReLU(m)
            self.layer5 = scn.OutputLayer(DIMENSION)
        finally:
            scn.set_default_bn_momentum(prev)

    def forward(self, x):
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        return self.layer5(x)


class HipLinear(nn.Linear):
    """nn.Linear parameters, forward/backward on the row-wise HIP kernels (csrc/point.hip)."""

    def forward(self, x):
        return ops.LinearFunction.apply(x, self.weight, self.bias)


class L2G_classifier_3D(nn.Module):
    def __init__(self, input_channels, num_classes):
        super().__init__()
        self.linear_point = HipLinear(input_channels, num_classes)
        self.linear_global = HipLinear(input_channels, num_classes)  # unused by forward, kept for checkpoint parity
        self.dow = nn.AvgPool1d(kernel_size=3, stride=1, padding=1)

    def forward(self, input_3D_feature):
        return {"feats": input_3D_feature, "seg_logit_point": self.linear_point(input_3D_feature)}


class Net3DSeg(nn.Module):
    def __init__(self, num_classes, dual_head=True, backbone_3d_kwargs=None):
        super().__init__()
        self.linear_rgb_mask = HipLinear(3, 1)
        self.net_3d = UNetSCN(**(backbone_3d_kwargs or {}))
        self.linear = HipLinear(self.net_3d.out_channels, num_classes)
        self.dual_head = dual_head
        self.aux = L2G_classifier_3D(16, num_classes)

    def prepare(self, data_batch, side_stream=None, after=None):
        """Optional: build the sparse metadata of ``data_batch`` ahead of the forward (TrainModel overlaps it with the 2D
        branch on a side stream).  ``forward`` finds it on the coordinate tensor."""
        layer = self.net_3d.layer1  # scn.InputLayer
        return scn.prebuild_metadata(data_batch["x"][0], layer.spatial_size, side_stream, after, layer.prebuild_levels)

    def begin_metadata(self, data_batch):
        """Phase one of a pipelined metadata build for ``data_batch`` (scn.begin_metadata): returns the pending Metadata."""
        layer = self.net_3d.layer1
        return scn.begin_metadata(data_batch["x"][0], layer.spatial_size, layer.prebuild_levels)

    def forward(self, data_batch):
        coords, feats = data_batch["x"][0], data_batch["x"][1]
        gated, mask_rgb = ops.GateFunction.apply(feats, self.linear_rgb_mask.weight, self.linear_rgb_mask.bias)
        # the reference gates in place on the batch dict (model.py:48): keep that visible side effect
        data_batch["x"][1] = gated
        out_3D_feature = self.net_3d([coords, gated])
        preds = {"seg_logit": self.linear(out_3D_feature), "confidence": mask_rgb}
        return preds, out_3D_feature, self.aux(out_3D_feature)


Model = Net3DSeg
