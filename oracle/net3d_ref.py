"""CPU oracle: the 3D branch composition (TEST INFRASTRUCTURE ONLY).

Restates EXP/3d_net/scn_unet.py:8-126 (UNet / UNetSCN) and EXP/3d_net/model.py:23-95
(Net3DSeg, L2G_classifier_3D) over ``oracle.scn_ref``.  The layer order, the channel plan
``n_planes = [(n+1)*m]`` (scn_unet.py:104), the VGG/ResNet block choice (scn_unet.py:36-53)
and the ``[enc | dec]`` join (scn_unet.py:61-83) are what must match; module attribute
names are chosen so that ``state_dict`` keys equal the reference's
(``net_3d.layer3.1.1.2.weight`` ...), which tests/test_oracle_wiring.py checks against
the reference's own ``scn_unet.UNet`` run over these primitives.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import scn_ref as scn


def _block(seq, a, b, residual, leak):
    if residual:
        seq.add(
            scn.ConcatTable()
            .add(scn.Identity() if a == b else scn.NetworkInNetwork(a, b, False))
            .add(
                scn.Sequential()
                .add(scn.BatchNormLeakyReLU(a, leakiness=leak))
                .add(scn.SubmanifoldConvolution(3, a, b, 3, False))
                .add(scn.BatchNormLeakyReLU(b, leakiness=leak))
                .add(scn.SubmanifoldConvolution(3, b, b, 3, False))
            )
        ).add(scn.AddTable())
    else:
        seq.add(
            scn.Sequential()
            .add(scn.BatchNormLeakyReLU(a, leakiness=leak))
            .add(scn.SubmanifoldConvolution(3, a, b, 3, False))
        )


def build_unet(planes, reps=1, residual=False, leak=0, n_in=-1):
    """Recursive U (scn_unet.py:55-84): reps blocks, then [Identity | BN-Conv-U-BN-Deconv], join, reps blocks."""
    seq = scn.Sequential()
    p0 = planes[0]
    for r in range(reps):
        _block(seq, n_in if (n_in != -1 and r == 0) else p0, p0, residual, leak)
    if len(planes) > 1:
        inner = (
            scn.Sequential()
            .add(scn.BatchNormLeakyReLU(p0, leakiness=leak))
            .add(scn.Convolution(3, p0, planes[1], 2, 2, False))
            .add(build_unet(planes[1:], reps, residual, leak))
            .add(scn.BatchNormLeakyReLU(planes[1], leakiness=leak))
            .add(scn.Deconvolution(3, planes[1], p0, 2, 2, False))
        )
        seq.add(scn.ConcatTable().add(scn.Identity()).add(inner))
        seq.add(scn.JoinTable())
        for r in range(reps):
            _block(seq, p0 * (2 if r == 0 else 1), p0, residual, leak)
    return seq


class UNetSCNRef(nn.Module):
    """scn_unet.py:90-126."""

    def __init__(self, in_channels=1, m=16, block_reps=1, residual_blocks=False, full_scale=4096, num_planes=7):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, m
        planes = [(i + 1) * m for i in range(num_planes)]
        self.layer1 = scn.InputLayer(3, full_scale, mode=4)
        self.layer2 = scn.SubmanifoldConvolution(3, in_channels, m, 3, False)
        self.layer3 = build_unet(planes, block_reps, residual_blocks)
        self.layer4 = scn.BatchNormReLU(m)
        self.layer5 = scn.OutputLayer(3)

    def forward(self, x):
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4, self.layer5):
            x = layer(x)
        return x


class AuxHead3DRef(nn.Module):
    """L2G_classifier_3D (model.py:61-95): only linear_point is used; linear_global/dow are dead params."""

    def __init__(self, input_channels, num_classes):
        super().__init__()
        self.linear_point = nn.Linear(input_channels, num_classes)
        self.linear_global = nn.Linear(input_channels, num_classes)

    def forward(self, feat):
        return {"feats": feat, "seg_logit_point": self.linear_point(feat)}


class Net3DSegRef(nn.Module):
    """model.py:23-58.  Gate: feats *= sigmoid(Linear(3,1)(feats)) IN PLACE on the batch dict."""

    def __init__(self, num_classes, dual_head=True, backbone_3d_kwargs=None):
        super().__init__()
        self.linear_rgb_mask = nn.Linear(3, 1)
        self.net_3d = UNetSCNRef(**(backbone_3d_kwargs or {}))
        self.linear = nn.Linear(self.net_3d.out_channels, num_classes)
        self.dual_head = dual_head
        self.aux = AuxHead3DRef(16, num_classes)

    def forward(self, data_batch):
        raw = data_batch["x"][1].clone()
        mask_rgb = torch.sigmoid(self.linear_rgb_mask(raw))
        # The reference multiplies in place (model.py:48).  In fp32 that invalidates the tensor
        # nn.Linear saved for its weight gradient (torch raises); it only trains under AMP, where
        # the saved tensor is the fp16 copy of the UN-gated feats.  The oracle therefore computes
        # the gate out of place (same forward values, gradient w.r.t. the un-gated feats) and then
        # mirrors the mutation of the batch dict.
        gated = raw * mask_rgb
        data_batch["x"][1].copy_(gated.detach())
        feat = self.net_3d([data_batch["x"][0], gated])
        preds = {"seg_logit": self.linear(feat), "confidence": mask_rgb}
        return preds, feat, self.aux(feat)
