"""2D RGB-D branch: two ResNet34 encoders (RGB, sparse depth) + U-Net decoder + 5x5-avgpool/1x1 heads + lifting.

Mirrors the plugin the reference loads by name (``2d_net``: /root/reference/.../2d_net/model.py:35-180,
backbones.py:13-65): same constructor arguments, same ``state_dict`` keys (the encoder keys equal torchvision's
resnet34: ``layer1.0.conv1.weight`` ...), same return tuple.  torchvision is not a dependency: the ResNet34
BasicBlock stack [3,4,6,3] is built here.  Recorded parity decisions:
  * ``segm_last`` is bound also when no padding was needed (the reference raises UnboundLocalError then);
  * no ImageNet weights exist offline: ``pretrained=True`` only selects the 3-channel stem.
"""
from __future__ import annotations

import numpy as np
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nn2d
from .lifting import PixelIndex, lift

signature = (
    {"img": np.zeros([1, 3, 480, 640], dtype=np.float32)},
    {"segm": np.zeros([1, 1, 480, 640], dtype=np.float32)},
)
dependencies = [f"numpy>={np.__version__}", f"torch=={torch.__version__}"]


class FrozenBatchNorm2d(nn.Module):
    """Affine with fixed statistics (torchvision.ops.FrozenBatchNorm2d semantics: eps 1e-5, buffers only)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))

    def forward(self, x):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn2d.BatchNorm2d
        self.conv1 = nn2d.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn2d.ReLU(inplace=True)
        self.conv2 = nn2d.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = downsample
        self.stride = stride
        nn2d.feeds_bn(self.conv1, self.bn1)  # batch statistics in the convolution's epilogue (nn2d._wants_stats)
        nn2d.feeds_bn(self.conv2, self.bn2)
        self._fused = isinstance(self.bn1, nn2d.BatchNorm2d)
        if self._fused:  # ReLU after bn1, and (+identity, ReLU) after bn2, run inside the BN apply kernel
            self.bn1.relu = True
            self.bn2.relu = True

    def _identity(self, x):
        if self.downsample is None:
            return x
        if (self._fused and not nn2d.fp32_mode() and torch.is_grad_enabled() and getattr(x, "_mm_handoff", None) is not None
                and isinstance(self.downsample[0], nn2d.Conv2d)):
            # x is read by conv1 AND by the 1x1 downsample: the downsample's data gradient goes to x's producer through its
            # hand-off slot (nn2d.GradHandoff) instead of an autograd add of two full maps
            identity = self.downsample[0](x, handoff=x._mm_handoff)
            for m in list(self.downsample)[1:]:
                identity = m(identity)
            return identity
        return self.downsample(x)

    @staticmethod
    def forward_pair(b1, b2, x1, x2, out1=None, out2=None):
        """(b1(x1, out1), b2(x2, out2)) for the same block of the two backbones, their 3x3 stride-1 convolutions as pairs in one
        launch each (nn2d.conv_pair): same arithmetic, the persistent kernel's last round shared between the two."""
        if not (b1._fused and b2._fused) or nn2d.fp32_mode():
            return b1(x1, out=out1), b2(x2, out=out2)
        if (b1.downsample is not None and b2.downsample is not None and len(b1.downsample) == 2 and len(b2.downsample) == 2
                and isinstance(b1.downsample[0], nn2d.Conv2d) and isinstance(b2.downsample[0], nn2d.Conv2d) and torch.is_grad_enabled()
                and getattr(x1, "_mm_handoff", None) is not None and getattr(x2, "_mm_handoff", None) is not None):
            # the 1x1 downsample convolutions one after the other (implicit GEMM, handing their data gradients to the inputs'
            # producers as in _identity), their batch norms as a pair
            id1, id2 = nn2d.bn_pair(b1.downsample[1], b2.downsample[1], b1.downsample[0](x1, handoff=x1._mm_handoff),
                                    b2.downsample[0](x2, handoff=x2._mm_handoff))
        else:
            id1, id2 = b1._identity(x1), b2._identity(x2)
        c1, c2 = nn2d.conv_pair(b1.conv1, b2.conv1, x1, x2)
        y1, y2 = nn2d.bn_pair(b1.bn1, b2.bn1, c1, c2)
        c1, c2 = nn2d.conv_pair(b1.conv2, b2.conv2, y1, y2)
        return nn2d.bn_pair(b1.bn2, b2.bn2, c1, c2, id1, id2, out1, out2, residual_shared=b1.downsample is None)

    def forward(self, x, out=None):
        """``out``: optional NHWC channel slice the block's result is written into (see nn2d.CatBuffer)."""
        identity = self._identity(x)
        if self._fused:
            y = self.bn1(self.conv1(x))
            return self.bn2(self.conv2(y), identity, out=out, residual_shared=self.downsample is None)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


def _make_layer(inplanes, planes, blocks, stride, norm_layer):
    down = None
    if stride != 1 or inplanes != planes:
        down = nn.Sequential(nn2d.Conv2d(inplanes, planes, 1, stride, bias=False), (norm_layer or nn2d.BatchNorm2d)(planes))
        nn2d.feeds_bn(down[0], down[1])
    layers = [BasicBlock(inplanes, planes, stride, down, norm_layer)]
    layers += [BasicBlock(planes, planes, norm_layer=norm_layer) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


def _resnet_init(module):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class Backbone(nn.Module):
    """ResNet34 encoder whose stem does NOT downsample: conv 7x7 stride 1 (backbones.py:23-25)."""

    def __init__(self, num_channel=3, pretrained=True, norm_layer=None):
        super().__init__()
        self.conv1 = nn2d.Conv2d(num_channel, 64, kernel_size=7, stride=1, padding=3, bias=False)
        self.bn1 = (norm_layer or nn2d.BatchNorm2d)(64)
        nn2d.feeds_bn(self.conv1, self.bn1)
        self.relu = nn2d.ReLU(inplace=True)
        self.maxpool = nn2d.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = _make_layer(64, 64, 3, 1, norm_layer)
        self.layer2 = _make_layer(64, 128, 4, 2, norm_layer)
        self.layer3 = _make_layer(128, 256, 6, 2, norm_layer)
        self.layer4 = _make_layer(256, 512, 3, 2, norm_layer)
        self.dropout = nn2d.Dropout(p=0.4)
        self._fused = isinstance(self.bn1, nn2d.BatchNorm2d)
        if self._fused:
            self.bn1.relu = True
        _resnet_init(self)

    @property
    def channels(self):
        return 64, 64, 128, 256, 512

    @staticmethod
    def _run(layer, x, out):
        for blk in list(layer)[:-1]:
            x = blk(x)
        last = layer[-1]
        return last(x, out=out) if (out is not None and getattr(last, "_fused", False)) else last(x)

    def forward(self, x, outs=None, pad_to=None):
        """``outs``: optional destinations (channel slices of the decoder's concat buffers) for feats 0..2.
        ``pad_to``: (Hp, Wp) - the network runs on the image zero-padded to that size; the stem's staging kernel writes the zeros."""
        feats = []
        o = (list(outs) + [None] * 3)[:3] if outs is not None else [None] * 3
        if self._fused:
            x, xp = nn2d.bn_pool(self.bn1, self.maxpool, self.conv1(x, pad_to=pad_to) if pad_to is not None else self.conv1(x), out=o[0])
        else:
            x = self.relu(self.bn1(self.conv1(x)))
            xp = self.maxpool(x)
        feats.append(x)
        x = self._run(self.layer1, xp, o[1])
        feats.append(x)
        x = self._run(self.layer2, x, o[2])
        feats.append(x)
        x = self.dropout(self.layer3(x))
        feats.append(x)
        x = self.dropout(self.layer4(x))
        feats.append(x)
        return feats


PAIR_LAYER1 = [os.environ.get("MM_PAIR_LAYER1", "1") != "0"]  # layer1 in lockstep too (its weight gradients pair): A/B switch


def backbone_pair(r, d, img, hints, outs_r=None, outs_d=None, pad_to=None):
    """(r(img, outs_r, pad_to), d(hints, outs_d, pad_to)) with layers 2-4 of the two encoders walked in lockstep, each pair of
    3x3 stride-1 convolutions in one launch (BasicBlock.forward_pair).  The stems run one after the other; layer1 (64 -> 64:
    weights-resident kernel, thousands of work items per layer) is walked in lockstep for its weight gradients' sake only.  Dropout draws in the order r.l3, d.l3, r.l4, d.l4."""
    if not (r._fused and d._fused) or nn2d.fp32_mode() or not nn2d._c2d.PAIR[0]:
        return r(img, outs=outs_r, pad_to=pad_to), d(hints, outs=outs_d, pad_to=pad_to)
    o_r = (list(outs_r) + [None] * 3)[:3] if outs_r is not None else [None] * 3
    o_d = (list(outs_d) + [None] * 3)[:3] if outs_d is not None else [None] * 3
    fr, fd = [], []
    xs = []
    for net, x, o, feats in ((r, img, o_r, fr), (d, hints, o_d, fd)):
        x, xp = nn2d.bn_pool(net.bn1, net.maxpool, net.conv1(x, pad_to=pad_to) if pad_to is not None else net.conv1(x), out=o[0])
        feats.append(x)
        xs.append(xp)
    xr, xd = xs

    def run_pair(lr, ld, xr, xd, out_r, out_d):
        br, bd = list(lr), list(ld)
        for i, (b1, b2) in enumerate(zip(br, bd)):
            last = i == len(br) - 1
            xr, xd = BasicBlock.forward_pair(b1, b2, xr, xd, out_r if last else None, out_d if last else None)
        return xr, xd

    # layer1 (64 -> 64, weights-resident kernel: XCDs 0-3 run one encoder's items, XCDs 4-7 the other's); its 75 MB maps are too large
    # for a batch-norm pair (the library runs them singly)
    if PAIR_LAYER1[0]:
        xr, xd = run_pair(r.layer1, d.layer1, xr, xd, o_r[1], o_d[1])
    else:
        xr, xd = Backbone._run(r.layer1, xr, o_r[1]), Backbone._run(d.layer1, xd, o_d[1])
    fr.append(xr), fd.append(xd)
    xr, xd = run_pair(r.layer2, d.layer2, xr, xd, o_r[2], o_d[2])
    fr.append(xr), fd.append(xd)
    xr, xd = run_pair(r.layer3, d.layer3, xr, xd, None, None)
    xr, xd = r.dropout(xr), d.dropout(xd)
    fr.append(xr), fd.append(xd)
    xr, xd = run_pair(r.layer4, d.layer4, xr, xd, None, None)
    xr, xd = r.dropout(xr), d.dropout(xd)
    fr.append(xr), fd.append(xd)
    return fr, fd


def _pixel_index(data_batch, h, w, device):
    idx = data_batch.get("_pixel_index")
    if idx is None or (idx.H, idx.W) != (h, w) or idx.device != device:
        idx = PixelIndex(data_batch["img_indices"], h, w, device)
        data_batch["_pixel_index"] = idx
    return idx


class L2G_classifier_2D(nn.Module):
    def __init__(self, input_channels, num_classes):
        super().__init__()
        self.con1_1_avg = nn2d.Conv2d(input_channels, num_classes, kernel_size=1, stride=1)
        self.linear = nn.Linear(input_channels, num_classes)  # unused by forward, kept for checkpoint parity
        self.dow_avg = nn2d.AvgPool2d((5, 5), stride=(1, 1), padding=(2, 2))

    def forward(self, input_2D_feature, pixel_index, avg=None):
        if avg is None:  # standalone use; inside Net2DSeg both heads are computed by one fused pass
            h, w = input_2D_feature.shape[2], input_2D_feature.shape[3]
            avg = nn2d.fused_heads(input_2D_feature, h, w, self.con1_1_avg, self.con1_1_avg)[0]
        return {"seg_logit_avg": lift(avg, pixel_index), "seg_logit_avg_2d": avg}


class Net2DSeg(nn.Module):
    def __init__(self, num_classes, pretrained=True, frozen_batch_norm=False):
        super().__init__()
        feat_channels = 64
        norm_layer = FrozenBatchNorm2d if frozen_batch_norm else None
        self.rgb_backbone = Backbone(pretrained=pretrained, norm_layer=norm_layer)
        self.depth_backbone = Backbone(num_channel=1, pretrained=False)
        _, self.dec_t_conv_stage5 = self.dec_stage(self.rgb_backbone.layer4, num_concat=1, num_concat_t=2)
        self.dec_conv_stage4, self.dec_t_conv_stage4 = self.dec_stage(self.rgb_backbone.layer3, num_concat=3)
        self.dec_conv_stage3, self.dec_t_conv_stage3 = self.dec_stage(self.rgb_backbone.layer2, num_concat=3)
        self.dec_conv_stage2, self.dec_t_conv_stage2 = self.dec_stage(self.rgb_backbone.layer1, num_concat=3)
        self.dec_conv_stage1 = nn2d.Conv2d(3 * 64, 64, kernel_size=3, padding=1)
        self.dow_avg = nn2d.AvgPool2d((5, 5), stride=(1, 1), padding=(2, 2))
        self.con1_1_avg = nn2d.Conv2d(64, num_classes, kernel_size=1, stride=1)
        self.aux = L2G_classifier_2D(feat_channels, num_classes)

    @staticmethod
    def dec_stage(enc_stage, num_concat, num_concat_t=1):
        cin = enc_stage[0].conv1.in_channels
        cout = enc_stage[-1].conv2.out_channels
        # index 2 of each Sequential is the reference's ReLU; here it is fused into the BatchNorm apply kernel
        conv = nn.Sequential(nn2d.Conv2d(num_concat * cout, cout, kernel_size=3, padding=1), nn2d.BatchNorm2d(cout, relu=True),
                             nn2d.FusedAway())
        t_conv = nn.Sequential(nn2d.ConvTranspose2d(cout * num_concat_t, cin, kernel_size=2, stride=2),
                               nn2d.BatchNorm2d(cin, relu=True), nn2d.FusedAway())
        nn2d.feeds_bn(conv[0], conv[1])
        nn2d.feeds_bn(t_conv[0], t_conv[1])
        return conv, t_conv

    def _forward_fp32(self, data_batch, img, hints, img_indices, h, w):
        """`precision: 32` (config/run/test.yaml:8): the same modules and parameters, fp32 kernels (nn2d.set_precision(32)),
        the decoder's concatenations as plain copies (2d_net/model.py:104-127)."""
        r = self.rgb_backbone(img)
        d = self.depth_backbone(hints)
        cat = nn2d.cat_channels
        x = self.dec_t_conv_stage5(cat([d[4], r[4]]))
        x = self.dec_conv_stage4(cat([d[3], x, r[3]]))
        x = self.dec_t_conv_stage4(x)
        x = self.dec_conv_stage3(cat([d[2], x, r[2]]))
        x = self.dec_t_conv_stage3(x)
        x = self.dec_conv_stage2(cat([d[1], x, r[1]]))
        x = self.dec_t_conv_stage2(x)
        x = self.dec_conv_stage1(cat([d[0], x, r[0]]))
        segm_last = x[:, :, 0:h, 0:w]
        segm, avg = nn2d.fused_heads(x, h, w, self.con1_1_avg, self.aux.con1_1_avg)
        pix = _pixel_index(data_batch, h, w, segm.device)
        preds = {"seg_logit": lift(segm, pix), "seg_logit_2d": segm}
        return preds, segm_last, img_indices, self.aux(segm_last, pix, avg)

    def forward(self, data_batch):
        img, hints, img_indices = data_batch["img"], data_batch["depth"], data_batch["img_indices"]
        h, w = img.shape[2], img.shape[3]
        # the point -> pixel index first: its host arrays are uploaded asynchronously from pinned staging (lifting.PixelIndex)
        # before any of this forward is queued, so the host never waits for the convolutions to drain
        pix = _pixel_index(data_batch, h, w, img.device)
        if nn2d.fp32_mode():
            pad_h, pad_w = (-h) % 16, (-w) % 16
            if pad_h or pad_w:
                img = F.pad(img, [0, pad_w, 0, pad_h])
                hints = F.pad(hints, [0, pad_w, 0, pad_h])
            return self._forward_fp32(data_batch, img, hints, img_indices, h, w)
        from . import graph2d

        if graph2d.usable(self, img, hints):
            # the static-shape trunk (stems ... heads) as two HIP graphs - forward and backward - replayed per step: ~600 launches of
            # Python / autograd / ctypes work per step become two graph launches (mm2d3d_amd/graph2d.py)
            x, segm, avg = graph2d.run(self, img, hints, h, w, pix)
        else:
            x, segm, avg = self._trunk(img, hints, h, w, pix)
        segm_last = x[:, :, 0:h, 0:w]  # crop of the padding (a view; the heads read the padded map with bounds h, w)
        preds = {"seg_logit": lift(segm, pix), "seg_logit_2d": segm}
        return preds, segm_last, img_indices, self.aux(segm_last, pix, avg)

    def _trunk(self, img, hints, h, w, pix):
        """Images -> (decoder output map x [B, 64, Hp, Wp] 16-bit NHWC, main head logits, aux head logits [B, nc, h, w] fp32): every
        layer whose shapes depend on the image size only (the lifting to points does not belong to it)."""
        pad_h, pad_w = (-h) % 16, (-w) % 16
        fused_stems = self.rgb_backbone._fused and self.depth_backbone._fused
        if (pad_h or pad_w) and not fused_stems:
            img = F.pad(img, [0, pad_w, 0, pad_h])
            hints = F.pad(hints, [0, pad_w, 0, pad_h])
        # The three full-resolution decoder concats [depth | up | rgb] are never copied: their buffers exist up front and
        # the producing BatchNorm layers (stem / layer1 / layer2 of both backbones, the transposed-conv stages) write
        # straight into their channel slices; the backbones keep reading those slices as pitched NHWC maps.
        Bn, Hp, Wp = img.shape[0], h + pad_h, w + pad_w
        pad_to = (Hp, Wp) if (fused_stems and (pad_h or pad_w)) else None  # the stems' staging kernels write the padding zeros
        cb = [nn2d.CatBuffer(Bn, (c, c, c), Hp >> l, Wp >> l, img.device) for l, c in enumerate(self.rgb_backbone.channels[:3])]
        r, d = backbone_pair(self.rgb_backbone, self.depth_backbone, img, hints, [b.slot(2) for b in cb], [b.slot(0) for b in cb], pad_to)
        for l in range(3):
            cb[l].put(0, d[l], shared=True)  # the backbones keep consuming these maps
            cb[l].put(2, r[l], shared=True)
        cat = nn2d.cat_channels

        def up(stage, x, buf):  # ConvTranspose2d + BatchNorm(+ReLU) writing into the middle slice of the next concat
            bn = stage[1]
            if isinstance(bn, nn2d.BatchNorm2d):
                y = bn(stage[0](x), out=buf.slot(1))
                for m in list(stage)[2:]:
                    y = m(y)
                buf.put(1, y)
                return y
            return stage(x)

        # decoder: concat order is [depth, upsampled, rgb] (model.py:107,112,117,122)
        x = self.dec_t_conv_stage5(cat([d[4], r[4]]))
        x = self.dec_conv_stage4(cat([d[3], x, r[3]]))
        x = up(self.dec_t_conv_stage4, x, cb[2])
        x = self.dec_conv_stage3(cb[2].cat([d[2], x, r[2]]))
        x = up(self.dec_t_conv_stage3, x, cb[1])
        x = self.dec_conv_stage2(cb[1].cat([d[1], x, r[1]]))
        x = up(self.dec_t_conv_stage2, x, cb[0])
        x = self.dec_conv_stage1(cb[0].cat([d[0], x, r[0]]))
        segm, avg = nn2d.fused_heads(x, h, w, self.con1_1_avg, self.aux.con1_1_avg, pix)
        return x, segm, avg


Model = Net2DSeg

