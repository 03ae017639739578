"""CPU oracle: the 2D RGB-D branch (TEST INFRASTRUCTURE ONLY).

Functional restatement over a ``state_dict`` of /root/reference/.../2d_net/model.py:84-180 and backbones.py:43-65
(ResNet34 BasicBlock stack [3,4,6,3] with a stride-1 7x7 stem; decoder concat order [depth, up, rgb]; 5x5 avg-pool +
1x1 heads; per-sample pixel gather).  torch CPU ``F.conv2d / conv_transpose2d / batch_norm / max_pool2d /
avg_pool2d`` are the arithmetic oracle (an independent third-party implementation).
torchvision is absent, so the ResNet34 topology is restated from its published definition; the state_dict key names
are torchvision's, which is what the reference's checkpoints use.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))

# ``emulate_bf16``: round to bfloat16 exactly where the HIP branch stores bfloat16 (input images, convolution weights,
# every convolution output, every BatchNorm(+add+ReLU) output) while all arithmetic stays fp32.  Against this oracle the
# HIP 2D net differs only by accumulation order (and the 1-ulp bf16 flips that order causes), not by the storage format.
_EMULATE = [False]


def _q(t):
    e = _EMULATE[0]
    if not e:
        return t
    return t.to(torch.bfloat16 if e is True else e).float()  # True = bfloat16; torch.float16 for the fp16 build of the kernels


def _bn(sd, pre, x, training, stats_out=None):
    """nn.BatchNorm2d: eps 1e-5, momentum 0.1.  In training the updated running stats go to ``stats_out[pre]``."""
    w, b = sd[pre + ".weight"], sd[pre + ".bias"]
    if not training:
        return F.batch_norm(x, sd[pre + ".running_mean"], sd[pre + ".running_var"], w, b, False, 0.1, 1e-5)
    if stats_out is None:
        return F.batch_norm(x, None, None, w, b, True, 0.1, 1e-5)
    rm, rv = sd[pre + ".running_mean"].clone(), sd[pre + ".running_var"].clone()
    stats_out[pre] = (rm, rv)
    return F.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)


def _block(sd, pre, x, stride, has_down, training, so):
    idt = x
    out = _q(F.relu(_bn(sd, pre + ".bn1", _q(F.conv2d(x, _q(sd[pre + ".conv1.weight"]), None, stride, 1)), training, so)))
    out = _bn(sd, pre + ".bn2", _q(F.conv2d(out, _q(sd[pre + ".conv2.weight"]), None, 1, 1)), training, so)
    if has_down:
        idt = _q(_bn(sd, pre + ".downsample.1", _q(F.conv2d(x, _q(sd[pre + ".downsample.0.weight"]), None, stride, 0)), training, so))
    return _q(F.relu(out + idt))


def backbone(sd, pre, x, training, so, dropout_masks=None):
    feats = []
    x = _q(F.relu(_bn(sd, pre + ".bn1", _q(F.conv2d(_q(x), _q(sd[pre + ".conv1.weight"]), None, 1, 3)), training, so)))
    feats.append(x)
    x = F.max_pool2d(x, 3, 2, 1)
    inpl = 64
    for li, (planes, nblk, stride) in enumerate(LAYERS, 1):
        for b in range(nblk):
            s = stride if b == 0 else 1
            x = _block(sd, f"{pre}.layer{li}.{b}", x, s, b == 0 and (s != 1 or inpl != planes), training, so)
        inpl = planes
        if li >= 3 and dropout_masks is not None:  # dropout p=0.4 after layer3 / layer4 (train mode): masks supplied
            x = _q(x * dropout_masks[(pre, li)])
        feats.append(x)
    return feats


def _dec_conv(sd, pre, x, training, so):
    return _q(F.relu(_bn(sd, pre + ".1", _q(F.conv2d(x, _q(sd[pre + ".0.weight"]), sd[pre + ".0.bias"], 1, 1)), training, so)))


def _dec_tconv(sd, pre, x, training, so):
    return _q(F.relu(_bn(sd, pre + ".1", _q(F.conv_transpose2d(x, _q(sd[pre + ".0.weight"]), sd[pre + ".0.bias"], 2)), training, so)))


def lift(seg, img_indices):
    """model.py:131-137: permute(0,2,3,1)[i][rows, cols] per sample, then cat."""
    out = []
    for i in range(seg.shape[0]):
        ix = torch.as_tensor(img_indices[i])
        out.append(seg.permute(0, 2, 3, 1)[i][ix[:, 0], ix[:, 1]])
    return torch.cat(out, 0)


def net2d_forward(sd, data_batch, training=False, stats_out=None, dropout_masks=None, emulate_bf16=False):
    old = _EMULATE[0]
    _EMULATE[0] = emulate_bf16 if isinstance(emulate_bf16, torch.dtype) else bool(emulate_bf16)
    try:
        return _net2d_forward(sd, data_batch, training, stats_out, dropout_masks)
    finally:
        _EMULATE[0] = old


def _net2d_forward(sd, data_batch, training, stats_out, dropout_masks):
    img, hints, idx = data_batch["img"], data_batch["depth"], data_batch["img_indices"]
    h, w = img.shape[2], img.shape[3]
    pad_h, pad_w = (-h) % 16, (-w) % 16
    if pad_h or pad_w:
        img = F.pad(img, [0, pad_w, 0, pad_h])
        hints = F.pad(hints, [0, pad_w, 0, pad_h])
    so = stats_out
    r = backbone(sd, "rgb_backbone", img, training, so, dropout_masks)
    d = backbone(sd, "depth_backbone", hints, training, so, dropout_masks)
    x = _dec_tconv(sd, "dec_t_conv_stage5", torch.cat([d[4], r[4]], 1), training, so)
    x = _dec_conv(sd, "dec_conv_stage4", torch.cat([d[3], x, r[3]], 1), training, so)
    x = _dec_tconv(sd, "dec_t_conv_stage4", x, training, so)
    x = _dec_conv(sd, "dec_conv_stage3", torch.cat([d[2], x, r[2]], 1), training, so)
    x = _dec_tconv(sd, "dec_t_conv_stage3", x, training, so)
    x = _dec_conv(sd, "dec_conv_stage2", torch.cat([d[1], x, r[1]], 1), training, so)
    x = _dec_tconv(sd, "dec_t_conv_stage2", x, training, so)
    x = _q(F.conv2d(torch.cat([d[0], x, r[0]], 1), _q(sd["dec_conv_stage1.weight"]), sd["dec_conv_stage1.bias"], 1, 1))
    segm_last = x[:, :, :h, :w]
    segm = F.conv2d(F.avg_pool2d(segm_last, 5, 1, 2), sd["con1_1_avg.weight"], sd["con1_1_avg.bias"])
    avg = F.conv2d(F.avg_pool2d(segm_last, 5, 1, 2), sd["aux.con1_1_avg.weight"], sd["aux.con1_1_avg.bias"])
    preds = {"seg_logit": lift(segm, idx), "seg_logit_2d": segm}
    aux = {"seg_logit_avg": lift(avg, idx), "seg_logit_avg_2d": avg}
    return preds, segm_last, idx, aux
