"""Dense 2D layer modules of the RGB-D branch on hand-written gfx950 kernels.

One module class per torch.nn layer the reference's 2D net instantiates (2d_net/backbones.py:13-65,
2d_net/model.py:35-82): same constructor signatures, parameter names, shapes and default init, so ``state_dict``s
interchange with the reference (and with torchvision's resnet34 keys).  Activations are bfloat16 with torch's
``channels_last`` strides (NHWC in memory); parameters and BN statistics stay fp32.

Kernels: csrc/conv2d.hip (implicit-GEMM MFMA convolutions), csrc/bn2d.hip (BatchNorm + residual + ReLU),
csrc/misc2d.hip (concat, max-pool, fused segmentation heads).  GPU tensors always take the HIP path; CPU tensors are
refused (no CPU fallback).  Still on torch ops (interim, listed in DESIGN.md): Dropout's mask generation.
"""
from __future__ import annotations

import ctypes
import ctypes as C_
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, domains, gradsink
from . import conv2d as _c2d
from . import conv2d_f32 as _c2f
from ._lib import check, ptr, stream

CL = torch.channels_last
F32 = torch.float32


# 16: the 2D branch on the MFMA kernels over 16-bit maps (training hot path) - IEEE fp16 by default, the storage format of the
# reference's ``precision: 16`` (fp16 autocast + GradScaler, config/run/train.yaml:11); "bf16" selects bfloat16 maps (same
# kernels, same rate, 8 instead of 11 significand bits, no loss scale needed).  32: the whole 2D branch in fp32
# (config/run/test.yaml:8 `precision: 32`) on the fp32 implicit-GEMM convolutions of csrc/conv2d_f32.hip and the fp32
# batch-norm rows of csrc/bn.hip; pooling, padding, concatenation and the residual add are torch elementwise / pooling ops in
# that mode.
PRECISION = [16]
DEFAULT_PRECISION = 16


def set_precision(bits):
    """16 / "fp16": IEEE fp16 maps on the MFMA kernels (default; what the reference's ``precision: 16`` stores - the gradient maps
    want the loss scale of mm2d3d_amd/amp.py, which ``TrainModel`` installs, or Lightning's own GradScaler when the plugins run
    under the reference's trainer); "bf16": bfloat16 maps on the same kernels; 32: the exact-fp32 2D branch."""
    if isinstance(bits, str) and bits.isdigit():
        bits = int(bits)
    if bits in ("fp16", "f16", "half", 16):
        PRECISION[0] = 16
        _c2d.set_half(torch.float16)
        return
    if bits in ("bf16", "bfloat16"):
        PRECISION[0] = 16
        _c2d.set_half(torch.bfloat16)
        return
    if bits != 32:
        raise ValueError('precision must be 16 / "fp16" (IEEE fp16 MFMA, loss scale), "bf16" (bf16 MFMA) or 32 (exact fp32)')
    PRECISION[0] = 32


def half_kind():
    """"fp16" / "bf16" / "fp32": what the 2D branch currently stores."""
    if PRECISION[0] == 32:
        return "fp32"
    return "fp16" if _c2d.HALF[0] == torch.float16 else "bf16"


def fp32_mode():
    return PRECISION[0] == 32


def _need_gpu(x, what):
    if not x.is_cuda:
        raise RuntimeError(f"mm2d3d_amd.nn2d.{what}: input must be on the GPU (HIP path only, no CPU fallback)")


def _wants_stats(m):
    """A convolution directly in front of a training-mode BatchNorm2d files the batch statistics of its output in its epilogue
    (``feeds_bn`` is set by the model's constructor; conv2d.BN_PRE / MM_BN2D_PRE=0 turns the scheme off)."""
    return [None] if (m.feeds_bn and m.training and _c2d.BN_PRE[0] is not False) else None


def _with_stats(y, holder):
    if holder is not None and holder[0] is not None:
        y._mm_stats = holder[0]
    return y


def feeds_bn(conv, bn):
    """Marks ``conv`` as the producer of ``bn``'s input (see _wants_stats); returns conv."""
    if isinstance(bn, BatchNorm2d) and isinstance(conv, (Conv2d, ConvTranspose2d)):
        conv.feeds_bn = True
    return conv


class Conv2d(nn.Conv2d):
    feeds_bn = False

    def forward(self, x, handoff=None, pad_to=None):
        """``pad_to`` (stems only): see conv2d.StemConvFn."""
        _need_gpu(x, "Conv2d")
        k = self.kernel_size
        if fp32_mode():
            if not (self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1] and self.dilation == (1, 1)
                    and self.groups == 1 and self.padding_mode == "zeros"):
                raise NotImplementedError(f"mm2d3d_amd.nn2d.Conv2d (fp32): {self}")
            return _c2f.Conv2dF32Fn.apply(x, self.weight, self.bias, self.stride[0], self.padding[0])
        if pad_to is not None and not (k == (7, 7) and self.in_channels <= 8 and not fp32_mode()):
            raise NotImplementedError("nn2d.Conv2d(pad_to=): only the 7x7 stems stage their input (conv2d.StemConvFn)")
        if _c2d.hip_eligible(self.in_channels, self.out_channels, k[0], k[1], self.stride[0], self.padding[0], self.dilation[0],
                             self.groups) and self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1] \
                and self.padding_mode == "zeros":
            st = _wants_stats(self)
            return _with_stats(_c2d.Conv2dFn.apply(x, self.weight, self.bias, self.stride[0], self.padding[0], handoff, st), st)
        if k == (7, 7) and self.stride == (1, 1) and self.padding == (3, 3) and self.in_channels <= 8 and self.bias is None \
                and self.out_channels % 64 == 0 and self.groups == 1:
            st = _wants_stats(self)
            return _with_stats(_c2d.StemConvFn.apply(x, self.weight, st, pad_to, _c2d.CAPTURE_ANCHOR[0]), st)  # the two stems (backbones.py:23-25)
        raise NotImplementedError(f"mm2d3d_amd.nn2d.Conv2d: shape not on the hot path: {self}")  # (pad_to is the stems' alone)


def conv_pair(m1, m2, x1, x2):
    """(m1(x1), m2(x2)) for two Conv2d modules of one shape - the same layer of the two backbones - as ONE launch where the kernel
    allows it (conv2d.Conv2dPairFn: 3x3 stride 1 pad 1, no bias, dense 16-bit maps), else as two calls."""
    ok = (not fp32_mode() and isinstance(m1, Conv2d) and isinstance(m2, Conv2d) and m1.bias is None and m2.bias is None
          and m1.kernel_size == m2.kernel_size == (3, 3) and m1.stride == m2.stride == (1, 1) and m1.padding == m2.padding == (1, 1)
          and m1.dilation == m2.dilation == (1, 1) and m1.groups == m2.groups == 1 and m1.padding_mode == m2.padding_mode == "zeros"
          and x1.is_cuda and x2.is_cuda and m1.training == m2.training and _c2d.pairable(x1, x2, m1.weight, m2.weight))
    if not ok:
        return m1(x1), m2(x2)
    st1, st2 = _wants_stats(m1), _wants_stats(m2)
    if (st1 is None) != (st2 is None):
        st1 = st2 = None
    y1, y2 = _c2d.Conv2dPairFn.apply(x1, x2, m1.weight, m2.weight, st1, st2)
    return _with_stats(y1, st1), _with_stats(y2, st2)


class ConvTranspose2d(nn.ConvTranspose2d):
    feeds_bn = False

    def forward(self, x, output_size=None):
        _need_gpu(x, "ConvTranspose2d")
        if fp32_mode():
            if not (self.kernel_size == self.stride and self.padding == (0, 0) and self.output_padding == (0, 0) and self.groups == 1
                    and self.stride[0] == self.stride[1] and output_size is None):
                raise NotImplementedError(f"mm2d3d_amd.nn2d.ConvTranspose2d (fp32): {self}")
            return _c2f.ConvTranspose2dF32Fn.apply(x, self.weight, self.bias, self.stride[0])
        if not (self.kernel_size == (2, 2) and self.stride == (2, 2) and self.padding == (0, 0) and self.output_padding == (0, 0)
                and self.groups == 1 and self.in_channels % 64 == 0 and self.out_channels % 64 == 0 and output_size is None):
            raise NotImplementedError("hot path: ConvTranspose2d kernel 2, stride 2, channels multiple of 64")
        st = _wants_stats(self)
        return _with_stats(_c2d.ConvTranspose2dFn.apply(x, self.weight, self.bias, st), st)


class GradHandoff:
    """A map with two consumers (block output -> next conv + next residual add; backbone feature -> next stage + decoder
    concat) would have its two gradient contributions summed by an autograd add kernel.  Instead the second consumer's
    backward leaves its contribution here and returns None, and the producing BatchNorm's backward kernels read both
    (``dy + extra`` in fp32).  The producer node runs after both consumers (graph dependency), so the slot is filled."""

    __slots__ = ("extra",)

    def __init__(self):
        self.extra = []


class _BN2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, weight, bias, running_mean, running_var, training, eps, momentum, relu, nbt=None, out=None, handoff=None,
                res_handoff=None, pre=None):
        """``pre``: (slab, rows, n_first, B) - the statistics slab the producing convolution filled in its epilogue."""
        L = _c2d.lib2d()
        ctx.handoff, ctx.res_handoff = handoff, res_handoff
        x, ldx = _c2d.nhwc_pitch(x)
        B, C, H, W = x.shape
        N = B * H * W
        ldr = C
        if res is not None:
            res, ldr = _c2d.nhwc_pitch(res)
        # ``out`` = [view]: a channel slice of a wider NHWC buffer (the decoder's concat input) that this layer fills in
        # place.  It travels in a list so that autograd sees a fresh output, not an in-place modified input view.
        if out is not None:
            dst = out[0].detach()
            y, ldy = _c2d.nhwc_pitch(dst)
            if y.data_ptr() != dst.data_ptr() or tuple(y.shape) != tuple(x.shape) or y.dtype != _c2d.HALF[0]:
                raise ValueError("BatchNorm2d(out=): destination must be an NHWC bf16 channel slice of the output's shape")
        else:
            y, ldy = torch.empty((B, C, H, W), dtype=_c2d.HALF[0], device=x.device, memory_format=CL), C
        if training:
            nf = domains.current()  # jointly batched domains keep their own batch statistics
            Ns = nf * H * W if (nf is not None and 0 < nf < B) else N
            stats = torch.empty((2, 2 if Ns < N else 1, C), dtype=F32, device=x.device)
            ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(C)), x.device)
            ctx.Ns = Ns
            ctx.hd = hd = _lib.handle(x.device)  # barrier words / fault word / switches (include/mm2d3d.h mm_create); also the backward's
            if pre is not None and pre[2] == (nf if (nf is not None and 0 < nf < B) else B) and pre[3] == B and pre[0].shape[2] == C:
                # batch statistics from the producer's epilogue: no statistics pass over x, no grid barrier
                check(L.mm_bn2d_fwd_train_pre(ptr(x), ldx, ptr(res), ldr, N, Ns, C, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var),
                                              ptr(nbt), eps, momentum, 1 if relu else 0, ptr(y), ldy, ptr(stats[0]), ptr(stats[1]), ptr(pre[0]),
                                              pre[1], ptr(ws), ws.numel(), stream()), "bn2d_fwd_train_pre")
            else:
                check(L.mm_bn2d_fwd_train(hd.h, ptr(x), ldx, ptr(res), ldr, N, Ns, C, ptr(weight), ptr(bias), ptr(running_mean),
                                          ptr(running_var), ptr(nbt), eps, momentum, 1 if relu else 0, ptr(y), ldy, ptr(stats[0]), ptr(stats[1]),
                                          ptr(ws), ws.numel(), stream()), "bn2d_fwd_train")
            ctx.save_for_backward(x, y, weight, stats, bias)
            ctx.sinks = None
            if weight is not None and bias is not None and gradsink.claim(ctx, weight, ctx.needs_input_grad[2]):
                gradsink.claim(ctx, bias, True)
                ctx.sinks = (weight, bias)
        else:
            check(L.mm_bn2d_fwd_eval(ptr(x), ldx, ptr(res), ldr, N, C, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), eps,
                                     1 if relu else 0, ptr(y), ldy, stream()), "bn2d_fwd_eval")
        ctx.training, ctx.relu, ctx.has_res = training, relu, res is not None
        ctx.lds = (ldx, ldy)
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise RuntimeError("BatchNorm2d backward in eval mode is not part of the hot path")
        L = _c2d.lib2d()
        x, y, weight, stats, bias = ctx.saved_tensors
        dy, lddy = _c2d.nhwc_pitch(dy)
        dy2, lddy2 = None, 0
        if ctx.handoff is not None and ctx.handoff.extra:
            extra, ctx.handoff.extra = ctx.handoff.extra, []
            for e in extra[1:]:  # never more than one in this model; kept general
                dy = dy + e
                dy, lddy = _c2d.nhwc_pitch(dy)
            dy2, lddy2 = _c2d.nhwc_pitch(extra[0])
        ymask = y if (ctx.has_res or not ctx.relu or weight is None) else None  # no residual: mask recomputed from x
        ldx, ldy = ctx.lds
        B, C, H, W = x.shape
        N = B * H * W
        dx = torch.empty((B, C, H, W), dtype=_c2d.HALF[0], device=x.device, memory_format=CL)
        dres = torch.empty_like(dx) if ctx.has_res else None
        ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(C)), x.device)
        if ctx.sinks is not None:  # dgamma / dbeta accumulate straight into the optimiser's gradient arena
            wp, bp = ctx.sinks
            dw = db = None
            dwt, dbt, acc = wp._mm_sink, bp._mm_sink, 1
        else:
            dw = dwt = torch.empty(C, dtype=F32, device=x.device)
            db = dbt = torch.empty(C, dtype=F32, device=x.device)
            acc = 0
        if _lib.BARRIER_LISTENERS and L.mm_bn2d_single_launch(ctx.hd.h, N, ctx.Ns, C, 1):
            _lib.before_barrier_kernel(True)
        check(L.mm_bn2d_bwd(ctx.hd.h, ptr(x), ldx, ptr(dy), lddy, ptr(dy2), lddy2, ptr(ymask), ldy, 1 if ctx.relu else 0, N, ctx.Ns, C, ptr(weight),
                            ptr(bias),
                            ptr(stats[0]), ptr(stats[1]),
                            ptr(dx), C, ptr(dres), C, ptr(dwt), ptr(dbt), acc, ptr(ws), ws.numel(),
                            stream()), "bn2d_bwd")
        if ctx.sinks is not None:
            gradsink.done(wp)
            gradsink.done(bp)
        if dres is not None and ctx.res_handoff is not None:  # the residual's producer sums it in its own backward kernels
            ctx.res_handoff.extra.append(dres)
            dres = None
        return dx, dres, dw, db, None, None, None, None, None, None, None, None, None, None, None


BN_PAIR = [os.environ.get("MM_BN2D_PAIR", "1") != "0"]  # the same BatchNorm2d layer of the two encoders as one launch (bn_pair)


class _BN2dPairFn(torch.autograd.Function):
    """Training-mode BatchNorm2d(+residual)(+ReLU) of TWO maps of one shape with their own parameters - the same layer of the RGB and
    of the depth encoder - through mm_bn2d_fwd_train_pair / mm_bn2d_bwd_pair: one single-launch kernel for both where the maps allow
    it (the library decides; else it runs them one after the other).  Per problem the arithmetic of _BN2dFn."""

    @staticmethod
    def forward(ctx, x1, x2, res1, res2, w1, w2, b1, b2, cfg1, cfg2):
        """cfg = dict(running_mean, running_var, eps, momentum, relu, nbt, out, handoff, res_handoff)"""
        L = _c2d.lib2d()
        cfgs = (cfg1, cfg2)
        xs, ress, ws_, bs = [x1, x2], [res1, res2], (w1, w2), (b1, b2)
        ldx, ldr, ys, ldy = [0, 0], [0, 0], [None, None], [0, 0]
        for i in range(2):
            xs[i], ldx[i] = _c2d.nhwc_pitch(xs[i])
            B, C, H, W = xs[i].shape
            ldr[i] = C
            if ress[i] is not None:
                ress[i], ldr[i] = _c2d.nhwc_pitch(ress[i])
            out = cfgs[i]["out"]
            if out is not None:
                dst = out[0].detach()
                ys[i], ldy[i] = _c2d.nhwc_pitch(dst)
                if ys[i].data_ptr() != dst.data_ptr() or tuple(ys[i].shape) != tuple(xs[i].shape) or ys[i].dtype != _c2d.HALF[0]:
                    raise ValueError("BatchNorm2d(out=): destination must be an NHWC 16-bit channel slice of the output's shape")
            else:
                ys[i], ldy[i] = torch.empty((B, C, H, W), dtype=_c2d.HALF[0], device=xs[i].device, memory_format=CL), C
        B, C, H, W = xs[0].shape
        N = B * H * W
        nf = domains.current()
        Ns = nf * H * W if (nf is not None and 0 < nf < B) else N
        stats = [torch.empty((2, 2 if Ns < N else 1, C), dtype=F32, device=xs[0].device) for _ in range(2)]
        ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(C)), xs[0].device)
        ctx.Ns = Ns
        ctx.hd = hd = _lib.handle(xs[0].device)
        args = []
        for i in range(2):
            c = cfgs[i]
            args.append(_lib.Bn2dFwdArgs(ptr(xs[i]), ldx[i], ptr(ress[i]), ldr[i], ptr(ws_[i]), ptr(bs[i]), ptr(c["running_mean"]),
                                         ptr(c["running_var"]), ptr(c["nbt"]), ptr(ys[i]), ldy[i], ptr(stats[i][0]), ptr(stats[i][1])))
        check(L.mm_bn2d_fwd_train_pair(hd.h, C_.byref(args[0]), C_.byref(args[1]), N, Ns, C, cfg1["eps"], cfg1["momentum"], 1 if cfg1["relu"] else 0,
                                       ptr(ws), ws.numel(), stream()), "bn2d_fwd_train_pair")
        ctx.save_for_backward(xs[0], xs[1], ys[0], ys[1], w1, w2, stats[0], stats[1], b1, b2)
        ctx.sinks = [None, None]
        for i in range(2):
            if gradsink.claim(ctx, ws_[i], ctx.needs_input_grad[4 + i]):
                gradsink.claim(ctx, bs[i], True)
                ctx.sinks[i] = (ws_[i], bs[i])
        ctx.relu, ctx.has_res = cfg1["relu"], res1 is not None
        ctx.lds = (ldx, ldy)
        ctx.handoffs = (cfg1["handoff"], cfg2["handoff"])
        ctx.res_handoffs = (cfg1["res_handoff"], cfg2["res_handoff"])
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, dy1, dy2):
        L = _c2d.lib2d()
        x1, x2, y1, y2, w1, w2, st1, st2, b1, b2 = ctx.saved_tensors
        xs, ys, ws_, bs, sts = (x1, x2), (y1, y2), (w1, w2), (b1, b2), (st1, st2)
        ldx, ldy = ctx.lds
        B, C, H, W = x1.shape
        N = B * H * W
        dys, lddy, dy2s, lddy2 = [dy1, dy2], [0, 0], [None, None], [0, 0]
        for i in range(2):
            dys[i], lddy[i] = _c2d.nhwc_pitch(dys[i])
            h = ctx.handoffs[i]
            if h is not None and h.extra:
                extra, h.extra = h.extra, []
                for e in extra[1:]:
                    dys[i] = dys[i] + e
                    dys[i], lddy[i] = _c2d.nhwc_pitch(dys[i])
                dy2s[i], lddy2[i] = _c2d.nhwc_pitch(extra[0])
        ymask = [ys[i] if (ctx.has_res or not ctx.relu) else None for i in range(2)]  # no residual: mask recomputed from x
        dx = [torch.empty((B, C, H, W), dtype=_c2d.HALF[0], device=x1.device, memory_format=CL) for _ in range(2)]
        dres = [torch.empty_like(dx[i]) if ctx.has_res else None for i in range(2)]
        wsb = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(C)), x1.device)
        # the pair entry shares ``accumulate``: both problems into their sinks, or both into fresh tensors
        both_sinks = ctx.sinks[0] is not None and ctx.sinks[1] is not None
        tgt = []
        for i in range(2):
            if both_sinks:
                tgt.append((ctx.sinks[i][0]._mm_sink, ctx.sinks[i][1]._mm_sink))
            else:
                tgt.append((torch.empty(C, dtype=F32, device=x1.device), torch.empty(C, dtype=F32, device=x1.device)))
        if _lib.BARRIER_LISTENERS and L.mm_bn2d_single_launch(ctx.hd.h, N, ctx.Ns, C, 1):
            _lib.before_barrier_kernel(True)
        args = [_lib.Bn2dBwdArgs(ptr(xs[i]), ldx[i], ptr(dys[i]), lddy[i], ptr(dy2s[i]), lddy2[i], ptr(ymask[i]), ldy[i], ptr(ws_[i]), ptr(bs[i]),
                                 ptr(sts[i][0]), ptr(sts[i][1]), ptr(dx[i]), C, ptr(dres[i]), C, ptr(tgt[i][0]), ptr(tgt[i][1])) for i in range(2)]
        check(L.mm_bn2d_bwd_pair(ctx.hd.h, C_.byref(args[0]), C_.byref(args[1]), 1 if ctx.relu else 0, N, ctx.Ns, C, 1 if both_sinks else 0,
                                 ptr(wsb), wsb.numel(), stream()), "bn2d_bwd_pair")
        dw, db = [None, None], [None, None]
        for i in range(2):
            if both_sinks:
                gradsink.done(ctx.sinks[i][0])
                gradsink.done(ctx.sinks[i][1])
            elif ctx.sinks[i] is not None:  # only one of the two has a sink: add its fresh gradient there by hand
                ctx.sinks[i][0]._mm_sink.add_(tgt[i][0])
                ctx.sinks[i][1]._mm_sink.add_(tgt[i][1])
                gradsink.done(ctx.sinks[i][0])
                gradsink.done(ctx.sinks[i][1])
            else:
                dw[i], db[i] = tgt[i]
            if dres[i] is not None and ctx.res_handoffs[i] is not None:
                ctx.res_handoffs[i].extra.append(dres[i])
                dres[i] = None
        return dx[0], dx[1], dres[0], dres[1], dw[0], dw[1], db[0], db[1], None, None


def bn_pair(m1, m2, x1, x2, res1=None, res2=None, out1=None, out2=None, residual_shared=False):
    """(m1(x1, res1, out1), m2(x2, res2, out2)) for the same BatchNorm2d layer of the two encoders - one launch where the library can
    (``_BN2dPairFn``), else two calls."""
    ok = (BN_PAIR[0] and not fp32_mode() and isinstance(m1, BatchNorm2d) and isinstance(m2, BatchNorm2d) and m1.training and m2.training
          and m1.track_running_stats and m2.track_running_stats and m1.affine and m2.affine and x1.is_cuda and x2.is_cuda
          and x1.shape == x2.shape and (res1 is None) == (res2 is None) and bool(m1.relu) == bool(m2.relu)
          and float(m1.eps) == float(m2.eps) and m1.momentum == m2.momentum
          and getattr(x1, "_mm_stats", None) is None and getattr(x2, "_mm_stats", None) is None)
    if not ok:
        return (m1(x1, res1, out=out1, residual_shared=residual_shared), m2(x2, res2, out=out2, residual_shared=residual_shared))
    cfgs = []
    track = torch.is_grad_enabled() and x1.requires_grad and x2.requires_grad
    for m, res, out in ((m1, res1, out1), (m2, res2, out2)):
        cfgs.append(dict(running_mean=m.running_mean, running_var=m.running_var, eps=float(m.eps),
                         momentum=float(m.momentum if m.momentum is not None else 0.1), relu=bool(m.relu), nbt=m.num_batches_tracked,
                         out=[out] if out is not None else None, handoff=GradHandoff() if track else None,
                         res_handoff=getattr(res, "_mm_handoff", None) if (track and res is not None and residual_shared) else None))
    y1, y2 = _BN2dPairFn.apply(x1, x2, res1, res2, m1.weight, m2.weight, m1.bias, m2.bias, cfgs[0], cfgs[1])
    for y, c in ((y1, cfgs[0]), (y2, cfgs[1])):
        if c["handoff"] is not None:
            y._mm_handoff = c["handoff"]
    return y1, y2


class BatchNorm2d(nn.BatchNorm2d):
    """nn.BatchNorm2d parameters/buffers; ``relu=True`` fuses the following ReLU, ``forward(x, residual)`` fuses the
    BasicBlock's ``out + identity`` before it.  One statistics pass + one fused apply pass (csrc/bn2d.hip)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True, relu=False):
        super().__init__(num_features, eps, momentum, affine, track_running_stats)
        self.relu = relu

    def forward(self, x, residual=None, out=None, residual_shared=False):
        """``residual_shared``: the residual map has another consumer (it is the block input, also read by conv1), so its
        gradient contribution from here is handed to its producer (GradHandoff) instead of being summed by autograd."""
        _need_gpu(x, "BatchNorm2d")
        if fp32_mode():
            return _c2f.batch_norm_f32(x, self, residual, self.relu)
        use_batch = self.training or not self.track_running_stats
        nbt = self.num_batches_tracked if (self.training and self.track_running_stats) else None  # incremented in the kernel
        track = use_batch and torch.is_grad_enabled() and x.requires_grad
        handoff = GradHandoff() if track else None
        res_handoff = getattr(residual, "_mm_handoff", None) if (track and residual is not None and residual_shared) else None
        y = _BN2dFn.apply(x, residual, self.weight, self.bias, self.running_mean, self.running_var, use_batch, float(self.eps),
                          float(self.momentum if self.momentum is not None else 0.1), bool(self.relu), nbt,
                          [out] if out is not None else None, handoff, res_handoff, getattr(x, "_mm_stats", None) if use_batch else None)
        if handoff is not None:
            y._mm_handoff = handoff
        if getattr(x, "_mm_stats", None) is not None:
            x._mm_stats = None  # the slab has served
        return y


class FusedAway(nn.Module):
    """Placeholder that keeps a Sequential's indices: the preceding BatchNorm2d(relu=True) already applied the ReLU."""

    def forward(self, x):
        return x


class ReLU(nn.ReLU):
    pass


MAXPOOL_HANDOFF = [False]  # see MaxPool2d.forward


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, handoff=None):
        L = _c2d.lib2d()
        ctx.handoff = handoff
        x, ldx = _c2d.nhwc_pitch(x)
        B, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((B, C, Ho, Wo), dtype=_c2d.HALF[0], device=x.device, memory_format=CL)
        idx = torch.empty(B * Ho * Wo * C, dtype=torch.uint8, device=x.device)
        check(L.mm_maxpool3x3s2_fwd(ptr(x), ldx, B, H, W, C, ptr(y), ptr(idx), stream()), "maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = _c2d.lib2d()
        (idx,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        dy, ldy = _c2d.nhwc_pitch(dy)
        dy2, ldy2 = None, 0
        if ctx.handoff is not None and ctx.handoff.extra:  # the second consumer's contribution (see GradHandoff)
            extra, ctx.handoff.extra = ctx.handoff.extra, []
            for e in extra[1:]:
                dy = dy + e
                dy, ldy = _c2d.nhwc_pitch(dy)
            dy2, ldy2 = _c2d.nhwc_pitch(extra[0])
        dx = torch.empty((B, C, H, W), dtype=_c2d.HALF[0], device=dy.device, memory_format=CL)
        check(L.mm_maxpool3x3s2_bwd(ptr(dy), ldy, ptr(dy2), ldy2, ptr(idx), B, H, W, C, ptr(dx), stream()), "maxpool_bwd")
        return dx, None


class MaxPool2d(nn.MaxPool2d):
    def forward(self, x):
        _need_gpu(x, "MaxPool2d")
        if fp32_mode():
            return torch.nn.functional.max_pool2d(x.contiguous(), self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode)
        if (self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode) != (3, 2, 1, 1, False):
            raise NotImplementedError("hot path: MaxPool2d(3, 2, 1)")
        # layer1.0 reads the pooled map twice (conv1 and the residual add).  Summing the two gradients inside k_maxpool_bwd
        # (dy2) was measured SLOWER than the add kernel it replaces (round 4, rocprofv3: 141 -> 205 us per call against a 40 us add:
        # every pooled pixel is read by ~2.25 input pixels, so the second map is fetched 2.25 times): off unless asked for.
        handoff = GradHandoff() if (MAXPOOL_HANDOFF[0] and torch.is_grad_enabled() and x.requires_grad) else None
        y = _MaxPoolFn.apply(x, handoff)
        if handoff is not None:
            y._mm_handoff = handoff
        return y


# the stems' BatchNorm + ReLU + MaxPool as one pass: "1" forward and backward, "fwd" the forward pass only (backward = the separate
# max-pool and batch-norm kernels), "0" (default) off.  Measured on the headline step (round 5, same box, p50 of 20 steps x 2):
# off 35.19 / 35.17 ms, fwd 35.10 / 35.16, both 35.16 / 35.14 - bit-identical results, 0.7 GB less HBM traffic per encoder, no gain:
# the fused passes trade the streaming access of the separate kernels for window gathers (k_bn2d_apply_pool 230 us against 132 +
# 118 us; the gathering reduce / apply +76 / +77 us against the 151 us max-pool backward they replace).  Kept as an option.
BN_POOL = [{"1": True, "fwd": "fwd"}.get(os.environ.get("MM_BN2D_POOL", "0"), False)]


class _BnPoolFn(torch.autograd.Function):
    """Training-mode BatchNorm2d(+ReLU) whose batch statistics come from the producing convolution's epilogue, FUSED with the
    MaxPool2d(3, 2, 1) that follows (the two stems, backbones.py:43-47): mm_bn2d_fwd_train_pre_pool / mm_bn2d_bwd_pool.  Returns (y,
    pooled y); ``out``: the channel slice of the decoder's concat buffer that receives y.  Backward: the pooled map's gradient is
    gathered inside the batch norm's backward passes - no full-resolution gradient map between the two layers."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, nbt, out, handoff, pre):
        L = _c2d.lib2d()
        ctx.handoff = handoff
        ctx.set_materialize_grads(False)  # y's gradient normally arrives through the hand-off: no zero map in its place
        x, ldx = _c2d.nhwc_pitch(x)
        B, C, H, W = x.shape
        if out is not None:
            dst = out[0].detach()
            y, ldy = _c2d.nhwc_pitch(dst)
            if y.data_ptr() != dst.data_ptr() or tuple(y.shape) != tuple(x.shape) or y.dtype != _c2d.HALF[0]:
                raise ValueError("BatchNorm2d(out=): destination must be an NHWC 16-bit channel slice of the output's shape")
        else:
            y, ldy = torch.empty((B, C, H, W), dtype=_c2d.HALF[0], device=x.device, memory_format=CL), C
        Ho, Wo = H // 2, W // 2
        yp = torch.empty((B, C, Ho, Wo), dtype=_c2d.HALF[0], device=x.device, memory_format=CL)
        idx = torch.empty(B * Ho * Wo * C, dtype=torch.uint8, device=x.device)
        nf = domains.current()
        Bs = nf if (nf is not None and 0 < nf < B) else B
        stats = torch.empty((2, 2 if Bs < B else 1, C), dtype=F32, device=x.device)
        ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(C)), x.device)
        check(L.mm_bn2d_fwd_train_pre_pool(ptr(x), ldx, B, H, W, Bs, C, ptr(weight), ptr(bias), ptr(running_mean), ptr(running_var), ptr(nbt), eps,
                                           momentum, ptr(y), ldy, ptr(yp), ptr(idx), ptr(stats[0]), ptr(stats[1]), ptr(pre[0]), pre[1], ptr(ws),
                                           ws.numel(), stream()), "bn2d_fwd_train_pre_pool")
        ctx.save_for_backward(x, weight, stats, bias, idx)
        ctx.dims = (B, C, H, W, Bs, ldx)
        ctx.hd = _lib.handle(x.device)
        ctx.sinks = None
        if gradsink.claim(ctx, weight, ctx.needs_input_grad[1]):
            gradsink.claim(ctx, bias, True)
            ctx.sinks = (weight, bias)
        return y, yp

    @staticmethod
    def backward(ctx, dy, dyp):
        L = _c2d.lib2d()
        x, weight, stats, bias, idx = ctx.saved_tensors
        B, C, H, W, Bs, ldx = ctx.dims
        extra = []
        if ctx.handoff is not None and ctx.handoff.extra:
            extra, ctx.handoff.extra = ctx.handoff.extra, []
        if dy is not None:
            extra.append(dy)
        dy2, lddy2 = None, 0
        if extra:
            acc_ = extra[0]
            for e in extra[1:]:  # never more than one in this model; kept general
                acc_ = acc_ + e
            dy2, lddy2 = _c2d.nhwc_pitch(acc_)
        if dyp is None:
            dyp = torch.zeros((B, C, H // 2, W // 2), dtype=_c2d.HALF[0], device=x.device, memory_format=CL)
        dyp, lddyp = _c2d.nhwc_pitch(dyp)
        dx = torch.empty((B, C, H, W), dtype=_c2d.HALF[0], device=x.device, memory_format=CL)
        ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(C)), x.device)
        if ctx.sinks is not None:
            wp, bp = ctx.sinks
            dw = db = None
            dwt, dbt, acc = wp._mm_sink, bp._mm_sink, 1
        else:
            dw = dwt = torch.empty(C, dtype=F32, device=x.device)
            db = dbt = torch.empty(C, dtype=F32, device=x.device)
            acc = 0
        if BN_POOL[0] == "fwd":  # the separate kernels: the pooled gradient scattered to a full-resolution map, then mm_bn2d_bwd
            dxp = torch.empty((B, C, H, W), dtype=_c2d.HALF[0], device=x.device, memory_format=CL)
            check(L.mm_maxpool3x3s2_bwd(ptr(dyp), lddyp, None, 0, ptr(idx), B, H, W, C, ptr(dxp), stream()), "maxpool_bwd")
            Ns = Bs * H * W
            hd = ctx.hd  # the forward's handle (the stem's batch norm is the LAST one of the backward pass: exactly the "tail" in which
            # ddp.GradAllReducer(overlap="tail") has buckets in flight - the listeners must hear of this barrier kernel too)
            if _lib.BARRIER_LISTENERS and L.mm_bn2d_single_launch(hd.h, B * H * W, Ns, C, 1):
                _lib.before_barrier_kernel(True)
            check(L.mm_bn2d_bwd(hd.h, ptr(x), ldx, ptr(dxp), C, ptr(dy2), lddy2, None, C, 1, B * H * W, Ns, C, ptr(weight),
                                ptr(bias), ptr(stats[0]), ptr(stats[1]), ptr(dx), C, None, C, ptr(dwt), ptr(dbt), acc, ptr(ws), ws.numel(),
                                stream()), "bn2d_bwd")
        else:
            check(L.mm_bn2d_bwd_pool(ptr(x), ldx, ptr(dyp), lddyp, ptr(idx), B, H, W, Bs, ptr(dy2), lddy2, C, ptr(weight), ptr(bias),
                                     ptr(stats[0]), ptr(stats[1]), ptr(dx), C, ptr(dwt), ptr(dbt), acc, ptr(ws), ws.numel(), stream()),
                  "bn2d_bwd_pool")
        if ctx.sinks is not None:
            gradsink.done(wp)
            gradsink.done(bp)
        return dx, dw, db, None, None, None, None, None, None, None, None


def bn_pool(bn, pool, x, out=None):
    """(bn(x, out=out), pool(bn(x))) - one fused pass where the library has one (see _BnPoolFn), else the two modules."""
    pre = getattr(x, "_mm_stats", None)
    B, C, H, W = x.shape
    nf = domains.current()
    ok = (BN_POOL[0] and not fp32_mode() and isinstance(bn, BatchNorm2d) and isinstance(pool, MaxPool2d) and bn.training and bn.relu
          and bn.track_running_stats and bn.affine and x.is_cuda and pre is not None and H % 2 == 0 and W % 2 == 0 and C % 8 == 0
          and (pool.kernel_size, pool.stride, pool.padding, pool.dilation, pool.ceil_mode) == (3, 2, 1, 1, False)
          and torch.is_grad_enabled() and x.requires_grad and not MAXPOOL_HANDOFF[0]
          and pre[2] == (nf if (nf is not None and 0 < nf < B) else B) and pre[3] == B and pre[0].shape[2] == C)
    if not ok:
        y = bn(x, out=out)
        return y, pool(y)
    handoff = GradHandoff()
    y, yp = _BnPoolFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, float(bn.eps),
                            float(bn.momentum if bn.momentum is not None else 0.1), bn.num_batches_tracked,
                            [out] if out is not None else None, handoff, pre)
    y._mm_handoff = handoff
    x._mm_stats = None  # the slab has served
    return y, yp


class AvgPool2d(nn.AvgPool2d):
    """Only ever used fused with the 1x1 head convolution: see ``fused_heads``."""


class Dropout(nn.Dropout):
    pass


class _CatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *xs):
        L = _c2d.lib2d()
        xs = [_c2d.as_nhwc_bf16(x) for x in xs]
        B, _, H, W = xs[0].shape
        cs = [x.shape[1] for x in xs]
        Ct = sum(cs)
        out = torch.empty((B, Ct, H, W), dtype=_c2d.HALF[0], device=xs[0].device, memory_format=CL)
        n = len(xs)
        check(L.mm_concat_bf16((ctypes.c_void_p * n)(*[x.data_ptr() for x in xs]), (ctypes.c_int * n)(*cs), n, ptr(out), B * H * W, 0,
                               stream()), "concat")
        ctx.cs = cs
        return out

    @staticmethod
    def backward(ctx, dy):
        L = _c2d.lib2d()
        dy = _c2d.as_nhwc_bf16(dy)
        B, Ct, H, W = dy.shape
        outs = [torch.empty((B, c, H, W), dtype=_c2d.HALF[0], device=dy.device, memory_format=CL) for c in ctx.cs]
        n = len(outs)
        check(L.mm_concat_bf16((ctypes.c_void_p * n)(*[g.data_ptr() for g in outs]), (ctypes.c_int * n)(*ctx.cs), n, ptr(dy), B * H * W, 1,
                               stream()), "split")
        return tuple(outs)


class _CatFilledFn(torch.autograd.Function):
    """torch.cat(parts, 1) where every part already IS its channel slice of ``holder[0]`` (each producer wrote it in
    place through BatchNorm2d(out=)): no copy forward; backward hands out the channel slices of the gradient as views."""

    @staticmethod
    def forward(ctx, holder, handoffs, *parts):
        buf = holder[0]
        off = 0
        for q in parts:
            if q.data_ptr() != buf.data_ptr() + 2 * off or q.stride(3) != buf.shape[1]:
                raise ValueError("cat_filled: part is not the expected slice of the buffer")
            off += q.shape[1]
        ctx.cs = [q.shape[1] for q in parts]
        ctx.handoffs = handoffs
        return buf.detach()

    @staticmethod
    def backward(ctx, dy):
        dy = _c2d.as_nhwc_bf16(dy)
        outs, off = [], 0
        for c, h in zip(ctx.cs, ctx.handoffs):
            g = dy[:, off:off + c]
            if h is not None:  # the part has another consumer: its producer adds this slice in its backward kernels
                h.extra.append(g)
                g = None
            outs.append(g)
            off += c
        return (None, None, *outs)


class CatBuffer:
    """Pre-allocated NHWC bf16 concat buffer whose channel slices are handed to the producers as destinations."""

    def __init__(self, B, channels, H, W, device):
        self.buf = torch.empty((B, sum(channels), H, W), dtype=_c2d.HALF[0], device=device, memory_format=CL)
        self.channels = list(channels)
        self.parts = [None] * len(channels)
        self.shared = [False] * len(channels)

    def slot(self, i):
        off = sum(self.channels[:i])
        return self.buf[:, off:off + self.channels[i]]

    def put(self, i, t, shared=False):
        """Record producer output ``t`` for slot i (False if the producer could not write in place: ``cat`` then copies).
        ``shared``: the map has another consumer besides the concat (its gradient slice is handed to the producer)."""
        v = self.slot(i)
        if t.data_ptr() != v.data_ptr() or t.stride(3) != self.buf.shape[1]:
            return False
        self.parts[i], self.shared[i] = t, shared
        return True

    def cat(self, fallback):
        if all(q is not None for q in self.parts):
            handoffs = [getattr(q, "_mm_handoff", None) if sh else None for q, sh in zip(self.parts, self.shared)]
            return _CatFilledFn.apply([self.buf], handoffs, *self.parts)
        return cat_channels(fallback)


def cat_channels(xs):
    """torch.cat(xs, dim=1) for NHWC bf16 maps (decoder concat [depth, up, rgb])."""
    if fp32_mode():
        return torch.cat([_c2f.as_nhwc_f32(t) for t in xs], 1).contiguous(memory_format=CL)
    return _CatFn.apply(*xs)


class _HeadsFn(torch.autograd.Function):
    """[main | aux] = Conv1x1(AvgPool5x5(x[:, :, :h, :w])) for both heads in one pass (csrc/misc2d.hip header)."""

    @staticmethod
    def forward(ctx, x, h, w, w1, b1, w2, b2, index=None):
        """``index``: the batch's lifting.PixelIndex - its backward passes file the gradient maps' per-channel sums there."""
        L = _c2d.lib2d()
        ctx.index = index
        x = _c2d.as_nhwc_bf16(x)
        B, C, Hp, Wp = x.shape
        nc = w1.shape[0]
        Wj = torch.cat([w1.reshape(nc, C), w2.reshape(nc, C)], 0).float().contiguous()
        bj = torch.cat([b1, b2], 0).float().contiguous()
        out = torch.empty((B, h, w, 2 * nc), dtype=F32, device=x.device)  # NHWC; returned as logical [B,C,h,w] views
        ws = _lib.workspace.get(int(L.mm_head_ws_bytes(B, h, w, Hp, Wp, C, 2 * nc)), x.device)
        check(L.mm_head_fwd(ptr(x), B, Hp, Wp, C, h, w, C, ptr(Wj), ptr(bj), 2 * nc, ptr(out), ptr(ws), ws.numel(), stream()), "head_fwd")
        ctx.save_for_backward(x, Wj)
        ctx.dims = (h, w, nc, w1.shape)
        # gradient sinks for the four head parameters (all or none): their gradients then go straight into the optimiser's arena
        ps = (w1, b1, w2, b2)
        ctx.sinks = None
        if all(hasattr(q, "_mm_sink") for q in ps) and all(ctx.needs_input_grad[i] for i in (3, 4, 5, 6)):
            for q in ps:
                gradsink.claim(ctx, q, True)
            ctx.sinks = ps
        o = out.permute(0, 3, 1, 2)
        return o[:, :nc], o[:, nc:]

    @staticmethod
    def backward(ctx, d1, d2):
        L = _c2d.lib2d()
        x, Wj = ctx.saved_tensors
        h, w, nc, wshape = ctx.dims
        B, C, Hp, Wp = x.shape
        # the two lifting backward passes wrote the halves of ONE [B, h, w, 2 nc] buffer (lifting._LiftFn): take it as it is
        if (d1.dtype == F32 and d2.dtype == F32 and d2.data_ptr() == d1.data_ptr() + 4 * nc and d1.stride() == d2.stride()
                and d1.stride() == (h * w * 2 * nc, 1, w * 2 * nc, 2 * nc)):
            dout = torch.as_strided(d1, (B, h, w, 2 * nc), (h * w * 2 * nc, w * 2 * nc, 2 * nc, 1))
        else:
            dout = torch.empty((B, h, w, 2 * nc), dtype=F32, device=x.device)  # NHWC
            dout[..., :nc] = d1.permute(0, 2, 3, 1)
            dout[..., nc:] = d2.permute(0, 2, 3, 1)
        dx = torch.empty_like(x)
        dWj = torch.empty_like(Wj)
        ws = _lib.workspace.get(int(L.mm_head_ws_bytes(B, h, w, Hp, Wp, C, 2 * nc)), x.device)
        check(L.mm_head_bwd(ptr(x), B, Hp, Wp, C, h, w, C, ptr(Wj), 2 * nc, ptr(dout), ptr(dx), ptr(dWj), ptr(ws), ws.numel(), stream()),
              "head_bwd")
        from . import lifting

        s1, s2 = lifting.pop_colsum(ctx.index, d1), lifting.pop_colsum(ctx.index, d2)
        if s1 is not None and s2 is not None:  # bias gradients = sums over the points (lifting backward), not over the maps
            db1, db2 = s1, s2
        else:
            db = dout.sum((0, 1, 2))
            db1, db2 = db[:nc], db[nc:]
        if ctx.sinks is not None:  # what autograd's AccumulateGrad would do (in place into the arena slices), then the sinks' hooks
            for q, g in zip(ctx.sinks, (dWj[:nc].reshape(wshape), db1, dWj[nc:].reshape(wshape), db2)):
                q._mm_sink.add_(g)
                gradsink.done(q)
            return dx, None, None, None, None, None, None, None
        return dx, None, None, dWj[:nc].reshape(wshape), db1, dWj[nc:].reshape(wshape), db2, None


def fused_heads(x, h, w, conv_main: nn.Conv2d, conv_aux: nn.Conv2d, index=None):
    """(seg_logit_2d, seg_logit_avg_2d), each fp32 [B, num_classes, h, w], from the decoder output x (NHWC bf16, padded).
    ``index``: the lifting.PixelIndex the two maps will be lifted through (bias gradients = sums over the points)."""
    _need_gpu(x, "fused_heads")
    if fp32_mode():  # AvgPool2d(5, 1, 2) of the cropped map (torch pooling op), then the two 1x1 convolutions in fp32
        # (plain NCHW copy of the crop: torch 2.10+rocm7.0's avg_pool2d BACKWARD returns wrong values for a sliced
        # channels_last input - measured 1.18 relative error against the CPU - the contiguous layout is right)
        pooled = torch.nn.functional.avg_pool2d(x[:, :, :h, :w].contiguous(), 5, 1, 2)
        return (_c2f.Conv2dF32Fn.apply(pooled, conv_main.weight, conv_main.bias, 1, 0),
                _c2f.Conv2dF32Fn.apply(pooled, conv_aux.weight, conv_aux.bias, 1, 0))
    return _HeadsFn.apply(x, h, w, conv_main.weight, conv_main.bias, conv_aux.weight, conv_aux.bias, index)
