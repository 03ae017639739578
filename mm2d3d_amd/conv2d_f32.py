"""Exact-fp32 2D layers (csrc/conv2d_f32.hip): the ``precision: 32`` mode of the 2D branch (config/run/test.yaml:8).

The bf16 MFMA path (conv2d.py / nn2d.py) is the training hot path and matches the fp32 oracle to a few 1e-3; north_star
asks for logits within 1e-3 of the fp32 reference, so test / evaluation runs can switch the whole 2D branch to fp32
(``mm2d3d_amd.nn2d.set_precision(32)``): convolutions and transposed convolutions on the fp32 implicit-GEMM kernels of
csrc/conv2d_f32.hip, batch norm on the fp32 row kernels of csrc/bn.hip (an NHWC map is a [B*H*W, C] row matrix).  The
weights are the same fp32 parameters, so a checkpoint runs in either mode.
Maps are torch tensors of logical shape [B, C, H, W] with channels_last strides (NHWC in memory).
"""
from __future__ import annotations

import torch

from . import _lib, gradsink
from ._lib import check, ptr, stream

F32 = torch.float32
CL = torch.channels_last


def as_nhwc_f32(x):
    if x.dtype != F32:
        x = x.to(F32)
    if not x.is_contiguous(memory_format=CL) or (x.shape[1] == 1 and x.stride(1) != 1):
        x = x.contiguous(memory_format=CL)
    return x


def _conv(A, Hi, Wi, Ca, out, Ho, Wo, Cn, KH, KW, so, sgn, off, up, W, wstr, bias=None):
    Bn = A.shape[0]
    check(_lib.lib().mm_conv2d_f32(ptr(A), Bn, Hi, Wi, Ca, Ca, ptr(out), Ho, Wo, Cn, Cn, KH, KW, so, sgn, off, up, ptr(W), *wstr, ptr(bias),
                                   stream()), "conv2d_f32")


def _wgrad(G, Hg, Wg, Cg, A, Hi, Wi, Ca, KH, KW, so, sgn, off, up, dW, wstr, accumulate):
    L = _lib.lib()
    Bn = G.shape[0]
    ws = _lib.workspace.get(int(L.mm_conv2d_f32_wgrad_ws_bytes(Bn * Hg * Wg, Cg, Ca, KH, KW)), G.device)
    check(L.mm_conv2d_f32_wgrad(ptr(G), Bn, Hg, Wg, Cg, Cg, ptr(A), Hi, Wi, Ca, Ca, KH, KW, so, sgn, off, up, ptr(dW), *wstr,
                                1 if accumulate else 0, ptr(ws), ws.numel(), stream()), "conv2d_f32_wgrad")


def _colsum(dy, out, accumulate):
    Bn, C, H, W = dy.shape
    check(_lib.lib().mm_colsum_f32(ptr(dy), C, Bn * H * W, C, ptr(out), 1 if accumulate else 0, stream()), "colsum_f32")


class Conv2dF32Fn(torch.autograd.Function):
    """y = conv2d(x, w, b, stride, padding), square stride / padding, any kernel size and channel counts."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        _lib.require_cuda(x, "x")
        x = as_nhwc_f32(x)
        Bn, Cin, H, W = x.shape
        Cout, _, KH, KW = weight.shape
        Ho, Wo = (H + 2 * padding - KH) // stride + 1, (W + 2 * padding - KW) // stride + 1
        w = weight.detach().to(F32).contiguous()
        y = torch.empty((Bn, Cout, Ho, Wo), dtype=F32, device=x.device, memory_format=CL)
        b = bias.detach().to(F32).contiguous() if bias is not None else None
        # W[co][ci][ky][kx]: n = co, c = ci
        _conv(x, H, W, Cin, y, Ho, Wo, Cout, KH, KW, stride, 1, -padding, 1, w, (Cin * KH * KW, KH * KW, KW, 1), b)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding)
        ctx.wparam = weight if gradsink.claim(ctx, weight, ctx.needs_input_grad[1]) else None
        ctx.bparam = bias if (bias is not None and gradsink.claim(ctx, bias, ctx.needs_input_grad[2])) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding = ctx.cfg
        dy = as_nhwc_f32(dy)
        Bn, Cin, H, W = x.shape
        Cout, _, KH, KW = w.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((Bn, Cin, H, W), dtype=F32, device=x.device, memory_format=CL)
            # din[.., ci] = sum dout[(iy + p - ky) / s, .., co] W[co][ci][ky][kx]: n = ci, c = co
            _conv(dy, Ho, Wo, Cout, dx, H, W, Cin, KH, KW, 1, -1, padding, stride, w, (KH * KW, Cin * KH * KW, KW, 1))
        if ctx.needs_input_grad[1]:
            sink = ctx.wparam._mm_sink if ctx.wparam is not None else None
            dw = sink if sink is not None else torch.empty_like(w)
            _wgrad(dy, Ho, Wo, Cout, x, H, W, Cin, KH, KW, stride, 1, -padding, 1, dw, (Cin * KH * KW, KH * KW, KW, 1), sink is not None)
            if sink is not None:
                gradsink.done(ctx.wparam)
                dw = None
        if ctx.needs_input_grad[2]:
            sink = ctx.bparam._mm_sink if ctx.bparam is not None else None
            db = sink if sink is not None else torch.empty(Cout, dtype=F32, device=x.device)
            _colsum(dy, db, sink is not None)
            if sink is not None:
                gradsink.done(ctx.bparam)
                db = None
        return dx, dw, db, None, None


class ConvTranspose2dF32Fn(torch.autograd.Function):
    """y = conv_transpose2d(x, w, b, stride = kernel size, padding 0) (the decoder's k2 s2 stages)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        _lib.require_cuda(x, "x")
        x = as_nhwc_f32(x)
        Bn, Cin, H, W = x.shape
        _, Cout, KH, KW = weight.shape
        Ho, Wo = (H - 1) * stride + KH, (W - 1) * stride + KW
        w = weight.detach().to(F32).contiguous()
        y = torch.empty((Bn, Cout, Ho, Wo), dtype=F32, device=x.device, memory_format=CL)
        b = bias.detach().to(F32).contiguous() if bias is not None else None
        # out[Y, X, co] = sum [ (Y - ky) % s == 0 ] in[(Y - ky) / s, .., ci] W[ci][co][ky][kx]: n = co, c = ci
        _conv(x, H, W, Cin, y, Ho, Wo, Cout, KH, KW, 1, -1, 0, stride, w, (KH * KW, Cout * KH * KW, KW, 1), b)
        ctx.save_for_backward(x, w)
        ctx.stride = stride
        ctx.wparam = weight if gradsink.claim(ctx, weight, ctx.needs_input_grad[1]) else None
        ctx.bparam = bias if (bias is not None and gradsink.claim(ctx, bias, ctx.needs_input_grad[2])) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        s = ctx.stride
        dy = as_nhwc_f32(dy)
        Bn, Cin, H, W = x.shape
        _, Cout, KH, KW = w.shape
        Ho, Wo = dy.shape[2], dy.shape[3]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((Bn, Cin, H, W), dtype=F32, device=x.device, memory_format=CL)
            # din[y, x, ci] = sum dout[y*s + ky, .., co] W[ci][co][ky][kx]: n = ci, c = co
            _conv(dy, Ho, Wo, Cout, dx, H, W, Cin, KH, KW, s, 1, 0, 1, w, (Cout * KH * KW, KH * KW, KW, 1))
        if ctx.needs_input_grad[1]:
            sink = ctx.wparam._mm_sink if ctx.wparam is not None else None
            dw = sink if sink is not None else torch.empty_like(w)
            # dW[ci][co][ky][kx] = sum in[y, x, ci] dout[y*s + ky, x*s + kx, co]: G = x (n = ci), A = dout (c = co)
            _wgrad(x, H, W, Cin, dy, Ho, Wo, Cout, KH, KW, s, 1, 0, 1, dw, (Cout * KH * KW, KH * KW, KW, 1), sink is not None)
            if sink is not None:
                gradsink.done(ctx.wparam)
                dw = None
        if ctx.needs_input_grad[2]:
            sink = ctx.bparam._mm_sink if ctx.bparam is not None else None
            db = sink if sink is not None else torch.empty(Cout, dtype=F32, device=x.device)
            _colsum(dy, db, sink is not None)
            if sink is not None:
                gradsink.done(ctx.bparam)
                db = None
        return dx, dw, db, None


def batch_norm_f32(x, bn, residual=None, relu=False):
    """nn.BatchNorm2d semantics (+ residual add, + ReLU) on an fp32 NHWC map through the fp32 row kernels of csrc/bn.hip.
    Two statistics groups when a joint [source | target] batch is being processed (mm2d3d_amd/domains.py)."""
    from . import domains
    from .scn import ops

    x = as_nhwc_f32(x)
    Bn, C, H, W = x.shape
    rows = x.permute(0, 2, 3, 1).reshape(Bn * H * W, C)
    use_batch = bn.training or not bn.track_running_stats
    split = domains.current()
    seg_rows = split * H * W if (use_batch and split is not None and 0 < split < Bn) else None
    fused_relu = relu and residual is None
    # scn's momentum is the keep fraction of the running statistics (1 - torch's)
    mom = 1.0 - float(bn.momentum if bn.momentum is not None else 0.1)
    y = ops.BatchNormActFunction.apply(rows, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch, float(bn.eps), mom,
                                       0.0 if fused_relu else 1.0, seg_rows)
    if use_batch and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 2 if seg_rows is not None else 1
    y = y.reshape(Bn, H, W, C).permute(0, 3, 1, 2)
    if residual is not None:
        y = y + as_nhwc_f32(residual)
        if relu:
            y = torch.relu(y)
    return y
