"""2D -> 3D lifting: per-point gather of the dense logits at each point's pixel.

Reference: ``segm.permute(0,2,3,1)[i][img_indices[i][:,0], img_indices[i][:,1]]`` per sample + ``cat``
(2d_net/model.py:131-137, 166-173); backward = ``index_put_(accumulate=True)`` (duplicate pixels add up).
Here one gather kernel serves the whole batch; the backward is a segmented sum over a pixel->points CSR
prepared once per batch on the host from the (numpy) ``img_indices`` - no float atomics, bit-stable.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream

F32 = torch.float32


class PixelIndex:
    """Batch-level index built from ``img_indices`` (list of numpy int64 [n_i, 2] = (row, col))."""

    def __init__(self, img_indices, H, W, device):
        rows = [np.asarray(ix, dtype=np.int64) for ix in img_indices]
        for ix in rows:  # the reference asserts these bounds in the loader (nuscenes_dataloader.py:280-283)
            if len(ix) and (ix.min() < 0 or ix[:, 0].max() >= H or ix[:, 1].max() >= W):
                raise IndexError("img_indices out of the image bounds")
        b = np.concatenate([np.full(len(ix), i, np.int64) for i, ix in enumerate(rows)]) if rows else np.zeros(0, np.int64)
        rc = np.concatenate(rows, 0) if rows else np.zeros((0, 2), np.int64)
        self.n = len(b)
        self.H, self.W = H, W
        self.key = (b * H + rc[:, 0]) * W + rc[:, 1]  # flat pixel id b*H*W + r*W + c
        order = np.argsort(self.key, kind="stable")   # stable: ascending point order inside a pixel
        sk = self.key[order]
        first = np.ones(len(sk), bool)
        first[1:] = sk[1:] != sk[:-1]
        self.ukey = sk[first]
        off = np.flatnonzero(first)
        self.csr_off = torch.from_numpy(np.concatenate([off, [len(sk)]]).astype(np.int32)).to(device)
        self.csr_pts = torch.from_numpy(order.astype(np.int32)).to(device)
        self.device = device
        self._cache = {}

    def offsets(self, batch_stride, unique=False):
        """Element offset of channel 0 of every point's (or unique pixel's) pixel in a [B,C,H,W] map whose H,W dims are
        dense (stride W, 1) and whose batch stride is ``batch_stride`` elements."""
        k = (int(batch_stride), unique)
        if k not in self._cache:
            HW = self.H * self.W
            key = self.ukey if unique else self.key
            self._cache[k] = torch.from_numpy((key // HW) * int(batch_stride) + key % HW).to(self.device)
        return self._cache[k]


class _LiftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, index: PixelIndex):
        _lib.require_cuda(seg, "seg")
        L = _lib.lib()
        seg = seg.to(F32)
        B, C, H, W = seg.shape
        assert (H, W) == (index.H, index.W)
        if seg.stride(3) != 1 or seg.stride(2) != W or seg.stride(1) != H * W:  # channel slices of an NCHW map are fine as is
            seg = seg.contiguous()
        pix = index.offsets(seg.stride(0))
        out = torch.empty((index.n, C), dtype=F32, device=seg.device)
        check(L.mm_lift_gather(ptr(seg), H * W, ptr(pix), index.n, C, ptr(out), stream()), "lift_gather")
        ctx.index, ctx.shape = index, seg.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        index = ctx.index
        B, C, H, W = ctx.shape
        dout = dout.to(F32).contiguous()
        upix = index.offsets(C * H * W, unique=True)
        dseg = torch.zeros(ctx.shape, dtype=F32, device=dout.device)
        check(L.mm_lift_scatter(ptr(dout), C, ptr(upix), ptr(index.csr_off), ptr(index.csr_pts), len(index.ukey), H * W, ptr(dseg),
                                stream()), "lift_scatter")
        return dseg, None


def lift(seg, index: PixelIndex):
    """seg [B,C,H,W] -> [N_points, C] in the concatenated point order of the batch."""
    return _LiftFn.apply(seg, index)
