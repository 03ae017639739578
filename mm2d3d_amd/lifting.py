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

    def offsets(self, sb, sy, sx, unique=False):
        """Element offset of channel 0 of every point's (or unique pixel's) pixel in a [B,C,H,W] map with strides
        (sb, *, sy, sx) - NCHW and NHWC maps (and channel slices of either) are both served without a copy."""
        k = (int(sb), int(sy), int(sx), unique)
        if k not in self._cache:
            HW = self.H * self.W
            key = self.ukey if unique else self.key
            b, r = key // HW, key % HW
            self._cache[k] = torch.from_numpy(b * int(sb) + (r // self.W) * int(sy) + (r % self.W) * int(sx)).to(self.device)
        return self._cache[k]


class _LiftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, index: PixelIndex):
        _lib.require_cuda(seg, "seg")
        L = _lib.lib()
        seg = seg.to(F32)
        B, C, H, W = seg.shape
        assert (H, W) == (index.H, index.W)
        pix = index.offsets(seg.stride(0), seg.stride(2), seg.stride(3))
        out = torch.empty((index.n, C), dtype=F32, device=seg.device)
        check(L.mm_lift_gather(ptr(seg), seg.stride(1), ptr(pix), index.n, C, ptr(out), stream()), "lift_gather")
        ctx.index, ctx.shape = index, seg.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        index = ctx.index
        B, C, H, W = ctx.shape
        dout = dout.to(F32).contiguous()
        # gradient map in NHWC (what the fused heads' backward reads coalesced); logical shape stays [B,C,H,W]
        dseg = torch.zeros((B, H, W, C), dtype=F32, device=dout.device).permute(0, 3, 1, 2)
        upix = index.offsets(dseg.stride(0), dseg.stride(2), dseg.stride(3), unique=True)
        check(L.mm_lift_scatter(ptr(dout), C, ptr(upix), ptr(index.csr_off), ptr(index.csr_pts), len(index.ukey), dseg.stride(1),
                                ptr(dseg), stream()), "lift_scatter")
        return dseg, None


def lift(seg, index: PixelIndex):
    """seg [B,C,H,W] -> [N_points, C] in the concatenated point order of the batch."""
    return _LiftFn.apply(seg, index)
