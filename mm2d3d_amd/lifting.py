"""2D -> 3D lifting: per-point gather of the dense logits at each point's pixel.

Reference: ``segm.permute(0,2,3,1)[i][img_indices[i][:,0], img_indices[i][:,1]]`` per sample + ``cat``
(2d_net/model.py:131-137, 166-173); backward = ``index_put_(accumulate=True)`` (duplicate pixels add up).
Here one gather kernel serves the whole batch; the backward is a segmented sum over runs of equal pixels in a
stable key-sorted point order - no float atomics, bit-stable.  The index (one stable sort of the pixel keys) is built
on the GPU once per batch; the host only concatenates the numpy ``img_indices`` and never needs a device->host sync.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream

F32 = torch.float32


_STAGING = {}


def _staging(dev, n_int64):
    """One of two persistent pinned int64 buffers of the device, free for reuse (its previous upload has finished)."""
    ring = _STAGING.setdefault(dev.index, {"next": 0, "slots": [None, None]})
    i = ring["next"]
    ring["next"] = 1 - i
    slot = ring["slots"][i]
    if slot is not None:
        slot["event"].synchronize()
    if slot is None or slot["buf"].numel() < n_int64:
        slot = ring["slots"][i] = {"buf": torch.empty(max(int(n_int64 * 1.25), 1024), dtype=torch.int64).pin_memory(),
                                   "event": torch.cuda.Event()}
    return slot


class PixelIndex:
    """Batch-level index built from ``img_indices``: a list of int64 [n_i, 2] = (row, col) per scene, either numpy arrays
    (the reference's collate, lib/dataset/__init__.py:74,111) or tensors already on the device (the GPU data path,
    mm2d3d_amd/dataprep.prepare_batch / datasets.gpu_batch).  Device tensors are concatenated on the device: no host copy,
    no synchronisation; their bounds were checked by the kernel that produced them (``k_proj_index`` sets the batch's
    error flag) and are checked again here on the device, reported through :meth:`check`."""

    def __init__(self, img_indices, H, W, device):
        dev = torch.device(device)
        self.H, self.W, self.device = H, W, device
        self._bad = None
        if len(img_indices) and all(isinstance(ix, torch.Tensor) and ix.is_cuda for ix in img_indices):
            counts = [int(ix.shape[0]) for ix in img_indices]
            self.n = int(sum(counts))
            rc_d = torch.cat([ix.reshape(-1, 2) for ix in img_indices], 0).to(dev, torch.int64)
            cnt_d = torch.tensor(counts, dtype=torch.int64).to(dev, non_blocking=True)
            if self.n:
                self._bad = ((rc_d < 0).any() | (rc_d[:, 0] >= H).any() | (rc_d[:, 1] >= W).any())
                # whatever the caller passed, the gather / scatter kernels only ever see addresses inside the map
                rc_d = torch.stack([rc_d[:, 0].clamp(0, H - 1), rc_d[:, 1].clamp(0, W - 1)], 1)
            rows = img_indices
        else:
            rows = [np.asarray(ix.cpu() if isinstance(ix, torch.Tensor) else ix, dtype=np.int64).reshape(-1, 2) for ix in img_indices]
            for ix in rows:  # the reference asserts these bounds in the loader (nuscenes_dataloader.py:280-283)
                if len(ix) and (ix.min() < 0 or ix[:, 0].max() >= H or ix[:, 1].max() >= W):
                    raise IndexError("img_indices out of the image bounds")
            counts = [len(ix) for ix in rows]
            self.n = int(sum(counts))
            rc = np.concatenate(rows, 0) if rows else np.zeros((0, 2), np.int64)
            # Host -> device from PINNED staging, asynchronously on the current stream: a copy from pageable memory makes the host
            # wait until everything queued before it has run (measured: 12 ms per step inside this constructor when it was called
            # at the end of the 2D forward).  The staging buffers are persistent (two per device, reused alternately once their last
            # copy has completed): `Tensor.pin_memory()` registers fresh host pages on every call, 4.6-30 ms for the 9 MB of a
            # 16-scene batch.  No copy stream: the step stays on ONE stream (the single-launch batch norms need that, fused_bn.h).
            cur = torch.cuda.current_stream(dev)
            nb = len(counts)
            stage = _staging(dev, nb + 2 * self.n)
            host = stage["buf"].numpy()
            host[:nb] = counts
            host[nb : nb + 2 * self.n] = rc.reshape(-1)
            both = stage["buf"][: nb + 2 * self.n].to(dev, non_blocking=True)
            stage["event"].record(cur)
            cnt_d, rc_d = both[:nb], both[nb:].view(-1, 2)
        b = torch.repeat_interleave(torch.arange(len(rows), device=device), cnt_d, output_size=self.n)
        self.b, self.r, self.c = b, rc_d[:, 0], rc_d[:, 1]
        key = (b * H + self.r) * W + self.c                   # flat pixel id b*H*W + r*W + c
        self.skey, self.order = torch.sort(key, stable=True)  # stable: ascending point order inside a pixel
        first = torch.ones(self.n, dtype=torch.bool, device=device)
        if self.n > 1:
            first[1:] = self.skey[1:] != self.skey[:-1]
        self.first = first.to(torch.uint8)
        self._cache = {}

    def check(self):
        """Raises IndexError if a device-side index was out of the image (one small read-back; call it off the hot path)."""
        if self._bad is not None and bool(self._bad.item()):
            raise IndexError("img_indices out of the image bounds")

    def offsets(self, sb, sy, sx, sorted_order=False):
        """Element offset of channel 0 of every point's pixel in a [B,C,H,W] map with strides (sb, *, sy, sx); with
        ``sorted_order`` the offsets follow the key-sorted order (for the backward)."""
        k = (int(sb), int(sy), int(sx), sorted_order)
        if k not in self._cache:
            off = self.b * int(sb) + self.r * int(sy) + self.c * int(sx)
            self._cache[k] = off.index_select(0, self.order).contiguous() if sorted_order else off.contiguous()
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
        soff = index.offsets(dseg.stride(0), dseg.stride(2), dseg.stride(3), sorted_order=True)
        check(L.mm_lift_scatter_runs(ptr(dout), C, ptr(index.order), ptr(index.first), ptr(soff), index.n, dseg.stride(1), ptr(dseg),
                                     stream()), "lift_scatter_runs")
        return dseg, None


def lift(seg, index: PixelIndex):
    """seg [B,C,H,W] (any strides) -> [N_points, C] in the concatenated point order of the batch."""
    return _LiftFn.apply(seg, index)
