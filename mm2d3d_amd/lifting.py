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
        self._joint = {}  # joint gradient buffers of maps that are channel slices of one NHWC buffer (see _LiftFn.backward)
        # per-channel sums of the gradient maps the lifting backward passes of THIS batch produced, keyed by the map's address
        # (nn2d._HeadsFn.backward takes them for the 1x1 heads' bias gradients).  Scoped to the batch's index object (a
        # process-wide address-keyed table could hand a stale sum to an unrelated map the allocator placed at the same address)
        self._colsums = {}
        self._err = None  # device flag: an index was outside the map (device-side lists only; host lists are checked on the host)
        if len(img_indices) and all(isinstance(ix, torch.Tensor) and ix.is_cuda for ix in img_indices):
            counts = [int(ix.shape[0]) for ix in img_indices]
            self.n = int(sum(counts))
            rc_d = torch.cat([ix.reshape(-1, 2) for ix in img_indices], 0).to(dev, torch.int64)
            cnt_d = torch.tensor(counts, dtype=torch.int64).to(dev, non_blocking=True)
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
        # one device-side build (csrc/lift.hip): pixel keys, stable (key, point) radix sort - three launches (rounds 1-3: ~25 torch
        # launches: repeat_interleave, arange, key arithmetic, sort, run flags, offset tables)
        L = _lib.lib()
        self.rc, self.counts = rc_d.contiguous(), cnt_d.contiguous()
        self.key = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        self.skey = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        self.order = torch.empty(max(self.n, 1), dtype=torch.int32, device=dev)
        if self._err is None:
            self._err = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = _lib.workspace.get(int(L.mm_lift_index_ws_bytes(self.n)), dev)
        from .scn import metadata as _md

        check(L.mm_lift_index(ptr(self.rc), ptr(self.counts), len(counts), self.n, H, W, _md.NO_SPIN[0], ptr(self.key), ptr(self.skey),
                              ptr(self.order), ptr(self._err), ptr(ws), ws.numel(), stream()), "lift_index")

    def check(self):
        """Raises IndexError if a device-side index was out of the image (one small read-back; call it off the hot path)."""
        if self._err is not None and bool(self._err.item()):
            raise IndexError("img_indices out of the image bounds")


class _LiftFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seg, index: PixelIndex):
        _lib.require_cuda(seg, "seg")
        L = _lib.lib()
        seg = seg.to(F32)
        B, C, H, W = seg.shape
        assert (H, W) == (index.H, index.W)
        out = torch.empty((index.n, C), dtype=F32, device=seg.device)
        check(L.mm_lift_gather_key(ptr(seg), seg.stride(0), seg.stride(2), seg.stride(3), seg.stride(1), ptr(index.key), index.n, C, H, W,
                                   ptr(out), stream()), "lift_gather")
        ctx.index, ctx.shape = index, seg.shape
        # seg may be a channel slice of a wider NHWC map (the two heads of nn2d.fused_heads are the halves of ONE [B, h, w, 2 nc]
        # buffer): remember the layout, the backward then writes its slice of ONE joint gradient buffer instead of a map of its own
        ctx.slice = None
        if seg.stride(1) == 1 and seg.stride(3) > C and seg.stride(2) == W * seg.stride(3) and seg.stride(0) == H * W * seg.stride(3):
            off = seg.storage_offset() % seg.stride(3)
            ctx.slice = (seg.data_ptr() - 4 * off, int(seg.stride(3)), int(off))
        return out

    @staticmethod
    def backward(ctx, dout):
        L = _lib.lib()
        index = ctx.index
        B, C, H, W = ctx.shape
        dout = dout.to(F32).contiguous()
        # gradient map in NHWC (what the fused heads' backward reads coalesced); logical shape stays [B,C,H,W]
        if ctx.slice is not None:
            base, pitch, off = ctx.slice
            joint = index._joint.get(base)
            if joint is None or joint.shape != (B, H, W, pitch):
                joint = index._joint[base] = torch.zeros((B, H, W, pitch), dtype=F32, device=dout.device)
            dseg = joint[..., off:off + C].permute(0, 3, 1, 2)
        else:
            dseg = torch.zeros((B, H, W, C), dtype=F32, device=dout.device).permute(0, 3, 1, 2)
        check(L.mm_lift_scatter_key(ptr(dout), C, ptr(index.order), ptr(index.skey), index.n, H, W, dseg.stride(0), dseg.stride(2),
                                    dseg.stride(3), dseg.stride(1), ptr(dseg), stream()), "lift_scatter")
        # the per-channel sum of the dense gradient map = the sum over the points (what a bias behind the map needs): [N, C] instead
        # of a reduction over the whole map
        index._colsums[dseg.data_ptr()] = (dout.sum(0), dseg.shape)
        return dseg, None


def pop_colsum(index, dmap):
    """The per-channel sum the lifting backward of ``index``'s batch filed for gradient map ``dmap`` (popped), or None."""
    if index is None:
        return None
    hit = index._colsums.pop(dmap.data_ptr(), None)
    if hit is None or tuple(hit[1]) != tuple(dmap.shape):
        return None
    return hit[0]


def lift(seg, index: PixelIndex):
    """seg [B,C,H,W] (any strides) -> [N_points, C] in the concatenated point order of the batch."""
    return _LiftFn.apply(seg, index)
