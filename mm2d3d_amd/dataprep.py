"""GPU-side sample preparation and collation (csrc/dataprep.hip).

The reference prepares every sample in DataLoader workers with numpy (``NuScenesLidarSegSCN.__getitem__``,
lib/dataset/nuscenes_dataloader.py:236-369; ``augment_and_scale_3d``, lib/utils/augmentation_3d.py:83-158) and
collates on the host (``collate_scn_base``, lib/dataset/__init__.py:27-123).  Once a training step takes ~50 ms for 16
scenes that host path is the next wall, so here the per-point work of a whole batch runs as a short chain of HIP kernels
on arrays that were uploaded once (pinned host buffers -> device), and the result is the reference's batch dict, on the
device, bit-identical to the host restatement (mm2d3d_amd/projection.py + synthetic.collate):

    host (O(1) per scene, the reference's numpy RNG order)     device (per point / per pixel)
    -----------------------------------------------------     -------------------------------------------------------
    fliplr draw, rotation matrix, rand(3) translation draws     points.rot, *scale, -min, +offset, int cast, range mask,
    image decode / resize / normalise (caller)                  order-preserving compaction + batch index column (a1, a2)
                                                                pixel indices, fliplr remap, last-write-wins depth and
                                                                2D label maps, RGB features under the points (a16)
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream


def augmentation_draws(noisy_rot=0.0, flip_x=0.0, flip_y=0.0, rot_z=0.0, rot_y=0.0, transl=False):
    """The random part of ``augment_and_scale_3d`` (augmentation_3d.py:106-155) in the reference's draw order:
    randn(3,3) | randint x | randint y | rand z | rand y | rand(3).  Returns (rot float32 [3,3], u float64 [3] or None)."""
    rot = np.eye(3, dtype=np.float32)
    if noisy_rot > 0:
        rot += np.random.randn(3, 3) * noisy_rot
    if flip_x > 0:
        rot[0][0] *= np.random.randint(0, 2) * 2 - 1
    if flip_y > 0:
        rot[1][1] *= np.random.randint(0, 2) * 2 - 1
    if rot_z > 0:
        t = np.random.rand() * rot_z
        c, s = np.cos(t), np.sin(t)
        rot = rot.dot(np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float32))
    if rot_y > 0:
        t = np.random.rand() * rot_y
        c, s = np.cos(t), np.sin(t)
        rot = rot.dot(np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float32))
    u = np.random.rand(3) if transl else None
    return rot, u


def _offsets(lengths, dev):
    off = np.zeros(len(lengths) + 1, np.int32)
    np.cumsum(lengths, out=off[1:])
    return off, torch.from_numpy(off).to(dev)


def voxelize_batch(points, lengths, rots, us, scale=20, full_scale=4096):
    """points fp32 [n_total,3] on the GPU (scenes back to back), lengths = points per scene, rots / us = the per-scene
    draws of :func:`augmentation_draws`.  Returns dict(locs int64 [kept,4], keep int32 [kept], counts list, min_value
    fp32 [B,3], offset fp64 [B,3]); one small device->host copy (the kept counts) sizes the outputs."""
    _lib.require_cuda(points, "points")
    if points.dtype != torch.float32:
        raise TypeError("voxelize_batch: points must be float32 (the dtype the reference's pickles hold)")
    L = _lib.lib()
    dev = points.device
    points = points.contiguous()
    B, n = len(lengths), int(sum(lengths))
    assert points.shape == (n, 3)
    off_h, off_d = _offsets(lengths, dev)
    transl = any(u is not None for u in us)
    if transl and not all(u is not None for u in us):
        raise ValueError("voxelize_batch: translation must be drawn for every scene of the batch or for none")
    rot_d = torch.from_numpy(np.stack([np.asarray(r, np.float32).reshape(9) for r in rots])).to(dev)
    u_d = torch.from_numpy(np.stack([np.asarray(u if u is not None else np.zeros(3), np.float64) for u in us])).to(dev)
    locs = torch.empty((n, 4), dtype=torch.int64, device=dev)
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    counts = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    minv = torch.empty((B, 3), dtype=torch.float32, device=dev)
    offset = torch.empty((B, 3), dtype=torch.float64, device=dev)
    ws = _lib.workspace.get(int(L.mm_voxelize_ws_bytes(n, B)), dev)
    check(L.mm_voxelize_batch(ptr(points), ptr(off_d), off_h.ctypes.data, B, ptr(rot_d), ptr(u_d), 1 if transl else 0, float(scale),
                              int(full_scale), ptr(locs), ptr(keep), ptr(counts), ptr(minv), ptr(offset), ptr(ws), ws.numel(), stream()),
          "voxelize_batch")
    ch = counts.cpu().tolist()  # the only read-back: how many points survived the range mask
    kept = ch[B]
    return dict(locs=locs[:kept], keep=keep[:kept], counts=ch[:B], counts_dev=counts, min_value=minv, offset=offset,
                scene_off=(off_h, off_d))


def project_batch(points_img, depth_vals, labels, lengths, H, W, flips=None, want_seg2d=False, scene_off=None):
    """points_img fp32 [n_total,2] (row, col; already scaled to the network image), depth_vals fp32 [n_total] (camera z).
    Returns (img_indices int64 [n_total,2], depth fp32 [B,1,H,W], seg2d fp64 [B,H,W] or None)."""
    _lib.require_cuda(points_img, "points_img")
    L = _lib.lib()
    dev = points_img.device
    B = len(lengths)
    off_h, off_d = scene_off if scene_off is not None else _offsets(lengths, dev)
    n = int(off_h[B])
    points_img = points_img.to(torch.float32).contiguous()
    depth_vals = depth_vals.to(torch.float32).contiguous()
    flip_d = torch.tensor([1 if f else 0 for f in flips], dtype=torch.uint8, device=dev) if flips is not None and any(flips) else None
    idx = torch.empty((n, 2), dtype=torch.int64, device=dev)
    depth = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    seg2d = torch.empty((B, H, W), dtype=torch.float64, device=dev) if want_seg2d else None
    winner = torch.empty(B * H * W, dtype=torch.int32, device=dev)
    err = torch.zeros(1, dtype=torch.int32, device=dev)
    check(L.mm_project_batch(ptr(points_img), ptr(depth_vals), ptr(labels), ptr(off_d), off_h.ctypes.data, B, H, W, ptr(flip_d), ptr(idx),
                             ptr(depth), ptr(seg2d), ptr(winner), ptr(err), stream()), "project_batch")
    return idx, depth, seg2d, err


def prepare_batch(scenes, scale=20, full_scale=4096, augmentation=None, fliplr=0.0, want_seg2d=False, device="cuda", use_rgb=True):
    """The reference's ``__getitem__`` (per scene) + ``collate_scn_base`` for a list of decoded scenes, on the GPU.

    Each scene: dict(points [n,3] f32 = the coordinates that are voxelised (camera or LiDAR frame, as the dataset is
    configured), points_img [n,2] (row, col) scaled to the network image, depth [n] = camera z, seg_label [n] int64,
    img [3,H,W] f32 already normalised and, when this scene's fliplr draw says so, NOT yet flipped).  RNG draws per
    scene in the reference's order: fliplr ``rand()`` first (nuscenes_dataloader.py:291), then ``augment_and_scale_3d``'s;
    a scene that carries ``draws = (flip, rot, u)`` was drawn by the caller (datasets.gpu_batch draws scene by scene, between
    each scene's crop draws, as the reference's ``__getitem__`` sequence does).  ``use_rgb=False``: the constant feature of
    nuscenes_dataloader.py:365-368, ones [n, 1] with n = the scene's point count BEFORE the range mask (as in the reference).
    Returns the batch dict of lib/dataset/__init__.py:95-121 with every tensor on ``device`` (img_indices: list of
    device int64 [n_i,2]; use ``[t.cpu().numpy() for t in ...]`` where numpy arrays are required)."""
    L = _lib.lib()
    dev = torch.device(device)
    aug = dict(augmentation or {})
    B = len(scenes)
    lengths = [int(s["points"].shape[0]) for s in scenes]
    flips, rots, us = [], [], []
    for sc in scenes:
        if "draws" in sc:
            f, r, u = sc["draws"]
        else:
            f = bool(np.random.rand() < fliplr)
            r, u = augmentation_draws(**aug)
        flips.append(bool(f))
        rots.append(r)
        us.append(u)
    cat = lambda key, dt: torch.from_numpy(np.ascontiguousarray(np.concatenate([np.asarray(s[key]) for s in scenes], 0).astype(dt))).to(dev)
    pts = cat("points", np.float32)
    pimg = cat("points_img", np.float32)
    dvals = cat("depth", np.float32)
    labels = cat("seg_label", np.int64)
    img = torch.stack([torch.as_tensor(s["img"]) for s in scenes]).to(dev, torch.float32)
    if any(flips):
        img = torch.stack([im.flip(-1) if f else im for im, f in zip(img, flips)])
    H, W = img.shape[-2:]
    vox = voxelize_batch(pts, lengths, rots, us, scale, full_scale)
    idx_all, depth, seg2d, err = project_batch(pimg, dvals, labels, lengths, H, W, flips, want_seg2d, vox["scene_off"])
    kept = vox["locs"].shape[0]
    idx = torch.empty((kept, 2), dtype=torch.int64, device=dev)
    lab = torch.empty(kept, dtype=torch.int64, device=dev)
    feats = torch.empty((kept, img.shape[1]), dtype=torch.float32, device=dev) if use_rgb else None
    pkept = torch.empty((kept, 3), dtype=torch.float32, device=dev)
    check(L.mm_collect_points(ptr(vox["keep"]), ptr(vox["counts_dev"][B:]), kept, ptr(vox["locs"]), ptr(idx_all), ptr(labels), ptr(img.contiguous()),
                              img.shape[1], H, W, ptr(pts), ptr(idx), ptr(lab), ptr(feats), ptr(pkept), stream()), "collect_points")
    if int(err.item()) != 0:
        raise AssertionError("projected point outside the image (nuscenes_dataloader.py:279-283)")
    bounds = np.concatenate([[0], np.cumsum(vox["counts"])])
    if not use_rgb:
        feats = torch.ones((int(sum(lengths)), 1), dtype=torch.float32, device=dev)
    out = {
        "x": [vox["locs"], feats],
        "seg_label": lab,
        "img": img,
        "depth": depth,
        "img_indices": [idx[bounds[i] : bounds[i + 1]] for i in range(B)],
        "points": [pkept[bounds[i] : bounds[i + 1]] for i in range(B)],
        "min_values": vox["min_value"], "offsets": vox["offset"], "rotation_matrices": np.stack(rots), "fliplr": flips,
        "keep": vox["keep"],  # original row (in the concatenated input) of every kept point
    }
    if seg2d is not None:
        out["seg_labels_2d"] = seg2d
    return out
