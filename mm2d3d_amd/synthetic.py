"""Deterministic synthetic scenes with the shapes of the reference's datasets.

No dataset exists in the container, so benchmarks and parity tests use LiDAR-sweep-shaped clouds:
  NuScenes-shaped : 32 beams x 1090 azimuths  = 34,880 points
  KITTI-shaped    : 64 beams x 1900 azimuths  = 121,600 points
voxelised exactly like the reference loader does (lib/utils/augmentation_3d.py:83-158 followed by the
int cast and in-range mask of lib/dataset/nuscenes_dataloader.py:323-332), collated like
lib/dataset/__init__.py:27-123 (batch index appended as LAST coordinate column).
"""
from __future__ import annotations

import numpy as np
import torch

from .voxelize import augment_and_scale_3d, voxelize_points

SHAPES = {
    # name: (n_beams, n_az, elev_lo_deg, elev_hi_deg, sensor_height_m)
    "nuscenes": (32, 1090, -30.67, 10.67, 1.84),
    "kitti": (64, 1900, -24.8, 2.0, 1.73),
}


def lidar_sweep(seed: int, shape: str = "nuscenes") -> np.ndarray:
    """fp32 [n,3] xyz of one synthetic sweep: ground plane + 64 random walls, range noise 2 cm, r < 70 m."""
    n_beams, n_az, lo, hi, h = SHAPES[shape]
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(lo, hi, n_beams))
    az = np.linspace(0.0, 2.0 * np.pi, n_az, endpoint=False)
    E, A = np.meshgrid(elev, az, indexing="ij")
    wall = rng.uniform(5.0, 50.0, 64)[np.floor(A / (2.0 * np.pi) * 64).astype(np.int64) % 64]
    with np.errstate(divide="ignore", invalid="ignore"):
        r_ground = np.where(E < 0, h / np.sin(-E), np.inf)
    r = np.minimum(r_ground, wall / np.cos(E)) + rng.normal(0.0, 0.02, E.shape)
    keep = r < 70.0
    r, E, A = r[keep], E[keep], A[keep]
    xyz = np.stack([r * np.cos(E) * np.cos(A), r * np.cos(E) * np.sin(A), r * np.sin(E)], 1)
    return xyz.astype(np.float32)


def make_scene(seed: int, shape: str = "nuscenes", img_hw=(302, 480), num_classes: int = 6, scale: int = 20,
               full_scale: int = 4096, augment: bool = False, downsample: int = 0):
    """One sample dict with the keys the reference's ``__getitem__`` produces (nuscenes_dataloader.py:236-369)."""
    rng = np.random.default_rng(seed + 7_000_000)
    pts = lidar_sweep(seed, shape)
    if downsample and downsample < len(pts):
        pts = pts[np.sort(rng.choice(len(pts), downsample, replace=False))]
    H, W = img_hw
    n = len(pts)
    img = rng.random((3, H, W), dtype=np.float32)
    img_indices = np.stack([rng.integers(0, H, n), rng.integers(0, W, n)], 1).astype(np.int64)
    depth = np.zeros((H, W), np.float32)
    depth[img_indices[:, 0], img_indices[:, 1]] = np.linalg.norm(pts, axis=1)  # last write wins (loader :275-276)
    seg_label = rng.integers(0, num_classes, n).astype(np.int64)
    seg_label[rng.random(n) < 0.05] = -100
    if augment:
        st = np.random.get_state()
        np.random.seed(seed)
        aug = dict(noisy_rot=0.1, flip_x=0.5, rot_z=6.2831, transl=True)
    else:
        aug = {}
    coords, min_value, offset, rot = augment_and_scale_3d(pts, scale, full_scale, **aug)
    if augment:
        np.random.set_state(st)
    coords, idxs = voxelize_points(coords, full_scale)
    img_indices = img_indices[idxs]
    return {
        "coords": coords,
        "points": pts[idxs],
        "seg_label": seg_label[idxs],
        "img": img,
        "img_indices": img_indices,
        "depth": depth[None],
        "feats": np.ascontiguousarray(img[:, img_indices[:, 0], img_indices[:, 1]].T),
        "min_value": min_value, "offset": offset, "rot_matrix": rot,
    }


def collate(samples, device=None):
    """Batch dict in the reference's collate format (lib/dataset/__init__.py:95-121), pseudo-label keys included (:30-35,
    53-58, 91-95: ``pseudo_label_3d`` stays an empty list when the samples carry none)."""
    locs, feats, labels, imgs, depths, idxs = [], [], [], [], [], []
    pselab = "pseudo_label_2d" in samples[0]
    ps2d, ps3d, psens = [], [], []
    for b, s in enumerate(samples):
        c = torch.from_numpy(s["coords"])
        locs.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int64)], 1))
        feats.append(torch.from_numpy(s["feats"]))
        labels.append(torch.from_numpy(s["seg_label"]))
        imgs.append(torch.from_numpy(s["img"]))
        depths.append(torch.from_numpy(s["depth"]))
        idxs.append(s["img_indices"])
        if pselab:
            ps2d.append(torch.from_numpy(s["pseudo_label_2d"]))
            if s["pseudo_label_3d"] is not None:
                ps3d.append(torch.from_numpy(s["pseudo_label_3d"]))
            psens.append(torch.from_numpy(s["pseudo_label_ensemble"]))
    out = {
        "x": [torch.cat(locs, 0), torch.cat(feats, 0)],
        "seg_label": torch.cat(labels, 0),
        "img": torch.stack(imgs),
        "depth": torch.stack(depths),
        "img_indices": idxs,  # list of numpy int64 [n_i, 2] (row, col), as in the reference
    }
    if pselab:
        out["pseudo_label_2d"] = torch.cat(ps2d, 0)
        out["pseudo_label_3d"] = torch.cat(ps3d, 0) if ps3d else ps3d
        out["pseudo_label_ensemble"] = torch.cat(psens, 0)
    if device is not None:
        out["x"] = [out["x"][0].to(device), out["x"][1].to(device)]
        for k in ("seg_label", "img", "depth"):
            out[k] = out[k].to(device)
    return out


def make_batch(config_id: int, n_scenes: int, shape="nuscenes", img_hw=(302, 480), num_classes=6, rank=0, device=None,
               augment=False, first_scene=0, downsample=0):
    """Scene seed = 1000*config_id + scene_idx + 100000*rank."""
    samples = [
        make_scene(1000 * config_id + first_scene + i + 100000 * rank, shape, img_hw, num_classes, augment=augment,
                   downsample=downsample)
        for i in range(n_scenes)
    ]
    return collate(samples, device)
