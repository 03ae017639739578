"""Host-side voxeliser with the reference's semantics.

Restates lib/utils/augmentation_3d.py:83-158 (``augment_and_scale_3d``: numpy RNG call order, float32
rotation matrix, float64 promotion of the coordinates once ``randn`` noise is added) and the int cast +
in-range mask of lib/dataset/nuscenes_dataloader.py:323-332.  It is host code (numpy) in the reference and
stays host code here: it runs once per sample in the loader, before the batch reaches the GPU.
Pinned bit-exactly against the reference by tests/golden/voxelize_*.npz (tests/test_golden_leaves.py).
"""
from __future__ import annotations

import numpy as np


def augment_and_scale_3d(points, scale, full_scale, noisy_rot=0.0, flip_x=0.0, flip_y=0.0, rot_z=0.0, rot_y=0.0,
                         transl=False):
    """points [n,3] metres -> (coords float, min_value [3], offset [3] f64, rot_matrix [3,3] f32).

    RNG draws happen in the reference's order: randn(3,3) | randint x | randint y | rand z | rand y | rand(3).
    """
    rot = np.eye(3, dtype=np.float32)
    if noisy_rot > 0 or flip_x > 0 or flip_y > 0 or rot_z > 0 or rot_y > 0:
        if noisy_rot > 0:
            rot += np.random.randn(3, 3) * noisy_rot  # in-place: result stays float32
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
        points = points.dot(rot)
    coords = points * scale
    min_value = coords.min(0)
    coords -= min_value
    offset = np.zeros(3)
    if transl:
        offset = np.clip(full_scale - coords.max(0) - 0.001, a_min=0, a_max=None) * np.random.rand(3)
        coords += offset
    return coords, min_value, offset, rot


def voxelize_points(coords_float, full_scale):
    """astype(int64) truncation, keep rows with 0 <= c < full_scale (nuscenes_dataloader.py:323-332)."""
    coords = coords_float.astype(np.int64)
    idxs = (coords.min(1) >= 0) * (coords.max(1) < full_scale)
    return coords[idxs], idxs
