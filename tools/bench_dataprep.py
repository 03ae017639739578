"""Sample preparation of one 16-scene step: the host numpy loader path (mm2d3d_amd/projection.py, as in the reference's
DataLoader workers) against the GPU path (mm2d3d_amd/dataprep.py), NuScenes-shaped scenes at 480x302."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd import dataprep  # noqa: E402
from mm2d3d_amd.projection import make_sample  # noqa: E402
from mm2d3d_amd.synthetic import collate, lidar_sweep  # noqa: E402

H, W, B = 302, 480, 16
aug = dict(noisy_rot=0.1, flip_x=0.5, rot_y=6.2831, transl=True)
rng = np.random.default_rng(0)
scenes, host = [], []
for i in range(B):
    pts = lidar_sweep(i, "nuscenes")
    n = len(pts)
    pimg = np.stack([rng.uniform(0, H - 1e-3, n), rng.uniform(0, W - 1e-3, n)], 1).astype(np.float32)
    lab = rng.integers(0, 6, n).astype(np.int64)
    img = rng.random((H, W, 3), dtype=np.float32)
    scenes.append(dict(points=pts, points_img=pimg, depth=pts[:, 2].copy(), seg_label=lab, img=np.moveaxis(img, -1, 0).copy()))
    host.append((pts, pimg, lab, img))
t0 = time.perf_counter()
for _ in range(3):
    ref = collate([make_sample(p, p, pi, l, np.eye(3), im, camera_coords=True, noisy_rot=0.1, flip_x=0.5, rot=6.2831, transl=True, fliplr=0.5)
                   for p, pi, l, im in host])
t_host = (time.perf_counter() - t0) / 3
dataprep.prepare_batch(scenes, augmentation=aug, fliplr=0.5)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = dataprep.prepare_batch(scenes, augmentation=aug, fliplr=0.5)
torch.cuda.synchronize()
t_gpu = (time.perf_counter() - t0) / 10
print(f"host numpy loader + collate (1 core): {t_host * 1e3:.1f} ms per {B} scenes = {B / t_host:.0f} scenes/s")
print(f"GPU path incl. host concatenation and upload of pageable arrays: {t_gpu * 1e3:.1f} ms per {B} scenes = {B / t_gpu:.0f} scenes/s")
