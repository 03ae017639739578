"""Dataset-side projection of a preprocessed scene into the model's inputs.

Restates lib/dataset/nuscenes_dataloader.py:236-369 (``NuScenesLidarSegSCN.__getitem__``; the SemanticKITTI / A2D2 /
VirtualKITTI loaders run the same code, semantic_kitti.py:393-402,484-487, a2d2.py:334-344,394-397): scaled pixel
indices, sparse depth map (last write wins, numpy fancy assignment), 2D label map, horizontal flip remap, image
normalisation, 3D augmentation + voxelisation (``voxelize.py``), in-range filtering of every per-point array and the
RGB point features ``img[:, r, c].T``.  It is host code (numpy) in the reference and stays host code here: it runs once
per sample in the loader.  Image decoding / resizing (PIL) and colour jitter (torchvision) are the caller's: the
function takes the already decoded float image.  RNG draws happen in the reference's order (fliplr ``rand()`` first,
then the draws of ``augment_and_scale_3d``), so a seeded run selects the same augmentations.

Pinned by fixtures generated from the reference's own dataset classes (tests/golden/make_golden_loaders.py imports
lib.dataset over stand-ins for the absent pytorch_lightning / omegaconf / torchvision.transforms; tests/test_loader_golden.py)
and by hand-computed cases (tests/test_projection.py).  The dataset classes of mm2d3d_amd/datasets.py call the pieces below.
"""
from __future__ import annotations

import numpy as np

from .voxelize import augment_and_scale_3d, voxelize_points


def scale_image_points(points_img, orig_size_wh, resize_wh):
    """:259-268: rows scale with resize_h / orig_h, columns with resize_w / orig_w, both applied to floor(coordinate)."""
    p = np.array(points_img, dtype=np.float64 if points_img.dtype == np.float64 else np.float32, copy=True)
    if tuple(orig_size_wh) == tuple(resize_wh):
        return p
    if not orig_size_wh[0] > resize_wh[0]:
        raise AssertionError("resize must not enlarge the image")  # the reference's assert :261
    p[:, 0] = float(resize_wh[1]) / orig_size_wh[1] * np.floor(p[:, 0])
    p[:, 1] = float(resize_wh[0]) / orig_size_wh[0] * np.floor(p[:, 1])
    return p


def rasterise(points_img, depth_values, seg_label, H, W):
    """:274-283: int64 pixel indices (truncation), depth[r, c] = z and seg_labels_2d[r, c] = label, last write wins."""
    img_indices = points_img.astype(np.int64)
    if not (np.all(img_indices[:, 0] >= 0) and np.all(img_indices[:, 1] >= 0) and np.all(img_indices[:, 0] < H)
            and np.all(img_indices[:, 1] < W)):
        raise AssertionError("projected point outside the image")
    depth = np.zeros((H, W))
    depth[img_indices[:, 0], img_indices[:, 1]] = depth_values
    seg2d = np.ones((H, W)) * (-100)
    if seg_label is not None:
        seg2d[img_indices[:, 0], img_indices[:, 1]] = seg_label
    return img_indices, depth, seg2d


def flip_lr(image_hwc, img_indices, depth, seg2d, intrinsics):
    """:291-297 (including the reference's intrinsics[1, 2] update, which reads intrinsics[0, 1])."""
    image_hwc = np.ascontiguousarray(np.fliplr(image_hwc))
    img_indices = img_indices.copy()
    img_indices[:, 1] = image_hwc.shape[1] - 1 - img_indices[:, 1]
    depth = np.ascontiguousarray(np.fliplr(depth))
    intrinsics = intrinsics.copy()
    intrinsics[0, 2] = image_hwc.shape[1] - intrinsics[0, 2]
    intrinsics[1, 2] = image_hwc.shape[0] - intrinsics[0, 1]
    seg2d = np.ascontiguousarray(np.fliplr(seg2d))
    return image_hwc, img_indices, depth, seg2d, intrinsics


def point_feats(img_chw, img_indices):
    """:361-364: the image values under the points, [n, C]."""
    return img_chw[:, img_indices[:, 0], img_indices[:, 1]].T


def make_sample(points, pts_cam_coord, points_img, seg_label, intrinsics, image_hwc, *, orig_size_wh=None, resize_wh=None,
                scale=20, full_scale=4096, camera_coords=True, noisy_rot=0.0, flip_x=0.0, rot=0.0, transl=False, fliplr=0.0,
                image_normalizer=None, use_rgb=True, output_orig=False):
    """One loader sample (``out_dict`` of __getitem__) from decoded arrays.

    points / pts_cam_coord [n,3], points_img [n,2] (row, col) in the ORIGINAL image, seg_label [n] int, intrinsics [3,3],
    image_hwc float32 [H,W,3] in 0..1 ALREADY resized to ``resize_wh`` (colour jitter applied by the caller if any).
    """
    H, W = image_hwc.shape[0], image_hwc.shape[1]
    pts = (pts_cam_coord if camera_coords else points).copy()
    seg_label = seg_label.astype(np.int64)
    intr = intrinsics.copy()
    pimg = points_img.copy()
    if resize_wh is not None and orig_size_wh is not None and tuple(orig_size_wh) != tuple(resize_wh):
        pimg = scale_image_points(pimg, orig_size_wh, resize_wh)
        intr[:2] /= 4  # :271 (hard-coded in the reference)
    img_indices, depth, seg2d = rasterise(pimg, pts_cam_coord[:, 2], seg_label, H, W)
    image = image_hwc
    if np.random.rand() < fliplr:
        image, img_indices, depth, seg2d, intr = flip_lr(image, img_indices, depth, seg2d, intr)
    if image_normalizer:
        mean, std = (np.asarray(v, dtype=np.float32) for v in image_normalizer)
        image = (image - mean) / std
    out = {"img": np.moveaxis(image, -1, 0), "depth": depth[None].astype(np.float32)}
    coords, min_value, offset, rot_matrix = augment_and_scale_3d(
        pts, scale, full_scale, noisy_rot=noisy_rot, flip_x=flip_x, rot_z=rot if not camera_coords else 0,
        rot_y=rot if camera_coords else 0, transl=transl)
    coords, idxs = voxelize_points(coords, full_scale)
    out.update(coords=coords, points=pts[idxs], seg_label=seg_label[idxs], img_indices=img_indices[idxs], intrinsics=intr,
               seg_labels_2d=seg2d, min_value=min_value, offset=offset, rot_matrix=rot_matrix)
    out["_idxs"] = idxs  # the range mask (consumers that filter further per-point arrays; dropped by the dataset class)
    if output_orig:
        out.update(orig_seg_label=seg_label, orig_points_idx=idxs)
    out["feats"] = point_feats(out["img"], out["img_indices"]) if use_rgb else np.ones([len(idxs), 1], np.float32)
    return out
