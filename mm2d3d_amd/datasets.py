"""Preprocessed-scene datasets and collation: the reference's four loaders.

Drop-in classes with the reference's names, constructor keywords and output dicts:

    NuScenesLidarSegSCN   lib/dataset/nuscenes_dataloader.py:172-369     resize, jitter, fliplr
    SemanticKITTISCN      lib/dataset/semantic_kitti.py:207-492           bottom_crop / rand_crop (+ crop_size), jitter, fliplr
    A2D2SCN               lib/dataset/a2d2.py:193-408                     rand_crop = (prob, hmin, hmax, wmin, wmax), resize
    VirtualKITTISCN       lib/dataset/virtual_kitti_dataloader.py:80-364  downsample, projection with the fixed camera matrix,
                                                                          random_weather, crops
    collate_scn_base      lib/dataset/__init__.py:27-123                  the batch dict

All four share one per-sample pipeline here (the reference repeats it four times).  A sample passes through

    front end (per dataset)   load the pickled scene, class merging, [downsample, project, weather], open the image,
                              [crop], [resize]: everything that decides WHICH points and WHICH image window are used
    finish (shared)           rasterise (pixel indices, last-write-wins depth / 2D label maps) -> colour jitter -> float
                              image -> fliplr -> normalise -> 3D augmentation + voxelisation -> range mask -> features

with every numpy RNG draw in the reference's order, so a seeded run picks the same crops / flips / rotations
(tests/test_loader_golden.py compares with fixtures produced by the reference's own classes, tests/golden/loader_*.npz).
``gpu_batch(indices)`` runs the front end on the host and hands the per-point work of the whole batch to the HIP chain
(mm2d3d_amd/dataprep.prepare_batch -> csrc/dataprep.hip); it returns the same batch dict with device tensors.

File formats (written by the reference's offline preprocessing, lib/dataset/preprocess_nuscenes_lidarseg.py:229-239,
preprocces_virtual_kitti.py:81-87): ``<split>.pkl`` = pickled list of dicts with ``points [n,3]``, ``pts_cam_coord [n,3]``,
``points_img [n,2]`` (row, col), ``seg_labels [n]``, ``camera_path`` and either ``calib['cam_intrinsic']`` (NuScenes) or
``intrinsics`` (SemanticKITTI); VirtualKITTI scenes carry ``points``, ``seg_labels``, ``scene_id``, ``frame_id`` only.
Pseudo-label files: a pickled object array of dicts ``probs_2d / pseudo_label_2d / probs_3d / pseudo_label_3d /
probs_ensemble / pseudo_label_ensemble``, refined at load time (lib/utils/refine_pseudo_labels.py:4-21).
"""
from __future__ import annotations

import json
import os
import pickle
from functools import partial

import numpy as np
import torch

from . import label_maps, projection
from .color_jitter import ColorJitter
from .voxelize import augment_and_scale_3d, voxelize_points


def refine_pseudo_labels(probs, pseudo_label, ignore_label=-100):
    """Per class, the less confident half of the pseudo labels (below min(median, 0.9)) becomes ``ignore_label``
    (lib/utils/refine_pseudo_labels.py:4-21; torch.median = the LOWER median for even counts)."""
    probs = np.asarray(probs)
    out = np.array(pseudo_label, copy=True)
    for cls in np.unique(out):
        idx = np.nonzero(out == cls)[0]
        p = probs[idx]
        med = np.sort(p)[(len(p) - 1) // 2]  # torch.median returns the lower of the two middle values
        thresh = min(med, 0.9)
        out[idx[p < thresh]] = ignore_label
    return out


class _Work:
    """One scene on its way through the pipeline: per-point arrays stay row-aligned, ``keep`` records which of the
    scene's original points survived the crop (the pseudo labels are filtered with it, semantic_kitti.py:462-475)."""

    __slots__ = ("points", "cam", "pimg", "label", "intr", "image", "keep")

    def __init__(self, points, cam, pimg, label, intr, image):
        self.points, self.cam, self.pimg, self.label, self.intr, self.image = points, cam, pimg, label, intr, image
        self.keep = np.ones(len(points), dtype=bool)

    def select(self, mask):
        self.points, self.cam, self.pimg = self.points[mask], self.cam[mask], self.pimg[mask]
        if self.label is not None:
            self.label = self.label[mask]
        self.keep = mask


def _inside(pimg, top, left, bottom, right):
    return (pimg[:, 0] >= top) & (pimg[:, 0] < bottom) & (pimg[:, 1] >= left) & (pimg[:, 1] < right)


def _random_window(dims, W, H):
    """A window of random relative size ``dims = (hmin, hmax, wmin, wmax)`` at a random place: rand(2), rand(), rand() in
    this order (semantic_kitti.py:335-349, a2d2.py:268-276)."""
    ch, cw = dims[0::2] + np.random.rand(2) * (dims[1::2] - dims[0::2])
    top = np.random.rand() * (1 - ch) * H
    left = np.random.rand() * (1 - cw) * W
    bottom, right = top + ch * H, left + cw * W
    return int(top), int(left), int(bottom), int(right)


def _find_window(w, draw, tries=10):
    """Up to ten draws until more than 100 points fall into the window (semantic_kitti.py:324-360)."""
    for _ in range(tries):
        top, left, bottom, right = draw()
        mask = _inside(w.pimg, top, left, bottom, right)
        if np.sum(mask) > 100:
            return (top, left, bottom, right), mask
    return None, None


def _apply_window(w, window, mask, shift_intrinsics):
    top, left, bottom, right = window
    if shift_intrinsics:  # semantic_kitti.py:363-365 (the reference subtracts `top` from the x centre and `left` from y)
        w.intr[0, 2] -= top
        w.intr[1, 2] -= left
    w.image = w.image.crop((left, top, right, bottom))
    w.select(mask)
    w.pimg[:, 0] -= top
    w.pimg[:, 1] -= left


def _resize(w, size_wh):
    """Scale the (floored) pixel coordinates and resize the image with PIL's BILINEAR filter."""
    from PIL import Image

    ow, oh = w.image.size
    w.pimg[:, 0] = float(size_wh[1]) / oh * np.floor(w.pimg[:, 0])
    w.pimg[:, 1] = float(size_wh[0]) / ow * np.floor(w.pimg[:, 1])
    w.image = w.image.resize(tuple(size_wh), Image.BILINEAR)


class _Scenes:
    """Shared by the four datasets: pkl loading, short runs, pseudo labels and the second half of the sample pipeline."""

    has_pselab = True

    def _load_splits(self, split, directory, short_run=False, reduce_factor=1):
        self.split = [split] if isinstance(split, str) else list(split)
        self.data = []
        for s in self.split:
            with open(os.path.join(directory, s + ".pkl"), "rb") as f:
                self.data.extend(pickle.load(f))
        if "train" in self.split[0] and short_run:  # every reduce_factor-th scene in camera-path order
            order = sorted(range(len(self.data)), key=lambda i: self.data[i]["camera_path"])
            self.data = [self.data[i] for j, i in enumerate(order) if j % reduce_factor == 0]

    def _load_pseudo_labels(self, path, length_key):
        """nuscenes_dataloader.py:96-162 / semantic_kitti.py:143-205: refine over the WHOLE dataset, then cut back per scene."""
        self.pselab_data = None
        if not path:
            return
        data = [dict(d) for d in np.load(path, allow_pickle=True)]
        if len(data) != len(self.data):
            raise AssertionError("pseudo-label file and dataset have different lengths")
        for d, s in zip(data, self.data):
            if len(d["pseudo_label_2d"]) != len(s[length_key]):
                raise AssertionError("pseudo labels and points of a scene have different lengths")

        def refined(prob_key, lab_key):
            probs = np.concatenate([d[prob_key] for d in data])
            labs = np.concatenate([d[lab_key] for d in data]).astype(int)
            return refine_pseudo_labels(probs, labs)

        lab2d = refined("probs_2d", "pseudo_label_2d")
        lab3d = refined("probs_3d", "pseudo_label_3d") if data[0]["probs_3d"] is not None else None
        labens = refined("probs_ensemble", "pseudo_label_ensemble")
        left = 0
        for d in data:
            right = left + len(d["probs_2d"])
            d["pseudo_label_2d"] = lab2d[left:right]
            d["pseudo_label_3d"] = lab3d[left:right] if lab3d is not None else None
            d["pseudo_label_ensemble"] = labens[left:right]
            left = right
        self.pselab_data = data

    def _common(self, scale, full_scale, noisy_rot, flip_x, rot, transl, fliplr, color_jitter, image_normalizer, camera_coords, use_rgb,
                output_orig=False):
        self.scale, self.full_scale = scale, full_scale
        self.noisy_rot, self.flip_x, self.rot, self.transl = noisy_rot, flip_x, rot, transl
        self.fliplr = fliplr
        self.color_jitter = ColorJitter(*color_jitter) if color_jitter else None
        self.image_normalizer = image_normalizer
        self.camera_coords, self.use_rgb, self.output_orig = camera_coords, use_rgb, output_orig
        self.pselab_data = getattr(self, "pselab_data", None)

    def __len__(self):
        return len(self.data)

    # ------------------------------------------------------------------ second half, host (numpy) form
    def _augmentation(self):
        return dict(noisy_rot=self.noisy_rot, flip_x=self.flip_x, transl=self.transl,
                    rot_z=self.rot if not self.camera_coords else 0, rot_y=self.rot if self.camera_coords else 0)

    def _float_image(self, image):
        if self.color_jitter is not None:
            image = self.color_jitter(image)
        return np.array(image, dtype=np.float32) / 255.0

    def _normalise(self, image):
        if self.image_normalizer:
            mean, std = (np.asarray(v, dtype=np.float32) for v in self.image_normalizer)
            image = (image - mean) / std
        return image

    def _finish(self, index, w):
        W, H = w.image.size
        img_indices, depth, seg2d = projection.rasterise(w.pimg, w.cam[:, 2], w.label, H, W)
        image = self._float_image(w.image)
        intr = w.intr
        if np.random.rand() < self.fliplr:
            image, img_indices, depth, seg2d, intr = projection.flip_lr(image, img_indices, depth, seg2d, intr)
        image = self._normalise(image)
        out = {"img": np.moveaxis(image, -1, 0), "depth": depth[None].astype(np.float32)}
        coords, min_value, offset, rot_matrix = augment_and_scale_3d(w.points, self.scale, self.full_scale, **self._augmentation())
        coords, idxs = voxelize_points(coords, self.full_scale)
        out.update(coords=coords, points=w.points[idxs], img_indices=img_indices[idxs], intrinsics=intr, seg_labels_2d=seg2d,
                   min_value=min_value, offset=offset, rot_matrix=rot_matrix)
        if w.label is not None:
            out["seg_label"] = w.label[idxs]
        if self.has_pselab and self.pselab_data is not None:
            p = self.pselab_data[index]
            for k in ("pseudo_label_2d", "pseudo_label_3d", "pseudo_label_ensemble"):
                out[k] = None if p[k] is None else p[k][w.keep][idxs]
        if self.output_orig:
            out.update(orig_seg_label=w.label, orig_points_idx=idxs)
        # the reference sizes the constant feature by the MASK's length, not by the kept rows (nuscenes_dataloader.py:365-368)
        out["feats"] = projection.point_feats(out["img"], out["img_indices"]) if self.use_rgb else np.ones([len(idxs), 1], np.float32)
        return out

    def __getitem__(self, index):
        return self._finish(index, self._front(index))

    sample = __getitem__

    # ------------------------------------------------------------------ second half on the GPU, whole batch
    def gpu_batch(self, indices, device="cuda", want_seg2d=False):
        """``collate_scn_base([self[i] for i in indices])`` with the per-point work on the GPU (csrc/dataprep.hip).  Scene by
        scene the host runs the front end, the colour jitter / float conversion / normalisation of the image and draws the
        flip and the 3D augmentation in the reference's order; pixel indices, depth / label maps, flip remap, rotation,
        voxelisation, range mask, point features and the concatenation run as kernels over the whole batch."""
        from . import dataprep

        scenes, intrinsics, works = [], [], []
        for i in indices:
            w = self._front(i)
            image = self._float_image(w.image)
            flip = bool(np.random.rand() < self.fliplr)
            rot, u = dataprep.augmentation_draws(**self._augmentation())
            image = self._normalise(image)  # commutes with the flip the GPU applies
            intr = w.intr
            if flip:
                intr = intr.copy()
                intr[0, 2] = image.shape[1] - intr[0, 2]
                intr[1, 2] = image.shape[0] - intr[0, 1]
            if w.label is None:
                raise ValueError("gpu_batch needs labelled scenes (the 2D label map and seg_label are part of the batch)")
            if w.points.dtype != np.float32:
                raise NotImplementedError("gpu_batch voxelises float32 points; this configuration (VirtualKITTI with camera_coords) "
                                          "produces float64 points in the reference: use the host path")
            # the kernel truncates float32 pixel coordinates; truncating here first keeps float64 inputs (VirtualKITTI) exact
            scenes.append(dict(points=np.ascontiguousarray(w.points), points_img=np.trunc(w.pimg), depth=w.cam[:, 2], seg_label=w.label,
                               img=np.ascontiguousarray(np.moveaxis(image, -1, 0)), draws=(flip, rot, u)))
            intrinsics.append(intr)
            works.append(w)
        batch = dataprep.prepare_batch(scenes, self.scale, self.full_scale, None, 0.0, want_seg2d, device, use_rgb=self.use_rgb)
        batch["intrinsics"] = torch.from_numpy(np.stack(intrinsics))
        batch["points"] = torch.cat(batch["points"], 0) if batch["points"] else batch["points"]
        batch["coords"] = batch["x"][0][:, :3]
        batch["rotation_matrices"] = torch.from_numpy(np.stack(batch["rotation_matrices"]))
        if "seg_labels_2d" in batch:
            batch["seg_labels_2d"] = batch["seg_labels_2d"].float()
        if self.output_orig:  # labels before the range mask and the mask itself, per scene
            kept = batch["keep"].cpu().numpy()
            off = np.concatenate([[0], np.cumsum([len(w.points) for w in works])])
            masks = []
            for b, w in enumerate(works):
                m = np.zeros(len(w.points), dtype=bool)
                m[kept[(kept >= off[b]) & (kept < off[b + 1])] - off[b]] = True
                masks.append(m)
            batch["orig_seg_label"] = [w.label for w in works]
            batch["orig_points_idx"] = masks
        if self.has_pselab and self.pselab_data is not None:
            keep = batch["keep"].cpu().numpy()  # rows of the concatenated post-crop scenes that passed the range mask

            def cat(key):
                return np.concatenate([np.asarray(self.pselab_data[i][key])[w.keep] for i, w in zip(indices, works)])[keep]

            batch["pseudo_label_2d"] = torch.from_numpy(cat("pseudo_label_2d")).to(device)
            batch["pseudo_label_ensemble"] = torch.from_numpy(cat("pseudo_label_ensemble")).to(device)
            has3d = self.pselab_data[indices[0]]["pseudo_label_3d"] is not None
            batch["pseudo_label_3d"] = torch.from_numpy(cat("pseudo_label_3d")).to(device) if has3d else []
        return batch


class NuScenesLidarSegSCN(_Scenes):
    """lib/dataset/nuscenes_dataloader.py:172-369."""

    def __init__(self, split, preprocess_dir, nuscenes_dir="", pselab_paths=None, merge_classes=False, scale=20, full_scale=4096,
                 resize=(400, 225), image_normalizer=None, noisy_rot=0.0, flip_x=0.0, rot=0.0, transl=False, fliplr=0.0,
                 color_jitter=None, output_orig=False, short_run=False, reduce_factor=1, camera_coords=False, use_rgb=False,
                 label_mapping=None):
        self._load_splits(split, preprocess_dir, short_run, reduce_factor)
        self._load_pseudo_labels(pselab_paths, "seg_labels")
        self.class_names = list(label_maps.NUSCENES_RAW)
        self.label_mapping = None
        if merge_classes:
            self.label_mapping, self.class_names = label_maps.merged(label_maps.NUSCENES_MERGE, len(label_maps.NUSCENES_RAW))
        if label_mapping is not None:  # an explicit table (not in the reference): used by the miniature datasets of the tests
            self.label_mapping = np.asarray(label_mapping, dtype=np.int64)
        self.nuscenes_dir = nuscenes_dir
        self.resize = tuple(resize) if resize else None
        self._common(scale, full_scale, noisy_rot, flip_x, rot, transl, fliplr, color_jitter, image_normalizer, camera_coords, use_rgb,
                     output_orig)

    def _front(self, index):
        from PIL import Image

        d = self.data[index]
        cam = d["pts_cam_coord"].copy()
        points = cam.copy() if self.camera_coords else d["points"].copy()
        label = d["seg_labels"].astype(np.int64)
        if self.label_mapping is not None:
            label = self.label_mapping[label]
        w = _Work(points, cam, d["points_img"].copy(), label, d["calib"]["cam_intrinsic"].copy(),
                  Image.open(os.path.join(self.nuscenes_dir, d["camera_path"])))
        if self.resize and tuple(w.image.size) != self.resize:
            if not w.image.size[0] > self.resize[0]:
                raise AssertionError("resize must not enlarge the image")  # :260
            _resize(w, self.resize)
            w.intr[:2] /= 4  # :271, hard-coded in the reference
        return w


class PreprocessedScenes(NuScenesLidarSegSCN):
    """Round-2 name of the NuScenes-format dataset: ``split`` may be a string, the image root is ``image_dir``."""

    def __init__(self, split, preprocess_dir, image_dir="", **kw):
        super().__init__(split, preprocess_dir, nuscenes_dir=image_dir, **kw)


class SemanticKITTISCN(_Scenes):
    """lib/dataset/semantic_kitti.py:207-492.  ``resize`` and ``downsample`` are accepted and ignored, as in the reference."""

    def __init__(self, split, preprocess_dir, semantic_kitti_dir="", pselab_paths=None, merge_classes_style=None, merge_classes=None,
                 scale=20, full_scale=4096, image_normalizer=None, noisy_rot=0.0, flip_x=0.0, rot=0.0, transl=False, crop_size=tuple(),
                 bottom_crop=False, rand_crop=tuple(), fliplr=0.0, color_jitter=None, output_orig=False, resize=tuple(),
                 downsample=(-1,), short_run=False, reduce_factor=1, camera_coords=False, use_rgb=False):
        self._load_splits(split, preprocess_dir, short_run, reduce_factor)
        self._load_pseudo_labels(pselab_paths, "points")
        if not merge_classes_style:
            raise NotImplementedError("The merge classes style needs to be provided, e.g. A2D2.")
        self.label_mapping, self.class_names = label_maps.merged(label_maps.SEMANTIC_KITTI_MERGE[merge_classes_style],
                                                                 label_maps.SEMANTIC_KITTI_TABLE)
        self.semantic_kitti_dir = semantic_kitti_dir
        self.crop_size, self.bottom_crop, self.rand_crop = _crop_config(crop_size, bottom_crop, rand_crop)
        self._common(scale, full_scale, noisy_rot, flip_x, rot, transl, fliplr, color_jitter, image_normalizer, camera_coords, use_rgb,
                     output_orig)

    def _front(self, index):
        from PIL import Image

        d = self.data[index]
        cam = d["pts_cam_coord"].copy()
        points = cam.copy() if self.camera_coords else d["points"].copy()
        label = d["seg_labels"]
        if label is not None:
            label = self.label_mapping[label.astype(np.int64)]
        w = _Work(points, cam, d["points_img"].copy(), label, d["intrinsics"].copy(),
                  Image.open(os.path.join(self.semantic_kitti_dir, d["camera_path"])))
        _crop(w, self.crop_size, self.bottom_crop, self.rand_crop, d.get("camera_path"))
        return w


def _crop_config(crop_size, bottom_crop, rand_crop):
    crop_size = tuple(crop_size) if crop_size else tuple()
    if crop_size:
        if bottom_crop == bool(rand_crop):
            raise AssertionError("Exactly one crop method needs to be active if crop size is provided!")
    elif bottom_crop or rand_crop:
        raise AssertionError("No crop size, but crop method is provided is provided!")
    rand_crop = np.array(rand_crop)
    if len(rand_crop) not in (0, 4):
        raise AssertionError("rand_crop = (min_crop_height, max_crop_height, min_crop_width, max_crop_width)")
    return crop_size, bottom_crop, rand_crop


def _crop(w, crop_size, bottom_crop, rand_crop, name):
    """semantic_kitti.py:321-392 = virtual_kitti_dataloader.py:215-286: a bottom crop of ``crop_size`` at a random column, or
    a random window resized to ``crop_size``; the principal point moves with the window."""
    if not crop_size:
        return
    W, H = w.image.size
    if bottom_crop:
        def draw():
            left = int(np.random.rand() * (W + 1 - crop_size[0]))
            return H - crop_size[1], left, H, left + crop_size[0]
    else:
        draw = partial(_random_window, rand_crop, W, H)
    window, mask = _find_window(w, draw)
    if window is None:
        print("No valid crop found for image", name)
        return
    _apply_window(w, window, mask, shift_intrinsics=True)
    if len(rand_crop) > 0:
        _resize(w, crop_size)


class A2D2SCN(_Scenes):
    """lib/dataset/a2d2.py:193-408.  ``crop_size`` / ``bottom_crop`` are accepted and ignored, as in the reference (the shipped
    yaml passes them, datasets/a2d2_semantic_kitti.yaml:35-37)."""

    has_pselab = False
    INTRINSICS = ((1687.3369140625, 0.0, 965.43414055823814), (0.0, 1783.428466796875, 684.4193604186803), (0.0, 0.0, 1.0))  # a2d2.py:257-263

    def __init__(self, split, preprocess_dir, merge_classes=True, merge_classes_style="A2D2", scale=20, full_scale=4096, resize=(480, 302),
                 image_normalizer=None, noisy_rot=0.0, flip_x=0.0, rot=0.0, transl=False, rand_crop=tuple(), crop_size=tuple(),
                 bottom_crop=False, fliplr=0.0, color_jitter=None, short_run=False, reduce_factor=1, camera_coords=False, use_rgb=False):
        self.preprocess_dir = preprocess_dir
        with open(os.path.join(preprocess_dir, "cams_lidars.json")) as f:
            self.config = json.load(f)
        self._load_splits(split, os.path.join(preprocess_dir, "preprocess"), short_run, reduce_factor)
        with open(os.path.join(preprocess_dir, "class_list.json")) as f:
            self.class_names = list(json.load(f).values())  # hex colour -> class name, in label-index order
        self.label_mapping = None
        if merge_classes:
            table = {cat: [n for n in self.class_names if label_maps.a2d2_match(n, pats)] for cat, pats in label_maps.A2D2_MERGE.items()}
            lookup = {n: i for i, n in enumerate(self.class_names)}
            self.label_mapping, self.class_names = label_maps.merged({c: [lookup[n] for n in ns] for c, ns in table.items()},
                                                                     len(self.class_names) + 1)
        self.resize = tuple(resize) if resize else None
        self.crop_prob = rand_crop[0] if rand_crop else 0.0
        self.crop_dims = np.array(rand_crop[1:]) if rand_crop else None
        self._common(scale, full_scale, noisy_rot, flip_x, rot, transl, fliplr, color_jitter, image_normalizer, camera_coords, use_rgb)

    def _front(self, index):
        from PIL import Image

        d = self.data[index]
        cam = d["pts_cam_coord"].copy()
        points = cam.copy() if self.camera_coords else d["points"].copy()
        label = d["seg_labels"].astype(np.int64)
        if self.label_mapping is not None:
            label = self.label_mapping[label]
        w = _Work(points, cam, d["points_img"].copy(), label, np.array(self.INTRINSICS),
                  Image.open(os.path.join(self.preprocess_dir, d["camera_path"])))
        if np.random.rand() < self.crop_prob:  # drawn even when cropping is off (a2d2.py:266)
            W, H = w.image.size
            window, mask = _find_window(w, partial(_random_window, self.crop_dims, W, H))
            if window is None:
                print("No valid crop found for image", d["camera_path"])
            else:
                _apply_window(w, window, mask, shift_intrinsics=False)
        if self.resize and tuple(w.image.size) != self.resize:
            if not w.image.size[0] > self.resize[0]:
                raise AssertionError("resize must not enlarge the image")
            _resize(w, self.resize)
            w.intr[:2] /= 4
        return w


class VirtualKITTISCN(_Scenes):
    """lib/dataset/virtual_kitti_dataloader.py:80-364."""

    has_pselab = False
    proj_matrix = np.array([[725, 0, 620.5], [0, 725, 187], [0, 0, 1]], dtype=np.float32)  # :39-41

    def __init__(self, split, preprocess_dir, virtual_kitti_dir="", merge_classes=False, merge_classes_style="VirtualKITTI", scale=20,
                 full_scale=4096, image_normalizer=None, noisy_rot=0.0, flip_x=0.0, rot=0.0, transl=False, downsample=(-1,),
                 crop_size=tuple(), bottom_crop=False, rand_crop=tuple(), fliplr=0.0, color_jitter=None,
                 random_weather=("clone", "fog", "morning", "overcast", "rain", "sunset"), short_run=False, reduce_factor=1,
                 camera_coords=False, use_rgb=False):
        self._load_splits(split, preprocess_dir)
        if not merge_classes:
            raise NotImplementedError
        self.label_mapping, self.class_names = label_maps.merged(label_maps.VIRTUAL_KITTI_MERGE, len(label_maps.VIRTUAL_KITTI_RAW))
        self.virtual_kitti_dir = virtual_kitti_dir
        self.downsample = downsample[0] if len(downsample) == 1 else tuple(downsample)
        self.crop_size, self.bottom_crop, self.rand_crop = _crop_config(crop_size, bottom_crop, rand_crop)
        self.random_weather = random_weather
        self._common(scale, full_scale, noisy_rot, flip_x, rot, transl, fliplr, color_jitter, image_normalizer, camera_coords, use_rgb)

    def _front(self, index):
        from PIL import Image

        d = self.data[index]
        points = d["points"].copy()
        label = d["seg_labels"].astype(np.int64)
        n_keep = self.downsample
        if isinstance(n_keep, tuple):
            n_keep = np.random.randint(low=n_keep[0], high=n_keep[1])
        if n_keep > 0:  # uniform subsample without replacement (:181-186)
            if not n_keep < len(points):
                raise AssertionError("downsample must be smaller than the scene")
            choice = np.random.choice(len(points), size=n_keep, replace=False)
            points, label = points[choice], label[choice]
        label[label == 99] = len(self.label_mapping) - 1
        label = self.label_mapping[label]
        # LiDAR frame (x front, y left, z up) -> camera frame (x right, y down, z front); float64 from here on (:195-205)
        cam = np.array([-1, -1, 1]) * points[:, [1, 2, 0]]
        if self.camera_coords:
            points = cam.copy()
        pimg = (self.proj_matrix @ cam.T).T
        pimg = np.fliplr(pimg[:, :2] / np.expand_dims(pimg[:, 2], axis=1))  # (u, v) -> (row, col)
        weather = "clone"
        if self.random_weather:
            weather = self.random_weather[np.random.randint(len(self.random_weather))]
        path = os.path.join(self.virtual_kitti_dir, "vkitti_1.3.1_rgb", d["scene_id"], weather, d["frame_id"] + ".png")
        w = _Work(points, cam, pimg, label, self.proj_matrix.copy(), Image.open(path))
        _crop(w, self.crop_size, self.bottom_crop, self.rand_crop, path)
        return w


# ---------------------------------------------------------------------------------------------------- collation (a2)
def collate_scn_base(input_dict_list, output_orig, output_image=True):
    """lib/dataset/__init__.py:27-123: the batch index becomes the LAST coordinate column, per-point arrays are
    concatenated in scene order, per-scene arrays stacked, ``img_indices`` stays a list of numpy arrays."""
    first = input_dict_list[0]
    pselab = "pseudo_label_2d" in first
    t = torch.from_numpy
    locs = []
    for b, s in enumerate(input_dict_list):
        c = t(s["coords"])
        locs.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int64)], 1))
    rows = lambda key: [t(s[key]) for s in input_dict_list]
    out = {
        "x": [torch.cat(locs, 0), torch.cat(rows("feats"), 0)],
        "rotation_matrices": torch.stack(rows("rot_matrix")),
        "min_values": torch.stack(rows("min_value")),
        "offsets": torch.stack(rows("offset")),
        "points": torch.cat(rows("points"), 0),
        "coords": torch.cat(rows("coords"), 0),
        "intrinsics": torch.stack(rows("intrinsics")),
        "seg_labels_2d": torch.stack(rows("seg_labels_2d")).float(),
    }
    labels = [t(s["seg_label"]) for s in input_dict_list if "seg_label" in s]
    if labels:
        out["seg_label"] = torch.cat(labels, 0)
    if output_image:
        out["img"] = torch.stack(rows("img"))
        out["img_indices"] = [s["img_indices"] for s in input_dict_list]
        out["depth"] = torch.stack(rows("depth"))
    if output_orig:
        out["orig_seg_label"] = [s["orig_seg_label"] for s in input_dict_list]
        out["orig_points_idx"] = [s["orig_points_idx"] for s in input_dict_list]
    if pselab:
        out["pseudo_label_2d"] = torch.cat(rows("pseudo_label_2d"), 0)
        p3 = [t(s["pseudo_label_3d"]) for s in input_dict_list if s["pseudo_label_3d"] is not None]
        out["pseudo_label_3d"] = torch.cat(p3, 0) if p3 else p3
        out["pseudo_label_ensemble"] = torch.cat(rows("pseudo_label_ensemble"), 0)
    return out


def get_collate_scn(is_train):
    return partial(collate_scn_base, output_orig=not is_train)


def worker_init_fn(worker_id):
    np.random.seed(worker_id)  # lib/dataset/__init__.py:142-153


def load_datasets(name, cfg_source, cfg_target, ds_args=None, augmentations=None, short_run=False, reduce_factor=1, pselab_paths=None):
    """The dataset choice of ``load_datamodule`` (lib/dataset/__init__.py:156-296) without the Lightning wrapper: returns
    ``dict(train_source=, train_target=, val_target=, test=)``.  ``cfg_*`` are mappings with the keys of the DATASET_SOURCE /
    DATASET_TARGET blocks of datasets/*.yaml."""
    ds_args, aug = dict(ds_args or {}), dict(augmentations or {})
    short = dict(short_run=short_run, reduce_factor=reduce_factor)
    if name == "nuscenes":
        mk = lambda split, **kw: NuScenesLidarSegSCN(split=split, preprocess_dir=cfg_source["preprocess_dir"],
                                                     nuscenes_dir=cfg_source["nuscenes_dir"], **ds_args, **kw)
        return dict(
            train_source=mk(cfg_source["TRAIN"], output_orig=False, **short, **aug),
            train_target=NuScenesLidarSegSCN(split=cfg_target["TRAIN"], preprocess_dir=cfg_target["preprocess_dir"],
                                             nuscenes_dir=cfg_target["nuscenes_dir"], output_orig=False, pselab_paths=pselab_paths,
                                             **short, **ds_args, **aug),
            val_target=mk(cfg_target["VAL"], output_orig=True), test=mk(cfg_target["TEST"], output_orig=True))
    if name in ("ad2d_semantic_kitti", "vkitti_semantic_kitti"):
        if name == "ad2d_semantic_kitti":
            src = A2D2SCN(split=cfg_source["TRAIN"], preprocess_dir=cfg_source["preprocess_dir"], **short, **ds_args, **aug)
        else:
            src = VirtualKITTISCN(split=cfg_source["TRAIN"], preprocess_dir=cfg_source["preprocess_dir"],
                                  virtual_kitti_dir=cfg_source["virtual_kitti_dir"], **short, **ds_args, **aug)
        mk = lambda split, **kw: SemanticKITTISCN(split=split, preprocess_dir=cfg_target["preprocess_dir"],
                                                  semantic_kitti_dir=cfg_target["semantic_kitti_dir"], **ds_args, **kw)
        return dict(train_source=src, train_target=mk(cfg_target["TRAIN"], output_orig=False, pselab_paths=pselab_paths, **short, **aug),
                    val_target=mk(cfg_target["VAL"], output_orig=True), test=mk(cfg_target["TEST"], output_orig=True))
    raise ValueError(f"not found datamodule {name}")
