"""Preprocessed-scene datasets: the reference's on-disk format and its per-sample pipeline.

Restates the data side of lib/dataset/nuscenes_dataloader.py (the SemanticKITTI / A2D2 / VirtualKITTI loaders share the
format and the per-sample code, semantic_kitti.py / a2d2.py / virtual_kitti_dataloader.py):

  * ``<split>.pkl`` files written by the preprocessing scripts (lib/dataset/preprocess_nuscenes_lidarseg.py:229-239,
    preprocces_virtual_kitti.py:81-87): a pickled list of dicts with ``points [n,3]``, ``pts_cam_coord [n,3]``,
    ``points_img [n,2]`` (row, col), ``seg_labels [n] uint8``, ``camera_path`` (relative), ``calib['cam_intrinsic']``,
    plus bookkeeping strings (``lidar_path``, ``sample_token``, ``scene_name``);
  * camera images decoded and resized with PIL exactly like the loader (``Image.open`` / ``resize(.., BILINEAR)`` /
    ``np.array(.., float32) / 255``, nuscenes_dataloader.py:256-287);
  * optional class merging (``label_mapping``, :163-171) and pseudo-label files (``pselab_paths``: a pickled list of dicts
    with ``probs_2d / pseudo_label_2d / probs_3d / pseudo_label_3d / probs_ensemble / pseudo_label_ensemble``), refined at
    load time by :func:`refine_pseudo_labels` (:96-162, lib/utils/refine_pseudo_labels.py:4-21) and filtered per sample with
    the voxel range mask (:336-349).

Two consumers: ``sample(i)`` = the host path (mm2d3d_amd/projection.make_sample, one scene, numpy - what a DataLoader
worker of the reference produces) and ``gpu_batch(indices)`` = decoded scenes handed to the GPU preparation chain
(mm2d3d_amd/dataprep.prepare_batch), which returns the collated batch dict on the device.
"""
from __future__ import annotations

import os
import pickle

import numpy as np

from . import projection


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


class PreprocessedScenes:
    """The reference's ``NuScenesLidarSegSCN`` (and its three siblings) over ``<split>.pkl`` files."""

    def __init__(self, split, preprocess_dir, image_dir="", pselab_paths=None, label_mapping=None, scale=20, full_scale=4096,
                 resize=(400, 225), image_normalizer=None, noisy_rot=0.0, flip_x=0.0, rot=0.0, transl=False, fliplr=0.0,
                 output_orig=False, camera_coords=False, use_rgb=False, short_run=False, reduce_factor=1):
        self.split = [split] if isinstance(split, str) else list(split)
        self.image_dir = image_dir
        self.data = []
        for s in self.split:
            with open(os.path.join(preprocess_dir, s + ".pkl"), "rb") as f:
                self.data.extend(pickle.load(f))
        if "train" in self.split[0] and short_run:  # :85-95: every reduce_factor-th scene in camera-path order
            order = sorted(range(len(self.data)), key=lambda i: self.data[i]["camera_path"])
            self.data = [self.data[i] for j, i in enumerate(order) if j % reduce_factor == 0]
        self.label_mapping = None if label_mapping is None else np.asarray(label_mapping, dtype=np.int64)
        self.kw = dict(scale=scale, full_scale=full_scale, camera_coords=camera_coords, noisy_rot=noisy_rot, flip_x=flip_x, rot=rot,
                       transl=transl, fliplr=fliplr, image_normalizer=image_normalizer, use_rgb=use_rgb, output_orig=output_orig)
        self.resize = tuple(resize) if resize else None
        self.pselab_data = None
        if pselab_paths:
            self.pselab_data = self._load_pseudo_labels(pselab_paths)

    # ------------------------------------------------------------------ pseudo labels (:96-162)
    def _load_pseudo_labels(self, path):
        data = list(np.load(path, allow_pickle=True))
        if len(data) != len(self.data):
            raise AssertionError("pseudo-label file and dataset have different lengths")
        for d, s in zip(data, self.data):
            if len(d["pseudo_label_2d"]) != len(s["seg_labels"]):
                raise AssertionError("pseudo labels and points of a scene have different lengths")
        data = [dict(d) for d in data]

        def refined(prob_key, lab_key):
            probs = np.concatenate([d[prob_key] for d in data])
            labs = np.concatenate([d[lab_key] for d in data]).astype(np.int64)
            return refine_pseudo_labels(probs, labs)

        lab2d = refined("probs_2d", "pseudo_label_2d")
        lab3d = refined("probs_3d", "pseudo_label_3d") if data[0]["probs_3d"] is not None else None
        labens = refined("probs_ensemble", "pseudo_label_ensemble")
        left = 0
        for d in data:  # undo the concatenation
            right = left + len(d["probs_2d"])
            d["pseudo_label_2d"] = lab2d[left:right]
            d["pseudo_label_3d"] = lab3d[left:right] if lab3d is not None else None
            d["pseudo_label_ensemble"] = labens[left:right]
            left = right
        return data

    def __len__(self):
        return len(self.data)

    # ------------------------------------------------------------------ decoding (:236-287)
    def decode(self, index):
        """Arrays of one scene before any augmentation: image float32 [H,W,3] in 0..1 (resized), scaled ``points_img``."""
        from PIL import Image

        d = self.data[index]
        seg_label = d["seg_labels"].astype(np.int64)
        if self.label_mapping is not None:
            seg_label = self.label_mapping[seg_label]
        image = Image.open(os.path.join(self.image_dir, d["camera_path"]))
        orig_size = image.size  # (W, H)
        if self.resize and image.size != self.resize:
            if not image.size[0] > self.resize[0]:
                raise AssertionError("resize must not enlarge the image")
            image = image.resize(self.resize, Image.BILINEAR)
        return dict(points=d["points"], pts_cam_coord=d["pts_cam_coord"], points_img=d["points_img"], seg_label=seg_label,
                    intrinsics=d["calib"]["cam_intrinsic"], image=np.array(image, dtype=np.float32) / 255.0, orig_size=orig_size)

    def sample(self, index):
        """The reference's ``__getitem__`` (host numpy path)."""
        a = self.decode(index)
        out = projection.make_sample(a["points"], a["pts_cam_coord"], a["points_img"], a["seg_label"], a["intrinsics"], a["image"],
                                     orig_size_wh=a["orig_size"], resize_wh=self.resize, **self.kw)
        if self.pselab_data is not None:  # :336-349: pseudo labels follow the voxel range mask
            idxs = out["orig_points_idx"] if "orig_points_idx" in out else out["_idxs"]
            p = self.pselab_data[index]
            out["pseudo_label_2d"] = p["pseudo_label_2d"][idxs]
            out["pseudo_label_3d"] = None if p["pseudo_label_3d"] is None else p["pseudo_label_3d"][idxs]
            out["pseudo_label_ensemble"] = p["pseudo_label_ensemble"][idxs]
        out.pop("_idxs", None)
        return out

    __getitem__ = sample

    def gpu_batch(self, indices, device="cuda", want_seg2d=False):
        """Decoded scenes -> mm2d3d_amd.dataprep.prepare_batch: the collated batch dict on the device.  The image
        normalisation is applied here on the host arrays (it commutes with the flip the GPU path applies)."""
        from . import dataprep

        kw = self.kw
        scenes = []
        for i in indices:
            a = self.decode(i)
            pimg = a["points_img"]
            if self.resize and tuple(a["orig_size"]) != tuple(self.resize):
                pimg = projection.scale_image_points(pimg, a["orig_size"], self.resize)
            img = a["image"]
            if kw["image_normalizer"]:
                mean, std = (np.asarray(v, dtype=np.float32) for v in kw["image_normalizer"])
                img = (img - mean) / std
            pts = a["pts_cam_coord"] if kw["camera_coords"] else a["points"]
            scenes.append(dict(points=np.ascontiguousarray(pts, dtype=np.float32), points_img=pimg, depth=a["pts_cam_coord"][:, 2],
                               seg_label=a["seg_label"], img=np.ascontiguousarray(np.moveaxis(img, -1, 0))))
        aug = dict(noisy_rot=kw["noisy_rot"], flip_x=kw["flip_x"], transl=kw["transl"],
                   rot_z=kw["rot"] if not kw["camera_coords"] else 0, rot_y=kw["rot"] if kw["camera_coords"] else 0)
        batch = dataprep.prepare_batch(scenes, kw["scale"], kw["full_scale"], aug, kw["fliplr"], want_seg2d, device)
        if self.pselab_data is not None:
            import torch

            keep = batch["keep"].cpu().numpy()
            offs = np.concatenate([[0], np.cumsum([len(s["points"]) for s in scenes])])
            cat = lambda key: np.concatenate([np.asarray(self.pselab_data[i][key]) for i in indices])[keep]
            batch["pseudo_label_2d"] = torch.from_numpy(cat("pseudo_label_2d")).to(device)
            batch["pseudo_label_ensemble"] = torch.from_numpy(cat("pseudo_label_ensemble")).to(device)
            if self.pselab_data[indices[0]]["pseudo_label_3d"] is not None:
                batch["pseudo_label_3d"] = torch.from_numpy(cat("pseudo_label_3d")).to(device)
            else:
                batch["pseudo_label_3d"] = []
            del offs
        return batch
