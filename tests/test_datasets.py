"""Preprocessed-scene datasets (mm2d3d_amd/datasets.py): the reference's pkl format, PIL decoding / resizing, pseudo labels."""
import os
import pickle

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["even_odd", "small", "single_class"])
def test_refine_pseudo_labels_vs_reference_golden(name):
    from mm2d3d_amd.datasets import refine_pseudo_labels

    z = np.load(os.path.join(G, "pselab.npz"))
    out = refine_pseudo_labels(z[f"{name}/probs"], z[f"{name}/labels"])
    assert out.dtype == np.int64 and np.array_equal(out, z[f"{name}/refined"])


def _write_dataset(root, n_scenes=3, with_pselab=True):
    """A miniature dataset in the reference's on-disk format: <split>.pkl + JPEG camera images (+ a pseudo-label file)."""
    from PIL import Image

    from mm2d3d_amd.synthetic import lidar_sweep

    rng = np.random.default_rng(9)
    W0, H0 = 160, 90  # "original" camera size; the loader resizes to (80, 45)
    data, pselab = [], []
    os.makedirs(os.path.join(root, "cams"), exist_ok=True)
    for i in range(n_scenes):
        pts = lidar_sweep(60 + i, "nuscenes")[::50].copy()
        n = len(pts)
        img = (rng.random((H0, W0, 3)) * 255).astype(np.uint8)
        path = os.path.join("cams", f"img{i}.jpg")
        Image.fromarray(img).save(os.path.join(root, path), quality=92)
        data.append({
            "points": pts, "pts_cam_coord": (pts[:, [1, 2, 0]] * np.float32(1.0)).copy(),
            "points_img": np.stack([rng.uniform(0, H0 - 1e-2, n), rng.uniform(0, W0 - 1e-2, n)], 1).astype(np.float32),
            "seg_labels": rng.integers(0, 8, n).astype(np.uint8), "lidar_path": f"lidar{i}.bin", "camera_path": path,
            "sample_token": f"tok{i}", "scene_name": "scene-0001",
            "calib": {"cam_intrinsic": np.array([[1266.4, 0.0, 816.3], [0.0, 1266.4, 491.5], [0.0, 0.0, 1.0]])},
        })
        probs = rng.random((3, n)).astype(np.float32)
        pselab.append({"probs_2d": probs[0], "pseudo_label_2d": rng.integers(0, 4, n).astype(np.uint8),
                       "probs_3d": probs[1], "pseudo_label_3d": rng.integers(0, 4, n).astype(np.uint8),
                       "probs_ensemble": probs[2], "pseudo_label_ensemble": rng.integers(0, 4, n).astype(np.uint8)})
    with open(os.path.join(root, "train_day.pkl"), "wb") as f:
        pickle.dump(data, f)
    ps_path = None
    if with_pselab:
        ps_path = os.path.join(root, "pselab.npy")
        np.save(ps_path, np.array(pselab, dtype=object), allow_pickle=True)
    return data, pselab, ps_path


KW = dict(resize=(80, 45), image_normalizer=((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)), noisy_rot=0.1, flip_x=0.5, rot=6.2831,
          transl=True, fliplr=0.5, camera_coords=True, use_rgb=True, label_mapping=[0, 0, 1, 1, 2, 3, -100, 3])


def test_pkl_reader_sample_matches_the_reference_pipeline(tmp_path):
    """sample(i) = decode (PIL open + BILINEAR resize + /255) -> make_sample, with the class merging and the refined
    pseudo labels filtered by the voxel range mask (nuscenes_dataloader.py:236-369)."""
    from PIL import Image

    from mm2d3d_amd.datasets import PreprocessedScenes, refine_pseudo_labels
    from mm2d3d_amd.projection import make_sample

    data, pselab, ps_path = _write_dataset(str(tmp_path))
    ds = PreprocessedScenes("train_day", str(tmp_path), str(tmp_path), pselab_paths=ps_path, output_orig=True, **KW)
    assert len(ds) == 3
    np.random.seed(5)
    got = ds[1]
    d = data[1]
    image = Image.open(os.path.join(str(tmp_path), d["camera_path"]))
    assert image.size == (160, 90)
    image = np.array(image.resize((80, 45), Image.BILINEAR), dtype=np.float32) / 255.0
    lab = np.asarray(KW["label_mapping"])[d["seg_labels"].astype(np.int64)]
    np.random.seed(5)
    exp = make_sample(d["points"], d["pts_cam_coord"], d["points_img"], lab, d["calib"]["cam_intrinsic"], image, orig_size_wh=(160, 90),
                      resize_wh=(80, 45), scale=20, full_scale=4096, camera_coords=True, noisy_rot=0.1, flip_x=0.5, rot=6.2831,
                      transl=True, fliplr=0.5, image_normalizer=KW["image_normalizer"], use_rgb=True, output_orig=True)
    for k in ("coords", "points", "seg_label", "img", "img_indices", "depth", "feats", "intrinsics", "seg_labels_2d", "orig_points_idx"):
        assert np.array_equal(got[k], exp[k]), k
    assert got["img"].shape == (3, 45, 80) and got["depth"].shape == (1, 45, 80)
    # pseudo labels: refined over the WHOLE dataset (concatenated), then cut back per scene and masked
    all2d = refine_pseudo_labels(np.concatenate([p["probs_2d"] for p in pselab]),
                                 np.concatenate([p["pseudo_label_2d"] for p in pselab]).astype(np.int64))
    n0 = len(pselab[0]["probs_2d"])
    assert np.array_equal(got["pseudo_label_2d"], all2d[n0 : n0 + len(pselab[1]["probs_2d"])][exp["orig_points_idx"]])
    assert (got["pseudo_label_2d"] == -100).any() and got["pseudo_label_3d"] is not None


def test_collate_carries_pseudo_labels(tmp_path):
    from mm2d3d_amd.datasets import PreprocessedScenes
    from mm2d3d_amd.synthetic import collate

    _, _, ps_path = _write_dataset(str(tmp_path))
    ds = PreprocessedScenes("train_day", str(tmp_path), str(tmp_path), pselab_paths=ps_path, **KW)
    np.random.seed(2)
    b = collate([ds[0], ds[2]])
    n = b["x"][0].shape[0]
    assert b["pseudo_label_2d"].shape == (n,) and b["pseudo_label_ensemble"].shape == (n,) and b["pseudo_label_3d"].shape == (n,)


@pytest.mark.gpu
def test_gpu_batch_equals_host_samples(tmp_path):
    """The same scenes through decode -> mm2d3d_amd.dataprep (HIP) and through the host pipeline + collate."""
    from mm2d3d_amd.datasets import PreprocessedScenes
    from mm2d3d_amd.synthetic import collate

    _, _, ps_path = _write_dataset(str(tmp_path))
    ds = PreprocessedScenes("train_day", str(tmp_path), str(tmp_path), pselab_paths=ps_path, **KW)
    np.random.seed(31)
    ref = collate([ds[i] for i in (2, 0, 1)])
    np.random.seed(31)
    out = ds.gpu_batch([2, 0, 1])
    assert torch.equal(out["x"][0].cpu(), ref["x"][0]) and torch.equal(out["x"][1].cpu(), ref["x"][1])
    assert torch.equal(out["seg_label"].cpu(), ref["seg_label"])
    assert torch.equal(out["img"].cpu(), ref["img"]) and torch.equal(out["depth"].cpu(), ref["depth"])
    for a, b in zip(out["img_indices"], ref["img_indices"]):
        assert np.array_equal(a.cpu().numpy(), b)
    for k in ("pseudo_label_2d", "pseudo_label_3d", "pseudo_label_ensemble"):
        assert torch.equal(out[k].cpu(), ref[k]), k
