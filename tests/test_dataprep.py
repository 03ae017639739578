"""GPU-side sample preparation (csrc/dataprep.hip, mm2d3d_amd/dataprep.py) against the reference-generated golden vectors
(tests/golden/voxelize.npz) and against the host restatement of the loader (mm2d3d_amd/projection.py + synthetic.collate)."""
import os

import numpy as np
import pytest
import torch

from test_golden_leaves import G, VOX_CASES


@pytest.mark.parametrize("name", sorted(VOX_CASES))
def test_augmentation_draws_follow_the_reference_rng_order(name):
    """The host half of the GPU path: rotation matrix and translation draws, seeded like the golden fixture."""
    from mm2d3d_amd.dataprep import augmentation_draws

    z = np.load(os.path.join(G, "voxelize.npz"))
    np.random.seed(1234)
    rot, u = augmentation_draws(**VOX_CASES[name])
    assert rot.dtype == np.float32 and np.array_equal(rot, z[f"{name}/rot"])
    assert (u is not None) == bool(VOX_CASES[name].get("transl", False))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(VOX_CASES))
def test_gpu_voxeliser_bit_exact_vs_reference_golden(name):
    from mm2d3d_amd.dataprep import augmentation_draws, voxelize_batch

    z = np.load(os.path.join(G, "voxelize.npz"))
    pts = z["points"]
    np.random.seed(1234)
    rot, u = augmentation_draws(**VOX_CASES[name])
    # two copies of the scene in one batch: the second scene must give the same voxels with batch index 1
    both = torch.from_numpy(np.concatenate([pts, pts])).cuda()
    out = voxelize_batch(both, [len(pts), len(pts)], [rot, rot], [u, u], 20, 4096)
    vox, mask = z[f"{name}/voxels"], z[f"{name}/mask"]
    n = len(vox)
    assert out["counts"] == [n, n]
    locs = out["locs"].cpu().numpy()
    assert locs.dtype == np.int64 and locs.shape == (2 * n, 4)
    assert np.array_equal(locs[:n, :3], vox) and np.array_equal(locs[n:, :3], vox)
    assert np.all(locs[:n, 3] == 0) and np.all(locs[n:, 3] == 1)
    keep = out["keep"].cpu().numpy()
    assert np.array_equal(keep[:n], np.nonzero(mask)[0]) and np.array_equal(keep[n:], np.nonzero(mask)[0] + len(pts))
    assert np.array_equal(out["min_value"].cpu().numpy()[0], z[f"{name}/min_value"])
    assert np.array_equal(out["offset"].cpu().numpy()[1], z[f"{name}/offset"])


@pytest.mark.gpu
def test_gpu_voxeliser_range_mask_and_empty_scene():
    """Points that leave the receptive field are dropped in place (order kept); an empty scene in the batch is fine."""
    from mm2d3d_amd.dataprep import voxelize_batch
    from mm2d3d_amd.voxelize import augment_and_scale_3d, voxelize_points

    rng = np.random.default_rng(5)
    a = (rng.standard_normal((5000, 3)) * 25).astype(np.float32)  # ~180 m spread * 20 > 2048 voxels: many rows masked
    b = np.zeros((0, 3), np.float32)
    c = (rng.standard_normal((300, 3)) * 3).astype(np.float32)
    eye = np.eye(3, dtype=np.float32)
    out = voxelize_batch(torch.from_numpy(np.concatenate([a, b, c])).cuda(), [5000, 0, 300], [eye] * 3, [None] * 3, 20, 2048)
    ref = []
    for i, p in enumerate((a, b, c)):
        if len(p) == 0:
            ref.append(np.zeros((0, 4), np.int64))
            continue
        cf, _, _, _ = augment_and_scale_3d(p.copy(), 20, 2048)
        v, _ = voxelize_points(cf, 2048)
        ref.append(np.concatenate([v, np.full((len(v), 1), i, np.int64)], 1))
    assert out["counts"] == [len(r) for r in ref] and 0 < out["counts"][0] < 5000
    assert np.array_equal(out["locs"].cpu().numpy(), np.concatenate(ref))


@pytest.mark.gpu
def test_gpu_prepare_batch_equals_host_loader_and_collate():
    """Whole path: fliplr, augmentation, voxelisation, rasterised depth / 2D labels (duplicates: last write wins), RGB
    features, collate - every tensor identical to the host restatement run with the same seeds."""
    from mm2d3d_amd import dataprep
    from mm2d3d_amd.projection import make_sample
    from mm2d3d_amd.synthetic import collate, lidar_sweep

    H, W = 60, 96
    aug = dict(noisy_rot=0.1, flip_x=0.5, rot_y=6.2831, transl=True)
    rng = np.random.default_rng(11)
    scenes, host = [], []
    for i in range(3):
        pts = lidar_sweep(40 + i, "nuscenes")[:: 7 + i].copy()
        n = len(pts)
        pimg = np.stack([rng.uniform(0, H - 1e-3, n), rng.uniform(0, W - 1e-3, n)], 1).astype(np.float32)
        pimg[: n // 3] = pimg[n // 3 : 2 * (n // 3)]  # heavy duplicates: several points per pixel
        lab = rng.integers(0, 6, n).astype(np.int64)
        img = rng.random((H, W, 3), dtype=np.float32)
        scenes.append(dict(points=pts, points_img=pimg, depth=pts[:, 2].copy(), seg_label=lab, img=np.moveaxis(img, -1, 0).copy()))
        host.append((pts, pimg, lab, img))
    np.random.seed(70)
    ref = [make_sample(p, p, pi, l, np.eye(3), im, camera_coords=True, noisy_rot=0.1, flip_x=0.5, rot=6.2831, transl=True, fliplr=0.5)
           for p, pi, l, im in host]
    np.random.seed(70)
    out = dataprep.prepare_batch(scenes, augmentation=aug, fliplr=0.5, want_seg2d=True)
    assert any(out["fliplr"]) and not all(out["fliplr"]), "the seed exercises both flip states"
    cb = collate(ref)
    assert torch.equal(out["x"][0].cpu(), cb["x"][0])
    assert torch.equal(out["x"][1].cpu(), cb["x"][1])
    assert torch.equal(out["seg_label"].cpu(), cb["seg_label"])
    assert torch.equal(out["img"].cpu(), cb["img"])
    assert torch.equal(out["depth"].cpu(), cb["depth"])
    for a, b in zip(out["img_indices"], cb["img_indices"]):
        assert np.array_equal(a.cpu().numpy(), b)
    for i, r in enumerate(ref):
        assert np.array_equal(out["seg_labels_2d"][i].cpu().numpy(), r["seg_labels_2d"])
        assert np.array_equal(out["points"][i].cpu().numpy(), r["points"])
        assert np.array_equal(out["min_values"][i].cpu().numpy(), r["min_value"])
        assert np.array_equal(out["offsets"][i].cpu().numpy(), r["offset"])
        assert np.array_equal(out["rotation_matrices"][i], r["rot_matrix"])
