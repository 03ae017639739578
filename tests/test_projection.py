"""Dataset-side projection: hand-computed cases of the semantics restated in
mm2d3d_amd/projection.py from lib/dataset/nuscenes_dataloader.py:236-369."""
import numpy as np
import pytest

from mm2d3d_amd import projection as pj
from mm2d3d_amd.synthetic import collate


def test_pixel_index_scaling_floors_then_scales_then_truncates():
    p = np.array([[899.9, 1599.2], [10.7, 3.9], [0.0, 0.0]], dtype=np.float32)
    s = pj.scale_image_points(p, (1600, 900), (400, 225))  # rows * 225/900, cols * 400/1600, applied to floor()
    assert np.allclose(s, [[899 * 0.25, 1599 * 0.25], [10 * 0.25, 3 * 0.25], [0, 0]])
    assert np.array_equal(s.astype(np.int64), [[224, 399], [2, 0], [0, 0]])
    assert np.array_equal(pj.scale_image_points(p, (400, 225), (400, 225)), p)
    with pytest.raises(AssertionError):
        pj.scale_image_points(p, (300, 200), (400, 225))


def test_depth_and_label_maps_last_write_wins_and_bounds():
    pimg = np.array([[1.2, 2.9], [0.0, 0.0], [1.9, 2.1], [3.0, 4.0]], dtype=np.float32)  # points 0 and 2 share pixel (1, 2)
    idx, depth, seg = pj.rasterise(pimg, np.array([5.0, 6.0, 7.0, 8.0]), np.array([1, 2, 3, 4]), 4, 5)
    assert np.array_equal(idx, [[1, 2], [0, 0], [1, 2], [3, 4]])
    assert depth[1, 2] == 7.0 and seg[1, 2] == 3  # the later point wins
    assert depth[0, 0] == 6.0 and depth[3, 4] == 8.0 and depth.sum() == 21.0
    assert (seg == -100).sum() == 20 - 3
    with pytest.raises(AssertionError):
        pj.rasterise(np.array([[4.0, 0.0]], dtype=np.float32), np.ones(1), np.ones(1), 4, 5)


def test_flip_remaps_columns_and_keeps_point_pixel_correspondence():
    g = np.random.default_rng(0)
    img = g.random((4, 5, 3), dtype=np.float32)
    pimg = np.array([[1, 2], [0, 0], [3, 4]], dtype=np.float32)
    idx, depth, seg = pj.rasterise(pimg, np.array([5.0, 6.0, 7.0]), np.array([1, 2, 3]), 4, 5)
    K = np.array([[100.0, 0.5, 2.0], [0.0, 100.0, 1.5], [0, 0, 1]])
    fi, fidx, fd, fs, fK = pj.flip_lr(img, idx, depth, seg, K)
    assert np.array_equal(fidx[:, 1], 4 - idx[:, 1]) and np.array_equal(fidx[:, 0], idx[:, 0])
    for (r, c), (fr, fc), z in zip(idx, fidx, (5.0, 6.0, 7.0)):
        assert np.array_equal(fi[fr, fc], img[r, c]) and fd[fr, fc] == z
    assert fK[0, 2] == 5 - 2.0 and fK[1, 2] == 4 - 0.5  # the reference's formula reads intrinsics[0, 1]


def test_make_sample_filters_every_per_point_array_and_gathers_rgb():
    g = np.random.default_rng(1)
    n, H, W = 200, 12, 16
    pts = (g.random((n, 3), dtype=np.float32) - 0.5) * 8
    pts[0] = [500.0, 0, 0]  # 500 m * 20 vox/m > full_scale: filtered out
    pimg = np.stack([g.integers(0, H, n), g.integers(0, W, n)], 1).astype(np.float32) + 0.3
    lab = g.integers(0, 6, n)
    img = g.random((H, W, 3), dtype=np.float32)
    np.random.seed(3)
    s = pj.make_sample(pts, pts, pimg, lab, np.eye(3), img, scale=20, full_scale=4096, fliplr=0.0)
    keep = s["coords"].shape[0]
    assert keep == n - 1
    assert s["points"].shape == (keep, 3) and s["seg_label"].shape == (keep,) and s["img_indices"].shape == (keep, 2)
    assert s["img"].shape == (3, H, W) and s["depth"].shape == (1, H, W) and s["depth"].dtype == np.float32
    assert np.array_equal(s["seg_label"], lab[1:]) and np.array_equal(s["img_indices"], pimg[1:].astype(np.int64))
    assert np.array_equal(s["feats"], img[s["img_indices"][:, 0], s["img_indices"][:, 1]])  # [n, 3] RGB under the points
    assert s["coords"].min() >= 0 and s["coords"].max() < 4096 and s["coords"].dtype == np.int64
    # the collate format the networks consume (lib/dataset/__init__.py:95-121)
    b = collate([s, s])
    assert b["x"][0].shape == (2 * keep, 4) and b["x"][0][keep:, 3].eq(1).all() and b["img"].shape == (2, 3, H, W)


def test_flip_draw_precedes_the_3d_augmentation_draws():
    g = np.random.default_rng(2)
    n, H, W = 50, 8, 8
    pts = (g.random((n, 3), dtype=np.float32) - 0.5) * 4
    pimg = np.stack([g.integers(0, H, n), g.integers(0, W, n)], 1).astype(np.float32)
    img = g.random((H, W, 3), dtype=np.float32)
    np.random.seed(5)
    first = np.random.rand()
    np.random.seed(5)
    s = pj.make_sample(pts, pts, pimg, np.zeros(n, np.int64), np.eye(3), img, fliplr=0.5, noisy_rot=0.1, flip_x=0.5, rot=6.2831,
                       transl=True)
    flipped = first < 0.5
    assert np.array_equal(s["img_indices"][:, 1], (W - 1 - pimg[:, 1].astype(np.int64)) if flipped else pimg[:, 1].astype(np.int64))
    assert not np.allclose(s["rot_matrix"], np.eye(3))
