"""Loader rows against fixtures produced by the REFERENCE's own dataset classes.

tests/golden/make_golden_loaders.py imports the reference's ``lib.dataset`` (stand-ins only for the absent
pytorch_lightning / omegaconf / torchvision.transforms / matplotlib), writes four miniature datasets in the reference's
on-disk formats (tests/golden/mini_ds) and stores what ``NuScenesLidarSegSCN / SemanticKITTISCN / A2D2SCN /
VirtualKITTISCN.__getitem__`` and ``collate_scn_base`` returned for seeded index lists.  Here the same files go through
mm2d3d_amd.datasets: the host path must reproduce every array bit for bit (dtype included), and so must the GPU path
(``gpu_batch`` -> csrc/dataprep.hip, ``-m gpu``).
"""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
MINI = os.path.join(G, "mini_ds")
NORM = ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
AUG3D = dict(noisy_rot=0.1, flip_x=0.5, rot=6.2831, transl=True)

# the constructor keywords of make_golden_loaders.CASES (kept in step by test_cases_match_the_generator)
CASES = {
    "nuscenes_train": ("NuScenesLidarSegSCN", "nuscenes", dict(
        split=("train_usa",), preprocess_dir="{root}", nuscenes_dir="{root}", merge_classes=True, resize=(80, 45), image_normalizer=NORM,
        fliplr=0.5, color_jitter=(0.4, 0.4, 0.4), camera_coords=False, use_rgb=True, **AUG3D)),
    "nuscenes_val_pselab": ("NuScenesLidarSegSCN", "nuscenes", dict(
        split=("test_singapore",), preprocess_dir="{root}", nuscenes_dir="{root}", merge_classes=True, resize=(80, 45),
        pselab_paths="{root}/pselab_test_singapore.npy", output_orig=True, camera_coords=True, use_rgb=False)),
    "skitti_bottom_crop": ("SemanticKITTISCN", "semantic_kitti", dict(
        split=("train",), preprocess_dir="{root}", semantic_kitti_dir="{root}", merge_classes_style="A2D2", crop_size=(48, 30),
        bottom_crop=True, fliplr=0.5, color_jitter=(0.4, 0.4, 0.4), pselab_paths="{root}/pselab_train.npy", use_rgb=True, **AUG3D)),
    "skitti_rand_crop": ("SemanticKITTISCN", "semantic_kitti", dict(
        split=("train",), preprocess_dir="{root}", semantic_kitti_dir="{root}", merge_classes_style="VirtualKITTI", crop_size=(40, 24),
        rand_crop=(0.5, 0.9, 0.4, 0.8), fliplr=0.5, image_normalizer=NORM, use_rgb=True, camera_coords=True, **AUG3D)),
    "skitti_val": ("SemanticKITTISCN", "semantic_kitti", dict(
        split=("val",), preprocess_dir="{root}", semantic_kitti_dir="{root}", merge_classes_style="nuScenes", output_orig=True,
        pselab_paths="{root}/pselab_val.npy", use_rgb=True)),
    "a2d2_train": ("A2D2SCN", "a2d2", dict(
        split=("train",), preprocess_dir="{root}", merge_classes=True, resize=(96, 60), rand_crop=(0.7, 0.5, 0.9, 0.5, 0.9), fliplr=0.5,
        color_jitter=(0.4, 0.4, 0.4), crop_size=(96, 60), bottom_crop=True, use_rgb=True, **AUG3D)),
    "vkitti_train": ("VirtualKITTISCN", "virtual_kitti", dict(
        split=("train",), preprocess_dir="{root}", virtual_kitti_dir="{root}", merge_classes=True, downsample=(900,), crop_size=(96, 60),
        bottom_crop=True, fliplr=0.5, color_jitter=(0.4, 0.4, 0.4), random_weather=("clone", "fog"), use_rgb=True, **AUG3D)),
    "vkitti_rand_crop": ("VirtualKITTISCN", "virtual_kitti", dict(
        split=("train",), preprocess_dir="{root}", virtual_kitti_dir="{root}", merge_classes=True, downsample=(-1,), crop_size=(120, 40),
        rand_crop=(0.6, 0.95, 0.5, 0.9), random_weather=None, camera_coords=True, use_rgb=True, image_normalizer=NORM)),
}
GPU_CASES = [c for c in CASES if c != "vkitti_rand_crop"]  # float64 points (VirtualKITTI + camera_coords): host path only


def _dataset(name):
    from mm2d3d_amd import datasets

    cls, sub, kw = CASES[name]
    root = os.path.join(MINI, sub)
    kw = {k: (v.replace("{root}", root) if isinstance(v, str) else v) for k, v in kw.items()}
    return getattr(datasets, cls)(**kw), kw


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.dtype == b.dtype, f"{what}: dtype {a.dtype} != {b.dtype}"
    assert a.shape == b.shape, f"{what}: shape {a.shape} != {b.shape}"
    assert np.array_equal(a, b), f"{what}: values differ"


def test_cases_match_the_generator():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden_loaders", os.path.join(G, "make_golden_loaders.py"))
    src = open(spec.origin).read()
    for name, (cls, sub, kw) in CASES.items():
        assert f'"{name}": ("{cls}", "{sub}"' in src
    assert src.count('": ("') == len(CASES)


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_samples_and_collate_equal_the_reference(name):
    """``ds[i]`` for the fixture's index list under the fixture's seeds = the reference's ``__getitem__`` outputs, key by
    key (crops, resizes, jitter placement, flips, rotations, voxels, range mask, features, pseudo labels), and
    ``collate_scn_base`` of them = the reference's batch."""
    from mm2d3d_amd.datasets import collate_scn_base

    z = np.load(os.path.join(G, f"loader_{name}.npz"))
    ds, kw = _dataset(name)
    assert len(ds) == int(z["len"]) and list(ds.class_names) == list(z["class_names"])
    np.random.seed(int(z["seed"]))
    torch.manual_seed(int(z["seed"]))
    samples = [ds[int(i)] for i in z["indices"]]
    for j, s in enumerate(samples):
        ref_keys = {k.split("/", 1)[1] for k in z.files if k.startswith(f"s{j}/")}
        assert set(s.keys()) == ref_keys, (set(s.keys()) ^ ref_keys)
        for k in ref_keys:
            r = z[f"s{j}/{k}"]
            if r.dtype.kind == "U" and str(r) == "None":
                assert s[k] is None, k
            else:
                _same(s[k], r, f"sample {j} {k}")
    b = collate_scn_base(samples, output_orig=bool(kw.get("output_orig", False)))
    ref_keys = {k.split("/")[1] for k in z.files if k.startswith("batch/")}
    assert ({"x0", "x1"} | (set(b.keys()) - {"x"})) == ref_keys
    _same(b["x"][0].numpy(), z["batch/x0"], "locs")
    _same(b["x"][1].numpy(), z["batch/x1"], "feats")
    for k in ref_keys - {"x0", "x1"}:
        if f"batch/{k}/len" in z.files:
            assert isinstance(b[k], list) and len(b[k]) == int(z[f"batch/{k}/len"]), k
            for i, e in enumerate(b[k]):
                _same(e.numpy() if isinstance(e, torch.Tensor) else e, z[f"batch/{k}/{i}"], f"batch {k}[{i}]")
        else:
            _same(b[k].numpy(), z[f"batch/{k}"], f"batch {k}")


def test_bottom_crop_takes_the_image_bottom_and_keeps_only_points_inside():
    """Property check of the crop on top of the fixture equality (semantic_kitti.py:326-392)."""
    ds, _ = _dataset("skitti_bottom_crop")
    np.random.seed(3)
    torch.manual_seed(3)
    s = ds[0]
    assert s["img"].shape == (3, 30, 48) and s["depth"].shape == (1, 30, 48) and s["seg_labels_2d"].shape == (30, 48)
    assert s["img_indices"][:, 0].max() < 30 and s["img_indices"][:, 1].max() < 48 and s["img_indices"].min() >= 0
    assert len(s["coords"]) < len(ds.data[0]["points"])  # points outside the window are gone
    assert len(s["pseudo_label_2d"]) == len(s["coords"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", GPU_CASES)
def test_gpu_batch_equals_the_reference_batch(name):
    """The same index list through ``gpu_batch`` (front end on the host, per-point work in csrc/dataprep.hip): every tensor
    of the reference's collated batch, bit for bit."""
    z = np.load(os.path.join(G, f"loader_{name}.npz"))
    ds, kw = _dataset(name)
    np.random.seed(int(z["seed"]))
    torch.manual_seed(int(z["seed"]))
    b = ds.gpu_batch([int(i) for i in z["indices"]], want_seg2d=True)
    host = lambda t: t.cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    _same(host(b["x"][0]), z["batch/x0"], "locs")
    _same(host(b["x"][1]), z["batch/x1"], "feats")
    for k in ("seg_label", "img", "depth", "intrinsics", "seg_labels_2d", "min_values", "offsets", "rotation_matrices", "points", "coords"):
        _same(host(b[k]), z[f"batch/{k}"], k)
    lists = ["img_indices"] + (["orig_seg_label", "orig_points_idx"] if kw.get("output_orig") else [])
    for k in lists:
        assert len(b[k]) == int(z[f"batch/{k}/len"])
        for i, e in enumerate(b[k]):
            _same(host(e), z[f"batch/{k}/{i}"], f"{k}[{i}]")
    if "batch/pseudo_label_2d" in z.files:
        _same(host(b["pseudo_label_2d"]), z["batch/pseudo_label_2d"], "pseudo_label_2d")
        _same(host(b["pseudo_label_ensemble"]), z["batch/pseudo_label_ensemble"], "pseudo_label_ensemble")
        if "batch/pseudo_label_3d/len" in z.files:
            assert b["pseudo_label_3d"] == []
        else:
            _same(host(b["pseudo_label_3d"]), z["batch/pseudo_label_3d"], "pseudo_label_3d")


@pytest.mark.gpu
def test_training_step_runs_on_gpu_prepared_batches_and_equals_the_host_loader_step():
    """a ``gpu_batch()`` dict (device ``img_indices``) must feed ``TrainModel`` directly - no host round trip -
    and give the losses of the step on the host-collated batch of the same scenes (same seeds: identical inputs)."""
    import copy

    from mm2d3d_amd.datasets import collate_scn_base
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.train import TrainModel

    dev = torch.device("cuda:0")
    ds, _ = _dataset("nuscenes_train")
    torch.manual_seed(0)
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
    for m in n2.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
    loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {}}])
    tk = dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, gc_freeze=False)
    on_gpu = TrainModel({"2d_net": n2, "3d_net": n3}, None, loss, dict(tk))
    on_host = TrainModel({"2d_net": n2b, "3d_net": n3b}, None, loss, dict(tk))

    def host_batch(idx):
        b = collate_scn_base([ds[i] for i in idx], output_orig=False)
        out = dict(b)
        out["x"] = [b["x"][0].to(dev), b["x"][1].to(dev)]
        for k in ("seg_label", "img", "depth"):
            out[k] = b[k].to(dev)
        return out

    np.random.seed(77)
    torch.manual_seed(77)
    hb = {"source": host_batch([0, 1]), "target": host_batch([2, 0])}
    np.random.seed(77)
    torch.manual_seed(77)
    gb = {"source": ds.gpu_batch([0, 1]), "target": ds.gpu_batch([2, 0])}
    assert all(t.is_cuda for t in gb["source"]["img_indices"])
    lg = on_gpu.training_step(gb)
    lg.backward()
    lh = on_host.training_step(hb)
    lh.backward()
    torch.cuda.synchronize()
    for k, v in on_host.last_logs.items():
        assert on_gpu.last_logs[k].detach().item() == v.detach().item(), k
    for (name, p), (_, q) in zip(n3.named_parameters(), n3b.named_parameters()):
        assert (p.grad is None) == (q.grad is None) and (p.grad is None or torch.equal(p.grad, q.grad)), name
