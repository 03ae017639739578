"""Generates the golden fixtures in this directory by IMPORTING the reference (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (/root/reference) never travels to the GPU box; only the small .npz vectors written here do.
Leaves importable in this container: lib.utils.augmentation_3d, lib.losses, lib.optimizers.
Inputs come from this repo's deterministic generator (mm2d3d_amd.synthetic), outputs from the reference functions.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from lib.utils.augmentation_3d import augment_and_scale_3d as ref_aug  # noqa: E402
from lib.losses import Loss as RefLoss  # noqa: E402
from lib.optimizers import Optimizer as RefOptimizer  # noqa: E402

from mm2d3d_amd.synthetic import lidar_sweep  # noqa: E402


def voxelize_cases():
    pts = lidar_sweep(3, "nuscenes")[::40].copy()  # 872 points keeps the fixture small
    cases = {
        "plain": dict(),
        "train_nuscenes": dict(noisy_rot=0.1, flip_x=0.5, rot_z=6.2831, transl=True),
        "train_kitti_cam": dict(noisy_rot=0.1, flip_y=0.5, rot_y=6.2831, transl=True),
        "flip_only": dict(flip_x=0.5),
        "transl_only": dict(transl=True),
    }
    out = {"points": pts}
    for name, kw in cases.items():
        np.random.seed(1234)
        coords, min_value, offset, rot = ref_aug(pts.copy(), 20, 4096, **kw)
        ic = coords.astype(np.int64)
        idxs = (ic.min(1) >= 0) * (ic.max(1) < 4096)
        out[f"{name}/coords_f"] = coords
        out[f"{name}/min_value"] = min_value
        out[f"{name}/offset"] = offset
        out[f"{name}/rot"] = rot
        out[f"{name}/voxels"] = ic[idxs]
        out[f"{name}/mask"] = idxs
    np.savez_compressed(os.path.join(HERE, "voxelize.npz"), **out)


def loss_cases():
    g = torch.Generator().manual_seed(7)
    logits = torch.randn(257, 6, generator=g) * 3
    labels = torch.randint(0, 6, (257,), generator=g)
    labels[::9] = -100
    weights = {  # class weights of the three shipped experiments (config.yaml:45 of each copy)
        "usa_sing": [2.47956584, 4.26788384, 5.71114131, 3.80241668, 1.0],
        "none": None,
    }
    out = {"logits": logits.numpy(), "labels": labels.numpy()}
    w6 = [1.5, 2.0, 0.5, 3.0, 1.0, 0.25]
    for name, w in {"w6": w6, "none": None}.items():
        cfg = [{"name": "cross_entropy", "weight": 1.0, "target": "segmentation", "args": ({"weight": w} if w else {})}]
        loss = RefLoss(cfg)
        x = logits.clone().requires_grad_(True)
        v = loss("segmentation", pred=x, gt=labels)
        v.backward()
        out[f"ce_{name}/value"] = v.detach().numpy()
        out[f"ce_{name}/grad"] = x.grad.numpy()
        out[f"ce_{name}/weight"] = np.array(w if w else [], np.float32)
    np.savez_compressed(os.path.join(HERE, "loss_ce.npz"), **out)


def optimizer_cases():
    p = torch.nn.Parameter(torch.linspace(-1, 1, 10))
    opt = RefOptimizer("adamw", lr=0.001)
    opt.set_scheduler("one_cycle", max_lr=0.005, total_steps=50)
    o, s = opt.build([p])
    lrs, vals = [], []
    g = torch.Generator().manual_seed(3)
    for _ in range(49):
        p.grad = torch.randn(10, generator=g)
        o.step()
        s.step()
        lrs.append(o.param_groups[0]["lr"])
        vals.append(p.detach().clone().numpy())
    np.savez_compressed(os.path.join(HERE, "optimizer_adamw_onecycle.npz"), lrs=np.array(lrs), params=np.stack(vals),
                        p0=np.linspace(-1, 1, 10, dtype=np.float32))


if __name__ == "__main__":
    voxelize_cases()
    loss_cases()
    optimizer_cases()
    print("golden fixtures written to", HERE)


def wiring_case():
    """The reference's OWN composition (3d_net/scn_unet.py: UNet recursion, channel plan, join order) executed over the
    oracle's primitives registered as ``sparseconvnet``: pins layer order / state_dict naming / return value of the
    composition against the reference's source.  Arithmetic inside the primitives is pinned by test_oracle_dense.py."""
    import importlib.util

    from oracle import scn_ref

    sys.modules["sparseconvnet"] = scn_ref
    spec = importlib.util.spec_from_file_location(
        "ref_scn_unet", "/root/reference/experiments_USA_SING/rgbd_rgbxyz_sigmoid_for_rgb/3d_net/scn_unet.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    for tag, kw in (("vgg", dict(residual_blocks=False)), ("res", dict(residual_blocks=True))):
        torch.manual_seed(11)
        net = mod.UNetSCN(in_channels=3, m=4, block_reps=1, full_scale=64, num_planes=4, **kw)
        g = np.random.default_rng(5)
        coords = np.concatenate([g.integers(0, 64, (300, 3)), g.integers(0, 2, (300, 1))], 1).astype(np.int64)
        coords = np.concatenate([coords, coords[:40]], 0)  # duplicates
        feats = g.standard_normal((len(coords), 3)).astype(np.float32)
        net.train()
        y = net([torch.from_numpy(coords), torch.from_numpy(feats)])
        out[f"{tag}/coords"], out[f"{tag}/feats"], out[f"{tag}/out"] = coords, feats, y.detach().numpy()
        for k, v in net.state_dict().items():
            out[f"{tag}/sd/{k}"] = v.numpy()  # AFTER the forward: running stats included
    np.savez_compressed(os.path.join(HERE, "wiring_unet.npz"), **out)
    del sys.modules["sparseconvnet"]


if __name__ == "__main__":
    wiring_case()
    print("wiring fixture written")
