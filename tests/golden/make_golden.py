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
        for k, v in net.state_dict().items():
            out[f"{tag}/sd0/{k}"] = v.clone().numpy()  # BEFORE the forward: the initial running statistics
        ft = torch.from_numpy(feats).requires_grad_(True)
        y = net([torch.from_numpy(coords), ft])
        out[f"{tag}/coords"], out[f"{tag}/feats"], out[f"{tag}/out"] = coords, feats, y.detach().numpy()
        for k, v in net.state_dict().items():
            out[f"{tag}/sd/{k}"] = v.numpy()  # AFTER the forward: running stats included
        # backward of a fixed linear functional through the reference's composition (VERDICT r3 missing #3): input and
        # weight gradients
        wlin = g.standard_normal(tuple(y.shape)).astype(np.float32)
        (y * torch.from_numpy(wlin)).sum().backward()
        out[f"{tag}/wlin"], out[f"{tag}/dfeats"] = wlin, ft.grad.numpy()
        for k, p_ in net.named_parameters():
            out[f"{tag}/grad/{k}"] = p_.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "wiring_unet.npz"), **out)
    del sys.modules["sparseconvnet"]


def _torchvision_standin():
    """torchvision is absent from this image: a minimal stand-in (the published ResNet34 topology over plain torch.nn,
    torchvision's attribute / state_dict names) registered in sys.modules IN THIS CONTAINER ONLY so that the reference's
    2d_net imports unchanged.  The stand-in's arithmetic is torch CPU's."""
    import types

    import torch.nn as nn

    class BasicBlock(nn.Module):
        def __init__(self, inpl, planes, stride, down, norm):
            super().__init__()
            self.conv1 = nn.Conv2d(inpl, planes, 3, stride, 1, bias=False)
            self.bn1 = norm(planes)
            self.relu = nn.ReLU(inplace=True)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = norm(planes)
            self.downsample = down

        def forward(self, x):
            idt = x if self.downsample is None else self.downsample(x)
            y = self.relu(self.bn1(self.conv1(x)))
            return self.relu(self.bn2(self.conv2(y)) + idt)

    class ResNet34(nn.Module):
        def __init__(self, norm):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
            self.bn1 = norm(64)
            self.relu = nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool2d(3, 2, 1)
            inpl = 64
            for li, (planes, n, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
                blocks = []
                for b in range(n):
                    s = stride if b == 0 else 1
                    down = None
                    if s != 1 or inpl != planes:
                        down = nn.Sequential(nn.Conv2d(inpl, planes, 1, s, bias=False), norm(planes))
                    blocks.append(BasicBlock(inpl, planes, s, down, norm))
                    inpl = planes
                setattr(self, f"layer{li}", nn.Sequential(*blocks))

    class FrozenBatchNorm2d(nn.Module):
        def __init__(self, n, eps=1e-5):
            super().__init__()
            self.eps = eps
            for k, v in (("weight", torch.ones(n)), ("bias", torch.zeros(n)), ("running_mean", torch.zeros(n)),
                         ("running_var", torch.ones(n))):
                self.register_buffer(k, v)

        def forward(self, x):
            sc = self.weight * (self.running_var + self.eps).rsqrt()
            return x * sc.view(1, -1, 1, 1) + (self.bias - self.running_mean * sc).view(1, -1, 1, 1)

    def resnet34(pretrained=False, norm_layer=None, **_):
        return ResNet34(norm_layer or nn.BatchNorm2d)  # no network: never pretrained

    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.0+standin"
    tv.ops = types.ModuleType("torchvision.ops")
    tv.ops.FrozenBatchNorm2d = FrozenBatchNorm2d
    tv.models = types.ModuleType("torchvision.models")
    tv.models.resnet = types.ModuleType("torchvision.models.resnet")
    tv.models.resnet.resnet34 = resnet34
    return {"torchvision": tv, "torchvision.ops": tv.ops, "torchvision.models": tv.models,
            "torchvision.models.resnet": tv.models.resnet}


def wiring_case_2d():
    """The reference's OWN 2d_net (model.py + backbones.py, imported unchanged by bare name as train.py:522 does) on a
    tiny padded batch in eval mode, weights regenerated from key names (tests/golden/fill.py).  Pins the decoder wiring
    (concat order [depth, up, rgb], stage channel plan, pad-to-16/crop, heads, per-sample lifting, return tuple)."""
    import importlib

    from fill import fill_state_dict

    mods = _torchvision_standin()
    sys.modules.update(mods)
    exp = "/root/reference/experiments_USA_SING/rgbd_rgbxyz_sigmoid_for_rgb"
    sys.path.insert(0, exp)
    try:
        ref = importlib.import_module("2d_net")
        out = {}
        for tag, frozen in (("bn", False), ("frozen", True)):
            net = ref.Model(num_classes=6, pretrained=False, frozen_batch_norm=frozen)
            net.load_state_dict(fill_state_dict(net.state_dict()))
            net.eval()
            g = np.random.default_rng(21)
            B, H, W = 2, 40, 52
            img = g.random((B, 3, H, W), dtype=np.float32)
            depth = np.zeros((B, 1, H, W), np.float32)
            idx = []
            for b in range(B):
                n = 57 + 13 * b
                ix = np.stack([g.integers(0, H, n), g.integers(0, W, n)], 1).astype(np.int64)
                depth[b, 0, ix[:, 0], ix[:, 1]] = g.uniform(1, 50, n).astype(np.float32)
                idx.append(ix)
            with torch.no_grad():
                preds, segm_last, _, aux = net({"img": torch.from_numpy(img), "depth": torch.from_numpy(depth),
                                                "img_indices": idx})
            out[f"{tag}/img"], out[f"{tag}/depth"] = img, depth
            for b in range(B):
                out[f"{tag}/idx{b}"] = idx[b]
            out[f"{tag}/seg_logit"] = preds["seg_logit"].numpy()
            out[f"{tag}/seg_logit_avg"] = aux["seg_logit_avg"].numpy()
            if not frozen:  # dense maps once (fixture size)
                out[f"{tag}/seg_logit_2d"] = preds["seg_logit_2d"].numpy()
                out[f"{tag}/segm_last"] = segm_last.numpy()
                out[f"{tag}/seg_logit_avg_2d"] = aux["seg_logit_avg_2d"].numpy()
            if not frozen:
                # train mode (batch statistics, dropout p = 0 for determinism), backward of a fixed linear functional of the two
                # point-logit outputs: updated running statistics + a digest of every weight gradient (fill.grad_digest)
                from fill import grad_digest

                net.train()
                for m in net.modules():
                    if isinstance(m, torch.nn.Dropout):
                        m.p = 0.0
                # its own, larger batch (2 x 94 x 126: layer4 then normalises over 96 values per channel instead of 24 - with
                # fewer, batch-statistics amplify rounding so much that 16-bit gradients cannot be told from wrong ones)
                Bt, Ht, Wt = 2, 94, 126
                img_t = g.random((Bt, 3, Ht, Wt), dtype=np.float32)
                depth_t = np.zeros((Bt, 1, Ht, Wt), np.float32)
                idx_t = []
                for b in range(Bt):
                    n = 301 + 37 * b
                    ix = np.stack([g.integers(0, Ht, n), g.integers(0, Wt, n)], 1).astype(np.int64)
                    depth_t[b, 0, ix[:, 0], ix[:, 1]] = g.uniform(1, 50, n).astype(np.float32)
                    idx_t.append(ix)
                out["train/img"], out["train/depth"] = img_t, depth_t
                for b in range(Bt):
                    out[f"train/idx{b}"] = idx_t[b]
                preds, segm_last, _, aux = net({"img": torch.from_numpy(img_t), "depth": torch.from_numpy(depth_t), "img_indices": idx_t})
                w1 = g.standard_normal(tuple(preds["seg_logit"].shape)).astype(np.float32)
                w2 = g.standard_normal(tuple(aux["seg_logit_avg"].shape)).astype(np.float32)
                ((preds["seg_logit"] * torch.from_numpy(w1)).sum() + (aux["seg_logit_avg"] * torch.from_numpy(w2)).sum()).backward()
                out["train/w1"], out["train/w2"] = w1, w2
                out["train/seg_logit"] = preds["seg_logit"].detach().numpy()
                out["train/seg_logit_avg"] = aux["seg_logit_avg"].detach().numpy()
                out["train/segm_last_crop"] = segm_last.detach().numpy()[:, :, ::9, ::11].copy()  # a strided sample of the decoder map
                for k, v in net.state_dict().items():
                    if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                        out[f"train/sd/{k}"] = v.numpy()
                out["train/grad_keys"] = np.array([k for k, p_ in net.named_parameters() if p_.grad is not None])
                out["train/nograd_keys"] = np.array([k for k, p_ in net.named_parameters() if p_.grad is None])
                for k, v in grad_digest([(k, p_.grad) for k, p_ in net.named_parameters() if p_.grad is not None]).items():
                    out[f"train/grad/{k}"] = v
                net.eval()
            out[f"{tag}/keys"] = np.array(sorted(net.state_dict().keys()))
            out[f"{tag}/shapes"] = np.array([str(tuple(net.state_dict()[k].shape)) for k in sorted(net.state_dict())])
        np.savez_compressed(os.path.join(HERE, "wiring_net2d.npz"), **out)
    finally:
        sys.path.remove(exp)
        for k in mods:
            sys.modules.pop(k, None)


if __name__ == "__main__":
    wiring_case()
    wiring_case_2d()
    print("wiring fixtures written")


def pselab_case():
    """lib/utils/refine_pseudo_labels.py (imported from the reference): per-class median thresholding of pseudo labels."""
    from lib.utils.refine_pseudo_labels import refine_pseudo_labels as ref_refine

    rng = np.random.default_rng(21)
    out = {}
    for name, n, ncls in (("even_odd", 1001, 6), ("small", 12, 3), ("single_class", 40, 1)):
        probs = rng.random(n).astype(np.float32)
        probs[rng.random(n) < 0.2] = 0.95  # ties above the 0.9 cap
        labels = rng.integers(0, ncls, n).astype(np.int64)
        out[f"{name}/probs"] = probs
        out[f"{name}/labels"] = labels
        out[f"{name}/refined"] = ref_refine(probs.copy(), labels.copy())
    np.savez_compressed(os.path.join(HERE, "pselab.npz"), **out)


if __name__ == "__main__":
    pselab_case()
    print("pseudo-label fixture written")


def step_case():
    """The reference's OWN training step: ``TrainModel._generic_step`` and ``cross_modal_loss`` (train.py:157-184, 186-292),
    imported unchanged, over the reference's own 2d_net / 3d_net plugins (loaded by bare name through its ModelWrapper,
    train.py:508-568) and its own Loss registry.  Stand-ins, in this process only, for what the image lacks:
    ``pytorch_lightning.LightningModule`` -> an nn.Module with ``log_dict`` recording what is logged, ``global_step`` = 1 (step 0
    draws a figure from ``preds_3d_fe["confidence"]``, a key the USA_SING 3d_net never sets: SURVEY.md section 2.1);
    ``torchmetrics.JaccardIndex`` and ``lib.utils.visualize`` (matplotlib / plyfile) -> inert stubs, never called by the step;
    ``torchvision`` -> the ResNet34 stand-in above; ``sparseconvnet`` -> oracle/scn_ref.py (the dependency is un-vendored).
    Pins the composition of row a17 against the reference's source: which logits feed which KL term, the detach, the lambda
    weights, the six logged keys, the summed loss - and every parameter gradient of that loss (digest)."""
    import importlib
    import types

    import torch.nn as nn
    from fill import fill_state_dict, grad_digest

    from mm2d3d_amd.synthetic import collate, make_scene
    from oracle import scn_ref

    logged = {}

    class LightningModule(nn.Module):
        global_step = 1
        current_epoch = 0
        loggers = [None, None]
        device = torch.device("cpu")

        def log_dict(self, d, **kw):
            logged.update({k: float(v.detach()) for k, v in d.items()})

        def log(self, k, v, **kw):
            logged[k] = float(v)

    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = LightningModule
    tmx = types.ModuleType("torchmetrics")
    tmx.JaccardIndex = type("JaccardIndex", (nn.Module,), {"__init__": lambda self, *a, **k: nn.Module.__init__(self),
                                                           "reset": lambda self: None})
    viz = types.ModuleType("lib.utils.visualize")
    viz.draw_points_image_labels_with_confidence = lambda *a, **k: None
    mods = dict(_torchvision_standin())
    mods.update({"pytorch_lightning": pl, "torchmetrics": tmx, "lib.utils.visualize": viz, "sparseconvnet": scn_ref})
    sys.modules.update(mods)
    exp = "/root/reference/experiments_USA_SING/rgbd_rgbxyz_sigmoid_for_rgb"
    sys.path.insert(0, exp)
    try:
        for stale in ("2d_net", "2d_net.model", "2d_net.backbones", "3d_net", "3d_net.model", "3d_net.scn_unet", "train"):
            sys.modules.pop(stale, None)
        ref_train = importlib.import_module("train")
        weights = [1.9241476, 1.0, 2.16763851, 2.78254323, 1.54875664, 1.85686537]  # config.yaml:45
        loss = RefLoss([{"name": "cross_entropy", "weight": 1.0, "target": "segmentation", "args": {"weight": weights}}])
        kw3d = dict(in_channels=3, m=16, block_reps=1, residual_blocks=False, full_scale=4096, num_planes=7)  # config.yaml:22-29
        tm = ref_train.TrainModel(
            model_modules=["2d_net", "3d_net"], optimizer=None, loss=loss,
            train_kwargs={"class_names": [str(i) for i in range(6)], "class_palette": [[0, 0, 0]] * 6, "lambda_xm_src": 1.0,
                          "lambda_xm_trg": 0.1},  # config.yaml:105-106
            model_kwargs={"2d_net": {"num_classes": 6, "pretrained": False},
                          "3d_net": {"num_classes": 6, "dual_head": True, "backbone_3d_kwargs": kw3d}})
        tm.model.load_state_dict(fill_state_dict(tm.model.state_dict()))
        tm.train()
        for m in tm.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        # 2 + 2 NuScenes-shaped scenes, 4,000 points each, 94 x 126 images (not multiples of 16: the reference binds `segm_last`
        # only on the padded path, 2d_net/model.py:126-129)
        batch = {"source": collate([make_scene(91000 + i, "nuscenes", (94, 126), 6, downsample=4000) for i in range(2)]),
                 "target": collate([make_scene(92000 + i, "nuscenes", (94, 126), 6, downsample=4000) for i in range(2)])}
        out = {}
        for dom, b in batch.items():
            out[f"{dom}/coords"], out[f"{dom}/feats"] = b["x"][0].numpy(), b["x"][1].numpy().copy()
            out[f"{dom}/img"], out[f"{dom}/depth"], out[f"{dom}/seg_label"] = b["img"].numpy(), b["depth"].numpy(), b["seg_label"].numpy()
            for i, ix in enumerate(b["img_indices"]):
                out[f"{dom}/idx{i}"] = ix
        # 3d_net/model.py:46-48 multiplies the features IN PLACE after nn.Linear has saved them for its weight gradient: in fp32
        # autograd refuses that backward (the reference only ever trains under fp16 autocast, where Linear saves its own fp16 copy,
        # run/train.yaml:11).  Saving every tensor as a copy gives fp32 autograd what autocast gives the reference: the linear
        # layer's weight gradient sees the UNGATED features.  The reference source stays unchanged.
        with torch.autograd.graph.saved_tensors_hooks(lambda t: t.clone(), lambda t: t):
            total = tm._generic_step(batch, "train")
        total.backward()
        out["total"] = np.array(float(total.detach()))
        out["log_keys"] = np.array(list(logged.keys()))
        out["log_values"] = np.array([logged[k] for k in logged])
        # cross_modal_loss on its own (train.py:157-184): four random logit sets
        g = torch.Generator().manual_seed(5)
        a, b_, c, d = (torch.randn(211, 6, generator=g) * 2 for _ in range(4))
        l2d, l3d = tm.cross_modal_loss(a, b_, c, d)
        out["xm/args"] = torch.stack([a, b_, c, d]).numpy()
        out["xm/values"] = np.array([float(l2d), float(l3d)])
        sd = tm.state_dict()
        out["state_dict_keys"] = np.array(sorted(sd.keys()))  # "model.2d_net.model.*" / "model.3d_net.model.*"
        for k, v in sd.items():
            if k.endswith(("running_mean", "running_var")):
                out[f"sd/{k}"] = v.numpy()
        named = [(k, p_.grad) for k, p_ in tm.named_parameters() if p_.grad is not None]
        out["grad_keys"] = np.array([k for k, _ in named])
        out["nograd_keys"] = np.array([k for k, p_ in tm.named_parameters() if p_.grad is None])
        for k, v in grad_digest(named).items():
            out[f"grad/{k}"] = v
        np.savez_compressed(os.path.join(HERE, "step_ref.npz"), **out)
        print("step fixture:", {k: round(v, 6) for k, v in logged.items()}, "total", float(total.detach()))
    finally:
        sys.path.remove(exp)
        for k in list(mods) + ["train", "2d_net", "2d_net.model", "2d_net.backbones", "3d_net", "3d_net.model", "3d_net.scn_unet"]:
            sys.modules.pop(k, None)


if __name__ == "__main__":
    step_case()
    print("step fixture written")
