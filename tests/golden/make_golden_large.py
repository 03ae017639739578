"""Large reference-derived fixtures (round 5): gradients that can tell a right 16-bit gradient from a wrong one.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_large.py [2d] [step]

Run in the build container only (imports /root/reference; nothing of it travels).  Both fixtures regenerate their INPUTS from
seeds (tests/golden/fill.py, mm2d3d_amd.synthetic) - only outputs, running statistics and 32-projection gradient digests are stored.

wiring_net2d_large.npz - the reference's own 2d_net (model.py + backbones.py, imported unchanged over the torch.nn ResNet34
  stand-in of make_golden.py) in TRAIN mode on 4 x 222 x 286 images (padded to 224 x 288: layer4 normalises over 1,008 values per
  channel), backward of a fixed linear functional, three runs of the same source: fp32 ("f32"), under
  torch.autocast("cpu", dtype=torch.float16) ("f16": the reference's own training precision class, run/train.yaml:11, loss
  scaled by 256 as a GradScaler would) and under bfloat16 autocast ("b16").  The 16-bit runs say how far the REFERENCE's own 16-bit
  gradients sit from its fp32 ones on this batch - the yardstick for the HIP kernels' 16-bit gradients.
step_ref_large.npz - the reference's own TrainModel._generic_step (train.py:186-292, imported unchanged behind the stand-ins of
  make_golden.py:step_case) on 4 + 4 NuScenes-shaped scenes (4,000 points, 222 x 286 images), fp32: six logged terms, total,
  running statistics, gradient digest of every parameter.
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import fill  # noqa: E402
import make_golden as mg  # noqa: E402  (its __main__ blocks do not run on import)

EXP = "/root/reference/experiments_USA_SING/rgbd_rgbxyz_sigmoid_for_rgb"


def case_2d():
    mods = mg._torchvision_standin()
    sys.modules.update(mods)
    sys.path.insert(0, EXP)
    try:
        ref = importlib.import_module("2d_net")
        img, depth, idx = fill.large_inputs_2d()
        n_pts = sum(len(i) for i in idx)
        w1, w2 = fill.large_functional_2d(n_pts)
        out = {"n_points": np.array(n_pts)}
        for tag, dtype, scale in (("f32", None, 1.0), ("b16", torch.bfloat16, 1.0), ("f16", torch.float16, 256.0)):
            t0 = time.time()
            net = ref.Model(num_classes=6, pretrained=False, frozen_batch_norm=False)
            net.load_state_dict(fill.fill_state_dict(net.state_dict()))
            net.train()
            for m in net.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
            batch = {"img": torch.from_numpy(img), "depth": torch.from_numpy(depth), "img_indices": idx}
            if dtype is None:
                preds, segm_last, _, aux = net(batch)
            else:
                with torch.autocast("cpu", dtype=dtype):
                    preds, segm_last, _, aux = net(batch)
            loss = (preds["seg_logit"].float() * torch.from_numpy(w1)).sum() + (aux["seg_logit_avg"].float() * torch.from_numpy(w2)).sum()
            (loss * scale).backward()
            out[f"{tag}/seg_logit"] = preds["seg_logit"].detach().float().numpy()
            out[f"{tag}/seg_logit_avg"] = aux["seg_logit_avg"].detach().float().numpy()
            out[f"{tag}/segm_last_crop"] = segm_last.detach().float().numpy()[:, :, ::13, ::17].copy()
            if dtype is None:
                for k, v in net.state_dict().items():
                    if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                        out[f"{tag}/sd/{k}"] = v.numpy()
                out["grad_keys"] = np.array([k for k, p_ in net.named_parameters() if p_.grad is not None])
            named = [(k, p_.grad / scale) for k, p_ in net.named_parameters() if p_.grad is not None]
            for k, v in fill.grad_digest(named, nproj=fill.NPROJ_LARGE).items():
                out[f"{tag}/grad/{k}"] = v
            print(f"2d large [{tag}]: {time.time() - t0:.0f} s, loss {float(loss):.6f}", flush=True)
        np.savez_compressed(os.path.join(HERE, "wiring_net2d_large.npz"), **out)
    finally:
        sys.path.remove(EXP)
        for k in list(mods) + ["2d_net", "2d_net.model", "2d_net.backbones"]:
            sys.modules.pop(k, None)


def case_step(variants=("f32", "f16")):
    import types

    import torch.nn as nn
    from lib.losses import Loss as RefLoss

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
    mods = dict(mg._torchvision_standin())
    mods.update({"pytorch_lightning": pl, "torchmetrics": tmx, "lib.utils.visualize": viz, "sparseconvnet": scn_ref})
    sys.modules.update(mods)
    sys.path.insert(0, EXP)
    try:
        for stale in ("2d_net", "2d_net.model", "2d_net.backbones", "3d_net", "3d_net.model", "3d_net.scn_unet", "train"):
            sys.modules.pop(stale, None)
        ref_train = importlib.import_module("train")
        weights = [1.9241476, 1.0, 2.16763851, 2.78254323, 1.54875664, 1.85686537]  # config.yaml:45
        loss = RefLoss([{"name": "cross_entropy", "weight": 1.0, "target": "segmentation", "args": {"weight": weights}}])
        kw3d = dict(in_channels=3, m=16, block_reps=1, residual_blocks=False, full_scale=4096, num_planes=7)
        tm = ref_train.TrainModel(
            model_modules=["2d_net", "3d_net"], optimizer=None, loss=loss,
            train_kwargs={"class_names": [str(i) for i in range(6)], "class_palette": [[0, 0, 0]] * 6, "lambda_xm_src": 1.0,
                          "lambda_xm_trg": 0.1},
            model_kwargs={"2d_net": {"num_classes": 6, "pretrained": False},
                          "3d_net": {"num_classes": 6, "dual_head": True, "backbone_3d_kwargs": kw3d}})
        tm.model.load_state_dict(fill.fill_state_dict(tm.model.state_dict()))
        tm.train()
        for m in tm.modules():
            if isinstance(m, nn.Dropout):
                m.p = 0.0
        S = fill.LARGE_STEP
        batch = {"source": collate([make_scene(93000 + i, "nuscenes", (S["H"], S["W"]), 6, downsample=S["points"]) for i in range(S["scenes"])]),
                 "target": collate([make_scene(94000 + i, "nuscenes", (S["H"], S["W"]), 6, downsample=S["points"]) for i in range(S["scenes"])])}
        path = os.path.join(HERE, "step_ref_large.npz")
        out = dict(np.load(path)) if os.path.exists(path) else {}
        out["points"] = np.array([batch[d]["x"][0].shape[0] for d in ("source", "target")])
        initial = {k: v.clone() for k, v in tm.state_dict().items()}

        def fresh():
            b = {}
            for dom, d in batch.items():
                nb = dict(d)
                nb["x"] = [d["x"][0], d["x"][1].clone()]  # the 3D net gates the features in place
                b[dom] = nb
            return b

        if "f32" in variants:
            t0 = time.time()
            with torch.autograd.graph.saved_tensors_hooks(lambda t: t.clone(), lambda t: t):  # see make_golden.py:step_case
                total = tm._generic_step(fresh(), "train")
            total.backward()
            out["total"] = np.array(float(total.detach()))
            out["log_keys"] = np.array(list(logged.keys()))
            out["log_values"] = np.array([logged[k] for k in logged])
            for k, v in tm.state_dict().items():
                if k.endswith(("running_mean", "running_var")):
                    out[f"sd/{k}"] = v.numpy().copy()
            named = [(k, p_.grad) for k, p_ in tm.named_parameters() if p_.grad is not None]
            out["grad_keys"] = np.array([k for k, _ in named])
            for k, v in fill.grad_digest(named, nproj=fill.NPROJ_LARGE).items():
                out[f"grad/{k}"] = v
            print(f"step large [f32]: {time.time() - t0:.0f} s", {k: round(v, 6) for k, v in logged.items()}, "total", float(total.detach()), flush=True)
        if "f16" in variants:
            # The reference's own training precision class (run/train.yaml:11 precision: 16 = Lightning native AMP): the same
            # _generic_step under torch.autocast(float16), loss scaled by 1024 as a GradScaler would.  SparseConvNet's operators are
            # not autocast-aware - under AMP they compute in fp32 while the nn.Linear layers around them run in fp16 - so the
            # oracle primitives that stand in for them are kept out of autocast (a wrapper around the reference class's forward,
            # in this process only; its source stays as it is).
            t0 = time.time()
            tm.load_state_dict(initial)
            tm.zero_grad(set_to_none=True)
            logged.clear()
            unet = sys.modules["3d_net.scn_unet"].UNetSCN
            orig_forward = unet.forward

            def forward_fp32(self, x):
                with torch.autocast("cpu", enabled=False):
                    return orig_forward(self, [x[0], x[1].float()])

            unet.forward = forward_fp32
            try:
                with torch.autograd.graph.saved_tensors_hooks(lambda t: t.clone(), lambda t: t):
                    with torch.autocast("cpu", dtype=torch.float16):
                        total = tm._generic_step(fresh(), "train")
                (total.float() * 1024.0).backward()
            finally:
                unet.forward = orig_forward
            out["f16/total"] = np.array(float(total.detach()))
            out["f16/log_values"] = np.array([logged[k] for k in logged])
            named = [(k, p_.grad / 1024.0) for k, p_ in tm.named_parameters() if p_.grad is not None]
            for k, v in fill.grad_digest(named, nproj=fill.NPROJ_LARGE).items():
                out[f"f16/grad/{k}"] = v
            print(f"step large [f16]: {time.time() - t0:.0f} s", {k: round(v, 6) for k, v in logged.items()}, "total", float(total.detach()), flush=True)
        np.savez_compressed(path, **out)
    finally:
        sys.path.remove(EXP)
        for k in list(mods) + ["train", "2d_net", "2d_net.model", "2d_net.backbones", "3d_net", "3d_net.model", "3d_net.scn_unet"]:
            sys.modules.pop(k, None)


if __name__ == "__main__":
    which = sys.argv[1:] or ["2d", "step"]
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    if "2d" in which:
        case_2d()
    if "step" in which:
        case_step()
    if "step_f16" in which:  # adds the fp16-autocast variant to an existing step_ref_large.npz
        case_step(("f16",))
    print("large fixtures written to", HERE)
