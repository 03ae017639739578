"""Times the 3D branch (fwd+bwd) at a BASELINE.json configuration; used under rocprofv3 for kernel breakdowns."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd.net3d import Net3DSeg  # noqa: E402
from mm2d3d_amd.synthetic import make_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", type=int, default=8)
ap.add_argument("--shape", default="nuscenes")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--fwd-only", action="store_true")
ap.add_argument("--act16", action="store_true", help="16-bit sparse activations (BASELINE.json configs[4])")
ap.add_argument("--bench-batch", action="store_true",
                help="the joint [source | target] point set of bench.py's headline step (8 + 8 augmented NuScenes-shaped scenes, config ids "
                     "2 / 3): the PMC traffic record then refers to exactly the rule counts bench.py's roofline leg sees")
a = ap.parse_args()

dev = torch.device("cuda:0")
torch.manual_seed(0)
if a.act16:
    from mm2d3d_amd import scn

    scn.set_activation_dtype(torch.bfloat16)
net = Net3DSeg(6, True, dict(in_channels=3, m=16, full_scale=4096, num_planes=7)).to(dev)
if a.bench_batch:
    src = make_batch(2, 8, "nuscenes", img_hw=(32, 48), device=dev, augment=True)
    trg = make_batch(3, 8, "nuscenes", img_hw=(32, 48), device=dev, augment=True)
    ct = trg["x"][0].clone()
    ct[:, -1] += 8
    coords, feats = torch.cat([src["x"][0], ct], 0), torch.cat([src["x"][1], trg["x"][1]], 0)
else:
    batch = make_batch(2, a.scenes, a.shape, img_hw=(32, 48), device=dev)
    coords, feats = batch["x"]
print("points", coords.shape[0])


def step():
    preds, f, aux = net({"x": [coords, feats.clone()]})
    if not a.fwd_only:
        (preds["seg_logit"].sum() + aux["seg_logit_point"].sum()).backward()


for _ in range(a.warmup):
    step()
if not a.fwd_only:  # the algorithmic bytes of one step (SURVEY.md 8d formula over the actual rule counts)
    from mm2d3d_amd.scn import ops

    ops.PROFILE = []
    step()
    torch.cuda.synchronize()
    print("algorithmic_bytes_per_step", sum(r["bytes"] for r in ops.PROFILE))
    ops.PROFILE = None
torch.cuda.synchronize()
t0 = time.perf_counter()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    step()
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / a.steps * 1e3
print(f"3D branch {'fwd' if a.fwd_only else 'fwd+bwd'}: {e0.elapsed_time(e1) / a.steps:.3f} ms/step (gpu events), {wall:.3f} ms wall")
