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
a = ap.parse_args()

dev = torch.device("cuda:0")
torch.manual_seed(0)
if a.act16:
    from mm2d3d_amd import scn

    scn.set_activation_dtype(torch.bfloat16)
net = Net3DSeg(6, True, dict(in_channels=3, m=16, full_scale=4096, num_planes=7)).to(dev)
batch = make_batch(2, a.scenes, a.shape, img_hw=(32, 48), device=dev)
coords, feats = batch["x"]
print("points", coords.shape[0])


def step():
    preds, f, aux = net({"x": [coords, feats.clone()]})
    if not a.fwd_only:
        (preds["seg_logit"].sum() + aux["seg_logit_point"].sum()).backward()


for _ in range(a.warmup):
    step()
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
