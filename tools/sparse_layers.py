"""Per-layer listing of the sparse-conv engine calls of ONE training step of the 3D branch on the bench's joint batch (8 + 8 augmented
NuScenes-shaped scenes; --workload c4 / c5: BASELINE.json configs[3] / [4]): every call under its own HIP event pair (second backward
stream off), with its rule count, widths, engine and algorithmic GB/s (SURVEY.md 8d formula).  VERDICT r5 asked for this listing in
profiles/r06/.  usage: python tools/sparse_layers.py [--workload c2|c4|c5] [--reps 5]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd import scn  # noqa: E402
from mm2d3d_amd.net3d import Net3DSeg  # noqa: E402
from mm2d3d_amd.scn import ops  # noqa: E402
from mm2d3d_amd.synthetic import make_batch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2", choices=["c2", "c4", "c5"])
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--act", default="fp16")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
esize = 4
if a.workload == "c5":
    scn.set_activation_dtype(torch.float16 if a.act == "fp16" else torch.bfloat16)
    esize = 2
net = Net3DSeg(6, True, dict(in_channels=3, m=16, full_scale=4096, num_planes=7)).to(dev)
shape, n, down, cid = {"c2": ("nuscenes", 8, 0, (2, 3)), "c4": ("kitti", 4, 0, (4, 5)), "c5": ("kitti", 8, 10000, (6, 7))}[a.workload]
src = make_batch(cid[0], n, shape, img_hw=(32, 48), device=dev, augment=True, downsample=down)
trg = make_batch(cid[1], n, shape, img_hw=(32, 48), device=dev, augment=True)
ct = trg["x"][0].clone()
ct[:, -1] += n
coords, feats = torch.cat([src["x"][0], ct], 0), torch.cat([src["x"][1], trg["x"][1]], 0)
print("points", coords.shape[0], "workload", a.workload)
ops.BWD_OVERLAP[0] = False
ops.PROFILE_LEAD_CYCLES = int(60e6)  # the GPU sleeps ~30 ms at the first engine call: the host queues the whole step meanwhile, so the
# event pairs of the small layers time kernels, not the host's launch cadence


def step():
    preds, f, aux = net({"x": [coords, feats.clone()]})
    (preds["seg_logit"].float().sum() + aux["seg_logit_point"].float().sum()).backward()


for _ in range(3):
    step()
runs = []
for _ in range(a.reps):
    ops.PROFILE = []
    step()
    torch.cuda.synchronize()
    runs.append([(r["kind"], r["eng"], r["R"], r["K"], r["cin"], r["cout"], r["bytes"], r["e0"].elapsed_time(r["e1"]) * 1e3) for r in ops.PROFILE])
    ops.PROFILE = None
tot = {}
print(f"{'kind':5s} {'eng':4s} {'rules':>9s} {'K':>2s} {'cin':>4s} {'cout':>4s} {'MB':>7s} {'us':>7s} {'GB/s':>6s}")
for i, r0 in enumerate(runs[0]):
    us = statistics.median(run[i][7] for run in runs)
    kind, eng, R, K, cin, cout, b = r0[:7]
    print(f"{kind:5s} {eng:4s} {R:9d} {K:2d} {cin:4d} {cout:4d} {b / 1e6:7.1f} {us:7.1f} {b / us / 1e3:6.0f}")
    t = tot.setdefault((kind, eng), [0.0, 0.0, 0])
    t[0] += b
    t[1] += us
    t[2] += 1
print("--- by (pass, engine)")
for k, (b, us, c) in sorted(tot.items()):
    print(f"{k[0]:5s} {k[1]:4s} calls {c:3d}  {b / 1e6:8.1f} MB {us:8.1f} us  {b / us / 1e3:6.0f} GB/s")
B = sum(v[0] for v in tot.values())
U = sum(v[1] for v in tot.values())
print(f"total {B / 1e9:.3f} GB algorithmic, {U / 1e3:.3f} ms (event pairs, incl. ~4.7 us overhead each) -> {B / U / 1e3:.0f} GB/s = {B / U / 1e3 / 8000:.3f} of 8 TB/s")
