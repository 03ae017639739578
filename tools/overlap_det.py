"""Which output stops being reproducible when the sparse metadata is built on a side stream (train_kwargs overlap_metadata)?
Fixed weights, dropout off: every iteration of training_step + backward must give bit-identical logits, losses and gradients.
Prints the first iterations at which each quantity differs from iteration 0."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mm2d3d_amd.synthetic import make_batch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tm = bench.build_trainer(dev)
for m in tm.model.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
batch = {"source": make_batch(2, 8, "nuscenes", (302, 480), 6, device=dev, augment=True),
         "target": make_batch(3, 8, "nuscenes", (302, 480), 6, device=dev, augment=True)}
names = [n for n, p in tm.model.named_parameters() if p.requires_grad]
pick = names[:: max(1, len(names) // 24)]
params = dict(tm.model.named_parameters())
ref, first_bad = None, {}
for it in range(iters):
    for o in tm.optimizers:
        o.zero_grad()
    loss = tm.training_step(bench.fresh(batch))
    loss.backward()
    cur = {"loss": loss.detach().clone()}
    for k, v in tm.last_logs.items():
        cur[k] = v.detach().clone()
    for o, tag in zip(tm.optimizers, ("grad_arena_2d", "grad_arena_3d")):
        for j, a in enumerate(getattr(o, "_arenas", [])):
            if a is not None:
                cur[f"{tag}_{j}"] = a["g"].clone()
    if ref is None:
        ref = cur
        continue
    for k in cur:
        if k not in first_bad and not torch.equal(cur[k], ref[k]):
            first_bad[k] = (it, float((cur[k].double() - ref[k].double()).abs().max()))
torch.cuda.synchronize()
print("overlap_metadata =", tm.overlap_metadata, "iterations", iters)
print("first iteration at which a quantity differs from iteration 0 (max abs difference):", first_bad if first_bad else "none")
