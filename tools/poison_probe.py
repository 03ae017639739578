"""Does any result of the training step depend on uninitialised memory?  torch.empty / empty_like / new_empty are patched to
poison what they return (NaN for floating types, 0x7f7f... for integers) and the loss of three steps is compared with an unpatched
run of the same steps: it must be bit-identical (every buffer is fully written before it is read)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mm2d3d_amd.synthetic import make_batch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def run():
    tm = bench.build_trainer(dev)
    batch = {"source": make_batch(2, 8, "nuscenes", (302, 480), 6, device=dev, augment=True),
             "target": make_batch(3, 8, "nuscenes", (302, 480), 6, device=dev, augment=True)}
    return [float(tm.fit_step(bench.fresh(batch))) for _ in range(3)]


clean = run()
e0, el0 = torch.empty, torch.empty_like
ne0 = torch.Tensor.new_empty


def poison(t):
    if t.is_cuda and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64, torch.uint8, torch.int16):
            t.view(torch.uint8).fill_(0x7F)
    return t


torch.empty = lambda *a, **k: poison(e0(*a, **k))
torch.empty_like = lambda *a, **k: poison(el0(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: poison(ne0(self, *a, **k))
try:
    dirty = run()
finally:
    torch.empty, torch.empty_like, torch.Tensor.new_empty = e0, el0, ne0
print("clean :", clean)
print("poison:", dirty)
print("IDENTICAL" if clean == dirty else "DIFFERENT: some kernel reads memory it (or its producer) never wrote")
