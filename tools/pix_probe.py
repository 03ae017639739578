"""Host cost of lifting.PixelIndex (pinned staging vs pin_memory), the branch-only rates, a host profile of the 2D branch and
the full step with the cyclic collector on / off."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from mm2d3d_amd.synthetic import make_batch
from mm2d3d_amd.lifting import PixelIndex
dev = torch.device("cuda", 0)
b = make_batch(2, 16, "nuscenes", (302, 480), 6, device=dev, augment=True)
for _ in range(3):
    PixelIndex(b["img_indices"], 302, 480, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    PixelIndex(b["img_indices"], 302, 480, dev)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"PixelIndex host time {1e3*(t1-t0)/20:.2f} ms per call; with drain {1e3*(t2-t0)/20:.2f} ms")
rows = [np.asarray(ix, dtype=np.int64).reshape(-1, 2) for ix in b["img_indices"]]
t0 = time.perf_counter()
for _ in range(20):
    rc = np.concatenate(rows, 0)
t1 = time.perf_counter()
for _ in range(20):
    h = torch.from_numpy(rc).pin_memory()
t2 = time.perf_counter()
for _ in range(20):
    for ix in rows:
        ok = ix.min() < 0 or ix[:, 0].max() >= 302 or ix[:, 1].max() >= 480
t3 = time.perf_counter()
print(f"concatenate {1e3*(t1-t0)/20:.2f} ms, pin_memory {1e3*(t2-t1)/20:.2f} ms, bounds check {1e3*(t3-t2)/20:.2f} ms")
tm = bench.build_trainer(dev)
batch = {"source": make_batch(2, 8, "nuscenes", (302, 480), 6, device=dev, augment=True), "target": make_batch(3, 8, "nuscenes", (302, 480), 6, device=dev, augment=True)}
print(bench.branch_rates(tm, batch, dev))
import gc
gc.disable()
print('gc disabled:', bench.branch_rates(tm, batch, dev))
gc.enable()
import cProfile, pstats
src = batch["source"]
def one():
    for o in tm.optimizers: o.zero_grad()
    bb = bench.fresh({"source": src})["source"]
    preds = tm(bb, model_name="2d_net")[0]
    tm.loss("segmentation", pred=preds["seg_logit"], gt=bb["seg_label"]).backward()
one(); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): one()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
import time
for label, dis in (('gc on', False), ('gc off', True)):
    gc.disable() if dis else gc.enable()
    for _ in range(3): tm.fit_step(bench.fresh(batch))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): tm.fit_step(bench.fresh(batch))
    torch.cuda.synchronize(); print(label, 'full step', (time.perf_counter() - t0) / 20 * 1e3, 'ms')
gc.enable()
