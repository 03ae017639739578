"""Host-side enqueue time vs GPU time of the training step (is the step launch-bound?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mm2d3d_amd.synthetic import make_batch
dev = torch.device("cuda:0")
tm = bench.build_trainer(dev)
batch = {"source": make_batch(2, 8, "nuscenes", (302, 480), device=dev, augment=True),
         "target": make_batch(3, 8, "nuscenes", (302, 480), device=dev, augment=True)}
for _ in range(3):
    tm.fit_step(bench.fresh(batch))
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    tm.fit_step(bench.fresh(batch))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.1f} ms, until GPU idle {1e3*(t2-t0):.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); tm.fit_step(bench.fresh(batch)); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
