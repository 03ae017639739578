"""How far ahead of the GPU does the host run?  Per step: host time inside fit_step (it ends with no synchronisation of its own
except the two read-backs of the sparse metadata build) against the GPU time of the step (HIP events)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mm2d3d_amd.synthetic import make_batch  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
tm = bench.build_trainer(dev)
batch = {"source": make_batch(2, 8, "nuscenes", (302, 480), 6, device=dev, augment=True),
         "target": make_batch(3, 8, "nuscenes", (302, 480), 6, device=dev, augment=True)}
for _ in range(5):
    tm.fit_step(bench.fresh(batch))
torch.cuda.synchronize()
N = 20
host, marks = [], [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
t_all = time.perf_counter()
marks[0].record()
for i in range(N):
    b = bench.fresh(batch)
    t0 = time.perf_counter()
    tm.fit_step(b)
    host.append(time.perf_counter() - t0)
    marks[i + 1].record()
t_enq = time.perf_counter() - t_all
torch.cuda.synchronize()
t_wall = time.perf_counter() - t_all
gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(N)]
print(f"host inside fit_step: median {sorted(host)[N // 2] * 1e3:.1f} ms; GPU per step: median {sorted(gpu)[N // 2]:.1f} ms")
print(f"all {N} steps enqueued after {t_enq * 1e3:.0f} ms, finished after {t_wall * 1e3:.0f} ms")
# the same with the metadata read-backs taken out of the picture: time only the 2D forward enqueue
import cProfile
import pstats

pr = cProfile.Profile()
pr.enable()
for i in range(3):
    tm.fit_step(bench.fresh(batch))
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
