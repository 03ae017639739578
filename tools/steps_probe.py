"""Per-step GPU (HIP events) and host times of 40 training steps plus the cyclic collector's pauses (gc.callbacks):
finds stalls that an average hides.  FREEZE=1 freezes the long-lived objects first (what TrainModel does)."""
import os, sys, time
import torch
sys.path.insert(0, os.getcwd())
import bench
from mm2d3d_amd.synthetic import make_batch
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
tm = bench.build_trainer(dev)
batch = {"source": make_batch(2, 8, "nuscenes", (302, 480), 6, device=dev, augment=True), "target": make_batch(3, 8, "nuscenes", (302, 480), 6, device=dev, augment=True)}
for _ in range(5): tm.fit_step(bench.fresh(batch))
import gc
log, t_gc = [], [0.0]
def cb(phase, info):
    if phase == 'start': t_gc[0] = time.perf_counter()
    else: log.append((info['generation'], (time.perf_counter() - t_gc[0]) * 1e3, info['collected']))
gc.callbacks.append(cb)
if os.environ.get('FREEZE'):
    gc.collect(); gc.freeze()
torch.cuda.synchronize()
N = int(os.environ.get('STEPS', '40'))
marks = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host, allocs = [], []
marks[0].record()
for i in range(N):
    t0 = time.perf_counter()
    tm.fit_step(bench.fresh(batch))
    host.append((time.perf_counter() - t0) * 1e3)
    st = torch.cuda.memory_stats(dev)
    allocs.append((st.get('num_device_alloc', 0), st.get('num_device_free', 0), st.get('num_alloc_retries', 0), st['reserved_bytes.all.current'] >> 20))
    marks[i + 1].record()
torch.cuda.synchronize()
gpu = [marks[i].elapsed_time(marks[i + 1]) for i in range(N)]
print("gpu :", " ".join(f"{t:.0f}" for t in gpu))
print("gc gen>=1:", [(g, round(ms, 1), c) for g, ms, c in log if g >= 1 or ms > 2])
print("device allocs (count, frees, retries, reserved MiB) at the slow steps:", [(i, allocs[i], allocs[i-1]) for i in range(1, N) if gpu[i] > 1.3 * sorted(gpu)[N // 2]])
print("reserved MiB first/last:", allocs[0][3], allocs[-1][3], " device allocs first/last:", allocs[0][0], allocs[-1][0])
print("host:", " ".join(f"{t:.0f}" for t in host))
