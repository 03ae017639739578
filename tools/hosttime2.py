"""Host-side enqueue profile of the pipelined training step (next_batch passed, 2D trunk as HIP graphs): cProfile over 3 steps issued
into an empty queue each."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mm2d3d_amd.synthetic import make_batch
dev = torch.device("cuda:0")
tm = bench.build_trainer(dev)
mk = lambda s: {"source": make_batch(2, 8, "nuscenes", (302, 480), device=dev, augment=True, first_scene=s),
                "target": make_batch(3, 8, "nuscenes", (302, 480), device=dev, augment=True, first_scene=s)}
bs = [mk(0), mk(8)]
nxt = bench.fresh(bs[0])
for i in range(6):
    cur, nxt = nxt, bench.fresh(bs[(i + 1) % 2])
    tm.fit_step(cur, next_batch=nxt)
torch.cuda.synchronize()
ts = []
pr = cProfile.Profile()
for i in range(4):
    cur, nxt = nxt, bench.fresh(bs[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if i: pr.enable()
    tm.fit_step(cur, next_batch=nxt)
    if i: pr.disable()
    ts.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("host enqueue per step (empty queue):", [round(t, 2) for t in ts])
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
