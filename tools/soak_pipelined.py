"""Soak of the PIPELINED step (fit_step(batch, next_batch=): rulebooks of the next batch on the side stream beside the 3D backward,
2D trunk as HIP graphs): N steps over 4 rotating batches at a workload's size.  Run twice - identical output is the race detector;
a barrier fault raises.   python tools/soak_pipelined.py [steps] [c2|c4|c5]"""
import os, sys, math, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mm2d3d_amd.synthetic import make_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
wl = sys.argv[2] if len(sys.argv) > 2 else "c2"
dev = torch.device("cuda:0")
if wl == "c5":
    from mm2d3d_amd import scn
    scn.set_activation_dtype(torch.float16)
shape, B, kw = {"c2": ("nuscenes", 8, {}), "c4": ("kitti", 4, {}), "c5": ("kitti", 8, {})}[wl]
ncls = 6
tm = bench.build_trainer(dev, total_steps=steps + 10, train_kwargs={"sparse_activations": "fp16"} if wl == "c5" else None)
def mk(i):
    src = make_batch(2 + 2 * i, B, shape, (302, 480), ncls, device=dev, augment=True, **({"downsample": 10000} if wl == "c5" else {}))
    trg = make_batch(3 + 2 * i, B, shape, (302, 480), ncls, device=dev, augment=True)
    return {"source": src, "target": trg}
bs = [mk(i) for i in range(4)]
nxt = bench.fresh(bs[0])
t0 = time.time()
for i in range(steps):
    cur, nxt = nxt, bench.fresh(bs[(i + 1) % 4])
    loss = tm.fit_step(cur, next_batch=nxt)
    if i % 20 == 0 or i == steps - 1:
        v = float(loss)
        print(f"step {i:4d} loss {v:.9g}", flush=True)
        assert math.isfinite(v), "non-finite loss"
torch.cuda.synchronize()
print("OK", wl, steps, "steps")
