"""Which Python lines issue the small runtime copies / fills / torch kernels of a training step?  torch.profiler with stacks over ONE
eager step (MM_GRAPH2D=0 to see the 2D trunk's too):   python tools/launch_attrib.py [pattern ...]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mm2d3d_amd.synthetic import make_batch
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
tm = bench.build_trainer(dev)
mk = lambda s: {"source": make_batch(2, 8, "nuscenes", (302, 480), device=dev, augment=True, first_scene=s),
                "target": make_batch(3, 8, "nuscenes", (302, 480), device=dev, augment=True, first_scene=s)}
bs = [mk(0), mk(8)]
nxt = bench.fresh(bs[0])
for i in range(5):
    cur, nxt = nxt, bench.fresh(bs[(i + 1) % 2])
    tm.fit_step(cur, next_batch=nxt)
torch.cuda.synchronize()
pats = sys.argv[1:] or ["aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::cat", "aten::clone", "aten::contiguous", "aten::to", "aten::zeros", "aten::sum", "aten::mul", "aten::index", "aten::native_dropout", "aten::empty"]
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    cur, nxt = nxt, bench.fresh(bs[0])
    tm.fit_step(cur, next_batch=nxt)
torch.cuda.synchronize()
agg = collections.Counter()
for ev in prof.events():
    if any(ev.name == p or ev.name.startswith(p) for p in pats) and ev.cpu_parent is not None and not ev.cpu_parent.name.startswith("aten::"):
        st = [f for f in (ev.stack or []) if "mm2d3d_amd" in f or "bench.py" in f]
        agg[(ev.name, st[0].split("/repo/")[-1] if st else "?")] += 1
    elif any(ev.name == p or ev.name.startswith(p) for p in pats) and ev.cpu_parent is None:
        st = [f for f in (ev.stack or []) if "mm2d3d_amd" in f or "bench.py" in f]
        agg[(ev.name, st[0].split("/repo/")[-1] if st else "?")] += 1
for (name, where), n in sorted(agg.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{n:4d}  {name:28s} {where}")
