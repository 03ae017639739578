"""One SubmanifoldConvolution forward of the output-stationary engine, repeated: the target of rocprofv3 runs.
    python tools/os_one.py <level> <cin> <cout> [reps] [scenes]"""
import os
import sys

import torch

os.environ.setdefault("MM_OS_MIN_ROWS", "0")
os.environ.setdefault("MM_OS_UP", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd.scn import ops  # noqa: E402
from mm2d3d_amd.scn.metadata import Metadata  # noqa: E402
from mm2d3d_amd.synthetic import make_batch  # noqa: E402

l, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
scenes = int(sys.argv[5]) if len(sys.argv) > 5 else 16
dev = torch.device("cuda:0")
b = make_batch(2, scenes, "nuscenes", (32, 48), augment=True, device=dev)
md = Metadata(dev, 4096, 7)
md.build_levels(b["x"][0].contiguous())
md.build_rulebooks()
lv = md.levels[l]
x = torch.randn(lv.n, cin, device=dev)
w = torch.nn.Parameter(torch.randn(27, 1, cin, cout, device=dev) * 0.1)
with torch.no_grad():
    for _ in range(reps):
        y = ops.SparseConvFunction.apply(x, w, lv.subm, "subm", lv.n, lv.n)
torch.cuda.synchronize()
print("done", lv.n, lv.subm.n_rules, float(y.abs().mean()))
