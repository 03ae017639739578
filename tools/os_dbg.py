"""Times the output-stationary engine per layer (forward) with the bring-up switches of csrc/osconv.hip (MM_OS_DBG)."""
import os
import sys

import torch

os.environ.setdefault("MM_OS_MIN_ROWS", "0")
os.environ.setdefault("MM_OS_UP", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd.scn import ops  # noqa: E402
from mm2d3d_amd.scn.metadata import Metadata  # noqa: E402
from mm2d3d_amd.synthetic import make_batch  # noqa: E402
from tools.os_check import timeit  # noqa: E402

dev = torch.device("cuda:0")
b = make_batch(2, int(sys.argv[1]) if len(sys.argv) > 1 else 16, "nuscenes", (32, 48), augment=True, device=dev)
md = Metadata(dev, 4096, 7)
md.build_levels(b["x"][0].contiguous())
md.build_rulebooks()
for l, lv in enumerate(md.levels[:6]):
    p = 16 * (l + 1)
    t = lv.subm.os
    pc = torch.tensor([bin(int(v) & 0xFFFFFFFF).count("1") for v in t.tmask.cpu().tolist()], dtype=torch.float32)
    for cin, cout in ((p, p), (2 * p, p)):
        x = torch.randn(lv.n, cin, device=dev)
        w = torch.nn.Parameter(torch.randn(27, 1, cin, cout, device=dev) * 0.1)
        row = []
        with torch.no_grad():
            for dbg in (0, 1, 2, 3, 4, 6, 7):
                os.environ["MM_OS_DBG"] = str(dbg)
                row.append(timeit(lambda: ops.SparseConvFunction.apply(x, w, lv.subm, "subm", lv.n, lv.n)))
        os.environ["MM_OS_DBG"] = "0"
        print(f"L{l} {cin:3d}->{cout:3d} n={lv.n:7d} tiles={t.n_tiles:5d}x{t.tile_rows} k/tile={pc.mean():.1f} (p50 {pc.median():.0f} p90 {pc.quantile(0.9):.0f} max {pc.max():.0f})  full {row[0]:6.1f} | noMFMA {row[1]:6.1f} | "
              f"noGather {row[2]:6.1f} | noMFMA+noGather {row[3]:6.1f} | noSplit {row[4]:6.1f} | noGather+noSplit {row[5]:6.1f} | none {row[6]:6.1f}", flush=True)
