"""Stress for the grid barrier of the single-launch batch-norm kernels: BatchNorm2d forward + backward launches on the current
stream while a second stream keeps the GPU busy with small kernels of a chosen kind.  A launch whose grid cannot become resident
within 10 s traps (csrc/fused_bn.h), i.e. this script dies with a HIP error; otherwise it prints the rate.

    python tools/barrier_stress.py <kind> [seconds]      kind: none | sort | scan | elementwise | memset | metadata | metadata_onesweep

metadata: the sparse metadata build of a 16-scene batch (scn.prebuild_metadata) on the second stream, as TrainModel's
overlap_metadata option runs it (tile-table sort = merge sort); metadata_onesweep: the same with the Onesweep sort left on.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd import _lib  # noqa: E402
from mm2d3d_amd._lib import check, ptr, stream  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "sort"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
dev = torch.device("cuda", 0)
L = _lib.lib()
N, Ns, C = 145920, 72960, 128
bf = torch.bfloat16
x = torch.randn(N, C, device=dev).to(bf)
dy = torch.randn(N, C, device=dev).to(bf)
y, dx = torch.empty_like(x), torch.empty_like(x)
w, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)  # accumulate=1 below: dw / db grow, not compared
stats = torch.empty((2, 2, C), device=dev)
ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(C)), dev)
side = torch.cuda.Stream(dev)
keys = torch.randint(0, 1 << 27, (300000,), device=dev, dtype=torch.int32)
vals = torch.randn(1 << 20, device=dev)
small = torch.zeros(1 << 16, device=dev)
if kind.startswith("metadata"):
    from mm2d3d_amd import scn
    from mm2d3d_amd.synthetic import make_batch

    coords = make_batch(2, 16, "nuscenes", (32, 48), 6, device=dev)["x"][0]
    if kind == "metadata_onesweep":
        from mm2d3d_amd.scn import metadata as _md
        _md.NO_SPIN = type('K', (list,), {'__setitem__': lambda self, i, v: None})([0])  # leave Onesweep / look-back scans on


def side_work():
    with torch.cuda.stream(side):
        if kind == "sort":
            torch.sort(keys)
        elif kind == "scan":
            torch.cumsum(vals, 0)
        elif kind == "elementwise":
            vals.mul_(1.0001)
        elif kind == "memset":
            small.zero_()
    if kind.startswith("metadata"):
        c = coords.clone()
        scn.prebuild_metadata(c, 4096, side, torch.cuda.current_stream(dev).record_event())


ref = None
bad = 0
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < secs:
    for _ in range(20):
        if kind.startswith("metadata"):
            side_work()
        elif kind != "none":
            for _ in range(4):
                side_work()
        check(L.mm_bn2d_fwd_train(_lib.handle(x.device).h, ptr(x), C, None, C, N, Ns, C, ptr(w), ptr(b), ptr(rm), ptr(rv), None, 1e-5, 0.1, 1, ptr(y), C, ptr(stats[0]),
                                  ptr(stats[1]), ptr(ws), ws.numel(), stream()), "fwd")
        check(L.mm_bn2d_bwd(_lib.handle(x.device).h, ptr(x), C, ptr(dy), C, None, 0, None, C, 1, N, Ns, C, ptr(w), ptr(b), ptr(stats[0]), ptr(stats[1]), ptr(dx), C, None,
                            C, ptr(dw), ptr(db), 1, ptr(ws), ws.numel(), stream()), "bwd")
        n += 2
        if os.environ.get("CHECK"):  # same inputs every call: statistics, outputs and gradients must be bit-identical every time
            cur = (stats.clone(), y[::97].clone(), dx[::89].clone())
            if ref is None:
                ref = cur
            elif not all(torch.equal(a, b_) for a, b_ in zip(ref, cur)):
                bad += 1
    torch.cuda.synchronize()
print(f"{kind}: {bad} results differing from the first, " if os.environ.get("CHECK") else "", end="")
print(f"{kind}: {n} single-launch calls in {time.perf_counter() - t0:.1f} s ({(time.perf_counter() - t0) / n * 1e6:.1f} us per call), no stall")
