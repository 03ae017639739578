"""Per-shape timing of the 2D batch-norm kernels (train forward, backward) at the bench's layer shapes, as GB/s of the
compulsory traffic (fwd: read x twice + write y; bwd: read dy,x twice + write dx; bf16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd import nn2d

dev = torch.device("cuda:0")
B = 8
SHAPES = [(64, 304, 480), (64, 152, 240), (128, 76, 120), (256, 38, 60), (512, 19, 30)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for res in (False, True):
    for C, H, W in SHAPES:
        bn = nn2d.BatchNorm2d(C, relu=True).to(dev).train()
        x = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        r = torch.randn_like(x).requires_grad_(True) if res else None
        y = bn(x, r) if res else bn(x)
        g = torch.randn_like(y)
        nbytes = B * C * H * W * 2
        tf = timeit(lambda: bn(x, r) if res else bn(x))
        tb = timeit(lambda: torch.autograd.grad(y, [x] + ([r] if res else []), g, retain_graph=True))
        fb = nbytes * (3 + (1 if res else 0)); bb = nbytes * (5 + (1 if res else 0))
        print(f"C={C:4d} {H}x{W} res={int(res)}: fwd {tf*1e3:7.1f} us {fb/tf/1e6:7.1f} GB/s | bwd {tb*1e3:7.1f} us {bb/tb/1e6:7.1f} GB/s")
