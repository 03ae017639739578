"""Per-shape timing of the 2D convolution kernels (forward, data gradient, weight gradient) at the bench's layer shapes.

CAUTION: host-loop event timing of back-to-back identical launches; it over-stated the non-persistent 3x3 kernel by 1.5x
against its in-situ duration (rocprofv3 trace of bench.py).  Use the kernel trace of a real step for decisions."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd.conv2d import Conv2dFn

dev = torch.device("cuda:0")
B = 8
SHAPES = [  # (cin, cout, H, W, count per net forward)  3x3 s1 convs of one forward over both backbones + decoder
    (64, 64, 152, 240, 12), (128, 128, 76, 120, 14), (256, 256, 38, 60, 22), (512, 512, 19, 30, 10),
    (768, 256, 38, 60, 1), (384, 128, 76, 120, 1), (192, 64, 152, 240, 1), (192, 64, 304, 480, 1),
]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = [0.0, 0.0, 0.0]
for cin, cout, H, W, cnt in SHAPES:
    x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).requires_grad_(True)
    y = Conv2dFn.apply(x, w, None, 1, 1)
    g = torch.randn_like(y)
    fl = 2 * B * H * W * cin * cout * 9
    tf = timeit(lambda: Conv2dFn.apply(x, w, None, 1, 1))
    def bwd_x():
        torch.autograd.grad(y, x, g, retain_graph=True)
    def bwd_w():
        torch.autograd.grad(y, w, g, retain_graph=True)
    tx, tw = timeit(bwd_x), timeit(bwd_w)
    print(f"{cin:4d}->{cout:4d} @{H}x{W}: fwd {tf*1e3:7.1f} us {fl/tf/1e9:6.1f} TF | dgrad {tx*1e3:7.1f} us {fl/tx/1e9:6.1f} TF | wgrad {tw*1e3:7.1f} us {fl/tw/1e9:6.1f} TF")
    for i, t in enumerate((tf, tx, tw)): tot[i] += t * cnt
print(f"weighted per net-forward (B=8): fwd {tot[0]:.2f} ms, dgrad {tot[1]:.2f} ms, wgrad {tot[2]:.2f} ms  -> x2 per step")
