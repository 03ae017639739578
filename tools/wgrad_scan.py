"""Weight gradient of a 3x3 convolution against the number of images: time = fixed part + patches x per-patch cost.
(host-loop event timing of back-to-back launches, k_wgrad3x3n + k_wgrad_reduce; use for the slope, not for absolute numbers)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd.conv2d import Conv2dFn

dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cin, cout, H, W in ((256, 256, 38, 60), (64, 64, 152, 240), (512, 512, 19, 30)):
    for B in (1, 2, 4, 8, 16, 32):
        x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).requires_grad_(True)
        y = Conv2dFn.apply(x, w, None, 1, 1)
        g = torch.randn_like(y)
        t = timeit(lambda: torch.autograd.grad(y, w, g, retain_graph=True))
        npatch = B * ((H + 7) // 8) * ((W + 15) // 16)
        print(f"{cin}->{cout} @{H}x{W} B={B:2d} patches={npatch:5d}: {t:7.1f} us   {2*B*H*W*cin*cout*9/t/1e6:7.1f} TF/s")
