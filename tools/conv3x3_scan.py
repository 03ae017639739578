"""3x3 stride-1 convolution forward (k_conv3x3w / k_conv3x3r) at large batch, where the items balance over the workgroups:
time per call and TFLOP/s (host-loop event timing; use for A/B of kernel builds on one box)."""
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
for cin, cout, H, W, B in ((256, 256, 38, 60, 128), (128, 128, 76, 120, 64), (512, 512, 19, 30, 128), (192, 64, 152, 240, 16), (64, 64, 152, 240, 32),
                           (256, 256, 38, 60, 16), (128, 128, 76, 120, 16), (512, 512, 19, 30, 16)):
    x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    with torch.no_grad():
        t = timeit(lambda: Conv2dFn.apply(x, w, None, 1, 1))
    print(f"{cin}->{cout} @{H}x{W} B={B}: {t:7.1f} us  {2*B*H*W*cin*cout*9/t/1e6:7.0f} TF/s", flush=True)
