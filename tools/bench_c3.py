"""Forward-only timing of the 3x3 halo-tile convolution at the bench's layer shapes (B = 16: the joint pass)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd.conv2d import Conv2dFn
dev = torch.device("cuda:0")
B = 16
SHAPES = [(64, 64, 152, 240), (128, 128, 76, 120), (256, 256, 38, 60), (512, 512, 19, 30), (192, 64, 304, 480)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
for cin, cout, H, W in SHAPES:
    x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    with torch.no_grad():
        t = timeit(lambda: Conv2dFn.apply(x, w, None, 1, 1))
    fl = 2 * B * H * W * cin * cout * 9
    out.append(f"{cin}->{cout}@{H}x{W}: {t*1e3:7.1f} us {fl/t/1e9:6.1f} TF")
print(f"diag={os.environ.get('MM_C3_DIAG','0')}  " + " | ".join(out))
