"""Does one launch over 2B images beat two launches over B images on the low-resolution 3x3 layers (item-count quantisation of the
persistent kernel)?  Times mm_conv2d_3x3s1 forward on the layer3 / layer4 shapes of the bench step at B = 16 and B = 32."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd import _lib, conv2d as c2d  # noqa: E402
from mm2d3d_amd._lib import check, ptr, stream  # noqa: E402

dev = torch.device("cuda:0")
L = c2d.lib2d()
H16 = c2d.HALF[0]


def run(B, C, H, W, reps=30):
    x = torch.randn(B, H, W, C, device=dev).to(H16)
    y = torch.empty(B, H, W, C, device=dev, dtype=H16)
    wp = (torch.randn(C, 9, C, device=dev) * 0.02).to(H16)
    f = lambda: check(L.mm_conv2d_3x3s1(ptr(x), B, H, W, C, C, ptr(y), C, C, ptr(wp), None, 0, None, 0, stream()), "conv")
    for _ in range(5):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for C, H, W in ((256, 38, 60), (512, 19, 30), (128, 76, 120), (64, 152, 240)):
    t16, t32 = run(16, C, H, W), run(32, C, H, W)
    print(f"{C} ch @ {H}x{W}: B=16 {t16:7.1f} us   B=32 {t32:7.1f} us   two launches of 16 = {2 * t16:7.1f} us   saving {100 * (1 - t32 / (2 * t16)):5.1f} %")


def run_wgrad(B, C, H, W, reps=20):
    x = torch.randn(B, C, H, W, device=dev).to(H16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, C, H, W, device=dev).to(H16).contiguous(memory_format=torch.channels_last)
    dw = torch.zeros(C, C, 3, 3, device=dev)
    ty = [kh - 1 for kh in range(3) for _ in range(3)]
    tx = [kw - 1 for _ in range(3) for kw in range(3)]
    f = lambda: c2d._wgrad(x, B, H, W, C, dy, H, W, C, 1, ty, tx, dw, C * 9, 1, 9, accumulate=1)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("weight gradient (k_wgrad3x3n + k_wgrad_reduce):")
for C, H, W in ((256, 38, 60), (512, 19, 30), (128, 76, 120), (64, 152, 240)):
    t16, t32 = run_wgrad(16, C, H, W), run_wgrad(32, C, H, W)
    print(f"{C} ch @ {H}x{W}: B=16 {t16:7.1f} us   B=32 {t32:7.1f} us   two launches of 16 = {2 * t16:7.1f} us   saving {100 * (1 - t32 / (2 * t16)):5.1f} %")
