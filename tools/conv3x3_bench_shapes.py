"""The 3x3 stride-1 convolutions of the headline step at the BENCH's shapes (16 images per encoder; layers 2-4 as encoder pairs):
the round-6 kernels k_conv3x3s (flip bits 2-3 = 0, the default) and k_conv3x3v (8) against k_conv3x3w (4), interleaved in one
process (medians of --reps launches each).  Direct C-ABI calls on random 16-bit data.  usage: python tools/conv3x3_bench_shapes.py [--reps 15] [--half bf16]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd import _lib, conv2d as c2  # noqa: E402
from mm2d3d_amd._lib import check, ptr, stream  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=15)
ap.add_argument("--half", default="fp16")
args = ap.parse_args()
dt = torch.float16 if args.half == "fp16" else torch.bfloat16
c2.set_half(dt)
dev = torch.device("cuda:0")
L = c2.lib2d()

# (name, pair, Ca, Cn, H, W, launches per step)
SHAPES = [
    ("layer1 64->64 pair", True, 64, 64, 152, 240, 12),
    ("layer2 128->128 pair", True, 128, 128, 76, 120, 14),
    ("layer3 256->256 pair", True, 256, 256, 38, 60, 22),
    ("layer4 512->512 pair", True, 512, 512, 19, 30, 10),
    ("dec4 768->256", False, 768, 256, 38, 60, 1),
    ("dec4 dgrad 256->768", False, 256, 768, 38, 60, 1),
    ("dec3 384->128", False, 384, 128, 76, 120, 1),
    ("dec3 dgrad 128->384", False, 128, 384, 76, 120, 1),
    ("dec2 192->64", False, 192, 64, 152, 240, 1),
    ("dec2 dgrad 64->192", False, 64, 192, 152, 240, 1),
    ("dec1 192->64", False, 192, 64, 304, 480, 1),
    ("dec1 dgrad 64->192", False, 64, 192, 304, 480, 1),
]
B = 16
FLAGS = (0, 8, 4, 12)  # 12: k_conv3x3s also where the dispatch (flag 0) takes the weights-resident k_conv3x3r (64 -> 64)
tot = {f: 0.0 for f in FLAGS}
for name, pair, Ca, Cn, H, W, per_step in SHAPES:
    xs = [torch.randn(B, H, W, Ca, device=dev).to(dt) for _ in range(2)]
    ws = [(torch.randn(Cn, 9, Ca, device=dev) * (2.0 / (9 * Ca)) ** 0.5).to(dt) for _ in range(2)]
    ys = [torch.empty(B, H, W, Cn, device=dev, dtype=dt) for _ in range(2)]

    def launch(flag):
        if pair:
            check(L.mm_conv2d_3x3s1_pair(ptr(xs[0]), ptr(xs[1]), B, H, W, Ca, Ca, ptr(ys[0]), ptr(ys[1]), Cn, Cn, ptr(ws[0]), ptr(ws[1]), flag, None, None,
                                         B, stream()), "pair")
        else:
            check(L.mm_conv2d_3x3s1(ptr(xs[0]), B, H, W, Ca, Ca, ptr(ys[0]), Cn, Cn, ptr(ws[0]), None, flag, None, B, stream()), "single")

    ts = {f: [] for f in FLAGS}
    for flag in FLAGS:
        launch(flag)
    torch.cuda.synchronize()
    ref = None
    for r in range(args.reps):
        for flag in FLAGS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(flag)
            e1.record()
            torch.cuda.synchronize()
            ts[flag].append(e0.elapsed_time(e1) * 1e3)
            if r == 0:
                out = [y.clone() for y in (ys if pair else ys[:1])]
                if flag == 8:
                    outv = out
                elif flag == 4:
                    same = all(torch.equal(a, b) for a, b in zip(outv, out))
    gf = 2.0 * B * (2 if pair else 1) * H * W * Ca * Cn * 9 / 1e9
    m = {f: statistics.median(ts[f]) for f in ts}
    for f in m:
        tot[f] += m[f] * per_step
    print(f"{name:24s} {H:3d}x{W:3d}  s {m[0]:7.1f} us {gf / m[0] * 1e3:5.0f} TF/s | v {m[8]:7.1f} us {gf / m[8] * 1e3:5.0f} TF/s | w {m[4]:7.1f} us {gf / m[4] * 1e3:5.0f} TF/s | "
          f"s/w {m[0] / m[4]:.3f} v/w {m[8] / m[4]:.3f}  v==w {same} | s forced {m[12]:7.1f} us", flush=True)
print(f"per step (launch counts of the headline step): s {tot[0] / 1e3:.3f} ms, v {tot[8] / 1e3:.3f} ms, w {tot[4] / 1e3:.3f} ms")
