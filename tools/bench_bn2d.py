"""BatchNorm2d (csrc/bn2d.hip) per layer shape of the headline workload: which shapes a training step calls, and how long
the forward / backward entry points take on each, standalone (HIP events, 30 calls after 5 warm-up calls).

    python tools/bench_bn2d.py               # shapes of one step + timing table (both paths)
    python tools/bench_bn2d.py --no-shapes   # the recorded shape list
"""
import argparse
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def step_shapes(dev):
    import bench
    from mm2d3d_amd import nn2d
    from mm2d3d_amd.synthetic import make_batch

    nn2d.BN_PAIR[0] = False  # every layer through _BN2dFn, one problem per call: this tool times the layers singly
    tm = bench.build_trainer(dev)
    batch = {"source": make_batch(2, 8, "nuscenes", (302, 480), 6, device=dev, augment=True),
             "target": make_batch(3, 8, "nuscenes", (302, 480), 6, device=dev, augment=True)}
    tm.fit_step(bench.fresh(batch))
    seen = collections.Counter()
    f0, b0 = nn2d._BN2dFn.forward, nn2d._BN2dFn.backward

    def fwd(ctx, x, res, *a, **k):
        y = f0(ctx, x, res, *a, **k)
        B, C, H, W = x.shape
        ctx._shape = (B * H * W, getattr(ctx, "Ns", B * H * W), C, res is not None, bool(a[7]))
        return y

    def bwd(ctx, dy):
        two = ctx.handoff is not None and bool(ctx.handoff.extra)
        seen[ctx._shape + (two,)] += 1
        return b0(ctx, dy)

    nn2d._BN2dFn.forward, nn2d._BN2dFn.backward = staticmethod(fwd), staticmethod(bwd)
    try:
        tm.fit_step(bench.fresh(batch))
        torch.cuda.synchronize()
    finally:
        nn2d._BN2dFn.forward, nn2d._BN2dFn.backward = f0, b0
    return seen


def time_shape(dev, N, Ns, C, res, relu, two, iters=30):
    from mm2d3d_amd import _lib
    from mm2d3d_amd._lib import check, ptr, stream

    L = _lib.lib()
    bf = torch.bfloat16
    x = torch.randn(N, C, device=dev).to(bf)
    r = torch.randn(N, C, device=dev).to(bf) if res else None
    dy = torch.randn(N, C, device=dev).to(bf)
    dy2 = torch.randn(N, C, device=dev).to(bf) if two else None
    w = torch.rand(C, device=dev) + 0.5
    b = torch.randn(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if res else None
    dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    stats = torch.empty((2, 2 if Ns < N else 1, C), device=dev)
    ws = _lib.workspace.get(int(L.mm_bn2d_ws_bytes(C)), dev)
    ymask = y if (res or not relu) else None

    def f():
        check(L.mm_bn2d_fwd_train(_lib.handle(x.device).h, ptr(x), C, ptr(r), C, N, Ns, C, ptr(w), ptr(b), ptr(rm), ptr(rv), None, 1e-5, 0.1, int(relu), ptr(y), C,
                                  ptr(stats[0]), ptr(stats[1]), ptr(ws), ws.numel(), stream()), "fwd")

    def g():
        check(L.mm_bn2d_bwd(_lib.handle(x.device).h, ptr(x), C, ptr(dy), C, ptr(dy2), C if two else 0, ptr(ymask), C, int(relu), N, Ns, C, ptr(w), ptr(b),
                            ptr(stats[0]), ptr(stats[1]), ptr(dx), C, ptr(dres), C, ptr(dw), ptr(db), 1, ptr(ws), ws.numel(), stream()), "bwd")

    out = []
    for fn in (f, g):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / iters * 1e3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-shapes", action="store_true", help="use the recorded shape list instead of running a step")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    import mm2d3d_amd  # noqa: F401

    if a.no_shapes:
        shapes = collections.Counter({tuple(k): v for k, v in RECORDED})
    else:
        shapes = step_shapes(dev)
    from mm2d3d_amd import _lib

    L = _lib.lib()
    tot = [0.0, 0.0, 0.0, 0.0]
    print("three-kernel path (3k) against the single-launch kernels (1k); GB/s = single-pass traffic / time of the 1k column")
    print(f"{'rows':>9s} {'rows g0':>9s} {'C':>4s} res relu dy2 calls {'MB':>7s} {'fwd 3k':>8s} {'fwd 1k':>8s} {'bwd 3k':>8s} {'bwd 1k':>8s} {'fwd GB/s':>9s} {'bwd GB/s':>9s}")
    for (N, Ns, C, res, relu, two), n in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][2]):
        prev = _lib.bn2d_set_fused(0)
        tf0, tb0 = time_shape(dev, N, Ns, C, res, relu, two)
        _lib.bn2d_set_fused(3)
        tf, tb = time_shape(dev, N, Ns, C, res, relu, two)
        _lib.bn2d_set_fused(prev)
        mb = N * C * 2 / 1e6
        bf = mb * (2 + (1 if res else 0))  # x read once + y written (+ residual): the single-pass floor
        bb = mb * (3 + (1 if two else 0) + (2 if res else 0))  # x, dy (, dy2, yout) read + dx (, dres) written
        for i, t in enumerate((tf0, tf, tb0, tb)):
            tot[i] += t * n
        print(f"{N:9d} {Ns:9d} {C:4d} {int(res):3d} {int(relu):4d} {int(two):3d} {n:5d} {mb:7.1f} {tf0:8.1f} {tf:8.1f} {tb0:8.1f} {tb:8.1f} {bf / tf * 1e3:9.0f} {bb / tb * 1e3:9.0f}")
    print(f"per step, three-kernel: forward {tot[0] / 1e3:.2f} ms + backward {tot[2] / 1e3:.2f} ms = {(tot[0] + tot[2]) / 1e3:.2f} ms")
    print(f"per step, single-launch where the map fits: forward {tot[1] / 1e3:.2f} ms + backward {tot[3] / 1e3:.2f} ms = {(tot[1] + tot[3]) / 1e3:.2f} ms")
    print("RECORDED =", [(list(k), v) for k, v in shapes.items()])


# one headline training step (16 scenes, 480x302), recorded with this script on an MI355X box
RECORDED = [([2334720, 1167360, 64, False, True, False], 1), ([583680, 291840, 64, False, True, False], 8),
            ([145920, 72960, 128, False, True, False], 10), ([36480, 18240, 256, False, True, False], 14),
            ([9120, 4560, 512, True, True, False], 2), ([9120, 4560, 512, False, True, False], 6), ([9120, 4560, 512, True, True, True], 4),
            ([9120, 4560, 512, False, False, False], 2), ([36480, 18240, 256, True, True, False], 2),
            ([36480, 18240, 256, True, True, True], 10), ([36480, 18240, 256, False, False, False], 2),
            ([145920, 72960, 128, True, True, True], 8), ([145920, 72960, 128, False, False, False], 2),
            ([583680, 291840, 64, True, True, True], 6), ([2334720, 1167360, 64, False, True, True], 2)]

if __name__ == "__main__":
    main()
