"""Output-stationary engine (csrc/osconv.hip) against the rulebook engines (csrc/spconv.hip) on one synthetic batch:
forward and data gradient of the three convolutions, several widths; prints max relative differences and timings."""
import sys
import os

import torch

os.environ.setdefault("MM_OS_MIN_ROWS", "0")
os.environ.setdefault("MM_OS_UP", "1")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd.scn import ops  # noqa: E402
from mm2d3d_amd.scn.metadata import Metadata  # noqa: E402
from mm2d3d_amd.synthetic import make_batch  # noqa: E402


def timeit(fn, n=10):
    """GPU time per call: the launches are queued behind a GPU-side sleep, so host launch cadence does not leak in."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(int(2.0e9 * 0.02))
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    dev = torch.device("cuda:0")
    b = make_batch(2, scenes, "nuscenes", (32, 48), augment=True, device=dev)
    md = Metadata(dev, 4096, 7)
    md.build_levels(b["x"][0].contiguous())
    md.build_rulebooks()
    torch.manual_seed(0)
    worst = 0.0
    for l, lv in enumerate(md.levels[:6]):
        p, p2 = 16 * (l + 1), 16 * (l + 2)
        cases = [("subm", lv.subm, lv.n, lv.n, p, p), ("subm", lv.subm, lv.n, lv.n, 2 * p, p)]
        if lv.down is not None:
            cases += [("down", lv.down, lv.n, lv.coarse.n, p, p2), ("up", lv.down, lv.coarse.n, lv.n, p2, p)]
        for mode, rb, n_in, n_out, cin, cout in cases:
            K = rb.K
            x = torch.randn(n_in, cin, device=dev)
            w = torch.nn.Parameter(torch.randn(K, 1, cin, cout, device=dev) * (2.0 / cin / K) ** 0.5)
            g = torch.randn(n_out, cout, device=dev)
            res = {}
            for os_on in (False, True):
                ops.OS_ENABLED = os_on
                xx = x.clone().requires_grad_(True)
                ww = w
                y = ops.SparseConvFunction.apply(xx, ww, rb, mode, n_in, n_out)
                ww.grad = None
                y.backward(g)
                with torch.no_grad():
                    t_f = timeit(lambda: ops.SparseConvFunction.apply(x, w, rb, mode, n_in, n_out))
                res[os_on] = (y.detach(), xx.grad, ww.grad.clone(), t_f)
            (y0, dx0, dw0, t0), (y1, dx1, dw1, t1) = res[False], res[True]
            ey = float((y0 - y1).abs().max() / y0.abs().max())
            ex = float((dx0 - dx1).abs().max() / dx0.abs().max())
            ew = float((dw0 - dw1).abs().max() / dw0.abs().max())
            worst = max(worst, ey, ex, ew)
            print(f"L{l} {mode:4s} {cin:3d}->{cout:3d} n_in={n_in:7d} n_out={n_out:7d} R={rb.n_rules:8d}  "
                  f"err y {ey:.1e} dx {ex:.1e} dw {ew:.1e}   fwd {t0:7.1f} -> {t1:7.1f} us", flush=True)
    print("worst", worst)
    assert worst < 1e-5, worst


if __name__ == "__main__":
    main()
