"""Diagnostic (round 5): the whole training step (2D + 3D, forward + backward, one stream) with tests/helpers/squatter.hip hammering
the LDS of every CU from a second stream, against the same step alone.  Any kernel whose result depends on what else is
resident on its CU shows up as a loss / gradient that differs.  Usage: python tools/corun_net.py [mode] [lds_bytes] [grid]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.getcwd())
from mm2d3d_amd import graph2d
from mm2d3d_amd.losses import Loss
from mm2d3d_amd.net2d import Net2DSeg
from mm2d3d_amd.net3d import Net3DSeg
from mm2d3d_amd.synthetic import make_batch
from mm2d3d_amd.train import TrainModel
graph2d.ENABLED[0] = False
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lds = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
grid = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ticks = int(sys.argv[4]) if len(sys.argv) > 4 else 1000000  # of the 100 MHz clock, per squatter launch
nsquat = int(sys.argv[5]) if len(sys.argv) > 5 else 8
dev = torch.device("cuda:0")
sq = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libsquat.so"))
sq.squat.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
torch.manual_seed(0)
kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
for m in n2.modules():
    if isinstance(m, torch.nn.Dropout):
        m.p = 0.0
src = make_batch(6, 8, "nuscenes", (302, 480), 6, device=dev, augment=True)
trg = make_batch(7, 8, "nuscenes", (302, 480), 6, device=dev, augment=True)
loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {}}])
tm = TrainModel({"2d_net": n2, "3d_net": n3}, None, loss, dict(lambda_xm_src=0.1, lambda_xm_trg=0.01, precision="fp16", overlap_branches=0, overlap_rulebooks=0, gc_freeze=False))
buf = torch.randn(16 << 20, device=dev)
main, side = torch.cuda.current_stream(), torch.cuda.Stream(dev)
params = [(n, p) for net, pre in ((n2, "2d."), (n3, "3d.")) for n, p in ((pre + k, v) for k, v in net.named_parameters())]
def step(squat):
    for _, p in params:
        p.grad = None
    batch = {"source": dict(src, x=[src["x"][0], src["x"][1].clone()]), "target": dict(trg, x=[trg["x"][0], trg["x"][1].clone()])}
    if squat:
        side.wait_stream(main)
        for _ in range(nsquat):  # default 8 x 10 ms of squatting: longer than the step
            assert sq.squat(grid, lds, mode, ticks, buf.data_ptr(), buf.numel(), side.cuda_stream) == 0
    l = tm.training_step(batch)
    l.backward()
    main.wait_stream(side)
    torch.cuda.synchronize()
    return float(l), {k: float(v) for k, v in tm.last_logs.items()}, {n: p.grad.clone() for n, p in params if p.grad is not None}
for _ in range(2):
    ref = step(False)
again = step(False)
print("alone vs alone: loss equal", ref[0] == again[0], "grads differing", [n for n in ref[2] if not torch.equal(ref[2][n], again[2][n])][:5])
for it in range(3):
    got = step(True)
    bad = [n for n in ref[2] if not torch.equal(ref[2][n], got[2][n])]
    lbad = [k for k in ref[1] if ref[1][k] != got[1][k]]
    print(f"squatters (mode {mode}, {lds} B LDS, grid {grid}) run {it}: loss equal {ref[0] == got[0]}; loss terms differing {lbad}; {len(bad)} of {len(ref[2])} gradients differ")
    def rel(n):
        a, b = ref[2][n].float(), got[2][n].float()
        return float((a - b).norm() / (a.norm() + 1e-30))
    for n in bad[:400]:
        if n.endswith("weight") and ("conv" in n or "downsample.0" in n or n.startswith("3d")):
            print("     ", n, tuple(ref[2][n].shape), "rel", f"{rel(n):.2e}")
got = step(False)
print("alone again: loss equal", ref[0] == got[0], "grads differing", len([n for n in ref[2] if not torch.equal(ref[2][n], got[2][n])]))
