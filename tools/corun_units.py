"""Diagnostic (round 5): which kernels of the 2D branch change their results when small LDS-using workgroups of ANOTHER stream are
launched onto their CUs while they run (tests/helpers/squatter.hip, mode 1, many short launches)?  One layer at a time, forward + backward,
against the same layer alone.  Usage: python tools/corun_units.py [mode] [lds_bytes] [ticks] [nsquat]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.getcwd())
from mm2d3d_amd import nn2d
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lds = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
nsquat = int(sys.argv[4]) if len(sys.argv) > 4 else 60
dev = torch.device("cuda:0")
sq = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libsquat.so")  # hipcc -shared -fPIC tests/helpers/squatter.hip)
sq.squat.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
buf = torch.randn(16 << 20, device=dev)
main, side = torch.cuda.current_stream(), torch.cuda.Stream(dev)
torch.manual_seed(0)
C3 = lambda ci, co, **k: nn2d.Conv2d(ci, co, kernel_size=3, padding=1, bias=False, **k)
units = [
    ("conv3x3 64->64 @152x240 (k_conv3x3r, k_wgrad3x3n)", lambda: C3(64, 64), (16, 64, 152, 240)),
    ("conv3x3 192->64 @304x480 (k_conv3x3w<64>)", lambda: nn2d.Conv2d(192, 64, kernel_size=3, padding=1), (16, 192, 304, 480)),
    ("conv3x3 384->128 @76x120 (k_conv3x3w<128>)", lambda: C3(384, 128), (16, 384, 76, 120)),
    ("conv3x3 128->128 @76x120", lambda: C3(128, 128), (32, 128, 76, 120)),
    ("conv3x3 256->256 @38x60", lambda: C3(256, 256), (32, 256, 38, 60)),
    ("conv3x3 512->512 @19x30", lambda: C3(512, 512), (32, 512, 19, 30)),
    ("conv1x1 s2 64->128 @152x240 (k_conv_gemm)", lambda: nn2d.Conv2d(64, 128, kernel_size=1, stride=2, bias=False), (16, 64, 152, 240)),
    ("conv3x3 s2 64->128 @152x240 (k_conv_gemm)", lambda: nn2d.Conv2d(64, 128, kernel_size=3, stride=2, padding=1, bias=False), (16, 64, 152, 240)),
    ("convT 2x2 s2 128->64 @76x120", lambda: nn2d.ConvTranspose2d(128, 64, kernel_size=2, stride=2), (16, 128, 76, 120)),
    ("bn 64 @152x240", lambda: nn2d.BatchNorm2d(64), (16, 64, 152, 240)),
    ("bn 128 @76x120", lambda: nn2d.BatchNorm2d(128), (16, 128, 76, 120)),
    ("bn 64 @304x480", lambda: nn2d.BatchNorm2d(64), (16, 64, 304, 480)),
    ("bn 512 @19x30", lambda: nn2d.BatchNorm2d(512), (32, 512, 19, 30)),
]
def run(m, x, g, squat):
    x.grad = None
    for p in m.parameters():
        p.grad = None
    if squat:
        side.wait_stream(main)
        for _ in range(nsquat):
            assert sq.squat(1024, lds, mode, ticks, buf.data_ptr(), buf.numel(), side.cuda_stream) == 0
    y = m(x)
    (y.float() * g).sum().backward()
    main.wait_stream(side)
    torch.cuda.synchronize()
    out = {"y": y.detach().clone(), "dx": x.grad.clone()}
    for n, p in m.named_parameters():
        if p.grad is not None:
            out["d" + n] = p.grad.clone()
    return out
for name, make, shape in units:
    m = make().to(dev).train()
    x = torch.randn(*shape, device=dev).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    with torch.no_grad():
        yshape = m(x).shape
    g = torch.randn(*yshape, device=dev).contiguous(memory_format=torch.channels_last)
    ref = run(m, x, g, False)
    again = run(m, x, g, False)
    alone_ok = all(torch.equal(ref[k], again[k]) for k in ref)
    res = []
    for it in range(3):
        got = run(m, x, g, True)
        res.append({k: int((ref[k] != got[k]).sum()) for k in ref if not torch.equal(ref[k], got[k])})
    print(f"{name:58s} alone repeatable {alone_ok} | with squatters: {res}")
