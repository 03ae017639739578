"""Diagnostic (round 5): does a CO-RESIDENT foreign workgroup change what the LDS-DMA convolution kernels compute?
The last decoder convolution (192 -> 64 at 304 x 480, k_conv3x3w<64, 16>: 96 VGPRs, 139 KB of LDS - room for another kernel's
workgroups on the CU) runs on the main stream while tests/helpers/squatter.hip occupies every CU from a second stream; every output is
compared with the kernel's own output when it runs alone.  Usage: python tools/conv_corun.py  (needs tools/_bin/libsquat.so)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.getcwd())
from mm2d3d_amd import nn2d, _lib
dev = torch.device("cuda:0")
sq = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libsquat.so"))
sq.squat.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
torch.manual_seed(0)
shapes = [(16, 192, 64, 304, 480), (16, 64, 64, 152, 240), (32, 128, 128, 76, 120), (32, 256, 256, 38, 60)]
if len(sys.argv) > 1:
    shapes = shapes[: int(sys.argv[1])]
buf = torch.randn(64 << 20, device=dev)
main, side = torch.cuda.current_stream(), torch.cuda.Stream(dev)
for (B, Ci, Co, H, W) in shapes:
    conv = nn2d.Conv2d(Ci, Co, kernel_size=3, padding=1).to(dev)
    x = torch.randn(B, Ci, H, W, device=dev).half().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = conv(x).clone()
        torch.cuda.synchronize()
        print(f"--- conv {Ci}->{Co} @ {H}x{W} B={B}")
        for name, grid, lds, mode in (("none", 0, 0, 0), ("sleep, no LDS", 1024, 0, 0), ("sleep, 4 KB LDS", 1024, 4096, 0), ("sleep, 16 KB LDS", 512, 16384, 0),
                                      ("LDS hammer 1 KB", 1024, 1024, 1), ("global stream, no LDS", 2048, 0, 2), ("global stream, 4 KB LDS", 1024, 4096, 2), ("none", 0, 0, 0)):
            outs = []
            for rep in range(6):
                if grid:
                    side.wait_stream(main)
                    rc = sq.squat(grid, lds, mode, 60000, buf.data_ptr(), buf.numel(), side.cuda_stream)  # 600 us of the 100 MHz clock
                    assert rc == 0, rc
                for _ in range(3):
                    outs.append(conv(x))
                main.wait_stream(side)
            torch.cuda.synchronize()
            bad = [(i, int((o != ref).sum())) for i, o in enumerate(outs) if not torch.equal(o, ref)]
            print(f"  {name:26s} bad launches {len(bad):2d}/{len(outs)}", bad[:6])
