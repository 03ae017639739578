"""How do the persistent 3x3 convolution kernels behave while another kernel holds LDS on some CUs (as an RCCL collective
does during backward at N > 1)?  Times the conv layers with and without a 64-block LDS hog on a side stream."""
import ctypes, os, subprocess, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd.conv2d import Conv2dFn

here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/libhog.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-shared", "-fPIC", "--offload-arch=gfx950", os.path.join(here, "hog.hip"), "-o", so])
hog = ctypes.CDLL(so)
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
sink = torch.zeros(1, dtype=torch.int32, device=dev)
side = torch.cuda.Stream(dev)
B = 16
for cin, cout, H, W in [(64, 64, 152, 240), (128, 128, 76, 120), (256, 256, 38, 60), (512, 512, 19, 30)]:
    x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    with torch.no_grad():
        for _ in range(3):
            Conv2dFn.apply(x, w, None, 1, 1)
        res = []
        for blocks, lds in ((0, 0), (64, 65536), (64, 16384)):
            torch.cuda.synchronize()
            if blocks:
                hog.hog_launch(blocks, lds, 100_000_000 * 3 // 100, sink.data_ptr(), side.cuda_stream)  # ~30 ms at 100 MHz wall clock
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                Conv2dFn.apply(x, w, None, 1, 1)
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{cin}->{cout}@{H}x{W}: alone {res[0]:7.1f} us | with 64 x 64 KB hog {res[1]:7.1f} us | with 64 x 16 KB hog {res[2]:7.1f} us", flush=True)
