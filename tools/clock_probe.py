"""In-kernel clock of the 2D convolution kernels inside the real training step (diagnostic build: tools/diag_lib.sh clock -DMM_DIAG_CLOCK;
run with MM_LIB_PATH=tools/_bin/libmm2d3d_hip_clock.so MM_GRAPH2D=0).  After ~2 s of steps the per-workgroup (s_memtime, s_memrealtime)
deltas of the LAST launch of k_conv3x3w / k_conv3x3r / k_wgrad3x3n are read back: clock = d(memtime) / d(memrealtime) x 100 MHz."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mm2d3d_amd import _lib
from mm2d3d_amd.synthetic import make_batch
dev = torch.device("cuda:0")
tm = bench.build_trainer(dev)
batch = {"source": make_batch(2, 8, "nuscenes", (302, 480), device=dev, augment=True),
         "target": make_batch(3, 8, "nuscenes", (302, 480), device=dev, augment=True)}
t0 = time.time()
n = 0
while time.time() - t0 < 4.0 or n < 10:
    tm.fit_step(bench.fresh(batch)); n += 1
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
buf = np.zeros((3, 1024, 2), dtype=np.uint64)
rc = L.mm_diag_clock_read_f16(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
print("rc", rc, "steps", n)
for kid, name in enumerate(("k_conv3x3w", "k_conv3x3r", "k_wgrad3x3n")):
    m, r = buf[kid, :, 0].astype(np.float64), buf[kid, :, 1].astype(np.float64)
    ok = r > 0
    if not ok.any():
        print(name, "no data"); continue
    ghz = m[ok] / r[ok] * 0.1
    print(f"{name}: {ok.sum()} workgroups, in-kernel clock median {np.median(ghz):.3f} GHz (p10 {np.percentile(ghz,10):.3f}, p90 {np.percentile(ghz,90):.3f}); "
          f"loop time median {np.median(r[ok]) / 100.0:.1f} us")
