"""Forward time, bytes and FLOPs of every 2D convolution of the net that does NOT run on the 3x3 stride-1 kernels (k_conv_gemm:
strided 3x3, 1x1, transposed; stems) at the bench's shapes - eager forward of the 2D branch, HIP events around each module call,
median of 5.  Shows which of them are memory-bound and how far from the HBM rate they run.   python tools/gemm_layers.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mm2d3d_amd import graph2d, nn2d, domains
from mm2d3d_amd.net2d import Net2DSeg
from mm2d3d_amd.synthetic import make_batch
graph2d.ENABLED[0] = False
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Net2DSeg(6, pretrained=False).to(dev).train()
# the fused pair path of the two backbones bypasses module calls: time the single-encoder form
os.environ["MM_CONV_PAIR"] = "0"
b = make_batch(2, 16, "nuscenes", (302, 480), 6, device=dev, augment=True)
recs = {}
def wrap(name, m):
    orig = m.forward
    def fwd(x, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        y = orig(x, *a, **k)
        e1.record()
        yy = y[0] if isinstance(y, (tuple, list)) else y
        recs.setdefault(name, []).append((e0, e1, tuple(x.shape), tuple(yy.shape), m))
        return y
    m.forward = fwd
for name, m in net.named_modules():
    if isinstance(m, (nn2d.Conv2d, nn2d.ConvTranspose2d)):
        k = m.kernel_size[0]; s = m.stride[0]
        if not (isinstance(m, nn2d.Conv2d) and k == 3 and s == 1):
            wrap(name, m)
with torch.no_grad():
    for _ in range(6):
        net(b)
torch.cuda.synchronize()
print(f"{'module':44s} {'in':>22s} {'out':>22s}  us     GB/s   TF/s")
tot = 0
for name, L in recs.items():
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1, _, _, _ in L[1:])
    t = ts[len(ts) // 2]
    _, _, xi, yo, m = L[-1]
    bytes_ = (torch.tensor(xi).prod().item() + torch.tensor(yo).prod().item()) * 2 + m.weight.numel() * 2
    if isinstance(m, nn2d.ConvTranspose2d):
        fl = 2 * xi[0] * xi[2] * xi[3] * m.in_channels * m.out_channels * m.kernel_size[0] ** 2
    else:
        fl = 2 * yo[0] * yo[2] * yo[3] * m.in_channels * m.out_channels * m.kernel_size[0] ** 2
    tot += t
    print(f"{name:44s} {str(xi):>22s} {str(yo):>22s} {t:6.1f} {bytes_ / t / 1e3:7.0f} {fl / t / 1e6:6.0f}")
print("total forward us:", round(tot, 1))
