"""configs[4] weight gradient over 16-bit rows: k_dw_tr16 (rows through LDS + transpose reads, round 6) against k_dw_direct_s3 (element gathers;
mode | 8) on random rulebooks of the bench's sizes: time, bit identity of the slab sums, error against fp64.  usage: python tools/dw16_ab.py"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from mm2d3d_amd import _lib
from mm2d3d_amd._lib import check, ptr, stream
L = _lib.lib(); dev = torch.device("cuda:0")
torch.manual_seed(0)
def run(K, R, n_rows, cin, cout, dtype, mode, reps=1):
    g = torch.Generator().manual_seed(K * 1000 + cin + cout)
    per = [R // K + (1 if i < R % K else 0) for i in range(K)]
    off = np.zeros(K + 1, dtype=np.int32); off[1:] = np.cumsum(per)
    src = torch.randint(0, n_rows, (R,), generator=g, dtype=torch.int32).to(dev)
    dst = torch.randint(0, n_rows, (R,), generator=g, dtype=torch.int32).to(dev)
    x = torch.randn(n_rows, cin, generator=g).to(dtype).to(dev); d = torch.randn(n_rows, cout, generator=g).to(dtype).to(dev)
    dW = torch.zeros(K, cin, cout, device=dev)
    ws = torch.empty(int(L.mm_spconv_dw_ws_bytes(off.ctypes.data, K, cin, cout)), dtype=torch.uint8, device=dev)
    fn = L.mm_spconv_dw_f16 if dtype == torch.float16 else L.mm_spconv_dw_bf16
    def call():
        check(fn(ptr(x), cin, cin, ptr(d), cout, cout, ptr(src), ptr(dst), off.ctypes.data, K, ptr(dW), 0, mode, ptr(ws), ws.numel(), stream()), "dw")
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    ref = None
    return dW.clone(), e0.elapsed_time(e1) / reps * 1e3, (x, d, src, dst, off)
for (K, R, n, ci, co) in [(27, 3596049, 1050000, 16, 16), (27, 2968094, 700000, 32, 32), (27, 1912070, 400000, 96, 48), (27, 946950, 200000, 128, 64),
                          (8, 528960, 400000, 32, 48), (27, 396247, 80000, 160, 80), (27, 169573, 30000, 192, 96), (27, 72246, 8000, 112, 112), (8, 20663, 8000, 96, 112), (27, 1000, 300, 48, 80), (27, 33, 20, 16, 32)]:
    for dt in (torch.float16, torch.bfloat16):
        a, ta, ops = run(K, R, n, ci, co, dt, 8, 5)
        b, tb, _ = run(K, R, n, ci, co, dt, 0, 5)
        x, d, src, dst, off = ops
        same = torch.equal(a, b)
        # fp64 reference of one offset
        k = K // 2; sl = slice(int(off[k]), int(off[k + 1]))
        ref = x[src[sl].long()].double().t() @ d[dst[sl].long()].double()
        err = float((b[k].double() - ref).abs().max() / ref.abs().max().clamp_min(1e-9))
        print(f"K={K} R={R} {ci}->{co} {str(dt)[6:]}: elem {ta:7.1f} us  tr16 {tb:7.1f} us  ratio {tb/ta:.2f}  identical={same} rel.err vs fp64 {err:.2e}", flush=True)
