"""Co-residency of the LDS-DMA kernels of the 2D branch with foreign LDS-using workgroups.

Round 5 found that ``k_conv3x3w<64, *>`` computed wrong tiles whenever small LDS-using workgroups of ANOTHER stream were launched onto
its CU while its LDS-DMA ring was in flight; the cure was structural (every one-workgroup-per-CU kernel claims the whole LDS and
register file of its CU), the hardware mechanism was never pinned down.  The kernels that share their CU by design - ``k_conv_gemm``,
``k_conv_wgrad2``, ``k_stem7``, ``k_stem_wgrad``, the batch norms - and the owner kernels (``k_conv3x3s / v / r``, ``k_wgrad3x3n``) are
therefore all run here beside the trigger of that defect: hundreds of short launches of a 64-thread kernel that hammers 1 KB of LDS
(tests/helpers/squatter.hip, compiled on the spot), forward + backward, every output compared bit for bit with the same layer alone.
RCCL's kernels are such squatters: this is the N > 1 parity risk of a data-parallel run, tested on one GPU."""
import ctypes
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
_HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def squat():
    src = os.path.join(_HERE, "helpers", "squatter.hip")
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"libsquat_{os.getuid()}.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", out], check=True)
    lib = ctypes.CDLL(out)
    lib.squat.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_void_p]
    return lib


def _units():
    from mm2d3d_amd import nn2d

    C3 = lambda ci, co, **k: nn2d.Conv2d(ci, co, kernel_size=3, padding=1, bias=False, **k)
    return [
        ("conv3x3 64->64 (k_conv3x3s with resident weights, k_wgrad3x3n)", lambda: C3(64, 64), (16, 64, 152, 240), False),
        ("conv3x3 192->64 (k_conv3x3s<64>)", lambda: nn2d.Conv2d(192, 64, kernel_size=3, padding=1), (8, 192, 152, 240), False),
        ("conv3x3 384->128 (k_conv3x3s<128,16>)", lambda: C3(384, 128), (16, 384, 76, 120), False),
        ("conv3x3 256->256 (k_conv3x3s<128,32>, half items)", lambda: C3(256, 256), (32, 256, 38, 60), False),
        ("conv1x1 s2 64->128 (k_conv_gemm<128>, k_conv_wgrad2)", lambda: nn2d.Conv2d(64, 128, kernel_size=1, stride=2, bias=False), (16, 64, 152, 240), False),
        ("conv3x3 s2 64->128 (k_conv_gemm, dgrad by parity)", lambda: nn2d.Conv2d(64, 128, kernel_size=3, stride=2, padding=1, bias=False), (16, 64, 152, 240), False),
        ("convT 2x2 s2 128->64 (k_conv_gemm<64> zpar)", lambda: nn2d.ConvTranspose2d(128, 64, kernel_size=2, stride=2), (16, 128, 76, 120), False),
        ("stem 7x7 3->64 (k_stem7<4>, k_stem_wgrad<4>)", lambda: nn2d.Conv2d(3, 64, kernel_size=7, stride=1, padding=3, bias=False), (8, 3, 152, 240), True),
        ("stem 7x7 1->64 (k_stem7<1>, k_stem_wgrad<1>)", lambda: nn2d.Conv2d(1, 64, kernel_size=7, stride=1, padding=3, bias=False), (8, 1, 152, 240), True),
        ("BatchNorm2d 128 @76x120 (single launch, two grid barriers)", lambda: nn2d.BatchNorm2d(128), (16, 128, 76, 120), False),
        ("BatchNorm2d 64 @304x480 (three kernels)", lambda: nn2d.BatchNorm2d(64), (8, 64, 304, 480), False),
    ]


@pytest.mark.parametrize("idx", range(11))
@pytest.mark.parametrize("legacy", [0, 4])
def test_results_do_not_depend_on_foreign_lds_workgroups(idx, legacy, squat):
    from mm2d3d_amd import conv2d as c2

    name, make, shape, image = _units()[idx]
    if legacy and "conv3x3" not in name or (legacy and "64->64" in name):
        pytest.skip("the kernel choice only concerns the general 3x3 stride-1 kernels")
    dev = torch.device("cuda:0")
    torch.manual_seed(idx)
    c2.LEGACY3X3[0] = legacy  # 4: k_conv3x3w, the kernel that had the defect (it owns its CU since round 5)
    try:
        m = make().to(dev).train()
        if image:
            x = torch.rand(*shape, device=dev)
        else:
            x = torch.randn(*shape, device=dev).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        with torch.no_grad():
            yshape = m(x).shape
        g = torch.randn(*yshape, device=dev).contiguous(memory_format=torch.channels_last)
        buf = torch.randn(1 << 20, device=dev)
        main, side = torch.cuda.current_stream(), torch.cuda.Stream(dev)

        def run(with_squatters):
            if x.requires_grad:
                x.grad = None
            for p in m.parameters():
                p.grad = None
            if with_squatters:
                side.wait_stream(main)
                for _ in range(60):  # 1024 workgroups x 64 threads, 1 KB of LDS each, hammering it for 100 us
                    assert squat.squat(1024, 1024, 1, 10000, buf.data_ptr(), buf.numel(), side.cuda_stream) == 0
            y = m(x)
            (y.float() * g).sum().backward()
            main.wait_stream(side)
            torch.cuda.synchronize()
            out = {"y": y.detach().clone()}
            if x.requires_grad:
                out["dx"] = x.grad.clone()
            for n, p in m.named_parameters():
                if p.grad is not None:
                    out["d" + n] = p.grad.clone()
            return out

        ref = run(False)
        again = run(False)
        assert all(torch.equal(ref[k], again[k]) for k in ref), f"{name}: not repeatable alone"
        for it in range(3):
            got = run(True)
            bad = {k: int((ref[k] != got[k]).sum()) for k in ref if not torch.equal(ref[k], got[k])}
            assert not bad, f"{name}: outputs changed beside the squatters (differing elements: {bad})"
    finally:
        c2.LEGACY3X3[0] = 0
