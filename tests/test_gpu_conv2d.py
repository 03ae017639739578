"""GPU parity of the bf16 MFMA implicit-GEMM convolutions against torch fp32 convolutions of the SAME bf16-rounded
operands (so the only difference is fp32 accumulation order and the final bf16 rounding of the output)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    import mm2d3d_amd  # noqa: F401

    return torch.device("cuda:0")


def _rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("cin,cout,k,s,p,hw", [
    (64, 64, 3, 1, 1, (19, 23)), (64, 128, 3, 2, 1, (20, 28)), (64, 128, 1, 2, 0, (20, 28)), (128, 128, 3, 1, 1, (9, 13)),
    (192, 64, 3, 1, 1, (16, 16)), (256, 512, 3, 2, 1, (10, 14)), (768, 256, 3, 1, 1, (6, 8)), (64, 64, 3, 1, 1, (1, 1)),
])
def test_conv2d_fwd_bwd(cin, cout, k, s, p, hw):
    from mm2d3d_amd.conv2d import Conv2dFn

    dev = _dev()
    torch.manual_seed(cin + cout + k)
    B, (H, W) = 3, hw
    x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5)
    b = torch.randn(cout, device=dev)
    xr = x.float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, s, p)
    xh = x.clone().requires_grad_(True)
    wh = w.clone().requires_grad_(True)
    bh = b.clone().requires_grad_(True)
    yh = Conv2dFn.apply(xh, wh, bh, s, p)
    assert yh.shape == yr.shape and yh.dtype == torch.bfloat16
    assert _rel(yh, yr) < 1e-2  # bf16 output rounding: 2^-8 relative
    g = torch.randn_like(yr).bfloat16()
    yr.backward(g.float())
    yh.backward(g)
    assert _rel(xh.grad, xr.grad) < 1e-2
    assert _rel(wh.grad, wr.grad) < 5e-3  # fp32 out
    assert _rel(bh.grad, br.grad) < 5e-3


@pytest.mark.parametrize("cin,cout,hw", [(1024, 256, (5, 7)), (256, 128, (10, 14)), (64, 64, (12, 20))])
def test_conv_transpose2d_fwd_bwd(cin, cout, hw):
    from mm2d3d_amd.conv2d import ConvTranspose2dFn

    dev = _dev()
    torch.manual_seed(cin + cout)
    B, (H, W) = 2, hw
    x = torch.randn(B, cin, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(cin, cout, 2, 2, device=dev) / cin ** 0.5
    b = torch.randn(cout, device=dev)
    xr = x.float().requires_grad_(True)
    wr = w.bfloat16().float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, 2)
    xh, wh, bh = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yh = ConvTranspose2dFn.apply(xh, wh, bh)
    assert yh.shape == yr.shape
    assert _rel(yh, yr) < 1e-2
    g = torch.randn_like(yr).bfloat16()
    yr.backward(g.float())
    yh.backward(g)
    assert _rel(xh.grad, xr.grad) < 1e-2
    assert _rel(wh.grad, wr.grad) < 5e-3
    assert _rel(bh.grad, br.grad) < 5e-3


@pytest.mark.parametrize("cin", [3, 1])
def test_stem_conv7x7_fwd_and_weight_grad(cin):
    from mm2d3d_amd.conv2d import StemConvFn

    dev = _dev()
    torch.manual_seed(cin)
    B, H, W = 2, 32, 48
    img = torch.rand(B, cin, H, W, device=dev)
    w = torch.randn(64, cin, 7, 7, device=dev) / (cin * 49) ** 0.5
    xr = img.bfloat16().float()
    wr = w.bfloat16().float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 3)
    wh = w.clone().requires_grad_(True)
    yh = StemConvFn.apply(img, wh)
    assert yh.shape == yr.shape and _rel(yh, yr) < 1e-2
    g = torch.randn_like(yr).bfloat16()
    yr.backward(g.float())
    yh.backward(g)
    assert _rel(wh.grad, wr.grad) < 5e-3
