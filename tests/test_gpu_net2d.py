"""GPU parity of the 2D branch's memory-bound kernels and of the whole Net2DSeg against torch / the CPU oracle.

The branch computes in bf16 with fp32 accumulation (the reference runs it under fp16 AMP): operator tests compare with
torch fp32 ops on the SAME bf16-rounded inputs (tolerance = bf16 output rounding, 2^-8); the whole-net test compares
with the fp32 CPU oracle at a bf16-sized tolerance, stated below."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CL = torch.channels_last


def _dev():
    import mm2d3d_amd  # noqa: F401

    return torch.device("cuda:0")


def _rel(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


@pytest.mark.parametrize("C,relu,with_res", [(64, True, False), (128, True, True), (512, False, False), (64, False, True)])
def test_bn2d_train_eval_backward(C, relu, with_res):
    from mm2d3d_amd import nn2d

    dev = _dev()
    torch.manual_seed(C)
    B, H, W = 3, 13, 17
    x = (torch.randn(B, C, H, W, device=dev) * 2 + 0.3).bfloat16().contiguous(memory_format=CL)
    res = torch.randn(B, C, H, W, device=dev).bfloat16().contiguous(memory_format=CL) if with_res else None
    bn = nn2d.BatchNorm2d(C, relu=relu).to(dev)
    ref = torch.nn.BatchNorm2d(C).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    ref.load_state_dict(bn.state_dict())
    xh = x.clone().requires_grad_(True)
    rh = res.clone().requires_grad_(True) if with_res else None
    xr = x.float().requires_grad_(True)
    rr = res.float().requires_grad_(True) if with_res else None
    yh = bn(xh, rh)
    yr = ref(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    assert _rel(yh, yr) < 1e-2
    assert torch.allclose(bn.running_mean, ref.running_mean, atol=1e-4) and torch.allclose(bn.running_var, ref.running_var, atol=1e-3)
    assert int(bn.num_batches_tracked) == 1  # incremented inside the finalize kernel
    g = torch.randn_like(yr).bfloat16()
    # reference gradient with the bf16-rounded forward output deciding the ReLU mask, as the kernel does
    yr.backward(g.float())
    yh.backward(g)
    assert _rel(xh.grad, xr.grad) < 2e-2
    assert _rel(bn.weight.grad, ref.weight.grad) < 1e-2 and _rel(bn.bias.grad, ref.bias.grad) < 1e-2
    if with_res:
        assert _rel(rh.grad, rr.grad) < 1e-2
    bn.eval(), ref.eval()
    ye = bn(x, res)
    yre = ref(x.float())
    if with_res:
        yre = yre + res.float()
    if relu:
        yre = F.relu(yre)
    assert _rel(ye, yre) < 1e-2


def test_maxpool_cat_heads():
    from mm2d3d_amd import nn2d

    dev = _dev()
    torch.manual_seed(0)
    # max-pool: post-ReLU maps have many ties (zeros): the first maximum in scan order must receive the gradient
    x = F.relu(torch.randn(2, 64, 15, 18, device=dev)).bfloat16().contiguous(memory_format=CL)
    xh, xr = x.clone().requires_grad_(True), x.float().requires_grad_(True)
    yh, yr = nn2d.MaxPool2d(3, 2, 1)(xh), F.max_pool2d(xr, 3, 2, 1)
    assert torch.equal(yh.float(), yr)
    g = torch.randn_like(yr).bfloat16()
    yh.backward(g)
    yr.backward(g.float())
    assert _rel(xh.grad, xr.grad) < 1e-2
    # concat
    a = torch.randn(2, 64, 5, 7, device=dev).bfloat16().contiguous(memory_format=CL).requires_grad_(True)
    b = torch.randn(2, 128, 5, 7, device=dev).bfloat16().contiguous(memory_format=CL).requires_grad_(True)
    c = nn2d.cat_channels([a, b, a])
    assert torch.equal(c, torch.cat([a, b, a], 1))
    gg = torch.randn_like(c)
    c.backward(gg)
    assert torch.equal(a.grad.float(), (gg[:, :64].float() + gg[:, 192:].float()).bfloat16().float()) and torch.equal(b.grad, gg[:, 64:192])
    # fused heads == Conv1x1(AvgPool5x5(crop(x))) for both heads
    B, Hp, Wp, h, w = 2, 32, 48, 30, 44
    x = torch.randn(B, 64, Hp, Wp, device=dev).bfloat16().contiguous(memory_format=CL)
    c1, c2 = nn2d.Conv2d(64, 6, 1).to(dev), nn2d.Conv2d(64, 6, 1).to(dev)
    # reference on a plain NCHW-contiguous fp32 copy (torch's NHWC avg-pool backward on a cropped view is not trusted)
    xh, xr = x.clone().requires_grad_(True), x.float().contiguous(memory_format=torch.contiguous_format).requires_grad_(True)
    o1, o2 = nn2d.fused_heads(xh, h, w, c1, c2)
    pooled = F.avg_pool2d(xr[:, :, :h, :w], 5, 1, 2)
    r1, r2 = F.conv2d(pooled, c1.weight, c1.bias), F.conv2d(pooled, c2.weight, c2.bias)
    assert torch.allclose(o1, r1, atol=1e-4) and torch.allclose(o2, r2, atol=1e-4)
    g1, g2 = torch.randn_like(r1), torch.randn_like(r2)
    (o1 * g1).sum().add((o2 * g2).sum()).backward()
    gw = torch.autograd.grad((r1 * g1).sum() + (r2 * g2).sum(), [xr, c1.weight, c1.bias, c2.weight, c2.bias])
    assert _rel(xh.grad, gw[0]) < 1e-2 and torch.equal(xh.grad[:, :, h:, :].float(), torch.zeros_like(xh.grad[:, :, h:, :]).float())
    assert _rel(c1.weight.grad, gw[1]) < 1e-3 and _rel(c1.bias.grad, gw[2]) < 1e-3
    assert _rel(c2.weight.grad, gw[3]) < 1e-3 and _rel(c2.bias.grad, gw[4]) < 1e-3


@pytest.mark.parametrize("training", [False, True])
def test_net2d_vs_oracle(training):
    from mm2d3d_amd.net2d import Net2DSeg
    from oracle.net2d_ref import net2d_forward

    dev = _dev()
    torch.manual_seed(0)
    net = Net2DSeg(6, pretrained=False)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net.train(training)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    g = np.random.default_rng(1)
    B, H, W = 2, 46, 62  # padded to 48 x 64
    idx = [np.stack([g.integers(0, H, 300), g.integers(0, W, 300)], 1) for _ in range(B)]
    img, depth = torch.rand(B, 3, H, W), torch.rand(B, 1, H, W)
    so = {}
    pr, last_r, _, ar = net2d_forward(sd, {"img": img, "depth": depth, "img_indices": idx}, training=training, stats_out=so)
    net.to(dev)
    ph, last_h, _, ah = net({"img": img.to(dev), "depth": depth.to(dev), "img_indices": idx})
    assert last_h.shape == last_r.shape
    # bf16 activations through ~40 layers against the fp32 oracle; bounds = 2.5x what was measured on MI355X
    # (eval: logits 1.9e-3, decoder map 4.8e-3; batch statistics of this tiny 2 x 46 x 62 batch: logits 1.2e-2, map 1.6e-2)
    t_logit, t_map = (3e-2, 4e-2) if training else (5e-3, 1.2e-2)
    for a, b, what, tol in ((ph["seg_logit"], pr["seg_logit"], "seg_logit", t_logit), (ah["seg_logit_avg"], ar["seg_logit_avg"], "seg_logit_avg", t_logit),
                            (ph["seg_logit_2d"], pr["seg_logit_2d"], "seg_logit_2d", t_map), (last_h, last_r, "segm_last", t_map)):
        assert _rel(a.cpu(), b) < tol, (what, _rel(a.cpu(), b))
    # the same forward against the oracle that rounds to bf16 where the HIP branch stores bf16: what is left is accumulation
    # order and the isolated 1-ulp bf16 flips it causes -> an order of magnitude tighter than the fp32 comparison above
    so2 = {}
    pq, last_q, _, aq = net2d_forward(sd, {"img": img, "depth": depth, "img_indices": idx}, training=training, stats_out=so2,
                                      emulate_bf16=True)
    # measured: eval 1.6e-4 (logits) / 1.9e-3 (decoder map), train (batch statistics of a 2 x 46 x 62 batch) 2.3e-3 / 7.3e-3
    tol_logit, tol_map = (8e-3, 2.5e-2) if training else (1e-3, 6e-3)
    for a, b, what, tol in ((ph["seg_logit"], pq["seg_logit"], "seg_logit", tol_logit),
                            (ah["seg_logit_avg"], aq["seg_logit_avg"], "seg_logit_avg", tol_logit), (last_h, last_q, "segm_last", tol_map)):
        assert _rel(a.cpu(), b) < tol, ("bf16-emulating oracle", what, _rel(a.cpu(), b))
    if training:
        for pre, (rm, rv) in so.items():
            assert torch.allclose(net.state_dict()[pre + ".running_mean"].cpu(), rm, atol=2e-2, rtol=5e-2), pre
            assert torch.allclose(net.state_dict()[pre + ".running_var"].cpu(), rv, atol=2e-2, rtol=5e-2), pre
