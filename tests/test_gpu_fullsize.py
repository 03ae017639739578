"""BASELINE.json's full sizes (configs[1]: 8+8 NuScenes-shaped scenes, configs[3]: 4+4 KITTI-shaped scans of 121,600 points),
where the oracle would take minutes: size-independent properties of the voxel hash / rulebooks and of the sparse
convolution engines, checked on the GPU."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    import mm2d3d_amd  # noqa: F401

    return torch.device("cuda:0")


def _metadata(shape, scenes, split=None):
    from mm2d3d_amd import domains
    from mm2d3d_amd.scn.metadata import Metadata
    from mm2d3d_amd.synthetic import make_batch

    dev = _dev()
    batch = make_batch(7, scenes, shape, img_hw=(32, 48), device=dev, augment=True)
    coords = batch["x"][0].contiguous()
    md = Metadata(dev, 4096, 7)
    with domains.split(split):
        md.build_levels(coords)
        md.build_rulebooks()
    return md, coords


def _keys(c):
    c = c.long()
    return ((c[:, 3] * 4096 + c[:, 0]) * 4096 + c[:, 1]) * 4096 + c[:, 2]


@pytest.mark.parametrize("shape,scenes", [("nuscenes", 16), ("kitti", 8)])
def test_metadata_invariants_at_full_size(shape, scenes):
    md, coords = _metadata(shape, scenes, split=scenes // 2)
    lv0 = md.levels[0]
    # (1) active sites = distinct voxels; every point maps to the site that holds its voxel
    assert lv0.n == torch.unique(_keys(coords)).numel()
    assert torch.equal(lv0.coords.long()[lv0.item2vox.long()], coords.long())
    # first-occurrence order: the point index of a site's first item increases with the site id
    first_item = lv0.csr_items[lv0.csr_off[:-1].long()].long()
    assert bool((first_item[1:] > first_item[:-1]).all())
    fine = lv0
    for l, lv in enumerate(md.levels):
        c = lv.coords.long()
        assert lv.n == torch.unique(_keys(c)).numel(), l                      # ids are a bijection onto the voxels
        assert bool((c[1:, 3] >= c[:-1, 3]).all()), l                         # rows batch-sorted (joint BN groups rely on it)
        assert lv.seg_rows == int((c[:, 3] < scenes // 2).sum()), l           # statistics-group boundary
        rb = lv.subm
        off = rb.offsets_host.astype(np.int64)
        R = int(off[27])
        rin, rout = rb.rin[:R].long(), rb.rout[:R].long()
        # (2) submanifold rulebook: centre bucket is the identity on all sites; buckets k and 26-k are mirror images
        assert off[14] - off[13] == lv.n and torch.equal(rin[off[13]:off[14]], rout[off[13]:off[14]])
        assert torch.equal(rout[off[13]:off[14]], torch.arange(lv.n, device=c.device))
        for k in range(13):
            assert off[k + 1] - off[k] == off[27 - k] - off[26 - k], (l, k)
        k = 4
        a = torch.stack([rin[off[k]:off[k + 1]], rout[off[k]:off[k + 1]]], 1)
        b = torch.stack([rout[off[26 - k]:off[27 - k]], rin[off[26 - k]:off[27 - k]]], 1)
        assert torch.equal(a[a[:, 1].argsort()], b[b[:, 1].argsort()]), l      # (k,i,o) <-> (26-k,o,i)
        # every rule joins two sites whose coordinates differ by the bucket's offset, same batch item
        kk = torch.repeat_interleave(torch.arange(27, device=c.device), torch.as_tensor(np.diff(off), device=c.device))
        d = c[rin] - c[rout]
        assert torch.equal(d[:, 0], kk // 9 - 1) and torch.equal(d[:, 1], (kk // 3) % 3 - 1) and torch.equal(d[:, 2], kk % 3 - 1)
        assert bool((d[:, 3] == 0).all())
        assert bool((rout[1:] >= rout[:-1]).logical_or(kk[1:] != kk[:-1]).all())  # buckets sorted by destination
        if lv.coarse is not None:
            cz = lv.coarse.coords.long()
            rb8 = lv.down
            off8 = rb8.offsets_host.astype(np.int64)
            assert int(off8[8]) == lv.n                                        # each fine site has exactly one parent rule
            r8i, r8o = rb8.rin[:lv.n].long(), rb8.rout[:lv.n].long()
            assert torch.unique(r8i).numel() == lv.n
            assert torch.equal(c[r8i][:, :3] >> 1, cz[r8o][:, :3]) and torch.equal(c[r8i][:, 3], cz[r8o][:, 3])
            k8 = torch.repeat_interleave(torch.arange(8, device=c.device), torch.as_tensor(np.diff(off8), device=c.device))
            par = c[r8i][:, :3] & 1
            assert torch.equal(k8, (par[:, 0] * 2 + par[:, 1]) * 2 + par[:, 2])


@pytest.mark.parametrize("cin,cout", [(16, 16), (64, 32), (96, 96)])
def test_sparse_conv_linearity_and_adjoint_at_full_size(cin, cout):
    """conv(a x + b y) = a conv(x) + b conv(y), and <conv(x), g> = <x, conv^T(g)> (forward vs data gradient) on the
    level-0 rulebook of 16 NuScenes-shaped scenes: exact-fp32 engines below 64 input channels, split-bf16 above."""
    from mm2d3d_amd import scn
    from mm2d3d_amd.scn import SparseConvNetTensor

    dev = _dev()
    md, _ = _metadata("nuscenes", 16)
    lv = md.levels[1 if cin > 32 else 0]
    conv = scn.SubmanifoldConvolution(3, cin, cout, 3, False).to(dev)
    g = torch.Generator(device="cpu").manual_seed(cin * 7 + cout)
    x = torch.randn(lv.n, cin, generator=g).to(dev).requires_grad_(True)
    y = torch.randn(lv.n, cin, generator=g).to(dev)
    wrap = lambda f: SparseConvNetTensor(f, md, lv.spatial_size, lv)
    fx, fy = conv(wrap(x)).features, conv(wrap(y)).features
    fz = conv(wrap(2.0 * x.detach() - 0.5 * y)).features
    ref = 2.0 * fx.detach() - 0.5 * fy
    assert float((fz.detach() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    gout = torch.randn(lv.n, cout, generator=g).to(dev)
    (gx,) = torch.autograd.grad(fx, x, gout)
    lhs, rhs = float((fx.detach().double() * gout.double()).sum()), float((x.detach().double() * gx.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * (abs(lhs) + float(fx.detach().double().norm() * gout.double().norm()) * 1e-2), (lhs, rhs)


@pytest.mark.parametrize("level,cin,cout", [(0, 16, 16), (1, 64, 32), (2, 96, 48)])
def test_output_stationary_engine_equals_rulebook_engine_at_full_size(level, cin, cout):
    """At the bench's size (16 NuScenes-shaped scenes) the output-stationary engine (csrc/osconv.hip: rows sorted by
    neighbour mask, offsets accumulated in ascending k in registers) and the k-major rulebook engine + CSR reduce
    (csrc/spconv.hip) must give the same forward and data-gradient rows: bit-identical on the split-product widths (same
    products, same order), <= 2e-6 relative where the rulebook engine uses the exact-fp32 MFMA (below 32 input channels)."""
    from mm2d3d_amd.scn import ops

    dev = _dev()
    md, _ = _metadata("nuscenes", 16)
    lv = md.levels[level]
    assert lv.subm.os is not None, "the level is large enough for the output-stationary table"
    g = torch.Generator(device="cpu").manual_seed(level)
    x = torch.randn(lv.n, cin, generator=g).to(dev)
    w = torch.nn.Parameter((torch.randn(27, 1, cin, cout, generator=g) * (2.0 / cin / 27) ** 0.5).to(dev))
    gout = torch.randn(lv.n, cout, generator=g).to(dev)
    res = []
    try:
        for on in (True, False):
            ops.OS_ENABLED = on
            xx = x.clone().requires_grad_(True)
            y = ops.SparseConvFunction.apply(xx, w, lv.subm, "subm", lv.n, lv.n)
            (gx,) = torch.autograd.grad(y, xx, gout)
            res.append((y.detach(), gx))
    finally:
        ops.OS_ENABLED = True
    (y1, g1), (y0, g0) = res
    if cin >= 32 and cout >= 32:
        assert torch.equal(y1, y0) and torch.equal(g1, g0)
    else:
        assert float((y1 - y0).abs().max()) <= 2e-6 * float(y0.abs().max())
        assert float((g1 - g0).abs().max()) <= 2e-6 * float(g0.abs().max())


@pytest.mark.parametrize("level,cin,cout", [(3, 64, 64), (4, 160, 80), (5, 96, 96)])
def test_rulebook_engine_with_per_step_fragments_equals_per_call_pack(level, cin, cout):
    """The k-major engine takes its weights' three-term fragments either from the per-optimiser-step registry
    (mm_spconv_apply_packed, the layout of mm_spconv_os_pack_batch) or from its own pack launch (mm_spconv_apply): the same
    terms in the same places, so forward and data gradient are bit-identical."""
    from mm2d3d_amd.scn import ops

    dev = _dev()
    md, _ = _metadata("nuscenes", 16)
    lv = md.levels[level]
    assert lv.subm.os is None, "a level the k-major engines serve"
    g = torch.Generator(device="cpu").manual_seed(level)
    x = torch.randn(lv.n, cin, generator=g).to(dev)
    w = torch.nn.Parameter((torch.randn(27, 1, cin, cout, generator=g) * (2.0 / cin / 27) ** 0.5).to(dev))
    gout = torch.randn(lv.n, cout, generator=g).to(dev)
    res = []
    try:
        for on in (True, False):
            ops.OS_ENABLED = on  # off: no registry, the call packs its own fragments
            xx = x.clone().requires_grad_(True)
            y = ops.SparseConvFunction.apply(xx, w, lv.subm, "subm", lv.n, lv.n)
            (gx,) = torch.autograd.grad(y, xx, gout)
            res.append((y.detach(), gx))
    finally:
        ops.OS_ENABLED = True
    (y1, g1), (y0, g0) = res
    assert torch.equal(y1, y0) and torch.equal(g1, g0)
    assert float(y1.abs().max()) > 0


@pytest.mark.parametrize("cin,cout,H,W", [(64, 64, 152, 240), (128, 128, 76, 120), (256, 256, 38, 60), (512, 512, 19, 30), (192, 64, 152, 240)])
def test_conv3x3_at_bench_shapes_vs_torch(cin, cout, H, W, half2d):
    """The persistent 3x3 kernels at the joint-pass shapes of the bench (B = 16): forward and data gradient against torch's
    fp32 convolution on the same bf16-rounded operands (tolerance = bf16 output rounding), weight gradient likewise."""
    import torch.nn.functional as F

    from mm2d3d_amd.conv2d import Conv2dFn

    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin + cout + H)
    B = 16
    x = torch.randn(B, cin, H, W, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev).requires_grad_(True)
    y = Conv2dFn.apply(x, w, None, 1, 1)
    gy = torch.randn(B, cout, H, W, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last)
    gx, gw = torch.autograd.grad(y, [x, w], gy)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().to(half2d).float().requires_grad_(True)  # the kernels multiply the bf16-rounded weights
    yr = F.conv2d(xr, wr, None, 1, 1)
    gxr, gwr = torch.autograd.grad(yr, [xr, wr], gy.float())
    rel = lambda a, b: float((a.detach().float() - b.detach()).norm() / b.detach().norm())
    assert rel(y, yr) < 4e-3 and rel(gx, gxr) < 4e-3 and rel(gw, gwr) < 4e-3, (rel(y, yr), rel(gx, gxr), rel(gw, gwr))
    assert float((y.detach().float() - yr.detach()).abs().max()) <= 2e-2 * float(yr.detach().abs().max())


@pytest.mark.parametrize("split", [False, True])
def test_batchnorm2d_three_kernel_path_at_full_resolution_vs_torch(split, half2d):
    """a non-self comparison of a full-size 2D layer.  The three 299 MB maps of the headline step (16 x 64 x 304 x
    480: stem outputs and the last decoder stage) are too large for the single-launch batch norm and take the reduce / finalize /
    apply kernels; here that path at its own size, forward (+ReLU), running statistics and backward, one and two statistics groups,
    against torch's fp32 BatchNorm2d on the same 16-bit-rounded input."""
    import torch.nn.functional as F

    from mm2d3d_amd import _lib, domains, nn2d

    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(7)
    B, C, H, W = 16, 64, 304, 480
    x = (torch.randn(B, C, H, W, generator=g) * 1.5 + 0.25).to(half2d).to(dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, C, H, W, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last)
    bn = nn2d.BatchNorm2d(C, relu=True).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
    assert not _lib.lib().mm_bn2d_single_launch(_lib.handle(dev).h, B * H * W, B * H * W, C, 0), "this map must take the three-kernel path"
    nf = 8 if split else None
    xh = x.clone().requires_grad_(True)
    with domains.split(nf):
        yh = bn(xh)
    yh.backward(gy)
    groups = [(0, 8), (8, 16)] if split else [(0, 16)]
    dxr, dwr, dbr = [], torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    for b0, b1 in groups:  # what two consecutive calls of the reference's BatchNorm2d do (the two domains)
        xr = x[b0:b1].float().requires_grad_(True)
        ref = torch.nn.BatchNorm2d(C).to(dev)
        with torch.no_grad():
            ref.weight.copy_(bn.weight), ref.bias.copy_(bn.bias), ref.running_mean.copy_(rm), ref.running_var.copy_(rv)
        yr = F.relu(ref(xr))
        err = float((yh[b0:b1].detach().float() - yr.detach()).abs().max() / yr.detach().abs().max())
        assert err < 1e-2, err
        yr.backward(gy[b0:b1].float())
        dxr.append(xr.grad)
        dwr += ref.weight.grad
        dbr += ref.bias.grad
        rm, rv = ref.running_mean.clone(), ref.running_var.clone()
    dxr = torch.cat(dxr)
    rel = lambda a, b: float((a.float() - b).norm() / b.norm())
    assert rel(xh.grad, dxr) < 5e-3, rel(xh.grad, dxr)
    assert rel(bn.weight.grad, dwr) < 2e-3 and rel(bn.bias.grad, dbr) < 2e-3, (rel(bn.weight.grad, dwr), rel(bn.bias.grad, dbr))
    assert torch.allclose(bn.running_mean, rm, atol=1e-4) and torch.allclose(bn.running_var, rv, atol=1e-3)
    assert int(bn.num_batches_tracked) == len(groups)


def test_heads_at_full_resolution_vs_torch(half2d):
    """... and the fused heads (5 x 5 average pool of the crop + two 1 x 1 convolutions, k_head_* / k_box5) on the decoder's map at the
    bench's size, 16 x 64 x 304 x 480 cropped to 302 x 480, forward and backward against torch fp32 on the same 16-bit-rounded map."""
    import torch.nn.functional as F

    from mm2d3d_amd import nn2d

    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(11)
    B, Hp, Wp, h, w = 16, 304, 480, 302, 480
    x = torch.randn(B, 64, Hp, Wp, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last)
    c1, c2 = nn2d.Conv2d(64, 6, 1).to(dev), nn2d.Conv2d(64, 6, 1).to(dev)
    xh = x.clone().requires_grad_(True)
    o1, o2 = nn2d.fused_heads(xh, h, w, c1, c2)
    g1, g2 = torch.randn(o1.shape, generator=g).to(dev), torch.randn(o2.shape, generator=g).to(dev)
    (o1 * g1).sum().add((o2 * g2).sum()).backward()
    # reference on a plain NCHW-contiguous fp32 copy (torch's NHWC avg-pool backward on a cropped view is not trusted), in two halves
    gx = torch.empty(B, 64, h, w, device=dev)
    gw = [torch.zeros_like(c1.weight), torch.zeros_like(c1.bias), torch.zeros_like(c2.weight), torch.zeros_like(c2.bias)]
    for b0 in (0, 8):
        xr = x[b0:b0 + 8, :, :h, :w].float().contiguous().requires_grad_(True)
        pooled = F.avg_pool2d(xr, 5, 1, 2)
        r1, r2 = F.conv2d(pooled, c1.weight, c1.bias), F.conv2d(pooled, c2.weight, c2.bias)
        assert torch.allclose(o1[b0:b0 + 8], r1, atol=2e-4) and torch.allclose(o2[b0:b0 + 8], r2, atol=2e-4)
        got = torch.autograd.grad((r1 * g1[b0:b0 + 8]).sum() + (r2 * g2[b0:b0 + 8]).sum(), [xr, c1.weight, c1.bias, c2.weight, c2.bias])
        gx[b0:b0 + 8] = got[0]
        for a, t in zip(gw, got[1:]):
            a += t
    rel = lambda a, b: float((a.float() - b).norm() / b.norm())
    assert rel(xh.grad[:, :, :h, :w], gx) < 5e-3 and float(xh.grad[:, :, h:, :].float().abs().max()) == 0.0
    assert rel(c1.weight.grad, gw[0]) < 1e-3 and rel(c1.bias.grad, gw[1]) < 1e-3 and rel(c2.weight.grad, gw[2]) < 1e-3 and rel(c2.bias.grad, gw[3]) < 1e-3


@pytest.mark.parametrize("workload", ["c2", "c4"])
def test_full_joint_step_at_bench_size(workload, half2d):
    """One whole two-domain training step at BASELINE.json's sizes - configs[1]: 8 + 8 NuScenes-shaped scenes, 6 classes;
    configs[3]: 4 + 4 KITTI-shaped scans of 121,600 points, 10 classes (datasets/a2d2_semantic_kitti.yaml:19), both at
    480x302.  The jointly batched step (one pass per network over [source | target], per-domain batch-norm statistics)
    must reproduce the six loss terms of the reference's literal two-call sequence (train.py:186-292), and the fp32 3D
    segmentation loss must be bit-stable from run to run (every reduction of the step is order-fixed)."""
    import copy

    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = _dev()
    torch.manual_seed(0)
    shape, ncls, B = ("nuscenes", 6, 8) if workload == "c2" else ("kitti", 10, 4)
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    n2, n3 = Net2DSeg(ncls, pretrained=False).to(dev), Net3DSeg(ncls, True, kw).to(dev)
    for m in n2.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
    src = make_batch(2, B, shape, (302, 480), ncls, device=dev, augment=True)
    trg = make_batch(3, B, shape, (302, 480), ncls, device=dev, augment=True)

    def mk():
        f = lambda b: dict(b, x=[b["x"][0], b["x"][1].clone()])
        return {"source": f(src), "target": f(trg)}

    loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {"weight": [1.0 + 0.1 * i for i in range(ncls)]}}])
    tk = dict(lambda_xm_src=1.0, lambda_xm_trg=0.1)
    one = TrainModel({"2d_net": n2, "3d_net": n3}, None, loss, tk)
    two = TrainModel({"2d_net": n2b, "3d_net": n3b}, None, loss, dict(tk, joint_domains=False))
    t1 = one.training_step(mk())
    logs1 = {k: float(v) for k, v in one.last_logs.items()}
    t1.backward()
    g1 = n3.linear.weight.grad.clone() if hasattr(n3, "linear") else None
    t2 = two.training_step(mk())
    logs2 = {k: float(v) for k, v in two.last_logs.items()}
    assert len(logs1) == 6 and all(np.isfinite(v) for v in logs1.values())
    for k, v in logs2.items():
        # 3D: fp32 end to end; the others see the 16-bit 2D logits (round 5: both formats - IEEE fp16, the default, rounds 8x finer)
        tol = 1e-5 if k.endswith("segmentation_3d") else (3e-3 if half2d == torch.bfloat16 else 1e-3)
        assert abs(logs1[k] - v) <= tol * max(1.0, abs(v)), (k, logs1[k], v)
    # run-to-run: a second joint step from the same weights and inputs (running statistics have moved, the training-mode
    # arithmetic does not read them) gives the same bits for the fp32 3D loss and the same 3D head gradient
    for p in list(n2.parameters()) + list(n3.parameters()):
        p.grad = None
    t3 = one.training_step(mk())
    assert float(one.last_logs["train/loss_segmentation_3d"]) == logs1["train/loss_segmentation_3d"]
    t3.backward()
    if g1 is not None:
        assert torch.equal(n3.linear.weight.grad, g1)


@pytest.mark.parametrize("kind", ["bf16", "fp16"])
def test_full_joint_step_at_bench_size_c5_16bit_rows(kind):
    """BASELINE.json configs[4] at its bench size: 8 source scans downsampled to 10,000 points + 8 KITTI-shaped target scans
    (121,600 points), 480x302, sparse rows stored in 16 bits (`bench.py --workload c5 [--sparse-act fp16]`).  The joint
    [source | target] pass must give the losses of the literal two-call sequence; the step is repeatable bit for bit; three
    optimiser steps on the fixed batch lower the loss and stay finite.  fp16: IEEE half rows ("fp16 activations", the
    reference's ``precision: 16``) under the device-resident loss scale (mm2d3d_amd/amp.py), which must still stand at its
    initial 65536 after the four steps (no overflow on this workload) with every step taken."""
    import copy

    from mm2d3d_amd import scn
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = _dev()
    torch.manual_seed(0)
    scn.set_activation_dtype(torch.bfloat16 if kind == "bf16" else torch.float16)
    try:
        kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
        n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
        for m in n2.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
        src = make_batch(6, 8, "kitti", (302, 480), 6, device=dev, augment=True, downsample=10000)
        trg = make_batch(7, 8, "kitti", (302, 480), 6, device=dev, augment=True)
        assert src["x"][0].shape[0] <= 80000 and trg["x"][0].shape[0] > 900000

        def mk():
            f = lambda b: dict(b, x=[b["x"][0], b["x"][1].clone()])
            return {"source": f(src), "target": f(trg)}

        loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {}}])
        tk = dict(lambda_xm_src=0.1, lambda_xm_trg=0.01, precision=kind)  # the vKITTI experiment's weights (config.yaml:107-108); 2D maps in the same 16-bit kind as the rows
        opts = {}
        for k in ("2d_net", "3d_net"):
            o = Optimizer("adamw", lr=0.001)
            o.set_scheduler("one_cycle", max_lr=0.005, total_steps=1000)
            opts[k] = o
        one = TrainModel({"2d_net": n2, "3d_net": n3}, opts, loss, dict(tk, gc_freeze=False, sparse_activations=kind))
        two = TrainModel({"2d_net": n2b, "3d_net": n3b}, None, loss, dict(tk, joint_domains=False))
        t1 = one.training_step(mk())
        logs1 = {k: float(v) for k, v in one.last_logs.items()}
        t2 = two.training_step(mk())
        logs2 = {k: float(v) for k, v in two.last_logs.items()}
        assert len(logs1) == 6 and all(np.isfinite(v) for v in logs1.values())
        for k, v in logs2.items():
            assert abs(logs1[k] - v) <= 3e-3 * max(1.0, abs(v)), (k, logs1[k], v)  # 16-bit rows / bf16 2D logits on both sides
        t3 = one.training_step(mk())
        assert float(t3) == float(t1), "the 16-bit step is not repeatable"
        del t1, t2, t3, two
        losses = [float(one.fit_step(mk())) for _ in range(4)]
        assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
        if kind == "fp16":
            assert one.scaler is not None and one.scaler.get_scale() == 65536.0
            assert [one.scaler.steps_taken(o) for o in one.optimizers] == [4, 4]
        else:
            assert one.scaler is None
    finally:
        scn.set_activation_dtype(torch.float32)


@pytest.mark.parametrize("level,cin,cout", [
    (2, 48, 48),    # mid level, the k-major gather engines
    (4, 160, 80),   # a decoder layer after the concat: wide rows, three-term split-bf16 products on every engine
    (0, 16, 16),    # level 0: the output-stationary engine on the largest active set of the scan
])
def test_sparse_conv_at_kitti_size_vs_oracle(level, cin, cout):
    """SubmanifoldConvolution cin -> cout on one level of one KITTI-shaped scan (121,600 points): the HIP engines against the
    CPU oracle's rule-book convolution (forward, data gradient, weight gradient) on the scan's own active set of that level -
    "the engines equal each other" (the other tests of this file) tied to "the engines equal the oracle" at full size."""
    from mm2d3d_amd.scn import ops

    dev = _dev()
    from oracle import scn_ref

    md, coords_t = _metadata("kitti", 1)
    lv = md.levels[level]
    coords = coords_t.cpu().numpy()
    _, first = scn_ref.first_occurrence_ids(scn_ref.pack_keys(coords))
    rlv = scn_ref.Level(coords[first], 4096)
    for _ in range(level):
        rlv = scn_ref.down_rulebook(rlv)[1]
    assert rlv.n == lv.n
    rrb = scn_ref.subm_rulebook(rlv)
    assert rrb.n_rules == lv.subm.n_rules
    g = torch.Generator().manual_seed(4 + level)
    x = torch.randn(lv.n, cin, generator=g)
    w = torch.randn(27, 1, cin, cout, generator=g) * (2.0 / cin / 27) ** 0.5
    gout = torch.randn(lv.n, cout, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.reshape(27, cin, cout).clone().requires_grad_(True)
    yr = scn_ref.rule_conv(xr, wr, rrb, lv.n)
    yr.backward(gout)
    xh, wh = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    yh = ops.SparseConvFunction.apply(xh, wh, lv.subm, "subm", lv.n, lv.n)
    yh.backward(gout.to(dev))
    for a, b, what in ((yh, yr, "fwd"), (xh.grad, xr.grad, "dX"), (wh.grad.reshape(27, cin, cout), wr.grad, "dW")):
        err = float((a.detach().cpu() - b.detach()).abs().max())
        assert err <= 1e-3 * max(1.0, float(b.abs().max())), (what, err)
        rel = float((a.detach().cpu().double() - b.detach().double()).norm() / b.detach().double().norm())
        assert rel <= 2e-5, (what, rel)  # fp32-faithful products: the relative L2 distance is rounding noise
