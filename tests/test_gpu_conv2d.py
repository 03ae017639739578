"""GPU parity of the 16-bit MFMA implicit-GEMM convolutions against torch fp32 convolutions of the SAME 16-bit-rounded
operands (so the only difference is fp32 accumulation order and the final 16-bit rounding of the output).  Every test runs for
both builds of the kernels: IEEE fp16 maps (default, the reference's precision: 16) and bfloat16 maps (``half2d``)."""
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
def test_conv2d_fwd_bwd(cin, cout, k, s, p, hw, half2d):
    from mm2d3d_amd.conv2d import Conv2dFn

    dev = _dev()
    torch.manual_seed(cin + cout + k)
    B, (H, W) = 3, hw
    x = torch.randn(B, cin, H, W, device=dev).to(half2d).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5)
    b = torch.randn(cout, device=dev)
    xr = x.float().requires_grad_(True)
    wr = w.to(half2d).float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, s, p)
    xh = x.clone().requires_grad_(True)
    wh = w.clone().requires_grad_(True)
    bh = b.clone().requires_grad_(True)
    yh = Conv2dFn.apply(xh, wh, bh, s, p)
    assert yh.shape == yr.shape and yh.dtype == half2d
    assert _rel(yh, yr) < 1e-2  # bf16 output rounding: 2^-8 relative
    g = torch.randn_like(yr).to(half2d)
    yr.backward(g.float())
    yh.backward(g)
    assert _rel(xh.grad, xr.grad) < 1e-2
    assert _rel(wh.grad, wr.grad) < 5e-3  # fp32 out
    assert _rel(bh.grad, br.grad) < 5e-3


@pytest.mark.parametrize("cin,cout,hw", [(1024, 256, (5, 7)), (256, 128, (10, 14)), (64, 64, (12, 20))])
def test_conv_transpose2d_fwd_bwd(cin, cout, hw, half2d):
    from mm2d3d_amd.conv2d import ConvTranspose2dFn

    dev = _dev()
    torch.manual_seed(cin + cout)
    B, (H, W) = 2, hw
    x = torch.randn(B, cin, H, W, device=dev).to(half2d).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cin, cout, 2, 2, device=dev) / cin ** 0.5
    b = torch.randn(cout, device=dev)
    xr = x.float().requires_grad_(True)
    wr = w.to(half2d).float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, wr, br, 2)
    xh, wh, bh = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yh = ConvTranspose2dFn.apply(xh, wh, bh)
    assert yh.shape == yr.shape
    assert _rel(yh, yr) < 1e-2
    g = torch.randn_like(yr).to(half2d)
    yr.backward(g.float())
    yh.backward(g)
    assert _rel(xh.grad, xr.grad) < 1e-2
    assert _rel(wh.grad, wr.grad) < 5e-3
    assert _rel(bh.grad, br.grad) < 5e-3


@pytest.mark.parametrize("shape", [(2, 32, 48), (2, 19, 150), (1, 40, 64), (3, 9, 209)])
@pytest.mark.parametrize("cin", [3, 1])
def test_stem_conv7x7_fwd_and_weight_grad(cin, shape, half2d):
    """k_stem7 forward and k_stem_wgrad (round 5: the weight gradient from the raw strips of the staged image, 64-pixel steps along
    image rows) against torch on the same 16-bit-rounded operands: widths below, at and above the step length, not multiples of it."""
    from mm2d3d_amd.conv2d import StemConvFn

    dev = _dev()
    torch.manual_seed(cin)
    B, H, W = shape
    img = torch.rand(B, cin, H, W, device=dev)
    w = torch.randn(64, cin, 7, 7, device=dev) / (cin * 49) ** 0.5
    xr = img.to(half2d).float()
    wr = w.to(half2d).float().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 3)
    wh = w.clone().requires_grad_(True)
    yh = StemConvFn.apply(img, wh)
    assert yh.shape == yr.shape and _rel(yh, yr) < 1e-2
    g = torch.randn_like(yr).to(half2d)
    yr.backward(g.float())
    yh.backward(g)
    assert _rel(wh.grad, wr.grad) < 5e-3


def test_batched_weight_repack_equals_single_packs(half2d):
    """After an optimiser step every cached bf16 weight copy is refreshed by ONE launch (mm_pack_weights_bf16_batch); the
    result must equal the per-weight kernel for every layout in use (conv fwd/dgrad, transposed-conv fwd/dgrad), including
    sizes that are not a multiple of the 4096-element block chunk."""
    from mm2d3d_amd import conv2d as c2

    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    owners, specs = [], []
    for (cout, cin, t) in [(64, 64, 9), (128, 64, 9), (64, 192, 1), (192, 64, 1), (64, 128, 4)]:
        w = torch.nn.Parameter(torch.randn(cout, cin, t, generator=g).to(dev))
        owners.append(w)
        if t == 4:  # ConvTranspose2d weight [Cin, Cout, 2, 2] seen as (cin=cout here)
            specs.append([(4, cin, 1, cout, 1, 4, 0, cin * 4, "tfwd"), (1, cout, 4, cin, 0, cin * 4, 1, 4, "tdgrad")])
        else:
            specs.append([(1, cout, t, cin, 0, cin * t, 1, t, "fwd"), (1, cin, t, cout, 0, t, 1, cin * t, "dgrad")])
    first = [[c2._pack(w.data, *sp[:8], owner=w, kind=sp[8]).clone() for sp in sps] for w, sps in zip(owners, specs)]
    with torch.no_grad():
        for w in owners:
            w.mul_(1.5)  # bumps _version: every cached pack is stale now
    again = [[c2._pack(w.data, *sp[:8], owner=w, kind=sp[8]) for sp in sps] for w, sps in zip(owners, specs)]
    for w, sps, a, f in zip(owners, specs, again, first):
        for sp, got, old in zip(sps, a, f):
            exp = c2._pack(w.data, *sp[:8])
            assert torch.equal(got.view(torch.int16), exp.view(torch.int16)), sp
            assert not torch.equal(got.view(torch.int16), old.view(torch.int16))


@pytest.mark.parametrize("cin,cout,B,hw", [
    (256, 256, 8, (40, 48)),   # 16 tiles x 16 splits, 8 patches per workgroup: the four-stage ring wraps twice
    (512, 512, 5, (24, 40)),   # 64 tiles x 4 splits of 12, 12, 12 and 9 patches
    (512, 256, 3, (16, 33)),   # 3 patches per workgroup (prologue only), ragged right edge
    (512, 512, 2, (19, 30)),   # the bench's deepest map: 3 patches per workgroup, every patch touches the border
    (256, 256, 2, (24, 32)),   # 1 patch per workgroup
    (128, 64, 5, (33, 50)),    # Cn != Ck, 100 splits of one patch
    (64, 64, 6, (72, 112)),    # one tile, the splits fill the chip: 378 patches over 189 workgroups
])
def test_conv3x3_weight_grad_patch_ring(cin, cout, B, hw, half2d):
    """k_wgrad3x3n at patch counts per workgroup around and beyond its ring depth (1, 2, 3, 8, 12): the products are exact
    in fp32 (bf16 operands), so only the summation order differs from torch's fp32 weight gradient."""
    from mm2d3d_amd.conv2d import Conv2dFn

    dev = _dev()
    torch.manual_seed(cin * 7 + cout + B)
    H, W = hw
    x = torch.randn(B, cin, H, W, device=dev).to(half2d).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    g = torch.randn(B, cout, H, W, device=dev).to(half2d).contiguous(memory_format=torch.channels_last)
    wh = w.clone().requires_grad_(True)
    Conv2dFn.apply(x, wh, None, 1, 1).backward(g)
    ref = torch.nn.grad.conv2d_weight(x.float(), w.shape, g.float(), stride=1, padding=1)
    assert _rel(wh.grad, ref) < 1e-4
    again = w.clone().requires_grad_(True)
    Conv2dFn.apply(x, again, None, 1, 1).backward(g)
    assert torch.equal(again.grad, wh.grad)  # fixed summation order


@pytest.mark.parametrize("cin,cout,B,hw", [
    (128, 128, 20, (64, 64)),   # 320 items on 256 workgroups: 40 per XCD = one round of 32 + 8 items cut into 16 half items
    (128, 256, 9, (48, 80)),    # two cout blocks per tile: 270 items, 34 per XCD (the last XCD 32): 2 items -> 4 half items
    (256, 128, 17, (64, 64)),   # 272 items, four input chunks per item
])
def test_conv3x3_ragged_last_round_runs_as_half_items(cin, cout, B, hw, half2d):
    """k_conv3x3w<128, .>: when the last round of an XCD's items would keep at most half of its workgroups busy, those items run as
    two 64-cout halves on twice as many workgroups (csrc/conv2d.hip).  Forward (bias included) and data gradient (the same kernel
    with flipped taps) against torch fp32 on the same 16-bit-rounded operands."""
    from mm2d3d_amd.conv2d import Conv2dFn

    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin + cout + B)
    H, W = hw
    x = torch.randn(B, cin, H, W, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev).requires_grad_(True)
    b = torch.randn(cout, generator=g).to(dev).requires_grad_(True)
    y = Conv2dFn.apply(x, w, b, 1, 1)
    gy = torch.randn(B, cout, H, W, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last)
    (gx,) = torch.autograd.grad(y, [x], gy)
    xr = x.detach().float().requires_grad_(True)
    yr = F.conv2d(xr, w.detach().to(half2d).float(), b.detach(), 1, 1)
    (gxr,) = torch.autograd.grad(yr, [xr], gy.float())
    rel = lambda a, c: float((a.detach().float() - c.detach()).norm() / c.detach().norm())
    assert rel(y, yr) < 4e-3 and rel(gx, gxr) < 4e-3, (rel(y, yr), rel(gx, gxr))
    assert float((y.detach().float() - yr.detach()).abs().max()) <= 2e-2 * float(yr.detach().abs().max())
    y2 = Conv2dFn.apply(x, w, b, 1, 1)
    assert torch.equal(y2, y)  # same schedule, same sums


def _slab_totals(holder, C):
    slab, rows, nf, B = holder[0]
    assert slab.shape == (rows, 2, C) and rows % 2 == 0
    return slab.view(rows // 2, 2, 2, C).double().sum(0)  # [group][q][C]


def _expect_totals(y, nf):
    yf = y.detach().double()
    B = y.shape[0]
    out = torch.zeros(2, 2, y.shape[1], dtype=torch.float64, device=y.device)
    for g, (b0, b1) in enumerate(((0, nf), (nf, B))):
        if b1 > b0:
            out[g, 0] = yf[b0:b1].sum((0, 2, 3))
            out[g, 1] = (yf[b0:b1] ** 2).sum((0, 2, 3))
    return out


@pytest.mark.parametrize("kind,cin,cout,k,s,p,B,hw", [
    ("conv", 64, 64, 3, 1, 1, 5, (19, 23)),      # k_conv3x3r (weights resident)
    ("conv", 128, 64, 3, 1, 1, 3, (20, 28)),     # k_conv3x3w<64, .>
    ("conv", 128, 128, 3, 1, 1, 20, (64, 64)),   # k_conv3x3w<128, 16>, ragged last round -> half items
    ("conv", 128, 256, 3, 1, 1, 9, (48, 80)),    # two cout blocks per tile, half items
    ("conv", 256, 256, 3, 1, 1, 4, (8, 30)),     # k_conv3x3w<128, 32>
    ("conv", 64, 128, 3, 2, 1, 5, (20, 28)),     # implicit GEMM (stride 2)
    ("conv", 64, 128, 1, 2, 0, 5, (21, 27)),     # 1x1 downsample
    ("tconv", 128, 64, 2, 2, 0, 3, (9, 13)),     # transposed convolution: four parity GEMMs
    ("stem", 3, 64, 7, 1, 3, 3, (30, 41)),
    ("stem", 1, 64, 7, 1, 3, 4, (17, 50)),
])
@pytest.mark.parametrize("split", [False, True])
def test_conv_epilogue_files_batchnorm_statistics(kind, cin, cout, k, s, p, B, hw, split, half2d):
    """``stats=``: the convolutions file per-sub-block sums / sums of squares of their ROUNDED outputs (csrc/conv2d.hip stats_accum);
    summed over the sub-blocks they are the per-channel, per-group totals of the map - every slab element written exactly once
    (the slab starts as NaN), the map itself unchanged by the option."""
    from mm2d3d_amd import domains
    from mm2d3d_amd.conv2d import Conv2dFn, ConvTranspose2dFn, StemConvFn

    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin * 7 + cout + B)
    H, W = hw
    nf = (B // 2 or 1) if split else B
    if kind == "stem":
        x = torch.randn(B, cin, H, W, generator=g).to(dev)
        w = (torch.randn(cout, cin, 7, 7, generator=g) * 0.1).to(dev)
        run = lambda st: StemConvFn.apply(x, w, st)
    elif kind == "tconv":
        x = torch.randn(B, cin, H, W, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cin, cout, 2, 2, generator=g) * (1.0 / cin) ** 0.5).to(dev)
        b = torch.randn(cout, generator=g).to(dev)
        run = lambda st: ConvTranspose2dFn.apply(x, w, b, st)
    else:
        x = torch.randn(B, cin, H, W, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(dev)
        b = torch.randn(cout, generator=g).to(dev) if k == 3 and s == 1 else None
        run = lambda st: Conv2dFn.apply(x, w, b, s, p, None, st)
    import mm2d3d_amd.conv2d as c2d

    real_empty = torch.empty

    def nan_empty(*a, **kw):  # the slab must not rely on its initial contents
        t = real_empty(*a, **kw)
        if t.dtype == torch.float32 and t.dim() == 3 and t.shape[1] == 2:
            t.fill_(float("nan"))
        return t

    holder = [None]
    mode, c2d.BN_PRE[0] = c2d.BN_PRE[0], True  # every layer, not only the maps too large for the single-launch batch norm
    with domains.split(nf if split else None):
        c2d.torch.empty = nan_empty
        try:
            y = run(holder)
        finally:
            c2d.torch.empty = real_empty
            c2d.BN_PRE[0] = mode
        y0 = run(None)
    assert torch.equal(y, y0)
    assert holder[0][2] == nf and holder[0][3] == B
    got, want = _slab_totals(holder, cout), _expect_totals(y, nf)
    assert torch.isfinite(got).all()
    scale = want.abs().amax(-1, keepdim=True).clamp_min(1e-6)
    assert float(((got - want).abs() / scale).max()) < 2e-6, float(((got - want).abs() / scale).max())


@pytest.mark.parametrize("cin,cout,k,p,hw", [(64, 128, 3, 1, (20, 28)), (128, 256, 3, 1, (38, 60)), (64, 128, 1, 0, (20, 28)), (256, 512, 1, 0, (10, 14))])
def test_stride2_data_gradient_by_output_parity_is_bit_identical_with_the_generic_form(cin, cout, k, p, hw, half2d):
    """mm_conv2d_dgrad_s2: the data gradient of a stride-2 convolution as four tap windows (one per output parity) - the taps that
    reach a pixel, in the same order, instead of all k x k with zeros in between: the same fp32 sums, bit for bit; also against
    torch on the same 16-bit operands."""
    import mm2d3d_amd.conv2d as c2d
    from mm2d3d_amd.conv2d import Conv2dFn

    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin + cout + k)
    B, (H, W) = 3, hw
    x = torch.randn(B, cin, H, W, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, k, k, generator=g) * (2.0 / (k * k * cin)) ** 0.5).to(dev)
    gy = None
    out = []
    for mode in (True, False):
        was, c2d.DGRAD_S2[0] = c2d.DGRAD_S2[0], mode
        try:
            y = Conv2dFn.apply(x, w, None, 2, p)
            if gy is None:
                gy = torch.randn(y.shape, generator=g).to(half2d).to(dev).contiguous(memory_format=torch.channels_last)
            (gx,) = torch.autograd.grad(y, [x], gy)
        finally:
            c2d.DGRAD_S2[0] = was
        out.append(gx)
    assert torch.equal(out[0], out[1])
    xr = x.detach().float().requires_grad_(True)
    (gxr,) = torch.autograd.grad(F.conv2d(xr, w.to(half2d).float(), None, 2, p), [xr], gy.float())
    assert float((out[0].float() - gxr).norm() / gxr.norm()) < 4e-3


@pytest.mark.parametrize("cin,B,hw", [(3, 3, (30, 41)), (1, 4, (17, 50)), (3, 2, (64, 96)), (2, 2, (33, 20)), (5, 2, (16, 16)), (1, 1, (1, 1))])
def test_stem_kernel_is_bit_identical_with_the_generic_implicit_gemm(cin, B, hw, half2d):
    """k_stem7 (weights resident, the raw strip of a 16 x 16 tile staged once, pixel fragments read at shifted addresses) against the
    generic implicit GEMM on the same staged image (MM_CONV_STEM7=0): the same products in the same order, bit for bit - every
    (R, T) pair of the staging (C = 1: (8, 1), 2: (4, 2), 3: (2, 4), 5: (1, 7)), ragged tiles, the statistics slab included."""
    import mm2d3d_amd.conv2d as c2d
    from mm2d3d_amd.conv2d import StemConvFn

    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin * 13 + B)
    H, W = hw
    img = torch.randn(B, cin, H, W, generator=g).to(dev)
    w = (torch.randn(64, cin, 7, 7, generator=g) * 0.1).to(dev)
    outs, tots = [], []
    mode = c2d.BN_PRE[0]
    try:
        c2d.BN_PRE[0] = True
        for stem7 in (True, False):
            was, c2d.STEM7[0] = c2d.STEM7[0], stem7
            try:
                holder = [None]
                outs.append(StemConvFn.apply(img, w, holder))
                tots.append(_slab_totals(holder, 64))
            finally:
                c2d.STEM7[0] = was
    finally:
        c2d.BN_PRE[0] = mode
    assert torch.equal(outs[0], outs[1])
    want = _expect_totals(outs[0], B)
    scale = want.abs().amax(-1, keepdim=True).clamp_min(1e-6)
    for t in tots:
        assert float(((t - want).abs() / scale).max()) < 2e-6
    yr = F.conv2d(img.to(half2d).float(), w.to(half2d).float(), None, 1, 3)
    assert _rel(outs[0], yr) < 1e-2


def test_conv3x3_results_do_not_depend_on_kernels_of_other_streams():
    """Round 5: k_conv3x3w<64, *> (the decoder's 192 -> 64 convolutions: 96 registers per wave and 139 KB of LDS then, i.e. room on
    the CU for another kernel's workgroups) computed wrong tiles whenever small LDS-using workgroups of ANOTHER stream were launched
    onto its CU while its LDS-DMA ring was in flight - the sparse metadata kernels did it in 15 of 24 launches
    (tools/conv_corun.py, tools/corun_units.py).  The kernel now owns its CU (whole LDS, 128 registers per
    wave).  Here: the convolution and its data gradient on the main stream, the metadata build of a 280k-point batch on a second
    stream beside them, every output compared with the kernel's own output when it runs alone - bit for bit."""
    from mm2d3d_amd import nn2d
    from mm2d3d_amd.scn import metadata as md_mod
    from mm2d3d_amd.synthetic import make_batch

    dev = _dev()
    torch.manual_seed(5)
    conv = nn2d.Conv2d(192, 64, kernel_size=3, padding=1).to(dev)
    x = torch.randn(16, 192, 304, 480, device=dev).half().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    coords = make_batch(6, 8, "nuscenes", (302, 480), 6, device=dev)["x"][0].contiguous()
    g = torch.randn(16, 64, 304, 480, device=dev).half().contiguous(memory_format=torch.channels_last)

    def fwd_bwd():
        x.grad = None
        y = conv(x)
        y.backward(g)
        return y.detach(), x.grad

    ref_y, ref_dx = [t.clone() for t in fwd_bwd()]
    torch.cuda.synchronize()
    main, side = torch.cuda.current_stream(), torch.cuda.Stream(dev)
    bad = []
    for rep in range(4):
        side.wait_stream(main)
        outs = [fwd_bwd() for _ in range(3)]
        with torch.cuda.stream(side), md_mod.no_spin():
            md = md_mod.Metadata(dev, 4096, 7)
            md.build_levels(coords)
            md.build_rulebooks()
        outs += [fwd_bwd() for _ in range(3)]
        main.wait_stream(side)
        torch.cuda.synchronize()
        bad += [(rep, i) for i, (y, dx) in enumerate(outs) if not (torch.equal(y, ref_y) and torch.equal(dx, ref_dx))]
    assert not bad, f"convolution outputs changed beside another stream's kernels: launches {bad}"


@pytest.mark.parametrize("cin,cout,B,hw,split", [
    (128, 128, 20, (64, 64), True),    # <128, 16>, 320 items: ragged last round -> half items; two statistics groups
    (128, 256, 9, (48, 80), False),    # two cout blocks per tile, half items
    (256, 128, 17, (64, 64), True),    # four input chunks per item
    (256, 256, 4, (8, 30), False),     # <128, 32>, image smaller than a tile
    (512, 512, 6, (19, 30), True),     # layer4's shape: 8 x 32 tiles, eight chunks
    (192, 64, 3, (40, 56), True),      # <64, 16>: the decoder's concat convolutions
    (192, 64, 9, (152, 240), False),   # ... with several items per workgroup and an odd chunk count
    (64, 192, 2, (33, 47), False),     # <64, .> with three cout blocks (their data gradient)
    (128, 64, 5, (8, 100), False),     # <64, 32>
    (64, 64, 8, (152, 240), True),     # 64 -> 64: k_conv3x3s with RESIDENT weights (the default) against k_conv3x3r (flip | 4 and | 8), pair
                                       # list split at an XCD boundary, nine items per workgroup
    (64, 64, 5, (19, 23), False),      # ... ragged 8 x 32 tiles; 15 tiles per problem: the pair runs as two single launches
    (64, 64, 3, (1, 1), False),        # ... one pixel
])
def test_conv3x3_round6_kernels_against_the_first_kernel(cin, cout, B, hw, split, half2d):
    """Round 6: the 8-wave kernels with 128-pixel x 64-cout register tiles (four multiplying + four loader waves) against k_conv3x3w
    (flip | 4), which the tests above compare with torch: forward (with bias), data gradient and the BatchNorm statistics slab of the
    epilogue - whole items, half items, one or two statistics groups, pair mode.
      k_conv3x3v (flip | 8, 32x32x16 MFMAs): every output element is the same chain of the same MFMAs in the same order -> bit for bit;
      k_conv3x3s (the default, 16x16x32 MFMAs): one 32-deep MFMA where the others issue two 16-deep ones -> equal up to the fp32
      summation order, i.e. to one 16-bit ulp of the rounded outputs, and the slab sums to fp32 accuracy.
    64 -> 64 layers: flip | 4 and flip | 8 both select the round-3 weights-resident kernel k_conv3x3r, the default is k_conv3x3s with
    resident weights (one barrier per item)."""
    from mm2d3d_amd import conv2d as c2, domains
    from mm2d3d_amd.conv2d import Conv2dFn, Conv2dPairFn

    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(cin * 3 + cout + B)
    H, W = hw
    mk = lambda *s: torch.randn(*s, generator=g)
    xs = [mk(B, cin, H, W).to(half2d).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(2)]
    ws = [(mk(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5).to(dev).requires_grad_(True) for _ in range(2)]
    b = mk(cout).to(dev)
    gys = [mk(B, cout, H, W).to(half2d).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    old_pre = c2.BN_PRE[0]
    c2.BN_PRE[0] = True  # file the statistics whatever the map size

    def run(flag):
        c2.LEGACY3X3[0] = flag
        out = []
        with domains.split(B // 2 if split else None):
            st = [None]
            y = Conv2dFn.apply(xs[0], ws[0], b, 1, 1, None, st)
            (dx,) = torch.autograd.grad(y, [xs[0]], gys[0])
            out += [y.detach(), dx, st[0][0]]
            s1, s2 = [None], [None]
            y1, y2 = Conv2dPairFn.apply(xs[0], xs[1], ws[0], ws[1], s1, s2)
            d1, d2 = torch.autograd.grad([y1, y2], xs, gys)
            out += [y1.detach(), y2.detach(), d1, d2, s1[0][0], s2[0][0]]
        return out

    try:
        ref, v, s16, s12 = run(4), run(8), run(0), run(12)
    finally:
        c2.LEGACY3X3[0] = 0
        c2.BN_PRE[0] = old_pre
    for i, (a, r) in enumerate(zip(v, ref)):
        assert a.shape == r.shape and torch.equal(a, r), f"k_conv3x3v output {i} differs: {int((a != r).sum())} of {a.numel()} elements"
    # flip | 12 = k_conv3x3s in its streaming form for every shape: for 64 -> 64 the resident-weight form (the default) must give the
    # same bits (the same MFMAs on the same operands in the same order; only where the weight tiles come from differs)
    for i, (a, r) in enumerate(zip(s12, s16)):
        assert torch.equal(a, r), f"k_conv3x3s streaming / resident output {i} differs: {int((a != r).sum())} of {a.numel()} elements"
    ulp = 2.0 ** -10 if half2d == torch.float16 else 2.0 ** -7
    for i, (a, r) in enumerate(zip(s16, ref)):
        assert a.shape == r.shape
        a, r = a.float(), r.float()
        if a.dim() == 3:  # statistics slab [rows][2][C]: sums over 64 pixels of the rounded outputs (their squares)
            tol = 64 * 4 * ulp * float(r.abs().max().clamp_min(1.0)) / 8
            assert float((a - r).abs().max()) <= tol, (i, float((a - r).abs().max()), tol)
        else:
            err = (a - r).abs()
            assert float((err / r.abs().clamp_min(0.25)).max()) <= 2.1 * ulp, (i, float((err / r.abs().clamp_min(0.25)).max()))
            assert float((a != r).float().mean()) < 0.02, (i, float((a != r).float().mean()))  # a flipped rounding here and there
    yr = F.conv2d(xs[0].detach().float(), ws[0].detach().to(half2d).float(), b, 1, 1)
    assert float((s16[0].float() - yr).norm() / yr.norm()) < 4e-3
