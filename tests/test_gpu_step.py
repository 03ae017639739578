"""GPU parity of the losses, lifting, fused AdamW and the full two-domain step vs the CPU oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _dev():
    import mm2d3d_amd  # noqa: F401

    return torch.device("cuda:0")


@pytest.mark.parametrize("name", ["w6", "none"])
def test_cross_entropy_matches_reference_golden(name):
    from mm2d3d_amd.losses import Loss

    dev = _dev()
    z = np.load(os.path.join(G, "loss_ce.npz"))
    w = z[f"ce_{name}/weight"].tolist()
    cfg = [{"name": "cross_entropy", "weight": 1.0, "target": "segmentation", "args": ({"weight": w} if w else {})}]
    x = torch.from_numpy(z["logits"]).to(dev).requires_grad_(True)
    v = Loss(cfg)("segmentation", pred=x, gt=torch.from_numpy(z["labels"]).to(dev))
    v.backward()
    assert abs(v.item() - float(z[f"ce_{name}/value"])) < 1e-5
    assert np.allclose(x.grad.cpu().numpy(), z[f"ce_{name}/grad"], atol=1e-6)


def test_kl_matches_torch_kl_div():
    from mm2d3d_amd.losses import cross_modal_loss
    from oracle.step_ref import cross_modal_loss as ref

    dev = _dev()
    torch.manual_seed(1)
    a, b, c, d = (torch.randn(1000, 6) * 3 for _ in range(4))
    b.requires_grad_(True), d.requires_grad_(True)
    r1, r2 = ref(a, b, c, d)
    (r1 + 2 * r2).backward()
    bh, dh = b.detach().clone().to(dev).requires_grad_(True), d.detach().clone().to(dev).requires_grad_(True)
    h1, h2 = cross_modal_loss(a.to(dev), bh, c.to(dev), dh)
    (h1 + 2 * h2).backward()
    assert abs(h1.item() - r1.item()) < 1e-5 and abs(h2.item() - r2.item()) < 1e-5
    assert torch.allclose(bh.grad.cpu(), b.grad, atol=1e-7) and torch.allclose(dh.grad.cpu(), d.grad, atol=1e-7)


def test_lifting_gather_and_duplicate_accumulating_backward():
    from mm2d3d_amd.lifting import PixelIndex, lift
    from oracle.net2d_ref import lift as ref_lift

    dev = _dev()
    g = np.random.default_rng(0)
    B, C, H, W = 3, 6, 17, 23
    idx = [np.stack([g.integers(0, H, n), g.integers(0, W, n)], 1) for n in (400, 1, 700)]  # many duplicate pixels
    seg = torch.randn(B, C, H, W)
    sr = seg.clone().requires_grad_(True)
    sh = seg.clone().to(dev).requires_grad_(True)
    o_r = ref_lift(sr, idx)
    o_h = lift(sh, PixelIndex(idx, H, W, dev))
    assert torch.equal(o_h.cpu(), o_r)  # pure gather: bit-exact
    wgt = torch.randn_like(o_r)
    (o_r * wgt).sum().backward()
    (o_h * wgt.to(dev)).sum().backward()
    assert torch.allclose(sh.grad.cpu(), sr.grad, atol=1e-5)
    with pytest.raises(IndexError):
        PixelIndex([np.array([[H, 0]])], H, W, dev)


def test_flat_adamw_onecycle_matches_reference_golden():
    from mm2d3d_amd.optimizers import Optimizer

    dev = _dev()
    z = np.load(os.path.join(G, "optimizer_adamw_onecycle.npz"))
    p = torch.nn.Parameter(torch.from_numpy(z["p0"]).to(dev))
    opt = Optimizer("adamw", lr=0.001)
    opt.set_scheduler("one_cycle", max_lr=0.005, total_steps=50)
    o, s = opt.build([p])
    gen = torch.Generator().manual_seed(3)
    for i in range(49):
        o.zero_grad()
        (p * torch.randn(10, generator=gen).to(dev)).sum().backward()
        o.step()
        s.step()
        assert abs(o.param_groups[0]["lr"] - z["lrs"][i]) < 1e-12
        assert np.allclose(p.detach().cpu().numpy(), z["params"][i], atol=2e-6), i


def test_unused_parameters_keep_their_weights():
    from mm2d3d_amd.optimizers import FlatAdamW

    dev = _dev()
    a, b = torch.nn.Parameter(torch.ones(5, device=dev)), torch.nn.Parameter(torch.ones(7, device=dev))
    o = FlatAdamW([a, b], lr=0.1)
    o.zero_grad()
    (a * 2).sum().backward()
    o.step()
    assert torch.equal(b.detach().cpu(), torch.ones(7)) and not torch.equal(a.detach().cpu(), torch.ones(5))


@pytest.mark.parametrize("scaled", [False, True])
def test_adamw_touched_range_that_starts_unaligned_matches_torch_adamw(scaled):
    """an untouched 6-element parameter ahead of touched ones makes the touched range start at arena offset
    6 (not 16-byte aligned): the scalar instantiations k_adamw<false> / <false, true> run.  Every element must be updated
    exactly once per step, as torch.optim.AdamW does."""
    from mm2d3d_amd.amp import GradScaler
    from mm2d3d_amd.optimizers import FlatAdamW

    dev = _dev()
    g = torch.Generator().manual_seed(11)
    shapes = [(6,), (3, 1), (1,), (1037,), (16, 6), (6,), (5000,)]
    init = [torch.randn(s, generator=g) for s in shapes]
    hp = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    rp = [torch.nn.Parameter(t.clone()) for t in init]
    o = FlatAdamW(hp, lr=0.01, weight_decay=0.05)
    r = torch.optim.AdamW(rp[1:], lr=0.01, weight_decay=0.05)  # parameter 0 never receives a gradient
    scaler = GradScaler(dev, init_scale=1024.0) if scaled else None
    for step in range(4):
        ws = [torch.randn(s, generator=g) for s in shapes]
        o.zero_grad(), r.zero_grad()
        lh = sum((p * w.to(dev)).sum() for p, w in zip(hp[1:], ws[1:]))
        lr_ = sum((p * w).sum() for p, w in zip(rp[1:], ws[1:]))
        if scaled:
            scaler.scale(lh).backward()
            scaler.step(o)
            scaler.update()
        else:
            lh.backward()
            o.step()
        lr_.backward()
        r.step()
        lo = o._touched_ranges(o._arenas[0])[0][0]
        assert lo == 6 and (o._arenas[0]["p"][lo:].data_ptr() & 15) != 0  # the case under test
        assert torch.equal(hp[0].detach().cpu(), init[0])
        for a, b in zip(hp[1:], rp[1:]):
            assert torch.allclose(a.detach().cpu(), b.detach(), atol=2e-6, rtol=1e-6), step
    if scaled:
        assert scaler.steps_taken(o) == 4


@pytest.mark.parametrize("kind", ["fp16", "bf16"])
def test_full_step_losses_and_gradients_vs_oracle(kind):
    """One two-domain step on 1+1 small-image scenes: six loss terms within 1e-3, 3D/2D gradients consistent."""
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel
    from oracle.net3d_ref import Net3DSegRef
    from oracle.step_ref import generic_step

    dev = _dev()
    torch.manual_seed(0)
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    W = [1.9241476, 1.0, 2.16763851, 2.78254323, 1.54875664, 1.85686537]
    n2, n3 = Net2DSeg(6, pretrained=False), Net3DSeg(6, True, kw)
    for m in n2.modules():  # dropout is random: disable it on both sides for parity
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    ref3 = Net3DSegRef(6, True, kw)
    ref3.load_state_dict(n3.state_dict())
    sd2 = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in n2.state_dict().items()}
    cpu = {"source": make_batch(5, 1, "nuscenes", (48, 64)), "target": make_batch(6, 1, "nuscenes", (48, 64))}
    gpu = {"source": make_batch(5, 1, "nuscenes", (48, 64), device=dev), "target": make_batch(6, 1, "nuscenes", (48, 64), device=dev)}
    ref_total, ref_logs = generic_step(sd2, ref3, cpu, W)
    ref_total.backward()
    # the same step with the 2D branch rounding to bf16 where the HIP branch stores bf16 (forward values and gradients)
    import copy

    sd2e = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in sd2.items()}
    ref3e = copy.deepcopy(ref3)
    ref3e.zero_grad()
    cpu_e = {"source": make_batch(5, 1, "nuscenes", (48, 64)), "target": make_batch(6, 1, "nuscenes", (48, 64))}
    scale = 1024.0 if kind == "fp16" else 1.0  # IEEE fp16 gradient maps want the trainer's loss scale (mm2d3d_amd/amp.py)
    emu_total, emu_logs = generic_step(sd2e, ref3e, cpu_e, W, emulate_bf16=torch.float16 if kind == "fp16" else torch.bfloat16)
    (emu_total * scale).backward()
    if scale != 1.0:
        for v in list(sd2e.values()) + list(ref3e.parameters()):
            if v.grad is not None:
                v.grad.div_(scale)
    tm = TrainModel({"2d_net": n2.to(dev), "3d_net": n3.to(dev)}, None,
                    Loss([{"name": "cross_entropy", "target": "segmentation", "args": {"weight": W}}]),
                    dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, precision=kind))
    total = tm.training_step(gpu)
    (total * scale).backward()
    if scale != 1.0:
        for p_ in list(n2.parameters()) + list(n3.parameters()):
            if p_.grad is not None:
                p_.grad.div_(scale)
    # 3D-only terms: fp32 on both sides -> 1e-3 (north_star).  Terms fed by the 2D branch (bf16 activations, as the
    # reference's fp16 AMP) -> 3e-2.
    tol = {"loss_segmentation_3d": 1e-3}
    assert abs(total.item() - ref_total.item()) < 3e-2 * max(1.0, abs(ref_total.item()))
    for k, v in ref_logs.items():
        assert abs(tm.last_logs[f"train/{k}"].item() - v.item()) < tol.get(k, 3e-2) * max(1.0, abs(v.item())), k
    g3 = dict(ref3.named_parameters())
    worst = 0.0
    for name, p in n3.named_parameters():
        if p.grad is None:
            continue
        t = g3[name].grad
        worst = max(worst, ((p.grad.cpu() - t).abs().max() / max(1.0, t.abs().max())).item())
    assert worst < 5e-2, worst  # fp32 conditioning ~1e-2 (test_gpu_scn.py) + the bf16 2D logits entering the KL terms
    # against the bf16-EMULATING oracle what is left is accumulation order and the 1-ulp flips it causes: loss terms ...
    worst_loss = 0.0
    for k, v in emu_logs.items():
        worst_loss = max(worst_loss, abs(tm.last_logs[f"train/{k}"].item() - v.item()) / max(1.0, abs(v.item())))
    cos_f, cos_e = [], []
    for name, p in n2.named_parameters():
        if p.grad is None:
            continue
        t, te = sd2[name].grad, sd2e[name].grad
        g = p.grad.cpu().flatten().double()
        cf = torch.nn.functional.cosine_similarity(g, t.flatten().double(), dim=0).item()
        ce = torch.nn.functional.cosine_similarity(g, te.flatten().double(), dim=0).item()
        if t.norm() >= 1e-2:  # tiny early-layer gradients are the noisiest in bf16
            cos_f.append((cf, name))
            cos_e.append((ce, name))
    med = lambda v: float(np.median([c for c, _ in v]))
    print(f"2D gradients of the composed step: cosine vs fp32 oracle min {min(cos_f)[0]:.3f} median {med(cos_f):.3f}; "
          f"vs bf16-emulating oracle min {min(cos_e)[0]:.3f} median {med(cos_e):.3f}; loss terms vs emulating oracle {worst_loss:.2e}")
    # ... and the composed 2D gradients (bf16 activations AND bf16 gradients through ~40 layers; per-kernel backward parity is
    # pinned tightly in test_gpu_conv2d.py / test_gpu_net2d.py).  Measured: cosine min 0.976 / median 1.000 against the emulating
    # oracle (0.934 / 1.000 against the fp32 oracle; round 2 asserted 0.6); the six loss terms agree with the emulating oracle
    # to 4.5e-6 - i.e. within north_star's 1e-3 once the storage format is the same on both sides.
    assert min(cos_f)[0] > 0.85, sorted(cos_f)[:3]
    assert min(cos_e)[0] > 0.93 and med(cos_e) > 0.99, sorted(cos_e)[:3]
    assert worst_loss < 1e-4


def test_joint_domain_pass_equals_the_two_call_sequence(bf16_mode):
    """TrainModel batches [source | target] into one pass per network with per-domain batch-norm statistics
    (mm2d3d_amd/domains.py); losses, gradients, running statistics and batch counters must be those of the reference's
    literal two-call sequence (joint_domains=False)."""
    import copy

    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = _dev()
    torch.manual_seed(0)
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    W = [1.9241476, 1.0, 2.16763851, 2.78254323, 1.54875664, 1.85686537]
    n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
    for m in n2.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
    mk = lambda: {"source": make_batch(5, 2, "nuscenes", (48, 64), device=dev), "target": make_batch(6, 1, "nuscenes", (48, 64), device=dev)}
    loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {"weight": W}}])
    two = TrainModel({"2d_net": n2b, "3d_net": n3b}, None, loss, dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, joint_domains=False))
    one = TrainModel({"2d_net": n2, "3d_net": n3}, None, loss, dict(lambda_xm_src=1.0, lambda_xm_trg=0.1))
    t2 = two.training_step(mk())
    t2.backward()
    t1 = one.training_step(mk())
    t1.backward()
    for k, v in two.last_logs.items():
        tol = 1e-5 if k.endswith("segmentation_3d") else 2e-3  # 3D: fp32 end to end; the rest sees bf16 2D logits
        assert abs(one.last_logs[k].item() - v.item()) < tol * max(1.0, abs(v.item())), (k, one.last_logs[k].item(), v.item())
    for (name, a), (_, b) in zip(list(n2.state_dict().items()) + list(n3.state_dict().items()),
                                 list(n2b.state_dict().items()) + list(n3b.state_dict().items())):
        if name.endswith("num_batches_tracked"):
            assert int(a) == int(b) == 2, name
        elif "running" in name:
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-4), name
    for (name, p), (_, q) in zip(n3.named_parameters(), n3b.named_parameters()):
        if q.grad is None:
            assert p.grad is None, name
            continue
        assert (p.grad - q.grad).abs().max() <= 2e-2 * max(1e-3, float(q.grad.abs().max())), name
    for (name, p), (_, q) in zip(n2.named_parameters(), n2b.named_parameters()):
        if q.grad is None:
            continue
        cos = torch.nn.functional.cosine_similarity(p.grad.flatten().double(), q.grad.flatten().double(), dim=0).item()
        assert cos > 0.98 or q.grad.norm() < 1e-3, (name, cos)

@pytest.mark.parametrize("mode", [1, pytest.param(2, marks=pytest.mark.skipif(
    os.environ.get("MM_TEST_EXPERIMENTAL", "0") == "0",
    reason="overlap_branches=2 is experimental (a BatchNorm2d grid barrier can starve beside the 3D stream: DESIGN.md section 4); "
           "set MM_TEST_EXPERIMENTAL=1 to run it"))])
def test_branches_on_two_streams_equal_the_single_stream_step(mode):
    """train_kwargs["overlap_branches"]: the 3D branch runs on its own stream (forward and, through autograd's stream rules,
    backward).  Same kernels, same order within each branch: every loss term and every gradient must be BIT-identical to the
    single-stream step with the same batch-norm kernels (mode 1: three-kernel everywhere; mode 2, the default: three-kernel in the
    sparse branch, single-launch in the 2D branch) - a missing stream dependency shows up as a difference."""
    import copy

    from mm2d3d_amd import _lib
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = _dev()
    try:
        torch.manual_seed(1)
        kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
        W = [1.9241476, 1.0, 2.16763851, 2.78254323, 1.54875664, 1.85686537]
        n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
        for m in n2.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
        mk = lambda: {"source": make_batch(5, 2, "nuscenes", (96, 128), device=dev), "target": make_batch(6, 2, "nuscenes", (96, 128), device=dev)}
        loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {"weight": W}}])
        kwargs = dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, gc_freeze=False, bn2d_fused=0 if mode == 1 else 3, bn3d_fused=0)
        one = TrainModel({"2d_net": n2, "3d_net": n3}, None, loss, dict(kwargs, overlap_branches=0))
        two = TrainModel({"2d_net": n2b, "3d_net": n3b}, None, loss, dict(kwargs, overlap_branches=mode))
        for _ in range(3):  # several steps: the stream handoffs repeat with recycled allocator blocks
            for tm in (one, two):
                for p in tm.model.parameters():
                    p.grad = None
            t1 = one.training_step(mk())
            t1.backward()
            t2 = two.training_step(mk())
            t2.backward()
            torch.cuda.synchronize()
            for k, v in one.last_logs.items():
                assert two.last_logs[k].detach().item() == v.detach().item(), k
            for (name, p), (_, q) in zip(list(n2.named_parameters()) + list(n3.named_parameters()),
                                         list(n2b.named_parameters()) + list(n3b.named_parameters())):
                assert (p.grad is None) == (q.grad is None), name
                if p.grad is not None:
                    assert torch.equal(p.grad, q.grad), name
    finally:
        pass


def test_gradient_sinks_equal_autograd_accumulation():
    """With FlatAdamW installed, weight-gradient kernels accumulate straight into the flat arena (gradsink.py); the arena
    after backward must equal what plain autograd accumulation produces for the same step."""
    import copy

    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = _dev()
    torch.manual_seed(0)
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    W = [1.9241476, 1.0, 2.16763851, 2.78254323, 1.54875664, 1.85686537]
    n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
    for m in n2.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
    mk = lambda: {"source": make_batch(5, 1, "nuscenes", (48, 64), device=dev), "target": make_batch(6, 1, "nuscenes", (48, 64), device=dev)}
    loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {"weight": W}}])
    plain = TrainModel({"2d_net": n2b, "3d_net": n3b}, None, loss, dict(lambda_xm_src=1.0, lambda_xm_trg=0.1))
    plain.training_step(mk()).backward()
    opts = {k: Optimizer("adamw", lr=1e-3) for k in ("2d_net", "3d_net")}
    sunk = TrainModel({"2d_net": n2, "3d_net": n3}, opts, loss, dict(lambda_xm_src=1.0, lambda_xm_trg=0.1))
    sunk.configure_optimizers()
    for o in sunk.optimizers:
        o.zero_grad()
    sunk.training_step(mk()).backward()
    n_sink = 0
    for (name, p), (_, q) in zip(list(n2.named_parameters()) + list(n3.named_parameters()),
                                 list(n2b.named_parameters()) + list(n3b.named_parameters())):
        if q.grad is None:
            assert p._mm_pending == 0 and float(p._mm_sink.abs().max()) == 0.0, name
            continue
        n_sink += hasattr(p, "_mm_sink")
        assert p._mm_pending == 0, name
        ref = q.grad.float()
        assert torch.allclose(p._mm_sink, ref, rtol=2e-3, atol=2e-4 * float(ref.abs().max()) + 1e-7), name
    assert n_sink > 300
    for o in sunk.optimizers:  # every parameter that received a gradient is marked for the update
        for a in o._arenas:
            assert sum(a["touched"]) >= len(a["params"]) - 6


def test_deferred_sparse_weight_gradient_sums_equal_per_layer_sums():
    """scn.ops._DwBatch: the slab sums of every sparse layer of a backward pass in ONE launch (mm_spconv_dw_reduce_batch) give
    bit for bit the gradient arena of the per-layer launches (mm_spconv_dw), fp32 rows and 16-bit rows."""
    import copy

    from mm2d3d_amd import scn
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.scn import ops
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = _dev()
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    W = [1.9241476, 1.0, 2.16763851, 2.78254323, 1.54875664, 1.85686537]
    mk = lambda: {"source": make_batch(5, 2, "nuscenes", (48, 64), device=dev), "target": make_batch(6, 2, "nuscenes", (48, 64), device=dev)}
    loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {"weight": W}}])
    prev = ops.DW_BATCH[0]
    try:
        for act in (None, torch.bfloat16):
            torch.manual_seed(0)
            n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
            for m in n2.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
            arenas = []
            for batched in (True, False):
                ops.DW_BATCH[0] = batched
                a2, a3 = copy.deepcopy(n2), copy.deepcopy(n3)
                opts = {k: Optimizer("adamw", lr=1e-3) for k in ("2d_net", "3d_net")}
                kwargs = dict(lambda_xm_src=1.0, lambda_xm_trg=0.1)
                kwargs["sparse_activations"] = "bf16" if act is not None else "fp32"
                tm = TrainModel({"2d_net": a2, "3d_net": a3}, opts, loss, kwargs)
                tm.configure_optimizers()
                for o in tm.optimizers:
                    o.zero_grad()
                tm.training_step(mk()).backward()
                torch.cuda.synchronize()
                assert ops._DWB.items == [] and ops._DWB.expected == 0
                arenas.append({n: p._mm_sink.clone() for n, p in a3.named_parameters() if hasattr(p, "_mm_sink")})
                assert all(p._mm_pending == 0 for p in a3.parameters() if hasattr(p, "_mm_sink"))
            assert len(arenas[0]) > 50
            for n in arenas[0]:
                assert torch.equal(arenas[0][n], arenas[1][n]), (act, n)
    finally:
        ops.DW_BATCH[0] = prev
        scn.set_activation_dtype(torch.float32)


def test_grad_scaler_semantics_on_the_device():
    """mm2d3d_amd/amp.py restates torch.cuda.amp.GradScaler (what the reference's ``precision: 16`` trainer drives) with its state
    on the device: (1) a clean step applies exactly the update of ``step(grad_scale=1/scale)``; (2) a gradient holding an
    inf / nan leaves parameters, moments and the bias-correction step counter untouched and halves the scale; (3)
    ``growth_interval`` clean steps double it; (4) an optimiser with clean gradients still steps when another one overflowed."""
    from mm2d3d_amd.amp import GradScaler
    from mm2d3d_amd.optimizers import FlatAdamW

    dev = _dev()
    torch.manual_seed(0)
    mk = lambda: [torch.nn.Parameter(torch.randn(37, 5, device=dev)), torch.nn.Parameter(torch.randn(1001, device=dev))]
    pa, pb, pc = mk(), mk(), mk()
    for q, r, t in zip(pa, pb, pc):
        r.data.copy_(q.data)
        t.data.copy_(q.data)
    oa, ob, oc = (FlatAdamW(p, lr=1e-2, weight_decay=0.01) for p in (pa, pb, pc))
    sc = GradScaler(dev, init_scale=1024.0, growth_interval=2)
    g = [torch.randn_like(q) for q in pa]

    def load(opt, params, mult, poison=None):
        opt.zero_grad()
        for q, gg in zip(params, g):
            q.grad.copy_(gg * mult)
        if poison is not None:
            params[1].grad[17] = poison
        opt.mark_all_touched()

    # (1) clean step == the plain update on unscaled gradients
    load(oa, pa, 1024.0)
    load(ob, pb, 1024.0)
    sc.step(oa)
    sc.update()
    ob.step(grad_scale=1.0 / 1024.0)
    for q, r in zip(pa, pb):
        assert torch.allclose(q, r, rtol=0, atol=1e-7), float((q - r).abs().max())
    assert sc.steps_taken(oa) == 1 and sc.get_scale() == 1024.0
    # (2) overflow: nothing moves, the scale halves, the step counter stays; (4) the other optimiser steps
    keep = [q.detach().clone() for q in pa]
    m0 = oa._arenas[0]["m"].clone()
    for poison in (float("inf"), float("nan")):
        load(oa, pa, 1024.0, poison)
        load(oc, pc, sc.get_scale())
        before_c = [q.detach().clone() for q in pc]
        sc.step(oa)
        sc.step(oc)
        sc.update()
        for q, k in zip(pa, keep):
            assert torch.equal(q, k)
        assert torch.equal(oa._arenas[0]["m"], m0)
        assert any(not torch.equal(q, k) for q, k in zip(pc, before_c))
    assert sc.steps_taken(oa) == 1 and sc.steps_taken(oc) == 2
    assert sc.get_scale() == 256.0
    # (3) two clean steps (growth_interval) double the scale; the bias corrections continue from step 2
    for i in range(2):
        load(oa, pa, sc.get_scale())
        load(ob, pb, 1.0)
        sc.step(oa)
        sc.update()
        ob.step()
        for q, r in zip(pa, pb):
            assert torch.allclose(q, r, rtol=0, atol=2e-7), (i, float((q - r).abs().max()))
    assert sc.get_scale() == 512.0 and sc.steps_taken(oa) == 3
    assert oa.state_dict()["step"] == 3
    sd = sc.state_dict()
    sc2 = GradScaler(dev)
    sc2.load_state_dict(sd)
    assert sc2.get_scale() == 512.0 and sc2.growth_interval == 2


def test_training_steps_with_fp16_maps_and_rows_under_the_loss_scale():
    """``precision: "fp16"`` + ``sparse_activations: "fp16"``: both branches store IEEE fp16 (the reference's ``precision: 16``),
    the step runs under the device-resident GradScaler.  First-step losses agree with the bf16 configuration of the same
    weights within the storage formats' rounding; six steps on a fixed batch stay finite, lower the loss, take every step and
    leave the scale at 65536."""
    import copy

    from mm2d3d_amd import nn2d, scn
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = _dev()
    torch.manual_seed(0)
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    W = [1.9241476, 1.0, 2.16763851, 2.78254323, 1.54875664, 1.85686537]
    n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
    for m in n2.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
    mk = lambda: {"source": make_batch(5, 2, "nuscenes", (48, 64), device=dev), "target": make_batch(6, 2, "nuscenes", (48, 64), device=dev)}
    loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {"weight": W}}])
    try:
        ref = TrainModel({"2d_net": n2b, "3d_net": n3b}, None, loss, dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, precision="bf16",
                                                                             sparse_activations="bf16"))
        l_bf = float(ref.training_step(mk()))
        opts = {k: Optimizer("adamw", lr=1e-3) for k in ("2d_net", "3d_net")}
        tm = TrainModel({"2d_net": n2, "3d_net": n3}, opts, loss, dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, precision="fp16",
                                                                     sparse_activations="fp16", gc_freeze=False))
        assert nn2d._c2d.HALF[0] == torch.float16 and scn.ACTIVATION_DTYPE[0] == torch.float16
        l_fp = float(tm.training_step(mk()))
        assert abs(l_fp - l_bf) < 2e-2 * max(1.0, abs(l_bf)), (l_fp, l_bf)
        losses = [float(tm.fit_step(mk())) for _ in range(6)]
        assert all(np.isfinite(v) for v in losses) and losses[-1] < losses[0], losses
        assert tm.scaler is not None and tm.scaler.get_scale() == 65536.0
        assert [tm.scaler.steps_taken(o) for o in tm.optimizers] == [6, 6]
        for p in list(n2.parameters()) + list(n3.parameters()):
            assert bool(torch.isfinite(p).all())
        # checkpoint round trip: the loss scale travels under Lightning's key, the step counters resume from the optimisers'
        ck = tm.checkpoint()
        assert ck["native_amp_scaling_state"]["scale"] == 65536.0 and [sd["step"] for sd in ck["optimizer_states"]] == [6, 6]
        ck["native_amp_scaling_state"]["scale"] = 1024.0
        tm.load_checkpoint(ck)
        assert np.isfinite(float(tm.fit_step(mk())))
        assert tm.scaler.get_scale() == 1024.0 and [tm.scaler.steps_taken(o) for o in tm.optimizers] == [7, 7]
    finally:
        nn2d.set_precision(nn2d.DEFAULT_PRECISION)
        scn.set_activation_dtype(torch.float32)


def test_eval_confusion_iou_and_checkpoint_roundtrip(tmp_path):
    from mm2d3d_amd.metrics import SegIoU

    dev = _dev()
    torch.manual_seed(3)
    N, C = 5000, 6
    a, b = torch.randn(N, C) * 2, torch.randn(N, C) * 2
    y = torch.randint(0, C, (N,))
    y[::7] = -100
    m = SegIoU(C, dev)
    m.update(a[:3000].to(dev), b[:3000].to(dev), y[:3000].to(dev))
    m.update(a[3000:].to(dev), b[3000:].to(dev), y[3000:].to(dev))
    keep = y != -100
    ens = (F.softmax(a, 1) + F.softmax(b, 1)) / 2
    for i, pred in enumerate((a.argmax(1), b.argmax(1), ens.argmax(1))):
        cm = torch.zeros(C, C, dtype=torch.int64)
        cm.index_put_((y[keep], pred[keep]), torch.ones(int(keep.sum()), dtype=torch.int64), accumulate=True)
        assert torch.equal(m.cm[i].cpu(), cm)
        tp = cm.diag().double()
        iou = tp / (cm.sum(0) + cm.sum(1) - cm.diag()).double()
        assert torch.allclose(m.compute()[SegIoU.NAMES[i]].cpu().double(), iou, atol=1e-6)
    # checkpoint round trip of a tiny trainer (keys follow the reference: model.<net>.model.*)
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.train import TrainModel

    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=3)
    mk = lambda: TrainModel({"2d_net": torch.nn.Linear(2, 2).to(dev), "3d_net": Net3DSeg(6, True, kw).to(dev)},
                            {k: Optimizer("adamw", lr=1e-3) for k in ("2d_net", "3d_net")}, Loss("cross_entropy"), {})
    t1, t2 = mk(), mk()
    t1.configure_optimizers(), t2.configure_optimizers()
    t1.best["best_target_iou"] = 0.5
    ck = t1.checkpoint()
    assert any(k.startswith("model.3d_net.model.net_3d.layer2.weight") for k in ck["state_dict"])  # the reference's key path
    torch.save(ck, tmp_path / "last.ckpt")
    t2.load_checkpoint(torch.load(tmp_path / "last.ckpt", weights_only=False))
    for (k1, v1), (k2, v2) in zip(t1.model.state_dict().items(), t2.model.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    assert t2.best["best_target_iou"] == 0.5


def test_metadata_built_one_step_ahead_gives_the_same_steps():
    """fit_step(batch, next_batch=): the sparse metadata of the next batch is built during the current step (dedupe chain before
    the forward, rulebooks before the backward, asynchronous read-backs).  Same kernels on the same stream in a different
    order: losses and parameters after four optimiser steps must be BIT-identical to the in-line build."""
    import copy

    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = _dev()
    torch.manual_seed(3)
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
    for m in n2.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)

    def opts():
        out = {}
        for k in ("2d_net", "3d_net"):
            o = Optimizer("adamw", lr=0.001)
            o.set_scheduler("one_cycle", max_lr=0.005, total_steps=100)
            out[k] = o
        return out

    # different scenes per step: a stale or mismatched metadata object would show
    batches = [{"source": make_batch(5, 2, "nuscenes", (96, 128), device=dev, first_scene=4 * i),
                "target": make_batch(6, 2, "nuscenes", (96, 128), device=dev, first_scene=4 * i + 2)} for i in range(4)]
    clone = lambda b: {d: dict(v, x=[v["x"][0], v["x"][1].clone()]) for d, v in b.items()}
    loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {}}])
    tk = dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, gc_freeze=False)
    piped = TrainModel({"2d_net": n2, "3d_net": n3}, opts(), loss, dict(tk))
    plain = TrainModel({"2d_net": n2b, "3d_net": n3b}, opts(), loss, dict(tk))
    seq = [clone(b) for b in batches]
    for i in range(4):
        nxt = seq[i + 1] if i + 1 < 4 else None
        la = piped.fit_step(seq[i], next_batch=nxt)
        if nxt is not None:
            assert piped._pipelined is not None and piped._pipelined["key"] is nxt and piped._pipelined["phase"] == 2
        lb = plain.fit_step(clone(batches[i]))
        assert float(la.detach()) == float(lb.detach()), (i, float(la.detach()), float(lb.detach()))
    torch.cuda.synchronize()
    for a, b in zip(piped.optimizers, plain.optimizers):
        for x, y in zip(a._arenas, b._arenas):
            if x is not None:
                assert torch.equal(x["p"], y["p"])
