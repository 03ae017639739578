"""The RCCL path of the data-parallel trainer, executed on the one GPU a test box has: a process group of ONE rank over the
"nccl" (= RCCL) backend with the reducer forced on (mm2d3d_amd/ddp.py ``force``).  Every bucket is all-reduced by RCCL on
RCCL's stream, launched from the backward hooks beside the rest of backward - the launch / wait / stream-ordering machinery is
the real one; what a single card cannot show is the xGMI transfer itself (covered by logic on gloo, tests/test_ddp_gloo.py)."""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_forced_one_rank_rccl_step_equals_the_plain_step_bit_for_bit(monkeypatch):
    import copy

    from mm2d3d_amd import _lib
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = torch.device("cuda:0")
    L = _lib.lib()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        torch.manual_seed(0)
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

        mk = lambda: {"source": make_batch(5, 2, "nuscenes", (96, 128), device=dev), "target": make_batch(6, 2, "nuscenes", (96, 128), device=dev)}
        loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {}}])
        tk = dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, gc_freeze=False, bn2d_fused=0, bn3d_fused=0)  # both trainers' handles on the three-kernel batch norms (what DDP selects)
        monkeypatch.setenv("MM_DDP_FORCE", "1")
        monkeypatch.setenv("MM_DDP_OVERLAP", "1")  # buckets launched from the backward hooks, beside the rest of backward
        ddp = TrainModel({"2d_net": n2, "3d_net": n3}, opts(), loss, dict(tk))
        ddp.configure_optimizers()
        monkeypatch.setenv("MM_DDP_FORCE", "0")
        plain = TrainModel({"2d_net": n2b, "3d_net": n3b}, opts(), loss, dict(tk))
        plain.configure_optimizers()
        assert ddp.reducer.active and ddp.reducer.overlap and len(ddp.reducer.buckets) >= 2 and not plain.reducer.active
        for step in range(4):
            la, lb = ddp.fit_step(mk()), plain.fit_step(mk())
            torch.cuda.synchronize()
            assert float(la) == float(lb), (step, float(la), float(lb))
            st = ddp.reducer.stats
            assert st["buckets"] == len(ddp.reducer.order) or step == 0
            if step >= 1:  # from the second step on every bucket goes out DURING backward
                assert st["early"] == st["buckets"] > 0, st
        assert ddp.reducer.learned and ddp.reducer.consistent and len(ddp.reducer.unused) > 0
        for a, b in zip(ddp.optimizers, plain.optimizers):
            for x, y in zip(a._arenas, b._arenas):
                if x is not None:
                    assert torch.equal(x["p"], y["p"]), "parameters after 4 optimiser steps differ"
    finally:
        dist.destroy_process_group()
        pass


@pytest.mark.parametrize("overlap", [False, True, "tail"])
def test_fit_step_under_rccl_never_makes_the_host_wait_and_keeps_the_forward_batch_norms_single_launch(monkeypatch, overlap):
    """under data parallelism (i) no host synchronisation inside ``fit_step`` - ``Tensor.item`` / ``.cpu`` /
    ``.tolist`` / ``bool(tensor)`` / ``torch.cuda.synchronize`` are booby-trapped from the third step on (the collective "graph
    changed" flag is read one step late from pinned memory, ddp._read_flag); (ii) the forward batch norms stay on the single-launch
    kernels (no collective runs beside the forward pass: stream order), the backward ones take the three-kernel path; (iii) the
    losses follow the plain trainer's (whose backward uses the single-launch kernels: other summation order, same mathematics).
    ``overlap`` False (MM_DDP_OVERLAP=0): every bucket goes out in finish(), after backward - no collective beside a
    grid barrier, so every batch norm keeps its single-launch kernel and the step is the plain trainer's step bit for bit.
    ``overlap`` "tail" (the default since round 5): the same kernels, the same bits - and from the second step on every bucket but
    those that complete in the barrier-free tail's last moments leaves BEFORE finish(), after the last grid-barrier kernel of the
    backward pass (ddp.py); at least the 3D network's and the decoder / layer2-4 buckets of the 2D network."""
    import copy

    from mm2d3d_amd import _lib
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = torch.device("cuda:0")
    _lib.lib()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        torch.manual_seed(1)
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

        mk = lambda: {"source": make_batch(5, 2, "nuscenes", (96, 128), device=dev), "target": make_batch(6, 2, "nuscenes", (96, 128), device=dev)}
        loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {}}])
        tk = dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, gc_freeze=False)
        monkeypatch.setenv("MM_DDP_FORCE", "1")
        monkeypatch.delenv("MM_DDP_BN_FUSED", raising=False)
        monkeypatch.setenv("MM_DDP_OVERLAP", {False: "0", True: "1", "tail": "tail"}[overlap])
        if overlap == "tail":
            monkeypatch.delenv("MM_DDP_OVERLAP", raising=False)  # the default
        ddp = TrainModel({"2d_net": n2, "3d_net": n3}, opts(), loss, dict(tk))
        ddp.configure_optimizers()
        monkeypatch.setenv("MM_DDP_FORCE", "0")
        # the comparison trainer on ONE stream: the single-GPU default (overlap_branches=2) runs the sparse branch beside the 2D
        # branch with three-kernel sparse batch norms, whose forward statistics differ in the last bit from the single-launch kernel's
        plain = TrainModel({"2d_net": n2b, "3d_net": n3b}, opts(), loss, dict(tk, overlap_branches=0))
        plain.configure_optimizers()
        assert ddp.reducer.active and ddp.reducer.overlap == overlap
        if overlap is True:
            assert ddp.reducer.bn_path == "forward single-launch, backward three-kernel"
            assert ddp.handle.get(_lib.OPT_BN2D_FUSED) == 1 and ddp.handle.get(_lib.OPT_BN3D_FUSED) == 1
        else:
            assert "single-launch in both directions" in ddp.reducer.bn_path
            assert ddp.handle.get(_lib.OPT_BN2D_FUSED) == 3 and ddp.handle.get(_lib.OPT_BN3D_FUSED) == 3
        assert plain.handle.get(_lib.OPT_BN2D_FUSED) == 3

        def trap(name, orig=None):
            def boom(*a, **k):
                if orig is not None and not (a and isinstance(a[0], torch.Tensor) and a[0].is_cuda):
                    return orig(*a, **k)  # a host tensor: no device involved
                raise AssertionError(f"host synchronisation inside fit_step: {name}")
            return boom

        losses = []
        for step in range(6):
            batch_a, batch_b = mk(), mk()
            if step >= 2:
                with monkeypatch.context() as mp:
                    for name in ("item", "cpu", "tolist", "__bool__", "numpy"):
                        mp.setattr(torch.Tensor, name, trap("Tensor." + name, getattr(torch.Tensor, name)))
                    mp.setattr(torch.cuda, "synchronize", trap("torch.cuda.synchronize"))
                    mp.setattr(torch.cuda.Stream, "synchronize", trap("Stream.synchronize"))
                    la = ddp.fit_step(batch_a)
            else:
                la = ddp.fit_step(batch_a)
            lb = plain.fit_step(batch_b)
            torch.cuda.synchronize()
            losses.append((float(la), float(lb)))
        for step, (a, b) in enumerate(losses):
            assert (abs(a - b) <= 2e-3 * abs(b)) if overlap is True else (a == b), (step, a, b)
        st = ddp.reducer.stats
        if overlap is False:
            assert st["early"] == 0 and st["buckets"] == len(ddp.reducer.order)
        if overlap == "tail":
            # grid-barrier kernels were counted, and all buckets but (at most) the one that holds the stems' parameters - it completes
            # with the very last weight gradient of backward - left before finish()
            assert st["barrier_kernels_bwd"] > 20 and st["buckets"] == len(ddp.reducer.order)
            assert st["early"] >= st["buckets"] - 1 >= 1, st
        assert ddp.reducer.drain_flag() is False
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True])
def test_the_rccl_step_keeps_the_one_gpu_step_s_side_stream_and_graphs(monkeypatch, graph):
    """the step under the reducer is the one-GPU step.  The next batch's rulebooks are built on the side stream
    beside the 3D backward (on by default while every rank has a GPU of its own), and with ``ddp_graph`` the 2D trunk is replayed as
    its two HIP graphs; both give the plain trainer's losses and parameters bit for bit, and under the default "tail" schedule
    every bucket still leaves before finish()."""
    import copy

    from mm2d3d_amd import _lib, graph2d
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.synthetic import make_batch
    from mm2d3d_amd.train import TrainModel

    dev = torch.device("cuda:0")
    _lib.lib()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
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

        def batches():
            return [{"source": make_batch(5 + i, 2, "nuscenes", (96, 128), device=dev), "target": make_batch(40 + i, 2, "nuscenes", (96, 128), device=dev)}
                    for i in range(7)]

        loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {}}])
        tk = dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, gc_freeze=False, overlap_branches=0)
        monkeypatch.setenv("MM_DDP_FORCE", "1")
        monkeypatch.delenv("MM_DDP_OVERLAP", raising=False)
        monkeypatch.delenv("MM_DDP_META_SIDE", raising=False)
        ddp = TrainModel({"2d_net": n2, "3d_net": n3}, opts(), loss, dict(tk, ddp_graph=graph))
        ddp.configure_optimizers()
        monkeypatch.setenv("MM_DDP_FORCE", "0")
        plain = TrainModel({"2d_net": n2b, "3d_net": n3b}, opts(), loss, dict(tk))
        plain.configure_optimizers()
        assert ddp.reducer.active and ddp.reducer.overlap == "tail" and ddp.ddp_side_stream and not plain.reducer.active
        ba, bb = batches(), batches()
        for step in range(6):
            la = ddp.fit_step(ba[step], next_batch=ba[step + 1])
            lb = plain.fit_step(bb[step], next_batch=bb[step + 1])
            torch.cuda.synchronize()
            assert float(la) == float(lb), (step, float(la), float(lb))
        assert ddp._meta_stream is not None, "the rulebooks of the next batch were not built on the side stream"
        st2 = graph2d._STATE.get(id(n2))
        assert bool(st2 and st2["graphs"]) == graph, "HIP graphs of the 2D trunk under the reducer: only with ddp_graph"
        st = ddp.reducer.stats
        assert st["buckets"] == len(ddp.reducer.order) and st["early"] >= st["buckets"] - 1 >= 1, st
        assert ddp.reducer.drain_flag() is False
        for a, b in zip(ddp.optimizers, plain.optimizers):
            for x, y in zip(a._arenas, b._arenas):
                if x is not None:
                    assert torch.equal(x["p"], y["p"]), "parameters after 6 optimiser steps differ"
    finally:
        dist.destroy_process_group()
        from mm2d3d_amd import conv2d as _c2d

        _c2d.WGRAD_BATCH[0] = True


def test_a_flagged_step_is_skipped_on_the_device_and_the_relearn_step_does_not_raise_again(monkeypatch):
    """on RCCL the collective "graph changed" flag is read one step late, so the step in which a parameter
    learned as unused fires (its bucket is not reduced) used to go through the optimiser on every rank.  Now the optimiser kernels
    test the flag ON THE DEVICE (``skip_words``): the flagged step leaves every weight untouched, the RuntimeError arrives one step
    later, the re-learn step that follows is reduced correctly and does NOT raise a second time for the old flag, and
    ``drain_flag`` reports a flagged last step.  Both update forms: FlatAdamW.step(skip_words=) and GradScaler.step_all."""
    import torch.nn as nn

    from mm2d3d_amd import _lib
    from mm2d3d_amd.amp import GradScaler
    from mm2d3d_amd.ddp import GradAllReducer
    from mm2d3d_amd.optimizers import FlatAdamW

    dev = torch.device("cuda:0")
    _lib.lib()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1)
    try:
        for scaled in (False, True):
            torch.manual_seed(0)
            net = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4)).to(dev)
            extra = nn.Linear(4, 4).to(dev)  # unused at first: the find_unused_parameters case
            opt = FlatAdamW(list(net.parameters()) + list(extra.parameters()), lr=1e-2)
            red = GradAllReducer([opt], bucket_bytes=200, force=True, overlap=False)
            scaler = GradScaler(dev, init_scale=64.0) if scaled else None
            x = torch.randn(5, 8, device=dev)

            def run(use_extra):
                opt.zero_grad()
                y = net(x)
                loss = (y ** 2).sum() + (extra(y).sum() if use_extra else 0.0)
                (scaler.scale(loss) if scaled else loss).backward()
                red.finish()  # may raise
                if scaled:
                    scaler.step_all([opt], red.grad_scale, skip_words=red.skip_words())
                    scaler.update()
                else:
                    opt.step(grad_scale=red.grad_scale, skip_words=red.skip_words())

            snap = lambda: opt._arenas[0]["p"].detach().clone()
            run(False), run(False)
            assert red.learned and len(red.unused) == 2
            w0 = snap()
            run(False)
            w1 = snap()
            assert not torch.equal(w0, w1)  # an ordinary step moves the weights
            run(True)  # the graph changes: flagged on the device, the host does not know yet
            assert torch.equal(snap(), w1), "the flagged step was applied"
            with pytest.raises(RuntimeError, match="unused"):
                run(True)  # one step late: raises in finish(), before this step's update is queued
            assert not red.learned and torch.equal(snap(), w1)
            run(True)  # the re-learn step: reduced in full, must not raise for the (void) flag of the raising step
            w2 = snap()
            assert red.learned and len(red.unused) == 0 and not torch.equal(w2, w1)
            run(True)
            assert red.drain_flag() is False
            if scaled:
                assert scaler.steps_taken(opt) == 5 and scaler.get_scale() == 64.0  # the flag vetoes the step, not the loss scale
            # a flagged LAST step is found by drain_flag (TrainModel.checkpoint raises on it)
            extra2 = nn.Linear(4, 4).to(dev)
            del extra2
    finally:
        dist.destroy_process_group()


def test_grad_scaler_step_all_takes_one_decision_for_every_optimiser():
    """the reference's HybridOptim is ONE optimiser to Lightning's GradScaler (train.py:627-636) - a non-finite gradient in
    the 2D network must skip the 3D network's update too (and vice versa); ``step`` keeps torch's per-optimiser form."""
    from mm2d3d_amd.amp import GradScaler
    from mm2d3d_amd.optimizers import FlatAdamW

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    pa = [torch.nn.Parameter(torch.randn(100, device=dev))]
    pb = [torch.nn.Parameter(torch.randn(50, device=dev))]
    oa, ob = FlatAdamW(pa, lr=1e-2), FlatAdamW(pb, lr=1e-2)
    sc = GradScaler(dev, init_scale=8.0)

    def load(poison):
        for o, ps in ((oa, pa), (ob, pb)):
            o.zero_grad()
            ps[0].grad.copy_(torch.ones_like(ps[0]) * 8.0)
            o.mark_all_touched()
        if poison:
            pa[0].grad[3] = float("inf")

    a0, b0 = pa[0].detach().clone(), pb[0].detach().clone()
    load(True)
    sc.step_all([oa, ob])
    sc.update()
    assert torch.equal(pa[0], a0) and torch.equal(pb[0], b0), "an overflow in one optimiser must skip both"
    assert sc.get_scale() == 4.0 and sc.steps_taken(oa) == 0 and sc.steps_taken(ob) == 0
    load(False)
    sc.step_all([oa, ob])
    sc.update()
    assert not torch.equal(pa[0], a0) and not torch.equal(pb[0], b0)
    assert sc.steps_taken(oa) == 1 and sc.steps_taken(ob) == 1
    # an extra skip word vetoes the step without touching the loss scale
    a1, b1 = pa[0].detach().clone(), pb[0].detach().clone()
    load(False)
    sc.step_all([oa, ob], skip_words=torch.tensor([0, 1], dtype=torch.int32, device=dev))
    sc.update()
    assert torch.equal(pa[0], a1) and torch.equal(pb[0], b1) and sc.get_scale() == 4.0
    # the plain update form takes the same words
    load(False)
    oa.step(grad_scale=1.0, skip_words=torch.tensor([1], dtype=torch.int32, device=dev))
    assert torch.equal(pa[0], a1)
