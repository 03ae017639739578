"""The 2D trunk as two HIP graphs (mm2d3d_amd/graph2d.py): the replayed step must be the eager step, bit for bit."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(dev, n2, n3, dropout_p):
    from mm2d3d_amd.losses import Loss
    from mm2d3d_amd.optimizers import Optimizer
    from mm2d3d_amd.train import TrainModel

    for m in n2.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = dropout_p
    opts = {}
    for k in ("2d_net", "3d_net"):
        o = Optimizer("adamw", lr=0.001)
        o.set_scheduler("one_cycle", max_lr=0.005, total_steps=100)
        opts[k] = o
    loss = Loss([{"name": "cross_entropy", "target": "segmentation", "args": {}}])
    tm = TrainModel({"2d_net": n2, "3d_net": n3}, opts, loss, dict(lambda_xm_src=1.0, lambda_xm_trg=0.1, gc_freeze=False))
    tm.configure_optimizers()
    return tm


@pytest.mark.parametrize("dropout_p", [0.0, 0.4])
def test_graphed_trunk_equals_the_eager_trunk_bit_for_bit(dropout_p, half2d):
    """Seven optimiser steps on changing batches (point counts differ per batch, image shapes do not): with the trunk captured at the
    third call and replayed from then on, losses, every parameter, the running statistics and ``num_batches_tracked`` equal the eager
    trainer's to the last bit; with dropout on (torch's graph-safe Philox offsets) the graphed run is compared with itself for
    finiteness and a falling loss only - the eager run draws its masks from other offsets."""
    from mm2d3d_amd import graph2d
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg
    from mm2d3d_amd.synthetic import make_batch

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    n2, n3 = Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
    mk = lambda i: {"source": make_batch(5, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3)),
                    "target": make_batch(6, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3))}
    was = graph2d.ENABLED[0]
    try:
        graph2d.ENABLED[0] = True
        ga = _trainer(dev, n2, n3, dropout_p)
        la = [float(ga.fit_step(mk(i)).detach()) for i in range(7)]
        st = graph2d._STATE.get(id(n2))
        assert st is not None and len(st["graphs"]) == 1, "the trunk was not captured"
        g = next(iter(st["graphs"].values()))
        assert len(g.params) > 150  # every conv / batch-norm / head parameter of the trunk that receives a gradient
        assert all(np.isfinite(la))
        if dropout_p:
            return
        graph2d.ENABLED[0] = False
        eb = _trainer(dev, n2b, n3b, dropout_p)
        lb = [float(eb.fit_step(mk(i)).detach()) for i in range(7)]
        assert la == lb, (la, lb)
        for (k, a), (_, b) in zip(ga.model.state_dict().items(), eb.model.state_dict().items()):
            assert torch.equal(a, b), k
        for oa, ob in zip(ga.optimizers, eb.optimizers):
            for x, y in zip(oa._arenas, ob._arenas):
                if x is not None:
                    assert torch.equal(x["p"], y["p"]) and torch.equal(x["m"], y["m"]) and torch.equal(x["v"], y["v"])
    finally:
        graph2d.ENABLED[0] = was
        graph2d.reset()


def _nets(dev):
    from mm2d3d_amd.net2d import Net2DSeg
    from mm2d3d_amd.net3d import Net3DSeg

    torch.manual_seed(0)
    kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7)
    return Net2DSeg(6, pretrained=False).to(dev), Net3DSeg(6, True, kw).to(dev)


def _same_state(ga, eb):
    for (k, a), (_, b) in zip(ga.model.state_dict().items(), eb.model.state_dict().items()):
        assert torch.equal(a, b), k
    for oa, ob in zip(ga.optimizers, eb.optimizers):
        for x, y in zip(oa._arenas, ob._arenas):
            if x is not None:
                assert torch.equal(x["p"], y["p"]) and torch.equal(x["m"], y["m"]) and torch.equal(x["v"], y["v"])


def test_two_call_sequence_never_replays_a_graph_whose_backward_is_pending():
    """the literal two-call sequence (``joint_domains=False``: the 2D net on the source batch, then on the target
    batch, then ONE backward) with optimisers and same-shape images.  From the third call of the shape both calls would replay one
    graph - the target replay overwriting the activations the source's backward differentiates.  A graph with a forward in flight is
    not replayed again (the second call runs eagerly): six steps equal the eager trainer's bit for bit."""
    from mm2d3d_amd import graph2d
    from mm2d3d_amd.synthetic import make_batch

    dev = torch.device("cuda:0")
    n2, n3 = _nets(dev)
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
    mk = lambda i: {"source": make_batch(5, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3)),
                    "target": make_batch(6, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3))}
    was = graph2d.ENABLED[0]
    try:
        graph2d.ENABLED[0] = True
        ga = _trainer(dev, n2, n3, 0.0)
        ga.joint_domains = False
        la = [float(ga.fit_step(mk(i)).detach()) for i in range(6)]
        st = graph2d._STATE.get(id(n2))
        assert st is not None and len(st["graphs"]) >= 1, "no graph was captured: the test does not test what it says"
        graph2d.ENABLED[0] = False
        eb = _trainer(dev, n2b, n3b, 0.0)
        eb.joint_domains = False
        lb = [float(eb.fit_step(mk(i)).detach()) for i in range(6)]
        assert la == lb, (la, lb)
        _same_state(ga, eb)
    finally:
        graph2d.ENABLED[0] = was
        graph2d.reset()


def test_captured_forward_always_repacks_the_weights():
    """the capture must not depend on whether a pack happened to be stale when it was taken.  Two training_step
    calls WITHOUT an optimiser step, then fit_step (the capture: every pack is fresh at that moment), then more fit_steps: the
    replays must multiply with the weights AdamW has updated since - losses and parameters equal the eager trainer's."""
    from mm2d3d_amd import graph2d
    from mm2d3d_amd.synthetic import make_batch

    dev = torch.device("cuda:0")
    n2, n3 = _nets(dev)
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
    mk = lambda i: {"source": make_batch(5, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3)),
                    "target": make_batch(6, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3))}

    def run(tm):
        out = []
        for i in range(2):  # forward / backward only: gradients accumulate, no optimiser step, the packs stay fresh
            for o in tm.optimizers:
                o.zero_grad()
            loss = tm.training_step(mk(i), i)
            loss.backward()
            out.append(float(loss.detach()))
        out += [float(tm.fit_step(mk(i)).detach()) for i in range(2, 7)]
        return out

    was = graph2d.ENABLED[0]
    try:
        graph2d.ENABLED[0] = True
        ga = _trainer(dev, n2, n3, 0.0)
        la = run(ga)
        st = graph2d._STATE.get(id(n2))
        assert st is not None and len(st["graphs"]) == 1, "the trunk was not captured"
        graph2d.ENABLED[0] = False
        eb = _trainer(dev, n2b, n3b, 0.0)
        lb = run(eb)
        assert la == lb, (la, lb)
        _same_state(ga, eb)
    finally:
        graph2d.ENABLED[0] = was
        graph2d.reset()


def test_deferred_weight_gradient_sums_survive_a_backward_pass_that_raised():
    """the autograd engine skips the end-of-backward callbacks of a pass that raised, so conv2d._WgBatch.flush
    never ran; FlatAdamW.zero_grad -> reset() must also re-arm the callback, else no later backward queues it and the 2D convolution
    weight gradients stay zero.  An eager step with a backward that raises, then ordinary steps: equal to a trainer that skipped it."""
    from mm2d3d_amd import conv2d as c2, graph2d
    from mm2d3d_amd.synthetic import make_batch

    dev = torch.device("cuda:0")
    n2, n3 = _nets(dev)
    n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
    mk = lambda i: {"source": make_batch(5, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3)),
                    "target": make_batch(6, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3))}

    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    was, wb = graph2d.ENABLED[0], c2.WGRAD_BATCH[0]
    try:
        graph2d.ENABLED[0] = False
        c2.WGRAD_BATCH[0] = True
        ga = _trainer(dev, n2, n3, 0.0)
        for o in ga.optimizers:
            o.zero_grad()
        loss = ga.training_step(mk(0), 0)
        # the 2D convolutions' backward (deferred slab sums queued) runs before the failing node, which sits at the loss's root... so
        # make the failure come LAST: a leaf-side node of the 3D input cannot be reached; instead fail at the root and check the flag
        with pytest.raises(RuntimeError, match="boom"):
            (_Boom.apply(loss) * 1.0).backward()
        # a backward that raised after queuing work: emulate the engine's skipped callback (what ADVICE describes) and recover
        c2._WGB.cb_queued = True
        c2._WGB.items.append(None)
        la = [float(ga.fit_step(mk(i)).detach()) for i in range(1, 4)]
        eb = _trainer(dev, n2b, n3b, 0.0)
        for o in eb.optimizers:
            o.zero_grad()
        eb.training_step(mk(0), 0)  # same forward side effects (running statistics), no backward
        lb = [float(eb.fit_step(mk(i)).detach()) for i in range(1, 4)]
        assert la == lb, (la, lb)
        _same_state(ga, eb)
        w = next(p for n, p in n2.named_parameters() if n.endswith("layer2.1.conv1.weight"))
        assert float(w.grad.abs().sum()) > 0
    finally:
        graph2d.ENABLED[0], c2.WGRAD_BATCH[0] = was, wb
        graph2d.reset()
