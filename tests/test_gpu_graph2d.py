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
