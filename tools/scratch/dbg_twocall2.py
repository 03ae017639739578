import copy, os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_graph2d import _trainer, _nets
from mm2d3d_amd import graph2d
from mm2d3d_amd.synthetic import make_batch
dev = torch.device("cuda:0")
n2, n3 = _nets(dev)
mk = lambda i: {"source": make_batch(5, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3)),
                "target": make_batch(6, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3))}
def run(g, nsteps, side=None):
    graph2d.ENABLED[0] = g
    a, b = copy.deepcopy(n2), copy.deepcopy(n3)
    t = _trainer(dev, a, b, 0.0); t.joint_domains = False
    if side is not None: os.environ["MM_META_SIDE"] = side
    ls = [float(t.fit_step(mk(i)).detach()) for i in range(nsteps)]
    graph2d.reset()
    return t, ls
def diff(ta, tb, tag):
    out = []
    for net in ("2d_net", "3d_net"):
        pa, pb = dict(ta.model[net].named_parameters()), dict(tb.model[net].named_parameters())
        bad = [n for n in pa if pa[n].grad is not None and not torch.equal(pa[n].grad, pb[n].grad)]
        out.append((net, len(bad), len(pa)))
    print(tag, out)
for n in (1, 2, 3):
    e1, l1 = run(False, n); e2, l2 = run(False, n); g1, l3 = run(True, n)
    print("steps", n, "losses eager/eager/graph", l1, l2, l3)
    diff(e1, e2, " eager vs eager")
    diff(e1, g1, " eager vs graph")
