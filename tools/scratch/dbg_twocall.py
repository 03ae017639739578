import copy, os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from test_gpu_graph2d import _trainer, _nets
from mm2d3d_amd import graph2d
from mm2d3d_amd.synthetic import make_batch
dev = torch.device("cuda:0")
n2, n3 = _nets(dev); n2b, n3b = copy.deepcopy(n2), copy.deepcopy(n3)
mk = lambda i: {"source": make_batch(5, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3)),
                "target": make_batch(6, 2, "nuscenes", (94, 126), device=dev, first_scene=2 * (i % 3))}
graph2d.ENABLED[0] = True
ga = _trainer(dev, n2, n3, 0.0); ga.joint_domains = False
for i in range(2): ga.fit_step(mk(i))
graph2d.ENABLED[0] = False
eb = _trainer(dev, n2b, n3b, 0.0); eb.joint_domains = False
for i in range(2): eb.fit_step(mk(i))
# gradients of the last step are still in the arenas
for net in ("2d_net", "3d_net"):
    pa, pb = dict(ga.model[net].named_parameters()), dict(eb.model[net].named_parameters())
    bad = [(n, float((pa[n].grad - pb[n].grad).abs().max()), float(pb[n].grad.abs().max())) for n in pa if pa[n].grad is not None and not torch.equal(pa[n].grad, pb[n].grad)]
    print(net, "params with differing grads:", len(bad), "of", len(pa))
    for b in bad[:40]: print("   ", b)
sa, sb = ga.model.state_dict(), eb.model.state_dict()
print("buffers differing:", [k for k in sa if ("running" in k or "num_batches" in k) and not torch.equal(sa[k], sb[k])][:10])
