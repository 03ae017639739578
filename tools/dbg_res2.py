import os, sys, torch
sys.path.insert(0, '/root/repo')
from mm2d3d_amd.net3d import Net3DSeg
from mm2d3d_amd import scn
from mm2d3d_amd.synthetic import make_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
b = make_batch(1, 2, "nuscenes", img_hw=(32, 48))
kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7, residual_blocks=True)
net = Net3DSeg(6, True, kw).to(dev)
acts = {}
def hook(name):
    def h(m, i, o):
        if isinstance(o, scn.SparseConvNetTensor):
            acts[name] = o.features.detach().cpu().clone()
            o.features.retain_grad() if o.features.requires_grad else None
            acts["_t." + name] = o.features
    return h
for n, m in net.named_modules():
    if n: m.register_forward_hook(hook(n))
coords, feats = b["x"]
p, f, a = net({"x": [coords.to(dev), feats.clone().to(dev)]})
w = torch.randn(p["seg_logit"].shape, generator=torch.Generator().manual_seed(1)).to(dev)
((p["seg_logit"] * w).sum() + (a["seg_logit_point"] * w).sum()).backward()
out = {k: v for k, v in acts.items() if not k.startswith("_t.")}
for k, v in acts.items():
    if k.startswith("_t.") and v.grad is not None: out["grad." + k[3:]] = v.grad.detach().cpu()
torch.save(out, sys.argv[1])
