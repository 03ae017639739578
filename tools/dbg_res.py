import os, sys, torch
sys.path.insert(0, '/root/repo')
from mm2d3d_amd.net3d import Net3DSeg
from mm2d3d_amd.synthetic import make_batch
dev = torch.device('cuda:0')
torch.manual_seed(0)
b = make_batch(1, 2, "nuscenes", img_hw=(32, 48))
kw = dict(in_channels=3, m=16, full_scale=4096, num_planes=7, residual_blocks=True)
net = Net3DSeg(6, True, kw).to(dev)
coords, feats = b["x"]
p, f, a = net({"x": [coords.to(dev), feats.clone().to(dev)]})
w = torch.randn(p["seg_logit"].shape, generator=torch.Generator().manual_seed(1)).to(dev)
((p["seg_logit"] * w).sum() + (a["seg_logit_point"] * w).sum()).backward()
out = {"logit": p["seg_logit"].detach().cpu(), "feat": f.detach().cpu()}
for n, q in net.named_parameters():
    if q.grad is not None: out["g." + n] = q.grad.detach().cpu()
torch.save(out, sys.argv[1])
