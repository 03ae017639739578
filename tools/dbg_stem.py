import os, sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from mm2d3d_amd import scn
from mm2d3d_amd.synthetic import make_batch
from oracle import scn_ref
dev = torch.device('cuda:0')
torch.manual_seed(0)
b = make_batch(1, 2, "nuscenes", img_hw=(32, 48))
coords, feats = b["x"]
ri, rs = scn_ref.InputLayer(3, 4096, 4), scn_ref.SubmanifoldConvolution(3, 3, 16, 3, False)
hi, hs = scn.InputLayer(3, 4096, 4), scn.SubmanifoldConvolution(3, 3, 16, 3, False).to(dev)
hs.load_state_dict(rs.state_dict())
yr = rs(ri([coords, feats])).features
fh = feats.to(dev).requires_grad_(True)
yh = hs(hi([coords, fh])).features
d = (yh.detach().cpu() - yr.detach()).abs()
print('mode', os.environ.get('MM_NO_NARROW'), 'max abs diff', d.max().item(), 'rows with diff>1e-5:', int((d.max(1).values > 1e-5).sum()), 'of', len(d))
g = torch.randn_like(yr)
(yh * g.to(dev)).sum().backward()
fr = feats.clone().requires_grad_(True)
(rs(ri([coords, fr])).features * g).sum().backward()
print('dfeats max diff', (fh.grad.cpu() - fr.grad).abs().max().item(), 'dW diff', (hs.weight.grad.cpu() - rs.weight.grad).abs().max().item())
