import sys, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from mm2d3d_amd import nn2d
dev = torch.device('cuda:0'); CL = torch.channels_last
torch.manual_seed(0)
B, Hp, Wp, h, w = 1, 16, 16, 14, 12
x = torch.randn(B, 64, Hp, Wp, device=dev).bfloat16().contiguous(memory_format=CL)
c1, c2 = nn2d.Conv2d(64, 6, 1).to(dev), nn2d.Conv2d(64, 6, 1).to(dev)
xh, xr = x.clone().requires_grad_(True), x.float().requires_grad_(True)
o1, o2 = nn2d.fused_heads(xh, h, w, c1, c2)
g1, g2 = torch.randn_like(o1), torch.randn_like(o2)
(o1 * g1).sum().add((o2 * g2).sum()).backward()
pooled = F.avg_pool2d(xr[:, :, :h, :w], 5, 1, 2)
r1, r2 = F.conv2d(pooled, c1.weight, c1.bias), F.conv2d(pooled, c2.weight, c2.bias)
gw = torch.autograd.grad((r1 * g1).sum() + (r2 * g2).sum(), [xr, c1.weight])
print('dx hip', xh.grad[0, :4, 0, :4].float())
print('dx ref', gw[0][0, :4, 0, :4])
# manual: dz = box(dout)/25 ; dx = W^T dz
dout = torch.cat([g1, g2], 1)
dz = F.avg_pool2d(dout, 5, 1, 2)
Wj = torch.cat([c1.weight.reshape(6, 64), c2.weight.reshape(6, 64)], 0)
dxm = torch.einsum('bjyx,jc->bcyx', dz, Wj)
print('dx manual', dxm[0, :4, 0, :4])
print('dW hip', c1.weight.grad.reshape(6, 64)[:2, :4]); print('dW ref', gw[1].reshape(6, 64)[:2, :4])
