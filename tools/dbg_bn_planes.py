import sys, torch, torch.nn as nn
sys.path.insert(0, ".")
from freesound_classification_amd import functional as F
DEV = torch.device("cuda:0")
for (n, c, hw) in [(37, 11, 430), (37, 11, 861), (37, 11, 860), (37, 11, 1015), (37, 11, 1019), (37, 11, 1723), (8, 11, 861), (37, 1, 861), (16, 4, 861)]:
    g = torch.Generator(device="cpu").manual_seed(hw)
    x = (torch.randn(n, c, 1, hw, generator=g) * 1.7 + 0.4).to(DEV)
    dy = torch.randn(n, c, 1, hw, generator=g).to(DEV)
    bn = nn.BatchNorm2d(c).to(DEV); prelu = nn.PReLU(c).to(DEV)
    xr = x.clone().requires_grad_(True)
    rb = nn.BatchNorm2d(c).to(DEV); rp = nn.PReLU(c).to(DEV)
    rp(rb(xr)).backward(dy)
    st = F.bn_prepare(x, bn, True)
    dx, _, dg, db, dal, cs = F.bn_act_backward(dy, x, st, bn, prelu.weight, want_chan_sum=True)
    # fp64 reference of the sums
    xd, dyd = x.double(), dy.double()
    mean = xd.mean((0, 2, 3), keepdim=True); var = xd.var((0, 2, 3), unbiased=False, keepdim=True)
    xh = (xd - mean) / torch.sqrt(var + 1e-5)
    z = xh * bn.weight.double().view(1, -1, 1, 1) + bn.bias.double().view(1, -1, 1, 1)
    dz = torch.where(z > 0, dyd, dyd * prelu.weight.double().view(1, -1, 1, 1))
    print((n, c, hw), "dx err %.2e" % (dx - xr.grad).abs().max().item(), "db err %.2e" % (db.double() - dz.sum((0, 2, 3))).abs().max().item(),
          "dg err %.2e" % (dg.double() - (dz * xh).sum((0, 2, 3))).abs().max().item(),
          "torch db err %.2e" % (rb.bias.grad.double() - dz.sum((0, 2, 3))).abs().max().item(),
          "dx vs fp64 %.2e" % (dx.double() - (bn.weight.double().view(1,-1,1,1) / torch.sqrt(var + 1e-5)) * (dz - dz.mean((0,2,3),keepdim=True) - xh * (dz*xh).mean((0,2,3),keepdim=True))).abs().max().item())
