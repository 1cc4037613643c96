"""development: which tiles of conv_fwd_x3_kernel go wrong when a worker runs several items and c_in has a remainder chunk"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F
import torch.nn.functional as TF
dev = torch.device('cuda')
def rnd(t): return t.to(torch.bfloat16).float()
for (n, cin, cout, L, k) in [(256, 129, 64, 512, 3), (384, 129, 64, 512, 3), (256, 100, 64, 512, 3), (256, 129, 64, 512, 1), (256, 160, 64, 512, 3), (256, 129, 64, 256, 3)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, L, generator=g); w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** .5; b = torch.randn(cout, generator=g)
    ref = TF.conv1d(rnd(x).double(), rnd(w).double(), b.double(), padding=k // 2)
    F.set_conv_arith(1)
    y = F.conv_forward(x.to(dev).unsqueeze(2), w.to(dev).unsqueeze(2), b.to(dev)).squeeze(2).cpu()
    F.set_conv_arith(None)
    d = (y.double() - ref).abs()
    bad_img = (d.amax((1, 2)) > 1e-2)
    tiles = d.reshape(n, cout, -1, 256).amax((1, 3)) > 1e-2 if L % 256 == 0 else None
    buf = F.plan_name(F._desc(n, cin, cout, 1, L, 1, k, 1), 0)
    print((n, cin, cout, L, k), buf, 'bad images %d / %d' % (int(bad_img.sum()), n), 'bad tiles:', None if tiles is None else tiles.flatten().nonzero().flatten().tolist()[:24], 'of', None if tiles is None else tiles.numel())
    bad_ch = (d.amax((0, 2)) > 1e-2).nonzero().flatten().tolist()
    print('    bad channels', bad_ch[:20], 'max err', float(d.max()))
