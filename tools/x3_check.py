"""Development check of the split-bf16 conv kernel against an fp64 reference (GPU)."""
import os
import sys

import torch
import torch.nn.functional as TF

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402

CASES = [
    (2, 100, 100, 16, 43, 3, 3),
    (3, 150, 150, 12, 27, 3, 3),
    (2, 48, 64, 17, 33, 3, 3),
    (2, 60, 225, 9, 30, 3, 3),
    (4, 64, 48, 1, 300, 1, 3),
    (16, 337, 100, 4, 13, 3, 3),
    (2, 32, 48, 16, 16, 3, 3),
    (3, 150, 150, 18, 27, 3, 3),
    (2, 100, 150, 9, 53, 3, 3),
    (3, 64, 48, 1, 256, 1, 3),
]
dev = torch.device("cuda")
for case in CASES:
    n, cin, cout, h, w, kh, kw = case
    torch.manual_seed(sum(case))
    x = torch.randn(n, cin, h, w) * 37.0
    wt = torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout)
    gy = torch.randn(n, cout, h, w) * 3e-6
    y64 = TF.conv2d(x.double(), wt.double(), b.double(), padding=(kh // 2, kw // 2))
    dx64 = torch.nn.grad.conv2d_input(x.shape, wt.double(), gy.double(), padding=(kh // 2, kw // 2))
    y32 = TF.conv2d(x, wt, b, padding=(kh // 2, kw // 2))
    e32 = float((y32.double() - y64).abs().max())
    for mode in (0, 3, 6, 9):
        F.set_conv_arith(mode)
        d = F._desc(n, cin, cout, h, w, kh, kw)
        name = F.plan_name(d, 0)
        got = F.conv_forward(x.to(dev), wt.to(dev), b.to(dev)).cpu()
        dx = F.conv_dgrad(gy.to(dev), wt.to(dev), x.shape).cpu()
        dw = F.conv_wgrad(x.to(dev), gy.to(dev), wt.shape).cpu()
        if mode == 0:
            dw64 = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=(kh // 2, kw // 2))
            dw32 = torch.nn.grad.conv2d_weight(x, wt.shape, gy, padding=(kh // 2, kw // 2))
            ew32 = float((dw32.double() - dw64).abs().max())
        ew = float((dw.double() - dw64).abs().max())
        print("    wgrad %-40s maxerr %.3e (torch f32 %.3e) scale %.2f" % (F.plan_name(d, 2), ew, ew32, float(dw64.abs().max())))
        ef = float((got.double() - y64).abs().max())
        ed = float((dx.double() - dx64).abs().max())
        rms = float((got.double() - y64).pow(2).mean().sqrt())
        print("%-28s mode %d %-34s fwd maxerr %.3e rms %.3e (torch f32 %.3e)  dgrad maxerr %.3e" %
              (case, mode, name, ef, rms, e32, ed), flush=True)
