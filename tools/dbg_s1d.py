"""development: conv_s1d.hip's weight gradient against fp64 on the bf16-rounded operands, error by (tap, ci, co) region."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F
F.set_conv_arith("bf16")
r16 = lambda t: t.bfloat16().double()
for (n, cin, cout, L, k) in [(4, 129, 64, 431, 3), (2, 100, 100, 300, 1), (128, 476, 476, 3, 3), (4, 64, 64, 40, 3), (4, 64, 64, 32, 3), (1, 64, 64, 64, 3), (1, 64, 64, 64, 1), (2, 64, 64, 33, 1)]:
    torch.manual_seed(1)
    x = torch.randn(n, cin, 1, L); gy = torch.randn(n, cout, 1, L)
    d = F._desc(n, cin, cout, 1, L, 1, k)
    dw = F.conv_wgrad(x.cuda(), gy.cuda(), (cout, cin, 1, k)).cpu().double()
    ref = torch.nn.grad.conv2d_weight(r16(x), (cout, cin, 1, k), r16(gy), padding=(0, k // 2))
    err = (dw - ref).abs()
    print((n, cin, cout, L, k), F.plan_name(d, 2), "max err %.3e of %.3e" % (float(err.max()), float(ref.abs().max())),
          "per tap", [float(err[..., t].max()) for t in range(k)])
    # which (ci) rows / (co) cols are bad
    bad = err > 1e-3 * float(ref.abs().max())
    print("   bad fraction %.4f; bad per tap %s; bad ci range %s; bad co range %s" % (
        float(bad.double().mean()), [int(bad[..., t].sum()) for t in range(k)],
        (int(bad.any(0).any(-1).any(-1).nonzero().min()) if bad.any() else -1, int(bad.any(0).any(-1).any(-1).nonzero().max()) if bad.any() else -1),
        (int(bad.any(1).any(-1).any(-1).nonzero().min()) if bad.any() else -1, int(bad.any(1).any(-1).any(-1).nonzero().max()) if bad.any() else -1)))
