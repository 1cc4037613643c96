"""Randomised planner sweep of the conv kernels: split arithmetic against native fp32 MFMA (development tool).

    python tools/conv_sweep.py <seed> <trials> [<min channels> <max channels>]
"""
import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from freesound_classification_amd import functional as F
DEV = "cuda:0"
def maxdiff(a, b): return float((a.double() - b.double()).abs().max())
rng = np.random.RandomState(int(sys.argv[1]))
bad = 0
for trial in range(int(sys.argv[2])):
    kind = trial % 3
    kh, kw = [(3, 3), (1, 3), (1, 1)][kind]
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (32, 300)
    cin, cout = int(rng.randint(lo, hi)), int(rng.randint(lo, hi))
    n = int(rng.randint(1, 9))
    if kh == 3: h, w = int(rng.randint(1, 40)), int(rng.randint(3, 70))
    else: h, w = 1, int(rng.randint(8, 700))
    if kind == 2 and rng.rand() < 0.5: h, w = int(rng.randint(1, 30)), int(rng.randint(1, 40))
    torch.manual_seed(trial)
    x = torch.randn(n, cin, h, w, device=DEV); wt = torch.randn(cout, cin, kh, kw, device=DEV) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout, device=DEV); gy = torch.randn(n, cout, h, w, device=DEV); base = torch.randn_like(x)
    res = {}
    try:
        for mode in (0, 3, 6):
            F.set_conv_arith(mode)
            res[mode] = (F.conv_forward(x, wt, b), F.conv_dgrad(gy, wt, x.shape), F.conv_dgrad(gy, wt, x.shape, accumulate_into=base.clone()), F.conv_wgrad(x, gy, wt.shape))
    except Exception as e:
        print("EXC", (n, cin, cout, h, w, kh, kw), e); bad += 1; continue
    tols = (4e-6 * (cin*kh*kw) ** 0.5 + 1e-5, 4e-6 * (cout*kh*kw) ** 0.5 + 1e-5, 4e-6 * (cout*kh*kw) ** 0.5 + 1e-5, 6e-6 * (n*h*w) ** 0.5 + 2e-5)
    for mode in (3, 6):
        for name, a, bb, tol in zip(("fwd", "dgrad", "dgrad+acc", "wgrad"), res[0], res[mode], tols):
            dd = maxdiff(a, bb)
            if not dd < tol:
                print("BAD", trial, mode, name, (n, cin, cout, h, w, kh, kw), dd, tol); bad += 1
print("done, bad =", bad)
