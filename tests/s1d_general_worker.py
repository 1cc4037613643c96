"""Worker of tests/test_s1d_general_gpu.py: run with FSC_S1D_GENERAL=1 (read once per process by conv_s1d.hip's planner)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F

out = []
for arith, cases in (("bf16x9", [(128, 759, 759, 2, 6, 3), (128, 759, 759, 2, 6, 1), (16, 100, 70, 5, 9, 3), (3, 96, 80, 1, 50, 3)]),
                     ("bf16", [(16, 100, 70, 5, 9, 3), (8, 64, 48, 4, 4, 1)])):
    F.set_conv_arith(arith)
    rnd = (lambda t: t.bfloat16().double()) if arith == "bf16" else (lambda t: t.double())
    for (n, ci, co, h, w, k) in cases:
        torch.manual_seed(n + ci)
        kh = 1 if h == 1 else k
        x = torch.randn(n, ci, h, w); wt = torch.randn(co, ci, kh, k) / (ci * kh * k) ** 0.5; b = torch.randn(co)
        gy = torch.randn(n, co, h, w)
        d = F._desc(n, ci, co, h, w, kh, k)
        y = F.conv_forward(x.cuda(), wt.cuda(), b.cuda()).cpu().double()
        gx = F.conv_dgrad(gy.cuda(), wt.cuda(), (n, ci, h, w)).cpu().double()
        pad = (kh // 2, k // 2)
        y_ref = torch.nn.functional.conv2d(rnd(x), rnd(wt), b.double(), padding=pad)
        gx_ref = torch.nn.grad.conv2d_input((n, ci, h, w), rnd(wt), rnd(gy), padding=pad)
        out.append({"arith": arith, "case": [n, ci, co, h, w, k], "fwd": F.plan_name(d, 0), "dgrad": F.plan_name(d, 1),
                    "fwd_err": float((y - y_ref).abs().max() / y_ref.abs().max()),
                    "dgrad_err": float((gx - gx_ref).abs().max() / gx_ref.abs().max())})
print(json.dumps(out))
