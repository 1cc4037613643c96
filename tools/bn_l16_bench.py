"""Timing of the BN / PReLU producers in fp32-plane and L16 form (development tool)."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


dev = torch.device("cuda")
SHAPES = [(128, 100, 64, 215), (128, 150, 32, 107), (128, 337, 8, 26)]
if len(sys.argv) > 1 and sys.argv[1] == "cfg3":          # the 1-d model's planes (odd lengths)
    SHAPES = [(128, 64, 1, 3446), (128, 64, 1, 1723), (128, 80, 1, 861), (128, 100, 1, 430), (128, 125, 1, 215), (128, 195, 1, 53)]
for (n, c, h, w) in SHAPES:
    bn = nn.BatchNorm2d(c).to(dev)
    prelu = nn.PReLU(c).to(dev)
    x = torch.randn(n, c, h, w, device=dev)
    dy = torch.randn_like(x)
    st = F.bn_prepare(x, bn, True)
    gb = x.numel() * 4 / 1e9
    rows = [
        ("stats", lambda: F.bn_prepare(x, bn, True), 1),
        ("fwd f32", lambda: F.bn_act_forward(x, st, prelu.weight, with_amax=True), 2),
        ("fwd l16+f32", lambda: F.bn_act_forward(x, st, prelu.weight, l16=True), 3),
        ("fwd l16", lambda: F.bn_act_forward(x, st, prelu.weight, l16=True, want_f32=False), 2),
        ("bwd f32", lambda: F.bn_act_backward(dy, x, st, bn, prelu.weight, with_amax=True), 5),
        ("bwd l16+f32", lambda: F.bn_act_backward(dy, x, st, bn, prelu.weight, with_amax=True, l16=True), 6),
        ("bwd l16", lambda: F.bn_act_backward(dy, x, st, bn, prelu.weight, with_amax=True, l16=True, want_f32=False), 5),
        ("bwd f32 +csum", lambda: F.bn_act_backward(dy, x, st, bn, prelu.weight, with_amax=True, want_chan_sum=True), 5),
        ("bwd l16 +csum", lambda: F.bn_act_backward(dy, x, st, bn, prelu.weight, with_amax=True, l16=True, want_f32=False, want_chan_sum=True), 5),
    ]
    if len(sys.argv) > 1 and sys.argv[1] == "cfg3":
        res = torch.randn_like(x)
        rows = [rows[0], rows[1], rows[4],
                ("fwd f32 +res", lambda: F.bn_act_forward(x, st, prelu.weight, residual=res), 3),
                ("bwd f32 +res", lambda: F.bn_act_backward(dy, x, st, bn, prelu.weight, residual=res, want_dres=True), 8)]
    for name, fn, passes in rows:
        ms = timeit(fn)
        print("%-22s %-12s %7.3f ms  %5.2f TB/s (%d tensor passes)" % ((n, c, h, w), name, ms, passes * gb / ms, passes), flush=True)
