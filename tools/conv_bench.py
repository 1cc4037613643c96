"""Micro-benchmark of the conv kernels at benchmark shapes (development tool).

    python tools/conv_bench.py [fwd|dgrad|wgrad|all] [--shape n,cin,cout,h,w,k] [--iters 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402

SHAPES = {
    "b0e": (128, 2, 100, 128, 431, 3), "b0c1": (128, 100, 100, 64, 215, 1), "b0c2": (128, 100, 100, 64, 215, 3),
    "b1e": (128, 100, 150, 64, 215, 3), "b1c2": (128, 150, 150, 32, 107, 3), "b2e": (128, 150, 225, 32, 107, 3),
    "b2c2": (128, 225, 225, 16, 53, 3), "b3e": (128, 225, 337, 16, 53, 3), "b3c2": (128, 337, 337, 8, 26, 3),
    "b4e": (128, 337, 506, 8, 26, 3), "b4c2": (128, 506, 506, 4, 13, 3), "b5e": (128, 506, 759, 4, 13, 3),
    "b5c2": (128, 759, 759, 2, 6, 3), "b5c1": (128, 759, 759, 2, 6, 1),
    "t128": (128, 128, 128, 64, 215, 3), "t96": (128, 96, 96, 64, 215, 3),      # (development: 8 + 8 and 6 + 6 tiles -- balanced blocks)
}


def run(kind, shape, iters):
    n, cin, cout, h, w, k = shape
    dev = torch.device("cuda")
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    gy = torch.randn(n, cout, h, w, device=dev)
    xa = F.amax(x) if F.get_conv_arith() == 3 else None      # producers report it in the model path
    ga = F.amax(gy) if F.get_conv_arith() == 3 else None
    fn = {"fwd": lambda: F.conv_forward(x, wt, None, x_amax=xa), "dgrad": lambda: F.conv_dgrad(gy, wt, x.shape, dout_amax=ga),
          "wgrad": lambda: F.conv_wgrad(x, gy, wt.shape, x_amax=xa, dout_amax=ga)}[kind]
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * n * h * w * cin * cout * k * k
    import ctypes as C
    d = F._desc(n, cin, cout, h, w, k, k)
    name = F.plan_name(d, {"fwd": 0, "dgrad": 1, "wgrad": 2}[kind])
    print("%-6s %-28s %-30s %8.3f ms  %7.2f TFLOP/s" % (kind, shape, name, ms, flops / ms / 1e9), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", nargs="?", default="all")
    ap.add_argument("--shape", default=None)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--arith", default=None, help="f32 | f16x3 | bf16x6 | bf16x9")
    a = ap.parse_args()
    if a.arith:
        F.set_conv_arith(a.arith)
    kinds = ["fwd", "dgrad", "wgrad"] if a.kind == "all" else [a.kind]
    if a.shape in SHAPES:
        shapes = [SHAPES[a.shape]]
    elif a.shape:
        shapes = [tuple(int(v) for v in a.shape.split(","))]
    else:
        shapes = list(SHAPES.values())
    for s in shapes:
        for kd in kinds:
            run(kd, s, a.iters)
