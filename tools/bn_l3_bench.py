"""Development: the BatchNorm / PReLU producers of the three-limb formats, time and bytes per call (cfg-2 block-0 shape).
    python tools/bn_l3_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda")
    n, c, h, w = 128, 100, 64, 215
    torch.manual_seed(0)
    x = torch.randn(n, c, h, w, device=dev)
    dy = torch.randn(n, c, h, w, device=dev)
    bn = torch.nn.BatchNorm2d(c).to(dev).train()
    alpha = torch.full((c,), 0.25, device=dev)
    el = x.numel()
    for arith in ("f16x3", "bf16x9", "f16x6"):
        F.set_conv_arith(arith)
        st = F.bn_prepare(x, bn, True)
        limbs = F._l16_limbs()
        lb = 4 if limbs == 2 else 6
        ms = timeit(lambda: F.bn_act_forward(x, st, alpha, l16=True, want_f32=False))
        print("%-7s fwd  -> L16 only      %.3f ms  %.2f TB/s" % (arith, ms, el * (4 + lb) / ms / 1e9))
        ms = timeit(lambda: F.bn_act_backward(dy, x, st, bn, alpha, want_dres=False, want_chan_sum=True, with_amax=True, l16=True, want_f32=False))
        print("%-7s bwd  -> L16 only      %.3f ms  %.2f TB/s (reduce 8 B + apply 8 + %d B per element)" % (arith, ms, el * (16 + lb) / ms / 1e9, lb))


if __name__ == "__main__":
    main()
