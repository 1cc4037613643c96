"""Parity and timing of the L16 conv kernels against the fp32-input f16x3 kernels (development tool).

    python tools/l16_check.py [--iters 10] [shape names ...]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402
from tools.conv_bench import SHAPES  # noqa: E402


def timeit(fn, iters):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--wgrad", action="store_true")
    a = ap.parse_args()
    F.set_conv_arith("f16x3")
    dev = torch.device("cuda")
    names = a.names or [k for k in SHAPES if k != "b0e"]
    for name in names:
        n, cin, cout, h, w, k = SHAPES[name]
        if a.n:
            n = a.n
        torch.manual_seed(1)
        x = torch.randn(n, cin, h, w, device=dev)
        wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        bias = torch.randn(cout, device=dev)
        gy = torch.randn(n, cout, h, w, device=dev)
        d = F._desc(n, cin, cout, h, w, k, k, 3)
        if a.wgrad:
            if not F.conv_l16_wgrad_supported(d):
                print("%-5s wgrad unsupported" % name, flush=True)
                continue
            xa, ga = F.amax(x), F.amax(gy)
            x16, g16 = F.l16_pack(x, xa), F.l16_pack(gy, ga)
            ref = F.conv_wgrad(x, gy, wt.shape, x_amax=xa, dout_amax=ga)
            got = F.conv_l16_wgrad(x16, g16, wt.shape)
            torch.cuda.synchronize()
            diff = (ref - got).abs().max().item()
            ms_old = timeit(lambda: F.conv_wgrad(x, gy, wt.shape, x_amax=xa, dout_amax=ga), a.iters)
            ms_new = timeit(lambda: F.conv_l16_wgrad(x16, g16, wt.shape), a.iters)
            fl = 2.0 * n * h * w * cin * cout * k * k
            print("%-5s wgrad %-34s max|diff| %.3e (max|ref| %.3e) | old %-28s %7.3f ms %6.1f TF | new %7.3f ms %6.1f TF  x%.2f"
                  % (name, F.l16_wgrad_plan_name(d), diff, ref.abs().max().item(), F.plan_name(d, 2), ms_old, fl / ms_old / 1e9,
                     ms_new, fl / ms_new / 1e9, ms_old / ms_new), flush=True)
            continue
        for dgrad in (False, True):
            d = F._desc(n, cin, cout, h, w, k, k, 3)
            if not F.conv_l16_supported(d, int(dgrad)):
                print("%-5s %-5s unsupported" % (name, "dgrad" if dgrad else "fwd"), flush=True)
                continue
            src = gy if dgrad else x
            am = F.amax(src)
            t = F.l16_pack(src, am)
            if dgrad:
                ref = F.conv_dgrad(gy, wt, x.shape, dout_amax=am)
                got = F.conv_l16(t, wt, None, dgrad=True)
                f_old = lambda: F.conv_dgrad(gy, wt, x.shape, dout_amax=am)
                pp = F.conv_l16_pack(wt, n, h, w, True)
                f_new = lambda: F.conv_l16(t, wt, None, dgrad=True, prepacked=pp)
            else:
                ref = F.conv_forward(x, wt, bias, x_amax=am)
                got = F.conv_l16(t, wt, bias)
                f_old = lambda: F.conv_forward(x, wt, bias, x_amax=am)
                pp = F.conv_l16_pack(wt, n, h, w, False)
                f_new = lambda: F.conv_l16(t, wt, bias, prepacked=pp)
            torch.cuda.synchronize()
            diff = (ref - got).abs().max().item()
            nbad = int((ref != got).sum().item())
            rt = (F.l16_unpack(t) - src).abs().max().item()
            ms_old, ms_new = timeit(f_old, a.iters), timeit(f_new, a.iters)
            fl = 2.0 * n * h * w * cin * cout * k * k
            print("%-5s %-5s %-34s max|diff| %.3e (%d differ) roundtrip %.2e | old %7.3f ms %6.1f TF | new %7.3f ms %6.1f TF  x%.2f"
                  % (name, "dgrad" if dgrad else "fwd", F.l16_plan_name(d, int(dgrad)), diff, nbad, rt, ms_old, fl / ms_old / 1e9,
                     ms_new, fl / ms_new / 1e9, ms_old / ms_new), flush=True)


if __name__ == "__main__":
    main()
