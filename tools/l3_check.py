"""Parity and timing of the three-limb (bf16x9) pre-split conv kernels against the fp32-input bf16x9 kernels and fp64 (development tool).

    python tools/l3_check.py [--iters 10] [--n N] [--wgrad] [--f64] [shape names ...]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402
from tools.conv_bench import SHAPES  # noqa: E402
from tools.l16_check import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("names", nargs="*")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--f64", action="store_true", help="also compare with an fp64 convolution on the CPU (use a small --n)")
    ap.add_argument("--arith", type=int, default=9, help="9 (default), 8 / 6 (development builds with -DFSC_L3_ALL_PRODS), 10 (f16x6)")
    a = ap.parse_args()
    F.set_conv_arith("f16x6" if a.arith == 10 else "bf16x9")
    if a.arith not in (9, 10):
        F._l16_arith = lambda: a.arith
        F._l16_limbs = lambda arith=None: 3
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
        d = F._desc(n, cin, cout, h, w, k, k, a.arith)
        fl = 2.0 * n * h * w * cin * cout * k * k
        if a.wgrad:
            if not F.conv_l16_wgrad_supported(d):
                print("%-5s wgrad unsupported" % name, flush=True)
                continue
            x16, g16 = F.l16_pack(x), F.l16_pack(gy)
            F.USE_L16 = False
            ref = F.conv_wgrad(x, gy, wt.shape)
            F.USE_L16 = True
            got = F.conv_l16_wgrad(x16, g16, wt.shape)
            torch.cuda.synchronize()
            diff = (ref - got).abs().max().item()
            extra = ""
            if a.f64:
                r64 = torch.nn.grad.conv2d_weight(x.double().cpu(), wt.shape, gy.double().cpu(), padding=k // 2)
                extra = " | vs f64: new %.3e old %.3e" % ((got.double().cpu() - r64).abs().max().item(), (ref.double().cpu() - r64).abs().max().item())
            ms_old = timeit(lambda: F.conv_wgrad(x, gy, wt.shape), a.iters)
            ms_new = timeit(lambda: F.conv_l16_wgrad(x16, g16, wt.shape), a.iters)
            print("%-5s wgrad %-34s max|diff| %.3e (max|ref| %.3e)%s | old %-28s %7.3f ms %6.1f TF | new %7.3f ms %6.1f TF  x%.2f"
                  % (name, F.l16_wgrad_plan_name(d), diff, ref.abs().max().item(), extra, F.plan_name(d, 2), ms_old, fl / ms_old / 1e9,
                     ms_new, fl / ms_new / 1e9, ms_old / ms_new), flush=True)
            continue
        for dgrad in (False, True):
            if not F.conv_l16_supported(d, int(dgrad)):
                print("%-5s %-5s unsupported" % (name, "dgrad" if dgrad else "fwd"), flush=True)
                continue
            src = gy if dgrad else x
            t = F.l16_pack(src)
            rt = (F.l16_unpack(t) - src).abs().max().item()
            if dgrad:
                ref = F.conv_dgrad(gy, wt, x.shape)
                got = F.conv_l16(t, wt, None, dgrad=True)
                f_old = lambda: F.conv_dgrad(gy, wt, x.shape)
                pp = F.conv_l16_pack(wt, n, h, w, True)
                f_new = lambda: F.conv_l16(t, wt, None, dgrad=True, prepacked=pp)
            else:
                ref = F.conv_forward(x, wt, bias)
                got = F.conv_l16(t, wt, bias)
                f_old = lambda: F.conv_forward(x, wt, bias)
                pp = F.conv_l16_pack(wt, n, h, w, False)
                f_new = lambda: F.conv_l16(t, wt, bias, prepacked=pp)
            torch.cuda.synchronize()
            diff = (ref - got).abs().max().item()
            extra = ""
            if a.f64:
                if dgrad:
                    r64 = torch.nn.grad.conv2d_input(x.shape, wt.double().cpu(), gy.double().cpu(), padding=k // 2)
                else:
                    r64 = torch.nn.functional.conv2d(x.double().cpu(), wt.double().cpu(), bias.double().cpu(), padding=k // 2)
                extra = " | vs f64: new %.3e old %.3e" % ((got.double().cpu() - r64).abs().max().item(), (ref.double().cpu() - r64).abs().max().item())
            ms_old, ms_new = timeit(f_old, a.iters), timeit(f_new, a.iters)
            import ctypes as C
            mhz = C.c_double(0.0)
            F.call("fsc_conv_l16_last_clock", 2, C.byref(mhz))
            print("%-5s %-5s %-34s max|diff| %.3e roundtrip %.2e%s | old %7.3f ms %6.1f TF | new %7.3f ms %6.1f TF  x%.2f  clock %.0f MHz -> %.2f of bf16 peak at clock"
                  % (name, "dgrad" if dgrad else "fwd", F.l16_plan_name(d, int(dgrad)), diff, rt, extra, ms_old, fl / ms_old / 1e9,
                     ms_new, fl / ms_new / 1e9, ms_old / ms_new, mhz.value, 9 * fl / ms_new / 1e9 / (2500.0 * mhz.value / 2400.0)), flush=True)


if __name__ == "__main__":
    main()
