"""Per-layer timing of the cfg-3 convolutions (BASELINE.json configs[2]: 10 blocks, 64 x 1.25^k channels, rows of 3446 ... 3 frames,
batch 128, bf16) in forward / input gradient / weight gradient.  Run once with FSC_S1D_MAXPX=0 (the round-1 kernels of conv.hip
everywhere) and once without (conv_s1d.hip on the small layers): the environment is read once per process.
    python tools/s1d_bench.py [--from-block 2]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402


def layers():
    out, length, c_in = [], 3446, 129
    for k, depth in enumerate(int(1.25 ** k * 64) for k in range(10)):
        out.append(("b%de" % k, c_in, depth, length, 3))
        length //= 2
        out += [("b%dc1" % k, depth, depth, length, 1), ("b%dc2" % k, depth, depth, length, 3)]
        c_in = depth
    return out


def timed(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


ap = argparse.ArgumentParser()
ap.add_argument("--from-block", type=int, default=0)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
F.set_conv_arith("bf16")
dev = torch.device("cuda")
tot = [0.0, 0.0, 0.0]
print("FSC_S1D_MAXPX =", os.environ.get("FSC_S1D_MAXPX", "(default)"))
for name, cin, cout, length, k in layers():
    if int(name[1]) < a.from_block:
        continue
    x = torch.randn(128, cin, 1, length, device=dev)
    wt = torch.randn(cout, cin, 1, k, device=dev) / (cin * k) ** 0.5
    gy = torch.randn(128, cout, 1, length, device=dev)
    d = F._desc(128, cin, cout, 1, length, 1, k)
    pf = F.conv_pack(d, wt, 0); pd = F.conv_pack(d, wt, 1)
    out = torch.empty(128, cout, 1, length, device=dev); dx = torch.empty_like(x)
    import ctypes as C
    f = lambda: F.call("fsc_conv_fwd", C.byref(d), F.ptr(x), F.ptr(pf), None, 0, 0, F.ptr(out), None, F.stream_ptr())
    g = lambda: F.call("fsc_conv_fwd", C.byref(d), F.ptr(gy), F.ptr(pd), None, 1, 0, F.ptr(dx), None, F.stream_ptr())
    w = lambda: F.conv_wgrad(x, gy, wt.shape)
    t = [timed(f, a.iters), timed(g, a.iters), timed(w, a.iters)]
    for i in range(3):
        tot[i] += t[i]
    mb = 128 * length * 4 * (cin + cout) / 1e6
    print("%-5s %3d->%3d L%-4d k%d  %6.1f MB | fwd %6.1f us %-26s | dgrad %6.1f us | wgrad(+reduce) %6.1f us %s" % (
        name, cin, cout, length, k, mb, t[0], F.plan_name(d, 0), t[1], t[2], F.plan_name(d, 2)), flush=True)
print("sum: fwd %.1f us, dgrad %.1f us, wgrad %.1f us" % tuple(tot))
