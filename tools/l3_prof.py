"""Development: phase breakdown of the three-limb forward kernel (library built with -DFSC_L16_PROFILE, see conv_l3.hip).
    python tools/l3_prof.py [shape names]      (shapes of tools/conv_bench.py)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402
from tools.conv_bench import SHAPES  # noqa: E402

lib = F._lib.load()
HAVE_FWD = hasattr(lib, "fsc_debug_l3_prof")
if HAVE_FWD:
    prof = lib.fsc_debug_l3_prof
    prof.argtypes = [C.c_void_p]
buf = (C.c_ulonglong * 64)()
ARITH = int(os.environ.get("L3_ARITH", "9"))        # 9 (bf16x9) or 10 (f16x6)
F.set_conv_arith(ARITH)
dev = torch.device("cuda")
names = sys.argv[1:] or ["b0c2", "b1e", "b1c2", "b2c2", "b3c2", "b0c1"]
for name in (names if HAVE_FWD else []):
    n, cin, cout, h, w, k = SHAPES[name]
    torch.manual_seed(1)
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, device=dev)
    t = F.l16_pack(x)
    pp = F.conv_l16_pack(wt, n, h, w, False)
    d = F._desc(n, cin, cout, h, w, k, k, ARITH)
    for _ in range(2):
        F.conv_l16(t, wt, bias, prepacked=pp)
    prof(buf)
    iters = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        F.conv_l16(t, wt, bias, prepacked=pp)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    prof(buf)
    v = [[buf[wv * 8 + i] / iters for i in range(8)] for wv in range(8)]
    fl = 2.0 * n * h * w * cin * cout * k * k
    print("%s %s %s: %.3f ms %.1f TF fp32-eq; shader clock %.0f MHz" % (
        name, SHAPES[name], F.l16_plan_name(d, 0), ms, fl / ms / 1e9, 100.0 * v[0][5] / max(v[0][6], 1)))
    for wv in range(8):
        a = v[wv]
        steps = max(a[4], 1)
        print("   wave %d: kernel %.0f kcyc | per step: wait+barrier %.0f, copy issue %.0f, MFMA steps %.0f | epilogue %.1f%%, first fragments %.1f%% of kernel, "
              "steps %.0f, accounted %.1f%%" % (wv, a[5] / 1e3, a[7] / steps, a[0] / steps, a[1] / steps, 100 * a[2] / a[5], 100 * a[3] / a[5], steps,
                                               100 * (a[0] + a[1] + a[2] + a[3] + a[7]) / a[5]))


# ---- weight gradient on three limbs (the same library build instruments conv_l16_wgrad_kernel)
if hasattr(lib, "fsc_debug_l16w_prof"):
    wprof = lib.fsc_debug_l16w_prof
    wprof.argtypes = [C.c_void_p]
    for name in names:
        n, cin, cout, h, w, k = SHAPES[name]
        d = F._desc(n, cin, cout, h, w, k, k, ARITH)
        if not F.conv_l16_wgrad_supported(d):
            continue
        torch.manual_seed(1)
        x = torch.randn(n, cin, h, w, device=dev)
        gy = torch.randn(n, cout, h, w, device=dev)
        x16, g16 = F.l16_pack(x), F.l16_pack(gy)
        del x, gy
        for _ in range(2):
            F.conv_l16_wgrad(x16, g16, (cout, cin, k, k))
        wprof(buf)
        iters = 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            F.conv_l16_wgrad(x16, g16, (cout, cin, k, k))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        wprof(buf)
        fl = 2.0 * n * h * w * cin * cout * k * k
        print("%s wgrad %s: %.3f ms %.1f TF fp32-eq" % (name, F.l16_wgrad_plan_name(d), ms, fl / ms / 1e9))
        for wv in range(8):
            a = [buf[wv * 8 + i] / iters for i in range(5)]
            units = max(a[3], 1)
            print("   wave %d: kernel %.0f kcyc | per unit: wait + barrier %.0f, copy issue %.0f, MFMA k-steps %.0f | units %.0f, accounted %.1f%%"
                  % (wv, a[4] / 1e3, a[0] / units, a[1] / units, a[2] / units, units, 100 * (a[0] + a[1] + a[2]) / max(a[4], 1)))
