"""Development: shader clock the chip sustains under each L16 convolution of cfg 2 (fsc_conv_l16_last_clock)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F
from tools.conv_bench import SHAPES
F.set_conv_arith("f16x3")
for name in sys.argv[1:] or ["b0c2", "b1e", "b1c2", "b2e", "b2c2", "b3c2", "b0c1"]:
    n, cin, cout, h, w, k = SHAPES[name]
    for kind in ("fwd", "dgrad", "wgrad"):
        d = F._desc(n, cin, cout, h, w, k, k, 3)
        ok = F.conv_l16_wgrad_supported(d) if kind == "wgrad" else F.conv_l16_supported(d, int(kind == "dgrad"))
        if not ok:
            continue
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mhz = F.measure_l16_clock((n, cin, cout, h, w, k, k), kind, iters=40)
        dt = time.perf_counter() - t0
        print("%-5s %-5s %-34s shader clock %4.0f MHz" % (name, kind, F.l16_wgrad_plan_name(d) if kind == "wgrad" else F.l16_plan_name(d, int(kind == "dgrad")), mhz), flush=True)
