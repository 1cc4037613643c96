"""Timing and parity of the conv + max-pool fused L16 kernel against conv_l16 followed by the max-pool pass (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402
from tools.conv_bench import SHAPES  # noqa: E402
from tools.l16_check import timeit  # noqa: E402

F.set_conv_arith("f16x3")
dev = torch.device("cuda")
for name in sys.argv[1:] or ["b1e", "b2e", "b3e"]:
    n, cin, cout, h, w, k = SHAPES[name]
    torch.manual_seed(1)
    x = torch.randn(n, cin, h, w, device=dev)
    wt = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, device=dev)
    t = F.l16_pack(x, F.amax(x))
    pp = F.conv_l16_pack(wt, n, h, w, False)
    c = F.conv_l16(t, wt, bias, prepacked=pp)
    p_ref, i_ref = F.maxpool_forward(c, 2)
    got = F.conv_l16_pool(t, wt, bias, prepacked=pp)
    if got is None:
        print(name, "no fused tiling")
        continue
    same = torch.equal(got[0], p_ref) and torch.equal(got[1], i_ref)
    ms_conv = timeit(lambda: F.conv_l16(t, wt, bias, prepacked=pp), 10)
    ms_pool = timeit(lambda: F.maxpool_forward(c, 2), 10)
    ms_fused = timeit(lambda: F.conv_l16_pool(t, wt, bias, prepacked=pp), 10)
    print("%-4s identical %s | conv %.3f ms + pool %.3f ms = %.3f | fused %.3f ms" % (name, same, ms_conv, ms_pool, ms_conv + ms_pool, ms_fused),
          flush=True)
