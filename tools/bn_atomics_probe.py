"""Development: what the per-channel atomics of the BatchNorm backward apply pass cost on the 1-d model's planes (cfg 3): the same
backward with / without the channel sum of dx (conv bias gradient) and with / without max |dx|.   python tools/bn_atomics_probe.py"""
import sys
import torch
import torch.nn as nn
sys.path.insert(0, ".")
from freesound_classification_amd import functional as F

dev = torch.device("cuda:0")


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (n, c, h, w) in [(128, 64, 1, 1723), (128, 80, 1, 861), (128, 100, 1, 430), (128, 125, 1, 215), (128, 195, 1, 53), (128, 305, 1, 13)]:
    bn = nn.BatchNorm2d(c).to(dev)
    prelu = nn.PReLU(c).to(dev)
    x = torch.randn(n, c, h, w, device=dev)
    dy = torch.randn_like(x)
    st = F.bn_prepare(x, bn, True)
    for amax in (False, True):
        for csum in (False, True):
            us = timeit(lambda: F.bn_act_backward(dy, x, st, bn, prelu.weight, with_amax=amax, want_chan_sum=csum))
            print("%-20s amax=%d csum=%d  %7.1f us" % ((n, c, h, w), amax, csum, us), flush=True)
