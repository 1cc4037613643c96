"""Development: does the 256 MiB Infinity Cache serve a re-read?  fsc_amax (a pure read pass) over buffers of growing size,
repeated back to back, and A-then-B-then-A patterns."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F

dev = torch.device("cuda")
def t(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (32, 64, 96, 128, 160, 192, 224, 256, 384, 512, 1024, 2048):
    x = torch.randn(mb * 1024 * 1024 // 4, device=dev)
    ms = t(lambda: F.amax(x))
    # the same bytes as two halves read alternately with a big stranger in between (cache flushed)
    print("re-read of %5d MB: %.3f ms  %.2f TB/s" % (mb, ms, mb * 1.048576e-3 / ms), flush=True)
    del x
big = torch.randn(1024 * 1024 * 1024 // 4, device=dev)
for mb in (64, 128, 192):
    x = torch.randn(mb * 1024 * 1024 // 4, device=dev)
    y = torch.empty_like(x)
    # copy-like pass (read x write y) followed by a re-read of x: the pattern reduce -> apply
    def pat():
        F.amax(x); y.copy_(x)
    ms = t(pat)
    print("read + (read, write) of %4d MB: %.3f ms -> %.2f TB/s counting 3 x size" % (mb, ms, 3 * mb * 1.048576e-3 / ms), flush=True)
