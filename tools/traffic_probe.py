"""HBM-traffic probe for the roofline `traffic` field (development tool).

Runs (1) fsc_axpy on 256 Mi floats -- a kernel whose HBM bytes are known exactly (2 reads + 1 write
of 1 GiB) -- and (2) the dominant conv kernel at its benchmark shape, so that FETCH_SIZE / WRITE_SIZE
collected with `rocprofv3 --pmc` (separate passes) can be calibrated on the known kernel as
/opt/skills/guides/MI355X_MICROARCH.md prescribes, then applied to the conv kernel.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd._lib import call, ptr, stream_ptr  # noqa: E402

dev = torch.device("cuda")
n = 256 << 20
x = torch.randn(n, device=dev)
y = torch.randn(n, device=dev)
for _ in range(3):
    call("fsc_axpy", ptr(x), 0.5, ptr(y), n, stream_ptr())
torch.cuda.synchronize()
del x, y
shape = (128, 100, 150, 64, 215, 3)          # block-1 entry conv = conv_fwd_kernel<3,3,5,2>
nb, cin, cout, h, w, k = shape
xi = torch.randn(nb, cin, h, w, device=dev)
wt = torch.randn(cout, cin, k, k, device=dev) / 30
for _ in range(3):
    F.conv_forward(xi, wt, None)
torch.cuda.synchronize()
print("done")
