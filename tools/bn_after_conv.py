"""Development: does a BatchNorm pass run slower right behind an MFMA-bound convolution than behind another BatchNorm pass?
(cfg 2 stage timing: 4.45 TB/s in the step against 5.0 - 5.7 TB/s for the same passes in isolation.)"""
import os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F

dev = torch.device("cuda")
F.set_conv_arith("f16x3")
n, c, h, w = 128, 100, 64, 215
bn = nn.BatchNorm2d(c).to(dev); prelu = nn.PReLU(c).to(dev)
x = torch.randn(n, c, h, w, device=dev); dy = torch.randn_like(x)
st = F.bn_prepare(x, bn, True)
wt = torch.randn(c, c, 3, 3, device=dev) / 30
t16 = F.l16_pack(x, F.amax(x)); pp = F.conv_l16_pack(wt, n, h, w, False)
gb = x.numel() * 4 / 1e9


def bwd():
    return F.bn_act_backward(dy, x, st, bn, prelu.weight, with_amax=True, l16=True, want_f32=False)


def fwd():
    return F.bn_act_forward(x, st, prelu.weight, l16=True, want_f32=False)


def conv():
    return F.conv_l16(t16, wt, None, prepacked=pp)


def measure(pass_fn, before_fn, iters=20):
    for _ in range(3):
        if before_fn: before_fn()
        pass_fn()
    torch.cuda.synchronize()
    tot = 0.0
    evs = []
    for _ in range(iters):
        if before_fn: before_fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); pass_fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / iters


big = torch.empty(int(1.5e9) // 4, device=dev)             # 1.5 GB: flushes L2 and the 256 MB MALL
def flush():
    big.mul_(1.0)

for name, fn, passes in (("bwd l16", bwd, 5), ("fwd l16", fwd, 2)):
    a = measure(fn, None)
    b = measure(fn, conv)
    c_ = measure(fn, flush)
    print("%-8s back to back %.3f ms %.2f TB/s | behind an L16 conv %.3f ms %.2f TB/s | behind a 1.5 GB streaming pass %.3f ms %.2f TB/s" % (
        name, a, passes * gb / a, b, passes * gb / b, c_, passes * gb / c_))

# ---- the other direction: a 1x1 L16 convolution back to back, behind the BatchNorm apply pass that writes its operand, behind a flush
w1 = torch.randn(c, c, 1, 1, device=dev) / 10
pp1 = F.conv_l16_pack(w1, n, h, w, False)
fl = 2.0 * n * h * w * c * c
def conv1():
    return F.conv_l16(t16, w1, None, prepacked=pp1)
def producer():
    global t16
    _, t16 = F.bn_act_forward(x, st, prelu.weight, l16=True, want_f32=False)
a = measure(conv1, None); b = measure(conv1, producer); c_ = measure(conv1, flush); d_ = measure(conv1, conv)
mv = (x.numel() * 4 * 2) / 1e9
print("1x1 L16 conv 100->100: back to back %.3f ms %.2f TB/s | behind its producer (fwd_l16) %.3f ms %.2f TB/s | behind a flush %.3f ms | behind a 3x3 conv %.3f ms" % (
    a, mv / a, b, mv / b, c_, d_))
