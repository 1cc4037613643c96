"""Front-end timing: the wave-per-frame n_fft = 2048 kernel against the generic one (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd.ops.utils import make_mel_filterbanks  # noqa: E402


def t(fn, it=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


wav = 0.1 * torch.randn(128, 441000, device="cuda")
bands = F.MelBands(make_mel_filterbanks("mel_2048_1024_128"), wav.device)
a = F.frontend_logmel(wav, 2048, 1024, bands, True)
ms = t(lambda: F.frontend_logmel(wav, 2048, 1024, bands, True))
print("wave-per-frame: %.3f ms  (%.2f TB/s of the 254 MB algorithmic bytes)" % (ms, 0.254 / ms))
os.environ["FSC_FRONTEND_GENERIC"] = "1"
b = F.frontend_logmel(wav, 2048, 1024, bands, True)
print("generic:        %.3f ms" % t(lambda: F.frontend_logmel(wav, 2048, 1024, bands, True)))
print("max |new - generic| = %.3e" % (a - b).abs().max().item())
