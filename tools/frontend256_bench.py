"""Front-end timing at cfg 3: stft_256_128 log-magnitude of 128 x 441 000 samples (development tool)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from freesound_classification_amd import functional as F  # noqa: E402


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
a = F.frontend_stft(wav, 256, 128, True)
ms = t(lambda: F.frontend_stft(wav, 256, 128, True))
ref = torch.stft(wav[:4], 256, hop_length=128, window=torch.hann_window(256, device="cuda"), return_complex=True).abs()
print("stft_256_128: %.3f ms  (%.2f TB/s of the 453 MB algorithmic bytes)  max |log mag - torch| = %.2e" % (
    ms, 0.4534 / ms, (a[:4] - torch.log(ref + 1e-4)).abs().max().item()))
