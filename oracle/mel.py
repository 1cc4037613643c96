"""ORACLE (test infrastructure, not product code).

Restatement of the Slaney-style mel filterbank that the reference obtains from a
third-party dependency which is NOT vendored under /root/reference:

    librosa==0.6.3 (reference requirements.txt:35), ``librosa.filters.mel`` called at
    reference ops/utils.py:94-97 as mel(sr=44100, n_fft, n_mels, fmin=5, fmax=None).

Published algorithm (librosa 0.6.x docs, "filters.mel", htk=False, norm=1):
  * FFT bin centres: linspace(0, sr/2, 1 + n_fft//2)
  * Slaney mel scale: linear below 1 kHz with 200/3 Hz per mel, logarithmic above with
    step ln(6.4)/27 per mel
  * n_mels + 2 band edges equally spaced on that scale between fmin and fmax
  * triangular weights max(0, min(rising, falling)), then area normalisation
    2 / (f[i+2] - f[i]).

PARITY UNPINNED at this boundary: the reference holds no test or fixture for the
filterbank and librosa itself is absent here; the only published pin is the librosa
docstring value mel(22050, 2048)[0, 1] ~= 0.016, which tests/test_oracle_cpu.py checks.
The filterbank is a *data input* to the HIP mel kernel, so kernel parity is pinned
independently of this file (tests compare on the same matrix).
"""
import numpy as np

_F_SP = 200.0 / 3.0
_BREAK_HZ = 1000.0
_BREAK_MEL = _BREAK_HZ / _F_SP
_LOGSTEP = np.log(6.4) / 27.0


def hz_to_mel(hz):
    hz = np.asarray(hz, dtype=np.float64)
    lin = hz / _F_SP
    with np.errstate(divide="ignore", invalid="ignore"):
        log = _BREAK_MEL + np.log(np.maximum(hz, 1e-300) / _BREAK_HZ) / _LOGSTEP
    return np.where(hz >= _BREAK_HZ, log, lin)


def mel_to_hz(mel):
    mel = np.asarray(mel, dtype=np.float64)
    lin = mel * _F_SP
    log = _BREAK_HZ * np.exp(_LOGSTEP * (mel - _BREAK_MEL))
    return np.where(mel >= _BREAK_MEL, log, lin)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin=0.0, fmax=None):
    """(n_mels, 1 + n_fft//2) float64 filterbank, librosa-0.6.3 semantics."""
    if fmax is None:
        fmax = sr / 2.0
    n_bins = 1 + n_fft // 2
    bin_hz = np.linspace(0.0, sr / 2.0, n_bins)
    edges = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    widths = edges[1:] - edges[:-1]
    # distance of every bin from every edge: d[e, b] = edges[e] - bin_hz[b]
    d = edges[:, None] - bin_hz[None, :]
    rising = -d[:-2] / widths[:-1, None]
    falling = d[2:] / widths[1:, None]
    fb = np.clip(np.minimum(rising, falling), 0.0, None)
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb


def make_mel_filterbanks(descriptor, sr=44100):
    """Oracle of reference ops/utils.py:85-99 ("mel_{n_fft}_{hop}_{n_mel}", fmin=5)."""
    _, n_fft, _hop, n_mel = descriptor.split("_")
    return slaney_mel_filterbank(sr, int(n_fft), int(n_mel), fmin=5.0).astype(np.float32)
