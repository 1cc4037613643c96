"""Counterpart of the reference's ops/utils.py for the accelerated path: same public names
(`lwlrap`, `compute_torch_stft`, `make_mel_filterbanks`, `is_mel`, `is_stft`, `load_json`,
`get_class_names_from_classmap`), STFT on the HIP front-end kernel.
"""
import json

import numpy as np

from .. import functional as F


def lwlrap(truth, scores):
    """Label-weighted label-ranking average precision (reference ops/utils.py:17-26, which
    delegates to sklearn's LRAP with ties ranked 'max'); rows without positives are skipped.
    Host metric, fp64."""
    truth = np.asarray(truth) > 0
    scores = np.asarray(scores, dtype=np.float64)
    weights = truth.sum(axis=1)
    keep = weights > 0
    truth, scores, weights = truth[keep], scores[keep], weights[keep].astype(np.float64)
    n, c = truth.shape
    total = 0.0
    for row in range(n):
        pos = truth[row]
        k = int(weights[row])
        if k == c:
            total += float(k)
            continue
        s = scores[row]
        s_pos = s[pos]
        # rank among all labels / among positives, counting ties against the label ('max')
        rank_all = (s[None, :] >= s_pos[:, None]).sum(axis=1)
        rank_pos = (s_pos[None, :] >= s_pos[:, None]).sum(axis=1)
        total += float((rank_pos / rank_all).mean()) * k
    return total / float(weights.sum())


def load_json(file):
    with open(file, "r") as f:
        return json.load(f)


def get_class_names_from_classmap(classmap):
    by_index = sorted(classmap.items(), key=lambda kv: kv[1])
    return [name for name, _ in by_index]


def is_mel(descriptor):
    return descriptor.startswith("mel")


def is_stft(descriptor):
    return descriptor.startswith("stft")


def parse_features(descriptor):
    """"mel_{n_fft}_{hop}_{n_mel}" / "stft_{n_fft}_{hop}" / "raw" -> (kind, [ints])."""
    kind, *rest = descriptor.split("_")
    return kind, [int(v) for v in rest]


def _slaney_hz_to_mel(hz):
    hz = float(hz)
    if hz < 1000.0:
        return hz * 3.0 / 200.0
    return 15.0 + np.log(hz / 1000.0) * 27.0 / np.log(6.4)


def _slaney_mel_to_hz(mel):
    mel = np.asarray(mel, dtype=np.float64)
    hz = mel * 200.0 / 3.0
    upper = mel >= 15.0
    hz[upper] = 1000.0 * np.exp((mel[upper] - 15.0) * np.log(6.4) / 27.0)
    return hz


def make_mel_filterbanks(descriptor, sr=44100):
    """(n_mel, n_fft/2+1) float32 Slaney filterbank, fmin=5 Hz, fmax=sr/2, area-normalised
    triangles -- what the reference gets from librosa.filters.mel (ops/utils.py:85-99).
    `sr` defaults to 44100 for every clip, exactly like the reference's callers."""
    _, (n_fft, _hop, n_mel) = parse_features(descriptor)
    n_bins = n_fft // 2 + 1
    centres = np.linspace(0.0, sr / 2.0, n_bins)
    edges = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(5.0), _slaney_hz_to_mel(sr / 2.0), n_mel + 2))
    step = np.diff(edges)
    ramps = edges.reshape(-1, 1) - centres.reshape(1, -1)
    fb = np.zeros((n_mel, n_bins))
    for m in range(n_mel):
        up = -ramps[m] / step[m]
        down = ramps[m + 2] / step[m + 1]
        fb[m] = np.maximum(0.0, np.minimum(up, down)) * (2.0 / (edges[m + 2] - edges[m]))
    return fb.astype(np.float32)


def compute_torch_stft(audio, descriptor):
    """(N, T) device waveform -> (N, n_fft/2+1, 1 + T//hop) STFT magnitude
    (reference ops/utils.py:110-127: centred, reflect-padded, periodic Hann, one-sided)."""
    _, nums = parse_features(descriptor)
    n_fft, hop = nums[0], nums[1]
    return F.frontend_stft(audio, n_fft, hop, apply_log=False)
