"""Host-side waveform helpers with the reference's names (ops/audio.py).  These run inside
DataLoader worker processes and never touch the GPU; the device-side batched MixUp lives in
`freesound_classification_amd.functional.mixup_batch`.
"""
import random

import numpy as np


def read_audio(file):
    """Decode a file to (float32 mono waveform, sample rate).  The reference uses librosa
    (absent here); 16-bit / float WAV files are read with scipy."""
    from scipy.io import wavfile
    sr, data = wavfile.read(file)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    data = data.astype(np.float32)
    if data.ndim > 1:
        data = data.mean(axis=1)
    return data, sr


def mix_audio_and_labels(first_audio, second_audio, first_labels, second_labels):
    """MixUp with OR-ed labels (reference ops/audio.py:32-52).

    Equal lengths: plain average.  Otherwise the longer clip is scaled by a ~ U(0.4, 0.6) in
    place and the window [start, start + len(shorter)) is REPLACED by shorter * (1 - a): the
    reference's `=+` is an assignment, and that behaviour is kept.  Draw order: np.random
    first, then `random.randint` (only when lengths differ)."""
    labels = np.clip(first_labels + second_labels, 0, 1)
    a = np.random.uniform(0.4, 0.6)
    n1, n2 = first_audio.size, second_audio.size
    if n1 == n2:
        return (first_audio + second_audio) / 2, labels
    longer, shorter = (first_audio, second_audio) if n1 > n2 else (second_audio, first_audio)
    start = random.randint(0, longer.size - 1 - shorter.size)
    longer *= a
    longer[start:start + shorter.size] = shorter * (1 - a)
    return longer, labels


def _even_slices(n, n_packs):
    """sklearn.utils.gen_even_slices(n, n_packs): the first n % n_packs slices get one extra."""
    start = 0
    for pack in range(n_packs):
        size = n // n_packs + (1 if pack < n % n_packs else 0)
        if size > 0:
            yield slice(start, start + size)
            start += size


def shuffle_audio(audio, chunk_length=0.5, sr=None):
    """Permute `chunk_length`-second chunks (reference ops/audio.py:55-67)."""
    n_chunks = int((audio.size / sr) / chunk_length)
    if n_chunks in (0, 1):
        return audio
    pieces = list(_even_slices(audio.size, n_chunks))
    random.shuffle(pieces)
    return np.concatenate([audio[s] for s in pieces])
