"""Counterpart of the reference's ops/transforms.py: dict -> dict transforms with the protocol
``t(dataset=ds, **sample) -> dict`` used by SoundDataset and the training scripts.  Host code
(DataLoader workers); nothing here touches the GPU.

Provided: MapLabels, MixUp, LoadAudio, AudioFeatures, ShuffleAudio, SampleLongAudio, DropFields,
Compose (+ switch_off_augmentations), Identity, Augmentation.  `SyntheticAudio` replaces
`LoadAudio` for wav-free synthetic runs.  The sox-based AudioAugmentation and the transforms
the 2-d/1-d training pipelines never use are out of scope.
"""
import numpy as np

from .audio import mix_audio_and_labels, read_audio, shuffle_audio

SAMPLE_RATE = 44100


class Augmentation:
    """Base class: `Compose.switch_off_augmentations` zeroes `.p` of every instance."""
    pass


class MapLabels:
    """raw_labels (list of class names) -> multi-hot float32 `labels` (transforms.py:25-41)."""

    def __init__(self, class_map, drop_raw=True):
        self.class_map = class_map

    def __call__(self, dataset, **inputs):
        out = dict(inputs)
        hot = np.zeros(len(self.class_map), dtype=np.float32)
        for name in out.pop("raw_labels"):
            hot[self.class_map[name]] = 1.0
        out["labels"] = hot
        return out


class MixUp(Augmentation):
    """With probability p mix the clip with `dataset.random_clean_sample()` (transforms.py:44-65)."""

    def __init__(self, p):
        self.p = p

    def __call__(self, dataset, **inputs):
        out = dict(inputs)
        if np.random.uniform() < self.p:
            partner = dataset.random_clean_sample()
            out["audio"], out["labels"] = mix_audio_and_labels(
                inputs["audio"], partner["audio"], inputs["labels"], partner["labels"])
        return out


class LoadAudio:
    def __call__(self, dataset, **inputs):
        out = dict(inputs)
        out["audio"], out["sr"] = read_audio(inputs["filename"])
        return out


class SyntheticAudio:
    """Drop-in for LoadAudio on synthetic runs: `filename` is "synthetic:{seed}:{n_samples}[:{sr}]"
    and the clip is 0.1 * N(0, 1) float32 drawn from that seed."""

    def __call__(self, dataset, **inputs):
        out = dict(inputs)
        parts = str(inputs["filename"]).split(":")
        seed, n = int(parts[1]), int(parts[2])
        sr = int(parts[3]) if len(parts) > 3 else SAMPLE_RATE
        rng = np.random.RandomState(seed)
        out["audio"] = (0.1 * rng.standard_normal(n)).astype(np.float32)
        out["sr"] = sr
        return out


class AudioFeatures:
    """Parses the feature descriptor and emits `signal = audio[:, None]`; the spectrogram is
    computed on the device inside the model.  (The reference additionally runs a CPU STFT for
    "mel" and throws it away, transforms.py:220-226; that wasted work is not reproduced.)"""

    eps = 1e-4

    def __init__(self, descriptor, verbose=True):
        name, *args = descriptor.split("_")
        self.feature_type = name
        self.padding_value = 0.0
        if name == "stft":
            self.n_fft, self.hop_size = int(args[0]), int(args[1])
            self.n_features = self.n_fft // 2 + 1
        elif name == "mel":
            self.n_fft, self.hop_size, self.n_mel = int(args[0]), int(args[1]), int(args[2])
            self.n_features = self.n_mel
        elif name == "raw":
            self.n_features = 1
        else:
            raise ValueError("unknown feature descriptor %r" % descriptor)
        if verbose:
            print("\nUsing %s features: %s" % (name, descriptor))

    def __call__(self, dataset, **inputs):
        out = dict(inputs)
        out["signal"] = np.expand_dims(inputs["audio"], -1)
        return out


class ShuffleAudio(Augmentation):
    def __init__(self, chunk_length=0.5, p=0.5):
        self.chunk_length = chunk_length
        self.p = p

    def __call__(self, dataset, **inputs):
        out = dict(inputs)
        if np.random.uniform() < self.p:
            out["audio"] = shuffle_audio(out["audio"], self.chunk_length, sr=out["sr"])
        return out


class SampleLongAudio:
    """Random crop to max_length seconds when the clip is longer (transforms.py:292-309)."""

    def __init__(self, max_length):
        self.max_length = max_length

    def __call__(self, dataset, **inputs):
        out = dict(inputs)
        audio, sr = inputs["audio"], inputs["sr"]
        if audio.size / sr > self.max_length:
            keep = self.max_length * sr
            start = np.random.randint(0, audio.size - keep)
            out["audio"] = audio[start:start + keep]
        return out


class DropFields:
    def __init__(self, fields):
        self.to_drop = fields

    def __call__(self, dataset, **inputs):
        return {k: v for k, v in inputs.items() if k not in self.to_drop}


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def switch_off_augmentations(self):
        for t in self.transforms:
            if isinstance(t, Augmentation):
                t.p = 0.0

    def __call__(self, dataset=None, **inputs):
        for t in self.transforms:
            inputs = t(dataset=dataset, **inputs)
        return inputs


class Identity:
    def __call__(self, dataset=None, **inputs):
        return inputs
