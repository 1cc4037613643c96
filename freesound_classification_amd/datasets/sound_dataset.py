"""Counterpart of the reference's datasets/sound_dataset.py (same constructor and protocol)."""
import random

import numpy as np
import torch.utils.data as data


class SoundDataset(data.Dataset):
    """Items are dicts {filename, is_noisy[, raw_labels]} pushed through `transform`, which is
    called as transform(dataset=self, **sample) (reference datasets/sound_dataset.py:14-59)."""

    def __init__(self, audio_files, labels=None, transform=None, is_noisy=None, clean_transform=None):
        self.audio_files = audio_files
        self.labels = labels
        self.transform = transform
        self.clean_transform = clean_transform
        self.is_noisy = is_noisy or np.zeros(len(self.audio_files))

    def _raw(self, index):
        sample = dict(filename=self.audio_files[index], is_noisy=self.is_noisy[index])
        if self.labels is not None:
            sample["raw_labels"] = self.labels[index]
        return sample

    def __getitem__(self, index):
        sample = self._raw(index)
        if self.transform is not None:
            sample = self.transform(dataset=self, **sample)
        return sample

    def random_clean_sample(self):
        """MixUp partner: uniform index from python's `random`, through `clean_transform`."""
        sample = self._raw(random.randint(0, len(self) - 1))
        if self.clean_transform is not None:
            sample = self.clean_transform(dataset=self, **sample)
        return sample

    def __len__(self):
        return len(self.audio_files)
