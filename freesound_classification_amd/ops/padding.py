"""Counterpart of the reference's ops/padding.py: pad-to-longest collate and the
length-bucketing batch sampler.  Index work must be bit-exact with the reference (padding is
not masked downstream, so batch composition changes the numbers): same `random` draw order.
"""
import random

import numpy as np
from torch.utils.data.dataloader import default_collate


def make_collate_fn(padding_values):
    """padding_values: {field: constant | "edge"}; each named field is padded along axis 0 to
    the longest item of the batch (in place, like reference ops/padding.py:8-32), then the
    batch goes through torch's default_collate."""

    def _collate_fn(batch):
        for field, fill in padding_values.items():
            longest = max(len(item[field]) for item in batch)
            for item in batch:
                arr = item[field]
                missing = longest - len(arr)
                if missing == 0:
                    continue
                widths = [(0, missing)] + [(0, 0)] * (arr.ndim - 1)
                if fill == "edge":
                    item[field] = np.pad(arr, widths, mode="edge")
                else:
                    item[field] = np.pad(arr, widths, mode="constant", constant_values=fill)
        return default_collate(batch)

    return _collate_fn


class BucketingSampler:
    """Batches of similar-length items (reference ops/padding.py:36-81).

    Items are binned with np.digitize(lengths, buckets); only bins 1..len(buckets)-1 are kept
    (items shorter than buckets[0] or not shorter than buckets[-1] are dropped, as in the
    reference).  Inside a bin the ids are shuffled, then batches are filled greedily: the
    size check happens BEFORE an item is added, so a batch may overshoot `max_batch_elems` by
    one item.  Finally the batch list is shuffled.  Uses the global `random` state."""

    def __init__(self, dataset, max_batch_elems, buckets):
        self.dataset = dataset
        self.max_batch_elems = max_batch_elems
        self.buckets = buckets
        self._create_batches()

    def _create_batches(self):
        lengths = self.dataset.lengths
        which = np.digitize(lengths, self.buckets)
        self.n_bins = len(self.buckets)
        batches = []
        for b in range(1, self.n_bins):
            members = list(np.nonzero(which == b)[0])
            random.shuffle(members)
            batch, filled = [], 0
            for i in members:
                if filled >= self.max_batch_elems:
                    batches.append(batch)
                    batch, filled = [], 0
                batch.append(i)
                filled += lengths[i]
            if batch:
                batches.append(batch)
        random.shuffle(batches)
        self.batches = batches
        self.n_batches = len(batches)

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return self.n_batches
