"""ORACLE (test infrastructure, not product code).

numpy / pure-Python restatement of the host-side integer and index work on the path.
Bit-exact requirement.  Pinned against
goldens generated from the imported reference (tests/golden/make_golden.py).

Reference locations restated (relative to /root/reference):
  lwlrap                ops/utils.py:17-26 (sklearn LRAP, ties ranked 'max'; reading aid
                        relabel_noisy_data.py:91-114)
  mix_audio_and_labels  ops/audio.py:32-52   (note `=+` at :50 is an assignment)
  pad_collate           ops/padding.py:8-32
  bucket_batches        ops/padding.py:36-81
"""
import random

import numpy as np


def lwlrap(truth, scores):
    """Label-weighted label-ranking average precision; rows without positives skipped."""
    truth = np.asarray(truth) > 0
    scores = np.asarray(scores, dtype=np.float64)
    total_w = 0.0
    acc = 0.0
    for y, s in zip(truth, scores):
        npos = int(y.sum())
        if npos == 0:
            continue
        if npos == y.size:
            prec = 1.0
        else:
            pos = np.flatnonzero(y)
            # 'max' tie ranking: rank of label j = #labels with score >= s[j]
            rank_all = np.array([(s >= s[j]).sum() for j in pos], dtype=np.float64)
            rank_pos = np.array([(s[pos] >= s[j]).sum() for j in pos], dtype=np.float64)
            prec = float((rank_pos / rank_all).mean())
        acc += prec * npos
        total_w += npos
    return acc / total_w


def mix_audio_and_labels(first_audio, second_audio, first_labels, second_labels):
    """MixUp with OR-labels.  Consumes np.random.uniform once, then random.randint once
    when the lengths differ -- same draw order as the reference."""
    labels = np.clip(first_labels + second_labels, 0, 1)
    a = np.random.uniform(0.4, 0.6)
    if first_audio.size == second_audio.size:
        return (first_audio + second_audio) / 2, labels
    if first_audio.size > second_audio.size:
        long_, short_ = first_audio, second_audio
    else:
        long_, short_ = second_audio, first_audio
    start = random.randint(0, long_.size - 1 - short_.size)
    long_ *= a                                   # in place, like the reference
    long_[start:start + short_.size] = short_ * (1 - a)   # overlap is REPLACED
    return long_, labels


def pad_collate(batch, padding_values):
    """list of dict -> dict of stacked numpy arrays, named fields padded on axis 0."""
    out = {}
    for name in batch[0]:
        items = [np.asarray(s[name]) for s in batch]
        if name in padding_values:
            longest = max(len(x) for x in items)
            padded = []
            for x in items:
                extra = longest - len(x)
                if extra:
                    width = [(0, extra)] + [(0, 0)] * (x.ndim - 1)
                    if padding_values[name] == "edge":
                        x = np.pad(x, width, mode="edge")
                    else:
                        x = np.pad(x, width, mode="constant",
                                   constant_values=padding_values[name])
                padded.append(x)
            items = padded
        out[name] = np.stack(items)
    return out


def bucket_batches(lengths, max_batch_elems, buckets):
    """Length-bucketed batches; consumes python `random` exactly like the reference:
    one shuffle per non-edge bin (even empty ones), then one shuffle of the batch list."""
    lengths = np.asarray(lengths)
    bins = np.digitize(lengths, buckets)
    batches = []
    for b in range(1, len(buckets)):
        ids = [int(i) for i in np.nonzero(bins == b)[0]]
        random.shuffle(ids)
        cur, filled = [], 0
        for i in ids:
            if filled < max_batch_elems:
                cur.append(i)
                filled += int(lengths[i])
            else:
                batches.append(cur)
                cur, filled = [i], int(lengths[i])
        if cur:
            batches.append(cur)
    random.shuffle(batches)
    return batches
