"""Counterpart of the reference's networks/losses.py: same names and signatures, HIP kernels.

Only the two losses the training scripts can reach are provided (`lsep_loss`,
`binary_cross_entropy`); the reference's `focal_loss` / `lsep_loss_stable` are never called.
"""
from .. import functional as F


def lsep_loss(input, target, average=True):
    """Log-sum-exp pairwise ranking loss (reference networks/losses.py:47-58).

    input (N, C) logits, target (N, C); pair (i, j) counts when target[j] < target[i].
    Returns the mean over samples, or the (N,) per-sample losses when average=False.
    The exponentials are not stabilised, matching the reference."""
    per_sample = F.LsepFn.apply(input, target)
    return F.mean(per_sample) if average else per_sample


def binary_cross_entropy(input, target, raw=True):
    """Mean BCE over all elements (reference networks/losses.py:19-22); `raw` logits only."""
    if not raw:
        raise NotImplementedError("binary_cross_entropy on probabilities is not on the accelerated path")
    return F.BceFn.apply(input, target)
