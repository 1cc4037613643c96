"""Counterpart of the reference's ops/folds.py: `train_validation_data` (sklearn KFold, ops/folds.py:6-11) and
`train_validation_data_stratified` (multilabel iterative stratification, ops/folds.py:14-24).

The reference delegates the latter to `iterstrat.ml_stratifiers.MultilabelStratifiedKFold`
(iterative-stratification==0.1.6, requirements.txt:31), which is absent from this image and from /root/reference;
`multilabel_stratified_folds` restates its published algorithm (Sechidis et al. 2011 as implemented by that package:
labels with the fewest remaining examples first, each example to the fold that still wants the most examples of that
label, ties by the fold wanting most examples overall, then at random).  No fixture of the reference pins fold
membership, so fold identity against iterstrat is *parity-unpinned*; tests pin the properties (a partition, per-label
counts balanced to within the algorithm's guarantee, determinism under the seed).
Host code, integer work."""
import numpy as np


def train_validation_data(ids, labels, n_folds, seed):
    from sklearn.model_selection import KFold
    for train, valid in KFold(n_folds, shuffle=True, random_state=seed).split(ids, labels):
        yield train, valid


def _iterative_stratification(labels, r, rng):
    n = labels.shape[0]
    test_folds = np.zeros(n, dtype=int)
    want = r * n                                           # examples each fold still wants
    want_label = np.outer(r, labels.sum(axis=0)).astype(np.float64)   # ... per label
    todo = np.ones(n, dtype=bool)
    while todo.any():
        remaining = labels[todo].sum(axis=0)
        if remaining.sum() == 0:                           # only label-free rows left: fill the emptiest folds
            for i in np.where(todo)[0]:
                best = np.where(want == want.max())[0]
                f = best[rng.choice(best.shape[0])] if best.shape[0] > 1 else best[0]
                test_folds[i] = f
                want[f] -= 1
            break
        rare = np.where(remaining == remaining[np.nonzero(remaining)].min())[0]
        lab = rare[rng.choice(rare.shape[0])] if rare.shape[0] > 1 else rare[0]
        for i in np.where(np.logical_and(labels[:, lab], todo))[0]:
            col = want_label[:, lab]
            best = np.where(col == col.max())[0]
            if best.shape[0] > 1:
                sub = np.where(want[best] == want[best].max())[0]
                best = best[sub]
                if sub.shape[0] > 1:
                    best = best[[rng.choice(sub.shape[0])]]
            f = int(best[0])
            test_folds[i] = f
            todo[i] = False
            want_label[f, labels[i]] -= 1
            want[f] -= 1
    return test_folds


def multilabel_stratified_folds(binary_labels, n_folds, seed, shuffle=True):
    """Fold index per row for a (n, classes) multi-hot matrix."""
    y = np.asarray(binary_labels) > 0
    rng = np.random.RandomState(seed)
    order = np.arange(y.shape[0])
    if shuffle:
        rng.shuffle(order)
        y = y[order]
    r = np.asarray([1.0 / n_folds] * n_folds)
    folds = _iterative_stratification(y, r, rng)
    return folds[np.argsort(order)]


def train_validation_data_stratified(ids, labels, classmap, n_folds, seed):
    """`labels`: a pandas Series (or sequence) of comma-joined class names, as in the competition CSVs."""
    values = labels.values if hasattr(labels, "values") else labels
    binary = np.zeros((len(values), len(classmap)), dtype=np.float32)
    for k, item in enumerate(values):
        for name in (item.split(",") if isinstance(item, str) else item):
            binary[k, classmap[name]] = 1
    folds = multilabel_stratified_folds(binary, n_folds, seed)
    index = np.arange(len(values))
    for f in range(n_folds):
        yield index[folds != f], index[folds == f]
