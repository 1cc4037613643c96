"""development: record the discrete decisions of a training forward -- max-pool window indices, global-max indices -- so that two
configurations of the same batch (one process x 8 clips, two replicas x 4 clips) can be compared decision by decision
(tools/dbg_dp_syncbn.py; tests/dp_worker.py installs it when FSC_DP_DUMP names a file prefix)."""
import numpy as np

REC = []


def install():
    from freesound_classification_amd import functional as F

    def wrap(name, pick):
        orig = getattr(F, name)

        def f(*a, **k):
            out = orig(*a, **k)
            got = pick(out)
            if got is not None:
                REC.append((name, got.detach().cpu().numpy().copy()))
            return out
        setattr(F, name, f)

    wrap("maxpool_forward", lambda o: o[1])
    wrap("conv_pool_forward", lambda o: None if o is None else o[1])
    wrap("conv_l16_pool", lambda o: None if o is None else o[1])
    wrap("global_maxpool_forward", lambda o: o[1])
    wrap("bn_act_forward_rec", lambda o: o[2])
    # backward stages: the input gradient each BatchNorm backward returns (and its parameter gradients), every conv input gradient
    wrap("bn_act_backward", lambda o: o[0])
    wrap("bn_act_backward_unpool", lambda o: o[0])
    wrap("conv_dgrad", lambda o: o)
    wrap("bn_prepare", lambda o: None if getattr(o, "mean", None) is None else __import__("torch").stack([o.mean, o.invstd]))


def save(path):
    np.savez(path, **{"%02d_%s" % (i, n): a for i, (n, a) in enumerate(REC)})
