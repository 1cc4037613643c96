"""development: tests/test_dp_gpu.py's SyncBN comparison (two replicas on one GPU against the single-process global batch), per
tensor and per arithmetic (FSC_CONV_ARITH is inherited by the worker processes)."""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_dp_gpu as T
import dp_worker
from freesound_classification_amd import functional as F

print("arith", F.get_conv_arith())
dump = os.path.join(ROOT, "gpurun_out", "dp_dump")
os.environ["FSC_DP_DUMP"] = dump
with tempfile.TemporaryDirectory() as d:
    r0, r1 = T._run_two_replicas(d, sync_bn=True)
del os.environ["FSC_DP_DUMP"]
x, y = dp_worker.global_batch()
import dp_dump
dp_dump.install()
logits, per, grads, state = T._single_process(x, y, seed=5)
single = list(dp_dump.REC)
dp_dump.save(dump + ".single.npz")
z0, z1 = np.load(dump + ".rank0.npz"), np.load(dump + ".rank1.npz")
keys = sorted(z0.files)
print("decisions recorded: single %d, replicas %d" % (len(single), len(keys)))
for (name, a), k in zip(single, keys):
    if "bn_prepare" in k:
        print("  %-28s statistics: replica 0 vs single %.2e (relative), replica 1 vs single %.2e" % (
            k, float(np.abs(z0[k] / a - 1).max()), float(np.abs(z1[k] / a - 1).max())))
        continue
    both = np.concatenate([z0[k], z1[k]])
    if both.dtype.kind == "f":
        both = both / 2.0                                   # (a replica's loss is the mean over ITS four clips)
        d = np.abs(both - a).reshape(a.shape[0], -1).max(1) / max(1e-30, float(np.abs(a).max()))
        print("  %-28s %s: max |replicas / 2 - single| / max|single| per clip: %s" % (k, tuple(a.shape), ", ".join("%.1e" % v for v in d)))
        continue
    if both.shape != a.shape:
        print("  %-28s shapes differ %s %s" % (k, a.shape, both.shape)); continue
    ne = both != a
    per_clip = ne.reshape(ne.shape[0], -1).sum(1)
    print("  %-28s %s: %d of %d decisions differ; per clip %s" % (k, tuple(a.shape), int(ne.sum()), ne.size, per_clip.tolist()))
print("logits", float(np.abs(np.concatenate([r0["logits"], r1["logits"]]) - logits).max()))
rows = []
for k, g in grads.items():
    rows.append((float(np.abs(r0["grad." + k] - g).max()), float(np.abs(g).max()), k))
for d_, m_, k in sorted(rows, reverse=True)[:12]:
    print("%-40s diff %.3e  |g|max %.3e" % (k, d_, m_))
# the single process twice (run-to-run)
_, _, grads2, _ = T._single_process(x, y, seed=5)
print("single run-to-run worst:", max(float(np.abs(grads2[k] - grads[k]).max()) for k in grads))

# conditioning of the test point: the single process again on the input scaled by (1 + 2^-22) -- a constant factor on the waveform is
# a constant offset of the log-mel image, which the first BatchNorm removes: analytically the SAME gradients
_, _, grads3, _ = T._single_process(x * (1.0 + 2.0 ** -22), y, seed=5)
rows = sorted(((float(np.abs(grads3[k] - grads[k]).max()), k) for k in grads), reverse=True)[:6]
print("single process, input scaled by 1 + 2^-22:", ", ".join("%s %.2e" % (k, v) for v, k in rows))
