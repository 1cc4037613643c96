"""Development: the bench's cfg-3 loop (MixUp, optimizer steps) from a fresh model, many times, checking every step for non-finite
values: loss, logits, front-end output, gradients, parameters.  Reports the first bad step of each failing trial.
    python tools/nan_steps.py [trials] [steps] [one-cycle peak rate, default: the bench's]
Also reports, per trial, the largest |logit| and the largest score gap max_j z_j - min_i z_i any step reached: LSEP (reference
networks/losses.py:47-58, reproduced un-stabilised) overflows fp32 when a gap exceeds ~88.  FSC_BN_NO_TICKETS=1 takes the separate
BatchNorm finalisation launches (no cross-workgroup hand-off anywhere in the step): if the non-finite losses were a race in the
ticketed finalisation they would disappear with it."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd.networks.classifiers import HierarchicalCNNClassificationModel  # noqa: E402
from freesound_classification_amd.ops.training import make_step  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
w = dict(bench.WORKLOADS["cfg3"])
if len(sys.argv) > 3:
    w["scheduler"] = "1cycle_0.0001_%s" % sys.argv[3]
print("scheduler %s, BatchNorm tickets %s" % (w["scheduler"], "off (FSC_BN_NO_TICKETS)" if not F.BN_TICKETS else "on"), flush=True)
F.set_conv_arith(w["arith"])
dev = torch.device("cuda:0")
batch = w["batch"]
failed = 0
for trial in range(trials):
    torch.manual_seed(42)
    model = HierarchicalCNNClassificationModel(bench.make_experiment(w), device=str(dev))
    model.train()
    model.global_step = 0
    model.make_optimizer(max_steps=steps + 1)
    signal, labels = bench.synthetic_batch(w, batch, dev, 1234)
    rng = np.random.RandomState(7)
    peak_logit, peak_gap, peak_at, last_loss = 0.0, 0.0, -1, float("nan")
    for step in range(steps):
        model.global_step += 1
        make_step(model.scheduler, step=model.global_step)
        perm = torch.from_numpy(rng.permutation(batch)).to(dev)
        take = torch.from_numpy(rng.uniform(size=batch) < w["mixup"]).to(dev)
        partner = torch.where(take, perm, torch.arange(batch, device=dev))
        t = w["samples"]
        mixed, y = F.mixup_batch(signal.squeeze(-1), signal.squeeze(-1)[partner].contiguous(), [t] * batch, [t] * batch,
                                 [0] * batch, rng.uniform(0.4, 0.6, size=batch), labels, labels[partner].contiguous())
        x = mixed.unsqueeze(-1)
        logits, per, loss = model.training_step(x, y)
        torch.cuda.synchronize()
        lg = logits.detach()
        if torch.isfinite(lg).all():
            gap = float((lg.max(dim=1).values - lg.min(dim=1).values).max())
            if gap > peak_gap:
                peak_gap, peak_at = gap, step
            peak_logit = max(peak_logit, float(lg.abs().max()))
            last_loss = float(loss.detach())
        if not torch.isfinite(loss.detach()).all():
            failed += 1
            bad_l = int((~torch.isfinite(logits.detach())).sum())
            bad_x = int((~torch.isfinite(mixed)).sum())
            with torch.no_grad():
                fe = model.features(x)
            bad_fe = int((~torch.isfinite(fe)).sum())
            bad_p = [k for k, v in model.named_parameters() if not torch.isfinite(v).all()]
            bad_g = [k for k, v in model.named_parameters() if v.grad is not None and not torch.isfinite(v.grad).all()]
            bad_b = [k for k, v in model.named_buffers() if v.dtype.is_floating_point and not torch.isfinite(v).all()]
            print("trial %d step %d: loss %r; non-finite logits %d, mixed input %d, front-end (recomputed) %d; parameters %d %s; "
                  "gradients %d %s; buffers %d %s" % (trial, step, float(loss), bad_l, bad_x, bad_fe, len(bad_p), bad_p[:4],
                                                     len(bad_g), bad_g[:4], len(bad_b), bad_b[:4]), flush=True)
            break
    print("trial %2d: largest |logit| %8.2f, largest score gap %8.2f (step %d), last finite loss %.4f" % (trial, peak_logit, peak_gap, peak_at, last_loss),
          flush=True)
    model.close()
print("%d of %d trials hit a non-finite loss" % (failed, trials))
