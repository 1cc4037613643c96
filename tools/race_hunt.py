"""Development: the same training step (fixed inputs, no optimizer step, BatchNorm running statistics restored) many times; every
repetition must reproduce the first one up to the noise of atomic summation.  Reports repetitions whose logits or gradients differ.
    python tools/race_hunt.py [cfg3|cfg2] [repetitions]"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd.networks.classifiers import (  # noqa: E402
    HierarchicalCNNClassificationModel, TwoDimensionalCNNClassificationModel)

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
w = bench.WORKLOADS[name]
if w.get("arith"):
    F.set_conv_arith(w["arith"])
dev = torch.device("cuda:0")
torch.manual_seed(42)
cls = HierarchicalCNNClassificationModel if w.get("dims") == 1 else TwoDimensionalCNNClassificationModel
model = cls(bench.make_experiment(w), device=str(dev))
model.train()
model.global_step = 0
model.make_optimizer(max_steps=100)
signal, labels = bench.synthetic_batch(w, w["batch"], dev, 1234)
for _ in range(3):
    model.training_step(signal, labels)          # a few real steps: away from the initialisation
state = copy.deepcopy(model.state_dict())
dstate = copy.deepcopy(model._dropout_state) if hasattr(model, "_dropout_state") else None


def once():
    model.load_state_dict(state)
    if dstate is not None:
        model._dropout_state = copy.deepcopy(dstate)
    for p in model.parameters():
        p.grad = None
    logits, per, loss = model.training_step(signal, labels, step_optimizer=False)
    torch.cuda.synchronize()
    return logits.detach().clone(), {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}, float(loss)


l0, g0, loss0 = once()
print("reference loss", loss0, "finite", bool(torch.isfinite(l0).all()))
gmax = max(v.abs().max().item() for v in g0.values())
bad = 0
for r in range(reps):
    l, g, loss = once()
    dl = float((l - l0).abs().max()) if torch.isfinite(l).all() else float("nan")
    worst, wk = 0.0, None
    for k in g0:
        if g0[k].abs().max().item() < 1e-4 * gmax:       # (a convolution bias in front of a BatchNorm: its gradient is rounding noise)
            continue
        d = (g[k] - g0[k]).abs().max().item() / max(1e-12, g0[k].abs().max().item())
        if not (d <= worst):
            worst, wk = d, k
    if not (dl <= 1e-3 * max(1.0, float(l0.abs().max()))) or not (worst <= 5e-2):
        bad += 1
        if bad <= 12:
            first = [k for k in g0 if not torch.isfinite(g[k]).all()]
            print("rep %d: loss %r, max |dlogits| %r, worst gradient %s rel %r, non-finite gradients: %d %s"
                  % (r, loss, dl, wk, worst, len(first), first[:3]), flush=True)
print("%d of %d repetitions deviate" % (bad, reps))
