"""Development: the bench's cfg-3 loop (MixUp, one-cycle schedule) eager and replayed, loss per step."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from freesound_classification_amd import functional as F
from freesound_classification_amd.networks.classifiers import HierarchicalCNNClassificationModel
from freesound_classification_amd.ops.training import CapturedTrainingStep, make_step

mode = sys.argv[1]                    # eager | graph | graph_nomix | eager_nomix
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
w = bench.WORKLOADS["cfg3"]
F.set_conv_arith(w["arith"])
dev = torch.device("cuda:0")
batch = w["batch"]
torch.manual_seed(42)
_side = torch.cuda.Stream(device=dev); _side.wait_stream(torch.cuda.current_stream(dev)); torch.cuda.set_stream(_side)   # one non-default stream for everything
model = HierarchicalCNNClassificationModel(bench.make_experiment(w), device=str(dev))
model.train(); model.global_step = 0
model.make_optimizer(max_steps=int(sys.argv[3]) if len(sys.argv) > 3 else steps + 30)
signal, labels = bench.synthetic_batch(w, batch, dev, 1234)
rng = np.random.RandomState(7)
fn = [model.training_step]
losses, gaps = [], []
for step in range(steps):
    if step == 5 and mode.startswith("graph"):
        lg = per = loss = lgd = None               # no tensor of an earlier step's graph may be alive at the recording
        fn[0] = CapturedTrainingStep(model, signal, labels)
    model.global_step += 1; make_step(model.scheduler, step=model.global_step)
    x, y = signal, labels
    if not mode.endswith("nomix"):
        perm = torch.from_numpy(rng.permutation(batch)).to(dev)
        take = torch.from_numpy(rng.uniform(size=batch) < w["mixup"]).to(dev)
        partner = torch.where(take, perm, torch.arange(batch, device=dev))
        t = w["samples"]
        mixed, y = F.mixup_batch(signal.squeeze(-1), signal.squeeze(-1)[partner].contiguous(), [t] * batch, [t] * batch, [0] * batch,
                                 rng.uniform(0.4, 0.6, size=batch), labels, labels[partner].contiguous())
        x = mixed.unsqueeze(-1)
    lg, per, loss = fn[0](x, y)
    losses.append(float(loss.detach()))
    lgd = lg.detach()
    gaps.append(float((lgd.max(dim=1).values - lgd.min(dim=1).values).max()))
print(mode, "loss:", " ".join("%.3f" % v for v in losses))
print(mode, "gap: ", " ".join("%.0f" % v for v in gaps))
