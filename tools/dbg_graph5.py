"""Development: which host-side reads between replays a CapturedTrainingStep run needs (it must need none).
    python tools/dbg_graph5.py <reads: none|loss|all|eager_only|replay_only> [steps] [sync: 0|1]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from freesound_classification_amd import functional as F
from freesound_classification_amd.networks.classifiers import HierarchicalCNNClassificationModel
from freesound_classification_amd.ops.training import CapturedTrainingStep, make_step

reads = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 27
smode = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # 0 never, 1 every step, 2 eager steps only, 3 replays only
sync = smode != 0
w = bench.WORKLOADS["cfg3"]
F.set_conv_arith(w["arith"])
dev = torch.device("cuda:0")
batch = w["batch"]
torch.manual_seed(42)
model = HierarchicalCNNClassificationModel(bench.make_experiment(w), device=str(dev))
model.train(); model.global_step = 0
model.make_optimizer(max_steps=49)
signal, labels = bench.synthetic_batch(w, batch, dev, 1234)
rng = np.random.RandomState(7)
fn = [model.training_step]
seen = []
side = torch.cuda.Stream(device=dev)
if side is not None:
    side.wait_stream(torch.cuda.current_stream(dev))
    torch.cuda.set_stream(side)                 # everything below -- eager steps, recording, replays -- on ONE non-default stream
for step in range(steps):
    if step == 5:
        fn[0] = CapturedTrainingStep(model, signal, labels)
    model.global_step += 1; make_step(model.scheduler, step=model.global_step)
    perm = torch.from_numpy(rng.permutation(batch)).to(dev)
    take = torch.from_numpy(rng.uniform(size=batch) < w["mixup"]).to(dev)
    partner = torch.where(take, perm, torch.arange(batch, device=dev))
    t = w["samples"]
    mixed, y = F.mixup_batch(signal.squeeze(-1), signal.squeeze(-1)[partner].contiguous(), [t] * batch, [t] * batch, [0] * batch,
                             rng.uniform(0.4, 0.6, size=batch), labels, labels[partner].contiguous())
    out = fn[0](mixed.unsqueeze(-1), y)
    replay = step >= 5
    if smode == 1 or (smode == 2 and not replay) or (smode == 3 and replay):
        torch.cuda.synchronize()
    if reads == "all" or (reads == "eager_only" and not replay) or (reads == "replay_only" and replay):
        seen.append((float(out[2].detach()), float(out[0].detach().abs().max())))
    elif reads == "loss":
        seen.append((float(out[2].detach()), 0.0))
    out = None
torch.cuda.synchronize()
final = fn[0].outputs
print(reads, "sync" if sync else "nosync", "final loss %.3f max|logit| %.1f" % (float(final[2].detach()), float(final[0].detach().abs().max())))
