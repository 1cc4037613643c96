"""Development: which Python lines of a cfg-2 training step launch ATen fill / copy kernels (torch profiler with stacks)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from freesound_classification_amd import functional as F
from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
from freesound_classification_amd.ops.training import make_step
from torch.profiler import profile, ProfilerActivity

w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
if w.get("arith"):
    F.set_conv_arith(w["arith"])
dev = torch.device("cuda:0")
torch.manual_seed(42)
from freesound_classification_amd.networks.classifiers import HierarchicalCNNClassificationModel
cls = HierarchicalCNNClassificationModel if w.get("dims") == 1 else TwoDimensionalCNNClassificationModel
model = cls(bench.make_experiment(w), device=str(dev))
model.train(); model.global_step = 0
model.make_optimizer(max_steps=40)
signal, labels = bench.synthetic_batch(w, w["batch"], dev, 1234)
for _ in range(3):
    model.global_step += 1; make_step(model.scheduler, step=model.global_step)
    model.training_step(signal, labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    model.global_step += 1; make_step(model.scheduler, step=model.global_step)
    model.training_step(signal, labels)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::ones_like", "aten::clone", "aten::contiguous", "aten::add_", "aten::mul_"):
        frames = [f for f in ev.stack if "freesound_classification_amd" in f or "bench.py" in f]
        where = frames[0].strip() if frames else (ev.stack[0].strip() if ev.stack else "?")
        cnt[(ev.name, where[-110:], str(ev.input_shapes)[:60])] += 1
for (name, where, shp), k in cnt.most_common(45):
    print("%3d %-16s %-60s %s" % (k, name, shp, where))
