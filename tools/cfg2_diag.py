"""Development diagnostic: per-parameter gradient difference product vs CPU oracle at cfg-2 width (GPU)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from freesound_classification_amd import functional as F
from freesound_classification_amd.networks.classifiers import TwoDimensionalCNNClassificationModel
from freesound_classification_amd.networks.losses import lsep_loss
from oracle import ref_torch as oref
from test_oracle_cpu import cfg2_golden_inputs

class NS(dict):
    __getattr__ = dict.__getitem__

blocks = int(os.environ.get("BLOCKS", "6"))
base = int(os.environ.get("BASE", "100"))
exp = NS(config=NS(network=NS(num_conv_blocks=blocks, start_deep_supervision_on=1, conv_base_depth=base, growth_rate=1.5,
                              output_dropout=0.0, aggregation_type="max"),
                   data=NS(features="mel_2048_1024_128", _input_dim=128, _n_classes=80),
                   train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0, scheduler="1cycle_0.0001_0.005")))
g = dict(np.load(os.path.join(ROOT, "tests/golden/g12_cfg2_step.npz")))
torch.manual_seed(2024)
m = TwoDimensionalCNNClassificationModel(exp, device="cuda:0")
if os.environ.get("ARITH"):
    F.set_conv_arith(os.environ["ARITH"])
signal, labels = cfg2_golden_inputs(g)
if os.environ.get("NOTAIL"):
    gen = torch.Generator().manual_seed(4321)
    signal = 0.1 * torch.randn(4, 441000, 1, generator=gen)
ref = oref.TagCNN2d("mel_2048_1024_128", blocks, base, 1.5, 1, 80)
ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
ref.train(); m.train()
if os.environ.get("FEATS"):
    with torch.no_grad():
        feats = m.features(signal.cuda()).cpu()
    print("feature diff vs oracle front-end", float((feats - ref.front_end(signal)).abs().max()))
    rl = ref.output_transform(ref.trunk(feats))
else:
    rl = ref(signal)["class_logits"]
oref.lsep(rl, labels, average=False).mean().backward()
ml = m(signal.cuda())["class_logits"]
F.mean(lsep_loss(ml, labels.cuda(), average=False)).backward()
print("logits diff", float((ml.cpu() - rl).abs().max()))
rg = dict(ref.named_parameters())
rows = []
for k, p in m.named_parameters():
    d = (p.grad.cpu().double() - rg[k].grad.double()).abs()
    scale = max(1.0, float(rg[k].grad.abs().max()))
    rows.append((float(d.max()) / scale, float(d.max()), scale, k))
for r in rows[:14]:
    print("%.2e  %.2e  %.2e  %s" % r)
num = sum(float((p.grad.cpu().double() - rg[k].grad.double()).pow(2).sum()) for k, p in m.named_parameters())
den = sum(float(rg[k].grad.double().pow(2).sum()) for k, p in m.named_parameters())
print("global relative L2 of the gradient difference: %.3e" % ((num / den) ** 0.5))
print("worst:")
for r in sorted(rows, reverse=True)[:8]:
    print("%.2e  %.2e  %.2e  %s" % r)
