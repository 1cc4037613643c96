"""RNN aggregation head (reference networks/classifiers.py:514-522, 592-597 and, 1-d, :137-145, 202-207):
mean over frequency -> LayerNorm -> bidirectional GRU(128) final states, forward and backward (BPTT) on the GPU against
torch's CPU LayerNorm / GRU, and the full 2-d model with aggregation_type="rnn" against the reference's fixture g14."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd.networks.classifiers import (  # noqa: E402
    HierarchicalCNNClassificationModel, TwoDimensionalCNNClassificationModel)
from freesound_classification_amd.networks.losses import lsep_loss  # noqa: E402
from oracle import ref_torch as oref  # noqa: E402
from test_oracle_cpu import check_rnn_golden  # noqa: E402

DEV = "cuda:0"


def maxdiff(a, b):
    return float((a.detach().cpu().double() - b.detach().double()).abs().max())


@pytest.mark.parametrize("shape", [(3, 12, 8, 23), (5, 150, 4, 107), (128, 37, 1, 9), (2, 759, 2, 6)])
def test_rnn_head_against_torch(shape):
    n, c, h, w = shape
    torch.manual_seed(sum(shape))
    rnn = torch.nn.Sequential(torch.nn.LayerNorm((c,)), torch.nn.GRU(c, 128, batch_first=True, bidirectional=True))
    with torch.no_grad():
        rnn[0].weight.uniform_(0.5, 1.5)
        rnn[0].bias.normal_(0, 0.2)
    x = torch.randn(n, c, h, w, requires_grad=True)
    _, state = rnn(torch.mean(x, 2).permute(0, 2, 1))
    feat = state.permute(1, 0, 2).contiguous().view(n, -1)
    gy = torch.randn_like(feat)
    feat.backward(gy)
    import copy
    drnn = copy.deepcopy(rnn).to(DEV)
    for p in drnn.parameters():
        p.grad = None
    xd = x.detach().to(DEV).requires_grad_()
    fd = F.rnn_head(xd, drnn)
    assert fd.shape == (n, 256)
    assert maxdiff(fd, feat) < 1e-5
    fd.backward(gy.to(DEV))
    assert maxdiff(xd.grad, x.grad) < 1e-5
    ref = dict(rnn.named_parameters())
    for k, p in drnn.named_parameters():
        scale = max(1.0, float(ref[k].grad.abs().max()))
        assert maxdiff(p.grad, ref[k].grad) < 2e-5 * scale, k
    with torch.no_grad():                     # inference path: no gate buffers
        assert maxdiff(F.rnn_head(x.detach().to(DEV), drnn), feat) < 1e-5


class NS(dict):
    __getattr__ = dict.__getitem__


def _exp(features, blocks, base, growth, input_dim):
    return NS(config=NS(
        network=NS(num_conv_blocks=blocks, start_deep_supervision_on=1, conv_base_depth=base, growth_rate=growth,
                   output_dropout=0.0, aggregation_type="rnn"),
        data=NS(features=features, _input_dim=input_dim, _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0,
                 scheduler="1cycle_0.0001_0.005")))


def test_rnn_model_against_reference_golden(golden):
    g = golden("g14_rnn_head.npz")
    torch.manual_seed(int(g["seed"]))
    m = TwoDimensionalCNNClassificationModel(_exp("mel_1024_512_64", 2, 8, 1.5, 64), device=DEV)
    assert [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()] == golden("g14_state_keys.json")
    for k, v in m.state_dict().items():           # same seed, same registration order: the reference's parameters
        if ("init_sum." + k) in g:
            assert abs(float(v.double().sum()) - float(g["init_sum." + k])) < 1e-9 * max(1.0, float(g["init_abs." + k])), k
    signal, labels = torch.from_numpy(g["signal"]).to(DEV), torch.from_numpy(g["labels"]).to(DEV)
    m.train()
    logits = m(signal)["class_logits"]
    per = lsep_loss(logits, labels, average=False)
    F.mean(per).backward()
    m.eval()
    with torch.no_grad():
        ev = m(signal)["class_logits"]
    check_rnn_golden(g, [(k, p.grad.cpu().numpy()) for k, p in m.named_parameters()], logits.detach().cpu().numpy(),
                     per.detach().cpu().numpy(), ev.cpu().numpy(), 1e-3)


def test_rnn_1d_model_and_training_step_against_oracle():
    """The 1-d model's RNN heads (H == 1: the mean over frequency is the identity) and a full optimizer step."""
    torch.manual_seed(8)
    exp = _exp("stft_256_128", 3, 12, 1.5, 129)
    m = HierarchicalCNNClassificationModel(exp, device=DEV)
    ref = oref.TagCNN1d("stft_256_128", 3, 12, 1.5, 1, 80, input_dim=129, aggregation_type="rnn")
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    signal = 0.1 * torch.randn(6, 9000, 1)
    labels = torch.zeros(6, 80)
    labels[torch.arange(6), torch.randint(0, 80, (6,))] = 1.0
    ref.train()
    rl = ref(signal)["class_logits"]
    oref.lsep(rl, labels, average=False).mean().backward()
    m.train()
    m.make_optimizer(max_steps=10)
    logits, per, loss = m.training_step(signal.to(DEV), labels.to(DEV), step_optimizer=False)
    assert maxdiff(logits, rl) < 1e-3
    rg = dict(ref.named_parameters())
    for k, p in m.named_parameters():
        assert maxdiff(p.grad, rg[k].grad) < 1e-3, k
    m.optimizer.step()
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in m.parameters())
