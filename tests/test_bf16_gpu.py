"""BASELINE.json configs[2]: the 1-d raw-STFT path in bf16.  Arithmetic mode 1 of the conv kernels rounds every operand
to ONE bf16 value and multiplies on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (no limb split); weights, BatchNorm
statistics, activations in HBM and the optimizer state stay fp32 (mixed precision with fp32 masters).

* kernel level: against an fp64 convolution of the bf16-ROUNDED operands -- what an exact bf16-input / fp32-accumulate
  unit would produce -- to fp32-accumulation accuracy (this pins "bf16 products, fp32 sums", not merely "close");
* model level, against the fp32 CPU oracle on the 1-d hierarchical model (20 convolutions) and the 2-d model (12): the
  error budget of plain bf16 is set by the format, not by the kernel -- each operand carries a relative rounding error
  of 2^-9, every BatchNorm renormalises, so after L conv layers an O(1) activation carries ~sqrt(2 L) * 2^-9 ~= 1.2e-2
  of error (L = 20), and the largest of 32 x 80 logits sits at ~4 sigma: rms <= 2.5e-2, max <= 1e-1 (measured: 1.5e-2 /
  6.9e-2); the numbers are written to the report."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs an MI355X", allow_module_level=True)

import torch.nn.functional as TF  # noqa: E402

from freesound_classification_amd import functional as F  # noqa: E402
from freesound_classification_amd.networks.classifiers import (  # noqa: E402
    HierarchicalCNNClassificationModel, TwoDimensionalCNNClassificationModel)
from freesound_classification_amd.networks.losses import lsep_loss  # noqa: E402
from freesound_classification_amd.ops.utils import lwlrap  # noqa: E402
from oracle import ref_torch as oref  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.fixture
def bf16():
    mode0 = F.get_conv_arith()
    F.set_conv_arith("bf16")
    yield
    F.set_conv_arith(mode0)


def r16(t):
    return t.bfloat16().double()


@pytest.mark.parametrize("case", [(4, 129, 64, 1, 431, 1, 3), (3, 80, 100, 1, 215, 1, 3), (2, 100, 100, 1, 300, 1, 1),
                                  (2, 100, 150, 16, 43, 3, 3), (2, 150, 150, 16, 27, 1, 1), (16, 337, 100, 4, 13, 3, 3)])
def test_bf16_conv_is_bf16_products_with_fp32_accumulation(case, bf16):
    n, cin, cout, h, w, kh, kw = case
    torch.manual_seed(sum(case))
    pad = (kh // 2, kw // 2)
    x = torch.randn(n, cin, h, w)
    wt = torch.randn(cout, cin, kh, kw) / (cin * kh * kw) ** 0.5
    b = torch.randn(cout)
    gy = torch.randn(n, cout, h, w)
    d = F._desc(n, cin, cout, h, w, kh, kw)
    names = [F.plan_name(d, m) for m in (0, 1, 2)]
    # the bf16 matrix-core kernels: conv_fwd_x3_kernel<.., 1> (conv.hip), or -- 1-d rows with few positions -- conv_s1d.hip
    assert (names[0].startswith("conv_fwd_x3_kernel") and names[0].endswith(",1>")) or names[0].startswith("conv_s1d_fwd_kernel"), names
    if h == 1 and n * w <= 4096 or (h == 1 and w >= 13 and n * w <= 32768):
        assert names[0].startswith("conv_s1d_fwd_kernel") and names[2].startswith("conv_s1d_wgrad_kernel"), names
    y = F.conv_forward(x.to(DEV), wt.to(DEV), b.to(DEV)).cpu().double()
    dx = F.conv_dgrad(gy.to(DEV), wt.to(DEV), x.shape).cpu().double()
    dw = F.conv_wgrad(x.to(DEV), gy.to(DEV), wt.shape).cpu().double()
    y_ref = TF.conv2d(r16(x), r16(wt), b.double(), padding=pad)
    dx_ref = torch.nn.grad.conv2d_input(x.shape, r16(wt), r16(gy), padding=pad)
    dw_ref = torch.nn.grad.conv2d_weight(r16(x), wt.shape, r16(gy), padding=pad)
    eps = 2.0 ** -23
    assert float((y - y_ref).abs().max()) < 16 * eps * float(y_ref.abs().max()) + 1e-6
    assert float((dx - dx_ref).abs().max()) < 16 * eps * float(dx_ref.abs().max()) + 1e-6
    if names[2].startswith("conv_wgrad_x3_kernel") or names[2].startswith("conv_s1d_wgrad_kernel"):
        assert names[2].endswith(",1>") or names[2].startswith("conv_s1d_wgrad_kernel"), names
        assert float((dw - dw_ref).abs().max()) < 64 * eps * float(dw_ref.abs().max()) + 1e-5
    else:          # shapes the planner keeps on the native fp32 wgrad kernel: fp32 products
        dw32 = torch.nn.grad.conv2d_weight(x.double(), wt.shape, gy.double(), padding=pad)
        assert float((dw - dw32).abs().max()) < 64 * eps * float(dw32.abs().max()) + 1e-5
    # and it is NOT fp32 arithmetic: the bf16 rounding of the operands is visible against the unrounded product
    y32 = TF.conv2d(x.double(), wt.double(), b.double(), padding=pad)
    assert float((y - y32).abs().max()) > 50 * float((y - y_ref).abs().max())


class NS(dict):
    __getattr__ = dict.__getitem__


def _exp(features, blocks, base, growth, input_dim):
    return NS(config=NS(
        network=NS(num_conv_blocks=blocks, start_deep_supervision_on=1, conv_base_depth=base, growth_rate=growth,
                   output_dropout=0.0, aggregation_type="max"),
        data=NS(features=features, _input_dim=input_dim, _n_classes=80),
        train=NS(accumulation_steps=1, optimizer="adam", learning_rate=1e-3, weight_decay=0.0,
                 scheduler="1cycle_0.0001_0.005")))


@pytest.mark.parametrize("kind", ["1d", "2d"])
def test_bf16_model_within_the_stated_tolerance_of_the_fp32_oracle(kind, bf16):
    """Train-mode forward on a 32-clip batch: logits rms <= 2.5e-2 / max <= 1e-1 absolute; gradients: cosine similarity
    with the fp32 oracle's > 0.9 (bf16 roundings flip max-pool winners: measured 0.935 / 0.963); eval-mode predictions after the one BN update: probabilities <= 2.5e-2, lwlrap <= 1e-2."""
    torch.manual_seed(12)
    if kind == "1d":
        exp = _exp("stft_256_128", 5, 64, 1.5, 129)
        m = HierarchicalCNNClassificationModel(exp, device="cuda:0")
        ref = oref.TagCNN1d("stft_256_128", 5, 64, 1.5, 1, 80, input_dim=129)
        signal = 0.1 * torch.randn(32, 44100, 1)
    else:
        exp = _exp("mel_1024_512_64", 3, 64, 1.5, 64)
        m = TwoDimensionalCNNClassificationModel(exp, device="cuda:0")
        ref = oref.TagCNN2d("mel_1024_512_64", 3, 64, 1.5, 1, 80)
        signal = 0.1 * torch.randn(32, 66000, 1)
    ref.load_state_dict({k: v.cpu() for k, v in m.state_dict().items()})
    labels = torch.zeros(32, 80)
    labels[torch.arange(32), torch.randint(0, 80, (32,))] = 1.0
    used = F.plan_name(F._desc(32, 64, 96, 1 if kind == "1d" else 16, 172 if kind == "1d" else 32, 1 if kind == "1d" else 3, 3), 0)
    assert used.endswith(",1>") or used.startswith("conv_s1d_fwd_kernel"), used          # (a bf16 matrix-core kernel)
    ref.train()
    m.train()
    rl = ref(signal)["class_logits"]
    oref.lsep(rl, labels, average=False).mean().backward()
    ml = m(signal.to(DEV))["class_logits"]
    F.mean(lsep_loss(ml, labels.to(DEV), average=False)).backward()
    dl = (ml.detach().cpu() - rl.detach()).abs()
    assert float(dl.max()) < 1e-1 and float(dl.pow(2).mean().sqrt()) < 2.5e-2, (float(dl.max()), float(dl.pow(2).mean().sqrt()))
    rg = dict(ref.named_parameters())
    num = sum(float((p.grad.cpu().double() * rg[k].grad.double()).sum()) for k, p in m.named_parameters())
    den = (sum(float(p.grad.double().pow(2).sum()) for p in m.parameters()) *
           sum(float(p.grad.double().pow(2).sum()) for p in ref.parameters())) ** 0.5
    assert num / den > 0.9, num / den
    ref.eval()
    m.eval()
    with torch.no_grad():
        pe, re_ = torch.sigmoid(m(signal.to(DEV))["class_logits"]).cpu().numpy(), torch.sigmoid(ref(signal)["class_logits"]).numpy()
    d_lw = abs(lwlrap(labels.numpy(), pe) - lwlrap(labels.numpy(), re_))
    import os
    rep = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "cfg2_layer_parity.txt")
    os.makedirs(os.path.dirname(rep), exist_ok=True)
    with open(rep, "a") as f:
        f.write("bf16 %s model vs fp32 oracle: logits max %.2e rms %.2e, grad cosine %.6f, eval probs max %.2e, lwlrap diff %.2e\n" % (
            kind, float(dl.max()), float(dl.pow(2).mean().sqrt()), num / den, float(np.abs(pe - re_).max()), d_lw))
    assert np.abs(pe - re_).max() < 2.5e-2
    assert d_lw < 1e-2
